// engine.cpp -- host side of libconnect6_amd.so: engine object, weight loading, the round loop
// (select -> evaluate -> expand/backup -> move), and the extended C ABI of include/connect6_amd.h.
// The reference's orchestration being replaced: AsyncAgent::run (agent/async_agent/mod.rs:100-137),
// Agent::play (agent/agent_impl/mod.rs:121-178) and AlphaZero::next (alphazero_policy/mod.rs:464-489).
#include "../../include/connect6_amd.h"
#include "c6_device.h"
#include "c6_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

struct C6NetTimers;
C6NetTimers *c6_timers_new();
void c6_timers_free(C6NetTimers *t);
void c6_timers_enable(C6NetTimers *t, bool on);
void c6_timers_collect(C6NetTimers *t, double *ms, unsigned long long *launches, double *flops, double *pre_ms, double *exec_flops, double *post_ms,
                       bool reset);

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(x)                                                                                          \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess) return fail(C6_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));    \
    } while (0)

static std::mutex g_launch_mu;
static std::string g_launch_err;
void c6_note_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_launch_mu);
    if (g_launch_err.empty()) g_launch_err = std::string(what) + ": " + hipGetErrorString(e);
}
int c6_launch_status() {
    std::lock_guard<std::mutex> lk(g_launch_mu);
    if (g_launch_err.empty()) return C6_OK;
    const std::string m = "a kernel could not be launched (" + g_launch_err + "): the outputs of this call are not valid";
    g_launch_err.clear();
    return fail(C6_ERR_HIP, m);
}
int c6_set_error(int code, const char *msg) { return fail(code, msg); }   // for the other translation units (c6_engine.h)
extern "C" const char *c6_last_error(void) { return g_err.c_str(); }
extern "C" const char *c6_version(void) { return "connect6_amd 0.1 (gfx950)"; }
extern "C" const char *c6_rng_spec(void) { return C6_RNG_SPEC; }
extern "C" int c6_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int require_gpu(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(C6_ERR_NO_GPU, "libconnect6_amd: no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(C6_ERR_BAD_ARG, "bad device ordinal");
    HIPCHK(hipSetDevice(device));
    return C6_OK;
}

// ---- constant tables ----------------------------------------------------------------------

static uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static void rot_left(std::vector<int16_t> &b, int n) {   // augment.rs:7-15
    std::vector<int16_t> r(b.size());
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) r[(n - j - 1) * n + i] = b[i * n + j];
    b.swap(r);
}
static void rot_right(std::vector<int16_t> &b, int n) {  // augment.rs:17-25
    std::vector<int16_t> r(b.size());
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) r[j * n + (n - i - 1)] = b[i * n + j];
    b.swap(r);
}
static void flip_v(std::vector<int16_t> &b, int n) {     // augment.rs:27-35
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n / 2; j++) std::swap(b[i * n + j], b[i * n + (n - j - 1)]);
}

// aug[k][c]: image k of augment_way8 takes its cell c from board cell aug[k][c] (augment.rs:60-73)
// rec[k][c]: output cell c of recover_way8 takes image k's policy at cell rec[k][c] (augment.rs:75-92)
static void build_maps(int n, std::vector<int16_t> &aug, std::vector<int16_t> &rec) {
    const int cells = n * n;
    aug.assign(8 * cells, 0);
    rec.assign(8 * cells, 0);
    std::vector<int16_t> id(cells);
    for (int c = 0; c < cells; c++) id[c] = (int16_t)c;
    std::vector<int16_t> b = id;
    for (int i = 0; i < 4; i++) {
        rot_left(b, n);
        std::copy(b.begin(), b.end(), aug.begin() + (2 * i) * cells);
        std::vector<int16_t> f = b;
        flip_v(f, n);
        std::copy(f.begin(), f.end(), aug.begin() + (2 * i + 1) * cells);
    }
    for (int i = 0; i < 4; i++) {
        std::vector<int16_t> a = id, f = id;
        flip_v(f, n);
        for (int r = 0; r < i + 1; r++) { rot_right(a, n); rot_right(f, n); }
        std::copy(a.begin(), a.end(), rec.begin() + (2 * i) * cells);
        std::copy(f.begin(), f.end(), rec.begin() + (2 * i + 1) * cells);
    }
}

static void build_zobrist(std::vector<unsigned long long> &z) {
    z.resize(2 * C6_MAXCELLS);
    uint64_t s = 0xA11CE5EEDULL;
    for (auto &v : z) { do { v = splitmix64(s); } while (v == 0); }
}

// ---- device buffer bookkeeping ------------------------------------------------------------

struct DevPool {
    std::vector<void *> ptrs;
    size_t bytes = 0;
    template <typename T> int alloc(T **p, size_t n, bool zero = true) {
        void *q = nullptr;
        const size_t sz = std::max<size_t>(n * sizeof(T), 16);
        hipError_t e = hipMalloc(&q, sz);
        if (e != hipSuccess) return fail(C6_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        if (zero) hipMemset(q, 0, sz);
        ptrs.push_back(q);
        bytes += sz;
        *p = reinterpret_cast<T *>(q);
        return C6_OK;
    }
    void release() {
        for (void *p : ptrs) hipFree(p);
        ptrs.clear();
    }
    void free_one(void *p) {
        auto it = std::find(ptrs.begin(), ptrs.end(), p);
        if (it != ptrs.end()) { ptrs.erase(it); hipFree(p); }
    }
};
// device scratch of one call, freed when it returns
struct TmpDev {
    std::vector<void *> p;
    ~TmpDev() { for (void *q : p) hipFree(q); }
    template <typename T> T *get(size_t n, const T *src = nullptr) {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(n * sizeof(T), 16)) != hipSuccess) return nullptr;
        p.push_back(q);
        if (src) hipMemcpy(q, src, n * sizeof(T), hipMemcpyHostToDevice);
        else hipMemset(q, 0, std::max<size_t>(n * sizeof(T), 16));
        return reinterpret_cast<T *>(q);
    }
};

#define ALLOC(p, n)                                     \
    do {                                                \
        int rc_ = pool.alloc(&(p), (size_t)(n));        \
        if (rc_ != C6_OK) return rc_;                   \
    } while (0)

// A device array whose physical memory is mapped piece by piece behind one address (HIP virtual memory management): the record and
// table pools grow without a copy and without a second allocation living beside the first.  Without VMM support the array is one
// hipMalloc of its initial size and cannot grow.
// Every piece of an array has the SAME size (`unit`, a multiple of the mapping granularity) and sits at a multiple of it: on this runtime
// (ROCm 7.2) hipMemSetAccess refuses a piece whose size differs from its neighbours' at most offsets, and setting the access of a whole
// mapped prefix at once is accepted but leaves part of it unmapped (profiles/r03_vmm_probe.txt: "81 then 567", "mode 1").
struct VmRange {
    char *base = nullptr;
    size_t reserved = 0, mapped = 0, gran = 0, unit = 0;
    bool vmm = false;
    int device = 0;
    std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> parts;      // handle, bytes
    static bool supported(int device) {
        int v = 0;
        return hipDeviceGetAttribute(&v, hipDeviceAttributeVirtualMemoryManagementSupported, device) == hipSuccess && v != 0;
    }
    hipMemAllocationProp prop() const {
        hipMemAllocationProp p = {};
        p.type = hipMemAllocationTypePinned;
        p.location.type = hipMemLocationTypeDevice;
        p.location.id = device;
        return p;
    }
    // mapped in 2 MiB multiples (large pages behind the pools) unless the device asks for more
    static size_t granularity(int dev) {
        hipMemAllocationProp p = {};
        p.type = hipMemAllocationTypePinned;
        p.location.type = hipMemLocationTypeDevice;
        p.location.id = dev;
        size_t g = 0, gmin = 0;
        if (hipMemGetAllocationGranularity(&g, &p, hipMemAllocationGranularityRecommended) != hipSuccess) g = 0;
        if (hipMemGetAllocationGranularity(&gmin, &p, hipMemAllocationGranularityMinimum) != hipSuccess) gmin = 0;
        (void)hipGetLastError();      // a query this runtime does not answer is not an error of the engine
        return std::max<size_t>((size_t)2 << 20, std::max(g, gmin));
    }
    int reserve(size_t max_bytes, size_t unit_bytes, int dev, bool use_vmm) {
        device = dev;
        vmm = use_vmm;
        unit = unit_bytes;
        if (!vmm) { reserved = max_bytes; return C6_OK; }
        gran = granularity(dev);
        if (unit == 0 || unit % gran) return fail(C6_ERR_BAD_ARG, "pool unit is not a multiple of the mapping granularity");
        reserved = (max_bytes + unit - 1) / unit * unit;
        void *b = nullptr;
        hipError_t e = hipMemAddressReserve(&b, reserved, 0, nullptr, 0);
        if (e != hipSuccess) return fail(C6_ERR_HIP, std::string("hipMemAddressReserve: ") + hipGetErrorString(e));
        base = reinterpret_cast<char *>(b);
        return C6_OK;
    }
    // physical memory behind [0, bytes): bytes must be a multiple of gran (VMM) -- or the one and only allocation (no VMM)
    int map_to(size_t bytes) {
        if (bytes <= mapped) return C6_OK;
        if (!vmm) {
            if (base) return fail(C6_ERR_OVERFLOW, "this device has no virtual memory management: the pools cannot grow");
            void *b = nullptr;
            hipError_t e = hipMalloc(&b, bytes);
            if (e != hipSuccess) return fail(C6_ERR_HIP, std::string("hipMalloc (pool): ") + hipGetErrorString(e));
            base = reinterpret_cast<char *>(b);
            mapped = bytes;
            return C6_OK;
        }
        if (bytes > reserved || bytes % unit) return fail(C6_ERR_OVERFLOW, "pool beyond its reserved address range (or not in whole units)");
        const hipMemAllocationProp p = prop();
        while (mapped < bytes) {
            hipMemGenericAllocationHandle_t h;
            hipError_t e = hipMemCreate(&h, unit, &p, 0);
            if (e != hipSuccess) { (void)hipGetLastError(); return fail(C6_ERR_OVERFLOW, std::string("hipMemCreate: ") + hipGetErrorString(e)); }
            e = hipMemMap(base + mapped, unit, 0, h, 0);
            if (e != hipSuccess) { hipMemRelease(h); (void)hipGetLastError(); return fail(C6_ERR_HIP, std::string("hipMemMap: ") + hipGetErrorString(e)); }
            hipMemAccessDesc acc = {};
            acc.location = p.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(base + mapped, unit, &acc, 1);
            if (e != hipSuccess) { hipMemUnmap(base + mapped, unit); hipMemRelease(h); (void)hipGetLastError(); return fail(C6_ERR_HIP, std::string("hipMemSetAccess: ") + hipGetErrorString(e)); }
            parts.emplace_back(h, unit);
            mapped += unit;
        }
        return C6_OK;
    }
    void release() {
        if (!base) return;
        if (!vmm) { hipFree(base); base = nullptr; return; }
        size_t off = 0;
        for (auto &pr : parts) { hipMemUnmap(base + off, pr.second); hipMemRelease(pr.first); off += pr.second; }
        parts.clear();
        hipMemAddressFree(base, reserved);
        base = nullptr;
    }
};

// ---- self-describing weight blobs -------------------------------------------------------------------
// header = 8 floats in front of the payload: magic "C6WB" (bit pattern), layout version, N, C, num_block, payload floats (as two
// 24-bit halves) and a checksum of the payload (FNV-1a of the u32 bit patterns folded to 24 bits, exactly representable)
#define C6_BLOB_MAGIC 0x42573643u       /* "C6WB" little endian */
#define C6_BLOB_VERSION 1
static float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static uint32_t blob_checksum(const float *p, size_t n) {      // FNV-1a over the bit patterns: order-sensitive, folded to 24 bits
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) h = (h ^ f_bits(p[i])) * 0x100000001b3ull;
    return (uint32_t)((h ^ (h >> 24) ^ (h >> 48)) & 0xffffffu);
}
extern "C" size_t c6_net_blob_header_floats(void) { return C6_BLOB_HEADER_FLOATS; }
extern "C" int c6_net_blob_write_header(float *dst, size_t nfloats, int n, int C, int nb) {
    if (!dst || nfloats != c6_net_blob_floats(n, C, nb) + C6_BLOB_HEADER_FLOATS) return fail(C6_ERR_BAD_ARG, "blob buffer has the wrong size for this geometry");
    const size_t payload = nfloats - C6_BLOB_HEADER_FLOATS;
    dst[0] = bits_f(C6_BLOB_MAGIC); dst[1] = (float)C6_BLOB_VERSION; dst[2] = (float)n; dst[3] = (float)C; dst[4] = (float)nb;
    dst[5] = (float)(payload >> 24); dst[6] = (float)(payload & 0xffffffu);
    dst[7] = (float)blob_checksum(dst + C6_BLOB_HEADER_FLOATS, payload);
    return C6_OK;
}
extern "C" int c6_net_blob_check_header(const float *blob, size_t nfloats, int n, int C, int nb) {
    if (!blob || nfloats < C6_BLOB_HEADER_FLOATS) return fail(C6_ERR_BAD_ARG, "blob too short for a header");
    if (f_bits(blob[0]) != C6_BLOB_MAGIC) return fail(C6_ERR_BAD_ARG, "weight blob: bad magic (not written by save_blob / c6_net_blob_write_header)");
    if (blob[1] != (float)C6_BLOB_VERSION) return fail(C6_ERR_BAD_ARG, "weight blob: unknown layout version");
    if (blob[2] != (float)n || blob[3] != (float)C || blob[4] != (float)nb) return fail(C6_ERR_BAD_ARG, "weight blob was written for another board size / channel count / block count");
    // the size halves are converted only once they are known to be 24-bit integers (a float -> integer cast of a NaN / negative / huge
    // value of a corrupted header is undefined behaviour)
    for (int k = 5; k <= 7; k++)
        if (!(blob[k] >= 0.f && blob[k] < 16777216.f) || blob[k] != std::floor(blob[k])) return fail(C6_ERR_BAD_ARG, "weight blob: corrupted header (size / checksum fields)");
    const size_t payload = ((size_t)blob[5] << 24) | (size_t)blob[6];
    if (payload != nfloats - C6_BLOB_HEADER_FLOATS || payload != c6_net_blob_floats(n, C, nb)) return fail(C6_ERR_BAD_ARG, "weight blob: payload size does not match its header");
    if (blob[7] != (float)blob_checksum(blob + C6_BLOB_HEADER_FLOATS, payload)) return fail(C6_ERR_BAD_ARG, "weight blob: checksum mismatch (corrupted or re-ordered payload)");
    return C6_OK;
}

// ---- resident net ---------------------------------------------------------------------------

struct NetHost {
    C6NetDev dev{};
    DevPool pool;
    bool loaded = false;
};

extern "C" size_t c6_net_blob_floats(int n, int C, int nb) {
    const size_t n2 = (size_t)n * n, c = (size_t)C;
    size_t t = 0;
    t += c * 2 * 9 + c + 4 * c;
    t += (size_t)(nb - 1) * 2 * (c * c * 9 + c + 4 * c);
    t += 2 * c + 2 + 8 + n2 * 2 * n2 + n2;
    t += c + 1 + 4 * n2 + 256 * n2 + 256 + 256 + 1;
    return t;
}

template <typename T> static int upload(DevPool &pool, const std::vector<T> &h, const T **d) {
    T *p = nullptr;
    int rc = pool.alloc(&p, h.size(), false);
    if (rc != C6_OK) return rc;
    HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *d = p;
    return C6_OK;
}

// which kernel serves a C6_PREC_* (see include/connect6_amd.h)
static int conv_variant(int precision) {
    switch (precision) {
    case C6_PREC_FP32_DIRECT: return C6_CONV_F32N;
    case C6_PREC_BF16X3: return C6_CONV_B3S;
    case C6_PREC_BF16: return C6_CONV_BF16;
    case C6_PREC_F16F8: return C6_CONV_M8;
    default: return C6_CONV_WINO;      // C6_PREC_DEFAULT = C6_PREC_FP32: the reference's arithmetic (model.hpp:52-95 is f32)
    }
}

// Weight blob: flat f32, the tensors of WeightedPolicy (AlphaZero/cpp_weighted/model.hpp:10-50) in FORWARD order with each
// BatchNorm's running statistics beside its affine parameters -- NOT libtorch's registration order (model.hpp registers conv, norm,
// the two heads and only then the tower, and named_buffers() lists running statistics separately).  Per tensor PyTorch's own layout:
//   stem    conv.weight [C,2,3,3], conv.bias [C], norm.{weight, bias, running_mean, running_var} [C] each
//   tower   for i in 0 .. num_block-2, for j in 0, 1:  conv{i}_{j}.weight [C,C,3,3], .bias [C], norm{i}_{j}.{weight, bias, running_mean, running_var}
//   policy  policy_conv.weight [2,C,1,1], .bias [2], policy_norm.{weight, bias, running_mean, running_var} [2] each,
//           policy_linear.weight [N^2, 2N^2], .bias [N^2]
//   value   value_conv.weight [1,C,1,1], .bias [1], value_norm.{weight, bias, running_mean, running_var} [N^2] each,
//           value_linear1.weight [256, N^2], .bias [256], value_linear2.weight [1,256], .bias [1]
// alphazero_connect6_amd.blob_from_state_dict builds it from a state_dict (by NAME, so a permuted blob cannot happen there);
// c6_net_blob_check_header validates the optional 8-float header that save_blob writes (magic, version, N, C, blocks, count,
// checksum) so that a blob of the right size but the wrong layout or geometry is rejected before it is played with.
static int net_load(NetHost &nh, int n, int C, int nb, const float *blob, size_t nfloats, int precision = C6_PREC_DEFAULT) {
    if (C != 64 && C != 128 && C != 256) return fail(C6_ERR_BAD_ARG, "net_channels must be 64, 128 or 256");
    if (nb < 1) return fail(C6_ERR_BAD_ARG, "net_blocks must be >= 1");
    if (precision != C6_PREC_DEFAULT && precision != C6_PREC_FP32 && precision != C6_PREC_FP32_DIRECT && precision != C6_PREC_BF16X3 &&
        precision != C6_PREC_BF16 && precision != C6_PREC_F16F8)
        return fail(C6_ERR_BAD_ARG, "net_precision must be a C6_PREC_* value");
    if (nfloats == c6_net_blob_floats(n, C, nb) + C6_BLOB_HEADER_FLOATS) {      // self-describing blob (save_blob): check and skip the header
        int rc = c6_net_blob_check_header(blob, nfloats, n, C, nb);
        if (rc != C6_OK) return rc;
        blob += C6_BLOB_HEADER_FLOATS;
        nfloats -= C6_BLOB_HEADER_FLOATS;
    }
    if (nfloats != c6_net_blob_floats(n, C, nb)) return fail(C6_ERR_BAD_ARG, "weight blob has the wrong size");
    for (size_t i = 0; i < nfloats; i++)
        if (!std::isfinite(blob[i])) return fail(C6_ERR_BAD_ARG, "weight blob holds a non-finite value");
    nh.pool.release();
    nh.loaded = false;
    const int cells = n * n;
    const float eps = 1e-5f;
    const float *p = blob;
    auto take = [&](size_t k) { const float *r = p; p += k; return r; };
    C6NetDev &d = nh.dev;
    d = C6NetDev{};
    d.N = n; d.cells = cells; d.C = C; d.blocks = nb;
    d.variant = conv_variant(precision);
    {
        hipDeviceProp_t prop;
        int dev = 0;
        hipGetDevice(&dev);
        d.num_cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    const bool guard = d.variant == C6_CONV_M8;      // the reduced-range arithmetic: validate what it is given (c6_engine.h, C6_M8_RANGE_*)

    {   // stem
        const float *w = take((size_t)C * 2 * 9), *b = take(C), *g = take(C), *be = take(C), *m = take(C), *v = take(C);
        std::vector<float> sw((size_t)9 * 2 * C), sb(C);
        double bound = 0.0;
        for (int co = 0; co < C; co++) {
            const float s = g[co] / std::sqrt(v[co] + eps);
            sb[co] = (b[co] - m[co]) * s + be[co];
            double l1 = std::fabs((double)sb[co]);
            for (int ch = 0; ch < 2; ch++)
                for (int t = 0; t < 9; t++) {
                    sw[(size_t)(t * 2 + ch) * C + co] = w[((size_t)co * 2 + ch) * 9 + t] * s;
                    l1 += std::fabs((double)sw[(size_t)(t * 2 + ch) * C + co]);
                }
            bound = std::max(bound, l1);      // |stem output| <= sum |w| + |b|: its inputs are player / cells in {-1, 0, +1}
        }
        if (guard && !(bound <= C6_M8_RANGE_HI))
            return fail(C6_ERR_RANGE, "C6_PREC_F16F8: the stem's activations can exceed 3584 with these weights; use C6_PREC_FP32 or C6_PREC_BF16X3");
        if (guard && bound < C6_M8_RANGE_LO)
            return fail(C6_ERR_RANGE, "C6_PREC_F16F8: the stem's activations cannot reach 2^-3, the low end of this arithmetic's validated range; use C6_PREC_FP32 or C6_PREC_BF16X3");
        int rc;
        if ((rc = upload(nh.pool, sw, &d.stem_w)) || (rc = upload(nh.pool, sb, &d.stem_b))) return rc;
    }
    {   // tower: only the weight image of the selected kernel is built and kept
        const int L = 2 * (nb - 1), Lm = std::max(L, 1);
        std::vector<float> cb((size_t)Lm * C), folded((size_t)C * C * 9);
        std::vector<float> cw5, cww;
        std::vector<unsigned short> cw7;
        std::vector<unsigned char> cw9;
        d.wino6 = d.variant == C6_CONV_WINO && c6_w6_available(C, n, nb);
        const size_t w5 = c6_conv_packed_f32n_floats(C), ww = d.wino6 ? c6_w6_packed_floats() : c6_conv_packed_wino_floats(C), w7 = c6_conv_packed_b3s_units(C),
                     w9 = c6_conv_packed_m8_bytes(C);
        if (d.variant == C6_CONV_F32N) cw5.resize((size_t)Lm * w5);
        else if (d.variant == C6_CONV_WINO) cww.resize((size_t)Lm * ww);
        else if (d.variant == C6_CONV_M8) cw9.resize((size_t)Lm * w9);
        else cw7.resize((size_t)Lm * w7);
        for (int l = 0; l < L; l++) {
            const float *w = take((size_t)C * C * 9), *b = take(C), *g = take(C), *be = take(C), *m = take(C), *v = take(C);
            float wmax = 0.f, cmin = INFINITY;
            for (int co = 0; co < C; co++) {
                const float s = g[co] / std::sqrt(v[co] + eps);
                cb[(size_t)l * C + co] = (b[co] - m[co]) * s + be[co];
                float cmax = 0.f;
                for (size_t k = 0; k < (size_t)C * 9; k++) {
                    const float f = w[(size_t)co * C * 9 + k] * s;
                    folded[(size_t)co * C * 9 + k] = f;
                    cmax = std::max(cmax, std::fabs(f));
                }
                if (!std::isfinite(cmax) || !std::isfinite(cb[(size_t)l * C + co])) return fail(C6_ERR_BAD_ARG, "BatchNorm folding of a tower layer overflowed FP32");
                wmax = std::max(wmax, cmax);
                if (cmax > 0.f) cmin = std::min(cmin, cmax);
            }
            if (guard) {
                char msg[200];
                if (!(wmax < 65504.f)) {
                    snprintf(msg, sizeof msg, "C6_PREC_F16F8: folded weights of tower layer %d reach %.3g, outside fp16; use C6_PREC_FP32 or C6_PREC_BF16X3", l, wmax);
                    return fail(C6_ERR_RANGE, msg);
                }
                if (wmax < 1.0f / 512.0f) {
                    snprintf(msg, sizeof msg, "C6_PREC_F16F8: folded weights of tower layer %d stay below 2^-9 (largest %.3g): their fp16 parts are subnormal; "
                             "use C6_PREC_FP32 or C6_PREC_BF16X3", l, wmax);
                    return fail(C6_ERR_RANGE, msg);
                }
                if (cmin < wmax * (1.0f / 1024.0f)) {
                    snprintf(msg, sizeof msg, "C6_PREC_F16F8: output channels of tower layer %d span %.3g .. %.3g (more than 2^10): the layer's fp8 weight "
                             "planes share one shift; use C6_PREC_FP32 or C6_PREC_BF16X3", l, cmin, wmax);
                    return fail(C6_ERR_RANGE, msg);
                }
            }
            if (d.variant == C6_CONV_F32N) c6_pack_conv_weights_f32n(C, folded.data(), cw5.data() + (size_t)l * w5);
            else if (d.wino6) c6_w6_pack_weights(folded.data(), cww.data() + (size_t)l * ww);
            else if (d.variant == C6_CONV_WINO) c6_pack_conv_weights_wino(C, folded.data(), cww.data() + (size_t)l * ww);
            else if (d.variant == C6_CONV_M8) c6_pack_conv_weights_m8(C, folded.data(), cw9.data() + (size_t)l * w9);
            else c6_pack_conv_weights_b3s(C, folded.data(), cw7.data() + (size_t)l * w7);
        }
        int rc;
        if ((!cw5.empty() && (rc = upload(nh.pool, cw5, &d.conv_w5))) || (!cww.empty() && (rc = upload(nh.pool, cww, &d.conv_ww))) ||
            (!cw7.empty() && (rc = upload(nh.pool, cw7, &d.conv_w7))) || (!cw9.empty() && (rc = upload(nh.pool, cw9, &d.conv_w9))) ||
            (rc = upload(nh.pool, cb, &d.conv_b)))
            return rc;
        if (guard) {
            unsigned int *fl = nullptr;
            if ((rc = nh.pool.alloc(&fl, 1))) return rc;
            d.range_flag = fl;
        }
    }
    {   // policy head
        const float *w = take((size_t)2 * C), *b = take(2), *bn = take(8);
        std::vector<float> pw((size_t)2 * C), pb(2);
        for (int ch = 0; ch < 2; ch++) {
            const float s = bn[ch] / std::sqrt(bn[6 + ch] + eps);
            pb[ch] = (b[ch] - bn[4 + ch]) * s + bn[2 + ch];
            for (int c = 0; c < C; c++) pw[(size_t)ch * C + c] = w[(size_t)ch * C + c] * s;
        }
        const float *lw = take((size_t)cells * 2 * cells), *lb = take(cells);
        std::vector<float> wT((size_t)2 * cells * cells), lbv(lb, lb + cells);
        for (int o = 0; o < cells; o++)
            for (int k = 0; k < 2 * cells; k++) wT[(size_t)k * cells + o] = lw[(size_t)o * 2 * cells + k];
        int rc;
        if ((rc = upload(nh.pool, pw, &d.pc_w)) || (rc = upload(nh.pool, pb, &d.pc_b)) || (rc = upload(nh.pool, wT, &d.pl_wT)) ||
            (rc = upload(nh.pool, lbv, &d.pl_b)))
            return rc;
    }
    {   // value head
        const float *w = take(C), *b = take(1), *bn = take((size_t)4 * cells);
        std::vector<float> vw(w, w + C), sc(cells), sh(cells);
        d.vc_b = b[0];
        for (int i = 0; i < cells; i++) {
            sc[i] = bn[i] / std::sqrt(bn[3 * cells + i] + eps);
            sh[i] = bn[cells + i] - bn[2 * cells + i] * sc[i];
        }
        const float *w1 = take((size_t)256 * cells), *b1 = take(256), *w2 = take(256), *b2 = take(1);
        std::vector<float> w1T((size_t)cells * 256), b1v(b1, b1 + 256), w2v(w2, w2 + 256);
        for (int h = 0; h < 256; h++)
            for (int k = 0; k < cells; k++) w1T[(size_t)k * 256 + h] = w1[(size_t)h * cells + k];
        d.v2_b = b2[0];
        int rc;
        std::vector<float> hw((size_t)3 * C);      // [pc_w row 0 | pc_w row 1 | vc_w]
        if (hipMemcpy(hw.data(), d.pc_w, (size_t)2 * C * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return fail(C6_ERR_HIP, "head weights");
        std::copy(vw.begin(), vw.end(), hw.begin() + (size_t)2 * C);
        if ((rc = upload(nh.pool, hw, &d.head_w))) return rc;
        if ((rc = upload(nh.pool, vw, &d.vc_w)) || (rc = upload(nh.pool, sc, &d.vn_scale)) || (rc = upload(nh.pool, sh, &d.vn_shift)) ||
            (rc = upload(nh.pool, w1T, &d.v1_wT)) || (rc = upload(nh.pool, b1v, &d.v1_b)) || (rc = upload(nh.pool, w2v, &d.v2_w)))
            return rc;
    }
    nh.loaded = true;
    return C6_OK;
}

// ---- engine -----------------------------------------------------------------------------------

struct PhaseTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<hipEvent_t> pool;
    double ms = 0;
    unsigned long long n = 0;
    hipEvent_t get() {
        hipEvent_t e;
        if (!pool.empty()) { e = pool.back(); pool.pop_back(); } else hipEventCreate(&e);
        return e;
    }
    void collect() {
        for (auto &p : ev) {
            float m = 0;
            hipEventSynchronize(p.second);
            hipEventElapsedTime(&m, p.first, p.second);
            ms += m;
            n++;
            pool.push_back(p.first);
            pool.push_back(p.second);
        }
        ev.clear();
    }
    void destroy() {
        collect();
        for (auto e : pool) hipEventDestroy(e);
        pool.clear();
    }
};

// Host staging of the callback evaluator (8 G boards down, 8 G policies + values up, every simulation round): page-locked, so that the copies are
// DMA transfers at the link's rate instead of the runtime's pageable path through a bounce buffer (r5).  Grows, never shrinks; pageable if locking fails.
template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t cap = 0, n = 0;
    bool pinned = false;
    T *data() { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    void release() {
        if (p) { if (pinned) hipHostFree(p); else free(p); }
        p = nullptr; cap = n = 0;
    }
    bool resize(size_t want) {
        if (want > cap) {
            release();
            void *q = nullptr;
            if (hipHostMalloc(&q, want * sizeof(T), hipHostMallocDefault) == hipSuccess) pinned = true;
            else { (void)hipGetLastError(); q = malloc(want * sizeof(T)); pinned = false; }
            if (!q) return false;
            memset(q, 0, want * sizeof(T));
            p = static_cast<T *>(q);
            cap = want;
        }
        n = want;
        return true;
    }
    ~PinnedBuf() { release(); }
};

struct c6_engine {
    c6_config cfg{};
    C6Dev d{};
    DevPool pool;
    NetHost net;
    hipStream_t stream = nullptr;
    C6NetTimers *timers = nullptr;
    bool profiling = false;
    PhaseTimer t_select, t_expand, t_eval, t_move;
    c6_callback cb = nullptr;
    // net scratch
    float *act0 = nullptr, *act1 = nullptr, *act2 = nullptr, *ph = nullptr, *vh = nullptr;
    float *vbuf = nullptr;         // Winograd-transformed activations of one chunk (C6_PREC_FP32)
    int chunk_leaves = 0;
    // callback scratch
    float *img_boards = nullptr;   // device [G][8][cells]
    PinnedBuf<float> h_img, h_pol, h_val;
    std::vector<float> h_cbp, h_cbv;
    std::vector<int8_t> h_kind, h_turn;
    std::vector<C6GameHdr> h_hdr;      // host copy of the game headers (statistics, states)
    // owned tuple buffer
    uint8_t *own_tuple = nullptr;
    uint32_t *own_tuple_count = nullptr;
    size_t tuple_head = 0;            // records at the front of the tuple buffer already handed out by c6_engine_fetch_tuples (partial fetches)
    uint64_t rounds = 0;
    int round_in_move = 0;
    unsigned long long fetched = 0;   // entries of the finished-games queue already handed to the caller (c6_engine_fetch_games)
    bool said_finq = false;
    // the pools of record pages and table chunks (c6_engine.h): four arrays that grow by mapping more memory behind them
    VmRange vm_prior, vm_stat, vm_ht, vm_tag;
    size_t pages_max = 0, chunks_max = 0;      // what the rings / page_free / the address ranges are sized for
    size_t page_quantum = 1, chunk_quantum = 1; // growth steps keep every array's mapped size a multiple of the mapping granularity
    int hmax = 0;                              // absolute per-game table size
    int pages_per_game0 = 0;
    uint32_t pool_grows = 0;                   // times memory was added to the pools
    uint32_t cap_raises = 0;                   // times the per-game caps were raised
    bool grow_failed = false;      // the last c6_engine_maintain found parked games it could not help (caps at their maximum / the pools cannot grow)
    bool said_full = false;
};

extern "C" void c6_default_config(c6_config *cfg) {
    memset(cfg, 0, sizeof *cfg);
    cfg->board_size = 15;            // libconnect6/src/lib.rs:21
    cfg->num_games = 1;
    cfg->num_simulation = 800;       // HyperParameter::default, alphazero_policy/mod.rs:98-108
    cfg->epsilon = 0.25f;
    cfg->dirichlet_alpha = 0.03;
    cfg->c_puct = 1.0f;
    cfg->evaluator = C6_EVAL_RANDOM;
    cfg->net_channels = 256;         // AlphaZero/cpp_weighted/model.hpp:12
    cfg->net_blocks = 10;            // AlphaZero/cpp_weighted/main.cpp:12
}

extern "C" size_t c6_tuple_stride(int n) {
    const size_t cells = (size_t)n * n;
    return (cells + (cells & 1) + 2 * cells + 2 + 2 + 3) & ~(size_t)3;   // state | pad | pi u16 | pos u16 | turn | z
}

static size_t gcd_sz(size_t a, size_t b) { while (b) { const size_t t = a % b; a = b; b = t; } return a; }
static size_t lcm_sz(size_t a, size_t b) { return a / gcd_sz(a, b) * b; }
static int pow2_at_least(long v, int lo) { int h = lo; while (h < v) h <<= 1; return h; }

// pools up to `pages` record pages and `chunks` table chunks mapped (both multiples of their quanta or the final size); new pages'
// records are marked free (level -1 = 0xffff: a byte fill), new ids go onto the rings.  Stream idle.
static int pools_map(c6_engine *e, size_t pages, size_t chunks) {
    C6Dev &d = e->d;
    const size_t cells = d.cells, old_p = d.pages_total, old_c = d.chunks_total;
    int rc;
    if (pages > old_p) {
        if ((rc = e->vm_prior.map_to(pages * C6_PAGE_RECS * cells * 4)) || (rc = e->vm_stat.map_to(pages * C6_PAGE_RECS * sizeof(C6RecStat)))) return rc;
        d.rec_prior = reinterpret_cast<float *>(e->vm_prior.base);
        d.rec_stat = reinterpret_cast<C6RecStat *>(e->vm_stat.base);
        HIPCHK(hipMemset(d.rec_stat + old_p * C6_PAGE_RECS, 0xff, (pages - old_p) * C6_PAGE_RECS * sizeof(C6RecStat)));
    }
    if (chunks > old_c) {
        if ((rc = e->vm_ht.map_to(chunks * C6_CHUNK_SLOTS * sizeof(C6HtEntry))) || (rc = e->vm_tag.map_to(chunks * C6_CHUNK_SLOTS))) return rc;
        d.ht = reinterpret_cast<C6HtEntry *>(e->vm_ht.base);
        d.ht_tag = reinterpret_cast<uint8_t *>(e->vm_tag.base);
    }
    unsigned long long ctr[C6_POOL_N];
    HIPCHK(hipMemcpy(ctr, d.pool_ctr, sizeof ctr, hipMemcpyDeviceToHost));
    auto push = [&](uint32_t *ring, uint32_t mask, unsigned long long &tail, size_t from, size_t to) -> int {
        std::vector<uint32_t> ids(to - from);
        for (size_t i = 0; i < ids.size(); i++) ids[i] = (uint32_t)(from + i);
        size_t done = 0;
        while (done < ids.size()) {      // the ring wraps
            const size_t at = (size_t)(tail & mask), n = std::min(ids.size() - done, (size_t)mask + 1 - at);
            HIPCHK(hipMemcpy(ring + at, ids.data() + done, n * 4, hipMemcpyHostToDevice));
            done += n;
            tail += n;
        }
        return C6_OK;
    };
    if (pages > old_p && (rc = push(d.page_ring, d.page_ring_mask, ctr[C6_POOL_PAGE_TAIL], old_p, pages))) return rc;
    if (chunks > old_c && (rc = push(d.chunk_ring, d.chunk_ring_mask, ctr[C6_POOL_CHUNK_TAIL], old_c, chunks))) return rc;
    ctr[C6_POOL_PAGE_LIMIT] = ctr[C6_POOL_PAGE_TAIL];
    ctr[C6_POOL_CHUNK_LIMIT] = ctr[C6_POOL_CHUNK_TAIL];
    ctr[C6_POOL_PAGE_AVAIL] = ctr[C6_POOL_PAGE_TAIL] - ctr[C6_POOL_PAGE_HEAD];
    ctr[C6_POOL_CHUNK_AVAIL] = ctr[C6_POOL_CHUNK_TAIL] - ctr[C6_POOL_CHUNK_HEAD];
    if (old_p == 0) ctr[C6_POOL_PAGE_LOW] = pages;
    HIPCHK(hipMemcpy(d.pool_ctr, ctr, sizeof ctr, hipMemcpyHostToDevice));
    d.pages_total = (unsigned)pages;
    d.chunks_total = (unsigned)chunks;
    return C6_OK;
}

static int engine_alloc(c6_engine *e) {
    const c6_config &c = e->cfg;
    C6Dev &d = e->d;
    DevPool &pool = e->pool;
    const int N = c.board_size, cells = N * N, G = c.num_games;
    d.N = N; d.cells = cells; d.cellsP = (cells + 15) & ~15; d.G = G; d.Gpad = (G + 3) & ~3;
    d.S = c.num_simulation;
    d.maxd = d.cellsP;      // a select path adds a stone per step
    d.eps = c.epsilon; d.cpuct = c.c_puct; d.alpha = c.dirichlet_alpha;
    d.gam_d = (c.dirichlet_alpha + 1.0) - 1.0 / 3.0;       // Marsaglia-Tsang constants at shape alpha + 1
    d.gam_c = 1.0 / std::sqrt(9.0 * d.gam_d);
    d.seed_lo = (uint32_t)c.seed; d.seed_hi = (uint32_t)(c.seed >> 32);
    d.uid_base = c.game_uid_base;
    d.uid_stride = c.game_uid_stride ? c.game_uid_stride : (uint32_t)G;
    d.recycle = c.recycle;
    d.stress_every = c.stress_every > 0 ? c.stress_every : 1;

    // ---- the pools (c6_engine.h).  records_per_game is the AVERAGE the pool is sized for at creation (whole pages, and never less
    // than one move's searches + a page); a single game may hold up to max_records_per_game; the pool grows by mapping more memory.
    const int R0 = c.records_per_game > 0 ? c.records_per_game : 4 * c.num_simulation + 64;
    if (R0 > C6_MAX_RECORDS) return fail(C6_ERR_BAD_ARG, "records_per_game too large (max 65000)");
    const int move_pages = (d.S + C6_PAGE_RECS - 1) / C6_PAGE_RECS + 1;
    if (move_pages * C6_PAGE_RECS > C6_MAX_RECORDS) return fail(C6_ERR_BAD_ARG, "num_simulation too large for 16-bit record indices");
    e->pages_per_game0 = std::max((R0 + C6_PAGE_RECS - 1) / C6_PAGE_RECS, move_pages);
    int maxR = c.max_records_per_game > 0 ? std::min(C6_MAX_RECORDS, c.max_records_per_game) : C6_MAX_RECORDS;
    d.maxpages = std::max((maxR + C6_PAGE_RECS - 1) / C6_PAGE_RECS, move_pages);
    e->hmax = pow2_at_least(2L * ((long)d.maxpages * C6_PAGE_RECS + d.S), C6_CHUNK_SLOTS);
    d.maxchunks = e->hmax >> C6_CHUNK_SHIFT;
    d.pages_cap = std::min(d.maxpages, 2 * e->pages_per_game0);
    d.hcap = std::min(e->hmax, pow2_at_least(2L * ((long)d.pages_cap * C6_PAGE_RECS + d.S), C6_CHUNK_SLOTS));
    d.abandon_parked = 0;
    const size_t page_bytes = (size_t)C6_PAGE_RECS * ((size_t)cells * 4 + sizeof(C6RecStat)), chunk_bytes = (size_t)C6_CHUNK_SLOTS * (sizeof(C6HtEntry) + 1);
    const size_t pages0 = (size_t)G * e->pages_per_game0;
    const size_t chunks0 = (size_t)G * std::max<size_t>(pow2_at_least(2L * R0, C6_CHUNK_SLOTS) >> C6_CHUNK_SHIFT, pow2_at_least(2L * d.S, C6_CHUNK_SLOTS) >> C6_CHUNK_SHIFT);
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    // C6_NO_VMM=1: the pools as plain allocations of their initial size (what a device without virtual memory management gets): test hook
    static const bool no_vmm = [] { const char *v = getenv("C6_NO_VMM"); return v && v[0] == '1'; }();
    const bool vmm = !no_vmm && VmRange::supported(c.device);
    e->pages_max = vmm ? std::max(pages0, std::min((size_t)G * d.maxpages, (size_t)(0.9 * (double)tot / (double)page_bytes))) : pages0;
    e->chunks_max = vmm ? std::max(chunks0, std::min((size_t)G * d.maxchunks, e->pages_max / 4 + chunks0)) : chunks0;      // a table slot costs 17 B, a record >= 356
    auto round_up = [](size_t v, size_t q) { return (v + q - 1) / q * q; };
    if (vmm) {      // quanta: every array's mapped size stays a multiple of the mapping granularity
        const size_t g1 = VmRange::granularity(c.device);
        e->page_quantum = lcm_sz(g1 / gcd_sz(g1, (size_t)C6_PAGE_RECS * cells * 4), g1 / gcd_sz(g1, (size_t)C6_PAGE_RECS * sizeof(C6RecStat)));
        e->chunk_quantum = lcm_sz(g1 / gcd_sz(g1, (size_t)C6_CHUNK_SLOTS * sizeof(C6HtEntry)), g1 / gcd_sz(g1, (size_t)C6_CHUNK_SLOTS));
    }
    const size_t pages_init = round_up(pages0, e->page_quantum), chunks_init = round_up(chunks0, e->chunk_quantum);
    e->pages_max = round_up(std::max(e->pages_max, pages_init), e->page_quantum);
    e->chunks_max = round_up(std::max(e->chunks_max, chunks_init), e->chunk_quantum);
    if (e->pages_max > 0xfffffff0ull || e->chunks_max > 0xfffffff0ull) return fail(C6_ERR_BAD_ARG, "pool too large");
    int rc;
    // the pools grow in whole quanta; one quantum of an array = one mapped piece (VmRange)
    if ((rc = e->vm_prior.reserve(e->pages_max * C6_PAGE_RECS * cells * 4, e->page_quantum * C6_PAGE_RECS * cells * 4, c.device, vmm)) ||
        (rc = e->vm_stat.reserve(e->pages_max * C6_PAGE_RECS * sizeof(C6RecStat), e->page_quantum * C6_PAGE_RECS * sizeof(C6RecStat), c.device, vmm)) ||
        (rc = e->vm_ht.reserve(e->chunks_max * C6_CHUNK_SLOTS * sizeof(C6HtEntry), e->chunk_quantum * C6_CHUNK_SLOTS * sizeof(C6HtEntry), c.device, vmm)) ||
        (rc = e->vm_tag.reserve(e->chunks_max * C6_CHUNK_SLOTS, e->chunk_quantum * C6_CHUNK_SLOTS, c.device, vmm)))
        return rc;
    d.page_ring_mask = (unsigned)pow2_at_least((long)e->pages_max, 64) - 1u;
    d.chunk_ring_mask = (unsigned)pow2_at_least((long)e->chunks_max, 64) - 1u;
    ALLOC(d.page_free, e->pages_max);
    ALLOC(d.page_ring, (size_t)d.page_ring_mask + 1); ALLOC(d.chunk_ring, (size_t)d.chunk_ring_mask + 1);
    ALLOC(d.pool_ctr, C6_POOL_N);
    ALLOC(d.page_table, (size_t)G * d.maxpages); ALLOC(d.chunk_table, (size_t)G * d.maxchunks);
    HIPCHK(hipMemset(d.page_table, 0xff, (size_t)G * d.maxpages * 4));      // C6_NOPAGE
    d.pages_total = d.chunks_total = 0;
    if ((rc = pools_map(e, pages_init, chunks_init))) return rc;

    ALLOC(d.board_soa, (size_t)cells * d.Gpad);
    ALLOC(d.turn, d.Gpad); ALLOC(d.remain, d.Gpad); ALLOC(d.winner, d.Gpad); ALLOC(d.err, d.Gpad);
    ALLOC(d.hdr, G);
    ALLOC(d.root_board, (size_t)G * d.cellsP);
    ALLOC(d.moves, (size_t)G * cells); ALLOC(d.hist_pi, (size_t)G * cells * cells);
    ALLOC(d.rootc, (size_t)G * d.cellsP);
    ALLOC(d.leaf_board, (size_t)G * d.cellsP); ALLOC(d.path_rec, (size_t)G * d.maxd);
    ALLOC(d.eval_prior, (size_t)G * cells);
    ALLOC(d.move_pos, d.Gpad); ALLOC(d.last_move, d.Gpad); ALLOC(d.move_turn, G);
    d.finq_cap = std::max(4 * G, 1024);
    ALLOC(d.fin_winner, d.finq_cap); ALLOC(d.fin_len, d.finq_cap); ALLOC(d.fin_uid, d.finq_cap);
    ALLOC(d.fin_moves, (size_t)d.finq_cap * cells);
    ALLOC(d.counters, C6_CNT_N);
    d.tuple_stride = (int)c6_tuple_stride(N);

    std::vector<unsigned long long> zob;
    build_zobrist(zob);
    std::vector<int16_t> aug, rec;
    build_maps(N, aug, rec);
    if ((rc = upload(pool, zob, &d.zob)) || (rc = upload(pool, aug, &d.aug_map)) || (rc = upload(pool, rec, &d.rec_map))) return rc;

    if (c.evaluator == C6_EVAL_NET || c.evaluator == C6_EVAL_CALLBACK) {
        ALLOC(d.img_val, (size_t)G * 8);
        ALLOC(d.img_pol, (size_t)G * 8 * cells);
    }
    if (c.evaluator == C6_EVAL_CALLBACK) ALLOC(e->img_boards, (size_t)G * 8 * cells);
    if (c.evaluator == C6_EVAL_NET) {
        int chunk_boards = c.net_chunk_boards > 0 ? c.net_chunk_boards : 32768;
        int chunk_leaves = std::max(1, chunk_boards / 8);
        chunk_leaves = std::min(chunk_leaves, G);
        e->chunk_leaves = chunk_leaves;
        const size_t nb = (size_t)chunk_leaves * 8, act = nb * cells * c.net_channels + C6_ACT_SLACK;
        ALLOC(e->act0, act); ALLOC(e->act1, act); ALLOC(e->act2, act);
        ALLOC(e->ph, nb * 2 * cells); ALLOC(e->vh, nb * cells);
        if (conv_variant(c.net_precision) == C6_CONV_WINO) {
            // the largest buffer of the net (16/4 of an activation): not zero-filled, every float the GEMM reads is written first
            const size_t vf = c6_w6_available(c.net_channels, N, c.net_blocks) ? c6_w6_vbuf_floats(N, (int)nb) : c6_conv_wino_vbuf_floats(c.net_channels, N, (int)nb);
            int rc_ = pool.alloc(&e->vbuf, vf, false);
            if (rc_ != C6_OK) return rc_;
        }
    }
    return C6_OK;
}

extern "C" int c6_engine_create(const c6_config *cfg, c6_engine **out) {
    if (!cfg || !out) return fail(C6_ERR_BAD_ARG, "null argument");
    const int N = cfg->board_size;
    if (N != 9 && N != 15 && N != 19) return fail(C6_ERR_BAD_ARG, "board_size must be 9, 15 or 19");
    if (cfg->num_games < 1 || cfg->num_simulation < 1) return fail(C6_ERR_BAD_ARG, "num_games and num_simulation must be >= 1");
    if (cfg->evaluator < C6_EVAL_RANDOM || cfg->evaluator > C6_EVAL_STRESS) return fail(C6_ERR_BAD_ARG, "bad evaluator kind");
    if (!(cfg->dirichlet_alpha > 0.0)) return fail(C6_ERR_BAD_ARG, "dirichlet_alpha must be > 0 (Dirichlet::new_with_param asserts it)");
    int rc = require_gpu(cfg->device);
    if (rc != C6_OK) return rc;
    c6_engine *e = new c6_engine();
    e->cfg = *cfg;
    if (hipStreamCreate(&e->stream) != hipSuccess) { delete e; return fail(C6_ERR_HIP, "hipStreamCreate failed"); }
    e->timers = c6_timers_new();
    rc = engine_alloc(e);
    if (rc != C6_OK) { c6_engine_destroy(e); return rc; }
    c6_launch_reset_games(e->d, cfg->stagger, e->stream);
    if (hipStreamSynchronize(e->stream) != hipSuccess) { c6_engine_destroy(e); return fail(C6_ERR_HIP, "engine reset failed"); }
    *out = e;
    return C6_OK;
}

extern "C" void c6_engine_destroy(c6_engine *e) {
    if (!e) return;
    if (e->stream) hipStreamSynchronize(e->stream);
    e->t_select.destroy(); e->t_expand.destroy(); e->t_eval.destroy(); e->t_move.destroy();
    c6_timers_free(e->timers);
    e->vm_prior.release(); e->vm_stat.release(); e->vm_ht.release(); e->vm_tag.release();
    e->pool.release();
    e->net.pool.release();
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int c6_engine_load_weights(c6_engine *e, const float *blob, size_t nfloats) {
    if (!e || !blob) return fail(C6_ERR_BAD_ARG, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));      // rounds in flight still read the weights that are about to be replaced
    return net_load(e->net, e->cfg.board_size, e->cfg.net_channels, e->cfg.net_blocks, blob, nfloats, e->cfg.net_precision);
}

extern "C" int c6_engine_set_callback(c6_engine *e, c6_callback cb) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    e->cb = cb;
    return C6_OK;
}

extern "C" int c6_engine_set_tuple_buffer(c6_engine *e, void *dev_buf, size_t capacity_records, uint32_t *count_dev) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    HIPCHK(hipStreamSynchronize(e->stream));
    e->d.tuple_buf = reinterpret_cast<uint8_t *>(dev_buf);
    e->d.tuple_cap = capacity_records;
    e->d.tuple_count = count_dev;
    e->tuple_head = 0;
    return C6_OK;
}

extern "C" int c6_engine_swap_tuple_buffer(c6_engine *e, void *dev_buf, size_t capacity_records, uint32_t *count_dev) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    e->d.tuple_buf = reinterpret_cast<uint8_t *>(dev_buf);      // C6Dev travels by value with every launch: no drain needed
    e->d.tuple_cap = capacity_records;
    e->d.tuple_count = count_dev;
    e->tuple_head = 0;
    return C6_OK;
}

extern "C" void *c6_engine_stream(c6_engine *e) { return e ? (void *)e->stream : nullptr; }

static int ensure_own_tuples(c6_engine *e, size_t cap) {
    if (e->d.tuple_buf) return C6_OK;
    DevPool &pool = e->pool;
    ALLOC(e->own_tuple, cap * (size_t)e->d.tuple_stride);
    ALLOC(e->own_tuple_count, 1);
    e->d.tuple_buf = e->own_tuple;
    e->d.tuple_cap = cap;
    e->d.tuple_count = e->own_tuple_count;
    return C6_OK;
}

struct Scoped {
    PhaseTimer *t;
    hipStream_t s;
    hipEvent_t e0 = nullptr;
    Scoped(PhaseTimer *t_, hipStream_t s_, bool on) : t(on ? t_ : nullptr), s(s_) {
        if (t) { e0 = t->get(); hipEventRecord(e0, s); }
    }
    ~Scoped() {
        if (t) { hipEvent_t e1 = t->get(); hipEventRecord(e1, s); t->ev.emplace_back(e0, e1); }
    }
};

static int evaluate_callback(c6_engine *e) {
    C6Dev &d = e->d;
    const int G = d.G, cells = d.cells;
    if (!e->cb) return fail(C6_ERR_BAD_ARG, "evaluator is C6_EVAL_CALLBACK but no callback is set");
    c6_launch_augment(d, e->img_boards, e->stream);
    e->h_kind.resize(G); e->h_turn.resize(G);
    if (!e->h_img.resize((size_t)G * 8 * cells) || !e->h_val.resize((size_t)G * 8)) return fail(C6_ERR_HIP, "callback staging: out of host memory");
    e->h_hdr.resize(G);
    HIPCHK(hipMemcpyAsync(e->h_hdr.data(), d.hdr, (size_t)G * sizeof(C6GameHdr), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->h_img.data(), e->img_boards, e->h_img.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int g = 0; g < G; g++) { e->h_kind[g] = e->h_hdr[g].leaf_kind; e->h_turn[g] = e->h_hdr[g].leaf_turn; }
    // the callback takes ONE `player` for all boards (cppeval/mod.rs:8-13): batch per side to move.  Games of one call start together, so as a rule
    // every game waits for an evaluation for the same side: the callback then works on the staging buffers in place (no gather / scatter copies)
    bool uniform = G > 0;
    for (int g = 0; g < G && uniform; g++) uniform = e->h_kind[g] == C6_LEAF_EVAL && e->h_turn[g] == e->h_turn[0];
    if (uniform) {
        memset(e->h_val.data(), 0, e->h_val.size() * sizeof(float));      // values zero-filled, cppeval/mod.rs:34
        e->cb(e->h_turn[0], e->h_val.data(), e->h_img.data(), G * 8);     // (boards in, policies out: the same buffer, as the reference passes it)
        HIPCHK(hipMemcpyAsync(d.img_pol, e->h_img.data(), e->h_img.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(d.img_val, e->h_val.data(), e->h_val.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
        c6_launch_recover(d, e->stream);
        return C6_OK;
    }
    // mixed sides (or games without a leaf to evaluate): gather per side, scatter into a second staging buffer -- page-locked only now (the
    // common path above never touches it; 47 MB at 4096 games of 19x19)
    if (!e->h_pol.resize((size_t)G * 8 * cells)) return fail(C6_ERR_HIP, "callback staging: out of host memory");
    for (int player = -1; player <= 1; player += 2) {
        std::vector<int> idx;
        for (int g = 0; g < G; g++)
            if (e->h_kind[g] == C6_LEAF_EVAL && e->h_turn[g] == player) idx.push_back(g);
        if (idx.empty()) continue;
        const size_t per = (size_t)8 * cells;
        e->h_cbp.resize(idx.size() * per);
        e->h_cbv.assign(idx.size() * 8, 0.f);                 // values zero-filled, cppeval/mod.rs:34
        for (size_t k = 0; k < idx.size(); k++)
            memcpy(&e->h_cbp[k * per], &e->h_img[(size_t)idx[k] * per], per * sizeof(float));
        e->cb(player, e->h_cbv.data(), e->h_cbp.data(), (int)(idx.size() * 8));
        for (size_t k = 0; k < idx.size(); k++) {
            memcpy(&e->h_pol[(size_t)idx[k] * per], &e->h_cbp[k * per], per * sizeof(float));
            memcpy(&e->h_val[(size_t)idx[k] * 8], &e->h_cbv[k * 8], 8 * sizeof(float));
        }
    }
    HIPCHK(hipMemcpyAsync(d.img_pol, e->h_pol.data(), e->h_pol.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d.img_val, e->h_val.data(), e->h_val.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    c6_launch_recover(d, e->stream);
    return C6_OK;
}

static int evaluate(c6_engine *e) {
    C6Dev &d = e->d;
    switch (e->cfg.evaluator) {
    case C6_EVAL_NET: {
        if (!e->net.loaded) return fail(C6_ERR_BAD_ARG, "evaluator is C6_EVAL_NET but no weights are loaded");
        for (int l0 = 0; l0 < d.G; l0 += e->chunk_leaves) {
            const int nl = std::min(e->chunk_leaves, d.G - l0);
            c6_net_forward_dev(e->net.dev, nl, 8, d.leaf_board + (size_t)l0 * d.cellsP, d.cellsP, &d.hdr[l0].leaf_turn, (int)sizeof(C6GameHdr), d.aug_map,
                               e->act0, e->act1, e->act2, e->vbuf, e->ph, e->vh, d.img_pol + (size_t)l0 * 8 * d.cells,
                               d.img_val + (size_t)l0 * 8, e->stream, e->timers);
        }
        c6_launch_recover(d, e->stream);
        return C6_OK;
    }
    case C6_EVAL_CALLBACK: return evaluate_callback(e);
    default: return fail(C6_ERR_BAD_ARG, "bad evaluator");
    }
}

#ifndef C6_SEARCH_ROUNDS_PER_LAUNCH
#define C6_SEARCH_ROUNDS_PER_LAUNCH 64      // r4: 16 -> 64 (config 5, same box: 8 rounds per launch 0.146 ms per round, 16 0.139, 32 0.134, 64 0.130: a wave's start -- game
#endif                                      // registers, tag row, page table into LDS -- is paid once per launch; parked games ask again at every launch's head)
extern "C" int c6_engine_step(c6_engine *e, int rounds) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    C6Dev &d = e->d;
    const int ev = e->cfg.evaluator;
    if (ev == C6_EVAL_NET && !e->net.loaded) return fail(C6_ERR_BAD_ARG, "evaluator is C6_EVAL_NET but no weights are loaded");
    bool moved = false;
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)d.last_move, (int)0xFFFFFFFE, (size_t)d.Gpad, e->stream));      // -2: "did not move in this call"
    if (ev == C6_EVAL_RANDOM || ev == C6_EVAL_UNIFORM || ev == C6_EVAL_STRESS) {
        // built-in evaluators: whole rounds -- search() and, when its budget is spent, the move itself -- inside one kernel, several
        // rounds of every game per launch (tree_kernels.hip: c6_search_kernel)
        for (int r = 0; r < rounds;) {
            const int k = std::min(rounds - r, C6_SEARCH_ROUNDS_PER_LAUNCH);
            Scoped t(&e->t_select, e->stream, e->profiling);
            c6_launch_search(d, ev == C6_EVAL_RANDOM ? 1 : ev == C6_EVAL_UNIFORM ? 2 : 3, k, e->stream);
            r += k;
            e->rounds += k;
        }
        HIPCHK(hipGetLastError());
        return c6_launch_status();
    }
    for (int r = 0; r < rounds; r++) {
        { Scoped t(&e->t_select, e->stream, e->profiling); c6_launch_select(d, e->stream); }
        {
            Scoped t(&e->t_eval, e->stream, e->profiling);
            int rc = evaluate(e);
            if (rc != C6_OK) return rc;
        }
        { Scoped t(&e->t_expand, e->stream, e->profiling); c6_launch_expand(d, e->stream); }
        e->rounds++;
        e->round_in_move++;
        // K1 places the stones of the games whose policy() just ran.  It runs EVERY round (35 us beside a net evaluation): a game's
        // move round is a property of the game, not of the host's round counter -- staggered games, games resumed after waiting for
        // pages and slots recycled after an error are all out of step with the aligned ones, and a move_pos published in a round
        // without K1 would be finished by the next round's head with a stale winner / err.
        {
            Scoped t(&e->t_move, e->stream, e->profiling);
            c6_launch_board_step(d.N, d.Gpad, d.board_soa, d.turn, d.remain, d.move_pos, d.winner, d.err, e->stream);
            moved = true;
            if (e->round_in_move >= d.S) e->round_in_move = 0;
        }
    }
    if (moved) {      // callers see every applied move finished (root advanced / game queued); inside a call the next round's head does it
        Scoped t(&e->t_move, e->stream, e->profiling);
        c6_launch_post_move(d, e->stream);
    }
    HIPCHK(hipGetLastError());
    return c6_launch_status();
}

// Stones placed from OUTSIDE the engine's policy (MultiPolicy: another policy plays one colour, multi_policy/mod.rs:61-71;
// cpp_play_with: the human's stones, lib.rs:140-169): pos[g] = cell of game g, or -2 for "no move for this game".  Game::set
// + is_game_end on the GPU, then the same re-rooting as after the engine's own move -- the AlphaZero tree of the other
// colour keeps every node below the new position (the reference's policy is simply not consulted for this stone and
// finds its old nodes again at its next turn).
// pos_dev: int32 [Gpad] on the device (entries beyond G: -2)
static int apply_moves_dev(c6_engine *e, const int32_t *pos_dev) {
    C6Dev &d = e->d;
    HIPCHK(hipMemcpyAsync(d.move_pos, pos_dev, (size_t)d.Gpad * sizeof(int32_t), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d.move_turn, d.turn, (size_t)d.G, hipMemcpyDeviceToDevice, e->stream));   // the mover = the side to move now
    c6_launch_board_step(d.N, d.Gpad, d.board_soa, d.turn, d.remain, d.move_pos, d.winner, d.err, e->stream);
    c6_launch_post_move(d, e->stream);
    HIPCHK(hipGetLastError());
    return C6_OK;
}
extern "C" int c6_engine_apply_moves(c6_engine *e, const int32_t *pos) {
    if (!e || !pos) return fail(C6_ERR_BAD_ARG, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    C6Dev &d = e->d;
    std::vector<int32_t> h((size_t)d.Gpad, -2);
    for (int g = 0; g < d.G; g++) h[g] = pos[g];
    int32_t *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, (size_t)d.Gpad * sizeof(int32_t)));
    hipError_t he = hipMemcpy(tmp, h.data(), (size_t)d.Gpad * sizeof(int32_t), hipMemcpyHostToDevice);
    int rc = he == hipSuccess ? apply_moves_dev(e, tmp) : fail(C6_ERR_HIP, "apply_moves: copy failed");
    hipStreamSynchronize(e->stream);
    hipFree(tmp);
    return rc;
}

// ---- evaluation matches on the device: MultiPolicy(AlphaZero of this engine, opponent) ----------------------------------------------
// policy/multi_policy/mod.rs:61-71 gives each colour its own policy; here the engine's search plays `engine_colour` and the opponent --
// RandomPolicy (policy/random_policy/mod.rs:39-47) or DefaultPolicy (pure MCTS with random playouts, c6_rollout) -- the other one, for
// all games at once, and no board, move or result crosses the host between the first stone and the last: the opponent reads the engine's
// root positions in place, its stones reach K1 device to device (apply_moves_dev), and the host only reads one word per stone (the
// number of games still playing).
__global__ void c6_match_prepare_kernel(C6Dev E, int opp_colour, int8_t *boards, uint8_t *active, int *live) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    const bool playing = E.hdr[g].status == C6_ST_PLAYING;
    active[g] = playing && E.turn[g] == opp_colour;
    if (playing) atomicAdd(live, 1);
    for (int c = 0; c < E.cells; c++) boards[(size_t)g * E.cells + c] = E.root_board[(size_t)g * E.cellsP + c];
}
// RandomPolicy::next: a uniformly random empty cell (`possible.choose(rng)`); the unseeded thread_rng becomes C6RNG tag 8:
// word 0 of Philox(key = (seed_lo, seed_hi + game uid); ctr = (stones on the board, 0, 0, 8)), the k-th empty cell in row-major order with
// k = (w * len) >> 32 (as the DefaultPolicy picks of part 5)
#define C6_TAG_RANDOM_POLICY 8u
__global__ void c6_random_policy_kernel(C6Dev E, uint32_t seed_lo, uint32_t seed_hi, const int8_t *boards, const uint8_t *active, int32_t *pos) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    pos[g] = -1;
    if (!active[g]) return;
    const int8_t *b = boards + (size_t)g * E.cells;
    int len = 0;
    for (int c = 0; c < E.cells; c++) len += b[c] == 0;
    if (len == 0) return;
    const c6_u4 w = c6_philox((uint32_t)E.hdr[g].nstones, 0u, 0u, C6_TAG_RANDOM_POLICY, seed_lo, seed_hi + E.hdr[g].game_uid);
    int k = (int)(((unsigned long long)w.x * (unsigned long long)len) >> 32);
    for (int c = 0; c < E.cells; c++)
        if (b[c] == 0 && k-- == 0) { pos[g] = c; break; }
}
// the opponent's answers as K1 takes them: the cell, -1 (an active game without an empty cell: a draw) or -2 (not this game's turn)
__global__ void c6_match_moves_kernel(int G, int Gpad, const uint8_t *active, const int32_t *pos, int32_t *move) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < Gpad) move[g] = g < G && active[g] ? pos[g] : -2;
}

static int live_games(c6_engine *e, int *live, int *errors, int *overflow);
static int maintain(c6_engine *e);
extern "C" int c6_engine_play_match(c6_engine *e, int engine_colour, int opponent, c6_rollout *r, uint64_t opp_seed, long max_stones) {
    if (!e || (engine_colour != -1 && engine_colour != 1)) return fail(C6_ERR_BAD_ARG, "c6_engine_play_match: engine_colour is -1 (Black) or +1 (White)");
    if (opponent != C6_OPP_RANDOM && opponent != C6_OPP_ROLLOUT) return fail(C6_ERR_BAD_ARG, "c6_engine_play_match: opponent is C6_OPP_RANDOM or C6_OPP_ROLLOUT");
    if (opponent == C6_OPP_ROLLOUT && !r) return fail(C6_ERR_BAD_ARG, "c6_engine_play_match: C6_OPP_ROLLOUT needs a c6_rollout with as many slots as the engine has games");
    if (e->cfg.recycle || e->cfg.stagger) return fail(C6_ERR_BAD_ARG, "c6_engine_play_match: evaluation games are played once, in step (recycle = stagger = 0)");
    HIPCHK(hipSetDevice(e->cfg.device));
    C6Dev &d = e->d;
    TmpDev t;
    int8_t *boards = t.get<int8_t>((size_t)d.G * d.cells);
    uint8_t *active = t.get<uint8_t>(d.G);
    int32_t *pos = t.get<int32_t>(d.Gpad), *move = t.get<int32_t>(d.Gpad);
    int *live = t.get<int>(1);
    if (!boards || !active || !pos || !move || !live) return fail(C6_ERR_HIP, "hipMalloc failed");
    const unsigned gb = (unsigned)((d.Gpad + 255) / 256);
    // Every game is at the same stone of the same turn -- Black 1, then 2 each (game_impl/mod.rs:125-169) -- and the loop below keeps
    // them there: the precondition is CHECKED (no stone placed, Black to move, every game playing), and a game that is
    // parked at a move boundary (C6_ST_OVERFLOW: it waits for pages, chunks or higher caps, tree_kernels.hip reprovision_wave) is
    // served before the next stone is played -- otherwise it would resume a move late and the engine would search for the
    // opponent's colour in that game.
    {
        int live = 0, errs = 0;
        int rc = live_games(e, &live, &errs, nullptr);
        if (rc != C6_OK) return rc;
        std::vector<int8_t> h_turn((size_t)d.G);
        HIPCHK(hipMemcpy(h_turn.data(), d.turn, (size_t)d.G, hipMemcpyDeviceToHost));
        for (int g = 0; g < d.G; g++)
            if (e->h_hdr[g].status != C6_ST_PLAYING || e->h_hdr[g].nstones != 0 || e->h_hdr[g].sim_in_move != 0 || h_turn[g] != C6_BLACK)
                return fail(C6_ERR_BAD_ARG, "c6_engine_play_match: the engine must be at the start of its games (no stone placed, no search made, Black to move); create a fresh engine per match");
    }
    auto serve_parked = [&]() -> int {
        for (int attempt = 0;; attempt++) {
            int live = 0, errs = 0, ov = 0;
            int rc = live_games(e, &live, &errs, &ov);
            if (rc != C6_OK) return rc;
            if (ov == 0) return C6_OK;
            if (attempt >= 8 || e->grow_failed)
                return fail(C6_ERR_OVERFLOW, "c6_engine_play_match: a game's tree outgrew the pools (max_records_per_game / device memory); the match stops with that game unfinished");
            if ((rc = maintain(e)) != C6_OK) return rc;      // raise the caps / map more pool memory
            c6_launch_pool_snapshot(d, e->stream);
            c6_launch_post_move(d, e->stream);              // the round head: parked games ask again, nothing else happens
        }
    };
    int turn = C6_BLACK, remain = 1;
    for (long stone = 0; max_stones <= 0 || stone < max_stones; stone++) {
        {
            int rc = serve_parked();
            if (rc != C6_OK) return rc;
        }
        HIPCHK(hipMemsetAsync(live, 0, sizeof(int), e->stream));
        hipLaunchKernelGGL(c6_match_prepare_kernel, dim3(gb), dim3(256), 0, e->stream, d, -engine_colour, boards, active, live);
        int h_live = 0;
        HIPCHK(hipMemcpyAsync(&h_live, live, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (h_live == 0) break;
        if (turn == engine_colour) {
            int rc = c6_engine_step(e, d.S);      // S searches and the stone of every live game
            if (rc != C6_OK) return rc;
        } else {
            if (opponent == C6_OPP_RANDOM)
                hipLaunchKernelGGL(c6_random_policy_kernel, dim3(gb), dim3(256), 0, e->stream, d, (uint32_t)opp_seed, (uint32_t)(opp_seed >> 32), boards, active, pos);
            else {
                HIPCHK(hipStreamSynchronize(e->stream));
                int rc = c6_rollout_next(r, boards, active, pos, 1);      // device pointers; synchronises its own stream
                if (rc != C6_OK) return fail(rc, "c6_engine_play_match: c6_rollout_next failed (tree full?)");
            }
            hipLaunchKernelGGL(c6_match_moves_kernel, dim3(gb), dim3(256), 0, e->stream, d.G, d.Gpad, active, pos, move);
            int rc = apply_moves_dev(e, move);
            if (rc != C6_OK) return rc;
        }
        if (--remain <= 0) { remain = 2; turn = -turn; }
    }
    {
        int rc = serve_parked();      // the last stone's move boundary as well: no game is left waiting when the call says C6_OK
        if (rc != C6_OK) return rc;
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    return C6_OK;
}

// the cells the engine's own policy played during the last c6_engine_step call: pos[g] = cell, -1 = the policy had no move (the game
// ended as a draw), -2 = game g did not move in that call.  With c6_engine_apply_moves on a second engine this is MultiPolicy for two
// resident nets (net-vs-net evaluation games, policy/multi_policy/mod.rs:61-71).
extern "C" int c6_engine_last_moves(c6_engine *e, int32_t *pos) {
    if (!e || !pos) return fail(C6_ERR_BAD_ARG, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(pos, e->d.last_move, (size_t)e->d.G * sizeof(int32_t), hipMemcpyDeviceToHost));
    return C6_OK;
}

extern "C" int c6_engine_sync(c6_engine *e) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    HIPCHK(hipStreamSynchronize(e->stream));
    return C6_OK;
}

// C6_PREC_F16F8's runtime guard: the sticky device flag of the convolution epilogues / the per-layer sample (c6_engine.h).
// Call with the stream drained.
static int range_status(const NetHost &net) {
    if (!net.loaded || !net.dev.range_flag) return C6_OK;
    unsigned int f = 0;
    HIPCHK(hipMemcpy(&f, net.dev.range_flag, sizeof f, hipMemcpyDeviceToHost));
    if (f & 1u) return fail(C6_ERR_RANGE, "C6_PREC_F16F8: an activation exceeded 3584 (or was not finite): priors since the last check are not the FP32 net's; use C6_PREC_FP32 or C6_PREC_BF16X3");
    if (f & 2u) return fail(C6_ERR_RANGE, "C6_PREC_F16F8: a layer's activations fell below 2^-3, the low end of this arithmetic's validated range; use C6_PREC_FP32 or C6_PREC_BF16X3");
    return C6_OK;
}

static int live_games(c6_engine *e, int *live, int *errors, int *overflow) {
    e->h_hdr.resize(e->d.G);
    HIPCHK(hipMemcpyAsync(e->h_hdr.data(), e->d.hdr, (size_t)e->d.G * sizeof(C6GameHdr), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    int l = 0, er = 0, ov = 0;
    for (const C6GameHdr &h : e->h_hdr) { l += h.status == C6_ST_PLAYING; er += h.status == C6_ST_ERROR; ov += h.status == C6_ST_OVERFLOW; }
    *live = l;
    *errors = er + ov;
    if (overflow) *overflow = ov;
    return range_status(e->net);      // every host-synchronising entry point passes through here
}

// ---- the pools grow ------------------------------------------------------------------------------------------------------
// The reference's HashMap<u64, Node> grows without limit (alphazero_policy/mod.rs:186-190, 360-377).  Here a game takes pages of
// records and chunks of table slots from two pools at every move boundary, as many as its tree needs, and hands back what it no
// longer uses (tree_kernels.hip: reprovision_wave) -- all on the device.  The host's part (c6_engine_maintain, and c6_engine_run
// between its chunks of rounds):
//   * a game at the per-game caps (which size the LDS copies of its page table / tag row) is parked with hdr.park = C6_PARK_CAPS:
//     the caps are doubled, up to max_records_per_game;
//   * a pool that runs low (or a game parked with C6_PARK_POOL) gets more memory MAPPED behind the same addresses (VmRange):
//     nothing is copied, nothing is allocated twice, games in progress are not touched.
// A parked game has lost nothing -- it waits at a move boundary and asks again at the head of every round.
static int grow_pools(c6_engine *e, size_t need_pages, size_t need_chunks) {
    C6Dev &d = e->d;
    if (!e->vm_prior.vmm) return fail(C6_ERR_OVERFLOW, "the pools are plain allocations (no virtual memory management): they cannot grow");
    const bool pages_full = d.pages_total >= e->pages_max, chunks_full = d.chunks_total >= e->chunks_max;
    if ((need_pages == 0 || pages_full) && (need_chunks == 0 || chunks_full))
        return fail(C6_ERR_OVERFLOW, need_pages > 0 ? "record pool is at its maximum" : "table-chunk pool is at its maximum");
    size_t pages = d.pages_total, chunks = d.chunks_total;
    if (need_pages > 0 && !pages_full) {
        size_t add = std::max<size_t>(need_pages, d.pages_total / 4);
        add = std::max(add, std::min<size_t>(e->pages_max, ((size_t)1 << 30) / ((size_t)C6_PAGE_RECS * d.cells * 4)));      // >= 1 GiB of priors a step
        add = (add + e->page_quantum - 1) / e->page_quantum * e->page_quantum;
        pages = std::min(e->pages_max, d.pages_total + add);
        // chunks follow the pages in proportion
        chunks = std::min(e->chunks_max, (size_t)((double)pages / (double)d.pages_total * (double)d.chunks_total));
    }
    // ... and grow on their own when it is table chunks that games wait for (keyed on pages alone, a chunk shortage with the
    // pages at their maximum could never be served)
    if (need_chunks > 0 && !chunks_full) chunks = std::max(chunks, d.chunks_total + std::max<size_t>(need_chunks, d.chunks_total / 4));
    chunks = std::min(e->chunks_max, (chunks + e->chunk_quantum - 1) / e->chunk_quantum * e->chunk_quantum);
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    const size_t page_bytes = (size_t)C6_PAGE_RECS * ((size_t)d.cells * 4 + sizeof(C6RecStat)), chunk_bytes = (size_t)C6_CHUNK_SLOTS * (sizeof(C6HtEntry) + 1);
    const size_t reserve = (size_t)1 << 30;      // left to the rest of the process
    auto fits = [&]() { return (pages - d.pages_total) * page_bytes + (chunks - d.chunks_total) * chunk_bytes + reserve <= fr; };
    while (!fits() && (pages > d.pages_total || chunks > d.chunks_total)) {      // tight memory: halve the step, pages first (a chunk is 17 KB, a page >= 23 KB)
        if (pages > d.pages_total) pages = pages - d.pages_total > e->page_quantum ? d.pages_total + (pages - d.pages_total) / 2 / e->page_quantum * e->page_quantum : d.pages_total;
        else chunks = chunks - d.chunks_total > e->chunk_quantum ? d.chunks_total + (chunks - d.chunks_total) / 2 / e->chunk_quantum * e->chunk_quantum : d.chunks_total;
    }
    if (pages <= d.pages_total && chunks <= d.chunks_total) return fail(C6_ERR_OVERFLOW, "not enough device memory to grow the pools");
    int rc = pools_map(e, pages, chunks);
    if (rc != C6_OK) return rc;
    e->pool_grows++;
    return C6_OK;
}

// raise the caps / map more memory if games are waiting for it or the pools are running low; C6_OK also when there was nothing to do.
// (live_games has just refreshed e->h_hdr; the stream is idle.)
static int maintain(c6_engine *e) {
    C6Dev &d = e->d;
    int parked_caps = 0, parked_pages = 0, parked_chunks = 0, near_cap = 0;
    for (const C6GameHdr &h : e->h_hdr) {
        if (h.status == C6_ST_OVERFLOW) { parked_caps += h.park == C6_PARK_CAPS; parked_chunks += h.park == C6_PARK_CHUNKS; parked_pages += h.park != C6_PARK_CAPS && h.park != C6_PARK_CHUNKS; }
        else if (h.status == C6_ST_PLAYING && ((size_t)h.live_pages * 4 > (size_t)d.pages_cap * 3 || (size_t)h.hslots * 2 > (size_t)d.hcap)) near_cap++;
    }
    // Why a parked game cannot be served is a statement about NOW (one failed growth used to condemn every later parked game):
    // the bits are recomputed at every call.
    int cannot = 0;
    if (parked_caps > 0 || near_cap > 0) {
        const int pc = std::min(d.maxpages, 2 * d.pages_cap);
        const int hc = std::min(e->hmax, pow2_at_least(2L * ((long)pc * C6_PAGE_RECS + d.S), C6_CHUNK_SLOTS));
        if (pc != d.pages_cap || hc != d.hcap) { d.pages_cap = pc; d.hcap = hc; e->cap_raises++; }
        else if (parked_caps > 0) cannot |= C6_PARK_CAPS;      // a game at max_records_per_game: it stays parked (steady state: abandoned)
    }
    unsigned long long ctr[C6_POOL_N];
    HIPCHK(hipMemcpy(ctr, d.pool_ctr, sizeof ctr, hipMemcpyDeviceToHost));
    const size_t free_pages = (size_t)(ctr[C6_POOL_PAGE_TAIL] - ctr[C6_POOL_PAGE_HEAD]), free_chunks = (size_t)(ctr[C6_POOL_CHUNK_TAIL] - ctr[C6_POOL_CHUNK_HEAD]);
    const size_t move_pages = (size_t)(d.S + C6_PAGE_RECS - 1) / C6_PAGE_RECS + 1;
    const size_t move_chunks = (size_t)pow2_at_least(2L * d.S, C6_CHUNK_SLOTS) >> C6_CHUNK_SHIFT;      // what a fresh game's table takes
    const size_t low_pages = std::max<size_t>(64, (size_t)d.G * move_pages / 8);      // an eighth of the games could start a move
    const size_t low_chunks = std::max<size_t>(16, (size_t)d.G * move_chunks / 8);
    const bool want_pages = parked_pages > 0 || free_pages < low_pages, want_chunks = parked_chunks > 0 || free_chunks < low_chunks;
    if (want_pages || want_chunks) {
        int rc = grow_pools(e, want_pages ? (size_t)parked_pages * move_pages + low_pages : 0, want_chunks ? (size_t)parked_chunks * move_chunks + low_chunks : 0);
        if (rc != C6_OK) {
            // not an error of this call: the games wait, or leave as partial results (c6_engine_run).  Said once, on stderr.
            if (parked_pages > 0 || parked_chunks > 0) {
                if (!e->said_full) fprintf(stderr, "[connect6_amd] the pools cannot grow any further: %s\n", g_err.c_str());
                e->said_full = true;
                cannot |= (parked_pages > 0 ? C6_PARK_POOL : 0) | (parked_chunks > 0 ? C6_PARK_CHUNKS : 0);
            }
            g_err.clear();
            (void)hipGetLastError();
        }
    }
    e->grow_failed = cannot != 0;
    d.abandon_parked = d.recycle ? cannot : 0;
    return C6_OK;
}

extern "C" int c6_engine_maintain(c6_engine *e) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    int live = 0, errs = 0, ov = 0;
    int rc = live_games(e, &live, &errs, &ov);
    if (rc != C6_OK) return rc;
    return maintain(e);
}

extern "C" int c6_engine_run(c6_engine *e, long max_rounds) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    long done = 0;
    int errs = 0, stuck = 0;
    for (;;) {
        int live = 0, ov = 0;
        int rc = live_games(e, &live, &errs, &ov);
        if (rc != C6_OK) return rc;
        rc = maintain(e);
        if (rc != C6_OK) return rc;
        // parked games wait for pages: while others play, pages keep coming back; when nobody plays any more, only the host's growth
        // (just attempted) can help -- they get two more rounds to pick the new pages up
        if (live == 0 && (ov == 0 || e->grow_failed || stuck >= 2)) break;
        stuck = live == 0 ? stuck + 1 : 0;
        if (max_rounds > 0 && done >= max_rounds) break;
        int chunk = e->cfg.stagger ? std::min(e->d.S, 64) : e->d.S - e->round_in_move;
        if (live == 0) chunk = 1;
        if (max_rounds > 0) chunk = (int)std::min<long>(chunk, max_rounds - done);
        rc = c6_engine_step(e, chunk);
        if (rc != C6_OK) return rc;
        done += chunk;
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (errs > 0 && !(max_rounds > 0 && done >= max_rounds)) {
        // never silently: the stopped games leave as partial results (winner 0, stones so far) and the call says so
        c6_launch_flush_errors(e->d, e->stream);
        HIPCHK(hipStreamSynchronize(e->stream));
        char msg[200];
        snprintf(msg, sizeof msg, "%d game(s) stopped (record pool exhausted at max_records_per_game / device memory, or NaN scores); they are returned as "
                 "partial games with winner 0", errs);
        return fail(C6_ERR_OVERFLOW, msg);
    }
    return C6_OK;
}

extern "C" int c6_engine_stats(c6_engine *e, c6_stats *out) {
    if (!e || !out) return fail(C6_ERR_BAD_ARG, "null argument");
    unsigned long long c[C6_CNT_N];
    HIPCHK(hipMemcpyAsync(c, e->d.counters, sizeof c, hipMemcpyDeviceToHost, e->stream));
    int live = 0, errs = 0;
    int rc = live_games(e, &live, &errs, nullptr);
    if (rc != C6_OK) return rc;
    c[C6_CNT_SIMS] = c[C6_CNT_SELECT] = c[C6_CNT_STONES] = c[C6_CNT_MAXDEPTH] = c[C6_CNT_MAXREC] = 0;
    for (const C6GameHdr &h : e->h_hdr) {      // refreshed by live_games above
        c[C6_CNT_SIMS] += h.g_sims; c[C6_CNT_SELECT] += h.g_select; c[C6_CNT_STONES] += h.g_stones;
        c[C6_CNT_MAXDEPTH] = std::max<unsigned long long>(c[C6_CNT_MAXDEPTH], h.g_maxdepth);
        c[C6_CNT_MAXREC] = std::max<unsigned long long>(c[C6_CNT_MAXREC], h.g_maxrec);
    }
    out->rounds = e->rounds;
    out->simulations = c[C6_CNT_SIMS];
    out->select_steps = c[C6_CNT_SELECT];
    out->stones = c[C6_CNT_STONES];
    out->games_finished = c[C6_CNT_FINISHED];
    out->tuples = c[C6_CNT_TUPLES];
    out->live_games = (uint32_t)live;
    out->overflow_games = (uint32_t)errs + (uint32_t)c[C6_CNT_OVERFLOW];   // stopped now + abandoned / flushed so far: 0 means none was lost
    out->abandoned_games = c[C6_CNT_OVERFLOW];
    out->records_per_game = (uint32_t)(e->d.pages_cap * C6_PAGE_RECS);
    out->pool_grows = e->pool_grows;
    {
        unsigned long long pc[C6_POOL_N];
        HIPCHK(hipMemcpy(pc, e->d.pool_ctr, sizeof pc, hipMemcpyDeviceToHost));
        out->pool_pages = e->d.pages_total;
        out->pool_pages_free = (uint32_t)(pc[C6_POOL_PAGE_TAIL] - pc[C6_POOL_PAGE_HEAD]);
        out->pool_pages_low = (uint32_t)std::min<unsigned long long>(pc[C6_POOL_PAGE_LOW], 0xffffffffull);
        out->pool_bytes = (uint64_t)e->vm_prior.mapped + e->vm_stat.mapped + e->vm_ht.mapped + e->vm_tag.mapped;
        out->cap_raises = e->cap_raises;
        uint32_t parked = 0, mp = 0;
        for (const C6GameHdr &h : e->h_hdr) { parked += h.status == C6_ST_OVERFLOW; mp = std::max<uint32_t>(mp, (uint32_t)h.live_pages); }
        out->parked_games = parked;
        out->max_pages_held = mp;
    }
    out->max_records_used = (uint32_t)c[C6_CNT_MAXREC];
    out->max_depth = (uint32_t)c[C6_CNT_MAXDEPTH];
    return C6_OK;
}

extern "C" int c6_engine_fetch_games(c6_engine *e, int cap, int *count, int *winners, int *lens, uint32_t *uids, int16_t *moves) {
    if (!e || !count) return fail(C6_ERR_BAD_ARG, "null argument");
    C6Dev &d = e->d;
    HIPCHK(hipStreamSynchronize(e->stream));
    { int rc_ = range_status(e->net); if (rc_ != C6_OK) return rc_; }
    unsigned long long q = 0;
    HIPCHK(hipMemcpy(&q, d.counters + C6_CNT_FETCHQ, sizeof q, hipMemcpyDeviceToHost));
    if (q > (unsigned long long)d.finq_cap) {
        // the device keeps the first finq_cap (= max(4 G, 1024)) results between two drains and counts the rest: said, not hidden
        if (!e->said_finq) fprintf(stderr, "[connect6_amd] c6_engine_fetch_games: %llu finished games since the queue was last drained, it holds %d -- fetch more often\n", q, d.finq_cap);
        e->said_finq = true;
        q = d.finq_cap;
    }
    // r5: a caller with room for fewer games than are queued gets the NEXT `cap` of them at every call (e->fetched = the queue entries already
    // handed out); before, it was handed the first `cap` again and again.  The queue restarts at slot 0 once all of it has been handed out.
    const unsigned long long head = std::min<unsigned long long>(e->fetched, q);
    const int n = (int)std::min<unsigned long long>(q - head, (unsigned long long)std::max(cap, 0));
    if (n > 0) {
        if (!winners || !lens) return fail(C6_ERR_BAD_ARG, "c6_engine_fetch_games: winners / lens are null");
        HIPCHK(hipMemcpy(winners, d.fin_winner + head, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lens, d.fin_len + head, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        if (uids) HIPCHK(hipMemcpy(uids, d.fin_uid + head, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (moves) HIPCHK(hipMemcpy(moves, d.fin_moves + (size_t)head * d.cells, (size_t)n * d.cells * sizeof(int16_t), hipMemcpyDeviceToHost));
    }
    e->fetched = head + (unsigned long long)n;
    if (e->fetched == q) {
        HIPCHK(hipMemset(d.counters + C6_CNT_FETCHQ, 0, sizeof(unsigned long long)));
        e->fetched = 0;
    }
    *count = n;
    return C6_OK;
}

// Records [0, tuple_head) of the buffer have been handed out already.  Moves the rest to the front (the ranges may overlap: through a
// temporary) and rewrites the device counter.  Stream drained by the caller.
static int compact_tuples(c6_engine *e, size_t have) {
    C6Dev &d = e->d;
    const size_t head = std::min(e->tuple_head, have), left = have - head;
    if (head && left) {
        TmpDev t;
        uint8_t *tmp = t.get<uint8_t>(left * d.tuple_stride);
        if (!tmp) return fail(C6_ERR_HIP, "hipMalloc failed");
        HIPCHK(hipMemcpyAsync(tmp, d.tuple_buf + head * d.tuple_stride, left * d.tuple_stride, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(d.tuple_buf, tmp, left * d.tuple_stride, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    const uint32_t l32 = (uint32_t)left;
    HIPCHK(hipMemcpy(d.tuple_count, &l32, sizeof l32, hipMemcpyHostToDevice));
    e->tuple_head = 0;
    return C6_OK;
}

extern "C" int c6_engine_fetch_tuples(c6_engine *e, size_t cap, size_t *count, void *host_buf) {
    if (!e || !count) return fail(C6_ERR_BAD_ARG, "null argument");
    C6Dev &d = e->d;
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    { int rc_ = range_status(e->net); if (rc_ != C6_OK) return rc_; }
    if (!d.tuple_buf) { *count = 0; return C6_OK; }
    uint32_t n = 0;
    HIPCHK(hipMemcpy(&n, d.tuple_count, sizeof n, hipMemcpyDeviceToHost));
    const size_t have = std::min<size_t>(n, d.tuple_cap);      // (the engine stops writing at tuple_cap: records beyond it were never stored)
    const size_t head = std::min(e->tuple_head, have);
    const size_t m = host_buf ? std::min<size_t>(have - head, cap) : have - head;      // host_buf NULL: discards all
    if (m && host_buf) HIPCHK(hipMemcpy(host_buf, d.tuple_buf + head * d.tuple_stride, m * d.tuple_stride, hipMemcpyDeviceToHost));
    // A caller with room for fewer records than the buffer holds keeps the rest for its next call.  ADVICE r5: the remainder is no longer moved to the
    // front at every partial fetch (an allocation + two copies per call: O(n^2 / cap) for a caller draining in small pieces) -- a read offset advances
    // instead; the buffer is compacted only once the handed-out prefix passes half of it, and reset when it is drained.
    e->tuple_head = head + m;
    *count = m;
    if (e->tuple_head >= have || 2 * e->tuple_head > d.tuple_cap) return compact_tuples(e, have);
    return C6_OK;
}

// hand the tuples emitted so far to a replay buffer, device to device (no host round trip), and reset the counter
extern "C" int c6_engine_push_tuples(c6_engine *e, c6_replay *r) {
    if (!e || !r) return fail(C6_ERR_BAD_ARG, "null argument");
    C6Dev &d = e->d;
    if (!d.tuple_buf) return fail(C6_ERR_BAD_ARG, "no tuple buffer: call c6_engine_enable_tuples / c6_engine_set_tuple_buffer first");
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->tuple_head) {      // records a partial c6_engine_fetch_tuples has handed out already are not pushed again
        uint32_t n = 0;
        HIPCHK(hipMemcpy(&n, d.tuple_count, sizeof n, hipMemcpyDeviceToHost));
        int rc_ = compact_tuples(e, std::min<size_t>(n, d.tuple_cap));
        if (rc_ != C6_OK) return rc_;
    }
    int rc = c6_replay_push_dev(r, d.tuple_buf, d.tuple_count, (size_t)d.tuple_cap);
    if (rc != C6_OK) return fail(rc, "c6_replay_push_dev failed");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemset(d.tuple_count, 0, sizeof(uint32_t)));
    return C6_OK;
}

// enable tuple recording into an engine-owned buffer of `capacity_records`
extern "C" int c6_engine_enable_tuples(c6_engine *e, size_t capacity_records) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    HIPCHK(hipStreamSynchronize(e->stream));
    return ensure_own_tuples(e, capacity_records);
}

// Invariants of the two pools (a self-check for tests and soak runs): every page / chunk id below the pools' sizes is
// either held by exactly one game (a valid slot of its page table below npages / of its chunk list below hslots / 1024) or sits
// exactly once in its free ring between head and tail.  dup = ids seen more than once, leaked = ids seen nowhere.  Synchronises.
extern "C" int c6_engine_pool_audit(c6_engine *e, c6_pool_audit *out) {
    if (!e || !out) return fail(C6_ERR_BAD_ARG, "null argument");
    C6Dev &d = e->d;
    HIPCHK(hipStreamSynchronize(e->stream));
    const int G = d.G;
    std::vector<C6GameHdr> hdr(G);
    HIPCHK(hipMemcpy(hdr.data(), d.hdr, (size_t)G * sizeof(C6GameHdr), hipMemcpyDeviceToHost));
    unsigned long long ctr[C6_POOL_N];
    HIPCHK(hipMemcpy(ctr, d.pool_ctr, sizeof ctr, hipMemcpyDeviceToHost));
    size_t wp = 1, wc = 1;
    for (int g = 0; g < G; g++) { wp = std::max<size_t>(wp, hdr[g].npages); wc = std::max<size_t>(wc, hdr[g].hslots >> C6_CHUNK_SHIFT); }
    wp = std::min<size_t>(wp, d.maxpages); wc = std::min<size_t>(wc, d.maxchunks);
    std::vector<uint32_t> pt((size_t)G * wp), ct((size_t)G * wc);
    HIPCHK(hipMemcpy2D(pt.data(), wp * 4, d.page_table, (size_t)d.maxpages * 4, wp * 4, G, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy2D(ct.data(), wc * 4, d.chunk_table, (size_t)d.maxchunks * 4, wc * 4, G, hipMemcpyDeviceToHost));
    auto ring_ids = [&](const uint32_t *ring, uint32_t mask, unsigned long long head, unsigned long long tail, std::vector<uint32_t> &ids) -> int {
        std::vector<uint32_t> all((size_t)mask + 1);
        HIPCHK(hipMemcpy(all.data(), ring, all.size() * 4, hipMemcpyDeviceToHost));
        if (tail < head || tail - head > all.size()) return fail(C6_ERR_HIP, "pool ring: head beyond tail");
        for (unsigned long long t = head; t < tail; t++) ids.push_back(all[(size_t)(t & mask)]);
        return C6_OK;
    };
    memset(out, 0, sizeof *out);
    int rc;
    {
        std::vector<uint8_t> seen(d.pages_total, 0);
        std::vector<uint32_t> fr;
        if ((rc = ring_ids(d.page_ring, d.page_ring_mask, ctr[C6_POOL_PAGE_HEAD], ctr[C6_POOL_PAGE_TAIL], fr))) return rc;
        out->pages_total = d.pages_total;
        out->pages_free = (uint32_t)fr.size();
        auto mark = [&](uint32_t id) { if (id >= d.pages_total) out->pages_bad++; else if (seen[id]++) out->pages_dup++; };
        for (uint32_t id : fr) mark(id);
        for (int g = 0; g < G; g++)
            for (size_t k = 0; k < std::min<size_t>(hdr[g].npages, wp); k++) {
                const uint32_t id = pt[(size_t)g * wp + k];
                if (id != C6_NOPAGE) { out->pages_held++; mark(id); }
            }
        for (uint8_t v : seen) out->pages_leaked += v == 0;
    }
    {
        std::vector<uint8_t> seen(d.chunks_total, 0);
        std::vector<uint32_t> fr;
        if ((rc = ring_ids(d.chunk_ring, d.chunk_ring_mask, ctr[C6_POOL_CHUNK_HEAD], ctr[C6_POOL_CHUNK_TAIL], fr))) return rc;
        out->chunks_total = d.chunks_total;
        out->chunks_free = (uint32_t)fr.size();
        auto mark = [&](uint32_t id) { if (id >= d.chunks_total) out->chunks_bad++; else if (seen[id]++) out->chunks_dup++; };
        for (uint32_t id : fr) mark(id);
        for (int g = 0; g < G; g++)
            for (size_t k = 0; k < std::min<size_t>(hdr[g].hslots >> C6_CHUNK_SHIFT, wc); k++) { out->chunks_held++; mark(ct[(size_t)g * wc + k]); }
        for (uint8_t v : seen) out->chunks_leaked += v == 0;
    }
    return C6_OK;
}

__global__ void c6_root_probe_kernel(C6Dev E, int g, int32_t *visits, float *q, float *prior, int *root_stats, float *root_f) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int cells = E.cells;
    const int p = E.hdr[g].root_rec, H = (int)E.hdr[g].hslots;
    const uint32_t *pt = E.page_table + (size_t)g * E.maxpages, *ck = E.chunk_table + (size_t)g * E.maxchunks;
    auto rec = [&](int r) { return ((size_t)pt[r >> C6_PAGE_SHIFT] << C6_PAGE_SHIFT) | (size_t)(r & (C6_PAGE_RECS - 1)); };
    if (c == 0) {
        root_stats[0] = p >= 0 ? E.rec_stat[rec(p)].visit : 0;
        root_stats[1] = E.hdr[g].live_pages * C6_PAGE_RECS - E.hdr[g].free_top;
        root_f[0] = p >= 0 ? E.rec_stat[rec(p)].value : 0.f;
        root_f[1] = p >= 0 ? E.rec_stat[rec(p)].qsum : 0.f;
        root_f[2] = p >= 0 ? E.rec_stat[rec(p)].q : 0.f;
    }
    if (c >= cells) return;
    const int8_t v = E.root_board[(size_t)g * E.cellsP + c];
    visits[c] = -1; q[c] = 0.f; prior[c] = 0.f;
    if (v != 0 || p < 0 || H <= 0) { if (v == 0) visits[c] = 0; return; }
    const int turn = E.turn[g];
    const unsigned long long key = E.hdr[g].root_key ^ E.zob[(turn == C6_WHITE ? C6_MAXCELLS : 0) + c];
    uint32_t slot = (uint32_t)(key ^ (key >> 32)) & (uint32_t)(H - 1);
    int idx = -1;
    for (int it = 0; it < H; it++) {
        const C6HtEntry en = E.ht[((size_t)ck[slot >> C6_CHUNK_SHIFT] << C6_CHUNK_SHIFT) | (size_t)(slot & (C6_CHUNK_SLOTS - 1))];
        if (en.key == key) { idx = (int)en.idx1 - 1; break; }
        if (en.key == 0ull) break;
        slot = (slot + 1) & (uint32_t)(H - 1);
    }
    visits[c] = idx >= 0 ? E.rec_stat[rec(idx)].visit : 0;
    q[c] = idx >= 0 ? E.rec_stat[rec(idx)].q : 0.f;
    prior[c] = E.rec_prior[rec(p) * (size_t)cells + c];
}

extern "C" int c6_engine_root(c6_engine *e, int g, c6_root_info *info, int8_t *board, int32_t *visits, float *q, float *prior) {
    if (!e || g < 0 || g >= e->d.G) return fail(C6_ERR_BAD_ARG, "bad game index");
    C6Dev &d = e->d;
    const int cells = d.cells;
    HIPCHK(hipStreamSynchronize(e->stream));
    TmpDev tmp;      // (freed on every path out of the function)
    int32_t *dv = tmp.get<int32_t>(cells);
    float *dq = tmp.get<float>(cells), *dp = tmp.get<float>(cells), *drf = tmp.get<float>(3);
    int *drs = tmp.get<int>(2);
    if (!dv || !dq || !dp || !drf || !drs) return fail(C6_ERR_HIP, "hipMalloc failed");
    hipLaunchKernelGGL(c6_root_probe_kernel, dim3((cells + 63) / 64), dim3(64), 0, e->stream, d, g, dv, dq, dp, drs, drf);
    std::vector<int32_t> hv(cells); std::vector<float> hq(cells), hp(cells);
    int rs[2]; float rf[3];
    HIPCHK(hipMemcpy(hv.data(), dv, cells * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hq.data(), dq, cells * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hp.data(), dp, cells * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rs, drs, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rf, drf, 12, hipMemcpyDeviceToHost));
    if (visits) memcpy(visits, hv.data(), cells * 4);
    if (q) memcpy(q, hq.data(), cells * 4);
    if (prior) memcpy(prior, hp.data(), cells * 4);
    if (board) HIPCHK(hipMemcpy(board, d.root_board + (size_t)g * d.cellsP, cells, hipMemcpyDeviceToHost));
    if (info) {
        int8_t t, r, st, w; int16_t ns; uint32_t nsr;
        HIPCHK(hipMemcpy(&t, d.turn + g, 1, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&r, d.remain + g, 1, hipMemcpyDeviceToHost));
        C6GameHdr hh;
        HIPCHK(hipMemcpy(&hh, d.hdr + g, sizeof hh, hipMemcpyDeviceToHost));
        st = hh.status;
        HIPCHK(hipMemcpy(&w, d.winner + g, 1, hipMemcpyDeviceToHost));
        ns = hh.nstones;
        nsr = hh.nsearch;
        info->turn = t; info->remain = r; info->stones = ns; info->status = st; info->winner = w;
        info->root_visit = rs[0]; info->records_used = rs[1];
        info->root_value = rf[0]; info->root_q_sum = rf[1]; info->root_q_value = rf[2];
        info->nsearch = nsr;
    }
    return C6_OK;
}

// the canonical game state K1 works on (structure-of-arrays on the device): boards[g][N*N] (-1 black, 0 empty, +1 white), the side to move
// and the stones it still has to place (Game{turn, num_remain}, game_impl/mod.rs:117-121).  Synchronises.
extern "C" int c6_engine_boards(c6_engine *e, int8_t *boards, int8_t *turn, int8_t *remain) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    const C6Dev &d = e->d;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (boards) {
        std::vector<int8_t> soa((size_t)d.cells * d.Gpad);
        HIPCHK(hipMemcpy(soa.data(), d.board_soa, soa.size(), hipMemcpyDeviceToHost));
        for (int g = 0; g < d.G; g++)
            for (int c = 0; c < d.cells; c++) boards[(size_t)g * d.cells + c] = soa[(size_t)c * d.Gpad + g];
    }
    if (turn) HIPCHK(hipMemcpy(turn, d.turn, (size_t)d.G, hipMemcpyDeviceToHost));
    if (remain) HIPCHK(hipMemcpy(remain, d.remain, (size_t)d.G, hipMemcpyDeviceToHost));
    return C6_OK;
}

extern "C" int c6_engine_profile(c6_engine *e, int enable) {
    if (!e) return fail(C6_ERR_BAD_ARG, "null engine");
    e->profiling = enable != 0;
    c6_timers_enable(e->timers, e->profiling);
    return C6_OK;
}

extern "C" int c6_engine_profile_read(c6_engine *e, c6_profile *out, int reset) {
    if (!e || !out) return fail(C6_ERR_BAD_ARG, "null argument");
    HIPCHK(hipStreamSynchronize(e->stream));
    memset(out, 0, sizeof *out);
    unsigned long long n = 0;
    c6_timers_collect(e->timers, &out->conv_ms, &n, &out->conv_flops, &out->conv_pre_ms, &out->conv_exec_flops, &out->conv_post_ms, reset != 0);
    out->conv_launches = n;
    e->t_select.collect(); e->t_expand.collect(); e->t_eval.collect(); e->t_move.collect();
    out->select_ms = e->t_select.ms; out->select_launches = e->t_select.n;
    out->expand_ms = e->t_expand.ms; out->expand_launches = e->t_expand.n;
    out->eval_other_ms = e->t_eval.ms - out->conv_ms;
    out->move_ms = e->t_move.ms;
    if (reset) {
        for (PhaseTimer *t : {&e->t_select, &e->t_expand, &e->t_eval, &e->t_move}) { t->ms = 0; t->n = 0; }
    }
    return C6_OK;
}

// ---- stand-alone operators ------------------------------------------------------------------


static int check_n(int n) {
    if (n != 9 && n != 15 && n != 19) return fail(C6_ERR_BAD_ARG, "board_size must be 9, 15 or 19");
    return C6_OK;
}

// SoA host arrays may have any G; the kernels need G % 4 == 0, so the device copy is padded
static int8_t *soa_to_dev(TmpDev &t, int cells, int G, int Gpad, const int8_t *h) {
    int8_t *d = t.get<int8_t>((size_t)cells * Gpad);
    if (d && h) hipMemcpy2D(d, Gpad, h, G, G, cells, hipMemcpyHostToDevice);
    return d;
}

extern "C" int c6_board_step(int n, int G, int8_t *boards, int8_t *turn, int8_t *remain, const int32_t *pos, int8_t *winner,
                             int8_t *err) {
    int rc = check_n(n); if (rc) return rc;
    rc = require_gpu(0); if (rc) return rc;
    const int cells = n * n, Gpad = (G + 3) & ~3;
    TmpDev t;
    int8_t *db = soa_to_dev(t, cells, G, Gpad, boards), *dt = t.get<int8_t>(Gpad), *dr = t.get<int8_t>(Gpad), *dw = t.get<int8_t>(Gpad),
           *de = t.get<int8_t>(Gpad);
    int32_t *dp = t.get<int32_t>(Gpad);
    if (!db || !dt || !dr || !dw || !de || !dp) return fail(C6_ERR_HIP, "hipMalloc failed");
    HIPCHK(hipMemset(dp, 0xff, (size_t)Gpad * 4));   // padding games: pos = -1 (no move)
    HIPCHK(hipMemcpy(dt, turn, G, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dr, remain, G, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dp, pos, (size_t)G * 4, hipMemcpyHostToDevice));
    c6_launch_board_step(n, Gpad, db, dt, dr, dp, dw, de, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy2D(boards, G, db, Gpad, G, cells, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(turn, dt, G, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(remain, dr, G, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(winner, dw, G, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(err, de, G, hipMemcpyDeviceToHost));
    return C6_OK;
}

extern "C" int c6_search_winner(int n, int G, const int8_t *boards, int8_t *winner) {
    int rc = check_n(n); if (rc) return rc;
    rc = require_gpu(0); if (rc) return rc;
    const int cells = n * n, Gpad = (G + 3) & ~3;
    TmpDev t;
    int8_t *db = soa_to_dev(t, cells, G, Gpad, boards), *dw = t.get<int8_t>(Gpad);
    if (!db || !dw) return fail(C6_ERR_HIP, "hipMalloc failed");
    c6_launch_search_winner(n, Gpad, db, dw, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(winner, dw, G, hipMemcpyDeviceToHost));
    return C6_OK;
}

extern "C" int c6_legal_mask(int n, int G, const int8_t *boards, uint8_t *mask) {
    int rc = check_n(n); if (rc) return rc;
    rc = require_gpu(0); if (rc) return rc;
    const int cells = n * n, Gpad = (G + 3) & ~3, NB = (cells + 7) / 8;
    TmpDev t;
    int8_t *db = soa_to_dev(t, cells, G, Gpad, boards);
    uint8_t *dm = t.get<uint8_t>((size_t)NB * Gpad);
    if (!db || !dm) return fail(C6_ERR_HIP, "hipMalloc failed");
    c6_launch_legal_mask(n, Gpad, db, dm, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy2D(mask, G, dm, Gpad, G, NB, hipMemcpyDeviceToHost));
    return C6_OK;
}

extern "C" int c6_augment8(int n, int num, const int8_t *boards, int8_t *out) {
    int rc = check_n(n); if (rc) return rc;
    rc = require_gpu(0); if (rc) return rc;
    const int cells = n * n;
    std::vector<int16_t> aug, rec;
    build_maps(n, aug, rec);
    TmpDev t;
    int8_t *db = t.get<int8_t>((size_t)num * cells, boards), *dout = t.get<int8_t>((size_t)num * 8 * cells);
    int16_t *dm = t.get<int16_t>(aug.size(), aug.data());
    if (!db || !dout || !dm) return fail(C6_ERR_HIP, "hipMalloc failed");
    c6_launch_augment_op(n, num, db, dm, dout, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout, (size_t)num * 8 * cells, hipMemcpyDeviceToHost));
    return C6_OK;
}

extern "C" int c6_recover8(int n, int num, const int8_t *boards, const float *values8, const float *policies8, float *value, float *prior) {
    int rc = check_n(n); if (rc) return rc;
    rc = require_gpu(0); if (rc) return rc;
    const int cells = n * n;
    std::vector<int16_t> aug, rec;
    build_maps(n, aug, rec);
    TmpDev t;
    int8_t *db = t.get<int8_t>((size_t)num * cells, boards);
    float *dv8 = t.get<float>((size_t)num * 8, values8), *dp8 = t.get<float>((size_t)num * 8 * cells, policies8);
    float *dv = t.get<float>(num), *dp = t.get<float>((size_t)num * cells);
    int16_t *dm = t.get<int16_t>(rec.size(), rec.data());
    if (!db || !dv8 || !dp8 || !dv || !dp || !dm) return fail(C6_ERR_HIP, "hipMalloc failed");
    c6_launch_recover_op(n, num, db, dv8, dp8, dm, dv, dp, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(value, dv, (size_t)num * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(prior, dp, (size_t)num * cells * 4, hipMemcpyDeviceToHost));
    return C6_OK;
}

extern "C" int c6_dirichlet_noise(int n, int num, double alpha, const uint32_t *k0, const uint32_t *k1, const uint32_t *search,
                                  const uint32_t *depth, const int8_t *boards, float *noise) {
    int rc = check_n(n); if (rc) return rc;
    rc = require_gpu(0); if (rc) return rc;
    const int cells = n * n;
    TmpDev t;
    uint32_t *d0 = t.get<uint32_t>(num, k0), *d1 = t.get<uint32_t>(num, k1), *ds = t.get<uint32_t>(num, search), *dd = t.get<uint32_t>(num, depth);
    int8_t *db = t.get<int8_t>((size_t)num * cells, boards);
    float *dn = t.get<float>((size_t)num * cells);
    if (!d0 || !d1 || !ds || !dd || !db || !dn) return fail(C6_ERR_HIP, "hipMalloc failed");
    const double gd = (alpha + 1.0) - 1.0 / 3.0, gc = 1.0 / std::sqrt(9.0 * gd);
    c6_launch_noise_op(n, num, alpha, gd, gc, d0, d1, ds, dd, db, dn, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(noise, dn, (size_t)num * cells * 4, hipMemcpyDeviceToHost));
    return C6_OK;
}

extern "C" int c6_net_forward_prec(int n, int C, int blocks, int precision, const float *blob, size_t nfloats, int num,
                                   const int8_t *boards, const int8_t *player, float *policy, float *value);
extern "C" int c6_net_forward(int n, int C, int blocks, const float *blob, size_t nfloats, int num, const int8_t *boards,
                              const int8_t *player, float *policy, float *value) {
    return c6_net_forward_prec(n, C, blocks, C6_PREC_DEFAULT, blob, nfloats, num, boards, player, policy, value);
}
extern "C" int c6_net_forward_prec(int n, int C, int blocks, int precision, const float *blob, size_t nfloats, int num,
                                   const int8_t *boards, const int8_t *player, float *policy, float *value) {
    int rc = check_n(n); if (rc) return rc;
    rc = require_gpu(0); if (rc) return rc;
    const int cells = n * n;
    NetHost nh;
    rc = net_load(nh, n, C, blocks, blob, nfloats, precision);
    if (rc) { nh.pool.release(); return rc; }
    std::vector<int16_t> aug, rec;
    build_maps(n, aug, rec);
    TmpDev t;
    const size_t act = (size_t)num * cells * C + C6_ACT_SLACK;
    int8_t *db = t.get<int8_t>((size_t)num * cells, boards), *dpl = t.get<int8_t>(num, player);
    int16_t *dm = t.get<int16_t>(aug.size(), aug.data());
    float *a0 = t.get<float>(act), *a1 = t.get<float>(act), *a2 = t.get<float>(act), *ph = t.get<float>((size_t)num * 2 * cells),
          *vh = t.get<float>((size_t)num * cells), *dp = t.get<float>((size_t)num * cells), *dv = t.get<float>(num);
    float *vb = nh.dev.variant == C6_CONV_WINO ? t.get<float>(nh.dev.wino6 ? c6_w6_vbuf_floats(n, num) : c6_conv_wino_vbuf_floats(C, n, num)) : a0;
    if (!db || !dpl || !dm || !a0 || !a1 || !a2 || !ph || !vh || !dp || !dv || !vb) { nh.pool.release(); return fail(C6_ERR_HIP, "hipMalloc failed"); }
    c6_net_forward_dev(nh.dev, num, 1, db, cells, dpl, 1, dm, a0, a1, a2, vb, ph, vh, dp, dv, nullptr, nullptr);
    c6_note_hip(hipGetLastError(), "c6_net_forward");
    if ((rc = c6_launch_status()) != C6_OK) { hipDeviceSynchronize(); nh.pool.release(); return rc; }
    hipError_t he = hipDeviceSynchronize();
    if (he == hipSuccess && (rc = range_status(nh)) != C6_OK) { nh.pool.release(); return rc; }
    if (he == hipSuccess) he = hipMemcpy(policy, dp, (size_t)num * cells * 4, hipMemcpyDeviceToHost);
    if (he == hipSuccess) he = hipMemcpy(value, dv, (size_t)num * 4, hipMemcpyDeviceToHost);
    nh.pool.release();
    if (he != hipSuccess) return fail(C6_ERR_HIP, hipGetErrorString(he));
    return C6_OK;
}

// ---- conv micro-benchmark (development tool; also used to collect per-kernel profiles) ----------
__global__ void c6_fill_random_kernel(float *p, size_t n, uint32_t seed, float lo, float hi) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = lo + (hi - lo) * ((float)(x >> 8) * (1.0f / 16777216.0f));
    }
}
extern "C" int c6_bench_conv(int n, int C, int nboards, int iters, int precision, int ablate, int with_res, double *avg_ms) {
    int rc = check_n(n); if (rc) return rc;
    rc = require_gpu(0); if (rc) return rc;
    if (ablate != 0) return fail(C6_ERR_BAD_ARG, "c6_bench_conv: ablations were removed with the kernels they belonged to");
    if (C != 64 && C != 128 && C != 256) return fail(C6_ERR_BAD_ARG, "channels must be 64, 128 or 256");
    const int variant = conv_variant(precision);
    const size_t cells = (size_t)n * n, act = (size_t)nboards * cells * C + C6_ACT_SLACK,
                 wsz = std::max(c6_conv_packed_wino_floats(C), (c6_conv_packed_m8_bytes(C) + 3) / 4);      // the largest packed layout
    float *in, *res, *out, *w, *bias, *vb = nullptr;
    HIPCHK(hipMalloc(&in, act * 4)); HIPCHK(hipMalloc(&res, act * 4)); HIPCHK(hipMalloc(&out, act * 4));
    HIPCHK(hipMalloc(&w, wsz * 4)); HIPCHK(hipMalloc(&bias, C * 4));
    if (variant == C6_CONV_WINO) HIPCHK(hipMalloc(&vb, c6_conv_wino_vbuf_floats(C, n, nboards) * 4));
    hipLaunchKernelGGL(c6_fill_random_kernel, dim3(4096), dim3(256), 0, nullptr, in, act, 1u, 0.f, 1.f);
    hipLaunchKernelGGL(c6_fill_random_kernel, dim3(4096), dim3(256), 0, nullptr, res, act, 2u, 0.f, 1.f);
    hipLaunchKernelGGL(c6_fill_random_kernel, dim3(256), dim3(256), 0, nullptr, w, wsz, 3u, -0.02f, 0.02f);
    hipLaunchKernelGGL(c6_fill_random_kernel, dim3(1), dim3(256), 0, nullptr, bias, (size_t)C, 4u, -0.1f, 0.1f);
    hipDeviceProp_t prop;
    const int cus = hipGetDeviceProperties(&prop, 0) == hipSuccess ? prop.multiProcessorCount : 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    c6_conv_bench_launch(variant, C, n, nboards, in, w, bias, with_res ? res : nullptr, out, vb, cus, nullptr);
    HIPCHK(hipDeviceSynchronize());
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; i++) c6_conv_bench_launch(variant, C, n, nboards, in, w, bias, with_res ? res : nullptr, out, vb, cus, nullptr);
    hipEventRecord(e1, nullptr);
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *avg_ms = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(in); hipFree(res); hipFree(out); hipFree(w); hipFree(bias);
    if (vb) hipFree(vb);
    return C6_OK;
}

// ---- K1 micro-benchmark: G games resident in HBM, every game places one stone per launch -------------------------------
__global__ void c6_bench_pos_kernel(int cells, int G, int step, int32_t *pos) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) pos[g] = (int)(((unsigned)g * 2654435761u + (unsigned)step * 40503u) % (unsigned)cells);
}
static int bench_board(int n, int G, int iters, bool scan_only, double *avg_ms) {
    int rc = check_n(n); if (rc) return rc;
    rc = require_gpu(0); if (rc) return rc;
    const int cells = n * n, Gpad = (G + 3) & ~3;
    TmpDev t;
    int8_t *db = t.get<int8_t>((size_t)cells * Gpad), *dt = t.get<int8_t>(Gpad), *dr = t.get<int8_t>(Gpad), *dw = t.get<int8_t>(Gpad),
           *de = t.get<int8_t>(Gpad);
    int32_t *dp = t.get<int32_t>(Gpad);
    if (!db || !dt || !dr || !dw || !de || !dp) return fail(C6_ERR_HIP, "hipMalloc failed");
    HIPCHK(hipMemset(dt, 0xff, Gpad));   // black to move
    HIPCHK(hipMemset(dr, 1, Gpad));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float total = 0;
    for (int i = -1; i < iters; i++) {   // occupied cells are rejected by the kernel ("already set position"): still a full board read
        hipLaunchKernelGGL(c6_bench_pos_kernel, dim3((Gpad + 255) / 256), dim3(256), 0, nullptr, cells, Gpad, i + 1, dp);
        hipEventRecord(e0, nullptr);
        if (scan_only) c6_launch_search_winner(n, Gpad, db, dw, nullptr);      // is_game_end alone: N*N bytes read + 1 written
        else c6_launch_board_step(n, Gpad, db, dt, dr, dp, dw, de, nullptr);
        hipEventRecord(e1, nullptr);
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 0) total += ms;
    }
    *avg_ms = total / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return C6_OK;
}
extern "C" int c6_bench_board(int n, int G, int iters, double *avg_ms) { return bench_board(n, G, iters, false, avg_ms); }
// The pools' pop protocol on its own (tests/test_gpu_pool.py): `launches` launches of `waves` waves, each wave asking `tries` times for
// 1-8 ids from a ring of `ids` ids (ids << waves * tries * 4.5, so the ring runs dry early in every launch).  Out: ids handed out more
// than once (dup), ids handed out beyond the ring's content (over = popped - ids when positive), and the fewest ids a launch handed out.
extern "C" int c6_pool_ring_selftest(int ids, int waves, int tries, int launches, uint32_t seed, uint32_t *dup, uint32_t *over, uint32_t *min_popped) {
    if (ids < 8 || waves < 1 || tries < 1 || launches < 1 || !dup || !over || !min_popped) return fail(C6_ERR_BAD_ARG, "bad self-test geometry");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return fail(C6_ERR_NO_GPU, "no gfx950 device");
    uint32_t size = 64;
    while (size < (uint32_t)ids) size <<= 1;
    uint32_t *ring = nullptr, *count = nullptr, *popped = nullptr;
    unsigned long long *ctr = nullptr;
    HIPCHK(hipMalloc(&ring, size * 4)); HIPCHK(hipMalloc(&count, size * 4)); HIPCHK(hipMalloc(&popped, 4)); HIPCHK(hipMalloc(&ctr, C6_POOL_N * 8));
    std::vector<uint32_t> h(size), cnt(size);
    for (uint32_t i = 0; i < size; i++) h[i] = i;
    HIPCHK(hipMemcpy(ring, h.data(), size * 4, hipMemcpyHostToDevice));
    *dup = *over = 0; *min_popped = 0xffffffffu;
    unsigned long long head = 0;
    for (int l = 0; l < launches; l++) {
        unsigned long long c[C6_POOL_N] = {};
        c[C6_POOL_PAGE_HEAD] = head;                       // the ring keeps turning: launch l hands out ring[head .. head + ids)
        c[C6_POOL_PAGE_TAIL] = c[C6_POOL_PAGE_LIMIT] = head + (unsigned long long)ids;
        c[C6_POOL_PAGE_AVAIL] = (unsigned long long)ids;
        c[C6_POOL_PAGE_LOW] = ~0ull;
        HIPCHK(hipMemcpy(ctr, c, sizeof c, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(count, 0, size * 4)); HIPCHK(hipMemset(popped, 0, 4));
        c6_launch_ring_selftest(ring, size - 1, ctr, count, popped, waves, tries, seed + (uint32_t)l * 7919u, 0);
        HIPCHK(hipDeviceSynchronize());
        uint32_t np = 0;
        HIPCHK(hipMemcpy(cnt.data(), count, size * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(&np, popped, 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c, ctr, sizeof c, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < size; i++) if (cnt[i] > 1) *dup += cnt[i] - 1;
        if (np > (uint32_t)ids) *over += np - (uint32_t)ids;
        *min_popped = std::min(*min_popped, np);
        head = (head + (unsigned long long)ids + 13) & 0xffffffull;
    }
    hipFree(ring); hipFree(count); hipFree(popped); hipFree(ctr);
    return C6_OK;
}

extern "C" int c6_bench_board_scan(int n, int G, int iters, double *avg_ms) { return bench_board(n, G, iters, true, avg_ms); }
