// replay.hip -- device-resident replay buffer: the consumer of the pooled training tuples.
// Restates AlphaZero/cpp_weighted/ReplayBuffer.hpp:10-73 (ring of (winner, Path) with `tail` / `num_data`, uniform sampling
// with replacement into winners / players / boards / poses) for tuple records that already live in HBM, so that the
// self-play -> training hand-off needs no host round trip.  The reference seeds its sampler from std::random_device
// (unpinned); here index_i = Philox(seed; i, draw_id, tag 5).x % num_data (C6RNG), which the oracle reproduces.
#include "../../include/connect6_amd.h"
#include "c6_device.h"

#include <algorithm>
#include <string>
#include <vector>

#define C6_TAG_SAMPLE 5u

struct c6_replay {
    int n = 0, cells = 0, stride = 0, device = 0;      // device: the one that was current at create; every call selects it
    size_t max_size = 0;
    uint8_t *ring = nullptr;             // [max_size][stride]
    unsigned long long *state = nullptr; // device: [0] = tail, [1] = num_data
};

int c6_set_error(int code, const char *msg);      // engine.cpp: the message c6_last_error() returns
static int rfail(int code, const char *msg) { return c6_set_error(code, msg); }
#define RCHK(x) do { if ((x) != hipSuccess) return rfail(C6_ERR_HIP, #x); } while (0)
#define RDEV(r) RCHK(hipSetDevice((r)->device))

// one workgroup per record: paths[tail] = record; tail = (tail + 1) % max_size; num_data = min(num_data + count, max_size)
__global__ void c6_replay_push_kernel(uint8_t *ring, unsigned long long *state, unsigned long long max_size, int stride,
                                      const uint8_t *src, const uint32_t *count_dev, unsigned long long count_host, unsigned long long cap) {
    unsigned long long count = count_dev ? (unsigned long long)*count_dev : count_host;
    if (count > cap) count = cap;
    const unsigned long long tail = state[0];
    for (unsigned long long i = blockIdx.x; i < count; i += gridDim.x) {
        const unsigned long long slot = (tail + i) % max_size;
        const uint32_t *s = reinterpret_cast<const uint32_t *>(src + i * stride);
        uint32_t *d = reinterpret_cast<uint32_t *>(ring + slot * stride);
        for (int k = threadIdx.x; k < stride / 4; k += blockDim.x) d[k] = s[k];
    }
}
__global__ void c6_replay_advance_kernel(unsigned long long *state, unsigned long long max_size, const uint32_t *count_dev,
                                         unsigned long long count_host, unsigned long long cap) {
    unsigned long long count = count_dev ? (unsigned long long)*count_dev : count_host;
    if (count > cap) count = cap;
    state[0] = (state[0] + count) % max_size;
    const unsigned long long nd = state[1] + count;
    state[1] = nd < max_size ? nd : max_size;
}

__global__ void c6_replay_sample_kernel(const uint8_t *ring, const unsigned long long *state, int cells, int stride, int num_sample,
                                        uint32_t k0, uint32_t k1, uint32_t draw_id, float *winners, float *players, float *boards,
                                        float *poses, float *pi) {
    const int i = blockIdx.x;
    if (i >= num_sample) return;
    const unsigned long long num_data = state[1];
    if (num_data == 0) return;
    const uint32_t rnd = c6_philox((uint32_t)i, draw_id, 0u, C6_TAG_SAMPLE, k0, k1).x;
    const uint8_t *rec = ring + (unsigned long long)(rnd % num_data) * stride;     // gen() % num_data, ReplayBuffer.hpp:51
    const int off = cells + (cells & 1);
    for (int c = threadIdx.x; c < cells; c += blockDim.x) {
        boards[(size_t)i * cells + c] = (float)(int8_t)rec[c];
        if (pi) pi[(size_t)i * cells + c] = (float)reinterpret_cast<const uint16_t *>(rec + off)[c];
    }
    if (threadIdx.x == 0) {
        winners[i] = (float)(int8_t)rec[stride - 1];
        players[i] = (float)(int8_t)rec[stride - 2];
        const int pos = (int)*reinterpret_cast<const uint16_t *>(rec + stride - 4);
        poses[i] = (float)(pos < cells ? pos : cells - 1);      // host-encoded records are not trusted: the trainer indexes the policy with it
    }
}

extern "C" int c6_replay_create(int n, size_t max_size, c6_replay **out) {
    if (!out || max_size == 0 || (n != 9 && n != 15 && n != 19)) return rfail(C6_ERR_BAD_ARG, "c6_replay_create: bad argument");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return rfail(C6_ERR_NO_GPU, "no HIP device visible; this library has no CPU fallback");
    c6_replay *r = new c6_replay();
    r->n = n; r->cells = n * n; r->stride = (int)c6_tuple_stride(n); r->max_size = max_size;
    hipGetDevice(&r->device);
    if (hipMalloc(&r->ring, max_size * (size_t)r->stride) != hipSuccess || hipMalloc(&r->state, 16) != hipSuccess) {
        delete r;
        return rfail(C6_ERR_HIP, "hipMalloc failed");
    }
    hipMemset(r->state, 0, 16);
    *out = r;
    return C6_OK;
}

// the same on a named device (a host without the HIP headers -- the Rust side of a c6_group -- cannot select one itself)
extern "C" int c6_replay_create_on_device(int n, size_t max_size, int device, c6_replay **out) {
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return rfail(C6_ERR_NO_GPU, "no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= nd) return rfail(C6_ERR_BAD_ARG, "c6_replay_create_on_device: no such device");
    int prev = 0;
    hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return rfail(C6_ERR_HIP, "hipSetDevice failed");
    const int rc = c6_replay_create(n, max_size, out);
    hipSetDevice(prev);
    return rc;
}

extern "C" void c6_replay_destroy(c6_replay *r) {
    if (!r) return;
    hipSetDevice(r->device);
    hipDeviceSynchronize();
    hipFree(r->ring);
    hipFree(r->state);
    delete r;
}

static int push(c6_replay *r, const uint8_t *src_dev, const uint32_t *count_dev, size_t count_host, size_t cap) {
    RDEV(r);
    const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>(cap, 1), 4096);
    hipLaunchKernelGGL(c6_replay_push_kernel, dim3(grid), dim3(128), 0, nullptr, r->ring, r->state, (unsigned long long)r->max_size, r->stride,
                       src_dev, count_dev, (unsigned long long)count_host, (unsigned long long)cap);
    hipLaunchKernelGGL(c6_replay_advance_kernel, dim3(1), dim3(1), 0, nullptr, r->state, (unsigned long long)r->max_size, count_dev,
                       (unsigned long long)count_host, (unsigned long long)cap);
    RCHK(hipGetLastError());
    return C6_OK;
}

extern "C" int c6_replay_push_dev(c6_replay *r, const void *tuples_dev, const uint32_t *count_dev, size_t cap) {
    if (!r || !tuples_dev || !count_dev) return rfail(C6_ERR_BAD_ARG, "c6_replay_push_dev: null argument");
    if (cap > r->max_size) cap = r->max_size;     // a single push larger than the ring would overwrite itself
    return push(r, reinterpret_cast<const uint8_t *>(tuples_dev), count_dev, 0, cap);
}

extern "C" int c6_replay_push_host(c6_replay *r, const void *tuples_host, size_t count) {
    if (!r || (!tuples_host && count)) return rfail(C6_ERR_BAD_ARG, "c6_replay_push_host: null argument");
    RDEV(r);
    // pushes larger than the ring keep the LAST max_size records, like consecutive push_game calls
    const uint8_t *src = reinterpret_cast<const uint8_t *>(tuples_host);
    size_t done = 0;
    while (done < count) {
        const size_t m = std::min(count - done, r->max_size);
        uint8_t *tmp = nullptr;
        RCHK(hipMalloc(&tmp, m * (size_t)r->stride));
        RCHK(hipMemcpy(tmp, src + done * (size_t)r->stride, m * (size_t)r->stride, hipMemcpyHostToDevice));
        int rc = push(r, tmp, nullptr, m, m);
        hipDeviceSynchronize();
        hipFree(tmp);
        if (rc) return rc;
        done += m;
    }
    return C6_OK;
}

extern "C" int c6_replay_clear(c6_replay *r) {
    if (!r) return rfail(C6_ERR_BAD_ARG, "null replay buffer");
    RDEV(r);
    RCHK(hipMemset(r->state, 0, 16));
    return C6_OK;
}

extern "C" int c6_replay_size(c6_replay *r, size_t *num_data, size_t *tail) {
    if (!r) return rfail(C6_ERR_BAD_ARG, "null replay buffer");
    RDEV(r);
    unsigned long long st[2];
    RCHK(hipMemcpy(st, r->state, 16, hipMemcpyDeviceToHost));
    if (tail) *tail = (size_t)st[0];
    if (num_data) *num_data = (size_t)st[1];
    return C6_OK;
}

// ReplayBuffer::sample into DEVICE buffers on a caller's stream (the trainer's mini-batch, no host round trip)
extern "C" int c6_replay_sample_device(c6_replay *r, int num_sample, uint64_t seed, uint32_t draw_id, float *winners, float *players,
                                       float *boards, float *poses, void *stream) {
    if (!r || num_sample < 1 || !winners || !players || !boards || !poses) return rfail(C6_ERR_BAD_ARG, "c6_replay_sample_device: bad argument");
    RDEV(r);
    unsigned long long st[2] = {0, 0};
    RCHK(hipMemcpy(st, r->state, sizeof(st), hipMemcpyDeviceToHost));
    if (st[1] == 0) return rfail(C6_ERR_BAD_ARG, "c6_replay_sample_device: the buffer is empty");
    hipLaunchKernelGGL(c6_replay_sample_kernel, dim3(num_sample), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), r->ring, r->state, r->cells,
                       r->stride, num_sample, (uint32_t)seed, (uint32_t)(seed >> 32), draw_id, winners, players, boards, poses, (float *)nullptr);
    return C6_OK;
}
extern "C" int c6_replay_board_size(const c6_replay *r) { return r ? r->n : 0; }

extern "C" int c6_replay_sample(c6_replay *r, int num_sample, uint64_t seed, uint32_t draw_id, float *winners, float *players,
                                float *boards, float *poses, float *pi) {
    if (!r || num_sample < 1 || !winners || !players || !boards || !poses) return rfail(C6_ERR_BAD_ARG, "c6_replay_sample: bad argument");
    RDEV(r);
    {
        unsigned long long st[2] = {0, 0};
        RCHK(hipMemcpy(st, r->state, sizeof(st), hipMemcpyDeviceToHost));
        if (st[1] == 0) return rfail(C6_ERR_BAD_ARG, "c6_replay_sample: the buffer is empty (ReplayBuffer::sample on no data)");
    }
    const size_t nb = (size_t)num_sample * r->cells;
    float *dw, *dp, *db, *dq, *dpi = nullptr;
    RCHK(hipMalloc(&dw, num_sample * 4)); RCHK(hipMalloc(&dp, num_sample * 4)); RCHK(hipMalloc(&dq, num_sample * 4));
    RCHK(hipMalloc(&db, nb * 4));
    if (pi) RCHK(hipMalloc(&dpi, nb * 4));
    hipLaunchKernelGGL(c6_replay_sample_kernel, dim3(num_sample), dim3(128), 0, nullptr, r->ring, r->state, r->cells, r->stride, num_sample,
                       (uint32_t)seed, (uint32_t)(seed >> 32), draw_id, dw, dp, db, dq, dpi);
    RCHK(hipMemcpy(winners, dw, num_sample * 4, hipMemcpyDeviceToHost));
    RCHK(hipMemcpy(players, dp, num_sample * 4, hipMemcpyDeviceToHost));
    RCHK(hipMemcpy(poses, dq, num_sample * 4, hipMemcpyDeviceToHost));
    RCHK(hipMemcpy(boards, db, nb * 4, hipMemcpyDeviceToHost));
    if (pi) RCHK(hipMemcpy(pi, dpi, nb * 4, hipMemcpyDeviceToHost));
    hipFree(dw); hipFree(dp); hipFree(dq); hipFree(db); hipFree(dpi);
    return C6_OK;
}
