// Probe (round 5): do the REAL kernels of the FP32 Winograd tower overlap when they share a CU?
//   G  = c6_w6_gemm_kernel<true, false, false> (the 200-register instantiation: 8 waves, 133 KB of LDS -- leaves 112 registers per SIMD lane)
//   O  = the tower's last output transform (c6_w6_out_kernel's body) compiled for <= 112 registers, 4 waves per workgroup, persistent grid
// so that one workgroup of each fits on a CU together.  Times: G alone, O alone, O launched first (its waves are the OLDER ones) then G on
// another stream, G first then O.  together ~ max => the boundary kernel's memory time hides under the matrix work; ~ sum => it does not.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/probe/_w6_overlap tools/probe/w6_overlap.hip && tools/probe/_w6_overlap
#include "../../alphazero-connect6_amd/csrc/conv_wino6.hip"
#include <chrono>
#include <thread>
#include <cstdio>
#include <vector>
#include <atomic>
#include <fstream>
#include <glob.h>
#include <climits>
#include <cstdlib>
#include <string>
void c6_note_hip(hipError_t, const char *) {}

// one slice (b) of a tile at a time: 8 row sums of 1 KiB per wave instruction in flight, slice transform, bias, ReLU, the heads' three filters
template <int CLS>
__device__ __forceinline__ void out1_tile(const W6OutCtx &X, const float *zb, long bd, int ty, int x0, int wd, int lane, int b) {
    constexpr int C = W6_C, N = W6_N;
    constexpr int rows = CLS == W6_CLS_D || CLS == W6_CLS_DP ? 8 : CLS == W6_CLS_A ? 6 : 5;
    constexpr bool along_row = CLS == W6_CLS_D || CLS == W6_CLS_DP || CLS == W6_CLS_B;
    constexpr int npx = CLS == W6_CLS_D || CLS == W6_CLS_DP ? 6 : CLS == W6_CLS_A ? 4 : 3;
    if (!W6_SLICE_OK(CLS, b)) return;
    f32x4 zt[8];
#pragma unroll
    for (int i = 0; i < rows; i++) zt[i] = *reinterpret_cast<const f32x4 *>(zb + (size_t)(i * 4 + b) * C);
    f32x4 y[6];
    w6_slice_outputs(zt, y, CLS);
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < npx; k++) {
        const int py = 4 * ty + (along_row ? W6_SLICE_ROW(CLS, b) : k), cx = along_row ? k : b;
        float h0 = 0.f, h1 = 0.f, h2 = 0.f;
        if (py < N && cx < wd) {
            f32x4 v = y[k] + X.bi;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                v[r] = v[r] > 0.f ? v[r] : 0.f;
                h0 = __builtin_fmaf(v[r], X.hw0[r], h0);
                h1 = __builtin_fmaf(v[r], X.hw1[r], h1);
                h2 = __builtin_fmaf(v[r], X.hw2[r], h2);
            }
        }
        h0 = w4_wave_sum_row3(h0); h1 = w4_wave_sum_row3(h1); h2 = w4_wave_sum_row3(h2);
        if ((lane & 15) == k) { m0 = h0; m1 = h1; m2 = h2; }
    }
    const int k = lane & 15, py = 4 * ty + (along_row ? W6_SLICE_ROW(CLS, b) : k), cx = along_row ? k : b, px = x0 + cx;
    if (lane >= 48 && k < npx && py < N && cx < wd) {
        const int p = py * N + px;
        const float y0 = m0 + X.hd.pc_b[0], y1 = m1 + X.hd.pc_b[1];
        X.hd.ph[(size_t)bd * (2 * N * N) + p] = y0 > 0.f ? y0 : 0.f;
        X.hd.ph[(size_t)bd * (2 * N * N) + N * N + p] = y1 > 0.f ? y1 : 0.f;
        const float vv = (m2 + X.hd.vc_b) * X.hd.vn_scale[p] + X.hd.vn_shift[p];
        X.hd.vh[(size_t)bd * (N * N) + p] = vv > 0.f ? vv : 0.f;
    }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void out_lo_kernel(const float *__restrict__ M, const float *__restrict__ bias,
                                                                                            const float *__restrict__ res, long total_tiles, C6HeadFuse hd, int pad) {
    extern __shared__ float lds_pad[];      // (a request only: it decides how many of these workgroups a CU takes)
    if (pad == 12345) lds_pad[threadIdx.x] = 0.f;
    constexpr int C = W6_C;
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    W6OutCtx X;
    X.bi = *reinterpret_cast<const f32x4 *>(bias + lane * 4);
    X.hw0 = *reinterpret_cast<const f32x4 *>(hd.w + lane * 4);
    X.hw1 = *reinterpret_cast<const f32x4 *>(hd.w + C + lane * 4);
    X.hw2 = *reinterpret_cast<const f32x4 *>(hd.w + 2 * C + lane * 4);
    X.res = res;
    X.hd = hd;
    const W6Map map(total_tiles / W6_T);
#pragma unroll 1
    for (long it = wave0; it < 4 * total_tiles; it += nwaves) {      // item = one slice of a tile
        const long t = it >> 2;
        const int b = (int)(it & 3);
        const long bd = t / W6_T;
        const int tt = (int)(t - bd * W6_T), ty = tt / W6_TWX, tx = tt - ty * W6_TWX, cls = w6_cls(ty, tx);
        const int x0 = w6_x0(tx), wd = w6_width(tx);
        const float *zb = M + (size_t)map.slot(bd, ty, tx) * W6_ZTILE + lane * 4;
        if (cls == W6_CLS_D) out1_tile<W6_CLS_D>(X, zb, bd, ty, x0, wd, lane, b);
        else if (cls == W6_CLS_DP) out1_tile<W6_CLS_DP>(X, zb, bd, ty, x0, wd, lane, b);
        else if (cls == W6_CLS_A) out1_tile<W6_CLS_A>(X, zb, bd, ty, x0, wd, lane, b);
        else if (cls == W6_CLS_B) out1_tile<W6_CLS_B>(X, zb, bd, ty, x0, wd, lane, b);
        else out1_tile<W6_CLS_BP>(X, zb, bd, ty, x0, wd, lane, b);
    }
}
// a stand-in for the boundary kernel BETWEEN two layers (Z in, V out): per step a wave reads 8 KiB, applies FM packed multiply-adds per float4 and
// writes 12 KiB (V is 1.47 x Z), 1-KiB runs per wave instruction; persistent grid, 4 waves per workgroup, <= 96 registers
template <int FM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void xcopy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n4, int pad) {
    extern __shared__ float lds_pad[];
    if (pad == 12345) lds_pad[threadIdx.x] = 0.f;
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    const f32x4 m = {1.0000001f, 0.9999999f, 1.0000002f, 0.9999998f};
#pragma unroll 1
    for (size_t i = wave * 512 + lane; i + 448 < n4; i += nw * 512) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = src[i + k * 64];
        f32x4 o[12];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            f32x4 t = v[k & 7];
#pragma unroll
            for (int r = 0; r < FM; r++) t = __builtin_elementwise_fma(t, m, v[(k + r + 1) & 7]);
            o[k] = t;
        }
        const size_t ob = (i / 512) * 768 + lane;
#pragma unroll
        for (int k = 0; k < 12; k++) dst[ob + k * 64] = o[k];
    }
}
// co-residence check: a workgroup of O's shape that only sleeps for `ticks` of the 100 MHz wall clock
__global__ __launch_bounds__(256) void spin_kernel(long long ticks, int pad) {
    extern __shared__ float lds_pad[];
    if (pad == 12345) lds_pad[threadIdx.x] = 0.f;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void fill_kernel(float *p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((int)(x & 0xffff) - 32768) * (1.f / 65536.f);
    }
}
// ---- power / clock of the card while a kernel mix runs: the hwmon files of the amdgpu driver, sampled by a host thread every 10 ms -----------
struct Sampler {
    std::vector<std::string> pw, fq;
    std::atomic<bool> run{false};
    std::thread th;
    double psum = 0, pmax = 0, fsum = 0, fmin = 1e30; long n = 0;
    static std::vector<std::string> g(const char *pat) { glob_t gl; std::vector<std::string> r; if (glob(pat, 0, nullptr, &gl) == 0) for (size_t i = 0; i < gl.gl_pathc; i++) r.push_back(gl.gl_pathv[i]); globfree(&gl); return r; }
    static double rd(const std::string &f) { std::ifstream i(f); double v = -1; i >> v; return v; }
    Sampler() {
        // the card this process computes on: by its PCI address (the host shows every card's files)
        char bus[64] = {0};
        hipDeviceGetPCIBusId(bus, sizeof bus, 0);
        std::string want(bus);
        for (auto &ch : want) ch = (char)tolower(ch);
        std::string base;
        for (auto &d : g("/sys/class/drm/card*/device")) {
            char real[512];
            if (realpath(d.c_str(), real) && std::string(real).find(want) != std::string::npos) base = d;
        }
        printf("device 0 at %s -> %s\n", bus, base.c_str());
        if (!base.empty()) {
            for (const char *f : {"/hwmon/hwmon*/power1_average", "/hwmon/hwmon*/power1_input"}) { pw = g((base + f).c_str()); if (!pw.empty()) break; }
            fq = g((base + "/hwmon/hwmon*/freq1_input").c_str());
            for (auto &c : g((base + "/hwmon/hwmon*/power1_cap").c_str())) printf("power cap %.0f W\n", rd(c) / 1e6);
        }
    }
    void start() { psum = pmax = fsum = 0; fmin = 1e30; n = 0; run = true; th = std::thread([this] {
        while (run) { double p = 0, f = 0; for (auto &x : pw) p = std::max(p, rd(x)); for (auto &x : fq) f = std::max(f, rd(x));
            psum += p; pmax = std::max(pmax, p); fsum += f; fmin = std::min(fmin, f); n++; std::this_thread::sleep_for(std::chrono::milliseconds(10)); } }); }
    void stop(const char *what, double ms_per_iter) { run = false; th.join(); printf("%-58s %8.3f ms per iteration; power avg %6.0f W max %6.0f W; sclk avg %5.0f MHz min %5.0f MHz (%ld samples)\n", what, ms_per_iter, psum / n / 1e6, pmax / 1e6, fsum / n / 1e6, fmin / 1e6, n); }
};
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const int nboards = argc > 1 ? atoi(argv[1]) : 16384;
    const int out_reps = argc > 2 ? atoi(argv[2]) : 4;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const W6Map map(nboards);
    const int nblocks = (int)map.nblocks;
    const size_t vfl = (size_t)nblocks * W6_VBLOCK, zfl = (size_t)nblocks * W6_ZBLOCK;
    float *V, *Z, *Z2, *U, *bias, *hw, *scal, *ph, *vh;
    CK(hipMalloc(&V, vfl * 4)); CK(hipMalloc(&Z, zfl * 4)); CK(hipMalloc(&Z2, zfl * 4));
    CK(hipMalloc(&U, c6_w6_packed_floats() * 4));
    CK(hipMalloc(&bias, 1024 * 4)); CK(hipMalloc(&hw, 3 * 256 * 4)); CK(hipMalloc(&scal, 4096 * 4));
    CK(hipMalloc(&ph, (size_t)nboards * 2 * 361 * 4)); CK(hipMalloc(&vh, (size_t)nboards * 361 * 4));
    fill_kernel<<<2048, 256>>>(V, vfl, 1u); fill_kernel<<<2048, 256>>>(Z2, zfl, 2u); fill_kernel<<<256, 256>>>(U, c6_w6_packed_floats(), 3u);
    fill_kernel<<<4, 256>>>(bias, 1024, 4u); fill_kernel<<<4, 256>>>(hw, 768, 5u); fill_kernel<<<16, 256>>>(scal, 4096, 6u);
    CK(hipDeviceSynchronize());
    const C6HeadFuse hd{hw, scal, 0.1f, scal + 8, scal + 512, ph, vh};
    const long tiles = (long)nboards * W6_T;
    constexpr size_t gsmem = (size_t)2 * W6_BT * (W6_C + 4) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gsmem));
    hipStream_t sg, so; CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&so, hipStreamNonBlocking));
    const size_t osmem = 24 * 1024;
    // every block treated as a D' block (rows of five positions, 40 per tile): the instantiation that leaves registers free
    auto gemm = [&]() { hipLaunchKernelGGL(c6_w6_gemm_kernel<true, false, false>, dim3(cus), dim3(512), gsmem, sg, V, U, Z, 0, nblocks, (int)map.blkB, (int)map.blkBP, (int)map.blkD); };
    auto outp = [&](int grid) { for (int r = 0; r < out_reps; r++) hipLaunchKernelGGL(out_lo_kernel, dim3(grid), dim3(256), osmem, so, Z2, bias, (const float *)nullptr, tiles, hd, 0); };
    auto outfull = [&]() { for (int r = 0; r < out_reps; r++) hipLaunchKernelGGL(c6_w6_out_kernel, dim3(cus * 8), dim3(256), 0, so, Z2, bias, (const float *)nullptr, tiles, hd); };
    auto timeit = [&](auto &&f) {
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            f();
            hipStreamSynchronize(sg); hipStreamSynchronize(so);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        return best;
    };
    printf("CUs %d, %d boards, %d blocks; out kernel x %d\n", cus, nboards, nblocks, out_reps);
    if (argc > 3 && atoi(argv[3]) == 1) {      // power / clock under each kernel mix, ~2.5 s each
        Sampler S;
        float *V2; CK(hipMalloc(&V2, zfl * 4 * 3 / 2 + (1 << 20)));
        const size_t n4 = zfl / 4;
        auto loop = [&](const char *what, auto &&f) {
            hipDeviceSynchronize();
            S.start();
            auto t0 = std::chrono::steady_clock::now(); int it = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) { f(); hipStreamSynchronize(sg); hipStreamSynchronize(so); it++; }
            S.stop(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / it);
        };
        loop("idle (host sleeps 50 ms)", [&] { std::this_thread::sleep_for(std::chrono::milliseconds(50)); });
        loop("G: c6_w6_gemm_kernel<true, false, false>, all blocks", [&] { gemm(); });
        loop("O: c6_w6_out_kernel x reps (reads Z)", [&] { outfull(); });
        loop("X: read-transform-write stand-in x reps", [&] { for (int r = 0; r < out_reps; r++) hipLaunchKernelGGL(xcopy_kernel<6>, dim3(cus * 8), dim3(256), 0, so, (const f32x4 *)Z2, (f32x4 *)V2, n4, 0); });
        loop("G then X, one stream's worth (a layer: matrix phase + HBM phase)", [&] { gemm(); hipStreamSynchronize(sg); for (int r = 0; r < 2; r++) hipLaunchKernelGGL(xcopy_kernel<6>, dim3(cus * 8), dim3(256), 0, so, (const f32x4 *)Z2, (f32x4 *)V2, n4, 0); });
        return 0;
    }

    const double tg = timeit([&] { gemm(); });
    const double flops = 2.0 * 40.0 * nblocks * W6_BT * 256.0 * 256.0;
    printf("G alone (c6_w6_gemm_kernel<true, false, false>, all blocks as rows of five): %8.3f ms  %6.1f TFLOP/s\n", tg, flops / tg / 1e9);
    const double zbytes = (double)tiles * 32 * 1024 * out_reps;
    const double tof = timeit([&] { outfull(); });
    printf("the product's c6_w6_out_kernel (8 workgroups per CU) alone:     %8.3f ms  %5.2f TB/s\n", tof, zbytes / tof / 1e9);
    for (int per : {1, 2}) {
        const double to = timeit([&] { outp(cus * per); });
        printf("O alone (<= 96 registers, %d workgroup(s) of 4 waves per CU):    %8.3f ms  %5.2f TB/s\n", per, to, zbytes / to / 1e9);
    }
    const double to1 = timeit([&] { outp(cus); });
    const double t_og = timeit([&] { outp(cus); gemm(); });
    printf("O first (older), then G:  together %8.3f ms  (sum %8.3f, max %8.3f)\n", t_og, tg + to1, tg > to1 ? tg : to1);
    const double t_go = timeit([&] { gemm(); outp(cus); });
    printf("G first, then O (younger): together %8.3f ms  (sum %8.3f, max %8.3f)\n", t_go, tg + to1, tg > to1 ? tg : to1);
    // O started, G follows 2 ms later (so that O's workgroups are certainly resident first)
    const double t_og2 = timeit([&] { outp(cus); std::this_thread::sleep_for(std::chrono::microseconds(300)); gemm(); });
    printf("O first, G 0.3 ms later:  together %8.3f ms\n", t_og2);
    // ---- do a 4-wave workgroup (24 KB of LDS) and G share a CU at all?  A sleeper of 20 ms on every CU first, then G: 20 ms if they do, 32 if G must wait
    {
        const double ts = timeit([&] { hipLaunchKernelGGL(spin_kernel, dim3(cus), dim3(256), osmem, so, 2000000LL, 0); });
        const double tsg = timeit([&] { hipLaunchKernelGGL(spin_kernel, dim3(cus), dim3(256), osmem, so, 2000000LL, 0); std::this_thread::sleep_for(std::chrono::microseconds(300)); gemm(); });
        printf("sleeper alone %8.3f ms; sleeper first, then G: %8.3f ms (co-resident: ~%.1f, not: ~%.1f)\n", ts, tsg, ts, ts + tg);
        const double tsg8 = timeit([&] { hipLaunchKernelGGL(spin_kernel, dim3(cus * 4), dim3(256), osmem, so, 2000000LL, 0); std::this_thread::sleep_for(std::chrono::microseconds(300)); gemm(); });
        printf("4 sleepers per CU first, then G: %8.3f ms (G cannot fit beside four of them on a CU: LDS)\n", tsg8);
    }
    // ---- the stand-in for a boundary kernel between two layers: reads Z2 (x reps), writes 1.5 x as much into V2
    {
        float *V2; CK(hipMalloc(&V2, zfl * 4 * 3 / 2 + (1 << 20)));
        const size_t n4 = zfl / 4;
        auto xc = [&](int grid) { for (int r = 0; r < out_reps; r++) hipLaunchKernelGGL(xcopy_kernel<6>, dim3(grid), dim3(256), osmem, so, (const f32x4 *)Z2, (f32x4 *)V2, n4, 0); };
        const double xbytes = (double)zfl * 4 * 2.5 * out_reps;
        for (int per : {1, 2, 8}) {
            const double tx = timeit([&] { xc(cus * per); });
            printf("X alone (read 8 KiB / 6 packed FMAs per float4 / write 12 KiB; %d workgroup(s) of 4 waves per CU): %8.3f ms  %5.2f TB/s\n", per, tx, xbytes / tx / 1e9);
        }
        const double tx1 = timeit([&] { xc(cus); });
        const double t_xg = timeit([&] { xc(cus); gemm(); });
        printf("X first (older), then G:  together %8.3f ms  (sum %8.3f, max %8.3f)\n", t_xg, tg + tx1, tg > tx1 ? tg : tx1);
        const double t_gx = timeit([&] { gemm(); xc(cus); });
        printf("G first, then X (younger): together %8.3f ms  (sum %8.3f, max %8.3f)\n", t_gx, tg + tx1, tg > tx1 ? tg : tx1);
        const double t_xg2 = timeit([&] { xc(cus); std::this_thread::sleep_for(std::chrono::microseconds(300)); gemm(); });
        printf("X first, G 0.3 ms later:  together %8.3f ms\n", t_xg2);
    }
    return 0;
}
