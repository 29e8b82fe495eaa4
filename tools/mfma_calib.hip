// mfma_calib.hip -- calibration of the FP32 and bf16 matrix-core ceilings on this MI355X in the shapes the conv kernels use.
// Not part of the library.  hipcc --offload-arch=gfx950 -O3 mfma_calib.hip -o mfma_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool RANDOM>
__global__ __launch_bounds__(512) void k16(float *out, int iters, unsigned long long *clk) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f + 0.5f, b = 1.0f + blockIdx.x * 1e-4f;
    unsigned int lcg = threadIdx.x * 2654435761u + blockIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (RANDOM) {   // full-range random operands: the data the real kernel sees toggles every bit
            lcg = lcg * 1664525u + 1013904223u;
            a = __uint_as_float(0x3f800000u | (lcg >> 9)) - 1.5f;
            b = __uint_as_float(0x3f800000u | ((lcg * 747796405u) >> 9)) - 1.5f;
        }
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if (!RANDOM) a += 1e-6f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(512) void k32(float *out, int iters, unsigned long long *clk) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
    float a = threadIdx.x * 0.001f + 0.5f, b = 1.0f + blockIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// the split-bf16 conv's instruction: v_mfma_f32_32x32x16_bf16, 12 independent accumulator tiles, 3 MFMAs per tile and
// operand pair like the kernel (hi*hi, hi*mid, mid*hi); RANDOM toggles every operand bit each iteration
template <int NACC, bool RANDOM>
__global__ __launch_bounds__(512) void kb16(float *out, int iters, unsigned long long *clk) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
    union U { bf16x8 v; unsigned u[4]; } a0, a1, b0, b1;
    unsigned int lcg = threadIdx.x * 2654435761u + blockIdx.x;
    for (int q = 0; q < 4; q++) { a0.u[q] = 0x3f803f80u + q; a1.u[q] = 0x3c003c00u + q; b0.u[q] = 0x3f003f00u; b1.u[q] = 0x3b803b80u; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (RANDOM) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                lcg = lcg * 1664525u + 1013904223u;
                a0.u[q] = (lcg & 0x807f807fu) | 0x3f003f00u;            // random sign + mantissa, exponent ~1
                a1.u[q] = ((lcg >> 3) & 0x807f807fu) | 0x3b003b00u;
                b0.u[q] = ((lcg >> 5) & 0x807f807fu) | 0x3e803e80u;
                b1.u[q] = ((lcg >> 7) & 0x807f807fu) | 0x3a803a80u;
            }
        }
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0.v, a1.v, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1.v, a0.v, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0.v, a0.v, acc[i], 0, 0, 0);
        }
        if (!RANDOM) a0.u[0] ^= it & 1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// same work in the 16x16x32 shape: 4x as many accumulator tiles of 4 registers (the guide's DVFS item 7: the chip holds a
// higher clock on this shape under load)
template <int NACC, bool RANDOM>
__global__ __launch_bounds__(512) void kb16s(float *out, int iters, unsigned long long *clk) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    union U { bf16x8 v; unsigned u[4]; } a0, a1, b0, b1;
    unsigned int lcg = threadIdx.x * 2654435761u + blockIdx.x;
    for (int q = 0; q < 4; q++) { a0.u[q] = 0x3f803f80u + q; a1.u[q] = 0x3c003c00u + q; b0.u[q] = 0x3f003f00u; b1.u[q] = 0x3b803b80u; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (RANDOM) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                lcg = lcg * 1664525u + 1013904223u;
                a0.u[q] = (lcg & 0x807f807fu) | 0x3f003f00u;
                a1.u[q] = ((lcg >> 3) & 0x807f807fu) | 0x3b003b00u;
                b0.u[q] = ((lcg >> 5) & 0x807f807fu) | 0x3e803e80u;
                b1.u[q] = ((lcg >> 7) & 0x807f807fu) | 0x3a803a80u;
            }
        }
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0.v, a1.v, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1.v, a0.v, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0.v, a0.v, acc[i], 0, 0, 0);
        }
        if (!RANDOM) a0.u[0] ^= it & 1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// candidate for a cheaper 1e-5-grade product (open items): main term x_hi * w_hi in fp16 (11-bit significands)
// on v_mfma_f32_16x16x32_f16, the two cross terms x_hi * w_lo + x_lo * w_hi in MX-scaled fp8 on
// v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128 = [hi | lo] of two 32-channel tap steps).  Per TWO (tile, tap) units:
// 2 f16 MFMAs + 1 fp8 MFMA, against 6 bf16 MFMAs of the bf16x3 scheme.  FP8ONLY / F16ONLY isolate the two instructions.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int NACC, int MODE>      // MODE 0: mix, 1: f16 only (2 per pair), 2: fp8 only (1 per pair)
__global__ __launch_bounds__(512) void kmix(float *out, int iters, unsigned long long *clk) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    union H { f16x8 v; unsigned u[4]; } a0, a1, b0;
    union P { i32x8 v; unsigned u[8]; } pa, pb;
    unsigned int lcg = threadIdx.x * 2654435761u + blockIdx.x;
    for (int q = 0; q < 4; q++) { a0.u[q] = 0x3c003c00u; a1.u[q] = 0x3c003c00u; b0.u[q] = 0x38003800u; }
    for (int q = 0; q < 8; q++) { pa.u[q] = 0x38383838u; pb.u[q] = 0x30303030u; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {      // random sign + mantissa, exponents near 1 (fp16) / near 1 (e4m3)
            lcg = lcg * 1664525u + 1013904223u;
            a0.u[q] = (lcg & 0x83ff83ffu) | 0x38003800u;
            a1.u[q] = ((lcg >> 3) & 0x83ff83ffu) | 0x38003800u;
            b0.u[q] = ((lcg >> 5) & 0x83ff83ffu) | 0x34003400u;
            pa.u[q] = ((lcg >> 2) & 0x87878787u) | 0x30303030u;
            pa.u[q + 4] = ((lcg >> 4) & 0x87878787u) | 0x30303030u;
            pb.u[q] = ((lcg >> 6) & 0x87878787u) | 0x28282828u;
            pb.u[q + 4] = ((lcg >> 7) & 0x87878787u) | 0x28282828u;
        }
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (MODE < 2) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a1.v, acc[i], 0, 0, 0);
            }
            if (MODE == 3) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pb.v, pa.v, acc[i], 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);      // bf6 (e3m2) x bf6
            else if (MODE == 4) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pb.v, pa.v, acc[i], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // fp4 (e2m1) x fp4
            else if (MODE != 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pb.v, pa.v, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename F> void run(const char *name, F launch, int threads, double flop_per_wave_iter, int iters, double peak = 157.3) {
    float *out; unsigned long long *clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(out, iters / 10, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512); hipMemcpy(h.data(), clk, 256 * 16, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / (double)h[1] * 0.1;   // s_memrealtime ticks at 100 MHz
    double tf = flop_per_wave_iter * (threads / 64) * 256.0 * iters / (ms * 1e-3) / 1e12;
    printf("%-28s threads=%d  %.2f ms  %.1f TFLOP/s  in-kernel clock %.2f GHz  (%.1f%% of %.1f)\n", name, threads, ms, tf, ghz, tf / peak * 100.0, peak);
    hipFree(out); hipFree(clk);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    run("16x16x4 46acc 8 waves", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((k16<46, false>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 46 * 2048.0, iters);
    run("16x16x4 46acc 8w RANDOM data", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((k16<46, true>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 46 * 2048.0, iters);
    run("16x16x4 46acc 4 waves", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((k16<46, false>), dim3(256), dim3(256), 0, 0, o, it, c); }, 256, 46 * 2048.0, iters);
    run("32x32x2 12acc 8 waves", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL(k32<12>, dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 12 * 4096.0, iters);
    run("32x32x2 12acc 4 waves", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL(k32<12>, dim3(256), dim3(256), 0, 0, o, it, c); }, 256, 12 * 4096.0, iters);
    run("32x32x2 4acc 4 waves", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL(k32<4>, dim3(256), dim3(256), 0, 0, o, it, c); }, 256, 4 * 4096.0, iters * 3);
    run("bf16 32x32x16 12acc x3 8 waves", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kb16<12, false>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 36 * 32768.0, iters / 4, 2500.0);
    run("bf16 32x32x16 12acc x3 8w RANDOM", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kb16<12, true>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 36 * 32768.0, iters / 4, 2500.0);
    run("bf16 16x16x32 48acc x3 8 waves", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kb16s<48, false>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 144 * 16384.0, iters / 4, 2500.0);
    run("bf16 16x16x32 48acc x3 8w RANDOM", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kb16s<48, true>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 144 * 16384.0, iters / 4, 2500.0);
    run("bf16 32x32x16 12acc x3 8w RANDOM", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kb16<12, true>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 36 * 32768.0, iters / 4, 2500.0);
    run("bf16 16x16x32 48acc x3 8w RANDOM", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kb16s<48, true>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 144 * 16384.0, iters / 4, 2500.0);
    // (tile, tap) units per second: bf16x3 spends 3 MFMAs of 16 384 flop per unit; report the mixed scheme in the SAME unit
    // (algorithmic flop = 16 384 per unit, x3 for comparison with the "executed" bf16x3 figure above)
    run("MIX 2 f16 + 1 fp8(K128) per 2 units", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kmix<48, 0>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 48 * 2 * 3 * 16384.0, iters / 4, 2500.0);
    run("f16 16x16x32 only (2 per pair)", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kmix<48, 1>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 48 * 2 * 16384.0, iters / 4, 2500.0);
    run("fp8 16x16x128 scaled only", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kmix<48, 2>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 48 * 65536.0, iters / 4, 5000.0);
    run("bf6 (e3m2) 16x16x128 scaled only", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kmix<48, 3>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 48 * 65536.0, iters / 4, 10000.0);
    run("fp4 (e2m1) 16x16x128 scaled only", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kmix<48, 4>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 48 * 65536.0, iters / 4, 10000.0);
    run("bf16 16x16x32 48acc x3 8w RANDOM", [](float *o, int it, unsigned long long *c) { hipLaunchKernelGGL((kb16s<48, true>), dim3(256), dim3(512), 0, 0, o, it, c); }, 512, 144 * 16384.0, iters / 4, 2500.0);
    return 0;
}
