// mfma_f32_gemmlike.hip -- which FP32 MFMA shape sustains more under the Winograd GEMM's operand traffic?  Not part of the library.
// Per workgroup 8 waves; per "position" a 64 x 256 slab of V is read from LDS by every wave (ds_read_b128) and each wave streams its 32 KB of
// weights from a 9 MB L2-resident array (global_load_dwordx4), exactly the bytes per FLOP of c6_wino4_gemm_kernel, without its stores and DMA.
//   A: v_mfma_f32_32x32x2_f32, wave tile 32 couts x 64 tiles = 2 accumulators of 16 registers (the kernel's form)
//   B: v_mfma_f32_16x16x4_f32, the same wave tile = 8 accumulators of 4 registers
// hipcc --offload-arch=gfx950 -O3 mfma_f32_gemmlike.hip -o mfma_f32_gemmlike && ./mfma_f32_gemmlike
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int C = 256, CP = C + 4, TILES = 64, NPOS = 36;

template <bool SMALL>
__global__ __launch_bounds__(512) void probe(const float *__restrict__ U, const float *__restrict__ Vsrc, float *out, int sweeps, unsigned long long *clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [64][CP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < TILES * C; k += 512) lds[(k / C) * CP + k % C] = Vsrc[(size_t)(blockIdx.x % 64) * TILES * C + k];
    __syncthreads();
    f32x16 a32[2];
    f32x4 a16[8];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 16; j++) a32[i][j] = 0.f;
    for (int i = 0; i < 8; i++) a16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *ub = U + (size_t)wave * NPOS * 8192;      // [wave][pos][group][lane][4]
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int sw = 0; sw < sweeps; sw++)
        for (int p = 0; p < NPOS; p++) {
            const float *up = ub + (size_t)p * 8192 + lane * 4;
            if (!SMALL) {
                const float *vb = lds + (lane & 31) * CP + 4 * (lane >> 5);
#pragma unroll 8
                for (int u = 0; u < 32; u++) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(up + u * 256);
                    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(vb + 8 * u), v1 = *reinterpret_cast<const f32x4 *>(vb + 32 * CP + 8 * u);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        a32[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], v0[r], a32[0], 0, 0, 0);
                        a32[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], v1[r], a32[1], 0, 0, 0);
                    }
                }
            } else {
                const float *vb = lds + (lane & 15) * CP + 4 * (lane >> 4);
#pragma unroll 4
                for (int u = 0; u < 16; u++) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(up + (2 * u) * 256), w1 = *reinterpret_cast<const f32x4 *>(up + (2 * u + 1) * 256);
                    f32x4 v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = *reinterpret_cast<const f32x4 *>(vb + 16 * j * CP + 16 * u);
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            a16[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], v[j][r], a16[j], 0, 0, 0);
                            a16[4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], v[j][r], a16[4 + j], 0, 0, 0);
                        }
                }
            }
        }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 16; j++) s += a32[i][j];
    for (int i = 0; i < 8; i++) s += a16[i][0] + a16[i][1] + a16[i][2] + a16[i][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char **argv) {
    const size_t un = (size_t)8 * NPOS * 8192, vn = (size_t)64 * TILES * C;
    std::vector<float> hu(un), hv(vn);
    unsigned x = 12345u;
    for (auto &f : hu) { x = x * 1664525u + 1013904223u; f = ((int)(x >> 8) - (1 << 23)) * (0.04f / (1 << 24)); }
    for (auto &f : hv) { x = x * 1664525u + 1013904223u; f = ((int)(x >> 8) - (1 << 23)) * (4.0f / (1 << 24)); }
    float *U, *V, *out; unsigned long long *clk;
    hipMalloc(&U, un * 4); hipMalloc(&V, vn * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 16);
    hipMemcpy(U, hu.data(), un * 4, hipMemcpyHostToDevice); hipMemcpy(V, hv.data(), vn * 4, hipMemcpyHostToDevice);
    const size_t smem = (size_t)TILES * CP * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int sweeps = argc > 1 ? atoi(argv[1]) : 40;
    const double flops = 2.0 * 256 * 256 * TILES * NPOS * sweeps * 256;
    for (int rep = 0; rep < (argc > 2 ? atoi(argv[2]) : 2); rep++)
        for (int small = 0; small < 2; small++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](int sw) {
                if (small) hipLaunchKernelGGL(probe<true>, dim3(256), dim3(512), smem, 0, U, V, out, sw, clk);
                else hipLaunchKernelGGL(probe<false>, dim3(256), dim3(512), smem, 0, U, V, out, sw, clk);
            };
            launch(4); hipDeviceSynchronize();
            hipEventRecord(e0); launch(sweeps); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("%s: %.2f ms  %.1f TFLOP/s  in-kernel clock %.2f GHz\n", small ? "16x16x4 (8 x 4 regs) " : "32x32x2 (2 x 16 regs)", ms, flops / (ms * 1e-3) / 1e12,
                   (double)h[0] / (double)h[1] * 0.1);
        }
    return 0;
}
