// Probe (round 3): can vector-ALU / memory work of OTHER waves of the same workgroup make progress beside waves that issue FP32 MFMAs back to back?
// One workgroup per CU (LDS request forces it): waves 0-7 issue v_mfma_f32_32x32x2_f32 (two independent accumulators, as the Winograd GEMM),
// waves 8..8+X-1 run (A) packed-FMA chains, or (B) a float4 copy of a large buffer with a few packed FMAs per element (what a Winograd
// boundary transform is: HBM bound, little arithmetic).  Times: MFMA waves alone, side waves alone, both.  both ~ max => they overlap;
// both ~ sum => they share the pipe.      hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/probe/mfma_valu_coissue.hip && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int SIDE>   // 0: none, 1: FMA chains, 2: copy (addresses and 8 packed FMAs per float4 on the vector ALU), 3: copy WITHOUT a vector-ALU instruction (r4)
__global__ __launch_bounds__(1024) void probe(int mfma_iters, int side_iters, int do_mfma, int prio, const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                              size_t n4_per_block, float *sink, int side_first, int nm, int bf16) {
    extern __shared__ float lds[];
    const int wave0 = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nside = (blockDim.x >> 6) - nm;
    // r4: side_first puts the side waves at the LOW wave ids of the workgroup (dispatched first: the older waves of every SIMD)
    const int wave = side_first ? (wave0 < nside ? nm + wave0 : wave0 - nside) : wave0;      // < nm: an MFMA wave
    if (wave < nm) {
        if (!do_mfma) return;
        f32x16 a0 = {0}, a1 = {0};
        float x = 1.0f + lane * 1e-3f, y = 0.5f - lane * 1e-3f;
        if (!bf16) {
#pragma unroll 1
            for (int i = 0; i < mfma_iters; i++) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                }
            }
        } else {      // the bf16 matrix instruction of the same tile (its own pipe: 32 cycles per instruction)
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8 xb, yb;
            for (int q = 0; q < 8; q++) { xb[q] = (__bf16)(x + q); yb[q] = (__bf16)(y - q); }
#pragma unroll 1
            for (int i = 0; i < 2 * mfma_iters; i++) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yb, xb, a1, 0, 0, 0);
                }
            }
        }
        float s = 0.f;
        for (int k = 0; k < 16; k++) s += a0[k] + a1[k];
        if (s == 12345.678f) sink[0] = s;
        return;
    }
    if (SIDE == 0) return;
    if (prio) __builtin_amdgcn_s_setprio(3);
    const int sw = wave - nm;
    if (SIDE == 1) {
        f32x2 c0 = {1.f, 2.f}, c1 = {3.f, 4.f}, c2 = {5.f, 6.f}, c3 = {7.f, 8.f};
        const f32x2 m = {1.0000001f, 0.9999999f}, ad = {1e-7f, -1e-7f};
#pragma unroll 1
        for (int i = 0; i < side_iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                c0 = __builtin_elementwise_fma(c0, m, ad); c1 = __builtin_elementwise_fma(c1, m, ad);
                c2 = __builtin_elementwise_fma(c2, m, ad); c3 = __builtin_elementwise_fma(c3, m, ad);
            }
        }
        const float s = c0[0] + c0[1] + c1[0] + c1[1] + c2[0] + c2[1] + c3[0] + c3[1];
        if (s == 12345.678f) sink[1] = s;
    } else if (SIDE == 3) {
        // Round 4: the side waves move bytes as the Winograd GEMM itself does -- LDS-DMA in (global_load_lds_dwordx4: scalar base, one
        // loop-invariant lane offset in a VGPR), ds_read_b128, global_store_dwordx4 with a scalar base -- and issue NO vector-ALU instruction in their
        // loop: bases advance on the scalar ALU, the loop counter is scalar.  8 KiB per wave and step, two LDS slots per wave.
        const int swu = __builtin_amdgcn_readfirstlane(sw), nsu = __builtin_amdgcn_readfirstlane(nside);
        const float *sp = reinterpret_cast<const float *>(src + (size_t)blockIdx.x * n4_per_block) + (size_t)swu * 2048;      // this wave's first 8 KiB
        float *dp = reinterpret_cast<float *>(dst + (size_t)blockIdx.x * n4_per_block) + (size_t)swu * 2048;
        const size_t stride = (size_t)nsu * 2048;                     // floats between two of a wave's steps
        const int steps = __builtin_amdgcn_readfirstlane((int)(n4_per_block * 4 / stride));
        const unsigned voff = (unsigned)lane * 16u;                   // the only vector registers of the loop: set once
        const unsigned lbase = (unsigned)swu * 16384u;                // two 8-KiB slots per side wave
        const unsigned vlds = lbase + voff;
#define DMA8(SP, SLOT)                                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 8; k++)                                                                                           \
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"((SP) + k * 256), "s"(lbase + (SLOT) * 8192u + k * 1024u) : "memory");
        DMA8(sp, 0u)
#pragma unroll 1
        for (int it = 0; it < steps; it++) {
            const unsigned slot = (unsigned)(it & 1);
            const float *spn = sp + stride;
            if (it + 1 < steps) { DMA8(spn, slot ^ 1u) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }      // this step's 8 DMAs have landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            f32x4 v[8];
            // (offsets must be immediates: one unrolled copy per slot and k)
#define RD(K, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(v[K]) : "v"(vlds) : "memory");
            if (slot == 0) { RD(0, 0) RD(1, 1024) RD(2, 2048) RD(3, 3072) RD(4, 4096) RD(5, 5120) RD(6, 6144) RD(7, 7168) }
            else { RD(0, 8192) RD(1, 9216) RD(2, 10240) RD(3, 11264) RD(4, 12288) RD(5, 13312) RD(6, 14336) RD(7, 15360) }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(v[k]), "s"(dp + k * 256) : "memory");
            sp = spn;
            dp += stride;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef RD
#undef DMA8
    } else {
        const f32x4 *sp = src + (size_t)blockIdx.x * n4_per_block;
        f32x4 *dp = dst + (size_t)blockIdx.x * n4_per_block;
        const f32x4 m = {1.0000001f, 0.9999999f, 1.0000002f, 0.9999998f};
        // each side wave streams its share, 8 float4 (8 KiB per wave) in flight
        for (size_t i = (size_t)sw * 64 + lane; i + 7 * (size_t)nside * 64 < n4_per_block; i += (size_t)nside * 64 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = sp[i + (size_t)k * nside * 64];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                f32x4 t = v[k];
#pragma unroll
                for (int r = 0; r < 4; r++) t = __builtin_elementwise_fma(t, m, v[(k + 1) & 7]);      // 8 packed FMAs per float4
                dp[i + (size_t)k * nside * 64] = t;
            }
        }
    }
}

template <int SIDE>
static float run(int grid, int side_waves, int mfma_iters, int side_iters, int do_mfma, int prio, const f32x4 *src, f32x4 *dst, size_t n4pb, float *sink,
                 int side_first = 0, int nm = 8, int bf16 = 0) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 64 * (nm + side_waves);
    const size_t lds = 132 * 1024;      // one workgroup per CU (side waves of kind 3 use 16 KiB each)
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<SIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<SIDE>, dim3(grid), dim3(threads), lds, 0, mfma_iters, side_iters, do_mfma, prio, src, dst, n4pb, sink, side_first, nm, bf16);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const size_t bytes = (size_t)8 << 30;
    f32x4 *src, *dst; float *sink;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 64);
    hipMemset(src, 0, bytes);
    const size_t n4pb = bytes / 16 / cus;
    const int mi = 6000;      // 6000 x 32 MFMAs x 64 cycles / 2.4 GHz ~ 5 ms per wave pair on a SIMD
    printf("CUs %d\n", cus);
    const double mflop = 2.0 * 32 * 32 * 2 * 32.0 * mi * 8 * cus;
    float t_m = run<0>(cus, 0, mi, 0, 1, 0, src, dst, n4pb, sink);
    printf("MFMA waves alone (8 per CU):           %7.3f ms  %6.1f TFLOP/s\n", t_m, mflop / t_m / 1e9);
    for (int sw : {4, 8}) {
        for (int prio : {0, 1}) {
            const int si = 40000;
            float t_a = run<1>(cus, sw, mi, si, 0, prio, src, dst, n4pb, sink);
            float t_b = run<1>(cus, sw, mi, si, 1, prio, src, dst, n4pb, sink);
            printf("side=%d FMA-chain waves prio %d: alone %7.3f ms, with MFMA %7.3f ms (sum %7.3f, max %7.3f)\n", sw, prio, t_a, t_b, t_a + t_m, t_a > t_m ? t_a : t_m);
            float t_e = run<3>(cus, sw, mi, 0, 0, prio, src, dst, n4pb, sink);
            float t_f = run<3>(cus, sw, mi, 0, 1, prio, src, dst, n4pb, sink);
            printf("side=%d copy waves WITHOUT vector-ALU instructions (LDS-DMA in, ds_read, scalar-based stores; 8+8 GB) prio %d:  alone %7.3f ms (%5.2f TB/s), with MFMA %7.3f ms (sum %7.3f, max %7.3f)\n",
                   sw, prio, t_e, 2.0 * bytes / t_e / 1e9, t_f, t_e + t_m, t_e > t_m ? t_e : t_m);
            float t_c = run<2>(cus, sw, mi, 0, 0, prio, src, dst, n4pb, sink);
            float t_d = run<2>(cus, sw, mi, 0, 1, prio, src, dst, n4pb, sink);
            printf("side=%d copy waves (8+8 GB) prio %d:  alone %7.3f ms (%5.2f TB/s), with MFMA %7.3f ms (sum %7.3f, max %7.3f)\n", sw, prio, t_c, 2.0 * bytes / t_c / 1e9,
                   t_d, t_c + t_m, t_c > t_m ? t_c : t_m);
        }
    }
    // ---- round 4: what decides the serialisation? -------------------------------------------------------------------------------------------
    printf("\n# r4 variants (side = 8 copy waves without vector-ALU instructions unless said otherwise)\n");
    {
        float t_e = run<3>(cus, 8, mi, 0, 0, 0, src, dst, n4pb, sink, 1);
        float t_f = run<3>(cus, 8, mi, 0, 1, 0, src, dst, n4pb, sink, 1);
        printf("side waves FIRST in the workgroup (older than the MFMA waves): alone %7.3f ms, with MFMA %7.3f ms (sum %7.3f, max %7.3f)\n", t_e, t_f, t_e + t_m, t_e > t_m ? t_e : t_m);
        float t_c = run<2>(cus, 8, mi, 0, 0, 0, src, dst, n4pb, sink, 1);
        float t_d = run<2>(cus, 8, mi, 0, 1, 0, src, dst, n4pb, sink, 1);
        printf("  ... the plain float4 copy (vector-ALU addresses + FMAs), side first:    alone %7.3f ms, with MFMA %7.3f ms (sum %7.3f, max %7.3f)\n", t_c, t_d, t_c + t_m, t_c > t_m ? t_c : t_m);
    }
    {
        float t_m4 = run<0>(cus, 0, mi, 0, 1, 0, src, dst, n4pb, sink, 0, 4);
        float t_e = run<3>(cus, 8, mi, 0, 0, 0, src, dst, n4pb, sink, 0, 4);
        float t_f = run<3>(cus, 8, mi, 0, 1, 0, src, dst, n4pb, sink, 0, 4);
        printf("4 MFMA waves per CU (one per SIMD): alone %7.3f ms; side alone %7.3f ms, together %7.3f ms (sum %7.3f, max %7.3f)\n", t_m4, t_e, t_f, t_e + t_m4, t_e > t_m4 ? t_e : t_m4);
    }
    {
        float t_mb = run<0>(cus, 0, mi, 0, 1, 0, src, dst, n4pb, sink, 0, 8, 1);
        float t_e = run<3>(cus, 8, mi, 0, 0, 0, src, dst, n4pb, sink, 0, 8, 1);
        float t_f = run<3>(cus, 8, mi, 0, 1, 0, src, dst, n4pb, sink, 0, 8, 1);
        printf("bf16 MFMA waves (v_mfma_f32_32x32x16_bf16, 8 per CU): alone %7.3f ms; side alone %7.3f ms, together %7.3f ms (sum %7.3f, max %7.3f)\n", t_mb, t_e, t_f, t_e + t_mb,
               t_e > t_mb ? t_e : t_mb);
        float t_a = run<1>(cus, 8, mi, 40000, 0, 0, src, dst, n4pb, sink, 0, 8, 1);
        float t_b = run<1>(cus, 8, mi, 40000, 1, 0, src, dst, n4pb, sink, 0, 8, 1);
        printf("  ... beside FMA-chain waves: alone %7.3f ms, together %7.3f ms (sum %7.3f, max %7.3f)\n", t_a, t_b, t_a + t_mb, t_a > t_mb ? t_a : t_mb);
    }
    return 0;
}
