// conv_v2.hip -- K4, second generation of the 3x3 C->C convolution (FP32 MFMA implicit GEMM).
//
// Same arithmetic as c6_conv3x3_kernel (net_kernels.hip): AlphaZero/cpp_weighted/model.hpp:64-74,
// conv3x3 pad 1 + folded BatchNorm + (residual) + ReLU.  What changed is the schedule:
//
//  * activations are stored slice-major ("S32"): [board][C/32][pixel][32 channels] f32, so the
//    32-channel slice of one board is ONE contiguous block (N*N*128 B) that LDS-DMA
//    (global_load_lds_dwordx4) copies asynchronously with no registers, while the MFMAs of the
//    previous slice run;
//  * the slice is transposed LDS->LDS into a channel-major, zero-haloed grid  lds_in[ch][slot]
//    (slot = (y+1)*(N+2)+(x+1), row stride == 17 mod 32 banks): an A operand read is then
//    `ds_read_b32 v, addr offset:imm` with the tap AND the channel in the immediate -- no VALU;
//  * tiles are 16x16x4 (v_mfma_f32_16x16x4_f32): N*N = 361 rows pad to 368 instead of 384;
//  * wave w owns output channels [32w, 32w+32) for ALL pixels (23 x 2 tiles, 184 accumulator
//    VGPRs), so the B operand (weights) is private to the wave: it is loaded straight from L2 into
//    VGPRs, prefetched one 8-channel block ahead -- no weight staging in LDS, and only two barriers
//    per 32-channel slice instead of two per tap;
//  * the grid is persistent (one 64*C/32-thread workgroup per CU walks over boards), the next
//    board's first slice is in flight during the epilogue, and the accumulators start from
//    bias (+ residual) so that the epilogue is ReLU + store only.
#include "c6_device.h"
#include "c6_engine.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__host__ __device__ constexpr int v2_slots(int N) {   // (N+2)^2 rounded up to == 17 (mod 32)
    const int p = (N + 2) * (N + 2);
    int s = p;
    while (s % 32 != 17) s++;
    return s;
}

template <int C, int N>
struct V2Geom {
    static constexpr int CELLS = N * N;
    static constexpr int P = N + 2;
    static constexpr int MT = (CELLS + 15) / 16;          // 16-row m-tiles
    static constexpr int WAVES = C / 32;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int KS = C / 32;
    static constexpr int SLOTS = v2_slots(N);
    static constexpr int IN_FLOATS = 32 * SLOTS;
    static constexpr int PIECES = (CELLS * 8 + 63) / 64;  // 1-KiB DMA pieces per slice
    static constexpr int STG_FLOATS = PIECES * 256;
    static constexpr size_t LDS_BYTES = (size_t)(IN_FLOATS + STG_FLOATS) * 4;
};

template <int C, int N, bool RES>
__global__ __launch_bounds__(64 * (C / 32)) void c6_conv3x3_v2_kernel(const float *__restrict__ in, const float *__restrict__ wp,
                                                                      const float *__restrict__ bias, const float *__restrict__ res,
                                                                      float *__restrict__ out, int nboards) {
    using G = V2Geom<C, N>;
    constexpr int CELLS = G::CELLS, P = G::P, MT = G::MT, WAVES = G::WAVES, THREADS = G::THREADS, KS = G::KS, SLOTS = G::SLOTS;
    constexpr int MTA = (MT + 1) / 2;   // m-tiles are processed in two groups to bound the A-fragment registers
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *lds_in = smem;                       // [32][SLOTS]
    float *stg = smem + G::IN_FLOATS;           // [PIECES * 64][4]  == [pixel][32] as in global memory

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kk = lane >> 4;

    // A-operand address of m-tile mt, lane row i: pixel p = mt*16 + i sits at slot(p) = p + 2*(p / N) + P + 1.  Inside one
    // m-tile the row index p / N takes at most two values, so
    //   address = abase + [(mt*16 + 2*((mt*16) / N)) * 4 as an immediate] + 8 * wrap(mt)
    // with abase = (kk*SLOTS + i)*4 + tap offset and wrap(mt) = 1 iff pixel i of the tile is already on the next row.
    // (16 consecutive pixels span two rows when N >= 16 and up to three on smaller boards: wrap2)
    uint32_t wrapbits = 0, wrap2bits = 0;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int p = mt * 16 + i;
        const int dr = p < CELLS ? p / N - (mt * 16) / N : 0;
        wrapbits |= (uint32_t)(dr >= 1) << mt;
        wrap2bits |= (uint32_t)(dr >= 2) << mt;
    }
    int abase = (kk * SLOTS + i) * 4;   // tap 0 == top-left neighbour; advanced tap by tap

    for (int k = tid; k < G::IN_FLOATS; k += THREADS) lds_in[k] = 0.f;   // the halo stays zero for the whole kernel

    const int n0 = wave * 32;
    const int b0 = blockIdx.x, bstep = gridDim.x;
    const int nlocal = b0 < nboards ? (nboards - b0 + bstep - 1) / bstep : 0;   // boards of this workgroup
    const int total = nlocal * KS;                                             // slices, numbered g = local_board * KS + ks
    auto issue_dma = [&](int g) {
        const int b = b0 + (g / KS) * bstep, ks = g % KS;
        const float *src = in + ((size_t)b * KS + ks) * (size_t)(CELLS * 32);
        for (int pc = wave; pc < G::PIECES; pc += WAVES) {
            const float *gp = src + (size_t)(pc * 64 + lane) * 4;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)gp, (lds_void_t *)(stg + pc * 256), 16, 0, 0);
        }
    };
    auto lds_a = [&](int mt, int koff, int base0, int base8, int base16) -> float {   // mt, koff: compile-time after unrolling
        const int tile_off = (mt * 16 + 2 * ((mt * 16) / N)) * 4;
        int addr = ((wrapbits >> mt) & 1u) ? base8 : base0;   // 1 v_cndmask per read; the bases come through an opaque asm so
        if constexpr (N < 16) addr = ((wrap2bits >> mt) & 1u) ? base16 : addr;   // that MT addresses are never kept live
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(lds_in) + addr + tile_off + koff);
    };
    f32x4 acc[MT][2];
    auto epilogue = [&](int b) {   // C/D layout of 16x16: col = lane & 15, row = 4 * (lane >> 4) + r
        float *ob = out + ((size_t)b * KS + wave) * (size_t)(CELLS * 32);
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int p = mt * 16 + 4 * kk + r;
                if (p < CELLS) {
                    const float y0 = acc[mt][0][r], y1 = acc[mt][1][r];
                    ob[p * 32 + i] = y0 > 0.f ? y0 : 0.f;
                    ob[p * 32 + 16 + i] = y1 > 0.f ? y1 : 0.f;
                }
            }
    };
    if (total > 0) issue_dma(0);

#pragma unroll 1
    for (int g = 0; g < total; g++) {
        const int b = b0 + (g / KS) * bstep, ks = g % KS;
        // ---- slice boundary: the DMA of slice g has landed in stg; transpose it into lds_in ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int item = tid; item < CELLS * 8; item += THREADS) {
            const int p = item >> 3, q = item & 7;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(stg + item * 4);
            float *dst = lds_in + (4 * q) * SLOTS + (p / N + 1) * P + (p % N + 1);
            dst[0] = v[0];
            dst[SLOTS] = v[1];
            dst[2 * SLOTS] = v[2];
            dst[3 * SLOTS] = v[3];
        }
        __syncthreads();
        if (g + 1 < total) issue_dma(g + 1);   // streams behind the MFMAs of this slice
        if (ks == 0) {
            if (g > 0) epilogue(b - bstep);    // previous board: stores drain behind this board's first slice
            // accumulators start from bias (+ residual): the epilogue is then ReLU + store
            const float bi0 = bias[n0 + i], bi1 = bias[n0 + 16 + i];
            const float *rb = RES ? res + ((size_t)b * KS + wave) * (size_t)(CELLS * 32) : nullptr;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int p = mt * 16 + 4 * kk + r;
                    float v0 = bi0, v1 = bi1;
                    if (RES && p < CELLS) { v0 += rb[p * 32 + i]; v1 += rb[p * 32 + 16 + i]; }
                    acc[mt][0][r] = v0;
                    acc[mt][1][r] = v1;
                }
        }
        f32x2 bcur[2], bnxt[2];
        {
            const f32x2 *w2 = reinterpret_cast<const f32x2 *>(wp + (size_t)(ks * 9) * (32 * C) + (size_t)wave * 1024);
            bcur[0] = w2[lane];
            bcur[1] = w2[64 + lane];
        }
        // A fragments are software-pipelined: while the MFMAs of one half of the m-tiles run, the reads of the next half
        // (next 4-channel step / tap) are already in flight, so a wave hides its own LDS latency even when the two waves
        // of a SIMD run in lockstep.
        float a0[MTA], a1[MT - MTA];
        auto load_group = [&](int grp, int koff, int base0, int base8, int base16) {
            (void)base16;
            if (grp == 0) {
#pragma unroll
                for (int mt = 0; mt < MTA; mt++) a0[mt] = lds_a(mt, koff, base0, base8, base16);
            } else {
#pragma unroll
                for (int mt = MTA; mt < MT; mt++) a1[mt - MTA] = lds_a(mt, koff, base0, base8, base16);
            }
        };
        {
            int c0 = abase, c8 = abase + 8, c16 = abase + 16;
            asm volatile("" : "+v"(c0), "+v"(c8), "+v"(c16));
            load_group(0, 0, c0, c8, c16);
        }
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
            // address delta to the next tap: +1 slot, or to the start of the next row, or back to tap 0
            const int delta = (tap == 8 ? -(2 * P + 2) : (tap % 3 == 2 ? P - 2 : 1)) * 4;
#pragma unroll
            for (int k8 = 0; k8 < 4; k8++) {
                // prefetch the next 8-channel weight block of this wave: [stage][wave][k8][tile][lane][2]
                {
                    int nk8 = k8 + 1, ntap = tap;
                    if (nk8 == 4) { nk8 = 0; ntap = tap + 1; }
                    if (ntap < 9) {
                        const f32x2 *w2 = reinterpret_cast<const f32x2 *>(wp + (size_t)(ks * 9 + ntap) * (32 * C) + (size_t)wave * 1024 + nk8 * 256);
                        bnxt[0] = w2[lane];
                        bnxt[1] = w2[64 + lane];
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const int koff = (k8 * 8 + s * 4) * SLOTS * 4;   // channel block -> immediate offset
                    int c0 = abase, c8 = abase + 8, c16 = abase + 16;
                    asm volatile("" : "+v"(c0), "+v"(c8), "+v"(c16));
                    load_group(1, koff, c0, c8, c16);                // second half of this step, behind the MFMAs below
#pragma unroll
                    for (int mt = 0; mt < MTA; mt++) {
                        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mt], bcur[0][s], acc[mt][0], 0, 0, 0);
                        acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mt], bcur[1][s], acc[mt][1], 0, 0, 0);
                    }
                    // first half of the NEXT step (possibly of the next tap)
                    {
                        const bool last = (k8 == 3 && s == 1);
                        const int nkoff = last ? 0 : (k8 * 8 + s * 4 + 4) * SLOTS * 4;
                        int d0 = abase + (last ? delta : 0);
                        int d8 = d0 + 8, d16 = d0 + 16;
                        asm volatile("" : "+v"(d0), "+v"(d8), "+v"(d16));
                        load_group(0, nkoff, d0, d8, d16);
                    }
#pragma unroll
                    for (int mt = MTA; mt < MT; mt++) {
                        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt - MTA], bcur[0][s], acc[mt][0], 0, 0, 0);
                        acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt - MTA], bcur[1][s], acc[mt][1], 0, 0, 0);
                    }
                }
                bcur[0] = bnxt[0];
                bcur[1] = bnxt[1];
            }
            abase += delta;
        }
    }
    if (total > 0) epilogue(b0 + (nlocal - 1) * bstep);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// w: folded OIHW [C][C][3][3] -> [stage = ks*9 + tap][wave][k8][tile j][lane][s]
//    value = W[n = wave*32 + j*16 + (lane & 15)][k = ks*32 + k8*8 + s*4 + (lane >> 4)][tap]
void c6_pack_conv_weights_v2(int C, const float *w, float *packed) {
    const int KS = C / 32, WAVES = C / 32;
    for (int ks = 0; ks < KS; ks++)
        for (int t = 0; t < 9; t++)
            for (int wv = 0; wv < WAVES; wv++)
                for (int k8 = 0; k8 < 4; k8++)
                    for (int j = 0; j < 2; j++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int s = 0; s < 2; s++) {
                                const int n = wv * 32 + j * 16 + (lane & 15);
                                const int k = ks * 32 + k8 * 8 + s * 4 + (lane >> 4);
                                const size_t dst = ((((((size_t)(ks * 9 + t) * WAVES + wv) * 4 + k8) * 2 + j) * 64 + lane) * 2) + s;
                                packed[dst] = w[((size_t)n * C + k) * 9 + t];
                            }
}

template <int C, int N>
static void launch_v2(int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out, int grid,
                      hipStream_t s) {
    using G = V2Geom<C, N>;
    if (res) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_v2_kernel<C, N, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        hipLaunchKernelGGL((c6_conv3x3_v2_kernel<C, N, true>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, s, in, wp, bias, res, out, nboards);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_v2_kernel<C, N, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        hipLaunchKernelGGL((c6_conv3x3_v2_kernel<C, N, false>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, s, in, wp, bias, res, out, nboards);
    }
}

// activations in S32 layout; returns false for unsupported shapes
bool c6_conv_v2_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out,
                         int num_cus, hipStream_t s) {
    const int grid = nboards < num_cus ? nboards : num_cus;
    if (grid <= 0) return true;
#define V2CASE(CC, NN) if (C == CC && N == NN) { launch_v2<CC, NN>(nboards, in, wp, bias, res, out, grid, s); return true; }
    V2CASE(256, 19) V2CASE(256, 15) V2CASE(256, 9)
    V2CASE(128, 19) V2CASE(128, 15) V2CASE(128, 9)
    V2CASE(64, 19) V2CASE(64, 15) V2CASE(64, 9)
#undef V2CASE
    return false;
}
