// conv_bf16x3.hip -- K4, "split" arithmetic of the 3x3 C->C convolution (C6_PREC_BF16X3): FP32 operands are split into two bf16
// numbers (x = hi + mid + O(2^-17 x)) and every product is evaluated as  x_hi*w_hi + x_hi*w_mid + x_mid*w_hi  on the bf16 matrix
// cores (v_mfma_f32_16x16x32_bf16) with FP32 accumulation.  bf16 x bf16 products are exact in FP32, so the only error is
// the dropped cross terms: <= 3 * 2^-16 |x w| per product (tests hold the whole net to the same 1e-5 as the FP32 kernels;
// measured 2e-7 on the value, 4e-9 on the policy).  Three bf16 MFMAs replace one FP32 MFMA at 16x the rate.
//
// Arithmetic: AlphaZero/cpp_weighted/model.hpp:64-74 (conv3x3 pad 1 + folded BatchNorm + residual + ReLU), NHWC FP32
// activations in HBM; the split happens while a 32-channel slice is staged into LDS.
// Schedule ("N-split", shared with conv_f32n.hip / conv_m8.hip): persistent workgroups, one per CU, walking over boards;
//  * wave w owns output channels [32w, 32w+32) for EVERY pixel of the board, so its weights are private -> they stream
//    L2 -> VGPR (prefetched one step ahead); no weight staging in LDS, hence one barrier per 32-channel slice (8 per board);
//  * LDS holds two input buffers: the next slice is loaded (LDS-DMA), split and written in small portions behind the MFMAs
//    of the current one (no synchronous staging phase);
//  * M runs over a (N+1)-wide row grid (x = N is the zero column shared with the next row's x = -1), so that a pixel's LDS
//    slot is linear in m: every A address is  base(lane) + immediate(m-tile, tap, k);
//  * weights are the MFMA A operand (D = W x X^T): a lane's accumulators are 16-byte NHWC runs, the residual is the
//    accumulators' initial value, the epilogue is bias + ReLU + 16-byte stores.
// (Earlier forms -- an M x N wave grid with weight stages in LDS, and the same N-split schedule on 32x32x16 tiles -- were
// A/B partners in round 1 and are gone; their numbers are in DESIGN.md 5.)
#include "c6_device.h"
#include "c6_engine.h"

#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// c6_conv3x3_b3s_kernel: the N-split schedule on v_mfma_f32_16x16x32_bf16 ("small tiles").
//  Under a bf16-MFMA load on real data the chip is clock-limited (DVFS), and the clock it holds depends on the MFMA
//  shape: tools/mfma_calib.hip measures the SAME instruction mix from registers at 1.82-1.90 PFLOP/s (1.84-1.88 GHz) in
//  the 32x32x16 shape and 2.15-2.28 PFLOP/s (2.09-2.28 GHz) in 16x16x32.  A wave owns 32
//  output channels x all pixels (2 x PT accumulator tiles of 4 registers = the same 192 VGPRs at N = 19), one K step is a
//  whole 32-channel slice of one tap, a pixel fragment pair (hi, mid) read from LDS feeds 6 MFMAs (96 cycles, as before),
//  the weights of a step are 2 tiles x 2 planes = 16 VGPRs, prefetched one step ahead.
//  LDS slot = 160 bytes: with lane = (pixel j = l & 15, k-block l >> 4) reading 16 bytes at slot(j) + 16 kb, 160 is the
//  stride below 256 for which all four 16-lane groups of ds_read_b128 hit 16 distinct 4-bank groups (144 gives 2-way).
//  D = W x X^T: lane (j, q = l >> 4) of tile (ct, pt) holds channels 32 wave + 16 ct + 4 q + 0..3 of pixel row
//  16 pt + j -> 16-byte NHWC runs, lanes q = 0..3 completing 64 bytes; residual = initial accumulator value.
// ------------------------------------------------------------------------------------------------------------------
#define B3S_SLOT 80
template <int C, int N>
struct B3SGeom {
    static constexpr int W = N + 1;
    static constexpr int MROWS = N * W;
    static constexpr int PT = (MROWS + 15) / 16;           // 24 @19 (384 rows), 15 @15 (240 rows: no padding), 6 @9
    static constexpr int SLOTS = ((PT * 16 + 2 * W + 2 + 1) + 7) & ~7;
    static constexpr int IN_UNITS = SLOTS * B3S_SLOT;
    static constexpr int WAVES = C / 32;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int NITEM = (N * N * 8 + THREADS - 1) / THREADS;   // 16-byte staging items per thread and slice
    static constexpr int IPS = (NITEM + 5) / 6;                         // items per trickle step
    static constexpr size_t LDS_BYTES = (size_t)2 * IN_UNITS * 2 + (size_t)THREADS * IPS * 16;   // + LDS-DMA landing area
};

// SPLIT = false: plain bf16 products (x_hi * w_hi only, one MFMA instead of three) -- C6_PREC_BF16, a RELAXED-tolerance
// evaluator (net outputs ~1e-3, not the 1e-5 parity grade), SURVEY 8f rank 4
// HEAD = true (the tower's last convolution in the evaluator): the activations are not stored; the epilogue applies the
// three 1x1 head filters (policy conv 2 channels + value conv 1 channel, model.hpp:80,86) to the wave's 32 channels and
// writes per-wave partial sums head_part[board][filter][wave][GEMM row] into `out` (3 * C/32 * PT*16 floats per board, 10 %
// of an activation) -- c6_head_reduce_kernel adds the C/32 partials in wave order.  Saves the 1 KB per pixel the head
// convolution would read back and the store of an activation nobody else reads.
template <int C, int N, bool SPLIT = true, bool HEAD = false>
__global__ __launch_bounds__(64 * (C / 32)) void c6_conv3x3_b3s_kernel(const float *__restrict__ in, const unsigned short *__restrict__ wp,
                                                                       const float *__restrict__ bias, const float *__restrict__ res,
                                                                       float *__restrict__ out, int nboards, int relu,
                                                                       const float *__restrict__ head_w) {
    using G = B3SGeom<C, N>;
    constexpr int W = G::W, PT = G::PT, KS = C / 32, THREADS = G::THREADS, WAVES = G::WAVES, CELLS = N * N;
    constexpr int PH0 = (PT + 1) / 2;                      // a tap's pixel tiles run in two halves (trickle sub-stages)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short *lds0 = smem16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x, bstep = gridDim.x;
    const int nlocal = b0 < nboards ? (nboards - b0 + bstep - 1) / bstep : 0;
    const int total = nlocal * KS;

    for (int k = tid; k < G::IN_UNITS; k += THREADS) reinterpret_cast<uint32_t *>(lds0)[k] = 0u;   // both buffers, halo = 0

    auto item_load = [&](const float *slice, int k) -> f32x4 {
        int item = tid + k * THREADS;
        item = item < CELLS * 8 ? item : CELLS * 8 - 1;
        return *reinterpret_cast<const f32x4 *>(slice + (size_t)(item >> 3) * C + (item & 7) * 4);
    };
    auto item_store = [&](unsigned short *buf, int k, f32x4 v) {
        const int item = tid + k * THREADS;
        if (item < CELLS * 8) {
            const int p = item >> 3, q = item & 7;
            bf16x4 hi, mid;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                hi[r] = (__bf16)v[r];
                mid[r] = (__bf16)(v[r] - (float)hi[r]);
            }
            unsigned short *dst = buf + ((p / N + 1) * W + (p % N + 1)) * B3S_SLOT + q * 4;
            *reinterpret_cast<bf16x4 *>(dst) = hi;
            if (SPLIT) *reinterpret_cast<bf16x4 *>(dst + 32) = mid;
        }
    };
    constexpr int NITEM = G::NITEM, IPS = G::IPS;
    // landing area of the trickle's LDS-DMA: [wave][item slot][lane] x 16 bytes, behind the two input buffers
    float *stg = reinterpret_cast<float *>(smem16 + 2 * G::IN_UNITS) + (size_t)wave * IPS * 256;

    // weights of this wave: [stage = ks*9 + tap][wave][ct][plane][lane][8]
    auto wptr = [&](int stage, int ct, int plane) -> const bf16x8 * {
        return reinterpret_cast<const bf16x8 *>(wp + ((((size_t)stage * WAVES + wave) * 2 + ct) * 2 + plane) * 512) + lane;
    };

    if (total == 0) return;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < NITEM; k++) item_store(lds0, k, item_load(in + (size_t)b0 * (size_t)(CELLS * C), k));   // very first slice
    bf16x8 whi[2], wmid[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) { whi[ct] = *wptr(0, ct, 0); wmid[ct] = SPLIT ? *wptr(0, ct, 1) : whi[ct]; }

#pragma unroll 1
    for (int lb = 0; lb < nlocal; lb++) {
    // opaque per board: otherwise LICM hoists the row offsets of the unrolled tile loops out of the board loop
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int pj = lane_o & 15, n4 = wave * 32 + 4 * (lane_o >> 4);
    auto row_off = [&](int pt, bool &ok) -> size_t {
        const int m = pt * 16 + pj, y = m / W, x = m % W;
        ok = x < N && y < N;
        return (size_t)((ok ? y : 0) * N + (ok ? x : 0)) * C + n4;      // clamped: loads stay unconditional
    };
    const size_t b = (size_t)(b0 + lb * bstep);
    f32x4 acc[2][PT];
    if (res) {      // the residual is the accumulators' initial value: all loads in flight at once
        const float *resb = res + b * (size_t)CELLS * C;
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
            bool ok;
            const float *rp = resb + row_off(pt, ok);
            acc[0][pt] = *reinterpret_cast<const f32x4 *>(rp);
            acc[1][pt] = *reinterpret_cast<const f32x4 *>(rp + 16);
        }
    } else {
#pragma unroll
        for (int pt = 0; pt < PT; pt++) acc[0][pt] = acc[1][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int ks = 0; ks < KS; ks++) {
        const int gs = lb * KS + ks;
        __syncthreads();   // slice gs complete in its buffer; everybody has left the other buffer
        const unsigned short *lds_in = lds0 + (gs & 1) * G::IN_UNITS;
        unsigned short *lds_nx = lds0 + ((gs + 1) & 1) * G::IN_UNITS;
        const bool more = gs + 1 < total;
        const float *nslice = in + (size_t)(b0 + (ks + 1 < KS ? lb : lb + 1) * bstep) * (size_t)(CELLS * C) + (ks + 1 < KS ? (ks + 1) * 32 : 0);
        int abase = (lane & 15) * B3S_SLOT + 8 * (lane >> 4);   // tap 0 = top-left neighbour of row m = j; k-block l >> 4
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
            // prefetch the weights of the next step (of the next board's first slice at a board boundary)
            bf16x8 nhi[2], nmid[2];
            {
                int ntap = tap + 1, nks = ks;
                if (ntap == 9) { ntap = 0; nks++; }
                if (nks == KS) nks = 0;      // wraps to the next board's first stage (one harmless extra load at the very end)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) { nhi[ct] = *wptr(nks * 9 + ntap, ct, 0); nmid[ct] = SPLIT ? *wptr(nks * 9 + ntap, ct, 1) : nhi[ct]; }
            }
#pragma unroll
            for (int half = 0; half < 2; half++) {
                // the next input slice trickles in: item group k leaves global memory at sub-stage 3k by LDS-DMA
                // (global_load_lds_dwordx4: per-lane gather, no VGPRs -- a register-held copy was being spilled across the
                // MFMAs, and that scratch traffic showed up as +30 % HBM bytes) and is split + written to its slots at 3k + 2.
                // vmcnt completes in order: exactly the four weight loads of one step are younger than the DMA by then.
                if (more) {
                    const int sidx = tap * 2 + half, k = sidx / 3, ph = sidx % 3;
                    if (ph == 0) {
#pragma unroll
                        for (int q = 0; q < IPS; q++)
                            if (k * IPS + q < NITEM) {
                                int item = tid + (k * IPS + q) * THREADS;
                                item = item < CELLS * 8 ? item : CELLS * 8 - 1;
                                // Written as asm: through the builtin hipcc treats every later ds_read as a possible reader of
                                // the DMA's LDS bytes and puts s_waitcnt vmcnt(0) in front of the next pixel-fragment reads --
                                // the "asynchronous" copy was waited for on the spot (HBM latency, 6 times per slice).  The
                                // landing area is read only at ph == 2, behind the explicit vmcnt wait below.
                                const float *gp = nslice + (size_t)(item >> 3) * C + (item & 7) * 4;
                                // LDS byte address of this wave's landing slot (the dynamic segment starts at 0: no static LDS here)
                                const int lds_off = __builtin_amdgcn_readfirstlane((int)((2 * G::IN_UNITS) * 2 + (wave * IPS + q) * 1024));
                                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lds_off) : "memory", "m0");
                            }
                    } else if (ph == 2) {
                        if (SPLIT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // four weight loads per step are younger ...
                        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");             // ... two without the mid planes
#pragma unroll
                        for (int q = 0; q < IPS; q++)
                            if (k * IPS + q < NITEM) item_store(lds_nx, k * IPS + q, *reinterpret_cast<const f32x4 *>(stg + q * 256 + lane * 4));
                    }
                }
                // pixel fragments double-buffered by hand (sched_barrier keeps hipcc from serialising read -> wait -> MFMA)
                bf16x8 phi[2], pmid[2];
                auto load_p = [&](int slot, int pt) {
                    const unsigned short *ap = lds_in + abase + (pt * 16) * B3S_SLOT;
                    phi[slot] = *reinterpret_cast<const bf16x8 *>(ap);
                    if (SPLIT) pmid[slot] = *reinterpret_cast<const bf16x8 *>(ap + 32);
                };
                const int p_lo = half == 0 ? 0 : PH0, p_hi = half == 0 ? PH0 : PT;
                load_p(0, p_lo);
#pragma unroll
                for (int pt = p_lo, sl = 0; pt < p_hi; pt++, sl ^= 1) {
                    if (pt + 1 < p_hi) load_p(sl ^ 1, pt + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ct = 0; ct < 2; ct++) {
                        if (SPLIT) {
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[ct], pmid[sl], acc[ct][pt], 0, 0, 0);
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmid[ct], phi[sl], acc[ct][pt], 0, 0, 0);
                        }
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[ct], phi[sl], acc[ct][pt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ct++) { whi[ct] = nhi[ct]; wmid[ct] = nmid[ct]; }
            abase += (tap % 3 == 2 ? W - 2 : 1) * B3S_SLOT;   // next tap: +1 slot, or the start of the next row
        }
    }   // ks

        // epilogue: + bias, ReLU, 16-byte stores straight from the accumulators (rows with x == N are junk); the stores
        // drain behind the next board's first slice
        float *outb = out + b * (size_t)CELLS * C;
        // bias == nullptr and relu == 0: the training path (plain convolution: pre-BatchNorm output, input gradient)
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 bi0 = bias ? *reinterpret_cast<const f32x4 *>(bias + n4) : zero4, bi1 = bias ? *reinterpret_cast<const f32x4 *>(bias + n4 + 16) : zero4;
        const float floor_ = relu ? 0.f : -INFINITY;
        if (HEAD) {
#pragma unroll
            for (int pt = 0; pt < PT; pt++) {      // activations in place of the accumulators
                f32x4 y0 = acc[0][pt] + bi0, y1 = acc[1][pt] + bi1;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    y0[r] = y0[r] > floor_ ? y0[r] : floor_;
                    y1[r] = y1[r] > floor_ ? y1[r] : floor_;
                }
                acc[0][pt] = y0;
                acc[1][pt] = y1;
            }
            float *hp = out + b * (size_t)(3 * WAVES * PT * 16) + (size_t)wave * (PT * 16) + pj;
#pragma unroll 1
            for (int o = 0; o < 3; o++) {
                const f32x4 h0 = *reinterpret_cast<const f32x4 *>(head_w + o * C + n4), h1 = *reinterpret_cast<const f32x4 *>(head_w + o * C + n4 + 16);
#pragma unroll
                for (int pt = 0; pt < PT; pt++) {
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; r++) sum = __builtin_fmaf(acc[0][pt][r], h0[r], sum);
#pragma unroll
                    for (int r = 0; r < 4; r++) sum = __builtin_fmaf(acc[1][pt][r], h1[r], sum);
                    sum += __shfl_xor(sum, 16, 64);      // the four channel quarters of the pixel row
                    sum += __shfl_xor(sum, 32, 64);
                    if (lane_o < 16) hp[(size_t)o * (WAVES * PT * 16) + pt * 16] = sum;
                }
            }
            continue;
        }
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
            bool ok;
            float *op = outb + row_off(pt, ok);
            if (ok) {
                f32x4 y0 = acc[0][pt] + bi0, y1 = acc[1][pt] + bi1;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    y0[r] = y0[r] > floor_ ? y0[r] : floor_;
                    y1[r] = y1[r] > floor_ ? y1[r] : floor_;
                }
                *reinterpret_cast<f32x4 *>(op) = y0;
                *reinterpret_cast<f32x4 *>(op + 16) = y1;
            }
        }
    }
}

static inline unsigned short f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf16_to_f32(unsigned short h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

size_t c6_conv_packed_b3s_units(int C) { return (size_t)9 * (C / 32) * (C / 32) * 2 * 2 * 512; }

// w: folded OIHW [C][C][3][3] -> [stage = ks*9 + tap][wave][ct][plane][lane][j]
//    value = split(W[n = wave*32 + ct*16 + (lane & 15)][k = ks*32 + (lane >> 4)*8 + j][tap]), bf16 bits
void c6_pack_conv_weights_b3s(int C, const float *w, unsigned short *packed) {
    const int KS = C / 32, WAVES = C / 32;
    for (int ks = 0; ks < KS; ks++)
        for (int t = 0; t < 9; t++)
            for (int wv = 0; wv < WAVES; wv++)
                for (int ct = 0; ct < 2; ct++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int n = wv * 32 + ct * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + j;
                            const float x = w[((size_t)n * C + k) * 9 + t];
                            const unsigned short hi = f32_to_bf16_rne(x);
                            const unsigned short mid = f32_to_bf16_rne(x - bf16_to_f32(hi));
                            const size_t base = ((((size_t)(ks * 9 + t) * WAVES + wv) * 2 + ct) * 2) * 512;
                            packed[base + 0 * 512 + lane * 8 + j] = hi;
                            packed[base + 1 * 512 + lane * 8 + j] = mid;
                        }
}

template <int C, int N, bool SPLIT, bool HEAD>
static void launch_b3s_k(int nboards, int grid, const float *in, const unsigned short *wp, const float *bias, const float *res, float *out,
                         int relu, const float *head_w, hipStream_t s) {
    using G = B3SGeom<C, N>;
    const size_t smem = G::LDS_BYTES;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_b3s_kernel<C, N, SPLIT, HEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((c6_conv3x3_b3s_kernel<C, N, SPLIT, HEAD>), dim3(grid), dim3(G::THREADS), smem, s, in, wp, bias, res, out, nboards, relu, head_w);
}

template <int C, int N>
static void launch_b3s(int nboards, int grid, const float *in, const unsigned short *wp, const float *bias, const float *res, float *out,
                       int relu, int split, const float *head_w, hipStream_t s) {
    if (head_w) {
        if (split) launch_b3s_k<C, N, true, true>(nboards, grid, in, wp, bias, res, out, relu, head_w, s);
        else launch_b3s_k<C, N, false, true>(nboards, grid, in, wp, bias, res, out, relu, head_w, s);
    } else {
        if (split) launch_b3s_k<C, N, true, false>(nboards, grid, in, wp, bias, res, out, relu, nullptr, s);
        else launch_b3s_k<C, N, false, false>(nboards, grid, in, wp, bias, res, out, relu, nullptr, s);
    }
}

// GEMM rows per board of the head partials written by the HEAD instantiation (PT * 16)
int c6_conv_b3s_head_rows(int N) { return ((N * (N + 1) + 15) / 16) * 16; }

// head_w != nullptr: [3][C] head filters; `out` receives the head partials instead of the activations (see the kernel)
bool c6_conv_b3s_dispatch(int C, int N, int nboards, const float *in, const unsigned short *wp, const float *bias, const float *res,
                          float *out, int num_cus, hipStream_t s, int relu, int split, const float *head_w) {
    if (nboards <= 0) return true;
    const int per_cu = C == 256 ? 1 : (C == 128 ? 2 : 4);
    const int grid = nboards < num_cus * per_cu ? nboards : num_cus * per_cu;
#define B3SCASE(CC, NN) if (C == CC && N == NN) { launch_b3s<CC, NN>(nboards, grid, in, wp, bias, res, out, relu, split, head_w, s); return true; }
    B3SCASE(256, 19) B3SCASE(256, 15) B3SCASE(256, 9)
    B3SCASE(128, 19) B3SCASE(128, 15) B3SCASE(128, 9)
    B3SCASE(64, 19) B3SCASE(64, 15) B3SCASE(64, 9)
#undef B3SCASE
    return false;
}
