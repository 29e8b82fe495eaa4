// conv_bf16x3.hip -- K4, "split" variant of the 3x3 C->C convolution: FP32 operands are split into two bf16 numbers
// (x = hi + mid + O(2^-17 x)) and every product is evaluated as  x_hi*w_hi + x_hi*w_mid + x_mid*w_hi  on the bf16 matrix
// cores (v_mfma_f32_32x32x16_bf16) with FP32 accumulation.  bf16 x bf16 products are exact in FP32, so the only error is
// the dropped cross terms: <= 3 * 2^-16 |x w| per product (tests hold the whole net to the same 1e-5 as the FP32-MFMA
// kernel; measured 2e-7 on the value, 4e-9 on the policy).  Three bf16 MFMAs replace one FP32 MFMA at 16x the rate.
//
// Arithmetic: AlphaZero/cpp_weighted/model.hpp:64-74 (conv3x3 pad 1 + folded BatchNorm + residual + ReLU), same NHWC FP32
// activations in HBM as c6_conv3x3_kernel; the split happens while a 32-channel slice is staged into LDS.
// Structure: one 512-thread workgroup per board, 8 waves = 4 (M) x 2 (N), 3 x 4 tiles of 32x32 per wave, zero-haloed
// (N+2)^2 grid in LDS (taps = constant offsets), weight stages (pre-split on the host) streamed by LDS-DMA one stage
// ahead, LDS-transposed epilogue with 16-byte residual loads / output stores.
#include "c6_device.h"
#include "c6_engine.h"

#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define B3_THREADS 512
#define B3_SLOT 72          // ushort (2-byte) units per LDS pixel slot: 32 hi + 32 mid + 8 pad  (144 bytes)

// ABL: ablation switch used only by c6_bench_conv (0 = the product kernel): 1 no weight DMA, 2 no input staging,
//      3 neither and no barriers, 5 no epilogue stores (accumulators kept alive)
template <int C, int MTW, int ABL = 0>
__global__ __launch_bounds__(B3_THREADS) void c6_conv3x3_bf16x3_kernel(const float *__restrict__ in, const unsigned short *__restrict__ wp,
                                                                       const float *__restrict__ bias, const float *__restrict__ res,
                                                                       float *__restrict__ out, int N) {
    constexpr int NTW = C / 64;
    constexpr int KS = C / 32;
    constexpr int WSTAGE = 2 * 2 * 2 * C * 8;     // ushorts per weight stage: [plane][g][h][n][8]  (32 KiB at C = 256)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    const int cells = N * N, P = N + 2;
    const int in_units = P * P * B3_SLOT;
    unsigned short *lds_in = smem16;
    unsigned short *lds_w = smem16 + ((in_units + 7) & ~7);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const size_t b = blockIdx.x;
    const float *inb = in + b * (size_t)cells * C;

    int aoff[MTW];   // ushort units
#pragma unroll
    for (int mt = 0; mt < MTW; mt++) {
        int p = (wm * MTW + mt) * 32 + i;
        if (p >= cells) p = 0;
        aoff[mt] = ((p / N + 1) * P + (p % N + 1)) * B3_SLOT + 8 * h;
    }

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[mt][nt][v] = 0.f;

    for (int k = tid; k < in_units / 2; k += B3_THREADS) reinterpret_cast<uint32_t *>(lds_in)[k] = 0u;   // zero halo

    constexpr int WPIECES = WSTAGE * 2 / 1024;     // 1-KiB LDS-DMA pieces per stage
    auto dma_w = [&](int stage, int buf) {
        const unsigned short *src = wp + (size_t)stage * WSTAGE;
        unsigned short *dst = lds_w + buf * WSTAGE;
        for (int pc = wave; pc < WPIECES; pc += B3_THREADS / 64)
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)(src + pc * 512 + lane * 8),
                                             (__attribute__((address_space(3))) void *)(dst + pc * 512), 16, 0, 0);
    };

    if (ABL != 1 && ABL != 3) dma_w(0, 0);
    for (int ks = 0; ks < KS; ks++) {
        if (ABL != 3) __syncthreads();
        // stage + split the 32-channel input slice: x -> hi = bf16(x), mid = bf16(x - hi).  The loads of a thread go out in
        // batches of SB before the first conversion of the batch: ceil(6 / SB) HBM round trips per slice instead of 6.
        if (ABL != 2 && ABL != 3) {
            constexpr int SB = 3;
#pragma unroll 1
            for (int k0 = 0; k0 < 6; k0 += SB) {
                f32x4 v[SB];
#pragma unroll
                for (int k = 0; k < SB; k++) {
                    // unconditional loads (clamped): a branch around each load would make hipcc wait vmcnt(0) per element
                    int item = tid + (k0 + k) * B3_THREADS;
                    item = item < cells * 8 ? item : cells * 8 - 1;
                    v[k] = *reinterpret_cast<const f32x4 *>(inb + (size_t)(item >> 3) * C + ks * 32 + (item & 7) * 4);
                }
#pragma unroll
                for (int k = 0; k < SB; k++) {
                    const int item = tid + (k0 + k) * B3_THREADS;
                    if (item < cells * 8) {
                        const int p = item >> 3, q = item & 7;
                        bf16x4 hi, mid;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            hi[r] = (__bf16)v[k][r];
                            mid[r] = (__bf16)(v[k][r] - (float)hi[r]);
                        }
                        unsigned short *dst = lds_in + ((p / N + 1) * P + (p % N + 1)) * B3_SLOT + q * 4;
                        *reinterpret_cast<bf16x4 *>(dst) = hi;
                        *reinterpret_cast<bf16x4 *>(dst + 32) = mid;
                    }
                }
            }
        }
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
            const int stage = ks * 9 + tap, buf = stage & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ABL != 3) __syncthreads();
            if (ABL != 1 && ABL != 3) if (stage + 1 < KS * 9) dma_w(stage + 1, buf ^ 1);
            const int toff = ((tap / 3 - 1) * P + (tap % 3 - 1)) * B3_SLOT;
            const unsigned short *wb = lds_w + buf * WSTAGE;
#pragma unroll
            for (int g = 0; g < 2; g++) {       // two 16-deep k steps per 32-channel slice
                bf16x8 ahi[MTW], amid[MTW];
#pragma unroll
                for (int mt = 0; mt < MTW; mt++) {
                    const unsigned short *ap = lds_in + aoff[mt] + toff + g * 16;
                    ahi[mt] = *reinterpret_cast<const bf16x8 *>(ap);
                    amid[mt] = *reinterpret_cast<const bf16x8 *>(ap + 32);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) {
                    const int n = (wn * NTW + nt) * 32 + i;
                    const bf16x8 bhi = *reinterpret_cast<const bf16x8 *>(wb + (((0 * 2 + g) * 2 + h) * C + n) * 8);
                    const bf16x8 bmid = *reinterpret_cast<const bf16x8 *>(wb + (((1 * 2 + g) * 2 + h) * C + n) * 8);
#pragma unroll
                    for (int mt = 0; mt < MTW; mt++) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid[mt], bhi, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[mt], bmid, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[mt], bhi, acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    if (ABL == 5) {   // keep the accumulators alive without the store traffic
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < MTW; mt++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++)
#pragma unroll
                for (int v = 0; v < 16; v++) sum += acc[mt][nt][v];
        if (sum == 123.456f) out[tid] = sum;
        return;
    }
    // epilogue: transpose each 32x32 tile through a wave-private LDS patch -> 16-byte residual loads / stores
    __syncthreads();
    constexpr int PSTR = 36;
    float *patch = reinterpret_cast<float *>(smem16) + wave * (32 * PSTR);
    float *outb = out + b * (size_t)cells * C;
    const float *resb = res ? res + b * (size_t)cells * C : nullptr;
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const int n4 = (wn * NTW + nt) * 32 + c4;
        const f32x4 bi = *reinterpret_cast<const f32x4 *>(bias + n4);
#pragma unroll
        for (int mt = 0; mt < MTW; mt++) {
#pragma unroll
            for (int v = 0; v < 16; v++) patch[((v & 3) + 8 * (v >> 2) + 4 * h) * PSTR + i] = acc[mt][nt][v];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = j * 8 + r8;
                const int p = (wm * MTW + mt) * 32 + row;
                f32x4 y = *reinterpret_cast<const f32x4 *>(patch + row * PSTR + c4);
                if (p < cells) {
                    y = y + bi;
                    if (resb) y = y + *reinterpret_cast<const f32x4 *>(resb + (size_t)p * C + n4);
#pragma unroll
                    for (int q = 0; q < 4; q++) y[q] = y[q] > 0.f ? y[q] : 0.f;
                    *reinterpret_cast<f32x4 *>(outb + (size_t)p * C + n4) = y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// "N-split" variant (c6_conv3x3_b3n_kernel): wave w owns output channels [32w, 32w+32) for EVERY pixel of the board.
//  * its weights are private -> they stream L2 -> VGPR (one 16-byte load per 16-deep step and plane, prefetched one step
//    ahead); no weight staging in LDS, hence NO per-tap barrier: one barrier per 32-channel slice (8 per board, not 72);
//  * the LDS that the weight stages occupied holds a second input buffer: the next slice is loaded, split and written in
//    small portions behind the MFMAs of the current one (no synchronous staging phase);
//  * M runs over a 20-wide row grid (x = 19 is the zero column shared with the next row's x = -1), so that a pixel's LDS
//    slot is linear in m: every A address is  base(lane) + immediate(m-tile, tap, k).  19*20 = 380 -> 12 tiles of 32
//    rows, the same 384 rows the other kernels pad to.
// ------------------------------------------------------------------------------------------------------------------
template <int C, int N>
struct B3NGeom {
    static constexpr int W = N + 1;                        // row stride in slots
    static constexpr int MROWS = N * W;                    // m = y*W + x, x == N is junk
    static constexpr int MT = (MROWS + 31) / 32;
    static constexpr int SLOTS = ((MT * 32 + 2 * W + 2 + 1) + 7) & ~7;   // every tap of every (junk) row stays inside
    static constexpr int IN_UNITS = SLOTS * B3_SLOT;       // ushorts per input buffer
    static constexpr int WAVES = C / 32;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr size_t LDS_BYTES = (size_t)2 * IN_UNITS * 2;
};

template <int C, int N>
__global__ __launch_bounds__(64 * (C / 32)) void c6_conv3x3_b3n_kernel(const float *__restrict__ in, const unsigned short *__restrict__ wp,
                                                                       const float *__restrict__ bias, const float *__restrict__ res,
                                                                       float *__restrict__ out, int nboards) {
    using G = B3NGeom<C, N>;
    constexpr int W = G::W, MT = G::MT, KS = C / 32, THREADS = G::THREADS, WAVES = G::WAVES, CELLS = N * N;
    constexpr int MG = 1;                                  // m-tiles whose A fragments are live at once (x2: double-buffered)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short *lds0 = smem16;                         // two input buffers, then the epilogue patches

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // persistent: this workgroup walks boards b0, b0 + grid, ...; no workgroup turnover (whose store drain and relaunch
    // left the CU's wave slots empty ~20 % of the time), and the next board's first slice streams in behind the last
    // slice of the current one.  Slices are numbered gs = local_board * KS + ks; slice gs lives in buffer gs & 1.
    const int b0 = blockIdx.x, bstep = gridDim.x;
    const int nlocal = b0 < nboards ? (nboards - b0 + bstep - 1) / bstep : 0;
    const int total = nlocal * KS;

    for (int k = tid; k < G::IN_UNITS; k += THREADS) reinterpret_cast<uint32_t *>(lds0)[k] = 0u;   // both buffers, halo = 0

    auto item_load = [&](const float *slice /* board base + ks*32 */, int k) -> f32x4 {
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
            unsigned short *dst = buf + ((p / N + 1) * W + (p % N + 1)) * B3_SLOT + q * 4;
            *reinterpret_cast<bf16x4 *>(dst) = hi;
            *reinterpret_cast<bf16x4 *>(dst + 32) = mid;
        }
    };
    constexpr int NITEM = (CELLS * 8 + THREADS - 1) / THREADS;   // staging items per thread and slice
    constexpr int IPS = (NITEM + 5) / 6;                         // items per trickle step (6 steps of 3 sub-stages per slice)

    // weights of this wave: [stage = ks*9 + tap][wave][g][plane][lane][8]
    auto wptr = [&](int stage, int g, int plane) -> const bf16x8 * {
        return reinterpret_cast<const bf16x8 *>(wp + ((((size_t)stage * WAVES + wave) * 2 + g) * 2 + plane) * 512) + lane;
    };

    if (total == 0) return;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < NITEM; k++) item_store(lds0, k, item_load(in + (size_t)b0 * (size_t)(CELLS * C), k));   // very first slice
    bf16x8 bhi = *wptr(0, 0, 0), bmid = *wptr(0, 0, 1);
    f32x4 pend[IPS];
#pragma unroll
    for (int q = 0; q < IPS; q++) pend[q] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int lb = 0; lb < nlocal; lb++) {
    // opaque per board: otherwise LICM hoists the row offsets of the unrolled tile loops out of the board loop and
    // they sit in (spilled) registers across the whole MFMA loop
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int pi = lane_o & 31, n4 = wave * 32 + 4 * (lane_o >> 5);
    auto row_off = [&](int mt, bool &ok) -> size_t {
        const int m = mt * 32 + pi, y = m / W, x = m % W;
        ok = x < N && y < N;
        return (size_t)((ok ? y : 0) * N + (ok ? x : 0)) * C + n4;      // clamped: loads stay unconditional
    };
    const size_t b = (size_t)(b0 + lb * bstep);
    // acc[mt][4g + r] = channel 32 wave + 8g + 4h + r of pixel row m = 32 mt + i (weights are the MFMA "A" operand), i.e.
    // four 16-byte runs of the NHWC tensors.  The residual is therefore loaded STRAIGHT INTO the accumulators: all 4 MT
    // loads are in flight at once (one HBM latency per board; read tile by tile in the epilogue they were twelve
    // dependent round trips, 10 % of a residual launch).
    f32x16 acc[MT];
    if (res) {
        const float *resb = res + b * (size_t)CELLS * C;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            bool ok;
            const float *rp = resb + row_off(mt, ok);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(rp + 8 * g);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[mt][4 * g + r] = t[r];
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[mt][v] = 0.f;
    }
#pragma unroll 1
    for (int ks = 0; ks < KS; ks++) {
        const int gs = lb * KS + ks;
        __syncthreads();   // slice gs complete in its buffer; everybody has left the other buffer
        const unsigned short *lds_in = lds0 + (gs & 1) * G::IN_UNITS;
        unsigned short *lds_nx = lds0 + ((gs + 1) & 1) * G::IN_UNITS;
        const bool more = gs + 1 < total;
        // where the next slice comes from: same board, next 32 channels -- or the next board's first slice
        const float *nslice = in + (size_t)(b0 + (ks + 1 < KS ? lb : lb + 1) * bstep) * (size_t)(CELLS * C) + (ks + 1 < KS ? (ks + 1) * 32 : 0);
        int abase = i * B3_SLOT + 8 * h;                     // tap 0 = top-left neighbour of row m = i
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int g = 0; g < 2; g++) {
                // prefetch the weights of the next 16-deep step (of the next board's first slice at a board boundary)
                bf16x8 nhi = bhi, nmid = bmid;
                {
                    int ng = g + 1, ntap = tap, nks = ks;
                    if (ng == 2) { ng = 0; ntap++; }
                    if (ntap == 9) { ntap = 0; nks++; }
                    if (nks == KS) nks = 0;      // wraps to the next board's first stage (one harmless extra load at the very end)
                    nhi = *wptr(nks * 9 + ntap, ng, 0);
                    nmid = *wptr(nks * 9 + ntap, ng, 1);
                }
                // the next input slice trickles in: item k is loaded at sub-stage 3k and split + written at 3k + 1
                if (more) {
                    const int sidx = tap * 2 + g, k = sidx / 3, ph = sidx % 3;
                    if (ph == 0) {
#pragma unroll
                        for (int q = 0; q < IPS; q++) if (k * IPS + q < NITEM) pend[q] = item_load(nslice, k * IPS + q);
                    } else if (ph == 2) {
                        // two steps later: vmcnt completes in order, so this (HBM-latency) load must not be needed --
                        // directly or through a wait on a younger weight load -- any earlier
#pragma unroll
                        for (int q = 0; q < IPS; q++) if (k * IPS + q < NITEM) item_store(lds_nx, k * IPS + q, pend[q]);
                    }
                }
                // A fragments are double-buffered by hand: the reads of the next MG m-tiles are issued before the MFMAs of
                // the current ones (sched_barrier keeps hipcc from serialising read -> wait -> MFMA on one register set)
                bf16x8 ahi[2][MG], amid[2][MG];
                auto load_a = [&](int slot, int m0) {
#pragma unroll
                    for (int q = 0; q < MG; q++)
                        if (m0 + q < MT) {
                            const unsigned short *ap = lds_in + abase + ((m0 + q) * 32) * B3_SLOT + g * 16;
                            ahi[slot][q] = *reinterpret_cast<const bf16x8 *>(ap);
                            amid[slot][q] = *reinterpret_cast<const bf16x8 *>(ap + 32);
                        }
                };
                load_a(0, 0);
#pragma unroll
                for (int m0 = 0, sl = 0; m0 < MT; m0 += MG, sl ^= 1) {
                    if (m0 + MG < MT) load_a(sl ^ 1, m0 + MG);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < MG; q++)
                        if (m0 + q < MT) {
                            // weights are the MFMA "A" operand, pixels the "B": D = W x X^T, so a lane ends up holding
                            // 4 x 4 CONSECUTIVE channels of one pixel and the epilogue needs no transposition
                            acc[m0 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bhi, amid[sl][q], acc[m0 + q], 0, 0, 0);
                            acc[m0 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bmid, ahi[sl][q], acc[m0 + q], 0, 0, 0);
                            acc[m0 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bhi, ahi[sl][q], acc[m0 + q], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                bhi = nhi;
                bmid = nmid;
            }
            abase += (tap % 3 == 2 ? W - 2 : 1) * B3_SLOT;   // next tap: +1 slot, or the start of the next row
        }
    }   // ks

        // epilogue of this board: + bias, ReLU, four 16-byte stores per tile straight from the accumulators (lanes i / i+32
        // complete 32-byte sectors; rows with x == N are junk).  The stores drain behind the next board's first slice.
        float *outb = out + b * (size_t)CELLS * C;
        f32x4 bi[4];
#pragma unroll
        for (int g = 0; g < 4; g++) bi[g] = *reinterpret_cast<const f32x4 *>(bias + n4 + 8 * g);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            bool ok;
            float *op = outb + row_off(mt, ok);
            if (ok) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 yv;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float t = acc[mt][4 * g + r] + bi[g][r];
                        yv[r] = t > 0.f ? t : 0.f;
                    }
                    *reinterpret_cast<f32x4 *>(op + 8 * g) = yv;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// "small-tile" N-split variant (c6_conv3x3_b3s_kernel): the b3n kernel on v_mfma_f32_16x16x32_bf16.
//  Under a bf16-MFMA load on real data the chip is clock-limited (DVFS), and the clock it holds depends on the MFMA
//  shape: tools/mfma_calib.hip measures the SAME instruction mix from registers at 1.82-1.90 PFLOP/s (1.84-1.88 GHz) in
//  the 32x32x16 shape and 2.15-2.28 PFLOP/s (2.09-2.28 GHz) in 16x16x32.  Same data movement as b3n: a wave still owns 32
//  output channels x all pixels (2 x PT accumulator tiles of 4 registers = the same 192 VGPRs at N = 19), one K step is a
//  whole 32-channel slice of one tap, a pixel fragment pair (hi, mid) read from LDS feeds 6 MFMAs (96 cycles, as before),
//  the weights of a step are 2 tiles x 2 planes = 16 VGPRs, prefetched one step ahead.
//  LDS slot = 160 bytes: with lane = (pixel j = l & 15, k-block l >> 4) reading 16 bytes at slot(j) + 16 kb, 160 is the
//  stride below 256 for which all four 16-lane groups of ds_read_b128 hit 16 distinct 4-bank groups (144 gives 2-way).
//  D = W x X^T as in b3n: lane (j, q = l >> 4) of tile (ct, pt) holds channels 32 wave + 16 ct + 4 q + 0..3 of pixel row
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

size_t c6_conv_packed_b3_units(int C) { return (size_t)9 * (C / 32) * (2 * 2 * 2 * C * 8); }

// w: folded OIHW [C][C][3][3] -> [stage = ks*9 + tap][plane][g][h][n][j], k = ks*32 + g*16 + h*8 + j, bf16 bits
void c6_pack_conv_weights_b3(int C, const float *w, unsigned short *packed) {
    const int KS = C / 32;
    const size_t WSTAGE = (size_t)2 * 2 * 2 * C * 8;
    for (int ks = 0; ks < KS; ks++)
        for (int t = 0; t < 9; t++)
            for (int g = 0; g < 2; g++)
                for (int h = 0; h < 2; h++)
                    for (int n = 0; n < C; n++)
                        for (int j = 0; j < 8; j++) {
                            const int k = ks * 32 + g * 16 + h * 8 + j;
                            const float x = w[((size_t)n * C + k) * 9 + t];
                            const unsigned short hi = f32_to_bf16_rne(x);
                            const unsigned short mid = f32_to_bf16_rne(x - bf16_to_f32(hi));
                            const size_t base = (size_t)(ks * 9 + t) * WSTAGE;
                            packed[base + ((((size_t)0 * 2 + g) * 2 + h) * C + n) * 8 + j] = hi;
                            packed[base + ((((size_t)1 * 2 + g) * 2 + h) * C + n) * 8 + j] = mid;
                        }
}

template <int C>
static void launch_b3(int MTW, int nboards, size_t smem, const float *in, const unsigned short *wp, const float *bias, const float *res,
                      float *out, int N, hipStream_t s) {
#define B3CASE(M)                                                                                                                          \
    case M:                                                                                                                                \
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_bf16x3_kernel<C, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((c6_conv3x3_bf16x3_kernel<C, M>), dim3(nboards), dim3(B3_THREADS), smem, s, in, wp, bias, res, out, N);          \
        break;
    switch (MTW) { B3CASE(1) B3CASE(2) default: B3CASE(3) }
#undef B3CASE
}

void c6_conv_b3_ablate(int ablate, int nboards, const float *in, const unsigned short *wp, const float *bias, const float *res,
                       float *out, hipStream_t s) {
    const int N = 19, P = N + 2, C = 256;
    const size_t smem = ((size_t)((P * P * B3_SLOT + 7) & ~7) + 2 * (size_t)(2 * 2 * 2 * C * 8)) * sizeof(unsigned short);
#define B3ABL(A) case A: hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_bf16x3_kernel<256, 3, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((c6_conv3x3_bf16x3_kernel<256, 3, A>), dim3(nboards), dim3(B3_THREADS), smem, s, in, wp, bias, res, out, N); break;
    switch (ablate) { B3ABL(1) B3ABL(2) B3ABL(3) B3ABL(5) default: break; }
#undef B3ABL
}

bool c6_conv_b3_dispatch(int C, int N, int nboards, const float *in, const unsigned short *wp, const float *bias, const float *res,
                         float *out, hipStream_t s) {
    const int cells = N * N, MT = (cells + 31) / 32, MTW = (MT + 3) / 4, P = N + 2;
    size_t smem = ((size_t)((P * P * B3_SLOT + 7) & ~7) + 2 * (size_t)(2 * 2 * 2 * C * 8)) * sizeof(unsigned short);
    if (smem < (size_t)8 * 32 * 36 * sizeof(float)) smem = (size_t)8 * 32 * 36 * sizeof(float);
    switch (C) {
    case 64: launch_b3<64>(MTW, nboards, smem, in, wp, bias, res, out, N, s); return true;
    case 128: launch_b3<128>(MTW, nboards, smem, in, wp, bias, res, out, N, s); return true;
    case 256: launch_b3<256>(MTW, nboards, smem, in, wp, bias, res, out, N, s); return true;
    default: return false;
    }
}

size_t c6_conv_packed_b3n_units(int C) { return (size_t)9 * (C / 32) * (C / 32) * 2 * 2 * 512; }

// w: folded OIHW [C][C][3][3] -> [stage = ks*9 + tap][wave][g][plane][lane][j]
//    value = split(W[n = wave*32 + (lane & 31)][k = ks*32 + g*16 + (lane >> 5)*8 + j][tap]), bf16 bits
void c6_pack_conv_weights_b3n(int C, const float *w, unsigned short *packed) {
    const int KS = C / 32, WAVES = C / 32;
    for (int ks = 0; ks < KS; ks++)
        for (int t = 0; t < 9; t++)
            for (int wv = 0; wv < WAVES; wv++)
                for (int g = 0; g < 2; g++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int n = wv * 32 + (lane & 31), k = ks * 32 + g * 16 + (lane >> 5) * 8 + j;
                            const float x = w[((size_t)n * C + k) * 9 + t];
                            const unsigned short hi = f32_to_bf16_rne(x);
                            const unsigned short mid = f32_to_bf16_rne(x - bf16_to_f32(hi));
                            const size_t base = ((((size_t)(ks * 9 + t) * WAVES + wv) * 2 + g) * 2) * 512;
                            packed[base + 0 * 512 + lane * 8 + j] = hi;
                            packed[base + 1 * 512 + lane * 8 + j] = mid;
                        }
}

// w: folded OIHW [C][C][3][3] -> [stage = ks*9 + tap][wave][ct][plane][lane][j]
//    value = split(W[n = wave*32 + ct*16 + (lane & 15)][k = ks*32 + (lane >> 4)*8 + j][tap]), bf16 bits (same size as b3n)
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

template <int C, int N>
static void launch_b3n(int nboards, int grid, const float *in, const unsigned short *wp, const float *bias, const float *res, float *out,
                       hipStream_t s) {
    using G = B3NGeom<C, N>;
    const size_t smem = G::LDS_BYTES;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_b3n_kernel<C, N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((c6_conv3x3_b3n_kernel<C, N>), dim3(grid), dim3(G::THREADS), smem, s, in, wp, bias, res, out, nboards);
}

bool c6_conv_b3n_dispatch(int C, int N, int nboards, const float *in, const unsigned short *wp, const float *bias, const float *res,
                          float *out, int num_cus, hipStream_t s) {
    if (nboards <= 0) return true;
    // persistent grid: one workgroup per CU (small-C workgroups are narrow: several fit a CU)
    const int per_cu = C == 256 ? 1 : (C == 128 ? 2 : 4);
    const int grid = nboards < num_cus * per_cu ? nboards : num_cus * per_cu;
#define B3NCASE(CC, NN) if (C == CC && N == NN) { launch_b3n<CC, NN>(nboards, grid, in, wp, bias, res, out, s); return true; }
    B3NCASE(256, 19) B3NCASE(256, 15) B3NCASE(256, 9)
    B3NCASE(128, 19) B3NCASE(128, 15) B3NCASE(128, 9)
    B3NCASE(64, 19) B3NCASE(64, 15) B3NCASE(64, 9)
#undef B3NCASE
    return false;
}
