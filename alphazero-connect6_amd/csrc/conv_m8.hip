// conv_m8.hip -- K4, "mixed" variant of the 3x3 C->C convolution (C6_PREC_F16F8): every FP32 product is evaluated as
//
//      x * w  ~=  x_h * w_h                          v_mfma_f32_16x16x32_f16        (fp16 x fp16, exact in FP32)
//               + q8(x_h) * q8(w_l) + q8(x_l) * q8(w_h)   v_mfma_scale_f32_16x16x128_f8f6f4  (OCP e4m3 with E8M0 block scales)
//
// with x_h = fp16(x), x_l = x - x_h (|x_l| <= 2^-11 |x|), likewise w.  The main term carries 11-bit significands, so the
// two cross terms are 2^-11 of it and fp8 (4 significant bits) leaves 2^-15 of a product -- the accuracy class of the
// bf16 split (three bf16 MFMAs per product, conv_bf16x3.hip); the dropped x_l * w_l is 2^-22.  Emulated on the benchmark
// net before the kernel was written (tools/mixed_err.py): policy 8e-9, value 2e-7 against FP32 (bf16 split: 2e-9, 2e-7);
// tests/test_gpu_net.py holds it to the same 1e-5 as the other parity-grade paths.
// Cost: one fp8 MFMA has K = 128 = [q8(x_h) | q8(x_l)] of TWO taps (32 channels each), and runs at twice the bf16/fp16
// rate: per 32-channel slice and accumulator tile 9 fp16 MFMAs + 5 fp8 MFMAs (taps paired (0,1) (2,3) (4,5) (6,7) (8,-))
// = 9 * 16 + 5 * 32 = 304 matrix-pipe cycles against 27 * 16 = 432 of the bf16 split.  tools/mfma_calib.hip: the
// instruction mix sustains 1.66x the bf16 split's rate from registers on random operands (the fp8 MFMA holds 2.40 GHz).
//
// Arithmetic: AlphaZero/cpp_weighted/model.hpp:64-74 (conv3x3 pad 1 + folded BatchNorm + residual + ReLU).
// Schedule = c6_conv3x3_b3s_kernel's: persistent grid, wave w owns output channels [32w, 32w+32) for every pixel (2 x PT
// accumulator tiles of 4 registers), private weights L2 -> VGPR one to two sub-steps ahead, two zero-haloed input buffers on the
// (N+1)-wide row grid, the next slice trickled in by LDS-DMA behind the MFMAs, one barrier per slice, residual = initial
// accumulator, optional head epilogue.  A slice runs as the sub-steps  A0 B0 C0  A1 B1 C1 ... A3 B3 C3  A4 C4:
//   A_p = fp16 MFMAs of tap 2p, B_p = fp16 MFMAs of tap 2p+1, C_p = fp8 MFMAs of the pair (2p, 2p+1).
// LDS slot of a pixel (160 bytes): [x_h fp16 x 32 | q8(x_h 2^-3) x 32 | q8(x_l 2^8) x 32 | 32 pad].
// Operand layout of the fp8 MFMA (found with tools/probe/mfma_fp8_*.hip): lane (row r, group g) holds K = 16 g + i in
// bytes i = 0..15 and K = 64 + 16 g + i in bytes 16..31; the E8M0 scale of (row, K block b = K / 32) is read from lane
// r + 16 b.  Hence groups 0-1 carry tap 2p, groups 2-3 tap 2p+1; bytes 0-15 = q8(x_h) (weights: q8(w_l)), bytes 16-31 =
// q8(x_l) (weights: q8(w_h)), channel 16 (g & 1) + i; lanes 0-31 supply the scale of the first plane, lanes 32-63 of the second
// (the planes are shifted so that one scale pair fits both: see M8_XH_SHIFT below).
#include "c6_device.h"
#include "c6_engine.h"

#include "../../include/connect6_amd.h"

#include <cmath>
#include <cstring>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define M8_SLOT 80          // ushort (2-byte) units per LDS pixel slot (160 bytes: conflict-free stride of the b3s lane mapping)
// block scales (E8M0, bias 127) and the powers of two folded into the fp8 planes
// The fp8 planes hold  q8(x_h 2^-3)  [x up to 3584; beyond that x_h is clamped -- v_cvt_pk_fp8_f32 turns out-of-range values into
// NaN, not 448],  q8(x_l 2^8),  q8(w_l 2^(d+11)),  q8(w_h 2^d)  with a per-layer weight shift d chosen by the packer (max |w_h| 2^d
// just below 448; d sits in the 16-byte header of the packed image).  Both cross terms then carry 2^(d+8), so one pair of E8M0
// scales (2^-4 for the pixels, 2^-(d+4) for the weights) serves every block and lane.
#define M8_XH_SHIFT (-3)
#define M8_XL_SHIFT 8
#define M8_XH_MAX 3584.0f   // 448 * 2^3
#define M8_HEADER 16        // bytes in front of the weight planes: int32 d

// The MFMAs are issued as inline asm with the accumulator TIED (dst = src C).  Through the builtins hipcc picks the three-address
// form, renames accumulator tiles from one MFMA to the next and, at 256 VGPRs, ends up moving and spilling whole tiles inside
// the K loop (20 tile moves + 13 tile spills per tap pair in one instantiation, none in another: chaotic).  A tied tile is one
// read-modify-write live range.  The hazards hipcc would cover with s_nop are covered by hand: the accumulators are first read
// by VALU / stores after the K loop, behind M8_MFMA_DRAIN.
#define M8_MFMA16(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define M8_MFMA8(ACC, A, B, SA, SB) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(ACC) : "v"(A), "v"(B), "v"(SA), "v"(SB))
#define M8_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

template <int C, int N>
struct M8Geom {
    static constexpr int W = N + 1;
    static constexpr int MROWS = N * W;
    static constexpr int PT = (MROWS + 15) / 16;           // 24 @19 (384 rows), 15 @15, 6 @9
    static constexpr int SLOTS = ((PT * 16 + 2 * W + 2 + 1) + 7) & ~7;
    static constexpr int IN_UNITS = SLOTS * M8_SLOT;
    static constexpr int WAVES = C / 32;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int NITEM = (N * N * 8 + THREADS - 1) / THREADS;   // 16-byte staging items per thread and slice
    static constexpr int IPS = (NITEM + 4) / 5;                         // items per trickle step (one step per tap pair)
    static constexpr size_t LDS_BYTES = (size_t)2 * IN_UNITS * 2 + (size_t)THREADS * IPS * 16;   // + LDS-DMA landing area
};

template <int C, int N, bool HEAD = false>
__global__ __launch_bounds__(64 * (C / 32)) void c6_conv3x3_m8_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wp_,
                                                                      const float *__restrict__ bias, const float *__restrict__ res,
                                                                      float *__restrict__ out, int nboards, int relu,
                                                                      const float *__restrict__ head_w, unsigned int *__restrict__ range_flag,
                                                                      float range_hi) {
    using G = M8Geom<C, N>;
    const unsigned char *wp = wp_;
    constexpr int W = G::W, PT = G::PT, KS = C / 32, THREADS = G::THREADS, WAVES = G::WAVES, CELLS = N * N;
    constexpr int PH0 = (PT + 1) / 2;                      // a sub-step's pixel tiles run in two halves
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
    // split four channels of one pixel into the three planes of its slot
    auto item_store = [&](unsigned short *buf, int k, f32x4 v) {
        const int item = tid + k * THREADS;
        if (item < CELLS * 8) {
            const int p = item >> 3, q = item & 7;
            f16x4 h;
            float hf[4], lo[4], hc[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                h[r] = (_Float16)v[r];
                hf[r] = (float)h[r];
                lo[r] = v[r] - hf[r];
                hc[r] = __builtin_amdgcn_fmed3f(hf[r], -M8_XH_MAX, M8_XH_MAX);      // out of range would convert to NaN
            }
            // v_cvt_scalef32_pk_fp8_f32 converts src / scale (tools/probe/cvt_scale_fp8.hip): the shifts cost no multiply
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            s16x2 ha = {0, 0}, hb = {0, 0}, la = {0, 0}, lb = {0, 0};
            ha = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ha, hc[0], hc[1], 8.0f, false);
            hb = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(hb, hc[2], hc[3], 8.0f, false);
            la = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(la, lo[0], lo[1], 0.00390625f, false);
            lb = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(lb, lo[2], lo[3], 0.00390625f, false);
            const int h8 = (int)((unsigned)(unsigned short)ha[0] | ((unsigned)(unsigned short)hb[0] << 16));
            const int l8 = (int)((unsigned)(unsigned short)la[0] | ((unsigned)(unsigned short)lb[0] << 16));
            unsigned short *dst = buf + ((p / N + 1) * W + (p % N + 1)) * M8_SLOT;
            *reinterpret_cast<f16x4 *>(dst + q * 4) = h;
            *reinterpret_cast<int *>(dst + 32 + q * 2) = h8;
            *reinterpret_cast<int *>(dst + 48 + q * 2) = l8;
        }
    };
    constexpr int NITEM = G::NITEM, IPS = G::IPS;
    // landing area of the trickle's LDS-DMA: [wave][item slot][lane] x 16 bytes, behind the two input buffers
    float *stg = reinterpret_cast<float *>(smem16 + 2 * G::IN_UNITS) + (size_t)wave * IPS * 256;

    // weights of this wave: fp16 planes [stage = ks*9 + tap][wave][ct][lane][8 halfs], then fp8 planes
    // [ks*5 + pair][wave][ct][lane][32 bytes]
    constexpr size_t W16_BYTES = (size_t)KS * 10 * WAVES * 2 * 64 * 16;      // 10 stages per slice: tap 9 is all zeros
    auto w16ptr = [&](int stage, int ct) -> const f16x8 * {
        return reinterpret_cast<const f16x8 *>(wp + ((((size_t)stage * WAVES + wave) * 2 + ct) * 64) * 16) + lane;
    };
    auto w8ptr = [&](int stage, int ct) -> const i32x4 * {
        return reinterpret_cast<const i32x4 *>(wp + W16_BYTES + ((((size_t)stage * WAVES + wave) * 2 + ct) * 64 + lane) * 32);
    };
    auto load_w8 = [&](int stage, int ct) -> i32x8 {
        const i32x4 a = w8ptr(stage, ct)[0], b = w8ptr(stage, ct)[1];
        return i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };
    // E8M0 block scales.  Lanes 0-31 serve the first fp8 plane of a tap (q8(x_h 2^-3) with q8(w_l 2^(d+11))), lanes 32-63 the
    // second (q8(x_l 2^8) with q8(w_h 2^d)): both products carry 2^(d+8), so ONE pair of scales serves every block and lane
    static_assert(M8_XL_SHIFT - M8_XH_SHIFT == 11, "x_l is 2^-11 of x_h: the two cross terms must carry the same power of two");
    const int wshift = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(wp));      // the layer's weight shift d
    wp += M8_HEADER;
    const int scale_x = 127 - 4, scale_w = 127 - (wshift + M8_XL_SHIFT - 4);
    int vscale_w = scale_w, vscale_x = scale_x;      // the asm form takes the scales from VGPRs
    asm volatile("" : "+v"(vscale_w), "+v"(vscale_x));

    if (total == 0) return;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < NITEM; k++) item_store(lds0, k, item_load(in + (size_t)b0 * (size_t)(CELLS * C), k));   // very first slice
    f16x8 a16[2], b16[2];
    i32x8 w8[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) { a16[ct] = *w16ptr(0, ct); b16[ct] = a16[ct]; w8[ct] = i32x8{0, 0, 0, 0, 0, 0, 0, 0}; }

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
        const int base16 = (lane & 15) * M8_SLOT + 8 * (lane >> 4);           // fp16 fragment: 8 channels of k-block lane >> 4
        const int base8 = (lane & 15) * M8_SLOT + 32 + 8 * ((lane >> 4) & 1);  // fp8 fragment: 16 channels of half (g & 1)

        // one half of a sub-step's pixel tiles on the fp16 MFMAs (weights wv) at tap offset toff (slots)
        auto run16 = [&](const f16x8 (&wv)[2], int toff, auto half_c) __attribute__((always_inline)) {
            constexpr int half = decltype(half_c)::value;
            f16x8 px[2];
            constexpr int p_lo = half == 0 ? 0 : PH0, p_hi = half == 0 ? PH0 : PT;
            const unsigned short *ap = lds_in + base16 + toff * M8_SLOT;
            px[0] = *reinterpret_cast<const f16x8 *>(ap + (p_lo * 16) * M8_SLOT);
#pragma unroll
            for (int pt = p_lo, sl = 0; pt < p_hi; pt++, sl ^= 1) {
                if (pt + 1 < p_hi) px[sl ^ 1] = *reinterpret_cast<const f16x8 *>(ap + ((pt + 1) * 16) * M8_SLOT);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < 2; ct++)      // accumulator tied (dst = src C): see M8_MFMA16
                    M8_MFMA16(acc[ct][pt], wv[ct], px[sl]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ... and on the fp8 MFMA (weights w8) with the lane's own tap offset toff8 (groups 0-1: tap 2p, 2-3: tap 2p+1)
        auto run8 = [&](int toff8, auto half_c) __attribute__((always_inline)) {
            constexpr int half = decltype(half_c)::value;
            i32x8 px[2];
            constexpr int p_lo = half == 0 ? 0 : PH0, p_hi = half == 0 ? PH0 : PT;
            const unsigned short *ap = lds_in + base8 + toff8 * M8_SLOT;
            auto ld = [&](int pt) -> i32x8 {
                const i32x4 h = *reinterpret_cast<const i32x4 *>(ap + (pt * 16) * M8_SLOT), l = *reinterpret_cast<const i32x4 *>(ap + (pt * 16) * M8_SLOT + 16);
                return i32x8{h[0], h[1], h[2], h[3], l[0], l[1], l[2], l[3]};
            };
            px[0] = ld(p_lo);
#pragma unroll
            for (int pt = p_lo, sl = 0; pt < p_hi; pt++, sl ^= 1) {
                if (pt + 1 < p_hi) px[sl ^ 1] = ld(pt + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < 2; ct++) M8_MFMA8(acc[ct][pt], w8[ct], px[sl], vscale_w, vscale_x);
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        const std::integral_constant<int, 0> H0{};
        const std::integral_constant<int, 1> H1{};
        // one tap pair per iteration: A_p, B_p, C_p; the loads of all five iterations are identical (tap 9 has zero weights).
        // Every iteration issues the same loads (conditional assignments of loaded registers are merged with v_mov behind an
        // s_waitcnt): the fp8 weights of the pair and B_p's fp16 weights at its top, A_{p+1}'s fp16 weights before C_p.
#pragma unroll 1
        for (int p = 0; p < 5; p++) {
            const int t0 = 2 * p, t1 = p < 4 ? 2 * p + 1 : 8;
            const int off0 = (t0 / 3) * W + t0 % 3, off1 = (t1 / 3) * W + t1 % 3;
            const int off8 = lane < 32 ? off0 : off1;      // tap 8 has no partner: groups 2-3 read tap 8 again, their weights are 0
#pragma unroll
            for (int ct = 0; ct < 2; ct++) w8[ct] = load_w8(ks * 5 + p, ct);
#pragma unroll
            for (int ct = 0; ct < 2; ct++) b16[ct] = *w16ptr(ks * 10 + 2 * p + 1, ct);
            // trickle group p leaves global memory by LDS-DMA (per-lane gather into the wave's landing area, no VGPRs;
            // issued as asm: through the builtin hipcc puts s_waitcnt vmcnt(0) in front of every later ds_read) ...
            if (more) {
#pragma unroll
                for (int q = 0; q < IPS; q++)
                    if (p * IPS + q < NITEM) {
                        int item = tid + (p * IPS + q) * THREADS;
                        item = item < CELLS * 8 ? item : CELLS * 8 - 1;
                        const float *gp = nslice + (size_t)(item >> 3) * C + (item & 7) * 4;
                        const int lds_off = __builtin_amdgcn_readfirstlane((int)((2 * G::IN_UNITS) * 2 + (wave * IPS + q) * 1024));
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lds_off) : "memory", "m0");
                    }
            }
            run16(a16, off0, H0);
            run16(a16, off0, H1);
            if (p < 4) {      // tap 9 does not exist (its weights are zeros: skipping B_4 is worth 3 %; with the tied MFMAs the branch is harmless)
                run16(b16, off1, H0);
                run16(b16, off1, H1);
            }
            {   // A of the next pair: tap 2p+2, the next slice's (or board's) tap 0 after the last pair
                const int nks = ks + 1 < KS ? ks + 1 : 0;      // wraps to the next board's first stage (harmless at the very end)
                const int sa = p < 4 ? ks * 10 + t0 + 2 : nks * 10;
#pragma unroll
                for (int ct = 0; ct < 2; ct++) a16[ct] = *w16ptr(sa, ct);
            }
            // ... and is split and written to its slots: vmcnt completes in order and exactly the two fp16 weight loads above
            // are younger than the DMA (the loads at the top of the pair are older)
            if (more && p * IPS < NITEM) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#pragma unroll
                for (int q = 0; q < IPS; q++)
                    if (p * IPS + q < NITEM) item_store(lds_nx, p * IPS + q, *reinterpret_cast<const f32x4 *>(stg + q * 256 + lane * 4));
            }
            run8(off8, H0);
            run8(off8, H1);
        }
    }   // ks

        M8_MFMA_DRAIN();      // MFMA results -> VALU / store reads
        // (r5: every accumulator tile passes through an empty asm BEHIND the drain -- volatile statements keep their order, so no read of a tile can be
        //  scheduled between the last MFMA and the wait states; without it only the statement order, not a dependence, stood between them)
#pragma unroll
        for (int pt = 0; pt < PT; pt++) { asm volatile("" : "+v"(acc[0][pt])); asm volatile("" : "+v"(acc[1][pt])); }
        // epilogue: + bias, ReLU, 16-byte stores straight from the accumulators (rows with x == N are junk); the stores
        // drain behind the next board's first slice
        float *outb = out + b * (size_t)CELLS * C;
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 bi0 = bias ? *reinterpret_cast<const f32x4 *>(bias + n4) : zero4, bi1 = bias ? *reinterpret_cast<const f32x4 *>(bias + n4 + 16) : zero4;
        const float floor_ = relu ? 0.f : -INFINITY;
        // range guard of this arithmetic: an output beyond range_hi (or not finite) would reach the next layer's fp16 / fp8 planes
        // clamped or as inf -- never silently: the flag is sticky and the host turns it into C6_ERR_RANGE (engine.cpp)
        bool bad = false;
        if (HEAD) {      // see c6_conv3x3_b3s_kernel: per-wave partial sums of the three 1x1 head filters instead of the activations
#pragma unroll
            for (int pt = 0; pt < PT; pt++) {
                f32x4 y0 = acc[0][pt] + bi0, y1 = acc[1][pt] + bi1;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    y0[r] = y0[r] > floor_ ? y0[r] : floor_;
                    y1[r] = y1[r] > floor_ ? y1[r] : floor_;
                    bad = bad || !(__builtin_fabsf(y0[r]) <= range_hi) || !(__builtin_fabsf(y1[r]) <= range_hi);
                }
                acc[0][pt] = y0;
                acc[1][pt] = y1;
            }
            if (bad && range_flag) atomicOr(range_flag, 1u);
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
        // the row offsets are recomputed from a fresh opaque lane id: shared with the residual loads' they would stay alive
        // (48 VGPRs) across the whole K loop and push accumulators into scratch
        // Branch-free: the junk rows go to the slack behind the activation buffer (C6_ACT_SLACK floats; 32 bytes per lane).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int pje = lane_e & 15, n4e = wave * 32 + 4 * (lane_e >> 4);
        float *dump = out + (size_t)nboards * CELLS * C + lane_e * 8;
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
            const int m = pt * 16 + pje, y = m / W, x = m % W;
            const bool ok = x < N && y < N;
            float *op = ok ? outb + (size_t)(y * N + x) * C + n4e : dump;
            f32x4 y0 = acc[0][pt] + bi0, y1 = acc[1][pt] + bi1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                y0[r] = y0[r] > floor_ ? y0[r] : floor_;
                y1[r] = y1[r] > floor_ ? y1[r] : floor_;
                bad = bad || (ok && (!(__builtin_fabsf(y0[r]) <= range_hi) || !(__builtin_fabsf(y1[r]) <= range_hi)));
            }
            *reinterpret_cast<f32x4 *>(op) = y0;
            *reinterpret_cast<f32x4 *>(op + (ok ? 16 : 4)) = y1;
        }
        if (bad && range_flag) atomicOr(range_flag, 1u);
    }
}

// ---- host side: weight packing ---------------------------------------------------------------------------------------
// OCP e4m3 (no infinities, NaN = S.1111.111, max 448), round to nearest even, saturating
static inline unsigned char f32_to_e4m3(float f) {
    if (f != f) return 0x7f;
    const unsigned char s = std::signbit(f) ? 0x80 : 0x00;
    float a = std::fabs(f);
    if (a == 0.f) return s;
    if (a > 448.f) a = 448.f;
    int e;
    const float m = std::frexp(a, &e);      // a = m 2^e, m in [0.5, 1)
    int E = e + 6;                          // biased exponent of the normal form (1 + q/8) 2^(E - 7)
    if (E >= 1) {
        int q = (int)std::nearbyint((2.f * m - 1.f) * 8.f);
        if (q == 8) { q = 0; E++; }
        if (E > 15 || (E == 15 && q == 7)) return s | 0x7e;
        return (unsigned char)(s | (E << 3) | q);
    }
    const int q = (int)std::nearbyint(a * 512.f);      // subnormals: q 2^-9
    return (unsigned char)(s | (q >= 8 ? 0x08 : q));
}

static inline unsigned short f32_to_f16_bits(float f) {
    const _Float16 h = (_Float16)f;      // round to nearest even (host conversion)
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}

size_t c6_conv_packed_m8_bytes(int C) {
    const size_t KS = C / 32, WAVES = C / 32;
    return M8_HEADER + KS * 10 * WAVES * 2 * 64 * 16 + KS * 5 * WAVES * 2 * 64 * 32;
}

// w: folded OIHW [C][C][3][3] ->
//   fp16 planes [stage = ks*10 + tap][wave][ct][lane][8] (tap 9 = zeros):   fp16(W[n = wave*32 + ct*16 + (lane & 15)][k = ks*32 + (lane >> 4)*8 + j][tap])
//   fp8 planes  [ks*5 + pair][wave][ct][lane][32 bytes]:   lane (r, g): tap = 2 pair + (g >> 1) (zeros beyond tap 8),
//               k = ks*32 + 16 (g & 1) + i;  bytes i = q8(w_l * 2^(d+11)), bytes 16 + i = q8(w_h * 2^d); 16-byte header = int32 d
void c6_pack_conv_weights_m8(int C, const float *w, unsigned char *packed) {
    const int KS = C / 32, WAVES = C / 32;
    // header: the weight shift d -- the largest power of two that keeps max |w_h| 2^d within fp8 (416 leaves rounding headroom below 448)
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)C * C * 9; i++) wmax = std::fmax(wmax, std::fabs((float)(_Float16)w[i]));
    int d = 0;
    if (wmax > 0.f && std::isfinite(wmax)) {
        d = (int)std::floor(std::log2(416.0f / wmax));
        d = d < -16 ? -16 : (d > 24 ? 24 : d);
    }
    memset(packed, 0, M8_HEADER);
    memcpy(packed, &d, sizeof(int));
    unsigned short *p16 = reinterpret_cast<unsigned short *>(packed + M8_HEADER);
    unsigned char *p8 = packed + M8_HEADER + (size_t)KS * 10 * WAVES * 2 * 64 * 16;
    for (int ks = 0; ks < KS; ks++)
        for (int tap = 0; tap < 10; tap++)      // tap 9: zeros (the B sub-step beside the lone tap 8)
            for (int wave = 0; wave < WAVES; wave++)
                for (int ct = 0; ct < 2; ct++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int n = wave * 32 + ct * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + j;
                            p16[(((((size_t)(ks * 10 + tap) * WAVES + wave) * 2 + ct) * 64 + lane) * 8) + j] =
                                tap < 9 ? f32_to_f16_bits(w[((size_t)n * C + k) * 9 + tap]) : (unsigned short)0;
                        }
    for (int ks = 0; ks < KS; ks++)
        for (int pair = 0; pair < 5; pair++)
            for (int wave = 0; wave < WAVES; wave++)
                for (int ct = 0; ct < 2; ct++)
                    for (int lane = 0; lane < 64; lane++) {
                        unsigned char *dst = p8 + ((((size_t)(ks * 5 + pair) * WAVES + wave) * 2 + ct) * 64 + lane) * 32;
                        const int g = lane >> 4, tap = 2 * pair + (g >> 1), n = wave * 32 + ct * 16 + (lane & 15);
                        for (int i = 0; i < 16; i++) {
                            unsigned char lo = 0, hi = 0;
                            if (tap < 9) {
                                const int k = ks * 32 + 16 * (g & 1) + i;
                                const float x = w[((size_t)n * C + k) * 9 + tap];
                                const float xh = (float)(_Float16)x, xl = x - xh;
                                lo = f32_to_e4m3(std::ldexp(xl, d + 11));
                                hi = f32_to_e4m3(std::ldexp(xh, d));
                            }
                            dst[i] = lo;
                            dst[16 + i] = hi;
                        }
                    }
}

template <int C, int N, bool HEAD>
static void launch_m8_k(int nboards, int grid, const float *in, const unsigned char *wp, const float *bias, const float *res, float *out,
                        int relu, const float *head_w, unsigned int *range_flag, float range_hi, hipStream_t s) {
    using G = M8Geom<C, N>;
    const size_t smem = G::LDS_BYTES;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_m8_kernel<C, N, HEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((c6_conv3x3_m8_kernel<C, N, HEAD>), dim3(grid), dim3(G::THREADS), smem, s, in, wp, bias, res, out, nboards, relu, head_w, range_flag, range_hi);
}

// head_w != nullptr: [3][C] head filters; `out` receives the head partials instead of the activations (see c6_conv3x3_b3s_kernel)
bool c6_conv_m8_dispatch(int C, int N, int nboards, const float *in, const unsigned char *wp, const float *bias, const float *res,
                         float *out, int num_cus, hipStream_t s, int relu, const float *head_w, unsigned int *range_flag, float range_hi) {
    if (nboards <= 0) return true;
    const int per_cu = C == 256 ? 1 : (C == 128 ? 2 : 4);
    const int grid = nboards < num_cus * per_cu ? nboards : num_cus * per_cu;
#define M8CASE(CC, NN)                                                                                              \
    if (C == CC && N == NN) {                                                                                       \
        if (head_w) launch_m8_k<CC, NN, true>(nboards, grid, in, wp, bias, res, out, relu, head_w, range_flag, range_hi, s);   \
        else launch_m8_k<CC, NN, false>(nboards, grid, in, wp, bias, res, out, relu, nullptr, range_flag, range_hi, s);        \
        return true;                                                                                                \
    }
    M8CASE(256, 19) M8CASE(256, 15) M8CASE(256, 9)
    M8CASE(128, 19) M8CASE(128, 15) M8CASE(128, 9)
    M8CASE(64, 19) M8CASE(64, 15) M8CASE(64, 9)
#undef M8CASE
    return false;
}

// host-only hook for the tests: the packed image of one layer's folded weights (nbytes = c6_conv_pack_m8_bytes(channels))
extern "C" size_t c6_conv_pack_m8_bytes(int channels) { return (channels == 64 || channels == 128 || channels == 256) ? c6_conv_packed_m8_bytes(channels) : 0; }
extern "C" int c6_conv_pack_m8(int channels, const float *w_oihw, unsigned char *packed, size_t nbytes) {
    if (!w_oihw || !packed || c6_conv_pack_m8_bytes(channels) == 0 || nbytes != c6_conv_packed_m8_bytes(channels)) return C6_ERR_BAD_ARG;
    c6_pack_conv_weights_m8(channels, w_oihw, packed);
    return C6_OK;
}
