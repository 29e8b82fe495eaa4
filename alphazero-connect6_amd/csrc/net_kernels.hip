// net_kernels.hip -- K4: resident policy/value ResNet, forward only, FP32.
//
// Architecture = WeightedPolicy::forward, AlphaZero/cpp_weighted/model.hpp:52-95 (ctor :10-50):
//   stem conv3x3 2->C + BN + ReLU (:59); (blocks-1) residual blocks conv-BN-ReLU-conv-BN-(+x)-ReLU
//   (:61-78); policy head conv1x1 C->2, BN, ReLU, Linear(2N^2->N^2), softmax (:80-84); value head
//   conv1x1 C->1, BatchNorm over the N^2 features, ReLU, Linear(N^2->256), ReLU, Linear(256->1), tanh
//   (:86-92).  BatchNorm is eval-mode and folded into the preceding convolution on the host.
//
// The 3x3 convolutions are >99.9 % of the FLOPs.  c6_conv3x3_kernel is an implicit GEMM on the
// FP32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 products, f32 accumulate):
//   M = the N*N pixels of ONE board (padded to a multiple of 32), N = C output channels,
//   K = 9 taps x C input channels.
// One 512-thread workgroup (8 waves = 4 along M x 2 along N) owns one board and all C output
// channels, so activations are read from HBM exactly once per layer.  Activations are NHWC
// ([board][pixel][C]); a 32-channel slice of the board is staged in LDS on a zero-haloed
// (N+2)x(N+2) grid, so the nine taps are nine constant LDS offsets from the same staged slice
// (no im2col in HBM).  Weights are pre-packed on the host so that a (tap, 32-channel slice)
// stage is one contiguous 32*C-float block and a lane's 4 consecutive k values are one
// ds_read_b128; stages are double-buffered in LDS and prefetched into registers behind the MFMAs.
#include <cstdlib>

#include "c6_device.h"
#include "c6_engine.h"

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CONV_THREADS 512
#define SLOT_F 36   // floats per LDS pixel slot: 32 channels + 4 pad (bank spread for ds_read_b128)

struct C6NetTimers {
    bool enabled = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> conv;   // one pair per conv launch
    std::vector<hipEvent_t> pool;
    double conv_flops = 0.0;
    double acc_ms = 0.0;
    unsigned long long acc_n = 0;
};

// ----------------------------------------------------------------------------------------
// conv3x3 C->C, pad 1, + bias (+ residual) + ReLU
// ----------------------------------------------------------------------------------------
// ABL: ablation switch used only by c6_bench_conv (0 = the product kernel):
//   1 no weight staging, 2 no input staging, 3 neither and no barriers, 4 operands from registers (no LDS reads)
template <int C, int MTW, int ABL = 0>
__global__ __launch_bounds__(CONV_THREADS) void c6_conv3x3_kernel(const float *__restrict__ in, const float *__restrict__ wp,
                                                                 const float *__restrict__ bias, const float *__restrict__ res,
                                                                 float *__restrict__ out, int N, int relu) {
    constexpr int NTW = C / 64;          // 32-wide n-tiles per wave (2 waves along N)
    constexpr int KS = C / 32;           // 32-channel k-slices
    constexpr int WSTAGE = 32 * C;       // floats per weight stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cells = N * N, P = N + 2;
    const int in_floats = P * P * SLOT_F;
    float *lds_in = smem;
    float *lds_w = smem + ((in_floats + 3) & ~3);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const size_t b = blockIdx.x;
    const float *inb = in + b * (size_t)cells * C;

    int aoff[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++) {
        int p = (wm * MTW + mt) * 32 + i;
        if (p >= cells) p = 0;          // padding rows: read pixel 0 (in bounds), never stored
        aoff[mt] = ((p / N + 1) * P + (p % N + 1)) * SLOT_F + 4 * h;
    }

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[mt][nt][v] = 0.f;

    for (int k = tid; k < in_floats; k += CONV_THREADS) lds_in[k] = 0.f;   // halo stays zero for the whole kernel

    // weight stages (one contiguous 32*C-float block per (slice, tap)) stream into LDS by LDS-DMA
    // (global_load_lds_dwordx4: no registers, no ds_write), one stage ahead of the MFMAs, double-buffered
    constexpr int WPIECES = WSTAGE / 256;           // 1-KiB pieces per stage
    auto dma_w = [&](int stage, int buf) {
        const float *src = wp + (size_t)stage * WSTAGE;
        float *dst = lds_w + buf * WSTAGE;
        for (int pc = wave; pc < WPIECES; pc += CONV_THREADS / 64)
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)(src + pc * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(dst + pc * 256), 16, 0, 0);
    };

    if (ABL != 1 && ABL != 3) dma_w(0, 0);   // stage index = ks * 9 + tap
    for (int ks = 0; ks < KS; ks++) {
        if (ABL != 3) __syncthreads();   // previous slice fully consumed (and the zero fill done)
        // stage the 32-channel input slice: pixel p -> slot (y+1, x+1)
        if (ABL != 2 && ABL != 3)
        for (int item = tid; item < cells * 8; item += CONV_THREADS) {
            const int p = item >> 3, q = item & 7;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(inb + (size_t)p * C + ks * 32 + q * 4);
            *reinterpret_cast<f32x4 *>(lds_in + ((p / N + 1) * P + (p % N + 1)) * SLOT_F + q * 4) = v;
        }
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
            const int stage = ks * 9 + tap, buf = stage & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of `stage` have landed
            if (ABL != 3) __syncthreads();                     // ... and everybody's; the other buffer is free again
            if (ABL != 1 && ABL != 3) if (stage + 1 < KS * 9) dma_w(stage + 1, buf ^ 1);   // behind the MFMAs below
            const int toff = ((tap / 3 - 1) * P + (tap % 3 - 1)) * SLOT_F;
            const float *wb = lds_w + buf * WSTAGE;
#pragma unroll
            for (int kq = 0; kq < 4; kq++) {
                f32x4 a[MTW], bf[NTW];
#pragma unroll
                for (int mt = 0; mt < MTW; mt++) {
                    if (ABL == 4) { a[mt] = f32x4{(float)tap, (float)kq, (float)mt, 1.f}; asm volatile("" : "+v"(a[mt])); }
                    else a[mt] = *reinterpret_cast<const f32x4 *>(lds_in + aoff[mt] + toff + kq * 8);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) {
                    if (ABL == 4) { bf[nt] = f32x4{(float)ks, 0.5f, (float)nt, 2.f}; asm volatile("" : "+v"(bf[nt])); }
                    else bf[nt] = *reinterpret_cast<const f32x4 *>(wb + ((kq * 2 + h) * C + (wn * NTW + nt) * 32 + i) * 4);
                }
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int mt = 0; mt < MTW; mt++)
#pragma unroll
                        for (int nt = 0; nt < NTW; nt++)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][r], bf[nt][r], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // epilogue.  In the C/D layout a lane owns ONE channel (col = lane & 31) and 16 rows
    // (row = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)), which would make every residual load / output store a 4-byte
    // access.  Each 32x32 tile is therefore transposed through a wave-private LDS patch so that a lane handles four
    // consecutive channels of a pixel: residual and output move as 16-byte vectors, 1 KiB per wave-instruction.
    __syncthreads();                                   // every wave is done with the staging buffers
    constexpr int PSTR = 36;                           // patch row stride in floats (16-byte aligned rows)
    float *patch = smem + wave * (32 * PSTR);          // 8 x 4.5 KiB; the launcher sizes LDS for it
    float *outb = out + b * (size_t)cells * C;
    const float *resb = res ? res + b * (size_t)cells * C : nullptr;
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const int n4 = (wn * NTW + nt) * 32 + c4;
        // bias == nullptr / relu == 0: the training path (plain convolution)
        const f32x4 bi = bias ? *reinterpret_cast<const f32x4 *>(bias + n4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float floor_ = relu ? 0.f : -INFINITY;
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
                    for (int q = 0; q < 4; q++) y[q] = y[q] > floor_ ? y[q] : floor_;
                    *reinterpret_cast<f32x4 *>(outb + (size_t)p * C + n4) = y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch reads done before the next tile overwrites it
        }
    }
}

size_t c6_conv_packed_floats(int C) { return (size_t)9 * C * C; }

// w: folded OIHW [C][C][3][3]  ->  [tap][ks][kq][h][n][r], k = ks*32 + kq*8 + h*4 + r
void c6_pack_conv_weights(int C, const float *w, float *packed) {
    const int KS = C / 32;
    for (int t = 0; t < 9; t++)
        for (int ks = 0; ks < KS; ks++)
            for (int kq = 0; kq < 4; kq++)
                for (int h = 0; h < 2; h++)
                    for (int n = 0; n < C; n++)
                        for (int r = 0; r < 4; r++) {
                            const int k = ks * 32 + kq * 8 + h * 4 + r;
                            // stage index inside the kernel is ks*9 + tap
                            packed[((((size_t)(ks * 9 + t) * 4 + kq) * 2 + h) * C + n) * 4 + r] = w[((size_t)n * C + k) * 9 + t];
                        }
}

// ----------------------------------------------------------------------------------------
// stem: dihedral gather (augment_way8, augment.rs:60-73) fused with conv3x3 2->C + folded BN + ReLU
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void c6_stem_kernel(const int8_t *__restrict__ boards, int board_stride,
                                                     const int8_t *__restrict__ player, const int16_t *__restrict__ aug_map,
                                                     int images, const float *__restrict__ w /*[9][2][C]*/,
                                                     const float *__restrict__ bias, float *__restrict__ out, int N, int C,
                                                     int s32) {
    __shared__ float pl[21 * 21], bd[21 * 21];
    const int cells = N * N, P = N + 2;
    const size_t b = blockIdx.x;
    const size_t leaf = b / images;
    const int k = images == 8 ? (int)(b % 8) : 6;     // image 6 is the identity (four left rotations)
    const int8_t *board = boards + leaf * (size_t)board_stride;
    const float pv = (float)player[leaf];
    for (int s = threadIdx.x; s < P * P; s += 256) { pl[s] = 0.f; bd[s] = 0.f; }
    __syncthreads();
    for (int c = threadIdx.x; c < cells; c += 256) {
        const int s = (c / N + 1) * P + (c % N + 1);
        pl[s] = pv;                                     // model.hpp:56: player plane (zero padded like the board)
        bd[s] = (float)board[aug_map[k * cells + c]];
    }
    __syncthreads();
    // a thread owns 4 consecutive output channels (16-byte stores, 1 KiB per wave-instruction) and strides over pixels
    const int CQ = C / 4;
    const int cq = threadIdx.x % CQ, psub = threadIdx.x / CQ, PP = 256 / CQ;
    const int co = cq * 4;
    f32x4 w0[9], w1[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        w0[t] = *reinterpret_cast<const f32x4 *>(w + (t * 2 + 0) * C + co);
        w1[t] = *reinterpret_cast<const f32x4 *>(w + (t * 2 + 1) * C + co);
    }
    const f32x4 bi = *reinterpret_cast<const f32x4 *>(bias + co);
    float *ob = out + b * (size_t)cells * C;
    for (int p = psub; p < cells; p += PP) {
        const int s = (p / N + 1) * P + (p % N + 1);
        f32x4 acc = bi;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int o = s + (t / 3 - 1) * P + (t % 3 - 1);
            const float pv2 = pl[o], bv = bd[o];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[r] = __builtin_fmaf(w0[t][r], pv2, acc[r]);
                acc[r] = __builtin_fmaf(w1[t][r], bv, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = acc[r] > 0.f ? acc[r] : 0.f;
        // NHWC, or slice-major S32 = [C/32][pixel][32] (conv_v2.hip)
        const size_t o = s32 ? ((size_t)(co >> 5) * cells + p) * 32 + (co & 31) : (size_t)p * C + co;
        *reinterpret_cast<f32x4 *>(ob + o) = acc;
    }
}

// ----------------------------------------------------------------------------------------
// heads
// ----------------------------------------------------------------------------------------
// conv1x1 C->2 (policy, folded BN2d + ReLU) and C->1 (value, BatchNorm1d over cells + ReLU)
__global__ __launch_bounds__(256) void c6_head_conv_kernel(const float *__restrict__ act, const float *__restrict__ pc_w,
                                                          const float *__restrict__ pc_b, const float *__restrict__ vc_w,
                                                          float vc_b, const float *__restrict__ vn_scale,
                                                          const float *__restrict__ vn_shift, float *__restrict__ ph,
                                                          float *__restrict__ vh, int cells, int C, int s32) {
    const size_t b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *ab = act + b * (size_t)cells * C;
    for (int p = wave; p < cells; p += 4) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const size_t o = s32 ? ((size_t)(c >> 5) * cells + p) * 32 + (c & 31) : (size_t)p * C + c;
            const f32x4 x = *reinterpret_cast<const f32x4 *>(ab + o);
            const f32x4 a = *reinterpret_cast<const f32x4 *>(pc_w + c);
            const f32x4 d = *reinterpret_cast<const f32x4 *>(pc_w + C + c);
            const f32x4 e = *reinterpret_cast<const f32x4 *>(vc_w + c);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                s0 = __builtin_fmaf(x[r], a[r], s0);
                s1 = __builtin_fmaf(x[r], d[r], s1);
                s2 = __builtin_fmaf(x[r], e[r], s2);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s0 += __shfl_xor(s0, off, 64);
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        if (lane == 0) {
            const float y0 = s0 + pc_b[0], y1 = s1 + pc_b[1];
            ph[b * (size_t)(2 * cells) + p] = y0 > 0.f ? y0 : 0.f;             // flatten is channel-major (NCHW view)
            ph[b * (size_t)(2 * cells) + cells + p] = y1 > 0.f ? y1 : 0.f;
            const float v = (s2 + vc_b) * vn_scale[p] + vn_shift[p];
            vh[b * (size_t)cells + p] = v > 0.f ? v : 0.f;
        }
    }
}

// the same outputs from the per-wave partial sums the tower's last convolution leaves behind (c6_conv3x3_b3s_kernel<.., HEAD>):
// part[board][filter][wave][GEMM row], row m = y * (N + 1) + x; the partials are added in wave order
__global__ __launch_bounds__(256) void c6_head_reduce_kernel(const float *__restrict__ part, const float *__restrict__ pc_b, float vc_b,
                                                            const float *__restrict__ vn_scale, const float *__restrict__ vn_shift,
                                                            float *__restrict__ ph, float *__restrict__ vh, int N, int waves, int mp, int nb) {
    const int cells = N * N;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)nb * cells) return;
    const size_t b = i / cells;
    const int p = (int)(i - b * cells);
    const float *q = part + b * (size_t)(3 * waves * mp) + (p / N) * (N + 1) + p % N;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int w = 0; w < waves; w++) {
        s0 += q[(size_t)w * mp];
        s1 += q[(size_t)(waves + w) * mp];
        s2 += q[(size_t)(2 * waves + w) * mp];
    }
    const float y0 = s0 + pc_b[0], y1 = s1 + pc_b[1];
    ph[b * (size_t)(2 * cells) + p] = y0 > 0.f ? y0 : 0.f;             // flatten is channel-major (NCHW view)
    ph[b * (size_t)(2 * cells) + cells + p] = y1 > 0.f ? y1 : 0.f;
    const float v = (s2 + vc_b) * vn_scale[p] + vn_shift[p];
    vh[b * (size_t)cells + p] = v > 0.f ? v : 0.f;
}

#define HB 8   // boards per workgroup in the dense heads

// policy_linear (2N^2 -> N^2) + softmax, model.hpp:83-84
__global__ __launch_bounds__(256) void c6_head_policy_kernel(const float *__restrict__ ph, const float *__restrict__ wT,
                                                            const float *__restrict__ bias, float *__restrict__ pol, int cells,
                                                            int nboards) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = 2 * cells;
    float *x = sm;                 // [HB][K]
    float *lg = sm + HB * K;       // [HB][cells]
    const size_t b0 = (size_t)blockIdx.x * HB;
    for (int idx = threadIdx.x; idx < HB * K; idx += 256) {
        const size_t bb = b0 + idx / K;
        x[idx] = bb < (size_t)nboards ? ph[bb * K + idx % K] : 0.f;
    }
    __syncthreads();
    {
        // a thread owns outputs o1 = tid and (19x19: 361 > 256) o2 = tid + 256 and walks k ONCE for both: the LDS reads of x
        // are shared, and the second pass that ran with 105 of 256 threads active is gone (same k order per output:
        // bit-identical results)
        const int o1 = threadIdx.x, o2 = threadIdx.x + 256;
        const bool h1 = o1 < cells, h2 = o2 < cells;
        const int c1 = h1 ? o1 : 0, c2 = h2 ? o2 : c1;      // clamped: loads stay unconditional
        float acc1[HB], acc2[HB];
#pragma unroll
        for (int q = 0; q < HB; q++) { acc1[q] = bias[c1]; acc2[q] = bias[c2]; }
        if (cells > 256) {
            for (int k = 0; k < K; k++) {
                const float w1 = wT[(size_t)k * cells + c1], w2 = wT[(size_t)k * cells + c2];
#pragma unroll
                for (int q = 0; q < HB; q++) {
                    const float xv = x[q * K + k];
                    acc1[q] = __builtin_fmaf(w1, xv, acc1[q]);
                    acc2[q] = __builtin_fmaf(w2, xv, acc2[q]);
                }
            }
        } else {
            for (int k = 0; k < K; k++) {
                const float w1 = wT[(size_t)k * cells + c1];
#pragma unroll
                for (int q = 0; q < HB; q++) acc1[q] = __builtin_fmaf(w1, x[q * K + k], acc1[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < HB; q++) {
            if (h1) lg[q * cells + o1] = acc1[q];
            if (h2) lg[q * cells + o2] = acc2[q];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = wave; q < HB; q += 4) {
        const size_t bb = b0 + q;
        if (bb >= (size_t)nboards) continue;
        float mx = -INFINITY;
        for (int o = lane; o < cells; o += 64) mx = fmaxf(mx, lg[q * cells + o]);
        mx = c6_wave_max_f32(mx);
        float s = 0.f;
        for (int o = lane; o < cells; o += 64) {
            const float e = __expf(lg[q * cells + o] - mx);
            lg[q * cells + o] = e;
            s += e;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        for (int o = lane; o < cells; o += 64) pol[bb * cells + o] = lg[q * cells + o] / s;
    }
}

// value_linear1 (N^2 -> 256) + ReLU + value_linear2 (256 -> 1) + tanh, model.hpp:89-92
__global__ __launch_bounds__(256) void c6_head_value_kernel(const float *__restrict__ vh, const float *__restrict__ w1T,
                                                           const float *__restrict__ b1, const float *__restrict__ w2, float b2,
                                                           float *__restrict__ val, int cells, int nboards) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *x = sm;                     // [HB][cells]
    float *red = sm + HB * cells;      // [HB][4]
    const size_t b0 = (size_t)blockIdx.x * HB;
    for (int idx = threadIdx.x; idx < HB * cells; idx += 256) {
        const size_t bb = b0 + idx / cells;
        x[idx] = bb < (size_t)nboards ? vh[bb * cells + idx % cells] : 0.f;
    }
    __syncthreads();
    const int hdn = threadIdx.x;
    float acc[HB];
#pragma unroll
    for (int q = 0; q < HB; q++) acc[q] = b1[hdn];
    for (int k = 0; k < cells; k++) {
        const float w = w1T[(size_t)k * 256 + hdn];
#pragma unroll
        for (int q = 0; q < HB; q++) acc[q] = __builtin_fmaf(w, x[q * cells + k], acc[q]);
    }
    const float w2h = w2[hdn];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < HB; q++) {
        float v = (acc[q] > 0.f ? acc[q] : 0.f) * w2h;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[q * 4 + wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < HB) {
        const size_t bb = b0 + threadIdx.x;
        if (bb < (size_t)nboards) {
            const float *r = red + threadIdx.x * 4;
            val[bb] = tanhf(((r[0] + r[1]) + (r[2] + r[3])) + b2);
        }
    }
}

// ----------------------------------------------------------------------------------------
// host-side driver
// ----------------------------------------------------------------------------------------
template <int C>
static void launch_conv(int MTW, int nboards, size_t smem, const float *in, const float *wp, const float *bias, const float *res,
                        float *out, int N, hipStream_t s, int relu) {
    switch (MTW) {
    case 1:
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_kernel<C, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((c6_conv3x3_kernel<C, 1>), dim3(nboards), dim3(CONV_THREADS), smem, s, in, wp, bias, res, out, N, relu);
        break;
    case 2:
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_kernel<C, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((c6_conv3x3_kernel<C, 2>), dim3(nboards), dim3(CONV_THREADS), smem, s, in, wp, bias, res, out, N, relu);
        break;
    default:
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_kernel<C, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((c6_conv3x3_kernel<C, 3>), dim3(nboards), dim3(CONV_THREADS), smem, s, in, wp, bias, res, out, N, relu);
        break;
    }
}

static void conv_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res,
                          float *out, hipStream_t s, int relu = 1) {
    const int cells = N * N, MT = (cells + 31) / 32, MTW = (MT + 3) / 4, P = N + 2;
    size_t smem = ((size_t)((P * P * SLOT_F + 3) & ~3) + 2 * 32 * (size_t)C) * sizeof(float);
    if (smem < (size_t)8 * 32 * 36 * sizeof(float)) smem = (size_t)8 * 32 * 36 * sizeof(float);   // epilogue patches
    switch (C) {
    case 64: launch_conv<64>(MTW, nboards, smem, in, wp, bias, res, out, N, s, relu); break;
    case 128: launch_conv<128>(MTW, nboards, smem, in, wp, bias, res, out, N, s, relu); break;
    case 256: launch_conv<256>(MTW, nboards, smem, in, wp, bias, res, out, N, s, relu); break;
    default: break;
    }
}

// exact-FP32 convolution for the trainer (csrc/trainer.cpp): plain conv (bias nullable, relu selectable) + residual
bool c6_conv_fp32_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out,
                           hipStream_t s, int relu) {
    if (C != 64 && C != 128 && C != 256) return false;
    conv_dispatch(C, N, nboards, in, wp, bias, res, out, s, relu);
    return true;
}

static hipEvent_t timer_event(C6NetTimers *t) {
    hipEvent_t e;
    if (!t->pool.empty()) { e = t->pool.back(); t->pool.pop_back(); }
    else hipEventCreate(&e);
    return e;
}

void c6_net_forward_dev(const C6NetDev &net, int nleaf, int images, const int8_t *boards, int board_stride,
                        const int8_t *player, const int16_t *aug_map, float *act0, float *act1, float *act2, float *ph,
                        float *vh, float *pol, float *val, hipStream_t s, C6NetTimers *timers) {
    const int nb = nleaf * images, cells = net.cells, C = net.C;
    if (nb == 0) return;
    const int v2 = net.variant == 2 ? 1 : 0;
    hipLaunchKernelGGL(c6_stem_kernel, dim3(nb), dim3(256), 0, s, boards, board_stride, player, aug_map, images, net.stem_w,
                       net.stem_b, act0, net.N, C, v2);
    float *x = act0, *y = act1, *z = act2;
    const size_t wsz = c6_conv_packed_floats(C);
    const bool timed = timers && timers->enabled;
    static const bool fuse_heads = [] { const char *e = getenv("C6_HEAD_FUSE"); return !(e && e[0] == '0'); }();   // A/B switch
    bool head_fused = false;
    for (int i = 0; i < net.blocks - 1; i++) {     // res_block, model.hpp:61-78
        for (int half = 0; half < 2; half++) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (timed) { e0 = timer_event(timers); e1 = timer_event(timers); hipEventRecord(e0, s); }
            const int l = 2 * i + half;
            const float *cin = half == 0 ? x : y, *cres = half == 0 ? nullptr : x;
            float *cout = half == 0 ? y : z;
            if (net.variant == 5) c6_conv_f32n_dispatch(C, net.N, nb, cin, net.conv_w5 + (size_t)l * wsz, net.conv_b + (size_t)l * C, cres, cout, net.num_cus, s);
            else if (net.variant == 7 || net.variant == 8) {
                // the last convolution keeps its activations to itself and hands the heads their 1x1 partial sums
                head_fused = fuse_heads && l == 2 * (net.blocks - 1) - 1;
                c6_conv_b3s_dispatch(C, net.N, nb, cin, net.conv_w7 + (size_t)l * c6_conv_packed_b3n_units(C), net.conv_b + (size_t)l * C, cres, cout, net.num_cus, s, 1, net.variant == 7, head_fused ? net.head_w : nullptr);
            }
            else if (net.variant == 9) {
                head_fused = fuse_heads && l == 2 * (net.blocks - 1) - 1;
                c6_conv_m8_dispatch(C, net.N, nb, cin, net.conv_w9 + (size_t)l * c6_conv_packed_m8_bytes(C), net.conv_b + (size_t)l * C, cres, cout, net.num_cus, s, 1, head_fused ? net.head_w : nullptr);
            }
            else if (net.variant == 6) c6_conv_b3n_dispatch(C, net.N, nb, cin, net.conv_w6 + (size_t)l * c6_conv_packed_b3n_units(C), net.conv_b + (size_t)l * C, cres, cout, net.num_cus, s);
            else if (net.variant == 3) c6_conv_b3_dispatch(C, net.N, nb, cin, net.conv_w3 + (size_t)l * c6_conv_packed_b3_units(C), net.conv_b + (size_t)l * C, cres, cout, s);
            else if (v2) c6_conv_v2_dispatch(C, net.N, nb, cin, net.conv_w2 + (size_t)l * wsz, net.conv_b + (size_t)l * C, cres, cout, net.num_cus, s);
            else conv_dispatch(C, net.N, nb, cin, net.conv_w + (size_t)l * wsz, net.conv_b + (size_t)l * C, cres, cout, s);
            if (timed) {
                hipEventRecord(e1, s);
                timers->conv.emplace_back(e0, e1);
                timers->conv_flops += 2.0 * 9.0 * C * C * cells * (double)nb;
            }
        }
        float *t = x; x = z; z = t;
    }
    if (head_fused)
        hipLaunchKernelGGL(c6_head_reduce_kernel, dim3((unsigned)(((size_t)nb * cells + 255) / 256)), dim3(256), 0, s, x, net.pc_b, net.vc_b,
                           net.vn_scale, net.vn_shift, ph, vh, net.N, C / 32, c6_conv_b3s_head_rows(net.N), nb);
    else
        hipLaunchKernelGGL(c6_head_conv_kernel, dim3(nb), dim3(256), 0, s, x, net.pc_w, net.pc_b, net.vc_w, net.vc_b, net.vn_scale,
                           net.vn_shift, ph, vh, cells, C, v2);
    const int hgrid = (nb + HB - 1) / HB;
    const size_t sm_p = (size_t)(HB * 2 * cells + HB * cells) * sizeof(float);
    hipLaunchKernelGGL(c6_head_policy_kernel, dim3(hgrid), dim3(256), sm_p, s, ph, net.pl_wT, net.pl_b, pol, cells, nb);
    const size_t sm_v = (size_t)(HB * cells + HB * 4) * sizeof(float);
    hipLaunchKernelGGL(c6_head_value_kernel, dim3(hgrid), dim3(256), sm_v, s, vh, net.v1_wT, net.v1_b, net.v2_w, net.v2_b, val,
                       cells, nb);
}

// timers API used by engine.cpp
C6NetTimers *c6_timers_new() { return new C6NetTimers(); }
void c6_timers_free(C6NetTimers *t) {
    if (!t) return;
    for (auto &p : t->conv) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto e : t->pool) hipEventDestroy(e);
    delete t;
}
void c6_timers_enable(C6NetTimers *t, bool on) { t->enabled = on; }
// drains finished event pairs; returns (ms, launches, flops)
void c6_timers_collect(C6NetTimers *t, double *ms, unsigned long long *launches, double *flops, bool reset) {
    for (auto &p : t->conv) {
        float m = 0.f;
        hipEventSynchronize(p.second);
        hipEventElapsedTime(&m, p.first, p.second);
        t->acc_ms += m;
        t->acc_n += 1;
        t->pool.push_back(p.first);
        t->pool.push_back(p.second);
    }
    t->conv.clear();
    *ms = t->acc_ms;
    *launches = t->acc_n;
    *flops = t->conv_flops;
    if (reset) { t->acc_ms = 0; t->acc_n = 0; t->conv_flops = 0; }
}

// ---- micro-benchmark hook (c6_bench_conv in engine.cpp): one conv launch of a variant / ablation ----
void c6_conv_bench_launch(int variant, int ablate, int C, int N, int nboards, const float *in, const float *w1, const float *w2,
                          const float *bias, const float *res, float *out, int num_cus, hipStream_t s) {
    if (variant == 2) { c6_conv_v2_dispatch(C, N, nboards, in, w2, bias, res, out, num_cus, s); return; }
    if (variant == 5) { c6_conv_f32n_dispatch(C, N, nboards, in, w2, bias, res, out, num_cus, s); return; }
    if (variant == 6) { c6_conv_b3n_dispatch(C, N, nboards, in, reinterpret_cast<const unsigned short *>(w2), bias, res, out, num_cus, s); return; }
    if (variant == 7 || variant == 8) { c6_conv_b3s_dispatch(C, N, nboards, in, reinterpret_cast<const unsigned short *>(w2), bias, res, out, num_cus, s, 1, variant == 7); return; }
    if (variant == 9) { c6_conv_m8_dispatch(C, N, nboards, in, reinterpret_cast<const unsigned char *>(w2), bias, res, out, num_cus, s, 1, nullptr); return; }
    if (variant == 3 && ablate != 0 && C == 256 && N == 19) { c6_conv_b3_ablate(ablate, nboards, in, reinterpret_cast<const unsigned short *>(w2), bias, res, out, s); return; }
    if (variant == 3) { c6_conv_b3_dispatch(C, N, nboards, in, reinterpret_cast<const unsigned short *>(w2), bias, res, out, s); return; }
    if (C != 256 || N != 19 || ablate == 0) { conv_dispatch(C, N, nboards, in, w1, bias, res, out, s); return; }
    const int P = N + 2;
    const size_t smem = ((size_t)((P * P * SLOT_F + 3) & ~3) + 2 * 32 * (size_t)C) * sizeof(float);
#define ABLCASE(A) case A: hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_kernel<256, 3, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((c6_conv3x3_kernel<256, 3, A>), dim3(nboards), dim3(CONV_THREADS), smem, s, in, w1, bias, res, out, N, 1); break;
    switch (ablate) { ABLCASE(1) ABLCASE(2) ABLCASE(3) ABLCASE(4) default: break; }
#undef ABLCASE
}
