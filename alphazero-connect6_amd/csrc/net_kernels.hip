// net_kernels.hip -- K4: resident policy/value ResNet, forward only, FP32.
//
// Architecture = WeightedPolicy::forward, AlphaZero/cpp_weighted/model.hpp:52-95 (ctor :10-50):
//   stem conv3x3 2->C + BN + ReLU (:59); (blocks-1) residual blocks conv-BN-ReLU-conv-BN-(+x)-ReLU
//   (:61-78); policy head conv1x1 C->2, BN, ReLU, Linear(2N^2->N^2), softmax (:80-84); value head
//   conv1x1 C->1, BatchNorm over the N^2 features, ReLU, Linear(N^2->256), ReLU, Linear(256->1), tanh
//   (:86-92).  BatchNorm is eval-mode and folded into the preceding convolution on the host.
//
// The 3x3 convolutions are >99.9 % of the FLOPs; their kernels live in conv_wino4.hip / conv_wino.hip (FP32 Winograd F(4x4,3x3) at 256 channels / F(2x2,3x3) below: the default),
// conv_f32n.hip (FP32 direct), conv_bf16x3.hip (split bf16) and conv_m8.hip (fp16 + MX-fp8).  This file holds the stem, the
// heads and the forward driver.  Activations are NHWC ([board][pixel][C]).
#include <cstdlib>

#include "c6_device.h"
#include "c6_engine.h"

#include <vector>

#ifndef C6_AB_HEAD_FUSE
#define C6_AB_HEAD_FUSE 1
#endif
#ifndef C6_AB_WINO_MID
#define C6_AB_WINO_MID 2
#endif
#ifndef C6_AB_WINO_STEM
#define C6_AB_WINO_STEM 1
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct C6NetTimers {
    bool enabled = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> conv;   // one pair per conv layer (all its kernels)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pre;    // Winograd: the input-transform kernel of that layer
    std::vector<std::pair<hipEvent_t, hipEvent_t>> post;   // Winograd F(4x4): its output-transform kernel
    double pre_ms = 0.0, post_ms = 0.0, exec_flops = 0.0;
    std::vector<hipEvent_t> pool;
    double conv_flops = 0.0;
    double acc_ms = 0.0;
    unsigned long long acc_n = 0;
};

// ----------------------------------------------------------------------------------------
// stem: dihedral gather (augment_way8, augment.rs:60-73) fused with conv3x3 2->C + folded BN + ReLU
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void c6_stem_kernel(const int8_t *__restrict__ boards, int board_stride,
                                                     const int8_t *__restrict__ player, int player_stride, const int16_t *__restrict__ aug_map,
                                                     int images, const float *__restrict__ w /*[9][2][C]*/,
                                                     const float *__restrict__ bias, float *__restrict__ out, int N, int C) {
    __shared__ float pl[21 * 21], bd[21 * 21];
    const int cells = N * N, P = N + 2;
    const size_t b = blockIdx.x;
    const size_t leaf = b / images;
    const int k = images == 8 ? (int)(b % 8) : 6;     // image 6 is the identity (four left rotations)
    const int8_t *board = boards + leaf * (size_t)board_stride;
    const float pv = (float)player[leaf * (size_t)player_stride];
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
        *reinterpret_cast<f32x4 *>(ob + (size_t)p * C + co) = acc;
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
                                                          float *__restrict__ vh, int cells, int C) {
    const size_t b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *ab = act + b * (size_t)cells * C;
    for (int p = wave; p < cells; p += 4) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(ab + (size_t)p * C + c);
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
        // r5: the weights of sixteen k at a time, requested one block ahead -- the loop was a chain of 722 dependent L2 round trips, 180 us for a
        // workgroup on its own (one game's eight boards); the multiply-adds keep their order (k ascending per output: bit-identical results)
        // (blocks of sixteen, then at most one half block of eight whose loads go out together, then < 8 single k: 722 = 45 x 16 + 2)
        constexpr int KB = 16, KH = 8;
        if (cells > 256) {
            float wa[KB], wb[KB];
#pragma unroll
            for (int u = 0; u < KB; u++) { wa[u] = wT[(size_t)u * cells + c1]; wb[u] = wT[(size_t)u * cells + c2]; }      // (K = 2 cells >= 162: a whole first block)
            int k = 0;
            for (; k + KB <= K; k += KB) {
                float na[KB], nb_[KB];
                const int kn = k + KB + KB <= K ? k + KB : k;      // the next whole block (the last one: itself again, not used)
#pragma unroll
                for (int u = 0; u < KB; u++) { na[u] = wT[(size_t)(kn + u) * cells + c1]; nb_[u] = wT[(size_t)(kn + u) * cells + c2]; }
#pragma unroll
                for (int u = 0; u < KB; u++) {
#pragma unroll
                    for (int q = 0; q < HB; q++) {
                        const float xv = x[q * K + k + u];
                        acc1[q] = __builtin_fmaf(wa[u], xv, acc1[q]);
                        acc2[q] = __builtin_fmaf(wb[u], xv, acc2[q]);
                    }
                }
#pragma unroll
                for (int u = 0; u < KB; u++) { wa[u] = na[u]; wb[u] = nb_[u]; }
            }
            if (k + KH <= K) {
                float ra[KH], rb[KH];
#pragma unroll
                for (int u = 0; u < KH; u++) { ra[u] = wT[(size_t)(k + u) * cells + c1]; rb[u] = wT[(size_t)(k + u) * cells + c2]; }
#pragma unroll
                for (int u = 0; u < KH; u++) {
#pragma unroll
                    for (int q = 0; q < HB; q++) {
                        const float xv = x[q * K + k + u];
                        acc1[q] = __builtin_fmaf(ra[u], xv, acc1[q]);
                        acc2[q] = __builtin_fmaf(rb[u], xv, acc2[q]);
                    }
                }
                k += KH;
            }
            for (; k < K; k++) {
                const float w1 = wT[(size_t)k * cells + c1], w2 = wT[(size_t)k * cells + c2];
#pragma unroll
                for (int q = 0; q < HB; q++) {
                    const float xv = x[q * K + k];
                    acc1[q] = __builtin_fmaf(w1, xv, acc1[q]);
                    acc2[q] = __builtin_fmaf(w2, xv, acc2[q]);
                }
            }
        } else {
            float wa[KB];
#pragma unroll
            for (int u = 0; u < KB; u++) wa[u] = wT[(size_t)u * cells + c1];
            int k = 0;
            for (; k + KB <= K; k += KB) {
                float na[KB];
                const int kn = k + KB + KB <= K ? k + KB : k;
#pragma unroll
                for (int u = 0; u < KB; u++) na[u] = wT[(size_t)(kn + u) * cells + c1];
#pragma unroll
                for (int u = 0; u < KB; u++) {
#pragma unroll
                    for (int q = 0; q < HB; q++) acc1[q] = __builtin_fmaf(wa[u], x[q * K + k + u], acc1[q]);
                }
#pragma unroll
                for (int u = 0; u < KB; u++) wa[u] = na[u];
            }
            if (k + KH <= K) {
                float ra[KH];
#pragma unroll
                for (int u = 0; u < KH; u++) ra[u] = wT[(size_t)(k + u) * cells + c1];
#pragma unroll
                for (int u = 0; u < KH; u++) {
#pragma unroll
                    for (int q = 0; q < HB; q++) acc1[q] = __builtin_fmaf(ra[u], x[q * K + k + u], acc1[q]);
                }
                k += KH;
            }
            for (; k < K; k++) {
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
    {      // (r5: sixteen weights at a time, one block ahead, as in the policy head; same order of the multiply-adds)
        constexpr int KB = 16, KH = 8;
        float wa[KB];
#pragma unroll
        for (int u = 0; u < KB; u++) wa[u] = w1T[(size_t)u * 256 + hdn];
        int k = 0;
        for (; k + KB <= cells; k += KB) {
            float na[KB];
            const int kn = k + KB + KB <= cells ? k + KB : k;
#pragma unroll
            for (int u = 0; u < KB; u++) na[u] = w1T[(size_t)(kn + u) * 256 + hdn];
#pragma unroll
            for (int u = 0; u < KB; u++) {
#pragma unroll
                for (int q = 0; q < HB; q++) acc[q] = __builtin_fmaf(wa[u], x[q * cells + k + u], acc[q]);
            }
#pragma unroll
            for (int u = 0; u < KB; u++) wa[u] = na[u];
        }
        if (k + KH <= cells) {      // (361 = 22 x 16 + 8 + 1)
            float ra[KH];
#pragma unroll
            for (int u = 0; u < KH; u++) ra[u] = w1T[(size_t)(k + u) * 256 + hdn];
#pragma unroll
            for (int u = 0; u < KH; u++) {
#pragma unroll
                for (int q = 0; q < HB; q++) acc[q] = __builtin_fmaf(ra[u], x[q * cells + k + u], acc[q]);
            }
            k += KH;
        }
        for (; k < cells; k++) {
            const float w = w1T[(size_t)k * 256 + hdn];
#pragma unroll
            for (int q = 0; q < HB; q++) acc[q] = __builtin_fmaf(w, x[q * cells + k], acc[q]);
        }
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
static hipEvent_t timer_event(C6NetTimers *t) {
    hipEvent_t e;
    if (!t->pool.empty()) { e = t->pool.back(); t->pool.pop_back(); }
    else hipEventCreate(&e);
    return e;
}

void c6_net_forward_dev(const C6NetDev &net, int nleaf, int images, const int8_t *boards, int board_stride,
                        const int8_t *player, int player_stride, const int16_t *aug_map, float *act0, float *act1, float *act2, float *vbuf, float *ph,
                        float *vh, float *pol, float *val, hipStream_t s, C6NetTimers *timers) {
    const int nb = nleaf * images, cells = net.cells, C = net.C;
    if (nb == 0) return;
    float *x = act0, *y = act1, *z = act2;
    const bool timed = timers && timers->enabled;
    // The forms that lost their A/B comparisons are compile-time options of A/B builds only (tools/ab_build.sh -DC6_AB_...=...):
    // the product has one path per arithmetic.
    constexpr bool fuse_heads = C6_AB_HEAD_FUSE != 0;          // 0: the heads' 1x1 convolutions as a kernel of their own
    constexpr int mid_mode = C6_AB_WINO_MID;                   // boundary kernel: 0 off, 1 inside blocks, 2 also across blocks
    constexpr bool fuse_mid = mid_mode != 0;
    constexpr int stem_mode = C6_AB_WINO_STEM;                 // 0 the stem as a kernel of its own, 1 never stored (computed twice), 2 computed by the
    constexpr bool fuse_stem = stem_mode != 0;                 // first input transform and stored for the residual
    bool head_fused = false, heads_done = false, v_ready = false;
    // FP32 Winograd F(4x4) with at least two residual blocks: the stem's activations are never stored -- the first layer's input transform
    // computes them from the boards, and so does the boundary kernel behind the first block, which needs them as its residual
    const C6StemSrc stem_src{boards, board_stride, player, player_stride, aug_map, images, net.stem_w, net.stem_b};
    bool stem_inline = false;
    // (the boundary kernel across blocks must be on)
    if (net.variant == C6_CONV_WINO && fuse_stem && mid_mode >= 2 && net.blocks - 1 >= 2 && c6_conv_wino4_vbuf_floats(C, net.N, 1) > 0) stem_inline = true;
    if (!stem_inline)
        hipLaunchKernelGGL(c6_stem_kernel, dim3(nb), dim3(256), 0, s, boards, board_stride, player, player_stride, aug_map, images, net.stem_w,
                           net.stem_b, act0, net.N, C);
    if (net.variant == C6_CONV_WINO && net.wino6) {
        // 19x19, C = 256, >= 2 residual blocks: conv_wino6.hip (tiles six pixels wide, pipelined boundary kernel).  The chain of a pass:
        //   stem-in -> [GEMM -> boundary]* -> GEMM -> last output transform + heads; activations are stored only at block ends (the next
        //   block's residual), the stem's never (recomputed where it is the residual)
        const int Lw = 2 * (net.blocks - 1);
        const size_t wwf = c6_w6_packed_floats();
        const C6HeadFuse hf{net.head_w, net.pc_b, net.vc_b, net.vn_scale, net.vn_shift, ph, vh};
        for (int l = 0; l < Lw; l++) {
            const int half = l & 1;
            hipEvent_t e0 = nullptr, e1 = nullptr, em = nullptr, eg = nullptr;
            if (timed) { e0 = timer_event(timers); e1 = timer_event(timers); hipEventRecord(e0, s); }
            if (l == 0 && !c6_w6_dispatch(net.N, nb, nullptr, nullptr, nullptr, nullptr, vbuf, net.num_cus, s, 1, nullptr, &stem_src)) return;      // (noted: c6_launch_status)
            if (timed) {
                em = timer_event(timers);
                hipEventRecord(em, s);
                timers->pre.emplace_back(e0, em);
                timers->exec_flops += c6_w6_exec_flops(net.N, nb);
            }
            if (!c6_w6_dispatch(net.N, nb, net.conv_ww + (size_t)l * wwf, nullptr, nullptr, nullptr, vbuf, net.num_cus, s, 2)) return;
            if (timed) { eg = timer_event(timers); hipEventRecord(eg, s); }
            const float *cb = net.conv_b + (size_t)l * C;
            // residual of the block's second convolution: the block's input = the previous block's output (x), or the stem for block 0
            const float *cres = half == 0 ? nullptr : x;
            if (l < Lw - 1) {
                float *cout = half == 0 ? nullptr : z;      // a block's output is stored: the next block's residual
                if (!c6_w6_dispatch(net.N, nb, nullptr, cb, l == 1 ? nullptr : cres, cout, vbuf, net.num_cus, s, 16, nullptr, l == 1 ? &stem_src : nullptr)) return;
            } else
                if (!c6_w6_dispatch(net.N, nb, nullptr, cb, Lw == 2 ? nullptr : cres, nullptr, vbuf, net.num_cus, s, 4, &hf)) return;
            if (timed) {
                hipEvent_t eo = timer_event(timers);
                hipEventRecord(eo, s);
                timers->post.emplace_back(eg, eo);
                hipEventRecord(e1, s);
                timers->conv.emplace_back(e0, e1);
                timers->conv_flops += 2.0 * 9.0 * C * C * cells * (double)nb;
            }
            if (half == 1) { float *t = x; x = z; z = t; }
        }
        heads_done = true;
    } else
    for (int i = 0; i < net.blocks - 1; i++) {     // res_block, model.hpp:61-78
        for (int half = 0; half < 2; half++) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (timed) { e0 = timer_event(timers); e1 = timer_event(timers); hipEventRecord(e0, s); }
            const int l = 2 * i + half;
            const float *cin = half == 0 ? x : y, *cres = half == 0 ? nullptr : x;
            float *cout = half == 0 ? y : z;
            const float *cb = net.conv_b + (size_t)l * C;
            if (net.variant == C6_CONV_WINO) {
                if (l == 0 && stem_inline) c6_conv_wino_stem_transform(C, net.N, nb, stem_src, vbuf, stem_mode == 2 ? x : nullptr, net.num_cus, s);      // straight from the boards
                else if (!v_ready) c6_conv_wino_transform(C, net.N, nb, cin, vbuf, net.num_cus, s);      // else: the previous layer's fused transform left V
                v_ready = false;
                if (timed) {
                    hipEvent_t em = timer_event(timers);
                    hipEventRecord(em, s);
                    timers->pre.emplace_back(e0, em);
                    timers->exec_flops += c6_conv_wino_exec_flops(C, net.N, nb);
                }
                c6_conv_wino_gemm(C, net.N, nb, vbuf, net.conv_ww + (size_t)l * c6_conv_packed_wino_floats(C), cb, cres, cout, net.num_cus, s, 1);
                hipEvent_t eg = nullptr;
                if (timed) { eg = timer_event(timers); hipEventRecord(eg, s); }
                // first convolution of a block (no residual, one reader): its output transform and the next layer's input transform
                // are one kernel and the activations between them are never stored (F(4x4) form)
                bool has_out = false;
                const bool last_layer = l == 2 * (net.blocks - 1) - 1;
                if (fuse_mid && !last_layer && (half == 0 || mid_mode >= 2) &&
                    c6_conv_wino_mid(C, net.N, nb, vbuf, cb, half == 0 ? nullptr : cres, half == 0 ? nullptr : cout, net.num_cus, s,
                                     stem_inline && stem_mode != 2 && l == 1 ? &stem_src : nullptr)) {
                    v_ready = true;
                    has_out = true;
                } else {
                    // the tower's last output transform applies the heads' 1x1 filters itself (F(4x4) form) and stores no activations
                    const bool last = l == 2 * (net.blocks - 1) - 1;
                    const C6HeadFuse hf{net.head_w, net.pc_b, net.vc_b, net.vn_scale, net.vn_shift, ph, vh};
                    const int out_rc = c6_conv_wino_out(C, net.N, nb, vbuf, cb, cres, cout, net.num_cus, s, 1, last && fuse_heads ? &hf : nullptr);
                    has_out = out_rc != 0;
                    if (out_rc == 2) heads_done = true;
                }
                if (timed) {
                    if (has_out) {
                        hipEvent_t eo = timer_event(timers);
                        hipEventRecord(eo, s);
                        timers->post.emplace_back(eg, eo);
                    } else
                        timers->pool.push_back(eg);
                }
            }
            else if (net.variant == C6_CONV_F32N)
                c6_conv_f32n_dispatch(C, net.N, nb, cin, net.conv_w5 + (size_t)l * c6_conv_packed_f32n_floats(C), cb, cres, cout, net.num_cus, s);
            else if (net.variant == C6_CONV_B3S || net.variant == C6_CONV_BF16) {
                // the last convolution keeps its activations to itself and hands the heads their 1x1 partial sums
                head_fused = fuse_heads && l == 2 * (net.blocks - 1) - 1;
                c6_conv_b3s_dispatch(C, net.N, nb, cin, net.conv_w7 + (size_t)l * c6_conv_packed_b3s_units(C), cb, cres, cout, net.num_cus, s, 1,
                                     net.variant == C6_CONV_B3S, head_fused ? net.head_w : nullptr);
            } else {      // C6_CONV_M8
                head_fused = fuse_heads && l == 2 * (net.blocks - 1) - 1;
                c6_conv_m8_dispatch(C, net.N, nb, cin, net.conv_w9 + (size_t)l * c6_conv_packed_m8_bytes(C), cb, cres, cout, net.num_cus, s, 1,
                                    head_fused ? net.head_w : nullptr, net.range_flag, C6_M8_RANGE_HI);
                if (net.range_flag && !head_fused)        // a sample of boards: each one's output must reach the lower end of the range
                    c6_launch_range_sample(cout, nb < 32 ? nb : 32, (size_t)cells * C, C6_M8_RANGE_LO, C6_M8_RANGE_HI, net.range_flag, s);
            }
            if (timed) {
                hipEventRecord(e1, s);
                timers->conv.emplace_back(e0, e1);
                timers->conv_flops += 2.0 * 9.0 * C * C * cells * (double)nb;
                if (net.variant != C6_CONV_WINO) timers->exec_flops += 2.0 * 9.0 * C * C * (double)nb * (net.variant == C6_CONV_F32N ? 32 * ((net.N * (net.N + 1) + 31) / 32) : cells);
            }
        }
        float *t = x; x = z; z = t;
    }
    if (heads_done) {
    } else if (head_fused)
        hipLaunchKernelGGL(c6_head_reduce_kernel, dim3((unsigned)(((size_t)nb * cells + 255) / 256)), dim3(256), 0, s, x, net.pc_b, net.vc_b,
                           net.vn_scale, net.vn_shift, ph, vh, net.N, C / 32, c6_conv_b3s_head_rows(net.N), nb);
    else
        hipLaunchKernelGGL(c6_head_conv_kernel, dim3(nb), dim3(256), 0, s, x, net.pc_w, net.pc_b, net.vc_w, net.vc_b, net.vn_scale,
                           net.vn_shift, ph, vh, cells, C);
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
    for (auto &p : t->pre) hipEventDestroy(p.second);      // .first is shared with the layer's pair
    for (auto &p : t->post) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto e : t->pool) hipEventDestroy(e);
    delete t;
}
void c6_timers_enable(C6NetTimers *t, bool on) { t->enabled = on; }
// drains finished event pairs; returns (ms, launches, algorithmic flops) and the Winograd extras (input-transform ms, executed flops)
void c6_timers_collect(C6NetTimers *t, double *ms, unsigned long long *launches, double *flops, double *pre_ms, double *exec_flops, double *post_ms,
                       bool reset) {
    for (auto &p : t->post) {
        float m = 0.f;
        hipEventSynchronize(p.second);
        hipEventElapsedTime(&m, p.first, p.second);
        t->post_ms += m;
        t->pool.push_back(p.first);
        t->pool.push_back(p.second);
    }
    t->post.clear();
    for (auto &p : t->pre) {      // before the layer pairs: they recycle the shared start events
        float m = 0.f;
        hipEventSynchronize(p.second);
        hipEventElapsedTime(&m, p.first, p.second);
        t->pre_ms += m;
        t->pool.push_back(p.second);
    }
    t->pre.clear();
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
    *pre_ms = t->pre_ms;
    *post_ms = t->post_ms;
    *exec_flops = t->exec_flops;
    if (reset) { t->acc_ms = 0; t->acc_n = 0; t->conv_flops = 0; t->pre_ms = 0; t->post_ms = 0; t->exec_flops = 0; }
}

// range guard, second half: max |activation| of each board of a sample (one workgroup per board; see c6_engine.h)
__global__ __launch_bounds__(1024) void c6_range_sample_kernel(const float *__restrict__ act, size_t board_floats, float lo, float hi, unsigned int *flag) {
    __shared__ float red[16];
    const f32x4 *a4 = reinterpret_cast<const f32x4 *>(act + (size_t)blockIdx.x * board_floats);
    float mx = 0.f;
    bool bad = false;
    for (size_t i = threadIdx.x; i < board_floats / 4; i += 1024) {
        const f32x4 v = a4[i];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float a = __builtin_fabsf(v[r]);
            bad = bad || !(a <= hi);
            mx = a > mx ? a : mx;
        }
    }
    mx = c6_wave_max_f32(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    const bool anybad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int w = 0; w < 16; w++) m = red[w] > m ? red[w] : m;
        unsigned int f = 0;
        if (anybad) f |= 1u;
        if (m > 0.f && m < lo) f |= 2u;
        if (f) atomicOr(flag, f);
    }
}
void c6_launch_range_sample(const float *act, int nboards, size_t board_floats, float range_lo, float range_hi, unsigned int *range_flag, hipStream_t s) {
    if (nboards > 0) hipLaunchKernelGGL(c6_range_sample_kernel, dim3(nboards), dim3(1024), 0, s, act, board_floats, range_lo, range_hi, range_flag);
}

// ---- micro-benchmark hook (c6_bench_conv in engine.cpp): one conv launch of a kernel on random data ----
void c6_conv_bench_launch(int variant, int C, int N, int nboards, const float *in, const float *w, const float *bias, const float *res,
                          float *out, float *vbuf, int num_cus, hipStream_t s) {
    if (variant == C6_CONV_WINO) c6_conv_wino_dispatch(C, N, nboards, in, w, bias, res, out, vbuf, num_cus, s, 1);
    else if (variant == C6_CONV_F32N) c6_conv_f32n_dispatch(C, N, nboards, in, w, bias, res, out, num_cus, s);
    else if (variant == C6_CONV_B3S || variant == C6_CONV_BF16)
        c6_conv_b3s_dispatch(C, N, nboards, in, reinterpret_cast<const unsigned short *>(w), bias, res, out, num_cus, s, 1, variant == C6_CONV_B3S);
    else if (variant == C6_CONV_M8) c6_conv_m8_dispatch(C, N, nboards, in, reinterpret_cast<const unsigned char *>(w), bias, res, out, num_cus, s, 1, nullptr);
}
