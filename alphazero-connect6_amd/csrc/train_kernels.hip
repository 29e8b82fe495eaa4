// train_kernels.hip -- the training step on the other side of self-play.
//
// Reference being reproduced: WeightedPolicy::forward in train mode and ::loss (AlphaZero/cpp_weighted/model.hpp:52-108)
// differentiated by libtorch autograd, torch::optim::SGD with momentum (AlphaZero/cpp_weighted/main.cpp:107-109,151-157).
// Everything is FP32 and every reduction has a fixed order (chunk partials, then one thread per output sums the chunks
// in index order), so a step is bit-reproducible.
//
// Layouts: activations NHWC [B][cells][C] as in the evaluator; the policy head's pre-norm output is [B*cells][2], its
// flattened post-norm input of the linear layer [B][2*cells] (channel-major like the reference's view); parameters,
// gradients and momentum are three blobs in the documented order (oracle/c6_oracle_net.c).
//
// 3x3 convolutions: the forward pass and the input gradient run on the evaluator's split-bf16 kernel
// (c6_conv3x3_b3s_kernel with relu = 0, bias folded out, weights packed on the device each step by c6_pack_b3s_kernel --
// transposed and flipped for the input gradient); the weight gradient is c6_wgrad_b3_kernel below (split bf16) or, with
// C6_PREC_FP32, c6_wgrad_kernel (exact FP32 MFMA) next to the exact-FP32 convolution c6_conv3x3_f32n_kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "c6_train.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BN_EPS 1e-5f
#ifndef C6_AB_WGRAD_WS
#define C6_AB_WGRAD_WS 1          // A/B builds only: 0 = c6_wgrad_b3_kernel (two workgroups per CU) instead of the wave-specialised kernel
#endif

// ------------------------------------------------------------------------------------------------------------------
// weights OIHW [C][C][3][3] -> the b3s layout [stage = ks*9 + tap][wave][ct][plane][lane][8] (conv_bf16x3.hip), on the
// device.  transpose_flip: pack W'[n = ci][k = co][tap] = W[co][ci][8 - tap], the kernel of the input gradient.
// ------------------------------------------------------------------------------------------------------------------
__global__ void c6_pack_b3s_kernel(int C, const float *__restrict__ w, unsigned short *__restrict__ packed, int transpose_flip) {
    const int WAVES = C / 32, KS = C / 32;
    const size_t total = (size_t)KS * 9 * WAVES * 2 * 64 * 8;       // elements per plane
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int j = (int)(r % 8); r /= 8;
        const int lane = (int)(r % 64); r /= 64;
        const int ct = (int)(r % 2); r /= 2;
        const int wv = (int)(r % WAVES); r /= WAVES;
        const int t = (int)(r % 9); r /= 9;
        const int ks = (int)r;
        const int n = wv * 32 + ct * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + j;
        const float x = transpose_flip ? w[((size_t)k * C + n) * 9 + (8 - t)] : w[((size_t)n * C + k) * 9 + t];
        const __bf16 hi = (__bf16)x;
        const __bf16 mid = (__bf16)(x - (float)hi);
        const size_t base = ((((size_t)(ks * 9 + t) * WAVES + wv) * 2 + ct) * 2) * 512;
        packed[base + 0 * 512 + lane * 8 + j] = __builtin_bit_cast(unsigned short, hi);
        packed[base + 1 * 512 + lane * 8 + j] = __builtin_bit_cast(unsigned short, mid);
    }
}
void c6t_pack_b3s(int C, const float *w, unsigned short *packed, int transpose_flip, hipStream_t s) {
    hipLaunchKernelGGL(c6_pack_b3s_kernel, dim3(512), dim3(256), 0, s, C, w, packed, transpose_flip);
}

// the same for the exact-FP32 kernel (c6_conv3x3_kernel, net_kernels.hip): [stage = ks*9 + tap][kq][h][n][r],
// k = ks*32 + kq*8 + h*4 + r
__global__ void c6_pack_f32n_kernel(int C, const float *__restrict__ w, float *__restrict__ packed, int transpose_flip) {
    const int WAVES = C / 32;
    const size_t total = (size_t)9 * C * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t q = e;
        const int r = (int)(q % 4); q /= 4;
        const int lane = (int)(q % 64); q /= 64;
        const int u = (int)(q % 4); q /= 4;
        const int wv = (int)(q % WAVES); q /= WAVES;
        const int t = (int)(q % 9); q /= 9;
        const int ks = (int)q;
        const int n = wv * 32 + (lane & 31), k = ks * 32 + 8 * u + 4 * (lane >> 5) + r;
        packed[e] = transpose_flip ? w[((size_t)k * C + n) * 9 + (8 - t)] : w[((size_t)n * C + k) * 9 + t];
    }
}
void c6t_pack_f32n(int C, const float *w, float *packed, int transpose_flip, hipStream_t s) {
    hipLaunchKernelGGL(c6_pack_f32n_kernel, dim3(512), dim3(256), 0, s, C, w, packed, transpose_flip);
}

// ------------------------------------------------------------------------------------------------------------------
// stem: conv 2 -> C on the planes [player broadcast, board] (model.hpp:56-59), raw output (BatchNorm follows)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void c6t_stem_fwd_kernel(int N, int C, const float *__restrict__ boards, const float *__restrict__ players,
                                                           const float *__restrict__ w /*[C][2][3][3]*/, const float *__restrict__ bias,
                                                           float *__restrict__ y) {
    extern __shared__ float sm[];
    const int P = N + 2, cells = N * N;
    float *pl = sm, *bd = sm + P * P;
    const size_t b = blockIdx.x;
    const float pv = players[b];
    for (int s = threadIdx.x; s < P * P; s += blockDim.x) { pl[s] = 0.f; bd[s] = 0.f; }
    __syncthreads();
    for (int c = threadIdx.x; c < cells; c += blockDim.x) {
        const int s = (c / N + 1) * P + (c % N + 1);
        pl[s] = pv;
        bd[s] = boards[b * cells + c];
    }
    __syncthreads();
    for (int co = threadIdx.x; co < C; co += blockDim.x) {
        float w0[9], w1[9];
#pragma unroll
        for (int t = 0; t < 9; t++) { w0[t] = w[((size_t)co * 2 + 0) * 9 + t]; w1[t] = w[((size_t)co * 2 + 1) * 9 + t]; }
        const float bi = bias[co];
        for (int p = 0; p < cells; p++) {
            const int s = (p / N + 1) * P + (p % N + 1);
            float acc = bi;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int o = s + (t / 3 - 1) * P + (t % 3 - 1);
                acc = __builtin_fmaf(w0[t], pl[o], acc);
                acc = __builtin_fmaf(w1[t], bd[o], acc);
            }
            y[(b * cells + p) * (size_t)C + co] = acc;
        }
    }
}
void c6t_stem_fwd(int N, int C, int B, const float *boards, const float *players, const float *w, const float *bias, float *y, hipStream_t s) {
    const int P = N + 2;
    hipLaunchKernelGGL(c6t_stem_fwd_kernel, dim3(B), dim3(256), 2 * P * P * sizeof(float), s, N, C, boards, players, w, bias, y);
}

// dW[co][ch][tap] and db[co] of the stem from dy[B][cells][C]: partial[chunk][19][C] per chunk of boards
#define STEM_BWD_BOARDS 8
__global__ __launch_bounds__(256) void c6t_stem_bwd_kernel(int N, int C, int B, const float *__restrict__ boards, const float *__restrict__ players,
                                                           const float *__restrict__ dy, float *__restrict__ partial) {
    extern __shared__ float sm[];
    const int P = N + 2, cells = N * N;
    float *pl = sm, *bd = sm + P * P;
    const int chunk = blockIdx.x;
    float acc[19];
    for (int co0 = 0; co0 < C; co0 += blockDim.x) {
        const int co = co0 + threadIdx.x;
#pragma unroll
        for (int q = 0; q < 19; q++) acc[q] = 0.f;
        for (int bb = 0; bb < STEM_BWD_BOARDS; bb++) {
            const size_t b = (size_t)chunk * STEM_BWD_BOARDS + bb;
            if (b >= (size_t)B) break;
            __syncthreads();
            const float pv = players[b];
            for (int s = threadIdx.x; s < P * P; s += blockDim.x) { pl[s] = 0.f; bd[s] = 0.f; }
            __syncthreads();
            for (int c = threadIdx.x; c < cells; c += blockDim.x) {
                const int s = (c / N + 1) * P + (c % N + 1);
                pl[s] = pv;
                bd[s] = boards[b * cells + c];
            }
            __syncthreads();
            if (co < C)
                for (int p = 0; p < cells; p++) {
                    const int s = (p / N + 1) * P + (p % N + 1);
                    const float g = dy[(b * cells + p) * (size_t)C + co];
#pragma unroll
                    for (int t = 0; t < 9; t++) {
                        const int o = s + (t / 3 - 1) * P + (t % 3 - 1);
                        acc[t] = __builtin_fmaf(g, pl[o], acc[t]);
                        acc[9 + t] = __builtin_fmaf(g, bd[o], acc[9 + t]);
                    }
                    acc[18] += g;
                }
        }
        if (co < C)
#pragma unroll
            for (int q = 0; q < 19; q++) partial[((size_t)chunk * 19 + q) * C + co] = acc[q];
    }
}
// dw[co][2][9] and db[co] = sum of the chunk partials in chunk order
__global__ void c6t_stem_bwd_finalize_kernel(int C, int nchunk, const float *__restrict__ partial, float *__restrict__ dw, float *__restrict__ db) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 19 * C) return;
    const int q = idx / C, co = idx % C;
    float s = 0.f;
    for (int k = 0; k < nchunk; k++) s += partial[((size_t)k * 19 + q) * C + co];
    if (q == 18) db[co] = s;
    else dw[((size_t)co * 2 + q / 9) * 9 + q % 9] = s;
}
void c6t_stem_bwd(int N, int C, int B, const float *boards, const float *players, const float *dy, float *partial, float *dw, float *db,
                  hipStream_t s) {
    const int P = N + 2, nchunk = (B + STEM_BWD_BOARDS - 1) / STEM_BWD_BOARDS;
    hipLaunchKernelGGL(c6t_stem_bwd_kernel, dim3(nchunk), dim3(256), 2 * P * P * sizeof(float), s, N, C, B, boards, players, dy, partial);
    hipLaunchKernelGGL(c6t_stem_bwd_finalize_kernel, dim3((19 * C + 255) / 256), dim3(256), 0, s, C, nchunk, partial, dw, db);
}
size_t c6t_stem_bwd_partial_floats(int C, int B) { return (size_t)((B + STEM_BWD_BOARDS - 1) / STEM_BWD_BOARDS) * 19 * C; }

// ------------------------------------------------------------------------------------------------------------------
// column statistics of a row-major matrix x[R][C]: up to three weighted sums per column in one pass
//   mode 0 (forward BatchNorm):  s0 = sum x, s1 = sum x^2
//   mode 1 (BatchNorm backward): dz = dout * (aout > 0 or no mask); s0 = sum dz, s1 = sum dz * (y - mean) * invstd
//   mode 2 (plain column sum):   s0 = sum x
// partial[chunk][2][C]; finalize sums the chunks in order (double accumulators).
// ------------------------------------------------------------------------------------------------------------------
#define CS_ROWS 256
__global__ __launch_bounds__(256) void c6t_colstats_kernel(int mode, size_t R, int C, const float *__restrict__ x, const float *__restrict__ aout,
                                                           const float *__restrict__ y, const float *__restrict__ mean,
                                                           const float *__restrict__ var, float *__restrict__ partial) {
    const size_t r0 = (size_t)blockIdx.x * CS_ROWS;
    const size_t r1 = r0 + CS_ROWS < R ? r0 + CS_ROWS : R;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s0 = 0.f, s1 = 0.f;
        if (mode == 0) {
            for (size_t r = r0; r < r1; r++) { const float v = x[r * C + c]; s0 += v; s1 = __builtin_fmaf(v, v, s1); }
        } else if (mode == 1) {
            const float m = mean[c], is = 1.0f / sqrtf(var[c] + BN_EPS);
            for (size_t r = r0; r < r1; r++) {
                float dz = x[r * C + c];
                if (aout && !(aout[r * C + c] > 0.f)) dz = 0.f;
                s0 += dz;
                s1 = __builtin_fmaf(dz, (y[r * C + c] - m) * is, s1);
            }
        } else {
            for (size_t r = r0; r < r1; r++) s0 += x[r * C + c];
        }
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = s0;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = s1;
    }
}
// the same for the tower's shapes (C = 64 / 128 / 256): a thread owns FOUR channels (16-byte loads: 1 KiB per wave
// instruction) and every RL-th row of the chunk, RL = 256 / (C / 4) row lanes; the row lanes' sums are added in lane order
__global__ __launch_bounds__(256) void c6t_colstats4_kernel(int mode, size_t R, int C, const float *__restrict__ x, const float *__restrict__ aout,
                                                            const float *__restrict__ y, const float *__restrict__ mean,
                                                            const float *__restrict__ var, float *__restrict__ partial) {
    __shared__ f32x4 red[2][256];
    const int CQ = C / 4, RL = 256 / CQ;
    const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ, c = cq * 4;
    const size_t r0 = (size_t)blockIdx.x * CS_ROWS;
    const size_t r1 = r0 + CS_ROWS < R ? r0 + CS_ROWS : R;
    f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (mode == 0) {
        for (size_t r = r0 + rl; r < r1; r += RL) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + r * C + c);
            s0 = s0 + v;
#pragma unroll
            for (int q = 0; q < 4; q++) s1[q] = __builtin_fmaf(v[q], v[q], s1[q]);
        }
    } else if (mode == 1) {
        const f32x4 m = *reinterpret_cast<const f32x4 *>(mean + c), vv = *reinterpret_cast<const f32x4 *>(var + c);
        f32x4 is;
#pragma unroll
        for (int q = 0; q < 4; q++) is[q] = 1.0f / sqrtf(vv[q] + BN_EPS);
        for (size_t r = r0 + rl; r < r1; r += RL) {
            f32x4 dz = *reinterpret_cast<const f32x4 *>(x + r * C + c);
            const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + r * C + c);
            if (aout) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(aout + r * C + c);
#pragma unroll
                for (int q = 0; q < 4; q++) if (!(a[q] > 0.f)) dz[q] = 0.f;
            }
            s0 = s0 + dz;
#pragma unroll
            for (int q = 0; q < 4; q++) s1[q] = __builtin_fmaf(dz[q], (yv[q] - m[q]) * is[q], s1[q]);
        }
    } else {
        for (size_t r = r0 + rl; r < r1; r += RL) s0 = s0 + *reinterpret_cast<const f32x4 *>(x + r * C + c);
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < RL; l++) { s0 = s0 + red[0][l * CQ + cq]; s1 = s1 + red[1][l * CQ + cq]; }
        *reinterpret_cast<f32x4 *>(partial + ((size_t)blockIdx.x * 2 + 0) * C + c) = s0;
        *reinterpret_cast<f32x4 *>(partial + ((size_t)blockIdx.x * 2 + 1) * C + c) = s1;
    }
}
// mode 0: out0 = mean, out1 = biased variance; otherwise out0 = s0, out1 = s1.  32 columns x 32 chunk lanes per
// workgroup: lane ky sums chunks ky, ky + 32, ... in order, the 32 lane sums are then added in lane order (fixed order,
// double accumulators).  (8 chunk lanes took 70 us per call, 65 calls per step: the chain of ~225 dependent adds per lane.)
#define CSF_LANES 32
__global__ __launch_bounds__(32 * CSF_LANES) void c6t_colstats_finalize_kernel(int mode, size_t R, int C, int nchunk, const float *__restrict__ partial,
                                                                              float *__restrict__ out0, float *__restrict__ out1) {
    __shared__ double l0[CSF_LANES][32], l1[CSF_LANES][32];
    const int cx = threadIdx.x & 31, ky = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s0 = 0.0, s1 = 0.0;
    if (c < C)
        for (int k = ky; k < nchunk; k += CSF_LANES) { s0 += (double)partial[((size_t)k * 2 + 0) * C + c]; s1 += (double)partial[((size_t)k * 2 + 1) * C + c]; }
    l0[ky][cx] = s0; l1[ky][cx] = s1;
    __syncthreads();
    if (ky != 0 || c >= C) return;
    s0 = 0.0; s1 = 0.0;
    for (int q = 0; q < CSF_LANES; q++) { s0 += l0[q][cx]; s1 += l1[q][cx]; }
    if (mode == 0) {
        const double m = s0 / (double)R;
        double v = s1 / (double)R - m * m;
        if (v < 0.0) v = 0.0;
        out0[c] = (float)m;
        out1[c] = (float)v;
    } else {
        out0[c] = (float)s0;
        if (out1) out1[c] = (float)s1;
    }
}
// the ordered finalisation alone, for partials a producing kernel left (r6: c6_wino4_out_kernel<., false, true>): [nchunk][2][C]
void c6t_colstats_finalize(int mode, size_t R, int C, int nchunk, const float *partial, float *out0, float *out1, hipStream_t s) {
    hipLaunchKernelGGL(c6t_colstats_finalize_kernel, dim3((C + 31) / 32), dim3(32 * CSF_LANES), 0, s, mode, R, C, nchunk, partial, out0, out1);
}
size_t c6t_colstats_partial_floats(size_t R, int C) { return ((R + CS_ROWS - 1) / CS_ROWS) * 2 * (size_t)C; }
void c6t_colstats(int mode, size_t R, int C, const float *x, const float *aout, const float *y, const float *mean, const float *var,
                  float *partial, float *out0, float *out1, hipStream_t s) {
    const int nchunk = (int)((R + CS_ROWS - 1) / CS_ROWS);
    if (C == 64 || C == 128 || C == 256) hipLaunchKernelGGL(c6t_colstats4_kernel, dim3(nchunk), dim3(256), 0, s, mode, R, C, x, aout, y, mean, var, partial);
    else hipLaunchKernelGGL(c6t_colstats_kernel, dim3(nchunk), dim3(256), 0, s, mode, R, C, x, aout, y, mean, var, partial);
    hipLaunchKernelGGL(c6t_colstats_finalize_kernel, dim3((C + 31) / 32), dim3(32 * CSF_LANES), 0, s, mode, R, C, nchunk, partial, out0, out1);
}

// a = relu?( (y - mean) * invstd * gamma + beta (+ res) )      (train-mode BatchNorm with the batch statistics)
__global__ void c6t_bn_apply_kernel(size_t total, int C, const float *__restrict__ y, const float *__restrict__ mean, const float *__restrict__ var,
                                    const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ res,
                                    float *__restrict__ a, int relu) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        float v = (y[e] - mean[c]) * (1.0f / sqrtf(var[c] + BN_EPS)) * gamma[c] + beta[c];
        if (res) v += res[e];
        a[e] = relu ? (v > 0.f ? v : 0.f) : v;
    }
}
// the same, four channels per thread (16-byte loads and stores; C a multiple of 4: the tower's shapes) -- the same expression per element, so the same bits
__global__ __launch_bounds__(256) void c6t_bn_apply4_kernel(size_t total4, int C, const float *__restrict__ y, const float *__restrict__ mean, const float *__restrict__ var,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ res,
                                                            float *__restrict__ a, int relu) {
    const int CQ = C / 4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % CQ) * 4;
        const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + e * 4), m = *reinterpret_cast<const f32x4 *>(mean + c), vv = *reinterpret_cast<const f32x4 *>(var + c),
                    g = *reinterpret_cast<const f32x4 *>(gamma + c), b = *reinterpret_cast<const f32x4 *>(beta + c);
        f32x4 rv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (res) rv = *reinterpret_cast<const f32x4 *>(res + e * 4);
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float v = (yv[q] - m[q]) * (1.0f / sqrtf(vv[q] + BN_EPS)) * g[q] + b[q];
            if (res) v += rv[q];
            o[q] = relu ? (v > 0.f ? v : 0.f) : v;
        }
        *reinterpret_cast<f32x4 *>(a + e * 4) = o;
    }
}
void c6t_bn_apply(size_t R, int C, const float *y, const float *mean, const float *var, const float *gamma, const float *beta, const float *res,
                  float *a, int relu, hipStream_t s) {
    if (C % 64 == 0) hipLaunchKernelGGL(c6t_bn_apply4_kernel, dim3(2048), dim3(256), 0, s, R * C / 4, C, y, mean, var, gamma, beta, res, a, relu);
    else hipLaunchKernelGGL(c6t_bn_apply_kernel, dim3(2048), dim3(256), 0, s, R * C, C, y, mean, var, gamma, beta, res, a, relu);
}

// BatchNorm backward: dz = dout * (aout > 0); dy = gamma * invstd * (dz - sum_dz / R - xhat * sum_dzx / R); optionally
// stores dz (the gradient that also flows into the skip connection)
__global__ void c6t_bn_bwd_apply_kernel(size_t total, size_t R, int C, const float *__restrict__ dout, const float *__restrict__ aout,
                                        const float *__restrict__ y, const float *__restrict__ mean, const float *__restrict__ var,
                                        const float *__restrict__ gamma, const float *__restrict__ sum_dz, const float *__restrict__ sum_dzx,
                                        float *__restrict__ dy, float *__restrict__ dz_out) {
    const float invR = 1.0f / (float)R;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        float dz = dout[e];
        if (aout && !(aout[e] > 0.f)) dz = 0.f;
        const float is = 1.0f / sqrtf(var[c] + BN_EPS);
        const float xh = (y[e] - mean[c]) * is;
        dy[e] = gamma[c] * is * (dz - sum_dz[c] * invR - xh * sum_dzx[c] * invR);
        if (dz_out) dz_out[e] = dz;
    }
}
void c6t_bn_bwd_apply(size_t R, int C, const float *dout, const float *aout, const float *y, const float *mean, const float *var,
                      const float *gamma, const float *sum_dz, const float *sum_dzx, float *dy, float *dz_out, hipStream_t s) {
    hipLaunchKernelGGL(c6t_bn_bwd_apply_kernel, dim3(2048), dim3(256), 0, s, R * C, R, C, dout, aout, y, mean, var, gamma, sum_dz, sum_dzx, dy, dz_out);
}

// r6: the same pass with the column sums of dy -- the gradient of the convolution's bias in front of this BatchNorm (model.hpp:61-78: conv has a bias) -- taken on
// the way: a third pass over dy (c6t_colstats mode 2, 0.16 ms per tower layer) is gone.  Thread mapping, chunking and the order of every sum are
// c6t_colstats4_kernel's (a thread owns four channels and every RL-th row of a 256-row chunk; row lanes added in lane order; chunks by the ordered
// finalize kernel), and dy is computed by the same expression as above: dy AND the bias gradient are bit for bit what the two kernels gave.
__global__ __launch_bounds__(256) void c6t_bn_bwd_apply4_kernel(size_t R, int C, const float *__restrict__ dout, const float *__restrict__ aout,
                                                                const float *__restrict__ y, const float *__restrict__ mean, const float *__restrict__ var,
                                                                const float *__restrict__ gamma, const float *__restrict__ sum_dz,
                                                                const float *__restrict__ sum_dzx, float *__restrict__ dy, float *__restrict__ dz_out,
                                                                float *__restrict__ partial) {
    __shared__ f32x4 red[256];
    const int CQ = C / 4, RL = 256 / CQ;
    const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ, c = cq * 4;
    const size_t r0 = (size_t)blockIdx.x * CS_ROWS;
    const size_t r1 = r0 + CS_ROWS < R ? r0 + CS_ROWS : R;
    const float invR = 1.0f / (float)R;
    const f32x4 m = *reinterpret_cast<const f32x4 *>(mean + c), vv = *reinterpret_cast<const f32x4 *>(var + c), g = *reinterpret_cast<const f32x4 *>(gamma + c),
                sdz = *reinterpret_cast<const f32x4 *>(sum_dz + c), sdzx = *reinterpret_cast<const f32x4 *>(sum_dzx + c);
    f32x4 is;
#pragma unroll
    for (int q = 0; q < 4; q++) is[q] = 1.0f / sqrtf(vv[q] + BN_EPS);
    f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (size_t r = r0 + rl; r < r1; r += RL) {
        f32x4 dz = *reinterpret_cast<const f32x4 *>(dout + r * C + c);
        const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + r * C + c);
        if (aout) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(aout + r * C + c);
#pragma unroll
            for (int q = 0; q < 4; q++) if (!(a[q] > 0.f)) dz[q] = 0.f;
        }
        f32x4 d;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float xh = (yv[q] - m[q]) * is[q];
            d[q] = g[q] * is[q] * (dz[q] - sdz[q] * invR - xh * sdzx[q] * invR);
        }
        *reinterpret_cast<f32x4 *>(dy + r * C + c) = d;
        if (dz_out) *reinterpret_cast<f32x4 *>(dz_out + r * C + c) = dz;
        s0 = s0 + d;
    }
    red[threadIdx.x] = s0;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < RL; l++) s0 = s0 + red[l * CQ + cq];
        *reinterpret_cast<f32x4 *>(partial + ((size_t)blockIdx.x * 2 + 0) * C + c) = s0;
        *reinterpret_cast<f32x4 *>(partial + ((size_t)blockIdx.x * 2 + 1) * C + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
// dy (and dz) as c6t_bn_bwd_apply, and bias_grad[c] = sum over the rows of dy[.][c]
void c6t_bn_bwd_apply_colsum(size_t R, int C, const float *dout, const float *aout, const float *y, const float *mean, const float *var, const float *gamma,
                             const float *sum_dz, const float *sum_dzx, float *dy, float *dz_out, float *partial, float *bias_grad, hipStream_t s) {
    if (C == 64 || C == 128 || C == 256) {
        const int nchunk = (int)((R + CS_ROWS - 1) / CS_ROWS);
        hipLaunchKernelGGL(c6t_bn_bwd_apply4_kernel, dim3(nchunk), dim3(256), 0, s, R, C, dout, aout, y, mean, var, gamma, sum_dz, sum_dzx, dy, dz_out, partial);
        hipLaunchKernelGGL(c6t_colstats_finalize_kernel, dim3((C + 31) / 32), dim3(32 * CSF_LANES), 0, s, 2, R, C, nchunk, partial, bias_grad, (float *)nullptr);
    } else {
        c6t_bn_bwd_apply(R, C, dout, aout, y, mean, var, gamma, sum_dz, sum_dzx, dy, dz_out, s);
        c6t_colstats(2, R, C, dy, nullptr, nullptr, nullptr, nullptr, partial, bias_grad, nullptr, s);
    }
}

// running = (1 - m) * running + m * batch   (variance: unbiased, as torch's BatchNorm), momentum 0.1
__global__ void c6t_bn_running_kernel(int C, size_t R, const float *__restrict__ mean, const float *__restrict__ var, float *__restrict__ rmean,
                                      float *__restrict__ rvar) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float unb = R > 1 ? var[c] * ((float)R / (float)(R - 1)) : var[c];
    rmean[c] = 0.9f * rmean[c] + 0.1f * mean[c];
    rvar[c] = 0.9f * rvar[c] + 0.1f * unb;
}
void c6t_bn_running(int C, size_t R, const float *mean, const float *var, float *rmean, float *rvar, hipStream_t s) {
    hipLaunchKernelGGL(c6t_bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, s, C, R, mean, var, rmean, rvar);
}

// ------------------------------------------------------------------------------------------------------------------
// heads
// ------------------------------------------------------------------------------------------------------------------
// 1x1 convolutions C -> 2 (policy) and C -> 1 (value), raw outputs: yp[B*cells][2], yv[B][cells]; one wave per pixel
__global__ __launch_bounds__(256) void c6t_head_conv_fwd_kernel(size_t R, int C, const float *__restrict__ a, const float *__restrict__ wp,
                                                                const float *__restrict__ bp, const float *__restrict__ wv,
                                                                const float *__restrict__ bv, float *__restrict__ yp, float *__restrict__ yv) {
    const int lane = threadIdx.x & 63;
    for (size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (size_t)gridDim.x * 4) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float x = a[r * C + c];
            s0 = __builtin_fmaf(x, wp[c], s0);
            s1 = __builtin_fmaf(x, wp[C + c], s1);
            s2 = __builtin_fmaf(x, wv[c], s2);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s0 += __shfl_xor(s0, off, 64);
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        if (lane == 0) {
            yp[r * 2 + 0] = s0 + bp[0];
            yp[r * 2 + 1] = s1 + bp[1];
            yv[r] = s2 + bv[0];
        }
    }
}
void c6t_head_conv_fwd(size_t R, int C, const float *a, const float *wp, const float *bp, const float *wv, const float *bv, float *yp, float *yv,
                       hipStream_t s) {
    hipLaunchKernelGGL(c6t_head_conv_fwd_kernel, dim3(4096), dim3(256), 0, s, R, C, a, wp, bp, wv, bv, yp, yv);
}

// policy BatchNorm2d(2) + ReLU + flatten: hp[b][ch*cells + p] = relu(bn(yp[(b*cells + p)][ch]))
__global__ void c6t_policy_bn_apply_kernel(size_t R, int cells, const float *__restrict__ yp, const float *__restrict__ mean,
                                           const float *__restrict__ var, const float *__restrict__ gamma, const float *__restrict__ beta,
                                           float *__restrict__ hp) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < R * 2; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / 2;
        const int ch = (int)(e % 2);
        const size_t b = r / cells, p = r % cells;
        const float v = (yp[e] - mean[ch]) * (1.0f / sqrtf(var[ch] + BN_EPS)) * gamma[ch] + beta[ch];
        hp[b * (size_t)(2 * cells) + (size_t)ch * cells + p] = v > 0.f ? v : 0.f;
    }
}
void c6t_policy_bn_apply(size_t R, int cells, const float *yp, const float *mean, const float *var, const float *gamma, const float *beta, float *hp,
                         hipStream_t s) {
    hipLaunchKernelGGL(c6t_policy_bn_apply_kernel, dim3(1024), dim3(256), 0, s, R, cells, yp, mean, var, gamma, beta, hp);
}
// the inverse re-layout for the backward pass: dzp[(b*cells + p)][ch] = dhp[b][ch*cells + p] * (hp > 0), hpm likewise (mask source)
__global__ void c6t_policy_unflatten_kernel(size_t R, int cells, const float *__restrict__ dhp, const float *__restrict__ hp, float *__restrict__ dzp) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < R * 2; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / 2;
        const int ch = (int)(e % 2);
        const size_t b = r / cells, p = r % cells;
        const size_t o = b * (size_t)(2 * cells) + (size_t)ch * cells + p;
        dzp[e] = hp[o] > 0.f ? dhp[o] : 0.f;
    }
}
void c6t_policy_unflatten(size_t R, int cells, const float *dhp, const float *hp, float *dzp, hipStream_t s) {
    hipLaunchKernelGGL(c6t_policy_unflatten_kernel, dim3(1024), dim3(256), 0, s, R, cells, dhp, hp, dzp);
}

// generic small GEMM: Cm[m][n] = act( sum_k A(m,k) * Bm(k,n) + bias[n] ), A(m,k) = A[m*sAm + k*sAk], Bm(k,n) = Bm[k*sBk + n*sBn].
// 64x64 tiles, 256 threads x (4x4), k ascending: a fixed summation order.
__global__ __launch_bounds__(256) void c6t_gemm_kernel(int M, int N, int K, const float *__restrict__ A, long sAm, long sAk,
                                                       const float *__restrict__ Bm, long sBk, long sBn, const float *__restrict__ bias,
                                                       float *__restrict__ Cm, int ldc, int relu) {
    __shared__ float As[16][64 + 1], Bs[16][64 + 1];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int kk = e / 64, i = e % 64;
            As[kk][i] = (m0 + i < M && k0 + kk < K) ? A[(long)(m0 + i) * sAm + (long)(k0 + kk) * sAk] : 0.f;
            Bs[kk][i] = (n0 + i < N && k0 + kk < K) ? Bm[(long)(k0 + kk) * sBk + (long)(n0 + i) * sBn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { av[i] = As[kk][ty * 4 + i]; bv[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < M && n < N) {
                float v = acc[i][j] + (bias ? bias[n] : 0.f);
                Cm[(size_t)m * ldc + n] = relu ? (v > 0.f ? v : 0.f) : v;
            }
        }
}
void c6t_gemm(int M, int N, int K, const float *A, long sAm, long sAk, const float *Bm, long sBk, long sBn, const float *bias, float *Cm, int ldc,
              int relu, hipStream_t s) {
    hipLaunchKernelGGL(c6t_gemm_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, s, M, N, K, A, sAm, sAk, Bm, sBk, sBn, bias, Cm, ldc, relu);
}

// row softmax of logits[B][cells] -> prob; lossp[b] = -log(prob[b][pos] + 1e-7) (model.hpp:105-106); dlogits for the
// backward pass: dL/dp[b][k] = -(1/B) * onehot / (p + 1e-7), through the softmax: dlogit = p * (dp - sum_k dp * p)
__global__ __launch_bounds__(64) void c6t_softmax_loss_kernel(int B, int cells, const float *__restrict__ logits, const float *__restrict__ poses,
                                                              float *__restrict__ prob, float *__restrict__ lossp, float *__restrict__ dlogits) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *lg = logits + (size_t)b * cells;
    float mx = -INFINITY;
    for (int k = lane; k < cells; k += 64) mx = fmaxf(mx, lg[k]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float s = 0.f;
    for (int k = lane; k < cells; k += 64) s += expf(lg[k] - mx);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    int pos = (int)poses[b];
    pos = pos < 0 ? 0 : (pos >= cells ? cells - 1 : pos);      // device-resident batches (on_device, step_replay) are not validated on the host
    const float ppos = expf(lg[pos] - mx) / s;
    const float dp = -(1.0f / (float)B) / (ppos + 1e-7f);
    const float dot = dp * ppos;
    for (int k = lane; k < cells; k += 64) {
        const float p = expf(lg[k] - mx) / s;
        prob[(size_t)b * cells + k] = p;
        dlogits[(size_t)b * cells + k] = p * ((k == pos ? dp : 0.f) - dot);
    }
    if (lane == 0) lossp[b] = -logf(ppos + 1e-7f);
}
void c6t_softmax_loss(int B, int cells, const float *logits, const float *poses, float *prob, float *lossp, float *dlogits, hipStream_t s) {
    hipLaunchKernelGGL(c6t_softmax_loss_kernel, dim3(B), dim3(64), 0, s, B, cells, logits, poses, prob, lossp, dlogits);
}

// value head tail: v = tanh(h1 . w2 + b2) per row; one wave per row
__global__ __launch_bounds__(64) void c6t_value_out_kernel(int B, int H, const float *__restrict__ h1, const float *__restrict__ w2,
                                                           const float *__restrict__ b2, float *__restrict__ v) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int k = lane; k < H; k += 64) s = __builtin_fmaf(h1[(size_t)b * H + k], w2[k], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) v[b] = tanhf(s + b2[0]);
}
void c6t_value_out(int B, int H, const float *h1, const float *w2, const float *b2, float *v, hipStream_t s) {
    hipLaunchKernelGGL(c6t_value_out_kernel, dim3(B), dim3(64), 0, s, B, H, h1, w2, b2, v);
}

// loss = mean_{i,j} (w_j - v_i)^2 + mean_i lossp_i  (the [B] - [B,1] broadcast of model.hpp:103), and
// dpre[i] = dL/dv_i * (1 - v_i^2) with dL/dv_i = (2/B) (v_i - mean w).  One workgroup, fixed order (double accumulators).
__global__ __launch_bounds__(256) void c6t_loss_kernel(int B, const float *__restrict__ v, const float *__restrict__ winners,
                                                       const float *__restrict__ lossp, float *__restrict__ loss_out, float *__restrict__ dpre) {
    __shared__ double red[4][256];
    double sv = 0, sv2 = 0, sw = 0, sw2 = 0, sp = 0;
    for (int i = threadIdx.x; i < B; i += 256) {
        const double a = v[i], w = winners[i];
        sv += a; sv2 += a * a; sw += w; sw2 += w * w; sp += lossp[i];
    }
    __shared__ double redp[256];
    red[0][threadIdx.x] = sv; red[1][threadIdx.x] = sv2; red[2][threadIdx.x] = sw; red[3][threadIdx.x] = sw2; redp[threadIdx.x] = sp;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int q = 0; q < 4; q++) red[q][threadIdx.x] += red[q][threadIdx.x + off];
            redp[threadIdx.x] += redp[threadIdx.x + off];
        }
        __syncthreads();
    }
    const double n = (double)B, mv = red[0][0] / n, mv2 = red[1][0] / n, mw = red[2][0] / n, mw2 = red[3][0] / n;
    if (threadIdx.x == 0) loss_out[0] = (float)(mv2 - 2.0 * mv * mw + mw2 + redp[0] / n);
    if (dpre)
        for (int i = threadIdx.x; i < B; i += 256) {
            const float a = v[i];
            dpre[i] = (float)(2.0 / n * ((double)a - mw)) * (1.f - a * a);
        }
}
void c6t_loss(int B, const float *v, const float *winners, const float *lossp, float *loss_out, float *dpre, hipStream_t s) {
    hipLaunchKernelGGL(c6t_loss_kernel, dim3(1), dim3(256), 0, s, B, v, winners, lossp, loss_out, dpre);
}

// dh1[b][k] = dpre[b] * w2[k] * (h1 > 0)
__global__ void c6t_value_bwd_h1_kernel(int B, int H, const float *__restrict__ dpre, const float *__restrict__ w2, const float *__restrict__ h1,
                                        float *__restrict__ dh1) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)B * H; e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / H), k = (int)(e % H);
        dh1[e] = h1[e] > 0.f ? dpre[b] * w2[k] : 0.f;
    }
}
void c6t_value_bwd_h1(int B, int H, const float *dpre, const float *w2, const float *h1, float *dh1, hipStream_t s) {
    hipLaunchKernelGGL(c6t_value_bwd_h1_kernel, dim3(256), dim3(256), 0, s, B, H, dpre, w2, h1, dh1);
}

// gradient of the 1x1 head convolutions: partial[chunk][3][C] = sum over the chunk's rows of {dyp0, dyp1, dyv}[r] * a[r][c]
__global__ __launch_bounds__(256) void c6t_head_conv_bwd_w_kernel(size_t R, int C, const float *__restrict__ a, const float *__restrict__ dyp,
                                                                  const float *__restrict__ dyv, float *__restrict__ partial) {
    const size_t r0 = (size_t)blockIdx.x * CS_ROWS;
    const size_t r1 = r0 + CS_ROWS < R ? r0 + CS_ROWS : R;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (size_t r = r0; r < r1; r++) {
            const float x = a[r * C + c];
            s0 = __builtin_fmaf(dyp[r * 2 + 0], x, s0);
            s1 = __builtin_fmaf(dyp[r * 2 + 1], x, s1);
            s2 = __builtin_fmaf(dyv[r], x, s2);
        }
        partial[((size_t)blockIdx.x * 3 + 0) * C + c] = s0;
        partial[((size_t)blockIdx.x * 3 + 1) * C + c] = s1;
        partial[((size_t)blockIdx.x * 3 + 2) * C + c] = s2;
    }
}
__global__ __launch_bounds__(256) void c6t_head_conv_bwd_w_finalize_kernel(int C, int nchunk, const float *__restrict__ partial,
                                                                           float *__restrict__ dwp, float *__restrict__ dwv) {
    __shared__ double l0[8][32];
    const int cx = threadIdx.x & 31, ky = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + cx;        // (q, c) flattened
    const int q = idx / C, c = idx % C;
    double s = 0.0;
    if (idx < 3 * C)
        for (int k = ky; k < nchunk; k += 8) s += (double)partial[((size_t)k * 3 + q) * C + c];
    l0[ky][cx] = s;
    __syncthreads();
    if (ky != 0 || idx >= 3 * C) return;
    s = 0.0;
    for (int j = 0; j < 8; j++) s += l0[j][cx];
    if (q < 2) dwp[q * C + c] = (float)s; else dwv[c] = (float)s;
}
// da[r][c] = dyp[r][0] wp[0][c] + dyp[r][1] wp[1][c] + dyv[r] wv[c]
__global__ void c6t_head_conv_bwd_x_kernel(size_t R, int C, const float *__restrict__ dyp, const float *__restrict__ dyv, const float *__restrict__ wp,
                                           const float *__restrict__ wv, float *__restrict__ da) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < R * C; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / C;
        const int c = (int)(e % C);
        da[e] = __builtin_fmaf(dyp[r * 2 + 0], wp[c], __builtin_fmaf(dyp[r * 2 + 1], wp[C + c], dyv[r] * wv[c]));
    }
}
size_t c6t_head_conv_bwd_partial_floats(size_t R, int C) { return ((R + CS_ROWS - 1) / CS_ROWS) * 3 * (size_t)C; }
void c6t_head_conv_bwd(size_t R, int C, const float *a, const float *dyp, const float *dyv, const float *wp, const float *wv, float *partial,
                       float *dwp, float *dwv, float *da, hipStream_t s) {
    const int nchunk = (int)((R + CS_ROWS - 1) / CS_ROWS);
    hipLaunchKernelGGL(c6t_head_conv_bwd_w_kernel, dim3(nchunk), dim3(256), 0, s, R, C, a, dyp, dyv, partial);
    hipLaunchKernelGGL(c6t_head_conv_bwd_w_finalize_kernel, dim3((3 * C + 31) / 32), dim3(256), 0, s, C, nchunk, partial, dwp, dwv);
    hipLaunchKernelGGL(c6t_head_conv_bwd_x_kernel, dim3(2048), dim3(256), 0, s, R, C, dyp, dyv, wp, wv, da);
}

// ------------------------------------------------------------------------------------------------------------------
// SGD with momentum (torch::optim::SGD, dampening 0, no weight decay): buf = m * buf + g; p -= lr * buf; only where
// mask != 0 (BatchNorm running statistics live in the same blob and are not parameters)
// ------------------------------------------------------------------------------------------------------------------
__global__ void c6t_sgd_kernel(size_t n, float lr, float momentum, int first, const unsigned char *__restrict__ mask, const float *__restrict__ g,
                               float *__restrict__ buf, float *__restrict__ p) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        if (!mask[e]) continue;
        const float b = first ? g[e] : momentum * buf[e] + g[e];   // torch: the buffer is initialised with the first gradient
        buf[e] = b;
        p[e] = p[e] - lr * b;
    }
}
void c6t_sgd(size_t n, float lr, float momentum, int first, const unsigned char *mask, const float *g, float *buf, float *p, hipStream_t s) {
    hipLaunchKernelGGL(c6t_sgd_kernel, dim3(1024), dim3(256), 0, s, n, lr, momentum, first, mask, g, buf, p);
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient of a 3x3 convolution: dW[co][ci][tap] = sum_{b,p} dy[b][p][co] * x[b][p + off(tap)][ci], exact FP32 on
// v_mfma_f32_32x32x2_f32.
//  * workgroup (4 waves, 2 x 2) = a 64 (co) x 64 (ci) block of dW for ALL 9 taps and one K split (a range of boards);
//    a wave = 32 x 32 x 9 taps = 9 accumulator tiles (144 VGPRs);
//  * K runs over the pixels of a row band of one board on the zero-haloed (N+2)-wide grid: both operands are then
//    linear in k -- A = dy[k][co] (zero in the halo columns, so the padding positions add nothing), B_tap = x[k + off];
//    a 32x32x2 MFMA eats two consecutive k; lane (i, kb) reads element [k + kb][base + i]: 32 consecutive floats;
//  * per K step one A read and nine B reads from LDS feed nine MFMAs;
//  * two workgroups per CU (LDS <= 80 KB each): staging is synchronous inside a workgroup, the partner's MFMAs cover it;
//  * partial[split][tap][co][ci]; c6t_wgrad_reduce_kernel sums the splits in order into OIHW.
// ------------------------------------------------------------------------------------------------------------------
template <int N>
struct WGradGeom {
    static constexpr int P = N + 2;
    static constexpr int RB = N <= 9 ? N : (N <= 15 ? 8 : 5);      // output rows per band: LDS <= 80 KB, two workgroups per CU
                                                                    // (one stages its next band while the other runs MFMAs)
    static constexpr int BANDS = (N + RB - 1) / RB;
    static constexpr int KPAD = ((RB * P + 1) / 2) * 2;             // k positions per band (even)
    static constexpr int DY_ROWS = KPAD + 2;
    static constexpr int X_ROWS = (RB + 2) * P + 4;
    static constexpr size_t LDS_BYTES = (size_t)(DY_ROWS + X_ROWS) * 64 * sizeof(float);
};

template <int N>
__global__ __launch_bounds__(256, 2) void c6_wgrad_kernel(int C, int B, int boards_per_split, const float *__restrict__ dy, const float *__restrict__ x,
                                                       float *__restrict__ partial) {
    using G = WGradGeom<N>;
    constexpr int P = G::P, RB = G::RB;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *dyL = wsm;                       // [DY_ROWS][64]: band rows on the P-wide grid, halo columns zero
    float *xL = wsm + G::DY_ROWS * 64;      // [X_ROWS][64]: band rows -1 .. RB on the P-wide grid, zero halo

    const int nblk = C / 64;
    const int co0 = (blockIdx.x / nblk) * 64, ci0 = (blockIdx.x % nblk) * 64, split = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
    const int i = lane & 31, kb = lane >> 5;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[t][v] = 0.f;

    const int b_lo = split * boards_per_split, b_hi = (b_lo + boards_per_split < B) ? b_lo + boards_per_split : B;
    // the halo cells are written once: the fills below only touch interior cells (and, with several bands per board, the
    // rows a shorter last band leaves over are cleared again)
    for (int e = tid; e < (G::DY_ROWS + G::X_ROWS) * 16; e += 256) reinterpret_cast<f32x4 *>(wsm)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b = b_lo; b < b_hi; b++) {
        for (int band = 0; band < G::BANDS; band++) {
            const int y0 = band * RB, rows = (y0 + RB <= N) ? RB : N - y0;
            __syncthreads();
            if (G::BANDS > 1 && rows < RB) {      // stale rows of the previous (full) band
                for (int e = tid; e < (RB - rows) * P * 16; e += 256) {
                    reinterpret_cast<f32x4 *>(dyL + rows * P * 64)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                    reinterpret_cast<f32x4 *>(xL + (rows + 2) * P * 64)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            // staging in batches of 8 loads in flight per thread (one workgroup per CU: nothing else hides the latency)
            {
                const int cnt = rows * N * 16;
                for (int base = 0; base < cnt; base += 8 * 256) {
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        int e = base + u * 256 + tid;
                        e = e < cnt ? e : cnt - 1;
                        const int pix = e / 16, q = e % 16;
                        v[u] = *reinterpret_cast<const f32x4 *>(dy + ((size_t)b * N * N + (size_t)y0 * N + pix) * C + co0 + q * 4);
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int e = base + u * 256 + tid;
                        if (e < cnt) {
                            const int pix = e / 16, q = e % 16, yy = pix / N, xx = pix % N;
                            *reinterpret_cast<f32x4 *>(dyL + (yy * P + xx + 1) * 64 + q * 4) = v[u];     // k = yy*P + (xx+1)
                        }
                    }
                }
            }
            {
                const int cnt = (rows + 2) * N * 16;
                for (int base = 0; base < cnt; base += 8 * 256) {
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        int e = base + u * 256 + tid;
                        e = e < cnt ? e : cnt - 1;
                        const int pix = e / 16, q = e % 16, yy = pix / N, xx = pix % N, gy = y0 - 1 + yy;
                        const int cy = gy < 0 ? 0 : (gy >= N ? N - 1 : gy);                    // clamped: unconditional load
                        v[u] = *reinterpret_cast<const f32x4 *>(x + ((size_t)b * N * N + (size_t)cy * N + xx) * C + ci0 + q * 4);
                        if (gy < 0 || gy >= N) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};               // rows outside the board: zero
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int e = base + u * 256 + tid;
                        if (e < cnt) {
                            const int pix = e / 16, q = e % 16, yy = pix / N, xx = pix % N;
                            *reinterpret_cast<f32x4 *>(xL + (yy * P + xx + 1) * 64 + q * 4) = v[u];
                        }
                    }
                }
            }
            __syncthreads();
            // k position kk of dyL is grid cell (row kk / P of the band, column kk % P); the matching x cell for tap
            // (ty, tx) is xL row (kk / P + ty) , column (kk % P + tx - 1)  ->  xL index kk + ty * P + tx - 1
            const float *ap = dyL + kb * 64 + wco + i;
            const float *bp = xL + kb * 64 + wci + i;
            // K steps of two grid cells; a row's steps start at its first real cell (column 1), so only the right halo
            // cell (N odd: paired with the last pixel) is multiplied in vain -- not both halo columns of every row
            constexpr int SPR = (N + 1) / 2;                 // steps per row
            const int ksteps = rows * SPR;
            auto kpos = [&](int ks) -> int { return (ks / SPR) * P + 1 + 2 * (ks % SPR); };
            // operands of step ks + 1 are read before the MFMAs of step ks (the partner workgroup's waves aside, nothing else hides the LDS latency)
            float av = ap[kpos(0) * 64], bv[9];
#pragma unroll
            for (int t = 0; t < 9; t++) bv[t] = bp[(kpos(0) + (t / 3) * P + (t % 3) - 1) * 64];
#pragma unroll 1
            for (int ks = 0; ks < ksteps; ks++) {
                const int kn = kpos(ks + 1 < ksteps ? ks + 1 : ks);
                const float an = ap[kn * 64];
                float bn[9];
#pragma unroll
                for (int t = 0; t < 9; t++) bn[t] = bp[(kn + (t / 3) * P + (t % 3) - 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 9; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                av = an;
#pragma unroll
                for (int t = 0; t < 9; t++) bv[t] = bn[t];
            }
        }
    }
    // D tile: lane (i = column = ci, h = kb), register v -> row (co) = (v & 3) + 8 (v >> 2) + 4 h
    float *pp = partial + (size_t)split * 9 * C * C;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int co = co0 + wco + (v & 3) + 8 * (v >> 2) + 4 * kb, ci = ci0 + wci + i;
            pp[((size_t)t * C + co) * C + ci] = acc[t][v];
        }
}

// dW[co][ci][tap] (OIHW) = sum over the splits, in split order
__global__ void c6t_wgrad_reduce_kernel(int C, int nsplit, const float *__restrict__ partial, float *__restrict__ dw) {
    const size_t total = (size_t)9 * C * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; k++) s += partial[(size_t)k * total + e];
        const int t = (int)(e / ((size_t)C * C)), co = (int)((e / C) % C), ci = (int)(e % C);
        dw[((size_t)co * C + ci) * 9 + t] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix cores with split operands (dy = hi + mid, x = hi + mid,
// dy.x = hi.hi + hi.mid + mid.hi, FP32 accumulate): c6_wgrad_b3_kernel, v_mfma_f32_16x16x32_bf16.
//  * K = grid cells (pixels), 32 per MFMA: both operands must be K-CONTIGUOUS per lane although memory (NHWC) and the LDS
//    image are channel-contiguous -> the fragments are fetched with ds_read_b64_tr_b16 (a 4-row x 16-column block of
//    16-bit elements per 16-lane group, delivered column-major: lane i gets channel i of four consecutive cells).
//  * LDS image: one 256-byte row per grid cell = 16 chunks of 16 bytes: chunks 0-7 the hi plane of the block's 64
//    channels, 8-15 the mid plane; chunk ch of row r lives at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))) -- the
//    guide's image (b), conflict-free for the transposed reads at ANY row offset (the nine taps shift the x rows).
//    The swizzle depends on r mod 16 only and K steps advance by 32 rows: per lane two base addresses for dy and two per
//    tap for x; tile (+16 channels) and plane (+64 channels) are XORs of the address with 32 and 128.
//  * workgroup = 64 x 64 block of dW x 9 taps x one K split as in the FP32 kernel; wave = 32 x 32 = 2 x 2 tiles of
//    16 x 16 x 9 taps = 36 accumulator tiles (144 VGPRs); row bands sized for two workgroups per CU.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

template <int N>
struct WGradB3Geom {
    static constexpr int P = N + 2;
    static constexpr int RB = N <= 9 ? N : (N <= 15 ? 5 : 4);         // output rows per band
    static constexpr int BANDS = (N + RB - 1) / RB;
    static constexpr int KSTEPS = (RB * P + 31) / 32;
    static constexpr int DY_ROWS = KSTEPS * 32;
    static constexpr int X_ROWS = ((DY_ROWS + 2 * P + 3) + 3) & ~3;     // x image row j = grid cell j - 1 of the haloed band
    static constexpr size_t LDS_BYTES = (size_t)(DY_ROWS + X_ROWS) * 256;
};

__device__ __forceinline__ int wg_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <int N>
__global__ __launch_bounds__(256, 2) void c6_wgrad_b3_kernel(int C, int B, int boards_per_split, const float *__restrict__ dy,
                                                             const float *__restrict__ x, float *__restrict__ partial) {
    using G = WGradB3Geom<N>;
    constexpr int P = G::P, RB = G::RB;
    extern __shared__ __attribute__((aligned(16))) unsigned char wb[];
    unsigned char *dyL = wb, *xL = wb + (size_t)G::DY_ROWS * 256;

    const int nblk = C / 64;
    const int co0 = (blockIdx.x / nblk) * 64, ci0 = (blockIdx.x % nblk) * 64, split = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
    const int g = lane >> 4, i = lane & 15, q = (lane >> 2) & 3, p = lane & 3;

    f32x4 acc[2][2][9];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int tp = 0; tp < 9; tp++) acc[t][u][tp] = f32x4{0.f, 0.f, 0.f, 0.f};

    // lane bases of the transposed reads: lane 4q+p of group g supplies row (8g + q [+4]) , chunk c0 + (p >> 1), half p & 1
    int baseA[2], baseB[9][2];
#pragma unroll
    for (int ex = 0; ex < 2; ex++) {
        const int row = 8 * g + q + 4 * ex;
        baseA[ex] = wg_off(row, (wco >> 3) + (p >> 1)) + 8 * (p & 1);
#pragma unroll
        for (int tp = 0; tp < 9; tp++) baseB[tp][ex] = wg_off(row + (tp / 3) * P + (tp % 3), (wci >> 3) + (p >> 1)) + 8 * (p & 1);
    }

    for (int e = tid; e < (int)(G::LDS_BYTES / 16); e += 256) reinterpret_cast<f32x4 *>(wb)[e] = f32x4{0.f, 0.f, 0.f, 0.f};   // halo cells: once

    auto put = [&](unsigned char *img, int row, int qq, f32x4 v) {      // 4 channels (qq-th quad of the block's 64) of one cell
        bf16x4_t hi, mid;
#pragma unroll
        for (int r = 0; r < 4; r++) { hi[r] = (__bf16)v[r]; mid[r] = (__bf16)(v[r] - (float)hi[r]); }
        *reinterpret_cast<bf16x4_t *>(img + wg_off(row, qq >> 1) + 8 * (qq & 1)) = hi;
        *reinterpret_cast<bf16x4_t *>(img + wg_off(row, 8 + (qq >> 1)) + 8 * (qq & 1)) = mid;
    };

    const int b_lo = split * boards_per_split, b_hi = (b_lo + boards_per_split < B) ? b_lo + boards_per_split : B;
    for (int b = b_lo; b < b_hi; b++) {
        for (int band = 0; band < G::BANDS; band++) {
            const int y0 = band * RB, rows = (y0 + RB <= N) ? RB : N - y0;
            __syncthreads();
            if (G::BANDS > 1 && rows < RB) {      // rows a shorter last band leaves over from the previous full band
                for (int e = tid; e < (RB - rows) * P * 16; e += 256) {
                    reinterpret_cast<f32x4 *>(dyL + (size_t)rows * P * 256)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                    reinterpret_cast<f32x4 *>(xL + (size_t)((rows + 2) * P + 1) * 256)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                __syncthreads();
            }
            {   // dy rows of the band: cell (yy, xx) -> image row yy*P + xx + 1;  x rows y0-1 .. y0+rows: cell (yy, xx) ->
                // image row yy*P + xx + 2 (grid cell + 1).  All loads of the band are issued before the first conversion
                // (one HBM round trip per band instead of two or more: the kernel waits on exactly this)
                constexpr int DYI = (RB * N * 16 + 255) / 256, XI = ((RB + 2) * N * 16 + 255) / 256;
                const int cnt_d = rows * N * 16, cnt_x = (rows + 2) * N * 16;
                f32x4 vd[DYI], vx[XI];
#pragma unroll
                for (int u = 0; u < DYI; u++) {
                    int e = u * 256 + tid;
                    e = e < cnt_d ? e : cnt_d - 1;
                    vd[u] = *reinterpret_cast<const f32x4 *>(dy + ((size_t)b * N * N + (size_t)y0 * N + e / 16) * C + co0 + (e % 16) * 4);
                }
#pragma unroll
                for (int u = 0; u < XI; u++) {
                    int e = u * 256 + tid;
                    e = e < cnt_x ? e : cnt_x - 1;
                    const int pix = e / 16, yy = pix / N, xx = pix % N, gy = y0 - 1 + yy;
                    const int cy = gy < 0 ? 0 : (gy >= N ? N - 1 : gy);
                    vx[u] = *reinterpret_cast<const f32x4 *>(x + ((size_t)b * N * N + (size_t)cy * N + xx) * C + ci0 + (e % 16) * 4);
                    if (gy < 0 || gy >= N) vx[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < DYI; u++) {
                    const int e = u * 256 + tid;
                    if (e < cnt_d) { const int pix = e / 16; put(dyL, (pix / N) * P + pix % N + 1, e % 16, vd[u]); }
                }
#pragma unroll
                for (int u = 0; u < XI; u++) {
                    const int e = u * 256 + tid;
                    if (e < cnt_x) { const int pix = e / 16; put(xL, (pix / N) * P + pix % N + 2, e % 16, vx[u]); }
                }
            }
            __syncthreads();
            const int ksteps = (rows * P + 31) / 32;
            auto frag = [&](const unsigned char *img, int base0, int base1, int kofs, int x_or) -> bf16x8_t {
                // opaque: otherwise LICM hoists the 40 (base ^ constant) addresses out of the K loop -> spills
                asm volatile("" : "+v"(base0), "+v"(base1));
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(img + ((base0 ^ x_or) + kofs)));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(img + ((base1 ^ x_or) + kofs)));
                union { s16x4_t s[2]; bf16x8_t v; } uu;
                uu.s[0] = lo; uu.s[1] = hi;
                return uu.v;
            };
#pragma unroll 1
            for (int ks = 0; ks < ksteps; ks++) {
                const int kofs = ks * 32 * 256;
                bf16x8_t ah[2], am[2];
#pragma unroll
                for (int t = 0; t < 2; t++) { ah[t] = frag(dyL, baseA[0], baseA[1], kofs, 32 * t); am[t] = frag(dyL, baseA[0], baseA[1], kofs, 32 * t + 128); }
                // x fragments of tap tp + 1 are read before the MFMAs of tap tp (double-buffered by hand; sched_barrier keeps
                // hipcc from hoisting all nine taps' reads to the top: 72 fragment registers -> spills)
                bf16x8_t bh[2][2], bm[2][2];
                auto load_b = [&](int slot, int tp) {
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        bh[slot][u] = frag(xL, baseB[tp][0], baseB[tp][1], kofs, 32 * u);
                        bm[slot][u] = frag(xL, baseB[tp][0], baseB[tp][1], kofs, 32 * u + 128);
                    }
                };
                load_b(0, 0);
#pragma unroll
                for (int tp = 0, sl = 0; tp < 9; tp++, sl ^= 1) {
                    if (tp + 1 < 9) load_b(sl ^ 1, tp + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int t = 0; t < 2; t++) {
                            acc[t][u][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[t], bh[sl][u], acc[t][u][tp], 0, 0, 0);
                            acc[t][u][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bm[sl][u], acc[t][u][tp], 0, 0, 0);
                            acc[t][u][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh[sl][u], acc[t][u][tp], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // D tile (t, u): lane (column i = ci, g) holds rows co = 4g + r
    float *pp = partial + (size_t)split * 9 * C * C;
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int co = co0 + wco + 16 * t + 4 * g + r, ci = ci0 + wci + 16 * u + i;
                    pp[((size_t)tp * C + co) * C + ci] = acc[t][u][tp][r];
                }
}

// Wave-specialised weight gradient (the trainer's default): the arithmetic of c6_wgrad_b3_kernel (same split, same three
// products per accumulator in the same order), restructured around what the kernel above waits for.
//  * Roles.  Above, a workgroup stages a band (global -> split -> LDS) and then multiplies it, two workgroups per CU taking
//    turns; a staging phase is an HBM round trip plus ~50 VALU/LDS instructions per thread and the matrix pipes idled 63 %.
//    Here a workgroup is 8 waves: waves 0-3 own the 64 x 64 x 9-tap accumulators and only read fragments and issue MFMAs;
//    waves 4-7 (no accumulators: two whole bands fit in their registers) load band i + 2, split and store band i + 1 into the
//    other half of a double-buffered image while band i is multiplied.  One barrier per band, reached by both roles.
//  * Fragment addressing.  A lone compute wave per SIMD has nobody to hide its address arithmetic behind, and the XOR-swizzled
//    image above costs 2-3 VALU instructions per transposed read.  This image is LINEAR with a 288-byte row (256 + 32 pad):
//    tile, plane, tap and read-half are immediate offsets of ONE base register per operand.  Conflict-free because the K
//    slots of an MFMA are permuted (the same way for dy and x, so every product still meets its partner): lane group g reads
//    rows 4g + q and 16 + 4g + q of a 32-row K step instead of 8g + q and 8g + 4 + q, so the 32 lanes LDS serves together
//    touch 8 consecutive rows = 8 distinct 32-byte bank groups, at any tap shift.  (The K permutation reorders the f32 sums
//    inside an MFMA: last-bit differences from the kernel above, still bit-reproducible run to run.)
//  * Issue order.  Every MFMA is followed by at most one ds_read_tr of the fragments two taps ahead (three B slots), which
//    issues in that MFMA's shadow; consecutive MFMAs never share an accumulator; the K loop is unrolled by two so that every
//    register index is static.
//  * XCD-aware 1-D grid: all blocks of one K split (they read the same bands, each band C/64 times) get workgroup ids of one
//    residue mod 8 = one XCD = one L2.
template <int N>
struct WGradWSGeom {
    static constexpr int P = N + 2;
    static constexpr int RS = 288;                                     // image row stride in bytes
    static constexpr int RB = N <= 9 ? 5 : (N <= 15 ? 5 : 4);          // output rows per band (two images of a band must fit 160 KB)
    static constexpr int BANDS = (N + RB - 1) / RB;
    static constexpr int KSTEPS = (RB * P + 31) / 32;
    static constexpr int DY_ROWS = KSTEPS * 32;
    static constexpr int X_ROWS = ((DY_ROWS + 2 * P + 3) + 3) & ~3;     // x image row j = grid cell j - 1 of the haloed band
    static constexpr size_t IMG_BYTES = (size_t)(DY_ROWS + X_ROWS) * RS;
    static constexpr size_t LDS_BYTES = 2 * IMG_BYTES;
};

template <int N>
__global__ __launch_bounds__(512, 1) void c6_wgrad_b3w_kernel(int C, int B, int boards_per_split, const float *__restrict__ dy,
                                                              const float *__restrict__ x, float *__restrict__ partial) {
    using G = WGradWSGeom<N>;
    constexpr int P = G::P, RB = G::RB, RS = G::RS;
    extern __shared__ __attribute__((aligned(16))) unsigned char wb[];

    const int nblk = C / 64, nb2 = nblk * nblk, nsplit = gridDim.x / nb2;
    int blk, split;
    if (nsplit % 8 == 0) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; split = xcd + 8 * (j / nb2); blk = j % nb2; }
    else { blk = blockIdx.x % nb2; split = blockIdx.x / nb2; }
    const int co0 = (blk / nblk) * 64, ci0 = (blk % nblk) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;
    const int ltid = tid & 255;

    for (int e = tid; e < (int)(G::LDS_BYTES / 16); e += 512) reinterpret_cast<f32x4 *>(wb)[e] = f32x4{0.f, 0.f, 0.f, 0.f};   // halo cells: once

    const int b_lo = split * boards_per_split, b_hi = (b_lo + boards_per_split < B) ? b_lo + boards_per_split : B;
    const int nitems = b_hi > b_lo ? (b_hi - b_lo) * G::BANDS : 0;
    __syncthreads();

    if (loader) {
        auto put = [&](unsigned char *img, int row, int qq, f32x4 v) {      // 4 channels (qq-th quad of the block's 64) of one cell
            bf16x4_t hi, mid;
#pragma unroll
            for (int r = 0; r < 4; r++) { hi[r] = (__bf16)v[r]; mid[r] = (__bf16)(v[r] - (float)hi[r]); }
            *reinterpret_cast<bf16x4_t *>(img + row * RS + qq * 8) = hi;
            *reinterpret_cast<bf16x4_t *>(img + row * RS + 128 + qq * 8) = mid;
        };
        constexpr int DYI = (RB * N * 16 + 255) / 256, XI = ((RB + 2) * N * 16 + 255) / 256;
        struct Band { f32x4 vd[DYI], vx[XI]; };
        // the loads of band `it` (all in flight at once) ...
        auto issue = [&](int it, Band &r) {
            const int b = b_lo + it / G::BANDS, band = it % G::BANDS;
            const int y0 = band * RB, rows = (y0 + RB <= N) ? RB : N - y0;
            const int cnt_d = rows * N * 16, cnt_x = (rows + 2) * N * 16;
#pragma unroll
            for (int u = 0; u < DYI; u++) {
                int e = u * 256 + ltid;
                e = e < cnt_d ? e : cnt_d - 1;
                r.vd[u] = *reinterpret_cast<const f32x4 *>(dy + ((size_t)b * N * N + (size_t)y0 * N + e / 16) * C + co0 + (e % 16) * 4);
            }
#pragma unroll
            for (int u = 0; u < XI; u++) {
                int e = u * 256 + ltid;
                e = e < cnt_x ? e : cnt_x - 1;
                const int pix = e / 16, yy = pix / N, xx = pix % N, gy = y0 - 1 + yy;
                const int cy = gy < 0 ? 0 : (gy >= N ? N - 1 : gy);
                r.vx[u] = *reinterpret_cast<const f32x4 *>(x + ((size_t)b * N * N + (size_t)cy * N + xx) * C + ci0 + (e % 16) * 4);
            }
        };
        // ... and their split + store into the half of the image that band uses: dy cell (yy, xx) -> image row yy*P + xx + 1,
        // x rows y0-1 .. y0+rows: cell (yy, xx) -> image row yy*P + xx + 2 (grid cell + 1)
        auto commit = [&](int it, const Band &r) {
            unsigned char *dyL = wb + (size_t)(it & 1) * G::IMG_BYTES, *xL = dyL + (size_t)G::DY_ROWS * RS;
            const int band = it % G::BANDS;
            const int y0 = band * RB, rows = (y0 + RB <= N) ? RB : N - y0;
            const int cnt_d = rows * N * 16, cnt_x = (rows + 2) * N * 16;
            if (G::BANDS > 1 && rows < RB) {      // rows a shorter last band leaves over from a full band (disjoint from what is written below)
                for (int e = ltid; e < (RB - rows) * P * (RS / 16); e += 256) {
                    reinterpret_cast<f32x4 *>(dyL + (size_t)rows * P * RS)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                    reinterpret_cast<f32x4 *>(xL + (size_t)((rows + 2) * P + 1) * RS)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < DYI; u++) {
                const int e = u * 256 + ltid;
                if (e < cnt_d) { const int pix = e / 16; put(dyL, (pix / N) * P + pix % N + 1, e % 16, r.vd[u]); }
            }
#pragma unroll
            for (int u = 0; u < XI; u++) {
                const int e = u * 256 + ltid;
                if (e < cnt_x) {
                    const int pix = e / 16, gy = y0 - 1 + pix / N;
                    put(xL, (pix / N) * P + pix % N + 2, e % 16, (gy < 0 || gy >= N) ? f32x4{0.f, 0.f, 0.f, 0.f} : r.vx[u]);   // rows off the board: zero
                }
            }
        };
        Band ra, rb;
        if (nitems > 0) { issue(0, ra); if (nitems > 1) issue(1, rb); commit(0, ra); }
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < nitems; it += 2) {
            if (it + 2 < nitems) issue(it + 2, ra);
            if (it + 1 < nitems) commit(it + 1, rb);
            __syncthreads();
            if (it + 1 < nitems) {
                if (it + 3 < nitems) issue(it + 3, rb);
                if (it + 2 < nitems) commit(it + 2, ra);
                __syncthreads();
            }
        }
        return;
    }

    const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
    const int g = lane >> 4, i = lane & 15, q = (lane >> 2) & 3, p = lane & 3;
    f32x4 acc[2][2][9];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int tp = 0; tp < 9; tp++) acc[t][u][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane 4q + p of group g supplies 8 bytes (4 channels) of row 4g + q (second half of the fragment: + 16 rows)
    const int laneA = (4 * g + q) * RS + 2 * wco + 8 * p, laneB = (4 * g + q) * RS + 2 * wci + 8 * p;
    s16x4_t fa[2][2][2][2];   // [set = K step parity][t][plane hi/mid][half]
    s16x4_t fb[3][2][2][2];   // [slot = tap % 3][u][plane][half]
    auto cat = [](s16x4_t lo, s16x4_t hi) -> bf16x8_t {
        union { s16x4_t s[2]; bf16x8_t v; } uu;
        uu.s[0] = lo; uu.s[1] = hi;
        return uu.v;
    };
    auto rd = [&](const unsigned char *img, int base, int off) -> s16x4_t {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(img + base + off));
    };
    __syncthreads();   // item 0 staged
#pragma unroll 1
    for (int it = 0; it < nitems; it++) {
        const unsigned char *dyL = wb + (size_t)(it & 1) * G::IMG_BYTES, *xL = dyL + (size_t)G::DY_ROWS * RS;
        const int band = it % G::BANDS, y0 = band * RB, rows = (y0 + RB <= N) ? RB : N - y0;
        const int ksteps = (rows * P + 31) / 32;
        // read number r = 0..7 of a fragment set: (t or u) = r >> 2, plane = (r >> 1) & 1, half = r & 1
        auto rdA = [&](auto SET, int r, int base) {
            constexpr int set = decltype(SET)::value;
            fa[set][r >> 2][(r >> 1) & 1][r & 1] = rd(dyL, base, (r & 1) * 16 * RS + 32 * (r >> 2) + 128 * ((r >> 1) & 1));
        };
        auto rdB = [&](auto SLOT, int tp, int r, int base) {
            constexpr int slot = decltype(SLOT)::value;
            fb[slot][r >> 2][(r >> 1) & 1][r & 1] = rd(xL, base, ((tp / 3) * P + tp % 3 + (r & 1) * 16) * RS + 32 * (r >> 2) + 128 * ((r >> 1) & 1));
        };
        auto kstep = [&](auto PAR, int ks) {
            constexpr int par = decltype(PAR)::value;
            const int bA = laneA + ks * 32 * RS, bB = laneB + ks * 32 * RS;
            const bool nxt = ks + 1 < ksteps;
#pragma unroll
            for (int tp = 0; tp < 9; tp++) {
                const int sl = tp % 3;                                      // 9 taps: the slot of a tap does not depend on the K step
                const int ntp = (tp + 2) % 9, nb = bB + (tp >= 7 ? 32 * RS : 0);   // the tap two ahead (of the next K step from tap 7 on)
                const bool pre = tp < 7 || nxt;
#pragma unroll
                for (int m = 0; m < 12; m++) {
                    // product-major over the four accumulators of the tap; each accumulator still sees k = 0, 1, 2 in order
                    const int k = m / 4, u = (m >> 1) & 1, t = m & 1;
                    const bf16x8_t a = cat(fa[par][t][k == 0][0], fa[par][t][k == 0][1]);       // k 0: mid plane of dy, k 1, 2: hi
                    const bf16x8_t bq = cat(fb[sl][u][k == 1][0], fb[sl][u][k == 1][1]);        // k 1: mid plane of x
                    acc[t][u][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, acc[t][u][tp], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m < 8) {
                        if (pre) {
                            if (ntp % 3 == 0) rdB(std::integral_constant<int, 0>{}, ntp, m, nb);
                            else if (ntp % 3 == 1) rdB(std::integral_constant<int, 1>{}, ntp, m, nb);
                            else rdB(std::integral_constant<int, 2>{}, ntp, m, nb);
                        }
                    } else if (nxt && (tp == 5 || tp == 6)) rdA(std::integral_constant<int, par ^ 1>{}, (tp - 5) * 4 + (m - 8), bA + 32 * RS);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
#pragma unroll
        for (int r = 0; r < 8; r++) { rdA(std::integral_constant<int, 0>{}, r, laneA); rdB(std::integral_constant<int, 0>{}, 0, r, laneB); rdB(std::integral_constant<int, 1>{}, 1, r, laneB); }
#pragma unroll 1
        for (int ks = 0; ks < ksteps; ks += 2) {
            kstep(std::integral_constant<int, 0>{}, ks);
            if (ks + 1 < ksteps) kstep(std::integral_constant<int, 1>{}, ks + 1);
        }
        __syncthreads();   // band it multiplied, band it + 1 staged
    }
    // D tile (t, u): lane (column i = ci, g) holds rows co = 4g + r
    float *pp = partial + (size_t)split * 9 * C * C;
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int co = co0 + wco + 16 * t + 4 * g + r, ci = ci0 + wci + 16 * u + i;
                    pp[((size_t)tp * C + co) * C + ci] = acc[t][u][tp][r];
                }
}

int c6t_wgrad_splits(int C, int B) {
    const int blocks = (C / 64) * (C / 64);
    int splits = (512 + blocks - 1) / blocks;      // about two workgroups per CU
    if (splits > B) splits = B;
    if (splits < 1) splits = 1;
    return splits;
}
size_t c6t_wgrad_partial_floats(int C, int B) { return (size_t)c6t_wgrad_splits(C, B) * 9 * C * C; }

template <int N>
static void launch_wgrad(int C, int B, const float *dy, const float *x, float *partial, float *dw, int split_bf16, hipStream_t s) {
    const int splits = c6t_wgrad_splits(C, B), per = (B + splits - 1) / splits;
    constexpr int ws = C6_AB_WGRAD_WS;      // 0 (A/B builds only, tools/ab_build.sh -DC6_AB_WGRAD_WS=0): the two-workgroup kernel
    if (split_bf16 && ws) {
        using G = WGradWSGeom<N>;
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_wgrad_b3w_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        hipLaunchKernelGGL(c6_wgrad_b3w_kernel<N>, dim3((C / 64) * (C / 64) * splits), dim3(512), G::LDS_BYTES, s, C, B, per, dy, x, partial);
    } else if (split_bf16) {
        using G = WGradB3Geom<N>;
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_wgrad_b3_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        hipLaunchKernelGGL(c6_wgrad_b3_kernel<N>, dim3((C / 64) * (C / 64), splits), dim3(256), G::LDS_BYTES, s, C, B, per, dy, x, partial);
    } else {
        using G = WGradGeom<N>;
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_wgrad_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        hipLaunchKernelGGL(c6_wgrad_kernel<N>, dim3((C / 64) * (C / 64), splits), dim3(256), G::LDS_BYTES, s, C, B, per, dy, x, partial);
    }
    hipLaunchKernelGGL(c6t_wgrad_reduce_kernel, dim3(1024), dim3(256), 0, s, C, splits, partial, dw);
}
// split_bf16 = 0: exact FP32 MFMA; 1: split-bf16 products (3 bf16 MFMAs per product), the trainer's default precision
bool c6t_wgrad(int N, int C, int B, const float *dy, const float *x, float *partial, float *dw, hipStream_t s, int split_bf16) {
    if (C % 64) return false;
    switch (N) {
    case 9: launch_wgrad<9>(C, B, dy, x, partial, dw, split_bf16, s); return true;
    case 15: launch_wgrad<15>(C, B, dy, x, partial, dw, split_bf16, s); return true;
    case 19: launch_wgrad<19>(C, B, dy, x, partial, dw, split_bf16, s); return true;
    default: return false;
    }
}
