// c6_train.h -- launchers of train_kernels.hip (internal to the library; the C ABI is include/connect6_amd.h, part 3)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

void c6t_pack_b3s(int C, const float *w_oihw, unsigned short *packed, int transpose_flip, hipStream_t s);
void c6t_pack_f32n(int C, const float *w_oihw, float *packed, int transpose_flip, hipStream_t s);

void c6t_stem_fwd(int N, int C, int B, const float *boards, const float *players, const float *w, const float *bias, float *y, hipStream_t s);
size_t c6t_stem_bwd_partial_floats(int C, int B);
void c6t_stem_bwd(int N, int C, int B, const float *boards, const float *players, const float *dy, float *partial, float *dw, float *db, hipStream_t s);

// mode 0: out0 = column mean, out1 = biased variance; 1: BatchNorm-backward sums (dz = x masked by aout > 0); 2: column sum
size_t c6t_colstats_partial_floats(size_t R, int C);
void c6t_colstats(int mode, size_t R, int C, const float *x, const float *aout, const float *y, const float *mean, const float *var,
                  float *partial, float *out0, float *out1, hipStream_t s);
void c6t_colstats_finalize(int mode, size_t R, int C, int nchunk, const float *partial, float *out0, float *out1, hipStream_t s);
void c6t_bn_apply(size_t R, int C, const float *y, const float *mean, const float *var, const float *gamma, const float *beta, const float *res,
                  float *a, int relu, hipStream_t s);
void c6t_bn_bwd_apply(size_t R, int C, const float *dout, const float *aout, const float *y, const float *mean, const float *var,
                      const float *gamma, const float *sum_dz, const float *sum_dzx, float *dy, float *dz_out, hipStream_t s);
// ... with the column sums of dy (the bias gradient of the convolution in front) taken in the same pass (r6)
void c6t_bn_bwd_apply_colsum(size_t R, int C, const float *dout, const float *aout, const float *y, const float *mean, const float *var, const float *gamma,
                             const float *sum_dz, const float *sum_dzx, float *dy, float *dz_out, float *partial, float *bias_grad, hipStream_t s);
void c6t_bn_running(int C, size_t R, const float *mean, const float *var, float *rmean, float *rvar, hipStream_t s);

void c6t_head_conv_fwd(size_t R, int C, const float *a, const float *wp, const float *bp, const float *wv, const float *bv, float *yp, float *yv,
                       hipStream_t s);
void c6t_policy_bn_apply(size_t R, int cells, const float *yp, const float *mean, const float *var, const float *gamma, const float *beta, float *hp,
                         hipStream_t s);
void c6t_policy_unflatten(size_t R, int cells, const float *dhp, const float *hp, float *dzp, hipStream_t s);
void c6t_gemm(int M, int N, int K, const float *A, long sAm, long sAk, const float *Bm, long sBk, long sBn, const float *bias, float *Cm, int ldc,
              int relu, hipStream_t s);
void c6t_softmax_loss(int B, int cells, const float *logits, const float *poses, float *prob, float *lossp, float *dlogits, hipStream_t s);
void c6t_value_out(int B, int H, const float *h1, const float *w2, const float *b2, float *v, hipStream_t s);
void c6t_loss(int B, const float *v, const float *winners, const float *lossp, float *loss_out, float *dpre, hipStream_t s);
void c6t_value_bwd_h1(int B, int H, const float *dpre, const float *w2, const float *h1, float *dh1, hipStream_t s);
size_t c6t_head_conv_bwd_partial_floats(size_t R, int C);
void c6t_head_conv_bwd(size_t R, int C, const float *a, const float *dyp, const float *dyv, const float *wp, const float *wv, float *partial,
                       float *dwp, float *dwv, float *da, hipStream_t s);

void c6t_sgd(size_t n, float lr, float momentum, int first, const unsigned char *mask, const float *g, float *buf, float *p, hipStream_t s);

int c6t_wgrad_splits(int C, int B);
size_t c6t_wgrad_partial_floats(int C, int B);
bool c6t_wgrad(int N, int C, int B, const float *dy, const float *x, float *partial, float *dw, hipStream_t s, int split_bf16 = 0);
