// trainer.cpp -- host side of the training step: the C ABI c6_trainer_* of include/connect6_amd.h.
// Reference being reproduced: the train loop body of AlphaZero/cpp_weighted/main.cpp:148-160 --
//     optimizer.zero_grad(); loss = model->loss(winners, players, boards, poses); loss.backward(); optimizer.step();
// with model = WeightedPolicy in train() mode (model.hpp:52-108, BatchNorm on batch statistics, running statistics
// updated with momentum 0.1) and torch::optim::SGD(lr, momentum) (main.cpp:107-109).  Kernels: train_kernels.hip; the
// 3x3 convolutions of the forward pass and of the input gradient are the evaluator's c6_conv3x3_b3s_kernel.
#include "../../include/connect6_amd.h"
#include "c6_engine.h"
#include "c6_train.h"

#ifndef C6_AB_TRAIN_KEEP_V
#define C6_AB_TRAIN_KEEP_V 1      // A/B builds only (tools/ab_build.sh): 0 = the backward pass transforms x again
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string t_err;
static int tfail(int code, const std::string &msg) {
    t_err = msg;
    fprintf(stderr, "libconnect6_amd: %s\n", msg.c_str());
    return code;
}
#define TCHK(x)                                                                                         \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) return tfail(C6_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {
struct ConvBn { size_t w, b, g, be, rm, rv; };     // offsets into the blob
struct Bn { size_t g, be, rm, rv; };
}   // namespace

struct c6_trainer {
    int n = 0, cells = 0, C = 0, nb = 0, L = 0, B = 0, device = 0, num_cus = 256;
    float lr = 1e-3f, momentum = 0.9f;
    int precision = C6_PREC_FP32;       // of the 3x3 convolutions (forward, input gradient, weight gradient): f32 like the reference trainer
                                        // (main.cpp:107-157), the same meaning C6_PREC_DEFAULT has in the engine
    long steps = 0;
    size_t nfloats = 0, act = 0;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    float *P = nullptr, *G = nullptr, *M = nullptr;
    unsigned char *mask = nullptr;
    ConvBn stem;
    std::vector<ConvBn> tower;
    size_t pc_w = 0, pc_b = 0, pl_w = 0, pl_b = 0, vc_w = 0, vc_b = 0, v1_w = 0, v1_b = 0, v2_w = 0, v2_b = 0;
    Bn pn, vn;
    // batch
    float *winners = nullptr, *players = nullptr, *boards = nullptr, *poses = nullptr;
    // saved activations: a[l], y[l] for l = 0 (stem) .. L; batch statistics
    std::vector<float *> a, y;
    float *mean = nullptr, *var = nullptr;
    // heads
    float *yp = nullptr, *yv = nullptr, *hp = nullptr, *hv = nullptr, *logits = nullptr, *prob = nullptr, *lossp = nullptr, *dlogits = nullptr,
          *h1 = nullptr, *v = nullptr, *loss_dev = nullptr, *dpre = nullptr, *dh1 = nullptr, *dhv = nullptr, *dhp = nullptr, *dzp = nullptr,
          *dyp = nullptr, *dyv = nullptr, *pmean = nullptr, *pvar = nullptr, *vmean = nullptr, *vvar = nullptr;
    // scratch
    float *partial = nullptr, *sum0 = nullptr, *sum1 = nullptr, *X[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned short *wpack = nullptr;
    float *wpack32 = nullptr;
    float *wpackw = nullptr, *vbuf = nullptr;      // FP32 at 256 channels: Winograd F(4x4) (conv_wino4.hip), transformed weights + its scratch
    float *wgpart = nullptr;                        // ... and the partial sums of its weight gradient
    std::vector<float *> vkeep;                     // ... and every layer's transformed input, kept from the forward pass for the weight gradient
};

extern "C" size_t c6_net_blob_floats(int board_size, int channels, int blocks);

template <typename T>
static int talloc(c6_trainer *t, T **p, size_t count) {
    void *q = nullptr;
    TCHK(hipMalloc(&q, count * sizeof(T) + 256));
    TCHK(hipMemset(q, 0, count * sizeof(T) + 256));
    t->allocs.push_back(q);
    *p = reinterpret_cast<T *>(q);
    return C6_OK;
}

extern "C" int c6_trainer_create(int board_size, int channels, int blocks, int batch, float lr, float momentum, int device, c6_trainer **out) {
    if (!out) return tfail(C6_ERR_BAD_ARG, "c6_trainer_create: out == NULL");
    *out = nullptr;
    if (board_size != 9 && board_size != 15 && board_size != 19) return tfail(C6_ERR_BAD_ARG, "c6_trainer_create: board_size must be 9, 15 or 19");
    if (channels != 64 && channels != 128 && channels != 256) return tfail(C6_ERR_BAD_ARG, "c6_trainer_create: channels must be 64, 128 or 256");
    if (blocks < 1 || batch < 2) return tfail(C6_ERR_BAD_ARG, "c6_trainer_create: blocks >= 1 and batch >= 2 required");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return tfail(C6_ERR_NO_GPU, "libconnect6_amd: no HIP device visible; the trainer has no CPU fallback");
    if (device < 0 || device >= ndev) return tfail(C6_ERR_BAD_ARG, "c6_trainer_create: bad device ordinal");
    TCHK(hipSetDevice(device));

    c6_trainer *t = new c6_trainer();
    t->n = board_size; t->cells = board_size * board_size; t->C = channels; t->nb = blocks; t->L = 2 * (blocks - 1); t->B = batch;
    t->lr = lr; t->momentum = momentum; t->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) t->num_cus = prop.multiProcessorCount;
    const size_t n2 = t->cells, C = channels, B = batch;
    t->nfloats = c6_net_blob_floats(board_size, channels, blocks);

    // blob offsets (oracle/c6_oracle_net.c header) and the parameter mask
    std::vector<unsigned char> mask(t->nfloats, 1);
    size_t o = 0;
    auto take = [&](size_t k) { size_t r = o; o += k; return r; };
    auto take_bn = [&](size_t k, size_t &g, size_t &be, size_t &rm, size_t &rv) {
        g = take(k); be = take(k); rm = take(k); rv = take(k);
        std::fill(mask.begin() + rm, mask.begin() + rm + 2 * k, 0);
    };
    t->stem.w = take(C * 2 * 9); t->stem.b = take(C); take_bn(C, t->stem.g, t->stem.be, t->stem.rm, t->stem.rv);
    t->tower.resize(t->L);
    for (int l = 0; l < t->L; l++) {
        ConvBn &u = t->tower[l];
        u.w = take(C * C * 9); u.b = take(C); take_bn(C, u.g, u.be, u.rm, u.rv);
    }
    t->pc_w = take(2 * C); t->pc_b = take(2); take_bn(2, t->pn.g, t->pn.be, t->pn.rm, t->pn.rv);
    t->pl_w = take(n2 * 2 * n2); t->pl_b = take(n2);
    t->vc_w = take(C); t->vc_b = take(1); take_bn(n2, t->vn.g, t->vn.be, t->vn.rm, t->vn.rv);
    t->v1_w = take(256 * n2); t->v1_b = take(256); t->v2_w = take(256); t->v2_b = take(1);
    if (o != t->nfloats) { delete t; return tfail(C6_ERR_BAD_ARG, "c6_trainer_create: blob layout mismatch"); }

    int rc = C6_OK;
#define TA(p, cnt) if (rc == C6_OK) rc = talloc(t, &(p), (cnt))
    if (hipStreamCreate(&t->stream) != hipSuccess) { delete t; return tfail(C6_ERR_HIP, "hipStreamCreate"); }
    TA(t->P, t->nfloats); TA(t->G, t->nfloats); TA(t->M, t->nfloats); TA(t->mask, t->nfloats);
    TA(t->winners, B); TA(t->players, B); TA(t->boards, B * n2); TA(t->poses, B);
    t->act = B * n2 * C + C6_ACT_SLACK;
    t->a.resize(t->L + 1); t->y.resize(t->L + 1);
    for (int l = 0; l <= t->L; l++) { TA(t->a[l], t->act); TA(t->y[l], t->act); }
    TA(t->mean, (size_t)(t->L + 1) * C); TA(t->var, (size_t)(t->L + 1) * C);
    const size_t R = B * n2;
    TA(t->yp, R * 2); TA(t->yv, R); TA(t->hp, R * 2); TA(t->hv, R); TA(t->logits, R); TA(t->prob, R); TA(t->lossp, B); TA(t->dlogits, R);
    TA(t->h1, B * 256); TA(t->v, B); TA(t->loss_dev, 4); TA(t->dpre, B); TA(t->dh1, B * 256); TA(t->dhv, R); TA(t->dhp, R * 2); TA(t->dzp, R * 2);
    TA(t->dyp, R * 2); TA(t->dyv, R); TA(t->pmean, 2); TA(t->pvar, 2); TA(t->vmean, n2); TA(t->vvar, n2);
    size_t part = c6t_colstats_partial_floats(R, (int)C);
    part = std::max(part, c6t_colstats_partial_floats(R, 2));
    part = std::max(part, c6t_colstats_partial_floats(B, (int)std::max<size_t>(n2, 256)));
    part = std::max(part, c6t_stem_bwd_partial_floats((int)C, (int)B));
    part = std::max(part, c6t_head_conv_bwd_partial_floats(R, (int)C));
    part = std::max(part, c6t_wgrad_partial_floats((int)C, (int)B));
    part = std::max(part, (size_t)c6_conv_wino4_out_grid((int)board_size, (int)B, t->num_cus) * 2 * C);      // (r6: the forward convolutions' output kernel leaves the BatchNorm sums here)
    TA(t->partial, part);
    TA(t->sum0, std::max<size_t>(C, std::max<size_t>(n2, 256))); TA(t->sum1, std::max<size_t>(C, std::max<size_t>(n2, 256)));
    for (int q = 0; q < 4; q++) TA(t->X[q], t->act);
    TA(t->wpack, c6_conv_packed_b3s_units((int)C));
    TA(t->wpack32, c6_conv_packed_f32n_floats((int)C));
    if (c6_conv_wino4_vbuf_floats((int)C, t->n, (int)B)) {
        TA(t->wpackw, c6_conv_packed_wino4_floats((int)C));
        TA(t->vbuf, c6_conv_wino4_vbuf_floats((int)C, t->n, (int)B));
        TA(t->wgpart, c6_conv_wino4_wgrad_partial_floats((int)C, t->n, (int)B, t->num_cus));
        // the forward pass keeps every layer's B^T x B for the weight gradient (C6_AB_TRAIN_KEEP_V=0 in an A/B build: transform x again)
        const size_t vf = c6_conv_wino4_v_floats((int)C, t->n, (int)B);
        if (C6_AB_TRAIN_KEEP_V && vf * 4 * (size_t)t->L <= ((size_t)64 << 30)) {
            t->vkeep.assign(t->L, nullptr);
            for (int l = 0; l < t->L; l++) TA(t->vkeep[l], vf);
        }
    }
#undef TA
    if (rc == C6_OK && hipMemcpy(t->mask, mask.data(), t->nfloats, hipMemcpyHostToDevice) != hipSuccess) rc = tfail(C6_ERR_HIP, "mask upload");
    if (rc != C6_OK) {
        for (void *p : t->allocs) hipFree(p);
        hipStreamDestroy(t->stream);
        delete t;
        return rc;
    }
    *out = t;
    return C6_OK;
}

extern "C" void c6_trainer_destroy(c6_trainer *t) {
    if (!t) return;
    hipSetDevice(t->device);
    hipStreamSynchronize(t->stream);
    for (void *p : t->allocs) hipFree(p);
    hipStreamDestroy(t->stream);
    delete t;
}

extern "C" int c6_trainer_load_weights(c6_trainer *t, const float *blob, size_t nfloats) {
    if (!t || !blob) return tfail(C6_ERR_BAD_ARG, "c6_trainer_load_weights: NULL argument");
    if (nfloats == t->nfloats + C6_BLOB_HEADER_FLOATS) {      // self-describing blob: verify geometry + checksum, skip the header
        if (c6_net_blob_check_header(blob, nfloats, t->n, t->C, t->nb) != C6_OK) return tfail(C6_ERR_BAD_ARG, c6_last_error());
        blob += C6_BLOB_HEADER_FLOATS;
        nfloats -= C6_BLOB_HEADER_FLOATS;
    }
    if (nfloats != t->nfloats) return tfail(C6_ERR_BAD_ARG, "c6_trainer_load_weights: blob size does not match c6_net_blob_floats()");
    TCHK(hipSetDevice(t->device));
    TCHK(hipMemcpy(t->P, blob, nfloats * sizeof(float), hipMemcpyHostToDevice));
    TCHK(hipMemset(t->M, 0, nfloats * sizeof(float)));
    t->steps = 0;
    return C6_OK;
}

extern "C" int c6_trainer_export_weights(c6_trainer *t, float *blob, size_t nfloats) {
    if (!t || !blob) return tfail(C6_ERR_BAD_ARG, "c6_trainer_export_weights: NULL argument");
    if (nfloats != t->nfloats) return tfail(C6_ERR_BAD_ARG, "c6_trainer_export_weights: blob size does not match c6_net_blob_floats()");
    TCHK(hipSetDevice(t->device));
    TCHK(hipStreamSynchronize(t->stream));
    TCHK(hipMemcpy(blob, t->P, nfloats * sizeof(float), hipMemcpyDeviceToHost));
    return C6_OK;
}

extern "C" int c6_trainer_gradients(c6_trainer *t, float *blob, size_t nfloats) {
    if (!t || !blob) return tfail(C6_ERR_BAD_ARG, "c6_trainer_gradients: NULL argument");
    if (nfloats != t->nfloats) return tfail(C6_ERR_BAD_ARG, "c6_trainer_gradients: blob size does not match c6_net_blob_floats()");
    TCHK(hipSetDevice(t->device));
    TCHK(hipStreamSynchronize(t->stream));
    TCHK(hipMemcpy(blob, t->G, nfloats * sizeof(float), hipMemcpyDeviceToHost));
    return C6_OK;
}

// ---- forward in train mode (model.hpp:52-95 with batch statistics) + loss (model.hpp:97-108) ------------------------
// dW (OIHW) of a tower convolution from dy and the layer's input x: FP32 at 256 channels in the Winograd domain (conv_wino4.hip), else the direct kernels
static bool is_f32(const c6_trainer *t) { return t->precision == C6_PREC_FP32 || t->precision == C6_PREC_FP32_DIRECT; }
static bool wgrad(c6_trainer *t, const float *dy, const float *x, float *dw, int layer) {
    if (t->precision == C6_PREC_FP32 && t->wgpart)
        return c6_conv_wino4_wgrad(t->C, t->n, t->B, dy, x, t->vbuf, t->wgpart, dw, t->num_cus, t->stream, t->vkeep.empty() ? nullptr : t->vkeep[layer]);
    return c6t_wgrad(t->n, t->C, t->B, dy, x, t->partial, dw, t->stream, !is_f32(t));
}
// keep_layer >= 0: a forward convolution whose transformed input is kept for the weight gradient
// *stats_chunks (forward passes): set to the number of [2][C] partials of the outputs' column sums the convolution left in t->partial, or to 0 when the
// kernel taken cannot leave them (the caller then runs c6t_colstats over the stored outputs)
// bn_bwd (input-gradient passes): {aout, y, mean, var} of the BatchNorm + ReLU unit whose output gradient this convolution produces -- the partials are then
// that unit's backward sums (c6t_colstats mode 1) instead of the forward statistics
static int conv3x3(c6_trainer *t, const float *w_oihw, int transpose_flip, const float *in, const float *bias, const float *res, float *out, int keep_layer = -1,
                   int *stats_chunks = nullptr, const float *const *bn_bwd = nullptr) {
    bool ok;
    if (stats_chunks) *stats_chunks = 0;
    if (t->precision == C6_PREC_FP32 && t->vbuf) {      // 256 channels: the evaluator's default FP32 form (2.2x fewer ms than the direct kernel here)
        c6_pack_conv_weights_wino4_dev(t->C, w_oihw, t->wpackw, transpose_flip, t->stream);
        float *sp = stats_chunks ? t->partial : nullptr;
        if (keep_layer >= 0 && !t->vkeep.empty())
            ok = c6_conv_wino4_dispatch(t->C, t->n, t->B, in, t->wpackw, bias, res, out, t->vkeep[keep_layer], t->num_cus, t->stream, /*relu=*/0, 7, nullptr, nullptr,
                                        t->vbuf + c6_conv_wino4_v_floats(t->C, t->n, t->B), sp, bn_bwd);
        else
            ok = c6_conv_wino4_dispatch(t->C, t->n, t->B, in, t->wpackw, bias, res, out, t->vbuf, t->num_cus, t->stream, /*relu=*/0, 7, nullptr, nullptr, nullptr, sp, bn_bwd);
        if (ok && stats_chunks) *stats_chunks = c6_conv_wino4_out_grid(t->n, t->B, t->num_cus);
    } else if (is_f32(t)) {
        c6t_pack_f32n(t->C, w_oihw, t->wpack32, transpose_flip, t->stream);
        ok = c6_conv_f32n_dispatch(t->C, t->n, t->B, in, t->wpack32, bias, res, out, t->num_cus, t->stream, /*relu=*/0);
    } else {
        c6t_pack_b3s(t->C, w_oihw, t->wpack, transpose_flip, t->stream);
        ok = c6_conv_b3s_dispatch(t->C, t->n, t->B, in, t->wpack, bias, res, out, t->num_cus, t->stream, /*relu=*/0);
    }
    if (!ok) return tfail(C6_ERR_BAD_ARG, "trainer: no convolution kernel for this (channels, board_size)");
    return C6_OK;
}

static int forward_loss(c6_trainer *t, bool want_grad_seed) {
    const int C = t->C, n2 = t->cells, B = t->B, L = t->L;
    const size_t R = (size_t)B * n2;
    hipStream_t s = t->stream;
    float *P = t->P;
    // stem
    c6t_stem_fwd(t->n, C, B, t->boards, t->players, P + t->stem.w, P + t->stem.b, t->y[0], s);
    c6t_colstats(0, R, C, t->y[0], nullptr, nullptr, nullptr, nullptr, t->partial, t->mean, t->var, s);
    c6t_bn_apply(R, C, t->y[0], t->mean, t->var, P + t->stem.g, P + t->stem.be, nullptr, t->a[0], 1, s);
    c6t_bn_running(C, R, t->mean, t->var, P + t->stem.rm, P + t->stem.rv, s);
    // tower
    for (int l = 0; l < L; l++) {
        const ConvBn &u = t->tower[l];
        int chunks = 0;
        int rc = conv3x3(t, P + u.w, 0, t->a[l], P + u.b, nullptr, t->y[l + 1], l, &chunks);
        if (rc) return rc;
        float *m = t->mean + (size_t)(l + 1) * C, *v = t->var + (size_t)(l + 1) * C;
        if (chunks) c6t_colstats_finalize(0, R, C, chunks, t->partial, m, v, s);      // (r6: the sums came with the convolution's outputs)
        else c6t_colstats(0, R, C, t->y[l + 1], nullptr, nullptr, nullptr, nullptr, t->partial, m, v, s);
        const float *res = (l & 1) ? t->a[l - 1] : nullptr;              // second conv of a block: + block input, model.hpp:71
        c6t_bn_apply(R, C, t->y[l + 1], m, v, P + u.g, P + u.be, res, t->a[l + 1], 1, s);
        c6t_bn_running(C, R, m, v, P + u.rm, P + u.rv, s);
    }
    const float *top = t->a[L];
    // heads
    c6t_head_conv_fwd(R, C, top, P + t->pc_w, P + t->pc_b, P + t->vc_w, P + t->vc_b, t->yp, t->yv, s);
    c6t_colstats(0, R, 2, t->yp, nullptr, nullptr, nullptr, nullptr, t->partial, t->pmean, t->pvar, s);
    c6t_policy_bn_apply(R, n2, t->yp, t->pmean, t->pvar, P + t->pn.g, P + t->pn.be, t->hp, s);
    c6t_bn_running(2, R, t->pmean, t->pvar, P + t->pn.rm, P + t->pn.rv, s);
    c6t_gemm(B, n2, 2 * n2, t->hp, 2 * n2, 1, P + t->pl_w, 1, 2 * n2, P + t->pl_b, t->logits, n2, 0, s);      // hp x Wl^T + bl
    c6t_softmax_loss(B, n2, t->logits, t->poses, t->prob, t->lossp, t->dlogits, s);
    c6t_colstats(0, (size_t)B, n2, t->yv, nullptr, nullptr, nullptr, nullptr, t->partial, t->vmean, t->vvar, s);   // BatchNorm1d over the batch
    c6t_bn_apply((size_t)B, n2, t->yv, t->vmean, t->vvar, P + t->vn.g, P + t->vn.be, nullptr, t->hv, 1, s);
    c6t_bn_running(n2, (size_t)B, t->vmean, t->vvar, P + t->vn.rm, P + t->vn.rv, s);
    c6t_gemm(B, 256, n2, t->hv, n2, 1, P + t->v1_w, 1, n2, P + t->v1_b, t->h1, 256, 1, s);                     // relu(hv x W1^T + b1)
    c6t_value_out(B, 256, t->h1, P + t->v2_w, P + t->v2_b, t->v, s);
    c6t_loss(B, t->v, t->winners, t->lossp, t->loss_dev, want_grad_seed ? t->dpre : nullptr, s);
    return C6_OK;
}

static int backward(c6_trainer *t) {
    const int C = t->C, n2 = t->cells, B = t->B, L = t->L;
    const size_t R = (size_t)B * n2;
    hipStream_t s = t->stream;
    float *P = t->P, *G = t->G;
    // ---- policy head
    c6t_gemm(n2, 2 * n2, B, t->dlogits, 1, n2, t->hp, 2 * n2, 1, nullptr, G + t->pl_w, 2 * n2, 0, s);          // dWl = dlogits^T x hp
    c6t_colstats(2, (size_t)B, n2, t->dlogits, nullptr, nullptr, nullptr, nullptr, t->partial, G + t->pl_b, nullptr, s);
    c6t_gemm(B, 2 * n2, n2, t->dlogits, n2, 1, P + t->pl_w, 2 * n2, 1, nullptr, t->dhp, 2 * n2, 0, s);          // dhp = dlogits x Wl
    c6t_policy_unflatten(R, n2, t->dhp, t->hp, t->dzp, s);
    c6t_colstats(1, R, 2, t->dzp, nullptr, t->yp, t->pmean, t->pvar, t->partial, G + t->pn.be, G + t->pn.g, s);
    c6t_bn_bwd_apply(R, 2, t->dzp, nullptr, t->yp, t->pmean, t->pvar, P + t->pn.g, G + t->pn.be, G + t->pn.g, t->dyp, nullptr, s);
    // ---- value head
    c6t_gemm(1, 256, B, t->dpre, 0, 1, t->h1, 256, 1, nullptr, G + t->v2_w, 256, 0, s);                         // dw2 = dpre^T x h1
    c6t_colstats(2, (size_t)B, 1, t->dpre, nullptr, nullptr, nullptr, nullptr, t->partial, G + t->v2_b, nullptr, s);
    c6t_value_bwd_h1(B, 256, t->dpre, P + t->v2_w, t->h1, t->dh1, s);
    c6t_gemm(256, n2, B, t->dh1, 1, 256, t->hv, n2, 1, nullptr, G + t->v1_w, n2, 0, s);                          // dW1 = dh1^T x hv
    c6t_colstats(2, (size_t)B, 256, t->dh1, nullptr, nullptr, nullptr, nullptr, t->partial, G + t->v1_b, nullptr, s);
    c6t_gemm(B, n2, 256, t->dh1, 256, 1, P + t->v1_w, n2, 1, nullptr, t->dhv, n2, 0, s);                         // dhv = dh1 x W1
    c6t_colstats(1, (size_t)B, n2, t->dhv, t->hv, t->yv, t->vmean, t->vvar, t->partial, G + t->vn.be, G + t->vn.g, s);
    c6t_bn_bwd_apply((size_t)B, n2, t->dhv, t->hv, t->yv, t->vmean, t->vvar, P + t->vn.g, G + t->vn.be, G + t->vn.g, t->dyv, nullptr, s);
    // ---- 1x1 head convolutions -> gradient of the tower output
    float *cur = t->X[0];
    c6t_head_conv_bwd(R, C, t->a[L], t->dyp, t->dyv, P + t->pc_w, P + t->vc_w, t->partial, G + t->pc_w, G + t->vc_w, cur, s);
    c6t_colstats(2, R, 2, t->dyp, nullptr, nullptr, nullptr, nullptr, t->partial, G + t->pc_b, nullptr, s);
    c6t_colstats(2, R, 1, t->dyv, nullptr, nullptr, nullptr, nullptr, t->partial, G + t->vc_b, nullptr, s);
    // ---- tower, last block first
    float *dy = t->X[1], *dz = t->X[2], *dmid = t->X[3];
    int chunks = 0;      // r6: > 0 when the convolution that produced the gradient at hand left the next BatchNorm's backward sums in t->partial
    for (int blk = t->nb - 2; blk >= 0; blk--) {
        const int l0 = 2 * blk, l1 = 2 * blk + 1;
        {   // second conv unit: a[l1+1] = relu(bn(conv(a[l1])) + a[l0])
            const ConvBn &u = t->tower[l1];
            const float *m = t->mean + (size_t)(l1 + 1) * C, *v = t->var + (size_t)(l1 + 1) * C;
            if (chunks) c6t_colstats_finalize(1, R, C, chunks, t->partial, G + u.be, G + u.g, s);
            else c6t_colstats(1, R, C, cur, t->a[l1 + 1], t->y[l1 + 1], m, v, t->partial, G + u.be, G + u.g, s);
            c6t_bn_bwd_apply_colsum(R, C, cur, t->a[l1 + 1], t->y[l1 + 1], m, v, P + u.g, G + u.be, G + u.g, dy, dz, t->partial, G + u.b, s);
            if (!wgrad(t, dy, t->a[l1], G + u.w, l1)) return tfail(C6_ERR_BAD_ARG, "trainer: no weight-gradient kernel");
            // dmid = the gradient that reaches the block's FIRST unit: its backward sums come with it
            const float *nb_[4] = {t->a[l0 + 1], t->y[l0 + 1], t->mean + (size_t)(l0 + 1) * C, t->var + (size_t)(l0 + 1) * C};
            int rc = conv3x3(t, P + u.w, 1, dy, nullptr, nullptr, dmid, -1, &chunks, nb_);
            if (rc) return rc;
        }
        {   // first conv unit: a[l0+1] = relu(bn(conv(a[l0])))
            const ConvBn &u = t->tower[l0];
            const float *m = t->mean + (size_t)(l0 + 1) * C, *v = t->var + (size_t)(l0 + 1) * C;
            if (chunks) c6t_colstats_finalize(1, R, C, chunks, t->partial, G + u.be, G + u.g, s);
            else c6t_colstats(1, R, C, dmid, t->a[l0 + 1], t->y[l0 + 1], m, v, t->partial, G + u.be, G + u.g, s);
            c6t_bn_bwd_apply_colsum(R, C, dmid, t->a[l0 + 1], t->y[l0 + 1], m, v, P + u.g, G + u.be, G + u.g, dy, nullptr, t->partial, G + u.b, s);
            if (!wgrad(t, dy, t->a[l0], G + u.w, l0)) return tfail(C6_ERR_BAD_ARG, "trainer: no weight-gradient kernel");
            // cur = the gradient that reaches the unit in front of this block: the previous block's second unit (output a[l0]), or the stem
            const float *nb_[4] = {t->a[l0], t->y[l0], t->mean + (size_t)l0 * C, t->var + (size_t)l0 * C};
            int rc = conv3x3(t, P + u.w, 1, dy, nullptr, dz /* + the skip connection's gradient */, cur, -1, &chunks, nb_);
            if (rc) return rc;
        }
    }
    // ---- stem
    if (chunks) c6t_colstats_finalize(1, R, C, chunks, t->partial, G + t->stem.be, G + t->stem.g, s);
    else c6t_colstats(1, R, C, cur, t->a[0], t->y[0], t->mean, t->var, t->partial, G + t->stem.be, G + t->stem.g, s);
    c6t_bn_bwd_apply(R, C, cur, t->a[0], t->y[0], t->mean, t->var, P + t->stem.g, G + t->stem.be, G + t->stem.g, dy, nullptr, s);
    c6t_stem_bwd(t->n, C, B, t->boards, t->players, dy, t->partial, G + t->stem.w, G + t->stem.b, s);
    return C6_OK;
}

static int upload_batch(c6_trainer *t, const float *winners, const float *players, const float *boards, const float *poses, int on_device) {
    if (!winners || !players || !boards || !poses) return tfail(C6_ERR_BAD_ARG, "trainer: NULL batch pointer");
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (!on_device)
        for (int i = 0; i < t->B; i++)
            if (!(poses[i] >= 0.f && poses[i] < (float)t->cells)) return tfail(C6_ERR_BAD_ARG, "trainer: position outside the board");
    TCHK(hipMemcpyAsync(t->winners, winners, (size_t)t->B * 4, kind, t->stream));
    TCHK(hipMemcpyAsync(t->players, players, (size_t)t->B * 4, kind, t->stream));
    TCHK(hipMemcpyAsync(t->boards, boards, (size_t)t->B * t->cells * 4, kind, t->stream));
    TCHK(hipMemcpyAsync(t->poses, poses, (size_t)t->B * 4, kind, t->stream));
    return C6_OK;
}

static int read_loss(c6_trainer *t, float *loss_out) {
    if (loss_out) {
        TCHK(hipMemcpyAsync(loss_out, t->loss_dev, sizeof(float), hipMemcpyDeviceToHost, t->stream));
        TCHK(hipStreamSynchronize(t->stream));
    }
    TCHK(hipGetLastError());
    return C6_OK;
}

static int step_resident(c6_trainer *t, float *loss_out) {
    int rc = forward_loss(t, true);
    if (rc) return rc;
    TCHK(hipMemsetAsync(t->G, 0, t->nfloats * sizeof(float), t->stream));       // optimizer.zero_grad()
    rc = backward(t);
    if (rc) return rc;
    c6t_sgd(t->nfloats, t->lr, t->momentum, t->steps == 0, t->mask, t->G, t->M, t->P, t->stream);
    t->steps++;
    return read_loss(t, loss_out);
}

extern "C" int c6_trainer_step(c6_trainer *t, const float *winners, const float *players, const float *boards, const float *poses, int on_device,
                               float *loss_out) {
    if (!t) return tfail(C6_ERR_BAD_ARG, "c6_trainer_step: NULL trainer");
    TCHK(hipSetDevice(t->device));
    int rc = upload_batch(t, winners, players, boards, poses, on_device);
    if (rc) return rc;
    return step_resident(t, loss_out);
}

extern "C" int c6_trainer_loss(c6_trainer *t, const float *winners, const float *players, const float *boards, const float *poses, int on_device,
                               float *loss_out) {
    if (!t) return tfail(C6_ERR_BAD_ARG, "c6_trainer_loss: NULL trainer");
    TCHK(hipSetDevice(t->device));
    int rc = upload_batch(t, winners, players, boards, poses, on_device);
    if (rc) return rc;
    rc = forward_loss(t, false);
    if (rc) return rc;
    return read_loss(t, loss_out);
}

// sample a mini-batch from the device-resident replay buffer and train on it: ReplayBuffer::sample -> loss -> step without
// a host round trip (main.cpp:151-157)
extern "C" int c6_replay_sample_device(c6_replay *r, int num_sample, uint64_t seed, uint32_t draw_id, float *winners, float *players,
                                       float *boards, float *poses, void *stream);
extern "C" int c6_replay_board_size(const c6_replay *r);
extern "C" int c6_trainer_step_replay(c6_trainer *t, c6_replay *r, uint64_t seed, uint32_t draw_id, float *loss_out) {
    if (!t || !r) return tfail(C6_ERR_BAD_ARG, "c6_trainer_step_replay: NULL argument");
    if (c6_replay_board_size(r) != t->n) return tfail(C6_ERR_BAD_ARG, "c6_trainer_step_replay: board size mismatch");
    TCHK(hipSetDevice(t->device));
    int rc = c6_replay_sample_device(r, t->B, seed, draw_id, t->winners, t->players, t->boards, t->poses, t->stream);
    if (rc) return rc;
    return step_resident(t, loss_out);
}

extern "C" int c6_trainer_set_lr(c6_trainer *t, float lr, float momentum) {
    if (!t) return tfail(C6_ERR_BAD_ARG, "c6_trainer_set_lr: NULL trainer");
    t->lr = lr; t->momentum = momentum;
    return C6_OK;
}

extern "C" int c6_trainer_set_precision(c6_trainer *t, int precision) {
    if (!t) return tfail(C6_ERR_BAD_ARG, "c6_trainer_set_precision: NULL trainer");
    // C6_PREC_DEFAULT = C6_PREC_FP32, as in the engine: the reference trainer is f32 (main.cpp:107-157).  C6_PREC_FP32_DIRECT: the same
    // arithmetic on the direct kernels (forward / input gradient: c6_conv3x3_f32n_kernel, weight gradient: c6_wgrad_kernel) -- the yardstick
    // the Winograd-domain path is tested against.  C6_PREC_BF16X3: split-bf16 products (gradients within ~1e-3), the fast option.
    if (precision == C6_PREC_DEFAULT) precision = C6_PREC_FP32;
    if (precision != C6_PREC_FP32 && precision != C6_PREC_FP32_DIRECT && precision != C6_PREC_BF16X3)
        return tfail(C6_ERR_BAD_ARG, "c6_trainer_set_precision: C6_PREC_DEFAULT (= C6_PREC_FP32), C6_PREC_FP32, C6_PREC_FP32_DIRECT or C6_PREC_BF16X3");
    t->precision = precision;
    return C6_OK;
}

extern "C" const char *c6_trainer_last_error(void) { return t_err.c_str(); }

// Stand-alone operator for the parity tests: dW (OIHW [C][C][3][3]) of y = conv3x3(x, w) from dy and x (both NHWC [B][N*N][C], host
// buffers), in f32 -- method 0: the direct kernel (c6_wgrad_kernel, train_kernels.hip), 1: in the Winograd domain (c6_conv_wino4_wgrad,
// conv_wino4.hip; C = 256).  The two are independent implementations of the same sums.
extern "C" int c6_conv_wgrad_op(int n, int C, int B, const float *dy, const float *x, int method, float *dw) {
    if (!dy || !x || !dw) return tfail(C6_ERR_BAD_ARG, "c6_conv_wgrad_op: NULL argument");
    if ((n != 9 && n != 15 && n != 19) || C % 64 || B < 1 || (method != 0 && method != 1)) return tfail(C6_ERR_BAD_ARG, "c6_conv_wgrad_op: bad geometry / method");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return tfail(C6_ERR_NO_GPU, "c6_conv_wgrad_op: no HIP device (this library has no CPU fallback)");
    hipDeviceProp_t prop;
    const int cus = hipGetDeviceProperties(&prop, 0) == hipSuccess ? prop.multiProcessorCount : 256;
    const size_t act = (size_t)B * n * n * C;
    float *d_dy = nullptr, *d_x = nullptr, *d_dw = nullptr, *d_part = nullptr, *d_vbuf = nullptr;
    int rc = C6_OK;
    auto done = [&](int r) { hipFree(d_dy); hipFree(d_x); hipFree(d_dw); hipFree(d_part); hipFree(d_vbuf); return r; };
    if (hipMalloc(&d_dy, act * 4) != hipSuccess || hipMalloc(&d_x, act * 4) != hipSuccess || hipMalloc(&d_dw, (size_t)9 * C * C * 4) != hipSuccess)
        return done(tfail(C6_ERR_HIP, "c6_conv_wgrad_op: hipMalloc"));
    hipMemcpy(d_dy, dy, act * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_x, x, act * 4, hipMemcpyHostToDevice);
    bool ok;
    if (method == 1) {
        const size_t vb = c6_conv_wino4_vbuf_floats(C, n, B), pf = c6_conv_wino4_wgrad_partial_floats(C, n, B, cus);
        if (!vb) return done(tfail(C6_ERR_BAD_ARG, "c6_conv_wgrad_op: the Winograd-domain weight gradient needs C = 256"));
        if (hipMalloc(&d_vbuf, vb * 4) != hipSuccess || hipMalloc(&d_part, pf * 4) != hipSuccess) return done(tfail(C6_ERR_HIP, "c6_conv_wgrad_op: hipMalloc"));
        ok = c6_conv_wino4_wgrad(C, n, B, d_dy, d_x, d_vbuf, d_part, d_dw, cus, nullptr, nullptr);
    } else {
        if (hipMalloc(&d_part, c6t_wgrad_partial_floats(C, B) * 4) != hipSuccess) return done(tfail(C6_ERR_HIP, "c6_conv_wgrad_op: hipMalloc"));
        ok = c6t_wgrad(n, C, B, d_dy, d_x, d_part, d_dw, nullptr, 0);
    }
    if (!ok) rc = tfail(C6_ERR_BAD_ARG, "c6_conv_wgrad_op: no kernel for this geometry");
    if (hipDeviceSynchronize() != hipSuccess) rc = tfail(C6_ERR_HIP, "c6_conv_wgrad_op: kernel failed");
    if (rc == C6_OK) hipMemcpy(dw, d_dw, (size_t)9 * C * C * 4, hipMemcpyDeviceToHost);
    return done(rc);
}
