/*
 * c6_oracle_net.c -- CPU ORACLE (test infrastructure, NOT product code).
 * FP32 restatement of WeightedPolicy::forward,
 * AlphaZero/cpp_weighted/model.hpp:52-95 (constructor :10-50), with the
 * evaluator glue of AlphaZero/cpp_weighted/main.cpp:16-37.
 *
 * Weight blob = the module's parameters/buffers as flat f32 in this order
 * (PyTorch layouts, OIHW convolutions, [out][in] linears; BatchNorm = gamma,
 * beta, running_mean, running_var):
 *   conv.weight[C][2][3][3] conv.bias[C] norm[4][C]
 *   for i in 0..num_block-2: conv{i}_0.weight[C][C][3][3] .bias[C] norm{i}_0[4][C]
 *                            conv{i}_1.weight[C][C][3][3] .bias[C] norm{i}_1[4][C]
 *   policy_conv.weight[2][C] .bias[2] policy_norm[4][2]
 *   policy_linear.weight[n2][2*n2] .bias[n2]
 *   value_conv.weight[1][C] .bias[1] value_norm[4][n2]
 *   value_linear1.weight[256][n2] .bias[256] value_linear2.weight[1][256] .bias[1]
 * BatchNorm runs in eval mode (running statistics), eps = 1e-5.
 */
#include "c6_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BN_EPS 1e-5f
#define VHID 256

typedef struct {
    const float *w, *b;       /* conv OIHW, bias */
    const float *g, *be, *m, *v; /* batch norm */
    float *wt;                /* repacked [tap][ci][co] */
} convbn;

struct c6o_net {
    int n, cells, C, nb;
    float *blob;
    convbn stem;
    convbn *tower; /* 2*(nb-1) */
    const float *pc_w, *pc_b, *pn, *pl_w, *pl_b;
    const float *vc_w, *vc_b, *vn, *v1_w, *v1_b, *v2_w, *v2_b;
};

size_t c6o_net_blob_floats(int n, int C, int nb) {
    size_t n2 = (size_t)n * n, c = (size_t)C, t = 0;
    t += c * 2 * 9 + c + 4 * c;
    t += (size_t)(nb - 1) * 2 * (c * c * 9 + c + 4 * c);
    t += 2 * c + 2 + 8 + n2 * 2 * n2 + n2;
    t += c + 1 + 4 * n2 + VHID * n2 + VHID + VHID + 1;
    return t;
}

static const float *take(const float **p, size_t k) {
    const float *r = *p;
    *p += k;
    return r;
}

static void convbn_init(convbn *cb, const float **p, int cin, int cout) {
    cb->w = take(p, (size_t)cout * cin * 9);
    cb->b = take(p, (size_t)cout);
    cb->g = take(p, (size_t)cout);
    cb->be = take(p, (size_t)cout);
    cb->m = take(p, (size_t)cout);
    cb->v = take(p, (size_t)cout);
    cb->wt = (float *)malloc(sizeof(float) * (size_t)9 * cin * cout);
    for (int co = 0; co < cout; co++)
        for (int ci = 0; ci < cin; ci++)
            for (int t = 0; t < 9; t++) cb->wt[((size_t)t * cin + ci) * cout + co] = cb->w[((size_t)co * cin + ci) * 9 + t];
}

c6o_net *c6o_net_new(int n, int C, int nb, const float *blob) {
    c6o_net *net = (c6o_net *)calloc(1, sizeof *net);
    size_t total = c6o_net_blob_floats(n, C, nb);
    net->n = n;
    net->cells = n * n;
    net->C = C;
    net->nb = nb;
    net->blob = (float *)malloc(sizeof(float) * total);
    memcpy(net->blob, blob, sizeof(float) * total);
    const float *p = net->blob;
    size_t n2 = (size_t)net->cells;
    convbn_init(&net->stem, &p, 2, C);
    net->tower = (convbn *)calloc((size_t)(2 * (nb - 1)) + 1, sizeof(convbn));
    for (int i = 0; i < 2 * (nb - 1); i++) convbn_init(&net->tower[i], &p, C, C);
    net->pc_w = take(&p, (size_t)2 * C);
    net->pc_b = take(&p, 2);
    net->pn = take(&p, 8);
    net->pl_w = take(&p, n2 * 2 * n2);
    net->pl_b = take(&p, n2);
    net->vc_w = take(&p, (size_t)C);
    net->vc_b = take(&p, 1);
    net->vn = take(&p, 4 * n2);
    net->v1_w = take(&p, VHID * n2);
    net->v1_b = take(&p, VHID);
    net->v2_w = take(&p, VHID);
    net->v2_b = take(&p, 1);
    return net;
}

void c6o_net_free(c6o_net *net) {
    if (!net) return;
    free(net->stem.wt);
    for (int i = 0; i < 2 * (net->nb - 1); i++) free(net->tower[i].wt);
    free(net->tower);
    free(net->blob);
    free(net);
}

/* 3x3 conv, pad 1, NHWC activations, then BatchNorm(eval); optional residual; ReLU.
 * model.hpp:59 (stem) and :64-74 (residual block halves).
 * Register-blocked (6 pixels x 16 output channels per micro-tile, AVX2 FMA through GCC vector
 * extensions) so that the CPU baseline is a fair one; the arithmetic is the plain convolution sum. */
typedef float v8f __attribute__((vector_size(32), aligned(4)));

/* per-thread scratch: the CPU baseline runs one game per core, and malloc/calloc of ~0.5 MB blocks (mmap + page faults)
 * in every call serialises hundreds of threads in the kernel */
static __thread float *tl_buf;
static __thread size_t tl_cap;
static float *scratch(size_t floats) {
    if (floats > tl_cap) {
        free(tl_buf);
        tl_buf = (float *)malloc(sizeof(float) * floats);
        tl_cap = floats;
    }
    return tl_buf;
}

static void conv3_bn(const convbn *cb, int n, int cin, int cout, const float *in, const float *res, float *out, float *pad) {
    const int P = n + 2;
    /* zero-haloed copy of the input: taps become constant offsets (pad has (P*P + 8) * cin floats: the last
     * micro-tile over-reads a little) */
    memset(pad, 0, sizeof(float) * ((size_t)P * P + 8) * (size_t)cin);
    for (int y = 0; y < n; y++)
        memcpy(pad + ((size_t)(y + 1) * P + 1) * cin, in + (size_t)y * n * cin, sizeof(float) * (size_t)n * cin);
    float scale[512], shift[512];
    for (int co = 0; co < cout; co++) {
        scale[co] = cb->g[co] / sqrtf(cb->v[co] + BN_EPS);
        shift[co] = cb->be[co] - cb->m[co] * scale[co];
    }
    enum { PX = 6 };
    /* co outermost: the 9*cin*16 weight block of a channel group (147 KB at cin = 256) stays in the core's cache
     * while every pixel tile of the board streams past it */
    for (int co = 0; co < cout; co += 16)
        for (int y = 0; y < n; y++) {
            for (int x0 = 0; x0 < n; x0 += PX) {
                const int npx = n - x0 < PX ? n - x0 : PX;
                v8f acc[PX][2];
                for (int q = 0; q < PX; q++) { acc[q][0] = *(const v8f *)(cb->b + co); acc[q][1] = *(const v8f *)(cb->b + co + 8); }
                for (int t = 0; t < 9; t++) {
                    const float *ip = pad + ((size_t)(y + t / 3) * P + (x0 + t % 3)) * cin;
                    const float *wp = cb->wt + (size_t)t * cin * cout + co;
                    for (int ci = 0; ci < cin; ci++) {
                        const v8f w0 = *(const v8f *)(wp + (size_t)ci * cout), w1 = *(const v8f *)(wp + (size_t)ci * cout + 8);
                        for (int q = 0; q < PX; q++) {
                            /* pixels beyond the row read the halo / next row of `pad`: computed, never stored */
                            const float a = ip[(size_t)q * cin + ci];
                            const v8f av = {a, a, a, a, a, a, a, a};
                            acc[q][0] += av * w0;
                            acc[q][1] += av * w1;
                        }
                    }
                }
                for (int q = 0; q < npx; q++) {
                    float *op = out + ((size_t)y * n + x0 + q) * cout + co;
                    const float *rp = res ? res + ((size_t)y * n + x0 + q) * cout + co : NULL;
                    for (int k = 0; k < 16; k++) {
                        float v = (k < 8 ? acc[q][0][k] : acc[q][1][k - 8]);
                        v = v * scale[co + k] + shift[co + k];
                        if (rp) v += rp[k];
                        op[k] = v > 0.f ? v : 0.f;
                    }
                }
            }
        }
}

void c6o_net_forward(const c6o_net *net, float player, const float *boards, int len, float *policy, float *value) {
    int n = net->n, cells = net->cells, C = net->C;
    const size_t act = (size_t)cells * C, padf = ((size_t)(n + 2) * (n + 2) + 8) * (size_t)C;
    float *base = scratch(3 * act + padf + (size_t)cells * 5 + 64);
    float *x = base, *y = x + act, *z = y + act, *pad = z + act;
    float *in2 = pad + padf, *ph = in2 + (size_t)cells * 2, *vh = ph + (size_t)cells * 2;
    float hid[VHID];
    for (int b = 0; b < len; b++) {
        const float *bd = boards + (size_t)b * cells;
        /* model.hpp:56-57: planes [player broadcast, board] */
        for (int i = 0; i < cells; i++) { in2[2 * i] = player; in2[2 * i + 1] = bd[i]; }
        conv3_bn(&net->stem, n, 2, C, in2, NULL, x, pad);
        for (int i = 0; i < net->nb - 1; i++) { /* res_block :61-78 */
            conv3_bn(&net->tower[2 * i], n, C, C, x, NULL, y, pad);
            conv3_bn(&net->tower[2 * i + 1], n, C, C, y, x, z, pad);
            float *t = x; x = z; z = t;
        }
        /* policy head :80-84 (flatten is channel-major: NCHW view) */
        for (int ch = 0; ch < 2; ch++)
            for (int i = 0; i < cells; i++) {
                float a = net->pc_b[ch];
                const float *xp = x + (size_t)i * C, *w = net->pc_w + (size_t)ch * C;
                for (int c = 0; c < C; c++) a += xp[c] * w[c];
                float g = net->pn[ch], be = net->pn[2 + ch], m = net->pn[4 + ch], v = net->pn[6 + ch];
                a = (a - m) / sqrtf(v + BN_EPS) * g + be;
                ph[ch * cells + i] = a > 0.f ? a : 0.f;
            }
        float *pol = policy + (size_t)b * cells;
        float mx = -INFINITY;
        for (int o = 0; o < cells; o++) {
            float a = net->pl_b[o];
            const float *w = net->pl_w + (size_t)o * 2 * cells;
            for (int k = 0; k < 2 * cells; k++) a += ph[k] * w[k];
            pol[o] = a;
            if (a > mx) mx = a;
        }
        float s = 0.f;
        for (int o = 0; o < cells; o++) { pol[o] = expf(pol[o] - mx); s += pol[o]; }
        for (int o = 0; o < cells; o++) pol[o] /= s;
        /* value head :86-92 (BatchNorm over the n*n flattened features) */
        for (int i = 0; i < cells; i++) {
            float a = net->vc_b[0];
            const float *xp = x + (size_t)i * C;
            for (int c = 0; c < C; c++) a += xp[c] * net->vc_w[c];
            float g = net->vn[i], be = net->vn[cells + i], m = net->vn[2 * cells + i], v = net->vn[3 * cells + i];
            a = (a - m) / sqrtf(v + BN_EPS) * g + be;
            vh[i] = a > 0.f ? a : 0.f;
        }
        for (int h = 0; h < VHID; h++) {
            float a = net->v1_b[h];
            const float *w = net->v1_w + (size_t)h * cells;
            for (int i = 0; i < cells; i++) a += vh[i] * w[i];
            hid[h] = a > 0.f ? a : 0.f;
        }
        float a = net->v2_b[0];
        for (int h = 0; h < VHID; h++) a += hid[h] * net->v2_w[h];
        value[b] = tanhf(a);
    }
}

/* evaluator glue, AlphaZero/cpp_weighted/main.cpp:16-37 */
int c6o_net_eval(void *ctx, int turn, int n, int len, const int8_t *boards, float *values, float *policies) {
    const c6o_net *net = (const c6o_net *)ctx;
    int cells = n * n;
    float bf[8 * C6O_MAXCELLS];
    for (int done = 0; done < len; done += 8) {   /* the reference always evaluates 8 images (mod.rs:218-234) */
        const int m = len - done < 8 ? len - done : 8;
        for (int i = 0; i < m * cells; i++) bf[i] = (float)boards[(size_t)done * cells + i];
        c6o_net_forward(net, (float)turn, bf, m, policies + (size_t)done * cells, values + done);
    }
    return 0;
}
