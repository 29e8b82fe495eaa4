/*
 * c6_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Plain-C restatement of Connect6/libconnect6 (reference, Rust); see c6_oracle.h.
 * Compile with -ffp-contract=off: the float arithmetic below mirrors the
 * reference's unfused f32/f64 operations one by one.
 */
#include "c6_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================== */
/* game                                                                   */
/* ===================================================================== */

/* game/player/mod.rs:24-30 */
int c6o_player_switch(int p) { return p == C6O_BLACK ? C6O_WHITE : (p == C6O_WHITE ? C6O_BLACK : C6O_NONE); }

/* game/game_impl/mod.rs:125-131: Black to move, one stone, empty board */
void c6o_game_new(c6o_game *g, int n) {
    memset(g, 0, sizeof(*g));
    g->n = n;
    g->turn = C6O_BLACK;
    g->num_remain = 1;
}

/* game/game_impl/mod.rs:148-169 */
int c6o_game_set(c6o_game *g, int row, int col, c6o_set_result *out) {
    int n = g->n;
    /* usize arguments in the reference: a negative index cannot occur there; treat as invalid */
    if (row < 0 || col < 0 || row >= n || col >= n) return C6O_ERR_INVALID_POSITION;
    if (g->board[row * n + col] != C6O_NONE) return C6O_ERR_ALREADY_SET;
    g->board[row * n + col] = (int8_t)g->turn;
    g->num_remain -= 1;
    if (out) { /* SetResult::with_game: snapshot after the decrement, before the reset (:158-160) */
        out->player = g->turn;
        out->num_remain = g->num_remain;
        out->row = row;
        out->col = col;
    }
    if (g->num_remain <= 0) {
        g->num_remain = 2;
        g->turn = c6o_player_switch(g->turn);
    }
    return C6O_OK;
}

/* game/game_impl/mod.rs:66-100 (Error::description strings) */
const char *c6o_strerror(int err) {
    switch (err) {
    case C6O_OK: return "ok";
    case C6O_ERR_INVALID_POSITION: return "invalid position";
    case C6O_ERR_ALREADY_SET: return "already set position";
    default: return "unknown";
    }
}

/* search_winner/mod.rs:36-184: Cumulative + Block (two rolling rows with a 1-cell halo) */
typedef struct { int c[4]; } cumul; /* right, down, right_down, left_down */
typedef struct {
    int flag;
    cumul mem[2][C6O_MAXN + 2];
} block;

static void block_update_row(block *b, int n) { /* :176-183 */
    b->flag = 1 - b->flag;
    memset(b->mem[1 - b->flag], 0, sizeof(cumul) * (size_t)(n + 2));
}

/* :206-222 path_iter: returns 1 when some direction reaches >= 6 */
static int block_path_iter(block *b, int col0) {
    int col = col0 + 1;
    cumul *prev = b->mem[b->flag], *now = b->mem[1 - b->flag];
    for (int p = 0; p < 4; p++) {
        const cumul *src; /* Block::get_prev :131-139 */
        switch (p) {
        case 0: src = &now[col - 1]; break;  /* Right */
        case 1: src = &prev[col]; break;     /* Down */
        case 2: src = &prev[col - 1]; break; /* RightDown */
        default: src = &prev[col + 1]; break;/* LeftDown */
        }
        int updated = src->c[p] + 1;
        if (updated >= 6) return 1;
        now[col].c[p] = updated;
    }
    return 0;
}

/* search_winner/mod.rs:201-246 */
int c6o_search(const int8_t *board, int n) {
    block black, white;
    memset(&black, 0, sizeof black);
    memset(&white, 0, sizeof white);
    for (int row = 0; row < n; row++) {
        block_update_row(&black, n);
        block_update_row(&white, n);
        for (int col = 0; col < n; col++) {
            int8_t v = board[row * n + col];
            if (v == C6O_NONE) continue;
            if (v == C6O_BLACK) {
                if (block_path_iter(&black, col)) return C6O_BLACK;
            } else {
                if (block_path_iter(&white, col)) return C6O_WHITE;
            }
        }
    }
    return C6O_NONE;
}

int c6o_game_is_end(const c6o_game *g) { return c6o_search(g->board, g->n); }

/* ===================================================================== */
/* simulate                                                               */
/* ===================================================================== */

/* simulate/mod.rs:120-127 + Node::from_board :58-69 (row-major empties) */
void c6o_sim_from_game(c6o_sim *s, const c6o_game *g) {
    int n = g->n;
    s->n = n;
    s->turn = g->turn;
    s->num_remain = g->num_remain;
    memcpy(s->board, g->board, (size_t)(n * n));
    s->npossible = 0;
    for (int i = 0; i < n * n; i++)
        if (g->board[i] == C6O_NONE) s->possible[s->npossible++] = (int16_t)i;
}

int c6o_sim_validate(const c6o_sim *s, int row, int col) {
    if (row < 0 || col < 0 || row >= s->n || col >= s->n) return 0;
    return s->board[row * s->n + col] == C6O_NONE;
}

int c6o_sim_next_turn(const c6o_sim *s) { return s->num_remain <= 1 ? c6o_player_switch(s->turn) : s->turn; }

static void sim_remove_possible(c6o_sim *s, int cell) {
    int i = 0;
    while (i < s->npossible && s->possible[i] != cell) i++;
    if (i == s->npossible) { fprintf(stderr, "c6o: simulate on a non-empty cell\n"); abort(); } /* unwrap() */
    memmove(&s->possible[i], &s->possible[i + 1], sizeof(int16_t) * (size_t)(s->npossible - i - 1));
    s->npossible--;
}

/* simulate/mod.rs:262-274 */
void c6o_sim_simulate_in(c6o_sim *s, int row, int col) {
    int cell = row * s->n + col;
    sim_remove_possible(s, cell);
    s->board[cell] = (int8_t)s->turn;
    s->num_remain -= 1;
    if (s->num_remain <= 0) {
        s->num_remain = 2;
        s->turn = c6o_player_switch(s->turn);
    }
}

/* simulate/mod.rs:290-301 */
void c6o_sim_rollback_in(c6o_sim *s, int row, int col) {
    int cell = row * s->n + col;
    s->board[cell] = C6O_NONE;
    s->possible[s->npossible++] = (int16_t)cell;
    s->num_remain += 1;
    if (s->num_remain > 2) {
        s->num_remain = 1;
        s->turn = c6o_player_switch(s->turn);
    }
}

/* default_policy/mod.rs:74-83 */
int c6o_diff_board(const int8_t *b1, const int8_t *b2, int n) {
    for (int i = 0; i < n * n; i++)
        if (b1[i] != b2[i]) return i;
    return -1;
}

/* ===================================================================== */
/* augment                                                                */
/* ===================================================================== */

#define DEF_AUG(T, SUF)                                                                         \
    void c6o_rotate_left_##SUF(T *b, int n) { /* augment.rs:7-15 */                              \
        T r[C6O_MAXCELLS];                                                                      \
        for (int i = 0; i < n; i++)                                                             \
            for (int j = 0; j < n; j++) r[(n - j - 1) * n + i] = b[i * n + j];                  \
        memcpy(b, r, sizeof(T) * (size_t)(n * n));                                              \
    }                                                                                           \
    void c6o_rotate_right_##SUF(T *b, int n) { /* augment.rs:17-25 */                            \
        T r[C6O_MAXCELLS];                                                                      \
        for (int i = 0; i < n; i++)                                                             \
            for (int j = 0; j < n; j++) r[j * n + (n - i - 1)] = b[i * n + j];                  \
        memcpy(b, r, sizeof(T) * (size_t)(n * n));                                              \
    }                                                                                           \
    void c6o_flip_vertical_##SUF(T *b, int n) { /* augment.rs:27-35, axis | */                   \
        for (int i = 0; i < n; i++)                                                             \
            for (int j = 0; j < n / 2; j++) {                                                   \
                T t = b[i * n + (n - j - 1)];                                                   \
                b[i * n + (n - j - 1)] = b[i * n + j];                                          \
                b[i * n + j] = t;                                                               \
            }                                                                                   \
    }
DEF_AUG(int8_t, i8)
DEF_AUG(float, f32)

/* augment.rs:37-47, axis -- (unused on the path, kept for the known-answer test) */
void c6o_flip_horizontal_i8(int8_t *b, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n / 2; j++) {
            int8_t t = b[(n - j - 1) * n + i];
            b[(n - j - 1) * n + i] = b[j * n + i];
            b[j * n + i] = t;
        }
}

/* augment.rs:60-73 */
void c6o_augment_way8(const int8_t *board, int n, int8_t *out8) {
    int8_t b[C6O_MAXCELLS];
    size_t sz = (size_t)(n * n);
    memcpy(b, board, sz);
    for (int i = 0; i < 4; i++) {
        c6o_rotate_left_i8(b, n);
        memcpy(out8 + (size_t)(2 * i) * sz, b, sz);
        memcpy(out8 + (size_t)(2 * i + 1) * sz, b, sz);
        c6o_flip_vertical_i8(out8 + (size_t)(2 * i + 1) * sz, n);
    }
}

/* augment.rs:75-92; summation order kept exactly */
void c6o_recover_way8(float *probs8, int n, float *out) {
    size_t sz = (size_t)(n * n);
    for (size_t k = 0; k < sz; k++) out[k] = 0.f;
    for (int i = 0; i < 4; i++) {
        float *a = probs8 + (size_t)(2 * i) * sz, *f = probs8 + (size_t)(2 * i + 1) * sz;
        c6o_flip_vertical_f32(f, n);
        for (size_t k = 0; k < sz; k++) a[k] = a[k] + f[k];
        for (int r = 0; r < i + 1; r++) c6o_rotate_right_f32(a, n);
        for (size_t k = 0; k < sz; k++) out[k] = out[k] + a[k];
    }
    for (size_t k = 0; k < sz; k++) out[k] /= 8.f;
}

/* ===================================================================== */
/* C6RNG (spec in DESIGN.md): Philox4x32-10 / -7 + deterministic f32 math   */
/* ===================================================================== */

static void philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void c6o_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32(ctr, key, out, 10); }
/* C6RNG v5: the Dirichlet-noise draws (tag 1) -- one block per pair of attempts and empty cell at every tree level, the bulk of all draws -- use
 * the seven-round form (Salmon et al. 2011: Philox4x32-7 is the fewest rounds that pass BigCrush; 10 is the margin the other tags keep) */
void c6o_philox4x32_7(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32(ctr, key, out, 7); }

static inline double bits2d(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t d2bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* 53-bit uniform strictly inside (0,1) */
static inline double open01(uint32_t hi, uint32_t lo) {
    uint64_t v = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

static inline float bits2f(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* C6RNG v4: log and exp of the Gamma sampler -- polynomial forms (Cephes' logf / expf coefficients) on explicitly FUSED multiply-adds
 * (fmaf: one rounding; this file is compiled with -mfma and -ffp-contract=off, so exactly the fmaf written here are fused) and IEEE
 * + - *: the same operations, in the same order, as c6_log4 / c6_exp4 of the kernels (c6_device.h), hence the same bits. */
float c6o_log4(float x) {
    uint32_t b = f2bits(x);
    int e = (int)((b >> 23) & 0xff) - 127;
    float m = bits2f((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float fe = (float)e, y = m - 1.0f, z = y * y;
    float p = 7.0376836292e-2f;
    p = fmaf(p, y, -1.1514610310e-1f);
    p = fmaf(p, y, 1.1676998740e-1f);
    p = fmaf(p, y, -1.2420140846e-1f);
    p = fmaf(p, y, 1.4249322787e-1f);
    p = fmaf(p, y, -1.6668057665e-1f);
    p = fmaf(p, y, 2.0000714765e-1f);
    p = fmaf(p, y, -2.4999993993e-1f);
    p = fmaf(p, y, 3.3333331174e-1f);
    p = (p * y) * z;
    float r = fmaf(fe, -2.12194440e-4f, p);
    r = fmaf(z, -0.5f, r);
    r = y + r;
    return fmaf(fe, 0.693359375f, r);
}

/* exp(t) = p * 2^k, p in about [0.7, 1.42]; the exponent stays an integer (U^(1/alpha) reaches 2^-830 at alpha = 0.03) */
float c6o_exp4(float t, int *kout) {
    const float tt = t * 1.44269504f;
    const int k = (int)(tt + (tt >= 0.0f ? 0.5f : -0.5f));
    const float fk = (float)k;
    float r = fmaf(fk, -0.693359375f, t);
    r = fmaf(fk, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, z, r);
    *kout = k;
    return p + 1.0f;
}

/* 24-bit uniform strictly inside (0,1) */
static inline float open01f(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }

/* m * 2^k, exact for a normal m and a normal result */
static inline float scale2f(float m, int k) { return m * bits2f((uint32_t)(k + 127) << 23); }

/* One attempt of the alpha < 1 sampler on the uniforms (u1, u2); 1 and the variate m 2^k if it is accepted.  C6RNG v5 ("GS with a Pareto tail"):
 * rejection from the mixture of q1(x) = alpha x^(alpha-1) on (0, 1] and q2(x) = beta x^(-beta-1) on (1, inf), beta = 1 - alpha, with the weights
 * 1 : alpha / (e beta) -- then target / proposal is proportional to e^-x on the body and to x e^(1-x) on the tail, both <= 1:
 *     p = b u1, b = 1 + alpha / (e beta), E = -log(u2) (an Exp(1) variate)
 *     p <= 1 (98.9 %):  x = p^(1/alpha)  = exp(log(p) / alpha),              accepted iff x <= E                (<=> u2 <= e^-x)
 *     p >  1:           x = u'^(-1/beta) = exp(-log(u') / beta), u' = (b - p) e beta / alpha,
 *                                                                           accepted iff E >= (x - 1) - log(x)  (<=> u2 <= x e^(1-x))
 * Both branches are ONE sequence -- log of the branch's argument, times the branch's constant, exp, one comparison against E -- where v4's
 * Ahrens-Dieter tail (x = -log((b - p) / alpha), accepted iff E >= (1 - alpha) log x) needed a second logarithm that a wave executed whenever one
 * of its ~94 attempts fell into the tail.  Acceptance 0.972 (GS: 0.973). */
static int gs_attempt(float ia, float ib, float b, float ct, float u1, float u2, float *m_out, int *k_out) {
    const float p = b * u1;
    const int tail = !(p <= 1.0f);
    const float arg = tail ? (b - p) * ct : p;
    if (!(arg > 0.0f)) return 0;      /* p rounded up to b */
    const float L = c6o_log4(arg), E = 0.0f - c6o_log4(u2);
    const float t = L * (tail ? ib : ia);
    int k;
    const float m = c6o_exp4(t, &k);
    if (!tail) {
        if (!(k < -30 || scale2f(m, k) <= E)) return 0;      /* below 2^-30 every E (>= 2^-25) accepts */
    } else if (!(E >= (scale2f(m, k) - 1.0f) - t)) return 0;  /* (k <= 18: u' >= 2^-18) */
    *m_out = m;
    *k_out = k;
    return 1;
}

/*
 * Gamma(alpha, 1): what `Dirichlet::new_with_param(alpha, n).sample` draws per component
 * (rand 0.6, called from alphazero_policy/mod.rs:262,283).  The reference's generator is unseeded and rand's own
 * stream cannot be reproduced, so only the DISTRIBUTION is the reference's; the algorithm is C6RNG's (v5), in single
 * precision (the reference casts the noise to f32 before use, mod.rs:283); a variate is m * 2^k:
 *   alpha < 1:  gs_attempt above.  Attempts 2i and 2i + 1 share the Philox block with counter word c1 = i (words 0, 1 and 2, 3).
 *   alpha >= 1: Marsaglia-Tsang for shape alpha+1 times U^(1/alpha) (rand's GammaSmallShape form) with polar-method
 *               normals.  Words: polar v1, polar v2, squeeze u, power u.
 * Every block of these draws is Philox4x32-7 (c6o_philox4x32_7).
 */
void c6o_gamma_draw4(double alpha, uint32_t k0, uint32_t k1, uint32_t cell, uint32_t search, uint32_t depth, float *m_out, int *k_out) {
    const double shape = alpha + 1.0;
    const double dd = shape - 1.0 / 3.0;
    const double cd = 1.0 / sqrt(9.0 * dd);
    const float d = (float)dd, c = (float)cd, inv_alpha = 1.0f / (float)alpha;
    /* the launch's constants of gs_attempt (IEEE single precision, as the kernels compute them: c6_gs_consts, c6_device.h) */
    const float a = (float)alpha, beta = 1.0f - a, eb = 2.7182817f * beta, b = 1.0f + a / eb, ct = eb / a, ib = -1.0f / beta;
    uint32_t key[2] = {k0, k1};
    for (uint32_t att = 0;; att++) {
        /* alpha < 1: one Philox block serves the attempts 2i (words 0, 1) and 2i + 1 (words 2, 3) */
        uint32_t ctr[4] = {cell, alpha < 1.0 ? att >> 1 : att, search, (uint32_t)C6O_TAG_NOISE | (depth << 8)}, w[4];
        c6o_philox4x32_7(ctr, key, w);
        if (alpha < 1.0) {
            if (gs_attempt(inv_alpha, ib, b, ct, open01f(w[2 * (att & 1)]), open01f(w[2 * (att & 1) + 1]), m_out, k_out)) return;
            continue;
        }
        const float v1 = 2.0f * open01f(w[0]) - 1.0f, v2 = 2.0f * open01f(w[1]) - 1.0f;
        const float s = v1 * v1 + v2 * v2;
        if (s >= 1.0f || s == 0.0f) continue;
        const float x = v1 * sqrtf(-2.0f * c6o_log4(s) / s);
        const float vc = 1.0f + c * x;
        if (vc <= 0.0f) continue;
        const float v = vc * vc * vc;
        const float u = open01f(w[2]), u2 = open01f(w[3]);
        const float x2 = x * x;
        if (u < 1.0f - 0.0331f * x2 * x2 || c6o_log4(u) < 0.5f * x2 + d * (1.0f - v + c6o_log4(v))) {
            int k;
            *m_out = (d * v) * c6o_exp4(c6o_log4(u2) * inv_alpha, &k);
            *k_out = k;
            return;
        }
    }
}

/* the variate as a double (distribution tests) */
double c6o_gamma_draw(double alpha, uint32_t k0, uint32_t k1, uint32_t cell, uint32_t search, uint32_t depth) {
    float m;
    int k;
    c6o_gamma_draw4(alpha, k0, k1, cell, search, depth, &m, &k);
    return ldexp((double)m, k);
}

/* the wave's summation tree: xor-butterfly with ascending offsets 1 .. 32 over the 64 lane partial sums (part of the C6RNG spec) */
float c6o_butterfly_sum64f(const float part[64]) {
    float v[64], t[64];
    memcpy(v, part, sizeof v);
    for (int off = 1; off <= 32; off <<= 1) {
        for (int l = 0; l < 64; l++) t[l] = v[l] + v[l ^ off];
        memcpy(v, t, sizeof v);
    }
    return v[0];
}

/* Dirichlet noise of the empty cells (C6RNG v4): Gamma variates m 2^k scaled by 2^-kmax (kmax: the largest exponent of the vector; values
 * more than 2^100 below it are 0), lane partial sums (cell c belongs to lane c & 63, ascending c), the butterfly, noise = scaled * (1 / sum) */
void c6o_dirichlet_noise(double alpha, uint32_t k0, uint32_t k1, uint32_t search, uint32_t depth,
                         const int8_t *board, int n, float *noise) {
    float m[C6O_MAXCELLS], part[64];
    int k[C6O_MAXCELLS], kmax = -(1 << 30);
    int cells = n * n;
    for (int c = 0; c < cells; c++) {
        m[c] = 0.f;
        k[c] = -(1 << 30);
        if (board[c] == C6O_NONE) {
            c6o_gamma_draw4(alpha, k0, k1, (uint32_t)c, search, depth, &m[c], &k[c]);
            if (k[c] > kmax) kmax = k[c];
        }
    }
    for (int l = 0; l < 64; l++) part[l] = 0.f;
    for (int c = 0; c < cells; c++) {
        const int dk = k[c] - kmax;
        noise[c] = board[c] == C6O_NONE && dk >= -100 ? scale2f(m[c], dk) : 0.f;
        part[c & 63] = part[c & 63] + noise[c]; /* ascending within a lane: c, c+64, ... (cells that need none add 0) */
    }
    const float rs = 1.0f / c6o_butterfly_sum64f(part);
    for (int c = 0; c < cells; c++) noise[c] = noise[c] * rs;
}

/* ===================================================================== */
/* AlphaZero policy                                                       */
/* ===================================================================== */

/* alphazero_policy/mod.rs:26-36 */
typedef struct node {
    uint64_t h;
    int visit;
    float value, q_sum, q_value, n_prob;
    float *prob; /* NULL == all zeros (never expanded) */
    int num_player;
    int n_next;
    struct node **next; /* children in `possible` order; the reference stores their hashes */
    int16_t *next_cell;
    int8_t board[C6O_MAXCELLS];
} node;

struct c6o_az {
    int n, cells;
    c6o_hyper hp;
    int eval_kind;
    c6o_eval_fn fn;
    void *ctx;
    uint32_t k0, k1;
    uint32_t nsearch;    /* C6RNG `search` coordinate: searches since the game started */
    int root_level;      /* stones on the board at the last init() */
    uint64_t zob[2][C6O_MAXCELLS];
    node **tab;          /* open addressing, pointers */
    size_t cap, len;
    uint64_t depth_sum, depth_cnt;
};

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* stands in for alphazero_policy/mod.rs:456-460 (SipHash of the board): only used as a
 * table key; full boards are compared on every hit, so there are no collisions at all */
static uint64_t az_hash(const c6o_az *az, const int8_t *board) {
    uint64_t h = 0x1234567ULL;
    for (int i = 0; i < az->cells; i++)
        if (board[i] != C6O_NONE) h ^= az->zob[board[i] == C6O_WHITE][i];
    return h;
}

static node *map_get(const c6o_az *az, uint64_t h, const int8_t *board) {
    size_t mask = az->cap - 1, i = (size_t)h & mask;
    while (az->tab[i]) {
        node *nd = az->tab[i];
        if (nd->h == h && memcmp(nd->board, board, (size_t)az->cells) == 0) return nd;
        i = (i + 1) & mask;
    }
    return NULL;
}

static void map_put_raw(node **tab, size_t cap, node *nd) {
    size_t mask = cap - 1, i = (size_t)nd->h & mask;
    while (tab[i]) i = (i + 1) & mask;
    tab[i] = nd;
}

static void map_insert(c6o_az *az, node *nd) {
    if ((az->len + 1) * 2 > az->cap) {
        size_t ncap = az->cap * 2;
        node **nt = (node **)calloc(ncap, sizeof(node *));
        for (size_t i = 0; i < az->cap; i++)
            if (az->tab[i]) map_put_raw(nt, ncap, az->tab[i]);
        free(az->tab);
        az->tab = nt;
        az->cap = ncap;
    }
    map_put_raw(az->tab, az->cap, nd);
    az->len++;
}

static void node_free(node *nd) {
    free(nd->prob);
    free(nd->next);
    free(nd->next_cell);
    free(nd);
}

/* Node::new_with_num :62-74 */
static node *node_new(const c6o_az *az, const int8_t *board, uint64_t h, int num_player) {
    node *nd = (node *)calloc(1, sizeof(node));
    nd->h = h;
    nd->num_player = num_player;
    memcpy(nd->board, board, (size_t)az->cells);
    return nd;
}

static int count_stones(const int8_t *b, int cells) {
    int c = 0;
    for (int i = 0; i < cells; i++) c += b[i] != C6O_NONE;
    return c;
}

static int random_eval(void *ctx, int turn, int n, int len, const int8_t *boards, float *values, float *policies);
static int uniform_eval(void *ctx, int turn, int n, int len, const int8_t *boards, float *values, float *policies);

c6o_az *c6o_az_new(int n, const c6o_hyper *hp, int eval_kind, c6o_eval_fn fn, void *ctx, uint64_t seed,
                   uint32_t game_uid) {
    c6o_az *az = (c6o_az *)calloc(1, sizeof(*az));
    az->n = n;
    az->cells = n * n;
    az->hp = *hp;
    az->eval_kind = eval_kind;
    az->fn = fn;
    az->ctx = ctx;
    if (eval_kind == C6O_EVAL_RANDOM) { az->fn = random_eval; az->ctx = az; }
    if (eval_kind == C6O_EVAL_UNIFORM) { az->fn = uniform_eval; az->ctx = az; }
    az->k0 = (uint32_t)seed;
    az->k1 = (uint32_t)(seed >> 32) + game_uid;
    uint64_t s = 0xC6C6C6C6ULL;
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < C6O_MAXCELLS; i++) az->zob[c][i] = splitmix64(&s);
    az->cap = 1024;
    az->tab = (node **)calloc(az->cap, sizeof(node *));
    return az;
}

void c6o_az_free(c6o_az *az) {
    if (!az) return;
    for (size_t i = 0; i < az->cap; i++)
        if (az->tab[i]) node_free(az->tab[i]);
    free(az->tab);
    free(az);
}

size_t c6o_az_map_len(const c6o_az *az) { return az->len; }
uint32_t c6o_az_search_count(const c6o_az *az) { return az->nsearch; }
double c6o_az_mean_depth(const c6o_az *az) { return az->depth_cnt ? (double)az->depth_sum / (double)az->depth_cnt : 0.0; }

/* RandomEvaluator, alphazero_policy/mod.rs:137-169: value and every policy cell ~ U[-1,1) */
static inline float u11(uint32_t w) { return 2.0f * ((float)(w >> 8) * (1.0f / 16777216.0f)) - 1.0f; }

static int random_eval(void *ctx, int turn, int n, int len, const int8_t *boards, float *values, float *policies) {
    (void)turn; (void)boards;
    c6o_az *az = (c6o_az *)ctx;
    uint32_t key[2] = {az->k0, az->k1}, w[4];
    int cells = n * n;
    for (int k = 0; k < len; k++) {
        uint32_t cv[4] = {(uint32_t)k, 0, az->nsearch, C6O_TAG_EVAL_V};
        c6o_philox4x32_10(cv, key, w);
        values[k] = u11(w[0]);
        for (int c = 0; c < cells; c++) {
            if ((c & 3) == 0) {
                uint32_t cp[4] = {(uint32_t)(c >> 2), (uint32_t)k, az->nsearch, C6O_TAG_EVAL_P};
                c6o_philox4x32_10(cp, key, w);
            }
            policies[k * cells + c] = u11(w[c & 3]);
        }
    }
    return 0;
}

static int uniform_eval(void *ctx, int turn, int n, int len, const int8_t *boards, float *values, float *policies) {
    (void)ctx; (void)turn; (void)boards;
    int cells = n * n;
    for (int k = 0; k < len; k++) {
        values[k] = 0.f;
        for (int c = 0; c < cells; c++) policies[k * cells + c] = 1.0f / (float)cells;
    }
    return 0;
}

/* alphazero_policy/mod.rs:218-234 */
static int az_get_from(c6o_az *az, int turn, const int8_t *board, float *value, float *prob) {
    int8_t aug[8 * C6O_MAXCELLS];
    float vals[8], pols[8 * C6O_MAXCELLS];
    c6o_augment_way8(board, az->n, aug);
    memset(vals, 0, sizeof vals);
    /* cppeval/mod.rs:34-35: on entry `policies` holds the boards as floats */
    for (int i = 0; i < 8 * az->cells; i++) pols[i] = (float)aug[i];
    if (az->fn(az->ctx, turn, az->n, 8, aug, vals, pols) != 0) return -1;
    float s = 0.f;
    for (int k = 0; k < 8; k++) s = s + vals[k]; /* iter().sum::<f32>() */
    *value = s / 8.f;
    c6o_recover_way8(pols, az->n, prob);
    for (int i = 0; i < az->cells; i++) prob[i] *= (float)(board[i] == C6O_NONE); /* masking :226-231 */
    return 0;
}

/* alphazero_policy/mod.rs:302-306 */
void c6o_az_init(c6o_az *az, const c6o_sim *s) {
    uint64_t h = az_hash(az, s->board);
    az->root_level = count_stones(s->board, az->cells);
    if (!map_get(az, h, s->board)) map_insert(az, node_new(az, s->board, h, az->root_level)); /* Node::new :43-59 */
}

/* alphazero_policy/mod.rs:243-296; returns the chosen child */
static node *az_maximum_from(c6o_az *az, const c6o_sim *s) {
    node *tn = map_get(az, az_hash(az, s->board), s->board); /* unwrap() */
    if (!tn) { fprintf(stderr, "c6o: maximum_from on unknown node\n"); abort(); }
    int nc = tn->n_next;
    if (nc == 0) return NULL;
    if (nc == 1) return tn->next[0]; /* heuristic :255-258 */

    float epsilon = az->hp.epsilon, c_puct = az->hp.c_puct;
    uint32_t depth = (uint32_t)(count_stones(s->board, az->cells) - az->root_level);
    float noise[C6O_MAXCELLS];
    c6o_dirichlet_noise(az->hp.dirichlet_alpha, az->k0, az->k1, az->nsearch, depth, s->board, az->n, noise);

    int vs = 0;
    for (int i = 0; i < nc; i++) vs += tn->next[i]->visit;
    float visit_sum = (float)vs;
    if (s->turn == C6O_NONE) { fprintf(stderr, "c6o: maximum_from with player none\n"); abort(); }

    float best = 0.f, score[C6O_MAXCELLS];
    int have = 0;
    for (int i = 0; i < nc; i++) {
        const node *ch = tn->next[i];
        float nz = noise[tn->next_cell[i]];
        float visit = (float)ch->visit;
        float prob = epsilon * nz + (1.f - epsilon) * ch->n_prob;
        float puct = prob * sqrtf(visit_sum - visit) / (1.f + visit);
        float q = s->turn == C6O_BLACK ? -ch->q_value : ch->q_value;
        float sc = q + c_puct * puct;
        if (sc != sc) { fprintf(stderr, "c6o: NaN score (partial_cmp unwrap)\n"); abort(); }
        score[i] = sc;
        if (!have || sc > best) { best = sc; have = 1; }
    }
    int cnt = 0;
    for (int i = 0; i < nc; i++) cnt += score[i] == best;
    int pick = 0;
    if (cnt > 1) { /* `choose` among exact ties :291-295 -- uniform on the C6RNG TIE stream */
        uint32_t key[2] = {az->k0, az->k1}, w[4];
        uint32_t ctr[4] = {0, 0, az->nsearch, (uint32_t)C6O_TAG_TIE | (depth << 8)};
        c6o_philox4x32_10(ctr, key, w);
        pick = (int)(open01(w[0], w[1]) * (double)cnt);
        if (pick >= cnt) pick = cnt - 1;
    }
    for (int i = 0; i < nc; i++)
        if (score[i] == best && pick-- == 0) return tn->next[i];
    return NULL;
}

/* alphazero_policy/mod.rs:313-323 */
int c6o_az_select(c6o_az *az, const c6o_sim *s) {
    node *tn = map_get(az, az_hash(az, s->board), s->board);
    node *ch = az_maximum_from(az, s);
    if (!ch) return -1;
    return c6o_diff_board(ch->board, tn->board, az->n);
}

/* alphazero_policy/mod.rs:331-381 */
void c6o_az_expand(c6o_az *az, const c6o_sim *s) {
    node *parent = map_get(az, az_hash(az, s->board), s->board);
    if (!parent) { fprintf(stderr, "c6o: expand on unknown node\n"); abort(); }
    float cost = (float)s->turn;
    if (parent->num_player == az->cells) {
        int winner = c6o_search(s->board, az->n);
        if (winner == s->turn) parent->value = cost;
        else if (winner != C6O_NONE) parent->value = -cost;
        return;
    }
    float value, prob[C6O_MAXCELLS];
    if (az_get_from(az, s->turn, s->board, &value, prob) != 0) {
        fprintf(stderr, "alpha_zero::expand couldn't get value, prob\n");
        abort();
    }
    parent->value = value;
    if (!parent->prob) parent->prob = (float *)malloc(sizeof(float) * (size_t)az->cells);
    memcpy(parent->prob, prob, sizeof(float) * (size_t)az->cells);
    parent->visit += 1;
    int child_num = parent->num_player + 1;

    int np = s->npossible;
    node **kids = (node **)malloc(sizeof(node *) * (size_t)(parent->n_next + np));
    int16_t *kcell = (int16_t *)malloc(sizeof(int16_t) * (size_t)(parent->n_next + np));
    if (parent->n_next) { /* next_node.push appends (:376-378) */
        memcpy(kids, parent->next, sizeof(node *) * (size_t)parent->n_next);
        memcpy(kcell, parent->next_cell, sizeof(int16_t) * (size_t)parent->n_next);
    }
    int8_t b[C6O_MAXCELLS];
    memcpy(b, s->board, (size_t)az->cells);
    uint64_t ph = parent->h;
    for (int i = 0; i < np; i++) {
        int cell = s->possible[i];
        b[cell] = (int8_t)s->turn; /* sim.simulate(row, col) :361 */
        uint64_t h = ph ^ az->zob[s->turn == C6O_WHITE][cell];
        node *ch = map_get(az, h, b);
        if (!ch) { ch = node_new(az, b, h, child_num); map_insert(az, ch); } /* entry().or_insert :366-369 */
        ch->n_prob = prob[cell];                                              /* overwrite :370 */
        kids[parent->n_next + i] = ch;
        kcell[parent->n_next + i] = (int16_t)cell;
        b[cell] = C6O_NONE;
    }
    free(parent->next);
    free(parent->next_cell);
    parent->next = kids;
    parent->next_cell = kcell;
    parent->n_next += np;
}

/* alphazero_policy/mod.rs:386-408 */
void c6o_az_update(c6o_az *az, const c6o_sim *s, const int16_t *path, int len) {
    if (len == 0) return;
    node *leaf = map_get(az, az_hash(az, s->board), s->board);
    float value = leaf->value;
    c6o_sim c = *s; /* deep_clone */
    int n = az->n;
    {
        c6o_sim c2 = *s;
        c6o_sim_rollback_in(&c2, path[len - 1] / n, path[len - 1] % n);
        node *p = map_get(az, az_hash(az, c2.board), c2.board);
        p->q_sum += value;
    }
    for (int i = len - 1; i >= 0; i--) {
        c6o_sim_rollback_in(&c, path[i] / n, path[i] % n);
        node *p = map_get(az, az_hash(az, c.board), c.board);
        p->visit += 1;
        p->q_value = p->q_sum / (float)p->visit; /* recalc_q :77-79 */
    }
}

/* alphazero_policy/mod.rs:411-426 */
void c6o_az_search(c6o_az *az, const c6o_sim *root) {
    c6o_az_init(az, root);
    c6o_sim s = *root; /* deep_clone (simulate/mod.rs:133-143): `possible` rebuilt row-major */
    s.npossible = 0;
    for (int i = 0; i < az->cells; i++)
        if (s.board[i] == C6O_NONE) s.possible[s.npossible++] = (int16_t)i;
    int16_t path[C6O_MAXCELLS];
    int len = 0, cell;
    while ((cell = c6o_az_select(az, &s)) >= 0) {
        path[len++] = (int16_t)cell;
        c6o_sim_simulate_in(&s, cell / az->n, cell % az->n);
    }
    az->depth_sum += (uint64_t)len;
    az->depth_cnt += 1;
    c6o_az_expand(az, &s);
    c6o_az_update(az, &s, path, len);
    az->nsearch += 1;
}

/* alphazero_policy/mod.rs:433-452 */
int c6o_az_policy(c6o_az *az, const c6o_sim *s) {
    node *tn = map_get(az, az_hash(az, s->board), s->board);
    if (!tn || tn->n_next == 0) return -1;
    int vs = 0;
    for (int i = 0; i < tn->n_next; i++) vs += tn->next[i]->visit;
    float visit_sum = (float)vs;
    int best = -1;
    float bp = 0.f;
    for (int i = 0; i < tn->n_next; i++) {
        float visit = (float)tn->next[i]->visit;
        float p = visit / (visit_sum - visit + 1.f);
        if (best < 0 || !(p < bp)) { best = i; bp = p; } /* Iterator::max_by keeps the LAST maximum */
    }
    return c6o_diff_board(s->board, tn->next[best]->board, az->n);
}

/* alphazero_policy/mod.rs:473-487: drop every node with the root's stone count, re-insert the root */
void c6o_az_prune(c6o_az *az, const c6o_sim *s) {
    node *root = map_get(az, az_hash(az, s->board), s->board);
    int num_player = root->num_player;
    node **nt = (node **)calloc(az->cap, sizeof(node *));
    size_t nlen = 0;
    for (size_t i = 0; i < az->cap; i++) {
        node *nd = az->tab[i];
        if (!nd) continue;
        if (nd->num_player == num_player && nd != root) { node_free(nd); continue; }
        map_put_raw(nt, az->cap, nd);
        nlen++;
    }
    free(az->tab);
    az->tab = nt;
    az->len = nlen;
}

/* alphazero_policy/mod.rs:464-489 */
int c6o_az_next(c6o_az *az, const c6o_game *g) {
    c6o_sim s;
    c6o_sim_from_game(&s, g);
    c6o_az_init(az, &s);
    for (int i = 0; i < az->hp.num_simulation; i++) c6o_az_search(az, &s);
    int res = c6o_az_policy(az, &s);
    c6o_az_prune(az, &s);
    return res;
}

void c6o_az_node_info(const c6o_az *az, const int8_t *board, c6o_node_info *out) {
    memset(out, 0, sizeof *out);
    const node *nd = map_get(az, az_hash(az, board), board);
    if (!nd) return;
    out->found = 1;
    out->visit = nd->visit;
    out->value = nd->value;
    out->q_sum = nd->q_sum;
    out->q_value = nd->q_value;
    out->n_prob = nd->n_prob;
    out->num_player = nd->num_player;
    out->n_next = nd->n_next;
    out->has_prob = nd->prob != NULL;
}

int c6o_az_node_prob(const c6o_az *az, const int8_t *board, float *prob) {
    const node *nd = map_get(az, az_hash(az, board), board);
    if (!nd) return 0;
    for (int i = 0; i < az->cells; i++) prob[i] = nd->prob ? nd->prob[i] : 0.f;
    return 1;
}

int c6o_az_children(const c6o_az *az, const int8_t *board, int *visit, float *q_value, float *n_prob) {
    const node *nd = map_get(az, az_hash(az, board), board);
    for (int i = 0; i < az->cells; i++) { visit[i] = -1; q_value[i] = 0.f; n_prob[i] = 0.f; }
    if (!nd) return 0;
    for (int i = 0; i < nd->n_next; i++) {
        int c = nd->next_cell[i];
        visit[c] = nd->next[i]->visit;
        q_value[c] = nd->next[i]->q_value;
        n_prob[c] = nd->next[i]->n_prob;
    }
    return 1;
}

/* ===================================================================== */
/* agent                                                                  */
/* ===================================================================== */

/* agent/agent_impl/mod.rs:121-178 */
void c6o_agent_play(int n, c6o_policy_fn fn, void *ctx, c6o_play_result *out) {
    c6o_game g;
    c6o_game_new(&g, n);
    int cap = 64;
    out->winner = C6O_NONE;
    out->len = 0;
    out->err = C6O_OK;
    out->pi = NULL;
    out->path = (c6o_path *)malloc(sizeof(c6o_path) * (size_t)cap);
    for (;;) {
        int row = -1, col = -1;
        if (!fn(ctx, &g, &row, &col)) break; /* policy couldn't generate a selection :135-137 */
        if (out->len == cap) { cap *= 2; out->path = (c6o_path *)realloc(out->path, sizeof(c6o_path) * (size_t)cap); }
        c6o_path *p = &out->path[out->len++];
        p->turn = g.turn;
        p->row = row;
        p->col = col;
        memcpy(p->board, g.board, sizeof g.board);
        int err = c6o_game_set(&g, p->row, p->col, NULL);
        if (err != C6O_OK) { out->err = err; return; } /* Err(err) => return Err(err) :163 */
        int w = c6o_game_is_end(&g);
        if (w != C6O_NONE) { out->winner = w; break; }
    }
}

void c6o_play_result_free(c6o_play_result *r) {
    free(r->path);
    free(r->pi);
    r->path = NULL;
    r->pi = NULL;
}

typedef struct {
    c6o_az *az;
    int want_pi;
    int32_t *pi;
    int pi_cap, pi_len, cells;
} az_policy_ctx;

static int az_policy_fn(void *ctx, const c6o_game *g, int *row, int *col) {
    az_policy_ctx *pc = (az_policy_ctx *)ctx;
    if (!pc->want_pi) {
        int cell = c6o_az_next(pc->az, g);
        if (cell < 0) return 0;
        *row = cell / g->n;
        *col = cell % g->n;
        return 1;
    }
    /* next() with the root child visit counts captured before the prune */
    c6o_az *az = pc->az;
    c6o_sim s;
    c6o_sim_from_game(&s, g);
    c6o_az_init(az, &s);
    for (int i = 0; i < az->hp.num_simulation; i++) c6o_az_search(az, &s);
    int res = c6o_az_policy(az, &s);
    if (res >= 0) {
        if (pc->pi_len == pc->pi_cap) {
            pc->pi_cap = pc->pi_cap ? pc->pi_cap * 2 : 64;
            pc->pi = (int32_t *)realloc(pc->pi, sizeof(int32_t) * (size_t)pc->pi_cap * (size_t)pc->cells);
        }
        int32_t *dst = pc->pi + (size_t)pc->pi_len * (size_t)pc->cells;
        float q[C6O_MAXCELLS], np[C6O_MAXCELLS];
        int vis[C6O_MAXCELLS];
        c6o_az_children(az, s.board, vis, q, np);
        for (int i = 0; i < pc->cells; i++) dst[i] = vis[i] < 0 ? 0 : vis[i];
        pc->pi_len++;
    }
    c6o_az_prune(az, &s);
    if (res < 0) return 0;
    *row = res / g->n;
    *col = res % g->n;
    return 1;
}

typedef struct {
    int n;
    const c6o_hyper *hp;
    int eval_kind;
    c6o_eval_fn fn;
    void *ctx;
    uint64_t seed;
    uint32_t uid_base;
    int num, want_pi;
    c6o_play_result *out;
    int next;
    pthread_mutex_t mu;
} sp_job;

static void *sp_worker(void *arg) {
    sp_job *job = (sp_job *)arg;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        int id = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (id >= job->num) break;
        /* one policy per game: policy_gen, async_agent/mod.rs:105 */
        c6o_az *az = c6o_az_new(job->n, job->hp, job->eval_kind, job->fn, job->ctx, job->seed, job->uid_base + (uint32_t)id);
        az_policy_ctx pc = {az, job->want_pi, NULL, 0, 0, job->n * job->n};
        c6o_agent_play(job->n, az_policy_fn, &pc, &job->out[id]);
        job->out[id].pi = pc.pi;
        c6o_az_free(az);
    }
    return NULL;
}

/* agent/async_agent/mod.rs:100-137: `num` independent games on a pool of worker threads */
void c6o_self_play(int n, const c6o_hyper *hp, int eval_kind, c6o_eval_fn fn, void *ctx, uint64_t seed,
                   uint32_t uid_base, int num, int threads, int want_pi, c6o_play_result *out) {
    sp_job job = {n, hp, eval_kind, fn, ctx, seed, uid_base, num, want_pi, out, 0, PTHREAD_MUTEX_INITIALIZER};
    if (threads < 1) threads = 1;
    if (threads > num) threads = num;
    if (threads == 1) { sp_worker(&job); return; }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, sp_worker, &job);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th);
}

/* ===================================================================== */
/* replay buffer (AlphaZero/cpp_weighted/ReplayBuffer.hpp)                */
/* ===================================================================== */

typedef struct { int winner; c6o_path path; } replay_entry;
struct c6o_replay {
    int n;
    size_t max_size, tail, num_data;
    replay_entry *paths;
};

c6o_replay *c6o_replay_new(int n, size_t max_size) {
    c6o_replay *r = (c6o_replay *)calloc(1, sizeof *r);
    r->n = n;
    r->max_size = max_size;
    r->paths = (replay_entry *)calloc(max_size, sizeof(replay_entry));
    return r;
}

void c6o_replay_free(c6o_replay *r) {
    if (!r) return;
    free(r->paths);
    free(r);
}

/* ReplayBuffer.hpp:22-29 */
void c6o_replay_push_game(c6o_replay *r, int winner, const c6o_path *path, int len) {
    r->num_data = r->num_data + (size_t)len < r->max_size ? r->num_data + (size_t)len : r->max_size;
    for (int i = 0; i < len; i++) {
        r->paths[r->tail].winner = winner;
        r->paths[r->tail].path = path[i];
        r->tail = (r->tail + 1) % r->max_size;
    }
}

void c6o_replay_clear(c6o_replay *r) { r->num_data = r->tail = 0; }
size_t c6o_replay_size(const c6o_replay *r) { return r->num_data; }
size_t c6o_replay_tail(const c6o_replay *r) { return r->tail; }

/* ReplayBuffer.hpp:39-65 */
void c6o_replay_sample(const c6o_replay *r, int num_sample, uint64_t seed, uint32_t draw_id, float *winners, float *players,
                       float *boards, float *poses) {
    const int cells = r->n * r->n;
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int i = 0; i < num_sample; i++) {
        uint32_t ctr[4] = {(uint32_t)i, draw_id, 0, C6O_TAG_SAMPLE}, w[4];
        c6o_philox4x32_10(ctr, key, w);
        const replay_entry *e = &r->paths[w[0] % r->num_data];   /* gen() % num_data */
        winners[i] = (float)e->winner;
        players[i] = (float)e->path.turn;
        for (int c = 0; c < cells; c++) boards[(size_t)i * cells + c] = (float)e->path.board[c];
        poses[i] = (float)(e->path.row * r->n + e->path.col);
    }
}
