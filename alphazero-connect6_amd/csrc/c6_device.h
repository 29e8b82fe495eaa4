// c6_device.h -- device-side building blocks shared by the HIP kernels (gfx950 only).
//
//  * C6RNG: the counter-based random stream of DESIGN.md (Philox4x32-10 keyed by
//    (seed, game uid); counter = (cell, attempt, search, tag | depth << 8)) and the
//    Gamma(alpha < 1) sampler that replaces rand 0.6's `Dirichlet::sample`
//    (reference call sites: policy/alphazero_policy/mod.rs:262,283,294,145,163).
//    Only IEEE +,-,*,/ and sqrt are used (no fma contraction: this translation unit
//    is compiled with -ffp-contract=off) plus explicitly written fused multiply-adds (v4: correctly rounded on both
//    sides), so the stream is bit-reproducible.  C6RNG v4 / v5: Gamma variates are (float mantissa, integer exponent) pairs,
//    their normaliser is single precision on exponents relative to the largest of the vector -- no double anywhere.
//  * wave64 reductions used by the tree kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// the specification the random stream follows: games are a function of (this, the net's arithmetic, weights, seed, S) -- the
// whole-game parity tests pin engine and oracle to the same games for one specification
#define C6_RNG_SPEC "C6RNG v5"

#define C6_BLACK (-1)
#define C6_NONE 0
#define C6_WHITE 1

#define C6_TAG_NOISE 1u
#define C6_TAG_TIE 2u
#define C6_TAG_EVAL_V 3u
#define C6_TAG_EVAL_P 4u

#define C6_MAXCELLS 361
#define C6_MAXJ 6              // ceil(361 / 64): cells owned by one lane of a wave
#define C6_EMPTY_BOARD_KEY 0x9E3779B97F4A7C15ull

struct c6_u4 { uint32_t x, y, z, w; };

template <int ROUNDS = 10>
__device__ __forceinline__ c6_u4 c6_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32) instead of a high and a low multiplication
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c6_u4{c0, c1, c2, c3};
}

// 53-bit uniform strictly inside (0,1)
__device__ __forceinline__ double c6_open01(uint32_t hi, uint32_t lo) {
    uint64_t v = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

// U[-1,1) with 24 bits, RandomEvaluator (policy/alphazero_policy/mod.rs:141-168)
__device__ __forceinline__ float c6_u11(uint32_t w) { return 2.0f * ((float)(w >> 8) * (1.0f / 16777216.0f)) - 1.0f; }

// 24-bit uniform strictly inside (0,1)
__device__ __forceinline__ float c6_open01f(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// m * 2^k for a normal float m and -126 <= k <= 127 with a normal result: exact
__device__ __forceinline__ float c6_scale2f(float m, int k) { return __builtin_amdgcn_ldexpf(m, k); }      // v_ldexp_f32: one instruction, the same bits

// ---- C6RNG v5: the Gamma sampler -------------------------------------------------------------------------------------------
// log and exp of the sampler: polynomial forms (Cephes' logf / expf coefficients) on explicitly fused multiply-adds -- no division, no
// table; a value pair at a time (the two attempts of a Philox block) on the packed f32 instructions.  |error| ~ 1e-7: what the
// sampler owes the reference is a DISTRIBUTION (below); the routines are part of the spec, the oracle evaluates the same operations.
typedef float c6_f2 __attribute__((ext_vector_type(2)));
typedef int c6_i2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c6_f2 c6_fma2(c6_f2 a, c6_f2 b, c6_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ c6_f2 c6_fma2(c6_f2 a, c6_f2 b, float c) { return __builtin_elementwise_fma(a, b, c6_f2{c, c}); }
__device__ __forceinline__ c6_f2 c6_fma2(c6_f2 a, float b, c6_f2 c) { return __builtin_elementwise_fma(a, c6_f2{b, b}, c); }
// log(x), x > 0 normal
__device__ __forceinline__ c6_f2 c6_log4(c6_f2 x) {
    c6_f2 m, fe;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        // the spec's split -- e = exponent, mm = mantissa in [1, 2); mm > 1.41421356f (bits above 0x3fb504f3): mm / 2, e + 1 -- in three integer
        // operations (r4): with the bias 0x3f3504f4 the subtraction carries into the exponent exactly for the mantissas that are halved
        const uint32_t t = __float_as_uint(x[i]) - 0x3f3504f4u;
        m[i] = __uint_as_float((t & 0x007fffffu) + 0x3f3504f4u);
        fe[i] = (float)((int)t >> 23);
    }
    const c6_f2 y = m - 1.0f, z = y * y;
    c6_f2 p = c6_f2{7.0376836292e-2f, 7.0376836292e-2f};
    p = c6_fma2(p, y, -1.1514610310e-1f);
    p = c6_fma2(p, y, 1.1676998740e-1f);
    p = c6_fma2(p, y, -1.2420140846e-1f);
    p = c6_fma2(p, y, 1.4249322787e-1f);
    p = c6_fma2(p, y, -1.6668057665e-1f);
    p = c6_fma2(p, y, 2.0000714765e-1f);
    p = c6_fma2(p, y, -2.4999993993e-1f);
    p = c6_fma2(p, y, 3.3333331174e-1f);
    p = (p * y) * z;
    c6_f2 r = c6_fma2(fe, -2.12194440e-4f, p);
    r = c6_fma2(z, -0.5f, r);
    r = y + r;
    return c6_fma2(fe, 0.693359375f, r);
}
// exp(t) = p 2^k with p in about [0.7, 1.42]: the sampler's U^(1/alpha) reaches 2^-830 at alpha = 0.03, so the exponent stays an integer
__device__ __forceinline__ c6_f2 c6_exp4(c6_f2 t, c6_i2 &k) {
    const c6_f2 tt = t * 1.44269504f;
    c6_f2 fk;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        k[i] = (int)(tt[i] + __builtin_copysignf(0.5f, tt[i]));      // (= tt >= 0 ? +0.5 : -0.5 wherever it matters: (int)(-0.0f +- 0.5f) is 0 either way)
        fk[i] = (float)k[i];
    }
    c6_f2 r = c6_fma2(fk, -0.693359375f, t);
    r = c6_fma2(fk, 2.12194440e-4f, r);
    const c6_f2 z = r * r;
    c6_f2 p = c6_f2{1.9875691500e-4f, 1.9875691500e-4f};
    p = c6_fma2(p, r, 1.3981999507e-3f);
    p = c6_fma2(p, r, 8.3334519073e-3f);
    p = c6_fma2(p, r, 4.1665795894e-2f);
    p = c6_fma2(p, r, 1.6666665459e-1f);
    p = c6_fma2(p, r, 5.0000001201e-1f);
    p = c6_fma2(p, z, r);
    return p + 1.0f;
}
// the routines on one value (tests, the Marsaglia-Tsang branch)
__device__ __forceinline__ float c6_log4(float x) { return c6_log4(c6_f2{x, x})[0]; }
__device__ __forceinline__ float c6_exp4(float t, int &k) { c6_i2 kk; const c6_f2 p = c6_exp4(c6_f2{t, t}, kk); k = kk[0]; return p[0]; }

// a Gamma variate: m * 2^k (m a normal float or 0, k <= 0 for the small ones): no double, no underflow
struct c6_gam { float m; int k; };
#define C6_GAM_NONE (-(1 << 30))      // exponent of "no variate" (a cell that needs none): never the largest, scales to 0

// Gamma(alpha, 1).  C6RNG v5.  Attempts 2 i and 2 i + 1 of a cell come out of ONE Philox block (counter word c1 = i; words x, y serve
// the even attempt, z, w the odd one; the noise draws' blocks are Philox4x32-SEVEN) and are evaluated TOGETHER, a pair of values per packed
// instruction; a cell's variate is that of its first accepted attempt.  Attempts are independent of one another, so who evaluates them, and
// in which order, is free (c6_wave_gamma below spreads them over the wave).  Single precision (the reference casts the noise to f32 before it
// is used, alphazero_policy/mod.rs:283).
//  * alpha < 1 (the reference's 0.03): rejection from a two-piece proposal ("GS with a Pareto tail"): q1 = alpha x^(alpha-1) on (0, 1], q2 = beta
//    x^(-beta-1) on (1, inf), beta = 1 - alpha, weights 1 : alpha / (e beta).  b = 1 + alpha / (e beta), p = b u1, E = -log(u2) (an Exp(1) variate).
//      p <= 1:  x = p^(1/alpha)  = exp(log(p) / alpha),                                    accepted iff x <= E                (<=> u2 <= e^-x)
//      p >  1:  x = u'^(-1/beta) = exp(-log(u') / beta), u' = (b - p) e beta / alpha,     accepted iff E >= (x - 1) - log(x)  (<=> u2 <= x e^(1-x))
//    ONE sequence for both branches: a log of the branch's argument, a multiplication by the branch's constant, an exp, a comparison against
//    E.  97 % of the attempts are accepted, a pair fails once in ~1000.  (v4 was Ahrens-Dieter GS: its tail x = -log((b - p) / alpha), accepted iff
//    E >= (1 - alpha) log(x), needs a second logarithm -- executed by the whole wave whenever one of its ~94 attempts fell into the tail, 6 passes in
//    10.  The distribution is the same: tests/test_oracle_rng.py, Kolmogorov-Smirnov and moments on both branches.)
//  * alpha >= 1: Marsaglia-Tsang at shape alpha + 1 times U^(1/alpha) with polar normals (words: v1, v2, squeeze u, power u), one
//    attempt per Philox block (counter word c1 = attempt).
struct c6_gs_consts { float ia, ib, b, ct; };      // 1 / alpha, -1 / beta, b, e beta / alpha: IEEE single precision (the oracle computes the same bits)
__device__ __forceinline__ c6_gs_consts c6_gs_setup(float a) {
    const float beta = 1.0f - a, eb = 2.7182817f * beta;
    return c6_gs_consts{1.0f / a, -1.0f / beta, 1.0f + a / eb, eb / a};
}
__device__ __forceinline__ bool c6_gamma_gs2(const c6_gs_consts &G, c6_u4 w, c6_gam &out) {
    const c6_f2 u1 = {c6_open01f(w.x), c6_open01f(w.z)}, u2 = {c6_open01f(w.y), c6_open01f(w.w)};
    const c6_f2 p = u1 * G.b;
    const c6_f2 y = (G.b - p) * G.ct;
    const bool tl[2] = {!(p[0] <= 1.0f), !(p[1] <= 1.0f)};
    c6_f2 arg = {tl[0] ? y[0] : p[0], tl[1] ? y[1] : p[1]};
    const bool bad[2] = {!(arg[0] > 0.0f), !(arg[1] > 0.0f)};      // p rounded up to b: the attempt is rejected
    if (bad[0]) arg[0] = 1.0f;
    if (bad[1]) arg[1] = 1.0f;
    const c6_f2 L = c6_log4(arg), E = 0.0f - c6_log4(u2);
    const c6_f2 t = L * c6_f2{tl[0] ? G.ib : G.ia, tl[1] ? G.ib : G.ia};
    c6_i2 k;
    const c6_f2 m = c6_exp4(t, k);
    bool acc[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float x = c6_scale2f(m[i], !tl[i] && k[i] < -30 ? -30 : k[i]);      // body: below 2^-30 every E (>= 2^-25) accepts; tail: k <= 18
        acc[i] = !bad[i] && (tl[i] ? E[i] >= (x - 1.0f) - t[i] : (k[i] < -30 || x <= E[i]));
    }
    const int i = acc[0] ? 0 : 1;
    if (!acc[i]) return false;
    out.m = m[i];
    out.k = k[i];
    return true;
}

// one Marsaglia-Tsang attempt (alpha >= 1), Philox counter word c1 = att
__device__ __forceinline__ bool c6_gamma_mt(double alpha, double d_, double c_, uint32_t k0, uint32_t k1, uint32_t cell,
                                            uint32_t search, uint32_t depth, uint32_t att, c6_gam &out) {
    const uint32_t tag = C6_TAG_NOISE | (depth << 8);
    const float inv_alpha = 1.0f / (float)alpha;
    const c6_u4 w = c6_philox<7>(cell, att, search, tag, k0, k1);
    const float d = (float)d_, c = (float)c_;
    const float v1 = 2.0f * c6_open01f(w.x) - 1.0f, v2 = 2.0f * c6_open01f(w.y) - 1.0f;
    const float s = v1 * v1 + v2 * v2;
    if (s >= 1.0f || s == 0.0f) return false;
    const float x = v1 * sqrtf(-2.0f * c6_log4(s) / s);
    const float vc = 1.0f + c * x;
    if (vc <= 0.0f) return false;
    const float v = vc * vc * vc;
    const float u = c6_open01f(w.z), u2 = c6_open01f(w.w);
    const float x2 = x * x;
    if (u < 1.0f - 0.0331f * x2 * x2 || c6_log4(u) < 0.5f * x2 + d * (1.0f - v + c6_log4(v))) {
        int k;
        out.m = (d * v) * c6_exp4(c6_log4(u2) * inv_alpha, k);      // in [~1e-30, ~50] x 2^k: a normal float
        out.k = k;
        return true;
    }
    return false;
}

// attempts 2 * pair and 2 * pair + 1 of a cell: true and the variate of the first accepted one
__device__ __forceinline__ bool c6_gamma_try2(double alpha, const c6_gs_consts &G, double d_, double c_, uint32_t k0, uint32_t k1, uint32_t cell,
                                              uint32_t search, uint32_t depth, uint32_t pair, c6_gam &out) {
    if (alpha < 1.0) return c6_gamma_gs2(G, c6_philox<7>(cell, pair, search, C6_TAG_NOISE | (depth << 8), k0, k1), out);
    if (c6_gamma_mt(alpha, d_, c_, k0, k1, cell, search, depth, 2u * pair, out)) return true;
    return c6_gamma_mt(alpha, d_, c_, k0, k1, cell, search, depth, 2u * pair + 1u, out);
}

__device__ __forceinline__ c6_gam c6_gamma_draw(double alpha, double d, double c, uint32_t k0, uint32_t k1, uint32_t cell,
                                                uint32_t search, uint32_t depth) {
    c6_gam g;
    const c6_gs_consts G = c6_gs_setup((float)alpha);
    for (uint32_t pair = 0;; pair++)
        if (c6_gamma_try2(alpha, G, d, c, k0, k1, cell, search, depth, pair, g)) return g;
}

// ---- wave64 helpers ---------------------------------------------------------

// Wave-wide all-reduce without LDS round trips: xor-butterfly stages 1, 2 as quad permutes, 4 and 8 as row_half_mirror /
// row_mirror (equal to the xor partner's value once quads / octets are uniform), then the four row values by v_readlane,
// combined as (r0 op r1) op (r2 op r3) -- what butterfly stages 16 and 32 leave on every lane when op is commutative.
// (__shfl_xor compiles to ds_bpermute_b32: ~100 cycles of LDS latency per stage, 6 stages per reduction.)
template <int CTRL>
__device__ __forceinline__ int c6_dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float c6_dpp_f32(float v) { return __int_as_float(c6_dpp_i32<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double c6_dpp_f64(double v) {
    return __hiloint2double(c6_dpp_i32<CTRL>(__double2hiint(v)), c6_dpp_i32<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double c6_readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
#define C6_DPP_XOR1 0xB1          /* quad_perm [1,0,3,2] */
#define C6_DPP_XOR2 0x4E          /* quad_perm [2,3,0,1] */
#define C6_DPP_HALF_MIRROR 0x141
#define C6_DPP_MIRROR 0x140

// butterfly sum, offsets 1, 2, 4, ..., 32: every lane ends with the same value; the summation tree is part of the C6RNG spec
__device__ __forceinline__ double c6_butterfly_sum(double v) {
    v = v + c6_dpp_f64<C6_DPP_XOR1>(v);
    v = v + c6_dpp_f64<C6_DPP_XOR2>(v);
    v = v + c6_dpp_f64<C6_DPP_HALF_MIRROR>(v);
    v = v + c6_dpp_f64<C6_DPP_MIRROR>(v);
    const double r0 = c6_readlane_f64(v, 0), r1 = c6_readlane_f64(v, 16), r2 = c6_readlane_f64(v, 32), r3 = c6_readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// the same summation tree in single precision (C6RNG v4: the Dirichlet normaliser)
__device__ __forceinline__ float c6_butterfly_sum_f32(float v) {
    v = v + c6_dpp_f32<C6_DPP_XOR1>(v);
    v = v + c6_dpp_f32<C6_DPP_XOR2>(v);
    v = v + c6_dpp_f32<C6_DPP_HALF_MIRROR>(v);
    v = v + c6_dpp_f32<C6_DPP_MIRROR>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ int c6_wave_sum_i32(int v) {
    v += c6_dpp_i32<C6_DPP_XOR1>(v);
    v += c6_dpp_i32<C6_DPP_XOR2>(v);
    v += c6_dpp_i32<C6_DPP_HALF_MIRROR>(v);
    v += c6_dpp_i32<C6_DPP_MIRROR>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

__device__ __forceinline__ int c6_wave_max_i32(int v) {
    int o;
    o = c6_dpp_i32<C6_DPP_XOR1>(v); v = o > v ? o : v;
    o = c6_dpp_i32<C6_DPP_XOR2>(v); v = o > v ? o : v;
    o = c6_dpp_i32<C6_DPP_HALF_MIRROR>(v); v = o > v ? o : v;
    o = c6_dpp_i32<C6_DPP_MIRROR>(v); v = o > v ? o : v;
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int a = r1 > r0 ? r1 : r0, b = r3 > r2 ? r3 : r2;
    return b > a ? b : a;
}

__device__ __forceinline__ int c6_wave_min_i32(int v) {
    int o;
    o = c6_dpp_i32<C6_DPP_XOR1>(v); v = o < v ? o : v;
    o = c6_dpp_i32<C6_DPP_XOR2>(v); v = o < v ? o : v;
    o = c6_dpp_i32<C6_DPP_HALF_MIRROR>(v); v = o < v ? o : v;
    o = c6_dpp_i32<C6_DPP_MIRROR>(v); v = o < v ? o : v;
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int a = r1 < r0 ? r1 : r0, b = r3 < r2 ? r3 : r2;
    return b < a ? b : a;
}

__device__ __forceinline__ float c6_wave_max_f32(float v) {
    float o;
    o = c6_dpp_f32<C6_DPP_XOR1>(v); v = o > v ? o : v;
    o = c6_dpp_f32<C6_DPP_XOR2>(v); v = o > v ? o : v;
    o = c6_dpp_f32<C6_DPP_HALF_MIRROR>(v); v = o > v ? o : v;
    o = c6_dpp_f32<C6_DPP_MIRROR>(v); v = o > v ? o : v;
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    const float a = r1 > r0 ? r1 : r0, b = r3 > r2 ? r3 : r2;
    return b > a ? b : a;
}

// index of the n-th (0-based) set bit of m
__device__ __forceinline__ int c6_nth_set_bit(unsigned long long m, int n) {
    for (int i = 0; i < n; i++) m &= m - 1;
    return __ffsll((long long)m) - 1;
}

// position of the n-th (0-based) set bit of m by halving (n < popcount(m)); per-lane n without a data-dependent loop
__device__ __forceinline__ int c6_select_bit(unsigned long long m, int n) {
    int pos = 0;
    uint32_t w = (uint32_t)m;
    int c = __popc(w);
    if (n >= c) { n -= c; pos = 32; w = (uint32_t)(m >> 32); }
#pragma unroll
    for (int width = 16; width >= 1; width >>= 1) {
        const uint32_t low = w & ((1u << width) - 1u);
        c = __popc(low);
        if (n >= c) { n -= c; pos += width; w >>= width; } else w = low;
    }
    return pos;
}

// Gamma variates for the cells a wave owns (lane owns cells lane + 64 j where need[j]).  Same values as calling
// c6_gamma_draw per cell -- the attempt that is accepted for a cell does not depend on who evaluates it -- but the
// rejection loops are not run in lockstep: the cells still without a variate are packed onto lanes 0..k-1 in every round
// (occupied cells cost nothing: in the middle of a 9x9 game the 2 x 64 owned slots hold ~47 empty cells -- one pass instead of two).
// `scratch` = wave-private LDS: 64 J 8-byte words through which workers hand the variates (mantissa | exponent, or "rejected") back to
// the owners, followed by 64 J 16-bit words for the list of pending cells (C6_GAM_SCRATCH_DOUBLES(J) doubles in all).  Cells that need
// none come back as {0, C6_GAM_NONE}.
// The packing costs no division and no bit search: a pending cell's rank is a popcount below its lane (v_mbcnt) and it writes its own
// index into the list; every pending cell gets 2^s speculative attempt pairs, the largest power of two that fits the wave (item i ->
// cell list[i >> s], attempt pair att + (i & (2^s - 1)): shifts and masks).
#define C6_GAM_REJ 0x7fffffff
#define C6_GAM_SCRATCH_DOUBLES(J) (64 * (J) + 16 * (J))
template <int J>
__device__ __forceinline__ void c6_wave_gamma(const bool (&need)[J], c6_gam (&gam)[J], double *scratch_, int lane, double alpha,
                                              double d, double c, uint32_t k0, uint32_t k1, uint32_t search, uint32_t depth) {
    unsigned long long *scratch = reinterpret_cast<unsigned long long *>(scratch_);
    unsigned short *list = reinterpret_cast<unsigned short *>(scratch_ + 64 * J);
    unsigned long long pend[J];
    bool any = false;
    const c6_gs_consts G = c6_gs_setup((float)alpha);      // (wave-uniform: three divisions, hoisted out of the round loop by the compiler)
#pragma unroll
    for (int j = 0; j < J; j++) {
        gam[j].m = 0.f;
        gam[j].k = C6_GAM_NONE;
        pend[j] = __ballot(need[j]);
        any |= pend[j] != 0ull;
    }
    for (uint32_t att = 0; any;) {
        int tot = 0;
#pragma unroll
        for (int j = 0; j < J; j++) {
            if ((pend[j] >> lane) & 1ull)
                list[tot + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pend[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pend[j], 0u))] = (unsigned short)(lane + 64 * j);
            tot += __popcll(pend[j]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // LDS is in order within a wave; this binds the compiler
        // 2^s speculative attempt pairs (att .. att + 2^s - 1; c6_gamma_try2) per pending cell while they fit on the 64 lanes; the lowest
        // accepted attempt of a cell wins, exactly as if they had been tried one after the other
        const int s = tot > 32 ? 0 : 6 - (tot > 1 ? 32 - __builtin_clz((unsigned)(tot - 1)) : 0);
        const int k = 1 << s, items = tot << s;
        for (int base = 0; base < items; base += 64) {
            const int item = base + lane;
            const bool active = item < items;
            const int sub = item & (k - 1);
            const int cell = active ? (int)list[item >> s] : 0;
            c6_gam v = {0.f, 0};
            const bool ok = active && c6_gamma_try2(alpha, G, d, c, k0, k1, (uint32_t)cell, search, depth, att + (uint32_t)sub, v);
            // lanes of one cell are consecutive (sub = 0..k-1, one batch whenever k > 1): the first accepted one writes
            const unsigned long long okm = __ballot(ok);
            const unsigned long long mine = (okm >> (lane - sub)) & (k >= 64 ? ~0ull : ((1ull << k) - 1ull));
            if (active && sub == 0 && mine == 0ull) scratch[cell] = (unsigned long long)(uint32_t)C6_GAM_REJ << 32;
            if (ok && (mine & ((1ull << sub) - 1ull)) == 0ull) scratch[cell] = (unsigned long long)__float_as_uint(v.m) | ((unsigned long long)(uint32_t)v.k << 32);
        }
        att += (uint32_t)k;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        any = false;
#pragma unroll
        for (int j = 0; j < J; j++) {
            bool still = false;
            if ((pend[j] >> lane) & 1ull) {
                const unsigned long long v = scratch[lane + 64 * j];
                if ((int)(v >> 32) == C6_GAM_REJ) still = true;
                else { gam[j].m = __uint_as_float((uint32_t)v); gam[j].k = (int)(v >> 32); }
            }
            pend[j] = __ballot(still);
            any |= pend[j] != 0ull;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// Dirichlet noise from the Gamma variates of a wave's cells (C6RNG v4): exponents relative to the LARGEST of the vector (its scaled
// value lies in [0.7, 18): the sum cannot vanish, whatever the range of the others), values more than 2^100 below it are 0; lane
// partial sums in ascending j, the wave sum by the butterfly (ascending offsets), noise = scaled * (1 / sum).
template <int J>
__device__ __forceinline__ void c6_wave_dirichlet(const c6_gam (&gam)[J], float (&noise)[J]) {
    int kmax = C6_GAM_NONE;
#pragma unroll
    for (int j = 0; j < J; j++) kmax = gam[j].k > kmax ? gam[j].k : kmax;
    kmax = c6_wave_max_i32(kmax);
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int dk = gam[j].k - kmax;
        noise[j] = dk >= -100 ? c6_scale2f(gam[j].m, dk) : 0.f;
        part = part + noise[j];
    }
    const float rs = 1.0f / c6_butterfly_sum_f32(part);
#pragma unroll
    for (int j = 0; j < J; j++) noise[j] = noise[j] * rs;
}
