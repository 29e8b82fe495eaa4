// c6_device.h -- device-side building blocks shared by the HIP kernels (gfx950 only).
//
//  * C6RNG: the counter-based random stream of DESIGN.md (Philox4x32-10 keyed by
//    (seed, game uid); counter = (cell, attempt, search, tag | depth << 8)) and the
//    Gamma(alpha < 1) sampler that replaces rand 0.6's `Dirichlet::sample`
//    (reference call sites: policy/alphazero_policy/mod.rs:262,283,294,145,163).
//    Only IEEE +,-,*,/ and sqrt are used (no fma contraction: this translation unit
//    is compiled with -ffp-contract=off), so the stream is bit-reproducible.  The Gamma
//    variates are single precision (C6RNG v2), their normaliser double.
//  * wave64 reductions used by the tree kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

__device__ __forceinline__ c6_u4 c6_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
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

// single-precision log / exp of the Gamma sampler: explicit polynomials on IEEE + - * / (no contraction) -> the same
// bits on the CPU oracle (c6o_det_logf / c6o_det_expf_parts).  logf: x > 0 normal.
__device__ __forceinline__ float c6_det_logf(float x) {
    uint32_t b = __float_as_uint(x);
    int e = (int)((b >> 23) & 0xff) - 127;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = (m - 1.0f) / (m + 1.0f), f2 = f * f;
    float p = 1.0f / 11.0f;
    p = p * f2 + 1.0f / 9.0f;
    p = p * f2 + 1.0f / 7.0f;
    p = p * f2 + 1.0f / 5.0f;
    p = p * f2 + 1.0f / 3.0f;
    p = p * f2 + 1.0f;
    float lm = 2.0f * f * p;
    float de = (float)e;
    return de * 0.693145751953125f + (lm + de * 1.42860677e-6f);
}

// exp(x) = p * 2^k with p in about [0.7, 1.42]: the polynomial part in float, the exponent returned separately -- the
// sampler's U^(1/alpha) reaches 1e-250 at alpha = 0.03 and must not underflow (an all-zero noise vector would be NaN
// after normalisation), so the scaling is applied in double
__device__ __forceinline__ float c6_det_expf_parts(float x, int &k) {
    float t = x * 1.44269504f;
    k = (int)(t + (t >= 0.0f ? 0.5f : -0.5f));
    float dk = (float)k;
    float r = (x - dk * 0.693145751953125f) - dk * 1.42860677e-6f;
    float p = 1.0f / 40320.0f;
    p = p * r + 1.0f / 5040.0f;
    p = p * r + 1.0f / 720.0f;
    p = p * r + 1.0f / 120.0f;
    p = p * r + 1.0f / 24.0f;
    p = p * r + 1.0f / 6.0f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    return p;
}

// 24-bit uniform strictly inside (0,1)
__device__ __forceinline__ float c6_open01f(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// m * 2^k for a normal float m and -126 <= k <= 127 with a normal result: exact
__device__ __forceinline__ float c6_scale2f(float m, int k) { return m * __uint_as_float((uint32_t)(k + 127) << 23); }

// Gamma(alpha, 1).  C6RNG v3, one ATTEMPT of the spec (counter word c1 = att of this cell): true and the variate if it is
// accepted.  A cell's variate is that of its first accepted attempt; attempts are independent of one another, so who
// evaluates them, and in which order, is free (c6_wave_gamma below spreads them over the wave).  Single precision (the
// reference casts the noise to f32 before it is used, alphazero_policy/mod.rs:283; f64 vector arithmetic runs at half the
// f32 rate on gfx950), one Philox block per attempt.
//  * alpha < 1 (the reference's 0.03): Ahrens-Dieter GS -- p = (1 + alpha/e) u1; p <= 1: x = p^(1/alpha), accepted with
//    probability exp(-x); else x = -log((1 + alpha/e - p) / alpha), accepted with probability x^(alpha-1).  Two attempts per
//    Philox block (c6_gamma_try2).
//    One log and one exp per attempt, 97 % accepted at alpha = 0.03 (v2 drew Marsaglia-Tsang at alpha+1 times
//    U^(1/alpha) with polar normals as rand 0.6 does for small shapes: three logs, an exp, a sqrt and two divisions per
//    attempt, 75 % accepted; the distribution is the same, the select kernel spends its time here).
//  * alpha >= 1: Marsaglia-Tsang at shape alpha+1 times U^(1/alpha) with polar normals (words: v1, v2, squeeze u, power
//    u); d and c are its constants (double, rounded to float here).
// one GS attempt on the uniforms (u1, u2)
__device__ __forceinline__ bool c6_gamma_gs(float a, float inv_alpha, float b, float u1, float u2, double &out) {
    const float p = b * u1;
    if (p <= 1.0f) {
        int k;
        const float m = c6_det_expf_parts(c6_det_logf(p) * inv_alpha, k);      // x = m 2^k, m in about [0.7, 1.42]
        if (k >= -30) {      // below 2^-30, exp(-x) exceeds every u2 (<= 1 - 2^-25)
            int k2;
            const float e2 = c6_det_expf_parts(-c6_scale2f(m, k), k2);
            if (!(u2 <= c6_scale2f(e2, k2))) return false;
        }
        // 2^k exactly, in double (k >= -1000: far inside the normal range; smaller variates are 0)
        out = k < -1000 ? 0.0 : (double)m * __longlong_as_double((long long)(k + 1023) << 52);
        return true;
    }
    const float y = (b - p) * inv_alpha;      // in (0, 1/e]
    if (!(y > 0.0f)) return false;            // p rounded up to b
    const float x = 0.0f - c6_det_logf(y);
    int k2;
    const float e2 = c6_det_expf_parts((a - 1.0f) * c6_det_logf(x), k2);
    if (!(u2 <= c6_scale2f(e2, k2))) return false;
    out = (double)x;
    return true;
}

// one Marsaglia-Tsang attempt (alpha >= 1), Philox counter word c1 = att
__device__ __forceinline__ bool c6_gamma_mt(double alpha, double d_, double c_, uint32_t k0, uint32_t k1, uint32_t cell,
                                            uint32_t search, uint32_t depth, uint32_t att, double &out) {
    const uint32_t tag = C6_TAG_NOISE | (depth << 8);
    const float inv_alpha = 1.0f / (float)alpha;
    const c6_u4 w = c6_philox(cell, att, search, tag, k0, k1);
    const float d = (float)d_, c = (float)c_;
    const float v1 = 2.0f * c6_open01f(w.x) - 1.0f, v2 = 2.0f * c6_open01f(w.y) - 1.0f;
    const float s = v1 * v1 + v2 * v2;
    if (s >= 1.0f || s == 0.0f) return false;
    const float x = v1 * sqrtf(-2.0f * c6_det_logf(s) / s);
    const float vc = 1.0f + c * x;
    if (vc <= 0.0f) return false;
    const float v = vc * vc * vc;
    const float u = c6_open01f(w.z), u2 = c6_open01f(w.w);
    const float x2 = x * x;
    if (u < 1.0f - 0.0331f * x2 * x2 || c6_det_logf(u) < 0.5f * x2 + d * (1.0f - v + c6_det_logf(v))) {
        int k;
        const float g = (d * v) * c6_det_expf_parts(c6_det_logf(u2) * inv_alpha, k);      // in [~1e-30, ~50]: a normal float
        out = k < -1000 ? 0.0 : (double)g * __longlong_as_double((long long)(k + 1023) << 52);
        return true;
    }
    return false;
}

// attempts 2 * pair and 2 * pair + 1 of a cell: true and the variate of the first accepted one.  alpha < 1: both GS
// attempts come out of ONE Philox block (counter word c1 = pair; words x, y serve the even attempt, z, w the odd one) and
// the odd one is only evaluated where the even one was rejected (3 %): a pair fails once in ~1000, so a second packing
// round of c6_wave_gamma is rare.
__device__ __forceinline__ bool c6_gamma_try2(double alpha, double d_, double c_, uint32_t k0, uint32_t k1, uint32_t cell,
                                              uint32_t search, uint32_t depth, uint32_t pair, double &out) {
    if (alpha < 1.0) {
        const float a = (float)alpha, inv_alpha = 1.0f / a, b = 1.0f + a * 0.36787944f;
        const c6_u4 w = c6_philox(cell, pair, search, C6_TAG_NOISE | (depth << 8), k0, k1);
        if (c6_gamma_gs(a, inv_alpha, b, c6_open01f(w.x), c6_open01f(w.y), out)) return true;
        return c6_gamma_gs(a, inv_alpha, b, c6_open01f(w.z), c6_open01f(w.w), out);
    }
    if (c6_gamma_mt(alpha, d_, c_, k0, k1, cell, search, depth, 2u * pair, out)) return true;
    return c6_gamma_mt(alpha, d_, c_, k0, k1, cell, search, depth, 2u * pair + 1u, out);
}

__device__ __forceinline__ double c6_gamma_draw(double alpha, double d, double c, uint32_t k0, uint32_t k1, uint32_t cell,
                                                uint32_t search, uint32_t depth) {
    double g;
    for (uint32_t pair = 0;; pair++)
        if (c6_gamma_try2(alpha, d, c, k0, k1, cell, search, depth, pair, g)) return g;
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
// rejection loops are not run in lockstep: with one loop per lane the wave pays max-over-64-lanes attempts per j
// (~4.3 for the 21 % polar rejection, 26 evaluations at N = 19), here the cells still without a variate are packed
// onto lanes 0..k-1 in every round (~11 evaluations on an empty 19x19 board, ~4 instead of ~9 in a 9x9 middle game).
// `scratch` = cells doubles of wave-private LDS: workers hand the variates (or -1 = rejected) back to the owners.
template <int J>
__device__ __forceinline__ void c6_wave_gamma(const bool (&need)[J], double (&gam)[J], double *scratch, int lane, double alpha,
                                              double d, double c, uint32_t k0, uint32_t k1, uint32_t search, uint32_t depth) {
    unsigned long long pend[J];
    bool any = false;
#pragma unroll
    for (int j = 0; j < J; j++) {
        gam[j] = 0.0;
        pend[j] = __ballot(need[j]);
        any |= pend[j] != 0ull;
    }
    // every round -- the first included -- packs the pending cells onto the lanes: occupied cells cost nothing (in the
    // middle of a 9x9 game the 2 x 64 owned slots hold ~47 empty cells: one pass instead of two)
    for (uint32_t att = 0; any;) {
        int tot = 0;
#pragma unroll
        for (int j = 0; j < J; j++) tot += __popcll(pend[j]);
        // k speculative attempt pairs (att .. att+k-1; c6_gamma_try2) per pending cell while they fit on the 64 lanes; the lowest accepted
        // attempt of a cell wins, exactly as if they had been tried one after the other
        const int k = tot >= 64 ? 1 : 64 / tot;
        for (int base = 0; base < tot * k; base += 64) {
            const int item = base + lane;
            const bool active = item < tot * k;
            const int idx = item / k, sub = item - idx * k;
            int rem = idx, cell = -1;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int pc = __popcll(pend[j]);
                if (active && cell < 0) {
                    if (rem < pc) cell = c6_select_bit(pend[j], rem) + 64 * j;
                    else rem -= pc;
                }
            }
            double v = 0.0;
            const bool ok = active && c6_gamma_try2(alpha, d, c, k0, k1, (uint32_t)cell, search, depth, att + (uint32_t)sub, v);
            // lanes of one cell are consecutive (sub = 0..k-1, one batch whenever k > 1): the first accepted one writes
            const unsigned long long okm = __ballot(ok);
            const unsigned long long mine = (okm >> (lane - sub)) & (k >= 64 ? ~0ull : ((1ull << k) - 1ull));
            if (active && sub == 0 && mine == 0ull) scratch[cell] = -1.0;
            if (ok && (mine & ((1ull << sub) - 1ull)) == 0ull) scratch[cell] = v;
        }
        att += (uint32_t)k;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // LDS is in order within a wave; this binds the compiler
        any = false;
#pragma unroll
        for (int j = 0; j < J; j++) {
            bool still = false;
            if ((pend[j] >> lane) & 1ull) {
                const double v = scratch[lane + 64 * j];
                if (v < 0.0) still = true; else gam[j] = v;
            }
            pend[j] = __ballot(still);
            any |= pend[j] != 0ull;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}
