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

// Gamma(alpha, 1), alpha < 1: Marsaglia-Tsang at shape alpha+1 times U^(1/alpha) (rand 0.6 GammaSmallShape /
// GammaLargeShape), normal deviates by the polar method.  C6RNG v2: the variate is computed in SINGLE precision -- the
// reference casts the noise to f32 before it is used (alphazero_policy/mod.rs:283) and f64 vector arithmetic runs at half
// the f32 rate on gfx950 -- and one Philox block serves an attempt (words: polar v1, polar v2, squeeze u, power u).
// d and c are the sampler's constants (double, rounded to float here).
// One ATTEMPT of the spec (counter word c1 = att of this cell): true and the variate if it is accepted.  A cell's
// variate is that of its first accepted attempt; attempts are independent of one another, so who evaluates them, and in
// which order, is free (c6_wave_gamma below spreads them over the wave).
__device__ __forceinline__ bool c6_gamma_try(double alpha, double d_, double c_, uint32_t k0, uint32_t k1, uint32_t cell,
                                             uint32_t search, uint32_t depth, uint32_t att, double &out) {
    const uint32_t tag = C6_TAG_NOISE | (depth << 8);
    const float d = (float)d_, c = (float)c_, inv_alpha = 1.0f / (float)alpha;
    const c6_u4 w = c6_philox(cell, att, search, tag, k0, k1);
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
        // 2^k exactly, in double (k >= -1000: far inside the normal range; smaller variates are 0)
        out = k < -1000 ? 0.0 : (double)g * __longlong_as_double((long long)(k + 1023) << 52);
        return true;
    }
    return false;
}

__device__ __forceinline__ double c6_gamma_draw(double alpha, double d, double c, uint32_t k0, uint32_t k1, uint32_t cell,
                                                uint32_t search, uint32_t depth) {
    double g;
    for (uint32_t att = 0;; att++)
        if (c6_gamma_try(alpha, d, c, k0, k1, cell, search, depth, att, g)) return g;
}

// ---- wave64 helpers ---------------------------------------------------------

__device__ __forceinline__ double c6_shfl_xor_f64(double v, int off) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, off, 64);
    hi = __shfl_xor(hi, off, 64);
    return __hiloint2double(hi, lo);
}

// xor-butterfly sum: every lane ends with the same value; the summation tree is part of the C6RNG spec
__device__ __forceinline__ double c6_butterfly_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + c6_shfl_xor_f64(v, off);
    return v;
}

__device__ __forceinline__ int c6_wave_sum_i32(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ int c6_wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { int o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

__device__ __forceinline__ int c6_wave_min_i32(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { int o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
    return v;
}

__device__ __forceinline__ float c6_wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { float o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
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
        // k speculative attempts (att .. att+k-1) per pending cell while they fit on the 64 lanes; the lowest accepted
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
            const bool ok = active && c6_gamma_try(alpha, d, c, k0, k1, (uint32_t)cell, search, depth, att + (uint32_t)sub, v);
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
