// c6_device.h -- device-side building blocks shared by the HIP kernels (gfx950 only).
//
//  * C6RNG: the counter-based random stream of DESIGN.md (Philox4x32-10 keyed by
//    (seed, game uid); counter = (cell, attempt, search, tag | depth << 8)) and the
//    Gamma(alpha < 1) sampler that replaces rand 0.6's `Dirichlet::sample`
//    (reference call sites: policy/alphazero_policy/mod.rs:262,283,294,145,163).
//    Only IEEE +,-,*,/ and sqrt are used (no fma contraction: this translation unit
//    is compiled with -ffp-contract=off), so the stream is bit-reproducible.
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

__device__ __forceinline__ double c6_det_log(double x) {
    uint64_t b = (uint64_t)__double_as_longlong(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = (m - 1.0) / (m + 1.0), f2 = f * f;
    double p = 1.0 / 23.0;
    p = p * f2 + 1.0 / 21.0;
    p = p * f2 + 1.0 / 19.0;
    p = p * f2 + 1.0 / 17.0;
    p = p * f2 + 1.0 / 15.0;
    p = p * f2 + 1.0 / 13.0;
    p = p * f2 + 1.0 / 11.0;
    p = p * f2 + 1.0 / 9.0;
    p = p * f2 + 1.0 / 7.0;
    p = p * f2 + 1.0 / 5.0;
    p = p * f2 + 1.0 / 3.0;
    p = p * f2 + 1.0;
    double lm = 2.0 * f * p;
    double de = (double)e;
    return de * 6.93147180369123816490e-01 + (lm + de * 1.90821492927058770002e-10);
}

__device__ __forceinline__ double c6_det_exp(double x) {
    if (x < -746.0) return 0.0;
    double t = x * 1.44269504088896338700e+00;
    long long k = (long long)(t + (t >= 0 ? 0.5 : -0.5));
    double dk = (double)k;
    double r = (x - dk * 6.93147180369123816490e-01) - dk * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    long long k1 = k / 2, k2 = k - k1;
    double s1 = __longlong_as_double((k1 + 1023) << 52), s2 = __longlong_as_double((k2 + 1023) << 52);
    return (p * s1) * s2;
}

// Gamma(alpha, 1), alpha < 1: Marsaglia-Tsang at shape alpha+1 times U^(1/alpha) (rand 0.6 GammaSmallShape),
// normal deviates by the polar method.  d and c are the per-kernel constants of the sampler.
// One ATTEMPT of the C6RNG spec (counter words 2*att and 2*att+1 of this cell): true and the variate if it is accepted.
// A cell's variate is that of its first accepted attempt; attempts are independent of one another, so who evaluates
// them, and in which order, is free (c6_wave_gamma below spreads the retries over the wave).
__device__ __forceinline__ bool c6_gamma_try(double alpha, double d, double c, uint32_t k0, uint32_t k1, uint32_t cell,
                                             uint32_t search, uint32_t depth, uint32_t att, double &out) {
    const uint32_t tag = C6_TAG_NOISE | (depth << 8);
    c6_u4 w = c6_philox(cell, 2 * att, search, tag, k0, k1);
    double v1 = 2.0 * c6_open01(w.x, w.y) - 1.0, v2 = 2.0 * c6_open01(w.z, w.w) - 1.0;
    double s = v1 * v1 + v2 * v2;
    if (s >= 1.0 || s == 0.0) return false;
    double x = v1 * sqrt(-2.0 * c6_det_log(s) / s);
    double vc = 1.0 + c * x;
    if (vc <= 0.0) return false;
    double v = vc * vc * vc;
    w = c6_philox(cell, 2 * att + 1, search, tag, k0, k1);
    double u = c6_open01(w.x, w.y), u2 = c6_open01(w.z, w.w);
    double x2 = x * x;
    if (u < 1.0 - 0.0331 * x2 * x2 || c6_det_log(u) < 0.5 * x2 + d * (1.0 - v + c6_det_log(v))) {
        double g = d * v;
        out = g * c6_det_exp(c6_det_log(u2) / alpha);
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
