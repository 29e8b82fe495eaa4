// conv_wino6.hip -- the evaluator's FP32 Winograd tower at 19x19, C = 256: the tiling of conv_wino4.hip with tiles SIX pixels wide.
//
// Same layer and the same f32 arithmetic as conv_wino4.hip (conv3x3 C -> C, pad 1, + folded-BN bias (+ residual) + ReLU;
// AlphaZero/cpp_weighted/model.hpp:61-78; f32 operands, exact f32 products, f32 accumulate).  What changes is how a board is cut:
//     columns 19 = 6 + 6 + 4 + 3     F(6,3) (8 patch columns: points 0, +-1, +-2, +-1/2, inf), F(6,3), F(4,3), F(3,3)
//     rows    19 = 4 + 4 + 4 + 4 + 3 as before (the boundary kernels' rolling window holds 6 pixel rows of LDS: an 8-row window for tiles six
//                                    pixels TALL would be 180 KB)
// 20 tiles per board in four classes, each in its own region of V / Z so that a GEMM block of 64 tiles is of one class:
//     A   tx = 2, ty < 4    4 x 4 outputs, F(4) x F(4), 36 positions  p = 6 i + j
//     B   tx = 3, ty < 4    4 x 3, F(4) vertically, F(3) horizontally: 6 x 5 positions stored transposed, p = 6 j + i (j < 5)
//     B'  ty = 4, tx >= 2   3 x 4 (and the 3 x 3 corner as 3 x 4): F(3) vertically, F(4) horizontally: p = 6 i + j (i < 5)
//     D   tx < 2            4 x 6 (ty = 4: 3 x 6 as 4 x 6), F(4) vertically, F(6) horizontally: 6 x 8 positions stored transposed,
//                           p = 6 j + i (j < 8)
// The 6-long direction (F(4)'s six patch points) always runs INSIDE a row of positions, so the GEMM kernel's register half of the output
// transform is the one F(4) A^T of conv_wino4.hip for every class; a D block simply has 8 rows of positions and its own set of transformed
// weights.  804 instead of 846 positions per board (-5 % matrix work and V bytes), 536 instead of 564 row sums (Z).
// And a board has FOUR tile columns: the boundary kernels run 4 waves per workgroup, one per SIMD with the whole register file each --
// room to fetch the next tile row's Z while the current one is transformed (conv_wino4.hip's five waves share SIMDs and cannot).
//
// Kernels (the tower of a net with >= 2 residual blocks: net_kernels.hip):
//   c6_w6_stem_in_kernel   boards -> V of the first layer (the stem computed in place)
//   c6_w6_gemm_kernel      conv_wino4.hip's GEMM schedule over the four classes
//   c6_w6_mid_kernel       Z of layer l -> (activations) -> V of layer l + 1; producer and consumer waves
//   c6_w6_out_kernel       Z of the last layer -> the heads' 1x1 filters
// Rounding: F(6,3)'s matrices (entries up to 32 and 21/4) in ONE direction: a layer's error ~3e-6 rms of activations of order 1; the net's
// outputs stay within 1e-5 of a float64 model (tests/test_gpu_net.py).
#include "c6_wino.h"
#include <cstdlib>

#define W6_BT 64          // tiles per block
#define W6_C 256
#define W6_NPV 49         // 1-KiB rows of a tile's slot in V: 48 positions (class D uses all of them) + one of padding -- tile rows of a GEMM slab
#define W6_NZ 33          // and of its slot in Z: 32 row sums + one of padding -- then lie an ODD number of KiB apart (48 / 32 KiB strides
                          // put the 64 rows of a slab on a fraction of the memory channels)
#define W6_PF 7
enum { W6_CLS_A = 0, W6_CLS_B = 1, W6_CLS_BP = 2, W6_CLS_D = 3, W6_CLS_DP = 4 };
#define W6_SETS_POS (36 + 30 + 30 + 48 + 40)      // position matrices of the five weight sets

// The board's cut into tiles.  19 = 6 + 6 + 4 + 3 across, 4 + 4 + 4 + 4 + 3 down: 20 tiles, 788 positions (the benchmark's board).
// r6 -- 15 = 6 + 6 + 3 across, 4 + 4 + 4 + 3 down: 12 tiles, 6 x 48 + 2 x 40 + 3 x 30 + 30 = 488 positions, the REFERENCE's board
// (Connect6/libconnect6/src/lib.rs:21; conv_wino4.hip's F(4x4) tiling of it has 534): no class-A tile, everything else the same kernels.
// NA .. NDP: tiles of a class per board; COLS: window columns (board columns -1 .. N + 1).
template <int N> struct W6G;
template <> struct W6G<19> {
    static constexpr int N = 19, TWX = 4, TWY = 5, T = 20, NA = 4, NB = 4, NBP = 2, ND = 8, NDP = 2, COLS = 22, ROWF = COLS * W6_C;
    __host__ __device__ static constexpr int x0(int tx) { return tx == 0 ? 0 : tx == 1 ? 6 : tx == 2 ? 12 : 16; }
    __host__ __device__ static constexpr int width(int tx) { return tx < 2 ? 6 : tx == 2 ? 4 : 3; }
    __host__ __device__ static constexpr int cls(int ty, int tx) {
        return tx < 2 ? (ty == TWY - 1 ? W6_CLS_DP : W6_CLS_D) : ty == TWY - 1 ? W6_CLS_BP : tx == 2 ? W6_CLS_A : W6_CLS_B;
    }
};
template <> struct W6G<15> {
    static constexpr int N = 15, TWX = 3, TWY = 4, T = 12, NA = 0, NB = 3, NBP = 1, ND = 6, NDP = 2, COLS = 18, ROWF = COLS * W6_C;
    __host__ __device__ static constexpr int x0(int tx) { return 6 * tx; }
    __host__ __device__ static constexpr int width(int tx) { return tx < 2 ? 6 : 3; }
    __host__ __device__ static constexpr int cls(int ty, int tx) { return tx < 2 ? (ty == TWY - 1 ? W6_CLS_DP : W6_CLS_D) : ty == TWY - 1 ? W6_CLS_BP : W6_CLS_B; }
};
template <int N> struct W6Map {
    using G = W6G<N>;
    long blkB, blkBP, blkD, blkDP, nblocks;
    __host__ __device__ explicit W6Map(long nboards) {
        blkB = (nboards * G::NA + W6_BT - 1) / W6_BT;
        blkBP = blkB + (nboards * G::NB + W6_BT - 1) / W6_BT;
        blkD = blkBP + (nboards * G::NBP + W6_BT - 1) / W6_BT;
        blkDP = blkD + (nboards * G::ND + W6_BT - 1) / W6_BT;
        nblocks = blkDP + (nboards * G::NDP + W6_BT - 1) / W6_BT;
    }
    // the tile's slot: block slot / 64, tile slot % 64 of the block
    __host__ __device__ __forceinline__ long slot(long bd, int ty, int tx) const {
        if (tx < 2) return ty == G::TWY - 1 ? blkDP * W6_BT + bd * G::NDP + tx : blkD * W6_BT + bd * G::ND + ty * 2 + tx;
        if (ty == G::TWY - 1) return blkBP * W6_BT + bd * G::NBP + (tx - 2);
        if (G::NA && tx == 2) return bd * G::NA + ty;
        return blkB * W6_BT + bd * G::NB + ty;
    }
};
#define W6_VTILE ((size_t)W6_NPV * W6_C)
#define W6_VBLOCK ((size_t)W6_BT * W6_NPV * W6_C)
#define W6_ZTILE ((size_t)W6_NZ * W6_C)
#define W6_ZBLOCK ((size_t)W6_BT * W6_NZ * W6_C)

// ---- F(6,3), points 0, +-1, +-2, +-1/2, inf (the matrices of Lavin & Gray's generator) ---------------------------------------------
// row i of B^T (8 x 8) applied along the first index of x[8]:
//   [1 0 -21/4 0 21/4 0 -1 0; 0 1 1 -17/4 -17/4 1 1 0; 0 -1 1 17/4 -17/4 -1 1 0; 0 1/2 1/4 -5/2 -5/4 2 1 0; 0 -1/2 1/4 5/2 -5/4 -2 1 0;
//    0 2 4 -5/2 -5 1/2 1 0; 0 -2 4 5/2 -5 -1/2 1 0; 0 -1 0 21/4 0 -21/4 0 1]
// all eight at once, through the even / odd parts the rows share
template <typename V>
__device__ __forceinline__ void w6_bt6(const V (&x)[8], V (&r)[8]) {
    r[0] = w4_fma(5.25f, x[4] - x[2], x[0] - x[6]);
    r[7] = w4_fma(5.25f, x[3] - x[5], x[7] - x[1]);
    const V e1 = w4_fma(-4.25f, x[4], x[2] + x[6]), o1 = w4_fma(-4.25f, x[3], x[1] + x[5]);
    r[1] = e1 + o1;
    r[2] = e1 - o1;
    const V e2 = w4_fma(0.25f, x[2], w4_fma(-1.25f, x[4], x[6])), o2 = w4_fma(0.5f, x[1], w4_fma(-2.5f, x[3], x[5] + x[5]));
    r[3] = e2 + o2;
    r[4] = e2 - o2;
    const V e3 = w4_fma(4.f, x[2], w4_fma(-5.f, x[4], x[6])), o3 = w4_fma(2.f, x[1], w4_fma(-2.5f, x[3], 0.5f * x[5]));
    r[5] = e3 + o3;
    r[6] = e3 - o3;
}
// B^T d B of one patch -> the tile's positions in V (2 channels per pass, every window value read from LDS once: conv_wino4.hip).
// The patch's rows lie in the window slots slot[0 .. 5] (float offsets), wc = its first column (channel 0), vt = the tile in V.
// h0 .. h1 - 1: the channel halves this wave transforms (0, 2: both, one pass after the other; the first layer's kernel gives each half a wave)
__device__ __forceinline__ void w6_patch_transform(const float *wc, const int (&slot)[6], float *vt, int lane, int cls, int h0 = 0, int h1 = 2) {
    constexpr int C = W6_C;
#pragma unroll 1
    for (int h = h0; h < h1; h++) {
        const int co = h * (C / 2) + lane * 2;
        const float *wl = wc + co;
        float *vl = vt + co;
#define W6_D(a, b) (*reinterpret_cast<const f32x2 *>(wl + slot[a] + (b) * C))
#define W6_VST(pos, val) *reinterpret_cast<f32x2 *>(vl + (size_t)(pos) * C) = (val)
        if (cls == W6_CLS_D) {      // 6 patch rows (F(4) vertically) x 8 columns (F(6) horizontally), stored transposed: p = 6 j + i
            f32x2 t[6][8];
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const f32x2 d0 = W6_D(0, b), d1 = W6_D(1, b), d2 = W6_D(2, b), d3 = W6_D(3, b), d4 = W6_D(4, b), d5 = W6_D(5, b);
#pragma unroll
                for (int i = 0; i < 6; i++) t[i][b] = W4_BT_ROW(i, d0, d1, d2, d3, d4, d5);
            }
#pragma unroll
            for (int i = 0; i < 6; i++) {
                f32x2 r[8];
                w6_bt6(t[i], r);
#pragma unroll
                for (int jj = 0; jj < 8; jj++) W6_VST(jj * 6 + i, r[jj]);
            }
        } else if (cls == W6_CLS_DP) {      // 5 patch rows (F(3) vertically) x 8 columns (F(6) horizontally), stored transposed: p = 5 j + i
            f32x2 t[5][8];
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const f32x2 d0 = W6_D(0, b), d1 = W6_D(1, b), d2 = W6_D(2, b), d3 = W6_D(3, b), d4 = W6_D(4, b);
#pragma unroll
                for (int i = 0; i < 5; i++) t[i][b] = W4_BT3_ROW(i, d0, d1, d2, d3, d4);
            }
#pragma unroll
            for (int i = 0; i < 5; i++) {
                f32x2 r[8];
                w6_bt6(t[i], r);
#pragma unroll
                for (int jj = 0; jj < 8; jj++) W6_VST(jj * 5 + i, r[jj]);
            }
        } else if (cls == W6_CLS_A) {
            f32x2 t[6][6];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const f32x2 d0 = W6_D(0, b), d1 = W6_D(1, b), d2 = W6_D(2, b), d3 = W6_D(3, b), d4 = W6_D(4, b), d5 = W6_D(5, b);
#pragma unroll
                for (int i = 0; i < 6; i++) t[i][b] = W4_BT_ROW(i, d0, d1, d2, d3, d4, d5);
            }
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int jj = 0; jj < 6; jj++) W6_VST(i * 6 + jj, W4_BT_ROW(jj, t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5]));
        } else if (cls == W6_CLS_BP) {      // 5 patch rows (F(3) vertically) x 6 columns
            f32x2 t[5][6];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const f32x2 d0 = W6_D(0, b), d1 = W6_D(1, b), d2 = W6_D(2, b), d3 = W6_D(3, b), d4 = W6_D(4, b);
#pragma unroll
                for (int i = 0; i < 5; i++) t[i][b] = W4_BT3_ROW(i, d0, d1, d2, d3, d4);
            }
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int jj = 0; jj < 6; jj++) W6_VST(i * 6 + jj, W4_BT_ROW(jj, t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5]));
        } else {                            // B: 6 patch rows x 5 columns (F(3) horizontally), positions stored transposed
            f32x2 t[6][5];
#pragma unroll
            for (int b = 0; b < 5; b++) {
                const f32x2 d0 = W6_D(0, b), d1 = W6_D(1, b), d2 = W6_D(2, b), d3 = W6_D(3, b), d4 = W6_D(4, b), d5 = W6_D(5, b);
#pragma unroll
                for (int i = 0; i < 6; i++) t[i][b] = W4_BT_ROW(i, d0, d1, d2, d3, d4, d5);
            }
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int jj = 0; jj < 5; jj++) W6_VST(jj * 6 + i, W4_BT3_ROW(jj, t[i][0], t[i][1], t[i][2], t[i][3], t[i][4]));
        }
#undef W6_D
#undef W6_VST
        __builtin_amdgcn_sched_barrier(0);      // one pass at a time
    }
}
// window slot offsets of the six patch rows of tile row ty (board rows 4 ty - 1 .. 4 ty + 4; row y lives in slot (y + 1) % 6)
template <int N>
__device__ __forceinline__ void w6_slots(int ty, int (&slot)[6]) {
#pragma unroll
    for (int a = 0; a < 6; a++) slot[a] = ((4 * ty + a) % 6) * W6G<N>::ROWF;
}

// The boundary kernels' barriers order accesses to the LDS window only: wait for the wave's LDS operations, not for its global loads and
// stores (__syncthreads() also drains vmcnt: every V store and every prefetched Z load would have to land before the workgroup moves on)
#define W6_LDS_BARRIER()                                        \
    do {                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_s_barrier();                           \
        asm volatile("" ::: "memory");                          \
    } while (0)

// ---- first layer: the stem computed into the window, input transform -> V -----------------------------------------------------------
#define W6_THREADS(N) (128 * W6G<N>::TWX)      // 8 waves (15 x 15: 6): wave = (tile column, channel half).  r4: with four waves (a wave per tile column, both halves) the kernel kept ONE
                            // wave per SIMD, not waiting 69 % of the time -- the stem's arithmetic, the transform and the store issue all hung on it
template <int N>
__global__ __launch_bounds__(W6_THREADS(N)) void c6_w6_stem_in_kernel(C6StemSrc S, float *__restrict__ V, int nboards) {
    using G = W6G<N>;
    constexpr int C = W6_C, LPT = C / 4, P = N + 2, ROWF = G::ROWF, NT = W6_THREADS(N);
    extern __shared__ __attribute__((aligned(16))) float win[];                // [6][COLS][C] + the player-plane sums [9][C] + the board plane [P][P]
    float *s0 = win + 6 * ROWF, *bd = s0 + 9 * C;
    const int tid = threadIdx.x, cq = tid % LPT, wv = tid / LPT, tx = wv % G::TWX, hh = wv / G::TWX;
    const W6Map<N> map(nboards);
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < 6 * ROWF / 4; k += NT) reinterpret_cast<f32x4 *>(win)[k] = zero4;      // halo columns stay zero
    for (int k = tid; k < P * P; k += NT) bd[k] = 0.f;
    f32x4 sw[9];
#pragma unroll
    for (int t = 0; t < 9; t++) sw[t] = *reinterpret_cast<const f32x4 *>(S.w + (t * 2 + 1) * C + cq * 4);
    if (tid < LPT) w4_stem_player_sums(S.w, C, tid * 4, s0);
    const f32x4 sb = *reinterpret_cast<const f32x4 *>(S.bias + cq * 4);
    float pv = 0.f;
    // rows y0 .. y0 + nr - 1 of the stem's output into the window (rows outside the board are zeros)
    auto fill = [&](int y0, int nr) {
        for (int k = tid; k < nr * N * LPT; k += NT) {
            const int r = k / (N * LPT), px = (k / LPT) % N, y = y0 + r;
            const bool inb = y >= 0 && y < N;
            const f32x4 v = inb ? w4_stem_pixel<N>(bd, w4_fma(pv, *reinterpret_cast<const f32x4 *>(s0 + w4_border_class(y, px, N) * C + cq * 4), zero4), y, px, sb,
                                                   [&](int t) { return sw[t]; })
                                : zero4;
            *reinterpret_cast<f32x4 *>(win + ((y + 1) % 6) * ROWF + (px + 1) * C + cq * 4) = v;
        }
    };
#pragma unroll 1
    for (int b = blockIdx.x; b < nboards; b += gridDim.x) {
        W6_LDS_BARRIER();                                            // the previous board's last reads (and the zero fills)
        pv = w4_stem_plane<N>(S, b, bd, tid, NT);
        W6_LDS_BARRIER();
        fill(-1, 6);
#pragma unroll 1
        for (int ty = 0; ty < G::TWY; ty++) {
            W6_LDS_BARRIER();                                        // rows 4 ty - 1 .. 4 ty + 4 are in the window
            {
                const long t = map.slot(b, ty, tx);
                float *vb = V + (size_t)(t / W6_BT) * W6_VBLOCK + (size_t)(t % W6_BT) * W6_VTILE;
                int slot[6];
                w6_slots<N>(ty, slot);
                w6_patch_transform(win + G::x0(tx) * C, slot, vb, cq, G::cls(ty, tx), hh, hh + 1);
            }
            if (ty + 1 < G::TWY) {
                W6_LDS_BARRIER();                                    // rows 4 ty - 1 .. 4 ty + 2 are read no more
                fill(4 * ty + 5, 4);
            }
        }
    }
}

// the tail of a region's last block (tiles beyond the batch) is never written by the kernels above: zero it once per launch
template <int N>
__global__ void c6_w6_tail_kernel(float *__restrict__ V, int nboards) {
    using G = W6G<N>;
    const W6Map<N> map(nboards);
    const long first_blk[5] = {0, map.blkB, map.blkBP, map.blkD, map.blkDP},
               count[5] = {(long)nboards * G::NA, (long)nboards * G::NB, (long)nboards * G::NBP, (long)nboards * G::ND, (long)nboards * G::NDP};
    for (int r = 0; r < 5; r++) {
        const int used = (int)(count[r] % W6_BT);
        if (used == 0) continue;
        float *blk = V + (size_t)(first_blk[r] + count[r] / W6_BT) * W6_VBLOCK + (size_t)used * W6_VTILE;
        const size_t n = (size_t)(W6_BT - used) * W6_VTILE;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) blk[i] = 0.f;
    }
}
template <int N> static bool w6_has_tail(int nboards) {
    using G = W6G<N>;
    return ((long)nboards * G::NA) % W6_BT || ((long)nboards * G::NB) % W6_BT || ((long)nboards * G::NBP) % W6_BT || ((long)nboards * G::ND) % W6_BT ||
           ((long)nboards * G::NDP) % W6_BT;
}

// ---- the GEMMs of a block, with the column half of the output transform ("Z rows"): conv_wino4.hip's c6_wino4_gemm_kernel ----------
// over four classes: rows of positions 6 / 5 / 5 / 8, weight sets at 0 / 36 / 66 / 96 C^2, tile slots of 48 positions in V and 32 row
// sums in Z.  The schedule, the hand-issued memory operations and their wait counts are the original's (see there).
constexpr int w6_store_group(int k) { return 8 + k; }
constexpr int w6_wait(int i, bool stores, int nd = 8) {      // nd: slab DMAs per wave and position (8 waves: 8; the 4-wave jobs of tiny batches: 16)
    const int lo = i - W6_PF > 0 ? i - W6_PF : 0, hi = (nd - 1 < i - 1) ? nd - 1 : i - 1;
    const int dmas = hi >= lo ? hi - lo + 1 : 0;
    int st = 0;
    if (stores)
        for (int k = 0; k < 16; k++)
            if (w6_store_group(k) < i && w6_store_group(k) >= i - W6_PF) st++;
    return W6_PF + dmas + st;
}
static_assert(w6_store_group(15) < 32 - W6_PF, "a position's stores are older than every request that is waited for in the next one");
// the LAST position of a split job (below): the weight stream ends with the row and no next position is fetched -- younger than group i's request are
// only the groups i + 1 .. 31 and the result stores issued since
constexpr int w6_wait_last(int i) {
    int st = 0;
    for (int k = 0; k < 16; k++)
        if (w6_store_group(k) < i && w6_store_group(k) >= i - W6_PF) st++;
    return (31 - i < W6_PF ? 31 - i : W6_PF) + st;
}

// DP = false: the blocks [first, nblocks) of the classes with rows of six positions (A, B, B', D: 708 of a board's 788 positions); DP = true: the blocks of
// class D' (rows of five: 80 positions).  Two instantiations, two launches: with both row forms in one kernel the register allocator spilled 121 registers of the accumulators.
//
// SPLIT (r5): the unit of work is a ROW of positions instead of a block (a row's four row sums depend on nothing but its own six -- D': two rows' ten --
// positions).  A block on one CU is a chain of 48 positions x 13.7 us, whatever the batch: one game's eight boards took 0.66 ms per layer on 4 of the
// 256 CUs.  Split, its 28 rows run on 28 CUs at once.  Same instruction stream per position (same waits), only the addresses of the V rows, of the
// weight groups and of Z differ; the launcher takes this form while the blocks do not fill the chip a few times over (c6_w6_dispatch).
// HW (r5, split form only): half-width jobs for the tiniest batches -- a workgroup of FOUR waves multiplies a row for 128 of the 256 output channels
// (blockIdx.z picks the half), one wave per SIMD instead of two: a position takes 6.8 instead of 13.7 us, twice as many jobs.  Each wave fetches 16 of the
// slab's 64 rows (ND = 16: the wait counts take the DMA count as a parameter).  Same per-wave instruction stream: bit-identical sums.
template <bool DP, bool SPLIT, bool HW = false>
__global__ __launch_bounds__(HW ? 256 : 64 * (W6_C / 32)) void c6_w6_gemm_kernel(const float *__restrict__ V, const float *__restrict__ U, float *__restrict__ Zo,
                                                                                 int first, int nblocks, int blkB, int blkBP, int blkD) {
#define W6_BX ((int)blockIdx.x)
#define W6_GX ((int)gridDim.x)
#include "conv_wino6_gemm.inc"
#undef W6_BX
#undef W6_GX
}
template <bool DP, bool SPLIT, bool HW>
__device__ __forceinline__ void c6_w6_gemm_body(const float *__restrict__ V, const float *__restrict__ U, float *__restrict__ Zo, int first, int nblocks, int blkB, int blkBP,
                                                int blkD, const int bx, const int gx) {
#define W6_BX bx
#define W6_GX gx
#include "conv_wino6_gemm.inc"
#undef W6_BX
#undef W6_GX
}
// The half-width jobs of BOTH row forms in one launch (r5): at a handful of games a layer's two GEMM launches took 50 and 42 us side by side on two streams,
// and the fork / join between the streams cost as much again (12 .. 27 us before and after, profiles/r05_kernel_stats_1game.csv).  The first g6 workgroups
// along x run the rows of six, the rest the rows of five of class D' -- two inlined bodies behind a branch that is uniform over the workgroup, each with
// its own instruction stream and wait counts (four waves per workgroup: the register file has room for the larger of the two).
__global__ __launch_bounds__(256) void c6_w6_gemm_half_pair_kernel(const float *__restrict__ V, const float *__restrict__ U, float *__restrict__ Zo, int n6, int nblocks,
                                                                    int blkB, int blkBP, int blkD, int g6) {
    if ((int)blockIdx.x < g6) c6_w6_gemm_body<false, true, true>(V, U, Zo, 0, n6, blkB, blkBP, blkD, (int)blockIdx.x, g6);
    else c6_w6_gemm_body<true, true, true>(V, U, Zo, n6, nblocks, blkB, blkBP, blkD, (int)blockIdx.x - g6, (int)gridDim.x - g6);
}

// (the same for the eight-wave split jobs of small batches: both bodies fit the 256 registers of the larger one)
__global__ __launch_bounds__(64 * (W6_C / 32)) void c6_w6_gemm_split_pair_kernel(const float *__restrict__ V, const float *__restrict__ U, float *__restrict__ Zo, int n6, int nblocks,
                                                                                 int blkB, int blkBP, int blkD, int g6) {
    if ((int)blockIdx.x < g6) c6_w6_gemm_body<false, true, false>(V, U, Zo, 0, n6, blkB, blkBP, blkD, (int)blockIdx.x, g6);
    else c6_w6_gemm_body<true, true, false>(V, U, Zo, n6, nblocks, blkB, blkBP, blkD, (int)blockIdx.x - g6, (int)gridDim.x - g6);
}

// ---- output transform, the half the GEMM kernel left ------------------------------------------------------------------------------
// A tile's outputs (before bias) from its row sums Z_i[b] (i = Z's row, b = 0 .. 3 its second index):
//   A   y[a][b] = sum_i A4^T[a][i] Z_i[b]            (6 rows)      B'  the same with F(3)'s A^T over 5 rows -> output rows 0 .. 2
//   B   stored transposed: Z's row index is the patch COLUMN j (5 of them), b the output ROW: y[b][c] = sum_j A3^T[c][j] Z_j[b]
//   D   transposed, 8 rows: y[b][c] = sum_j A6^T[c][j] Z_j[b],  A6^T = [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 0; 0 1 1 4 4 1/4 1/4 0;
//       0 1 -1 8 -8 1/8 -1/8 0; 0 1 1 16 16 1/16 1/16 0; 0 1 -1 32 -32 1/32 -1/32 1]
// The transform is independent for every b: a SLICE takes the 8 / 6 / 5 row sums Z[.][b] and yields up to six pixels (a row of the tile
// for the transposed classes D and B, a column for A and B').  Pixels beyond the board are dropped by the callers.
template <typename VT>
__device__ __forceinline__ void w6_slice_outputs(const VT (&z)[8], VT (&y)[6], int cls) {
    if (cls == W6_CLS_D || cls == W6_CLS_DP) {
        const VT e1 = z[1] + z[2], o1 = z[1] - z[2], e2 = z[3] + z[4], o2 = z[3] - z[4], e3 = z[5] + z[6], o3 = z[5] - z[6];
        y[0] = (z[0] + e1) + (e2 + e3);
        y[1] = w4_fma(2.f, o2, w4_fma(0.5f, o3, o1));
        y[2] = w4_fma(4.f, e2, w4_fma(0.25f, e3, e1));
        y[3] = w4_fma(8.f, o2, w4_fma(0.125f, o3, o1));
        y[4] = w4_fma(16.f, e2, w4_fma(0.0625f, e3, e1));
        y[5] = w4_fma(32.f, o2, w4_fma(0.03125f, o3, o1)) + z[7];
    } else if (cls == W6_CLS_A) {
        const VT s12 = z[1] + z[2], d12 = z[1] - z[2], s34 = z[3] + z[4], d34 = z[3] - z[4];
        y[0] = z[0] + s12 + s34;
        y[1] = w4_fma(2.f, d34, d12);
        y[2] = w4_fma(4.f, s34, s12);
        y[3] = w4_fma(8.f, d34, d12) + z[5];
    } else {
        const VT s12 = z[1] + z[2], d12 = z[1] - z[2];
        y[0] = z[0] + s12 + z[3];
        y[1] = w4_fma(2.f, z[3], d12);
        y[2] = w4_fma(4.f, z[3], s12) + z[4];
    }
}
// ---- the tower's last output transform, with the heads' 1x1 filters (conv_wino4.hip's c6_wino4_out_kernel<N, true>) -----------------
// One wave per half tile at a time (two slices), lane = 4 channels; a pixel's three filter sums are wave sums; nothing but the heads'
// inputs is written.  Only one slice's pixels and filter sums are alive at a time beside the half tile's row sums and residual.
struct W6OutCtx {
    f32x4 bi, hw0, hw1, hw2;
    const float *res;
    C6HeadFuse hd;
};
// Class D' (the six-wide tiles of the last tile row: F(3) down) shares D's slot layout of 4 row sums per row of positions, of which the
// GEMM kernel fills b = 0, 1 and 3 (c6_w6_gemm_kernel: F(3)'s third row sum lives where F(4)'s fourth does): slice b = 2 does not exist and
// slice 3 is output row 2.
#define W6_SLICE_OK(CLS, b) (!((CLS) == W6_CLS_DP && (b) == 2))
#define W6_SLICE_ROW(CLS, b) ((CLS) == W6_CLS_DP && (b) == 3 ? 2 : (b))
// one tile of class CLS: everything about the class is a constant here (no run-time indexing of the register arrays)
template <int N, int CLS>
__device__ __forceinline__ void w6_out_tile(const W6OutCtx &X, const float *zb, long bd, int ty, int x0, int wd, int lane, int b0) {
    constexpr int C = W6_C;
    constexpr int rows = CLS == W6_CLS_D || CLS == W6_CLS_DP ? 8 : CLS == W6_CLS_A ? 6 : 5;
    constexpr bool along_row = CLS == W6_CLS_D || CLS == W6_CLS_DP || CLS == W6_CLS_B;      // the slice's pixels run along a row of the tile (else: along a column)
    constexpr int npx = CLS == W6_CLS_D || CLS == W6_CLS_DP ? 6 : CLS == W6_CLS_A ? 4 : 3;
    // two slices per wave (b = b0, b0 + 1): their row sums are requested at once (up to 16 KiB in flight per wave), then consumed one by one
    // (the residual first: asked for where it is added, the load would wait behind its slice's arithmetic)
    f32x4 rv[2][npx];
    if (X.res) {
#pragma unroll
        for (int bq = 0; bq < 2; bq++)
#pragma unroll
            for (int k = 0; k < npx; k++) {
                const int py = 4 * ty + (along_row ? W6_SLICE_ROW(CLS, b0 + bq) : k), cx = along_row ? k : b0 + bq;
                rv[bq][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (py < N && cx < wd && W6_SLICE_OK(CLS, b0 + bq)) rv[bq][k] = *reinterpret_cast<const f32x4 *>(X.res + ((size_t)bd * (N * N) + (size_t)(py * N + x0 + cx)) * C + lane * 4);
            }
    }
    f32x4 zt[2][8];
#pragma unroll
    for (int i = 0; i < rows; i++)
#pragma unroll
        for (int bq = 0; bq < 2; bq++) zt[bq][i] = *reinterpret_cast<const f32x4 *>(zb + (size_t)(i * 4 + b0 + bq) * C);
#pragma unroll
    for (int bq = 0; bq < 2; bq++) {
        const int b = b0 + bq;
        if (!W6_SLICE_OK(CLS, b)) continue;      // (uniform over the wave)
        f32x4 y[6];
        w6_slice_outputs(zt[bq], y, CLS);
        float hs[npx][3];
#pragma unroll
        for (int k = 0; k < npx; k++) {
            hs[k][0] = 0.f; hs[k][1] = 0.f; hs[k][2] = 0.f;
            const int py = 4 * ty + (along_row ? W6_SLICE_ROW(CLS, b) : k), cx = along_row ? k : b, px = x0 + cx;
            if (py >= N || cx >= wd) continue;
            f32x4 v = y[k] + X.bi;
            if (X.res) v = v + rv[bq][k];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                v[r] = v[r] > 0.f ? v[r] : 0.f;
                hs[k][0] = __builtin_fmaf(v[r], X.hw0[r], hs[k][0]);
                hs[k][1] = __builtin_fmaf(v[r], X.hw1[r], hs[k][1]);
                hs[k][2] = __builtin_fmaf(v[r], X.hw2[r], hs[k][2]);
            }
        }
#pragma unroll
        for (int k = 0; k < npx; k++)
#pragma unroll
            for (int f = 0; f < 3; f++) hs[k][f] = w4_wave_sum_row3(hs[k][f]);
        // lanes 48 .. have the sums: lane 48 + k writes pixel k of the slice
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < npx; k++)
            if ((lane & 15) == k) { m0 = hs[k][0]; m1 = hs[k][1]; m2 = hs[k][2]; }
        const int k = lane & 15, py = 4 * ty + (along_row ? W6_SLICE_ROW(CLS, b) : k), cx = along_row ? k : b, px = x0 + cx;
        if (lane >= 48 && k < npx && py < N && cx < wd) {
            const int p = py * N + px;
            const float y0 = m0 + X.hd.pc_b[0], y1 = m1 + X.hd.pc_b[1];
            X.hd.ph[(size_t)bd * (2 * N * N) + p] = y0 > 0.f ? y0 : 0.f;             // flatten is channel-major (NCHW view)
            X.hd.ph[(size_t)bd * (2 * N * N) + N * N + p] = y1 > 0.f ? y1 : 0.f;
            const float vv = (m2 + X.hd.vc_b) * X.hd.vn_scale[p] + X.hd.vn_shift[p];
            X.hd.vh[(size_t)bd * (N * N) + p] = vv > 0.f ? vv : 0.f;
        }
    }
}
template <int N>
__global__ __launch_bounds__(256) void c6_w6_out_kernel(const float *__restrict__ M, const float *__restrict__ bias, const float *__restrict__ res,
                                                        long total_tiles, C6HeadFuse hd) {
    using G = W6G<N>;
    constexpr int C = W6_C;
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    W6OutCtx X;
    X.bi = *reinterpret_cast<const f32x4 *>(bias + lane * 4);
    X.hw0 = *reinterpret_cast<const f32x4 *>(hd.w + lane * 4);
    X.hw1 = *reinterpret_cast<const f32x4 *>(hd.w + C + lane * 4);
    X.hw2 = *reinterpret_cast<const f32x4 *>(hd.w + 2 * C + lane * 4);
    X.res = res;
    X.hd = hd;
    const W6Map<N> map(total_tiles / G::T);
#pragma unroll 1
    for (long it = wave0; it < 2 * total_tiles; it += nwaves) {      // item = half a tile: two of its four slices
        const long t = it >> 1;
        const int b0 = (int)(it & 1) * 2;
        const long bd = t / G::T;
        const int tt = (int)(t - bd * G::T), ty = tt / G::TWX, tx = tt - ty * G::TWX, cls = G::cls(ty, tx);
        const int x0 = G::x0(tx), wd = G::width(tx);
        const float *zb = M + (size_t)map.slot(bd, ty, tx) * W6_ZTILE + lane * 4;
        if (cls == W6_CLS_D) w6_out_tile<N, W6_CLS_D>(X, zb, bd, ty, x0, wd, lane, b0);
        else if (cls == W6_CLS_DP) w6_out_tile<N, W6_CLS_DP>(X, zb, bd, ty, x0, wd, lane, b0);
        else if (G::NA && cls == W6_CLS_A) w6_out_tile<N, W6_CLS_A>(X, zb, bd, ty, x0, wd, lane, b0);
        else if (cls == W6_CLS_B) w6_out_tile<N, W6_CLS_B>(X, zb, bd, ty, x0, wd, lane, b0);
        else w6_out_tile<N, W6_CLS_BP>(X, zb, bd, ty, x0, wd, lane, b0);
    }
}

// ---- the boundary kernel with the two halves on different waves ---------------------------------------------------------------------
// 12 waves per workgroup (three per SIMD, 168 registers each), one board at a time:
//   waves 0 .. 7   PRODUCERS: wave = (tile column, channel half), lane = 2 channels.  Z tile -> pixels (A^T Z + bias (+ residual), ReLU),
//                  activations out, the pixels into the rolling 6-row window
//   waves 8 .. 11  CONSUMERS: wave = tile column, lane = 4 channels in two passes: window patch -> B^T d B -> V
// so that one half's memory waits (Z / residual loads; V stores) overlap the other half's arithmetic inside the workgroup -- the two never
// touch the matrix unit, the mutual exclusion profiles/r03_mfma_valu_coissue.txt measured does not apply.
// A board takes six phases, each ending in two LDS barriers (X: the consumers' reads of the phase are done / Y: the producers' writes are):
//   phase p  producers  compute the pixels of tile row p + 1 (p = 4: the zero row below the board; p = 5: tile row 0 of the NEXT board),
//                       after X write the three rows kept from phase p - 1 and the first row of this phase's tile into the window
//            consumers  transform the patches of tile row p - 1 (window rows 4p - 5 .. 4p), p >= 1
template <int N, int CLS, bool RS>
__device__ __forceinline__ void w6_produce_tile(const float *zb, long bd, int t, int x0, int wd, int c, f32x2 bi, const float *res, float *act, const float *plane, float pv,
                                                const float *swl, const float *s0, f32x2 sb, f32x2 (&r0)[6], f32x2 (&kn)[3][6]) {
    constexpr int C = W6_C;
    constexpr int rows = CLS == W6_CLS_D || CLS == W6_CLS_DP ? 8 : CLS == W6_CLS_A ? 6 : 5;
    constexpr bool along_row = CLS == W6_CLS_D || CLS == W6_CLS_DP || CLS == W6_CLS_B;
    constexpr int npx = CLS == W6_CLS_D || CLS == W6_CLS_DP ? 6 : CLS == W6_CLS_A ? 4 : 3;
    const f32x2 zero2 = f32x2{0.f, 0.f};
    if (CLS == W6_CLS_DP) {      // three output rows: the tile's fourth pixel row lies below the board (the next layer's zero padding)
#pragma unroll
        for (int k = 0; k < 6; k++) kn[2][k] = zero2;
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        // the residual of the half's pixels is requested BEFORE the row sums: both are in flight together (asked for where it is added, the
        // load waits behind the arithmetic of its slice -- a layer with a residual took 16.5 ms instead of 8.4)
        f32x2 rv[2][along_row ? 6 : 4];
        if (!RS && res) {
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int k = 0; k < (along_row ? 6 : 4); k++) {
                    const int a = along_row ? W6_SLICE_ROW(CLS, 2 * h + q) : k, cx = along_row ? k : 2 * h + q, py = 4 * t + a;
                    rv[q][k] = zero2;
                    if (cx < wd && k < npx && py < N && W6_SLICE_OK(CLS, 2 * h + q)) rv[q][k] = *reinterpret_cast<const f32x2 *>(res + ((size_t)bd * (N * N) + (size_t)(py * N + x0 + cx)) * C + c);
                }
        }
        f32x2 zt[2][8];
#pragma unroll
        for (int i = 0; i < rows; i++)
#pragma unroll
            for (int q = 0; q < 2; q++) zt[q][i] = *reinterpret_cast<const f32x2 *>(zb + (size_t)(i * 4 + 2 * h + q) * C);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int b = 2 * h + q;
            if (!W6_SLICE_OK(CLS, b)) continue;
            f32x2 y[6];
            w6_slice_outputs(zt[q], y, CLS);
#pragma unroll
            for (int k = 0; k < (along_row ? 6 : 4); k++) {
                const int a = along_row ? W6_SLICE_ROW(CLS, b) : k, cx = along_row ? k : b;      // pixel (row a, column cx) of the tile
                if (cx >= wd) continue;                                        // (uniform over the wave)
                const int py = 4 * t + a;
                f32x2 v = zero2;
                if (k < npx && py < N) {
                    v = y[k < npx ? k : 0] + bi;
                    const size_t o = ((size_t)bd * (N * N) + (size_t)(py * N + x0 + cx)) * C + c;
                    if (RS) {
                        constexpr int P = N + 2;
                        const f32x2 s0v = *reinterpret_cast<const f32x2 *>(s0 + w4_border_class(py, x0 + cx, N) * C + c);
                        f32x2 acc = sb + w4_fma(pv, s0v, zero2);              // (w4_stem_pixel's order of operations)
#pragma unroll
                        for (int tp = 0; tp < 9; tp++) {
                            const float bdv = plane[(py + tp / 3) * P + x0 + cx + tp % 3];
                            const f32x2 w = *reinterpret_cast<const f32x2 *>(swl + tp * C + c);
                            acc[0] = __builtin_fmaf(w[0], bdv, acc[0]);
                            acc[1] = __builtin_fmaf(w[1], bdv, acc[1]);
                        }
                        v[0] += acc[0] > 0.f ? acc[0] : 0.f;
                        v[1] += acc[1] > 0.f ? acc[1] : 0.f;
                    } else if (res) v = v + rv[q][k];
                    v[0] = v[0] > 0.f ? v[0] : 0.f;
                    v[1] = v[1] > 0.f ? v[1] : 0.f;
                    if (act) *reinterpret_cast<f32x2 *>(act + o) = v;      // a block's output is also the next block's residual
                }                                                          // (else: beyond the board, the next layer's zero padding)
                if (a == 0) r0[cx] = v;
                else kn[a - 1][cx] = v;
            }
        }
    }
}
#define W6_MID_THREADS(N) (192 * W6G<N>::TWX)      // 19 x 19: 8 + 4 waves; 15 x 15: 6 + 3
template <int N, bool RS>
__global__ __launch_bounds__(W6_MID_THREADS(N)) void c6_w6_mid_kernel(const float *__restrict__ M, const float *__restrict__ bias, const float *__restrict__ res,
                                                                      float *__restrict__ act, float *__restrict__ V, int nboards, C6StemSrc S) {
    using G = W6G<N>;
    constexpr int C = W6_C, ROWF = G::ROWF, NT = W6_MID_THREADS(N), PP = (N + 2) * (N + 2), TWX = G::TWX, TWY = G::TWY;
    extern __shared__ __attribute__((aligned(16))) float win[];                // [6][COLS][C] (+ RS: the stem's weights, sums and two board planes)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = w < 2 * TWX;
    const int tx = producer ? w % TWX : w - 2 * TWX;
    const int x0 = G::x0(tx), wd = G::width(tx);
    const int c = (producer ? w / TWX : 0) * (C / 2) + lane * 2;                         // a producer's channel pair
    const W6Map<N> map(nboards);
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x2 zero2 = f32x2{0.f, 0.f};
    const f32x2 bi = bias ? *reinterpret_cast<const f32x2 *>(bias + c) : zero2;
    for (int k = tid; k < 6 * ROWF / 4; k += NT) reinterpret_cast<f32x4 *>(win)[k] = zero4;      // halo columns stay zero
    float *wpix = win + (x0 + 1) * C + c;                  // window column of this producer's first pixel column (board column x0)
    float *swl = win + 6 * ROWF, *s0 = swl + 9 * C, *planes = s0 + 9 * C;      // RS: board-plane weights [9][C], player-plane sums [9][C], board planes [2][(N+2)^2]
    f32x2 sb = zero2;
    if (RS) {
        for (int k = tid; k < 9 * C / 4; k += NT) reinterpret_cast<f32x4 *>(swl)[k] = *reinterpret_cast<const f32x4 *>(S.w + ((k / (C / 4)) * 2 + 1) * C + (k % (C / 4)) * 4);
        if (tid < C / 4) w4_stem_player_sums(S.w, C, tid * 4, s0);
        for (int k = tid; k < 2 * PP; k += NT) planes[k] = 0.f;
        sb = *reinterpret_cast<const f32x2 *>(S.bias + c);
    }
    long cur = -1, nxt = blockIdx.x;
    int p = TWY;
    W6_LDS_BARRIER();
    // The two roles run separate loops (nothing of one role's state is alive in the other's registers); both pass the same barriers:
    // X = this phase's window reads are done, Y = the window holds the next phase's rows.
    if (producer) {
        constexpr int NP = 128 * TWX;                        // the producers' threads: tid 0 .. NP - 1
        int pbuf = 0;
        float pvc = 0.f, pvn = 0.f;
        if (RS && nxt < nboards) pvn = w4_stem_plane<N>(S, nxt, planes + (pbuf ^ 1) * PP, tid, NP);
        W6_LDS_BARRIER();
        f32x2 r0[6], ko[3][6], kn[3][6];
#pragma unroll
        for (int k = 0; k < 6; k++) { r0[k] = zero2; ko[0][k] = zero2; ko[1][k] = zero2; ko[2][k] = zero2; kn[0][k] = zero2; kn[1][k] = zero2; kn[2][k] = zero2; }
#pragma unroll 1
        for (;;) {
            const long b = p == TWY ? nxt : cur;
            const int t = p == TWY ? 0 : p + 1;
            if (p != TWY - 1 && b >= 0 && b < nboards) {
                const int cls = G::cls(t, tx);
                const float *zb = M + (size_t)map.slot(b, t, tx) * W6_ZTILE + c;
                const float *plane = planes + (p == TWY ? pbuf ^ 1 : pbuf) * PP;
                const float pv = p == TWY ? pvn : pvc;
                if (cls == W6_CLS_D) w6_produce_tile<N, W6_CLS_D, RS>(zb, b, t, x0, wd, c, bi, res, act, plane, pv, swl, s0, sb, r0, kn);
                else if (cls == W6_CLS_DP) w6_produce_tile<N, W6_CLS_DP, RS>(zb, b, t, x0, wd, c, bi, res, act, plane, pv, swl, s0, sb, r0, kn);
                else if (G::NA && cls == W6_CLS_A) w6_produce_tile<N, W6_CLS_A, RS>(zb, b, t, x0, wd, c, bi, res, act, plane, pv, swl, s0, sb, r0, kn);
                else if (cls == W6_CLS_B) w6_produce_tile<N, W6_CLS_B, RS>(zb, b, t, x0, wd, c, bi, res, act, plane, pv, swl, s0, sb, r0, kn);
                else w6_produce_tile<N, W6_CLS_BP, RS>(zb, b, t, x0, wd, c, bi, res, act, plane, pv, swl, s0, sb, r0, kn);
            } else if (p == TWY - 1) {
#pragma unroll
                for (int k = 0; k < 6; k++) r0[k] = zero2;                     // row 4 TWY: below the board
                if (RS && nxt < nboards) pvn = w4_stem_plane<N>(S, nxt, planes + (pbuf ^ 1) * PP, tid, NP);      // (last read in phase 3 of the board before)
            }
            W6_LDS_BARRIER();                                // X
            if (p == TWY) {
#pragma unroll
                for (int k = 0; k < 6; k++)
                    if (k < wd) {
                        *reinterpret_cast<f32x2 *>(wpix + 0 * ROWF + k * C) = zero2;      // slot 0 = board row -1
                        *reinterpret_cast<f32x2 *>(wpix + 1 * ROWF + k * C) = r0[k];      // slot 1 = row 0
                    }
            } else {
#pragma unroll
                for (int a = 1; a < 4; a++)
#pragma unroll
                    for (int k = 0; k < 6; k++)
                        if (k < wd) *reinterpret_cast<f32x2 *>(wpix + ((4 * p + a + 1) % 6) * ROWF + k * C) = ko[a - 1][k];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    if (k < wd) *reinterpret_cast<f32x2 *>(wpix + ((4 * p + 5) % 6) * ROWF + k * C) = r0[k];
            }
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int k = 0; k < 6; k++) ko[a][k] = kn[a][k];
            W6_LDS_BARRIER();                                // Y
            if (p == TWY) {
                if (nxt >= nboards) break;
                cur = nxt;
                nxt += gridDim.x;
                pvc = pvn;
                pbuf ^= 1;
                p = 0;
            } else p++;
        }
    } else {
        W6_LDS_BARRIER();
#pragma unroll 1
        for (;;) {
            if (p >= 1 && cur >= 0) {
                const int ty = p - 1;
                const long tt = map.slot(cur, ty, tx);
                float *vb = V + (size_t)(tt / W6_BT) * W6_VBLOCK + (size_t)(tt % W6_BT) * W6_VTILE;
                int slot[6];
                w6_slots<N>(ty, slot);
                w6_patch_transform(win + x0 * C, slot, vb, lane, G::cls(ty, tx));
            }
            W6_LDS_BARRIER();                                // X
            W6_LDS_BARRIER();                                // Y
            if (p == TWY) {
                if (nxt >= nboards) break;
                cur = nxt;
                nxt += gridDim.x;
                p = 0;
            } else p++;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
bool c6_w6_available(int C, int N, int blocks) {
    // (C6_W6_N15=0: a test hook -- 15 x 15 towers stay on conv_wino4.hip's F(4x4) tiling, which the trainer and the 9 x 9 / two-block towers use anyway;
    //  tests/test_gpu_variants.py compares the two tilings' outputs and keeps conv_wino4.hip's work decompositions covered at 15 x 15)
    static const bool n15 = [] { const char *e = getenv("C6_W6_N15"); return !(e && e[0] == '0'); }();
    return C == W6_C && (N == 19 || (N == 15 && n15)) && blocks - 1 >= 2;
}
size_t c6_w6_packed_floats(void) { return (size_t)W6_SETS_POS * W6_C * W6_C; }
static long w6_nblocks(int N, int nboards) { return N == 19 ? W6Map<19>(nboards).nblocks : W6Map<15>(nboards).nblocks; }
// V and Z, one behind the other
size_t c6_w6_vbuf_floats(int N, int nboards) { return (size_t)w6_nblocks(N, nboards) * (W6_VBLOCK + W6_ZBLOCK); }

// Transformed weights U = G_v g G_h^T (double, rounded once), four sets one behind the other: A (36 positions p = 6 i + j: F(4)'s G both
// ways), B (30, p = 6 j + i: G4 vertically, G3 horizontally), B' (30, p = 6 i + j: G3 vertically, G4 horizontally), D (48, p = 6 j + i: G4
// vertically, F(6)'s G horizontally), D' (40, p = 5 j + i: G3 vertically, F(6)'s G horizontally).  A set is packed [wave][position][group u][lane][r]: U_p[cout = 32 wave + (lane & 31)][cin = 8u + 4(lane >> 5) + r].
void c6_w6_pack_weights(const float *w, float *packed) {
    const double G4[6][3] = {{1.0 / 4, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const double G3[5][3] = {{1.0 / 2, 0.0, 0.0}, {-1.0 / 2, -1.0 / 2, -1.0 / 2}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0.0, 0.0, 1.0}};
    const double G6[8][3] = {{1.0, 0.0, 0.0},
                             {-2.0 / 9, -2.0 / 9, -2.0 / 9},
                             {-2.0 / 9, 2.0 / 9, -2.0 / 9},
                             {1.0 / 90, 1.0 / 45, 2.0 / 45},
                             {1.0 / 90, -1.0 / 45, 2.0 / 45},
                             {32.0 / 45, 16.0 / 45, 8.0 / 45},
                             {32.0 / 45, -16.0 / 45, 8.0 / 45},
                             {0.0, 0.0, 1.0}};
    constexpr int C = W6_C, NG = C / 8;
    const int npos[5] = {36, 30, 30, 48, 40}, first[5] = {0, 36, 66, 96, 144};
    for (int set = 0; set < 5; set++)
        for (int wv = 0; wv < C / 32; wv++)
            for (int pos = 0; pos < npos[set]; pos++) {
                const int L = set == 4 ? 5 : 6, p = pos / L, q = pos % L;      // D': rows of five positions, p = 5 j + i
                const double *gv = set == 0 ? G4[p] : set == 1 ? G4[q] : set == 2 ? G3[p] : set == 3 ? G4[q] : G3[q];      // the row of G on the vertical side
                const double *gh = set == 0 ? G4[q] : set == 1 ? G3[p] : set == 2 ? G4[q] : G6[p];                          // ... and on the horizontal side
                for (int u = 0; u < NG; u++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int r = 0; r < 4; r++) {
                            const int co = wv * 32 + (lane & 31), ci = 8 * u + 4 * (lane >> 5) + r;
                            const float *g = w + ((size_t)co * C + ci) * 9;
                            double sum = 0.0;
                            for (int a = 0; a < 3; a++)
                                for (int b = 0; b < 3; b++) sum += gv[a] * (double)g[a * 3 + b] * gh[b];
                            packed[(size_t)first[set] * C * C + ((((size_t)wv * npos[set] + pos) * NG + u) * 64 + lane) * 4 + r] = (float)sum;
                        }
            }
}

// the side stream of a caller's stream (small batches: the D' GEMM launch beside the other one), made on first use
#include <map>
struct W6Side { hipStream_t stream; hipEvent_t fork, join; };
static W6Side *w6_side_of(hipStream_t s) {
    static std::mutex mu;
    static std::map<std::pair<hipStream_t, int>, W6Side> sides;
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    auto it = sides.find({s, dev});
    if (it != sides.end()) return &it->second;
    W6Side x{};
    if (hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x.join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;      // no side stream: the launches stay in order on the caller's stream
    }
    return &sides.emplace(std::make_pair(s, dev), x).first->second;
}

// steps: 1 the first layer's input transform from the boards, 2 the GEMMs, 16 the boundary kernel, 4 the last output transform + heads
template <int N>
static bool w6_dispatch(int nboards, const float *wp, const float *bias, const float *res, float *act, float *vbuf, int num_cus, hipStream_t s, int steps,
                        const C6HeadFuse *heads, const C6StemSrc *stem) {
    using G = W6G<N>;
    if (nboards <= 0) return true;
    const W6Map<N> map(nboards);
    const int nblocks = (int)map.nblocks;
    const bool tail = w6_has_tail<N>(nboards);
    float *Vb = vbuf, *Mb = vbuf + (size_t)nblocks * W6_VBLOCK;
    constexpr size_t win_smem = (size_t)6 * G::ROWF * 4;
    const int grid_b = nboards < num_cus ? nboards : num_cus;
    if (steps & 1) {
        constexpr size_t smem = win_smem + (size_t)9 * W6_C * 4 + (size_t)(N + 2) * (N + 2) * 4;
        w4_smem_once(reinterpret_cast<const void *>(&c6_w6_stem_in_kernel<N>), (int)smem);
        hipLaunchKernelGGL(c6_w6_stem_in_kernel<N>, dim3(grid_b), dim3(W6_THREADS(N)), smem, s, *stem, Vb, nboards);
        if (tail) hipLaunchKernelGGL(c6_w6_tail_kernel<N>, dim3(256), dim3(256), 0, s, Vb, nboards);
    }
    if (steps & 2) {
        constexpr size_t smem = (size_t)2 * W6_BT * (W6_C + 4) * 4;
        const int n6 = (int)map.blkDP, n5 = nblocks - n6;      // blocks of the classes with rows of six positions / of class D'
        w4_smem_once(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<false, false>), (int)smem);
        w4_smem_once(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<true, false>), (int)smem);
        w4_smem_once(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<false, true>), (int)smem);
        w4_smem_once(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<true, true>), (int)smem);
        // r5 -- rows instead of blocks as the unit of work while a CU gets fewer than eight blocks (up to ~800 games): a block is a chain of 36 .. 48
        // positions on ONE CU whatever the batch (one game: 0.66 ms per layer on 4 CUs; split: 28 rows on 28 CUs), and with a few blocks per CU the block form
        // rounds up (256 games = 2.5 blocks per CU: 18.3 ms per round against 13.8).  From ~1000 games on the block form is as fast or faster (no prologue per
        // row: 4096 games 202.9 against 207.4 ms) -- profiles/r05_split_threshold.txt
        // (C6_W6_SPLIT = 0 / 1: a test hook that forces one form -- tests/test_gpu_net.py compares the two bit for bit)
        static const int force = [] { const char *e = getenv("C6_W6_SPLIT"); return e ? atoi(e) : -1; }();
        const bool split = force >= 0 ? force != 0 : n6 < 8 * num_cus;
        // the tiniest batches (a handful of games: twice the jobs still leave CUs idle): half-width jobs of four waves, a position in half the time
        static const int force_hw = [] { const char *e = getenv("C6_W6_HALF"); return e ? atoi(e) : -1; }();      // (test hook, as C6_W6_SPLIT)
        const bool half = split && (force_hw >= 0 ? force_hw != 0 : 2 * 8 * (n6 + n5) <= num_cus);
        // The split forms run BOTH row forms in one launch (c6_w6_gemm_split_pair_kernel / c6_w6_gemm_half_pair_kernel): D' -- a tenth of the layer's
        // products -- takes the CUs the other classes' jobs leave idle without a second stream.  (As two launches on two streams, fork / join by events,
        // the layer had been 8.9 -> 5.5 ms per round at 64 games against the launches in order; the fork / join itself cost 12 .. 27 us either side of the
        // GEMMs: one launch, 1 game 1.11 -> 0.95 ms per round, 8 games 1.33 -> 1.21, 64 .. 768 games -1 .. 2 %, profiles/r05_pair_ab.txt.)
        // C6_W6_PAIR=0 (test / A-B hook): the split form as the two launches it was.  The block form at a small batch (C6_W6_SPLIT=0 only) keeps the side
        // stream; at the benchmark's batch the chip is full either way and the launches stay in order on the caller's stream (21.7 + 2.5 ms).
        static const int force_pair = [] { const char *e = getenv("C6_W6_PAIR"); return e ? atoi(e) : -1; }();
        const bool pair = half || (split && force_pair != 0);
        W6Side *side = !pair && n5 > 0 && n6 < 2 * num_cus ? w6_side_of(s) : nullptr;
        if (side) hipEventRecord(side->fork, s);
        const int g6 = n6 < num_cus ? n6 : num_cus, g5 = n5 < num_cus ? n5 : num_cus;
        if (half) {
            w4_smem_once(reinterpret_cast<const void *>(&c6_w6_gemm_half_pair_kernel), (int)smem);
            hipLaunchKernelGGL(c6_w6_gemm_half_pair_kernel, dim3(g6 + g5, 8, 2), dim3(256), smem, s, Vb, wp, Mb, n6, nblocks, (int)map.blkB, (int)map.blkBP, (int)map.blkD, g6);
        } else if (pair) {
            w4_smem_once(reinterpret_cast<const void *>(&c6_w6_gemm_split_pair_kernel), (int)smem);
            hipLaunchKernelGGL(c6_w6_gemm_split_pair_kernel, dim3(g6 + g5, 8), dim3(64 * (W6_C / 32)), smem, s, Vb, wp, Mb, n6, nblocks, (int)map.blkB, (int)map.blkBP, (int)map.blkD, g6);
        } else if (split) {
            hipLaunchKernelGGL((c6_w6_gemm_kernel<false, true>), dim3(n6 < num_cus ? n6 : num_cus, 8), dim3(64 * (W6_C / 32)), smem, s, Vb, wp, Mb, 0, n6, (int)map.blkB,
                               (int)map.blkBP, (int)map.blkD);
        } else
            hipLaunchKernelGGL((c6_w6_gemm_kernel<false, false>), dim3(n6 < num_cus ? n6 : num_cus), dim3(64 * (W6_C / 32)), smem, s, Vb, wp, Mb, 0, n6, (int)map.blkB, (int)map.blkBP,
                               (int)map.blkD);
        if (n5 > 0 && !pair) {
            hipStream_t s5 = side ? side->stream : s;
            if (side) hipStreamWaitEvent(s5, side->fork, 0);
            if (split)
                hipLaunchKernelGGL((c6_w6_gemm_kernel<true, true>), dim3(n5 < num_cus ? n5 : num_cus, 8), dim3(64 * (W6_C / 32)), smem, s5, Vb, wp, Mb, n6, nblocks, (int)map.blkB,
                                   (int)map.blkBP, (int)map.blkD);
            else
                hipLaunchKernelGGL((c6_w6_gemm_kernel<true, false>), dim3(n5 < num_cus ? n5 : num_cus), dim3(64 * (W6_C / 32)), smem, s5, Vb, wp, Mb, n6, nblocks, (int)map.blkB, (int)map.blkBP,
                                   (int)map.blkD);
            if (side) {
                hipEventRecord(side->join, s5);
                hipStreamWaitEvent(s, side->join, 0);
            }
        }
    }
    if (steps & 16) {
        constexpr size_t rs_smem = win_smem + (size_t)18 * W6_C * 4 + (size_t)(N + 2) * (N + 2) * 4;
        if (stem) {      // the residual is the stem's output, recomputed from the boards
            constexpr size_t rs12 = rs_smem + (size_t)(N + 2) * (N + 2) * 4;
            w4_smem_once(reinterpret_cast<const void *>(&c6_w6_mid_kernel<N, true>), (int)rs12);
            hipLaunchKernelGGL((c6_w6_mid_kernel<N, true>), dim3(grid_b), dim3(W6_MID_THREADS(N)), rs12, s, Mb, bias, res, act, Vb, nboards, *stem);
        } else {
            w4_smem_once(reinterpret_cast<const void *>(&c6_w6_mid_kernel<N, false>), (int)win_smem);
            hipLaunchKernelGGL((c6_w6_mid_kernel<N, false>), dim3(grid_b), dim3(W6_MID_THREADS(N)), win_smem, s, Mb, bias, res, act, Vb, nboards, C6StemSrc{});
        }
        // (no c6_w6_tail_kernel here: the boundary kernel writes the tiles of the batch and nothing else, so the tail the pass's first step zeroed in this
        //  same V is still zero -- five launches less per pass, 5 us + a launch gap each, which a one-game round of 0.95 ms does notice)
    }
    if (steps & 4) {
        const long tiles = (long)nboards * G::T, want = (2 * tiles + 3) / 4;
        const int grid_out = (int)(want < (long)num_cus * 8 ? want : (long)num_cus * 8);
        hipLaunchKernelGGL(c6_w6_out_kernel<N>, dim3(grid_out), dim3(256), 0, s, Mb, bias, res, tiles, *heads);
    }
    const hipError_t le = hipGetLastError();      // a rejected launch (e.g. its LDS request): noted for c6_launch_status, and the caller stops the pass
    c6_note_hip(le, "launch of a c6_w6 kernel");
    return le == hipSuccess;
}
bool c6_w6_dispatch(int N, int nboards, const float *wp, const float *bias, const float *res, float *act, float *vbuf, int num_cus, hipStream_t s, int steps,
                    const C6HeadFuse *heads, const C6StemSrc *stem) {
    return N == 19 ? w6_dispatch<19>(nboards, wp, bias, res, act, vbuf, num_cus, s, steps, heads, stem)
                   : w6_dispatch<15>(nboards, wp, bias, res, act, vbuf, num_cus, s, steps, heads, stem);
}
// matrix FLOPs the GEMM kernel executes for `nboards` boards (whole 64-tile blocks x 36 / 30 / 30 / 48 / 40 positions x C x C MACs)
template <int N> static double w6_exec_flops(int nboards) {
    const W6Map<N> m(nboards);
    return 2.0 * (36.0 * m.blkB + 30.0 * (m.blkD - m.blkB) + 48.0 * (m.blkDP - m.blkD) + 40.0 * (m.nblocks - m.blkDP)) * W6_BT * (double)W6_C * W6_C;
}
double c6_w6_exec_flops(int N, int nboards) { return N == 19 ? w6_exec_flops<19>(nboards) : w6_exec_flops<15>(nboards); }
