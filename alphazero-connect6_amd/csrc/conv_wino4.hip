// conv_wino4.hip -- FP32 3x3 convolution by Winograd F(4x4, 3x3) on the FP32 matrix cores, C = 256 (the benchmark's tower).
//
// Same layer and the same f32 arithmetic as conv_wino.hip (conv3x3 C -> C, pad 1, + folded-BN bias (+ residual) + ReLU;
// AlphaZero/cpp_weighted/model.hpp:61-78; f32 operands, exact f32 products, f32 accumulate), with 6x6 patches / 4x4 outputs:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A      (points 0, +-1, +-2, inf: Lavin & Gray's matrices)
// 25 tiles x 36 positions = 900 products per (cin, cout) and 19x19 board instead of 1600 for F(2x2) and 3249 direct -- 846 with F(3,3)
// along the short side of the last tile row and column (W4Geom::MIX): 3.8x fewer matrix FLOPs than the direct form.  The price: a tile has 16 outputs per (tile, cout), so the output transform cannot be
// accumulated in registers position by position as conv_wino.hip does (16 x 8192 accumulators per wave).  Its column half can, one
// patch row at a time: 24 row sums Z per (tile, cout) go to memory and a boundary kernel applies the row half.  The kernels:
//   c6_wino4_gemm_kernel  Z[block][tile][row i][b][cout] = sum_j A^T[b][j] sum_cin U_ij[cout][cin] V[block][tile][ij][cin]:
//                         conv_wino.hip's schedule (persistent workgroups, wave w owns couts [32w, 32w+32) x 64 tiles, weights L2 ->
//                         VGPR through a ring of 8 groups, V slabs by LDS-DMA one position ahead, one barrier per position); positions
//                         j = 0 and j = 5 of a row accumulate straight into Z, j = 1 .. 4 aside + a few packed vector instructions;
//                         a row's Z is stored while the next row accumulates.  MFMA bound, at the part's power limit
//   c6_wino4_in_kernel    activations NHWC -> V[block][tile 64][pos 36][C]      HBM bound: reads 1x, writes 36/16 x the activations
//   c6_wino4_out_kernel   A^T Z + bias (+ residual), ReLU -> activations NHWC   HBM bound: reads 24/16 x + 1x, writes 1x;
//                         as the tower's last kernel it applies the heads' 1x1 filters instead of storing activations
//   c6_wino4_mid_kernel   out of layer l + in of layer l + 1 in one pass through an LDS window: what runs between two GEMM kernels
//                         inside the net (in / out are left with the tower's first and last layer, and with a layer on its own)
// Rounding: a layer's error is ~6x that of the direct f32 kernel (2.4e-6 rms of activations of order 1); over the benchmark's net
// the outputs move by < 1e-7 (tests/test_gpu_net.py holds every arithmetic to 1e-5 against the oracle and a float64 PyTorch model).
#include "c6_wino.h"
#include <cstdlib>


#define W4_BT 64        // tiles per block
#define W4_NP 36        // positions of a 6x6 patch
#ifndef W4_PF
#define W4_PF 7         // weight groups in flight ahead of the one being multiplied (ring of W4_PF + 1: 8 or 16)
#endif
#define W4_C 256

template <int N>
struct W4Geom {
    static constexpr int C = W4_C;
    static constexpr int TW = (N + 3) / 4;          // tiles per board side (5 / 4 / 3)
    static constexpr int T = TW * TW;
    static constexpr int CP = C + 4;                // LDS row: conflict-free ds_read_b128 with one tile per lane
    static constexpr int SLAB = W4_BT * CP;         // floats of one (block, position) slab in LDS
    static constexpr int POS_STRIDE = C;            // V in memory: [block][tile][pos][C]; Z: [block][tile][row 6][b 4][C]
    static constexpr int TILE_STRIDE = W4_NP * C;
    static constexpr size_t BLOCK_FLOATS = (size_t)W4_BT * W4_NP * C;
    static constexpr int WAVES = C / 32;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int NG = C / 8;
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 4;
    // Mixed tiling: when N = 4 (TW - 1) + 3 (19, 15) the last tile row / column has 3 output rows / columns, and F(3,3) -- 5 patch rows /
    // columns, points 0, +-1, 2, inf -- covers them with 30 instead of 36 positions per tile: 846 instead of 900 positions per 19x19 board.
    // Three classes of tiles, each in its own region of V / Z so that a GEMM block is of one class:
    //   A   ty, tx < TW - 1          F(4) x F(4), 36 positions  p = 6 i + j
    //   B   tx = TW - 1, ty < TW - 1  F(4) vertically, F(3) horizontally: 6 x 5 positions, stored TRANSPOSED, p = 6 j + i (j < 5)
    //   B'  ty = TW - 1              F(3) vertically, F(4) horizontally: 5 x 6 positions, p = 6 i + j (i < 5); the corner tile is of this class
    // so that for every class the 6-long direction runs inside a "row" of positions: the GEMM kernel's register half of the output
    // transform is the same F(4) A^T everywhere, and B / B' blocks simply have 5 rows.
    static constexpr bool MIX = (N % 4 == 3);
    static constexpr int NA = MIX ? (TW - 1) * (TW - 1) : T, NB = MIX ? TW - 1 : 0, NBP = MIX ? TW : 0;      // tiles of a board per class
};
#define W4_CLS_A 0
#define W4_CLS_B 1
#define W4_CLS_BP 2
#define W4_NP_EDGE 30   // positions of a B / B' tile
// blocks of the three regions for `nboards` boards
template <int N>
struct W4Map {
    long blkB, blkBP, nblocks;      // first block of region B, of region B', and the number of blocks
    __host__ __device__ explicit W4Map(long nboards) {
        using G = W4Geom<N>;
        blkB = (nboards * G::NA + W4_BT - 1) / W4_BT;
        blkBP = blkB + (nboards * G::NB + W4_BT - 1) / W4_BT;
        nblocks = blkBP + (nboards * G::NBP + W4_BT - 1) / W4_BT;
    }
    __device__ __forceinline__ static int cls(int ty, int tx) {
        using G = W4Geom<N>;
        return !G::MIX ? W4_CLS_A : ty == G::TW - 1 ? W4_CLS_BP : tx == G::TW - 1 ? W4_CLS_B : W4_CLS_A;
    }
    // the tile's slot: block slot / 64, tile slot % 64 of the block
    __device__ __forceinline__ long slot(long bd, int ty, int tx) const {
        using G = W4Geom<N>;
        if (!G::MIX) return bd * G::T + ty * G::TW + tx;
        if (ty == G::TW - 1) return blkBP * W4_BT + bd * G::NBP + tx;
        if (tx == G::TW - 1) return blkB * W4_BT + bd * G::NB + ty;
        return bd * G::NA + ty * (G::TW - 1) + tx;
    }
};


// ---- input transform: B^T d B, B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1] ----
// One workgroup per board at a time; the board's rows pass through a rolling 6-row window in LDS (row y in slot (y + 1) mod 6,
// columns x = -1 .. 4 TW with a zero halo): every activation is read from HBM once.  Per tile row the four new rows are prefetched
// into registers while the tiles of the current row are transformed -- 64 lanes per tile, each 4 channels; a result row i of the
// patch is formed from the window (B^T's rows have 3-4 non-zeros) and leaves as six 1-KiB runs.  (An LDS-free form -- a wave per tile,
// the 36 pixels straight from memory, the patches' overlap left to the caches -- fetched 21 GB instead of 13 for the 12 GB of
// activations and took 8.9-9.6 ms instead of 8.0 at 32 768 boards, however its tiles were dealt to the waves: a tile per wave from a
// global queue, a tile row of a board per workgroup, a whole board per workgroup.)
#define W4_IN_THREADS 320
// B^T d B of one patch -> the tile's positions in V.  The patch's rows lie in the window slots slot[0 .. 5] (float offsets), wc = its
// first column (channel 0), vt = the tile in V (channel 0).  A lane transforms 4 channels as two passes of 2 -- channels 2 lane, 2 lane + 1
// of each half of C -- so that the whole patch of a pass fits in registers (36 x 2) and every window value is read from LDS ONCE: formed
// one patch row at a time from the window (4 channels per lane, 132 reads of 16 bytes per tile) the LDS reads alone were a third of
// the kernels' time.  A pass reads and stores 512 contiguous bytes per wave instruction.  cls is uniform over the wave.
template <int C>
__device__ __forceinline__ void w4_patch_transform(const float *wc, const int (&slot)[6], float *vt, int lane, int cls) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int co = h * (C / 2) + lane * 2;
        const float *wl = wc + co;
        float *vl = vt + co;
#define W4_D(a, b) (*reinterpret_cast<const f32x2 *>(wl + slot[a] + (b) * C))
#define W4_VST(pos, val) *reinterpret_cast<f32x2 *>(vl + (size_t)(pos) * C) = (val)
        if (cls == W4_CLS_A) {
            f32x2 t[6][6];      // t[i][b] = sum_a B^T[i][a] d[a][b]
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const f32x2 d0 = W4_D(0, b), d1 = W4_D(1, b), d2 = W4_D(2, b), d3 = W4_D(3, b), d4 = W4_D(4, b), d5 = W4_D(5, b);
#pragma unroll
                for (int i = 0; i < 6; i++) t[i][b] = W4_BT_ROW(i, d0, d1, d2, d3, d4, d5);
            }
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int jj = 0; jj < 6; jj++) W4_VST(i * 6 + jj, W4_BT_ROW(jj, t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5]));
        } else if (cls == W4_CLS_BP) {      // 5 patch rows (F(3) vertically) x 6 columns
            f32x2 t[5][6];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const f32x2 d0 = W4_D(0, b), d1 = W4_D(1, b), d2 = W4_D(2, b), d3 = W4_D(3, b), d4 = W4_D(4, b);
#pragma unroll
                for (int i = 0; i < 5; i++) t[i][b] = W4_BT3_ROW(i, d0, d1, d2, d3, d4);
            }
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int jj = 0; jj < 6; jj++) W4_VST(i * 6 + jj, W4_BT_ROW(jj, t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5]));
        } else {                            // 6 patch rows x 5 columns (F(3) horizontally), positions stored transposed
            f32x2 t[6][5];
#pragma unroll
            for (int b = 0; b < 5; b++) {
                const f32x2 d0 = W4_D(0, b), d1 = W4_D(1, b), d2 = W4_D(2, b), d3 = W4_D(3, b), d4 = W4_D(4, b), d5 = W4_D(5, b);
#pragma unroll
                for (int i = 0; i < 6; i++) t[i][b] = W4_BT_ROW(i, d0, d1, d2, d3, d4, d5);
            }
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int jj = 0; jj < 5; jj++) W4_VST(jj * 6 + i, W4_BT3_ROW(jj, t[i][0], t[i][1], t[i][2], t[i][3], t[i][4]));
        }
#undef W4_D
#undef W4_VST
        __builtin_amdgcn_sched_barrier(0);      // one pass at a time
    }
}
template <int N>
__global__ __launch_bounds__(W4_IN_THREADS) void c6_wino4_in_kernel(const float *__restrict__ in, float *__restrict__ V, int nboards) {
    using G = W4Geom<N>;
    constexpr int C = G::C, LPT = C / 4, TPI = W4_IN_THREADS / LPT, T = G::T, TW = G::TW, COLS = 4 * TW + 2;
    constexpr int ROWF = COLS * C;
    constexpr int NPF = (4 * N * LPT + W4_IN_THREADS - 1) / W4_IN_THREADS;      // prefetch items (f32x4) per thread and step
    extern __shared__ __attribute__((aligned(16))) float win[];                // [6][COLS][C]
    const int tid = threadIdx.x, cq = tid % LPT, tsub = tid / LPT;
    const W4Map<N> map(nboards);
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < 6 * ROWF / 4; k += W4_IN_THREADS) reinterpret_cast<f32x4 *>(win)[k] = zero4;      // halo columns stay zero
    // item k of a load of rows y0 .. y0 + nr - 1: row y0 + k / (N LPT), pixel (k / LPT) % N, channel quad k % LPT
    auto item_src = [&](const float *ib, int y0, int nr, int k, bool &ok) -> const float * {
        const int r = k / (N * LPT), px = (k / LPT) % N, q = k % LPT, y = y0 + r;
        ok = k < nr * N * LPT && y >= 0 && y < N;
        return ib + (size_t)((ok ? y : 0) * N + px) * C + q * 4;
    };
    auto item_dst = [&](int y0, int k) -> float * {
        const int r = k / (N * LPT), px = (k / LPT) % N, q = k % LPT, y = y0 + r;
        return win + ((y + 1) % 6) * ROWF + (px + 1) * C + q * 4;
    };
#pragma unroll 1
    for (int bd = blockIdx.x; bd < nboards; bd += gridDim.x) {
        const float *ib = in + (size_t)bd * (N * N) * C;
        __syncthreads();                                            // the previous board's last reads (and the zero fill)
        // rows -1 .. 4 of the first tile row: -1, 0 and 1 .. 4
#pragma unroll 1
        for (int part = 0; part < 2; part++) {
            const int y0 = part == 0 ? -1 : 1, nr = part == 0 ? 2 : 4;
#pragma unroll
            for (int i = 0; i < NPF; i++) {
                const int k = tid + i * W4_IN_THREADS;
                bool ok;
                const float *sp = item_src(ib, y0, nr, k, ok);
                const f32x4 v = ok ? *reinterpret_cast<const f32x4 *>(sp) : zero4;
                if (k < nr * N * LPT) *reinterpret_cast<f32x4 *>(item_dst(y0, k)) = v;
            }
        }
#pragma unroll 1
        for (int ty = 0; ty < TW; ty++) {
            __syncthreads();                                        // rows 4 ty - 1 .. 4 ty + 4 are in the window
            f32x4 pf[NPF];
            const bool more = ty + 1 < TW;
            if (more) {
#pragma unroll
                for (int i = 0; i < NPF; i++) {
                    bool ok;
                    const float *sp = item_src(ib, 4 * ty + 5, 4, tid + i * W4_IN_THREADS, ok);
                    pf[i] = ok ? *reinterpret_cast<const f32x4 *>(sp) : zero4;
                }
            }
#pragma unroll 1
            for (int tx = tsub; tx < TW; tx += TPI) {
                const long t = map.slot(bd, ty, tx);
                float *vb = V + (size_t)(t / W4_BT) * G::BLOCK_FLOATS + (size_t)(t % W4_BT) * G::TILE_STRIDE;
                // window slot of patch row a (board row 4 ty - 1 + a) and the patch's first column
                const float *wc = win + (4 * tx) * C;
                int slot[6];
#pragma unroll
                for (int a = 0; a < 6; a++) slot[a] = ((4 * ty + a) % 6) * ROWF;
                w4_patch_transform<C>(wc, slot, vb, cq, W4Map<N>::cls(ty, tx));
            }
            if (more) {
                __syncthreads();                                    // rows 4 ty - 1 .. 4 ty + 2 are read no more: their slots take rows 4 ty + 5 .. 4 ty + 8
#pragma unroll
                for (int i = 0; i < NPF; i++) {
                    const int k = tid + i * W4_IN_THREADS;
                    if (k < 4 * N * LPT) *reinterpret_cast<f32x4 *>(item_dst(4 * ty + 5, k)) = pf[i];
                }
            }
        }
    }
}

// c6_wino4_in_kernel with the window rows computed from the board instead of loaded
template <int N>
__global__ __launch_bounds__(W4_IN_THREADS) void c6_wino4_stem_in_kernel(C6StemSrc S, float *__restrict__ V, float *__restrict__ act, int nboards) {
    using G = W4Geom<N>;
    constexpr int C = G::C, LPT = C / 4, TPI = W4_IN_THREADS / LPT, T = G::T, TW = G::TW, COLS = 4 * TW + 2, P = N + 2;
    constexpr int ROWF = COLS * C;
    static_assert(W4_IN_THREADS % LPT == 0, "a thread keeps its channel quad");
    extern __shared__ __attribute__((aligned(16))) float win[];                // [6][COLS][C] + the player-plane sums [9][C] + the board plane [P][P]
    float *s0 = win + 6 * ROWF, *bd = s0 + 9 * C;
    const int tid = threadIdx.x, cq = tid % LPT, tsub = tid / LPT;
    const W4Map<N> map(nboards);
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < 6 * ROWF / 4; k += W4_IN_THREADS) reinterpret_cast<f32x4 *>(win)[k] = zero4;      // halo columns stay zero
    for (int k = tid; k < P * P; k += W4_IN_THREADS) bd[k] = 0.f;
    f32x4 sw[9];
#pragma unroll
    for (int t = 0; t < 9; t++) sw[t] = *reinterpret_cast<const f32x4 *>(S.w + (t * 2 + 1) * C + cq * 4);
    if (tid < LPT) w4_stem_player_sums(S.w, C, tid * 4, s0);
    const f32x4 sb = *reinterpret_cast<const f32x4 *>(S.bias + cq * 4);
    float pv = 0.f;
    float *ab = act;
    // rows y0 .. y0 + nr - 1 of the stem's output into the window (rows outside the board are zeros) and, if asked for, to memory
    auto fill = [&](int y0, int nr) {
        for (int k = tid; k < nr * N * LPT; k += W4_IN_THREADS) {
            const int r = k / (N * LPT), px = (k / LPT) % N, y = y0 + r;
            const bool inb = y >= 0 && y < N;
            const f32x4 v = inb ? w4_stem_pixel<N>(bd, w4_fma(pv, *reinterpret_cast<const f32x4 *>(s0 + w4_border_class(y, px, N) * C + cq * 4), zero4), y, px, sb,
                                                   [&](int t) { return sw[t]; })
                                : zero4;
            *reinterpret_cast<f32x4 *>(win + ((y + 1) % 6) * ROWF + (px + 1) * C + cq * 4) = v;
            if (act && inb) *reinterpret_cast<f32x4 *>(ab + (size_t)(y * N + px) * C + cq * 4) = v;
        }
    };
#pragma unroll 1
    for (int b = blockIdx.x; b < nboards; b += gridDim.x) {
        __syncthreads();                                            // the previous board's last reads (and the zero fills)
        pv = w4_stem_plane<N>(S, b, bd, tid, W4_IN_THREADS);
        ab = act + (size_t)b * (N * N) * C;
        __syncthreads();
        fill(-1, 6);
#pragma unroll 1
        for (int ty = 0; ty < TW; ty++) {
            __syncthreads();                                        // rows 4 ty - 1 .. 4 ty + 4 are in the window
#pragma unroll 1
            for (int tx = tsub; tx < TW; tx += TPI) {
                const long t = map.slot(b, ty, tx);
                float *vb = V + (size_t)(t / W4_BT) * G::BLOCK_FLOATS + (size_t)(t % W4_BT) * G::TILE_STRIDE;
                const float *wc = win + (4 * tx) * C;
                int slot[6];
#pragma unroll
                for (int a = 0; a < 6; a++) slot[a] = ((4 * ty + a) % 6) * ROWF;
                w4_patch_transform<C>(wc, slot, vb, cq, W4Map<N>::cls(ty, tx));
            }
            if (ty + 1 < TW) {
                __syncthreads();                                    // rows 4 ty - 1 .. 4 ty + 2 are read no more
                fill(4 * ty + 5, 4);
            }
        }
    }
}

// the tail of a region's last block (tiles beyond the batch) is never written by the kernels above: zero it once per launch
template <int N>
__global__ void c6_wino4_tail_kernel(float *__restrict__ V, int nboards) {
    using G = W4Geom<N>;
    const W4Map<N> map(nboards);
    const long first_blk[3] = {0, map.blkB, map.blkBP}, count[3] = {(long)nboards * G::NA, (long)nboards * G::NB, (long)nboards * G::NBP};
    for (int r = 0; r < 3; r++) {
        const int used = (int)(count[r] % W4_BT);
        if (used == 0) continue;
        float *blk = V + (size_t)(first_blk[r] + count[r] / W4_BT) * G::BLOCK_FLOATS + (size_t)used * G::TILE_STRIDE;
        const size_t n = (size_t)(W4_BT - used) * G::TILE_STRIDE;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) blk[i] = 0.f;
    }
}
template <int N>
static bool w4_has_tail(int nboards) {
    using G = W4Geom<N>;
    return ((long)nboards * G::NA) % W4_BT || ((long)nboards * G::NB) % W4_BT || ((long)nboards * G::NBP) % W4_BT;
}

// ---- the 36 GEMMs of a block, with the column half of the output transform ("Z rows") ---------------------------------------
// Per block of 64 tiles and position p = (i, j) the GEMM  M[cout][tile] = sum_cin U_p[cout][cin] V_p[cin][tile]  on 32x32x2 MFMAs:
// persistent workgroups of 8 waves; wave w owns couts [32w, 32w + 32) x 64 tiles, so its transformed weights are private and stream
// L2 -> VGPR as one linear 1-KiB-per-group sequence 7 groups ahead (ring of 8; 16 changed nothing); V_p (64 tile rows of 1 KiB) lands in
// LDS by LDS-DMA one position ahead, double-buffered, rows padded to C + 4 floats (conflict-free ds_read_b128); one barrier per position.
//  * Vector-memory operations retire IN ORDER on one counter (loads, LDS-DMAs and stores alike), and the compiler's own s_waitcnt
//    counts do not see inline-asm operations: all of the loop's are issued by hand and every wait names exactly the operations younger
//    than the weight group it needs (w4_wait): the 7 groups behind it, the slab DMAs (group k < 8 issues DMA k of the next position) and
//    the result stores issued since (one per group, groups 8 .. 23 of the positions that store).
//  * The FP32 matrix instructions run on the lanes of the vector ALU (their peak IS the vector FP32 peak; a kernel of plain vector
//    instructions beside this one on another stream makes no progress at all): every vector instruction in the loop is matrix time
//    lost.  The weight stream advances a SCALAR pointer, the tile fragments alternate between two register sets instead of being
//    copied, accumulators start from the instruction's zero operand.  (With ~350 register copies / address adds per position the matrix
//    pipes were busy 0.864 at 2.20 GHz, without them 0.968 at 2.07: the part is at its power limit for this instruction mix.)
//  * Y = A^T M A, and the positions of a patch row i arrive one after the other: Z_i[b] = sum_j M[i][j] A^T[b][j] (b = 0 .. 3) is
//    accumulated in registers -- one row at a time, 4 x 32 registers -- and 24 instead of 36 values per (tile, cout) go to memory: a third
//    less for this kernel to write and for the boundary kernel behind it to read.  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0;
//    0 1 -1 8 -8 1]: position j = 0 accumulates straight into Z_0 and j = 5 straight into Z_3 (the MFMA's own accumulate); j = 1 .. 4
//    accumulate aside and cost 40 packed vector instructions per position on average -- the one exception to the rule above: +1.5 % on
//    this kernel, -19 % on the boundary kernels (same box, in the engine: 248.4 -> 236.8 ms per round).  Stores: Z_0, Z_1 during position
//    j = 5 of their row, Z_2, Z_3 during j = 0 of the next.  Result: Zo[block][tile 64][row i 6][b 4][C].
constexpr int w4_store_group(int k) { return 8 + k; }      // store k (0 .. 15) of a storing position goes out in group 8 + k
constexpr int w4_wait(int i, bool stores, int nd = 8) {      // nd: slab DMAs per wave and position (8 waves: 8; the four-wave jobs of the tiniest batches: 16)
    const int lo = i - W4_PF > 0 ? i - W4_PF : 0, hi = (nd - 1 < i - 1) ? nd - 1 : i - 1;
    const int dmas = hi >= lo ? hi - lo + 1 : 0;
    int st = 0;
    if (stores)
        for (int k = 0; k < 16; k++)
            if (w4_store_group(k) < i && w4_store_group(k) >= i - W4_PF) st++;
    return W4_PF + dmas + st;
}
static_assert(w4_store_group(15) < 32 - W4_PF, "a position's stores are older than every request that is waited for in the next one");
// the LAST position of a split job (SPLIT below, r5): the weight stream ends with the row and no next position is fetched -- younger than group i's
// request are only the groups i + 1 .. 31 and the result stores issued since
constexpr int w4_wait_last(int i) {
    int st = 0;
    for (int k = 0; k < 16; k++)
        if (w4_store_group(k) < i && w4_store_group(k) >= i - W4_PF) st++;
    return (31 - i < W4_PF ? 31 - i : W4_PF) + st;
}

// SPLIT (r5, as conv_wino6.hip's c6_w6_gemm_kernel<., true>): the job is a ROW of a block's positions (blockIdx.y), not the block -- a row's four row
// sums depend on its own six positions only, and a block is a chain of up to 36 positions on one CU whatever the batch.  Same instruction stream per
// position, bit-identical sums; every job has its own prologue, its last position requests nothing beyond the row (LP).  For small batches.
// HW (split form only): half-width jobs of FOUR waves (128 of the 256 output channels, blockIdx.z) for the tiniest batches, as conv_wino6.hip's.
template <int N, bool SPLIT, bool HW = false>
__global__ __launch_bounds__(HW ? 256 : 64 * (W4_C / 32)) void c6_wino4_gemm_kernel(const float *__restrict__ V, const float *__restrict__ U,
                                                                                    float *__restrict__ Zo, int nblocks, int blkB, int blkBP) {
    using G = W4Geom<N>;
    static_assert(!HW || SPLIT, "half-width jobs exist in the split form only");
    constexpr int C = G::C, CP = G::CP, SLAB = G::SLAB, WAVES = HW ? 4 : G::WAVES, NG = G::NG, PF = W4_PF;
    constexpr int ND = W4_BT / WAVES;
    static_assert(PF == 7 && NG == 32 && (ND == 8 || ND == 16), "ring of 8 weight groups; 8 (four waves: 16) DMAs per wave and position");
    constexpr int RM = PF;
    constexpr int ZTILE = 24 * C;                    // floats of a tile in Zo
    constexpr size_t ZBLOCK = (size_t)W4_BT * ZTILE;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int cw = HW ? wave + 4 * (int)blockIdx.z : wave;      // the wave's group of 32 output channels
    const int n4 = cw * 32 + 4 * h;
    int b = blockIdx.x;
    if (b >= nblocks) return;
    const int row0 = SPLIT ? (int)blockIdx.y : 0;

    // blocks [0, blkB) are of class A (6 rows of positions, weight set 0), [blkB, blkBP) of class B, [blkBP, nblocks) of class B' (5 rows,
    // weight sets at 36 C^2 and 66 C^2); a set is packed [wave][position][group][lane][4]
    auto rows_of = [&](int blk) { return blk < blkB ? 6 : 5; };
    auto wbase_of = [&](int blk) {
        return blk < blkB ? U + (size_t)cw * (W4_NP * NG * 256)
                          : U + (size_t)(blk < blkBP ? W4_NP : W4_NP + W4_NP_EDGE) * C * C + (size_t)cw * (W4_NP_EDGE * NG * 256);
    };
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned zoff0 = (unsigned)(((size_t)j * ZTILE + n4) * 4), zoff1 = zoff0 + (unsigned)((size_t)32 * ZTILE * 4);

    auto dma_chunk = [&](const float *src, int buf, int c0) {
        const int c = c0 * WAVES + wave;
        const float *gp = src + (size_t)c * G::TILE_STRIDE;
        const int lds_off = (buf * SLAB + c * CP) * 4;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(gp), "s"(lds_off) : "memory");
    };
    f32x4 ring[PF + 1];
    int ub = b, ungt = rows_of(b) * 6 * NG;      // the weight stream runs ahead of the multiplications: its own block and group counters
    const float *uptr = wbase_of(b);
    int gi = 0;
#define W4_UNEXT()                                                       \
    {                                                                    \
        uptr += 256;                                                     \
        if (!SPLIT) {                                                    \
            gi += 1;                                                     \
            if (gi == ungt) {                                            \
                gi = 0;                                                  \
                ub = ub + (int)gridDim.x < nblocks ? ub + (int)gridDim.x : ub; \
                ungt = rows_of(ub) * 6 * NG;                             \
                uptr = wbase_of(ub);                                     \
            }                                                            \
        }                                                                \
    }
    // the first position's V rows into slab 0, the first seven weight groups into the ring (SPLIT: per job; the empty asm keeps the job's group
    // addresses from being hoisted out of the block loop into scalar registers)
#define W4_PROLOGUE(ROW)                                                                                         \
    {                                                                                                            \
        _Pragma("unroll") for (int c0 = 0; c0 < ND; c0++) dma_chunk(V + (size_t)b * G::BLOCK_FLOATS + (size_t)(6 * (ROW)) * G::POS_STRIDE, 0, c0); \
        uptr = wbase_of(b) + (size_t)(6 * (ROW)) * (NG * 256);                                                   \
        if (SPLIT) asm volatile("" : "+s"(uptr));                                                                \
        _Pragma("unroll") for (int q = 0; q < PF; q++) { W4_ULOAD(ring[q], lane16, uptr); W4_UNEXT() }           \
    }
    if (!SPLIT) W4_PROLOGUE(0)

    // z[b][tile half]; the very first "previous row" is this block's row 0 itself (zeros, overwritten one row later)
    f32x16 z00, z01, z10, z11, z20, z21, z30, z31, m0, m1;
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    z20 = z16; z21 = z16; z30 = z16; z31 = z16;
    const float *zprev = Zo + (size_t)b * ZBLOCK;          // row base of the row whose Z_2, Z_3 are still to be stored
    const float *zcur = zprev;                              // row base of the row being accumulated

    // store k (0 .. 15) of a storing position: two Z sets, each 2 tile halves x 4 quads
#define W4Z_STORE1(SET0, SET1, BIDX, DST, k)                                                                     \
    {                                                                                                            \
        const f32x16 &sv = ((k) & 4) ? SET1 : SET0;                                                              \
        const f32x4 a = {sv[4 * ((k) & 3)], sv[4 * ((k) & 3) + 1], sv[4 * ((k) & 3) + 2], sv[4 * ((k) & 3) + 3]}; \
        /* SPLIT: the second tile half's offset rides in the scalar base (one vector register less) */         \
        const float *d_ = (DST) + (size_t)(BIDX) * C + (SPLIT && ((k) & 4) ? (size_t)32 * ZTILE : (size_t)0);   \
        const unsigned zo_ = SPLIT ? zoff0 : ((k) & 4) ? zoff1 : zoff0;                                          \
        if (((k) & 3) == 0) W4_STORE(zo_, a, d_, 0);                                                             \
        else if (((k) & 3) == 1) W4_STORE(zo_, a, d_, 32);                                                       \
        else if (((k) & 3) == 2) W4_STORE(zo_, a, d_, 64);                                                       \
        else W4_STORE(zo_, a, d_, 96);                                                                           \
    }
#define W4Z_LAST_ROW()                                                                                           \
    {                                                                                                            \
        _Pragma("unroll") for (int k = 0; k < 8; k++) W4Z_STORE1(z20, z21, 2, zprev, k)                           \
        _Pragma("unroll") for (int k = 0; k < 8; k++) W4Z_STORE1(z30, z31, 3, zprev, k)                           \
    }
    // JJ = position's column j: 0 accumulates into z0 from zero, 5 into z3 on top of it, 1 .. 4 into m from zero
#define W4Z_GROUP(q, u, NWAIT, DMA, JJ, BUF, LP)                                                                 \
            {                                                                                                    \
                if (!(LP) || (u) + PF < NG) {      /* (LP: a request nobody waits for lands in registers the compiler has given to something else) */ \
                    W4_ULOAD(ring[((q) + PF) & RM], lane16, uptr);                                               \
                    W4_UNEXT()                                                                                   \
                }                                                                                                \
                if ((u) + 1 < NG) {                                                                              \
                    va[((u) + 1) & 1][0] = *reinterpret_cast<const f32x4 *>(vb + 8 * ((u) + 1));                 \
                    va[((u) + 1) & 1][1] = *reinterpret_cast<const f32x4 *>(vb + 32 * CP + 8 * ((u) + 1));       \
                }                                                                                                \
                W4_UWAIT(ring[(q)], NWAIT);                                                                      \
                const f32x4 w = ring[(q)];                                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                if ((JJ) == 0) {                                                                                 \
                    z00 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va[(u) & 1][0][0], (u) == 0 ? z16 : z00, 0, 0, 0); \
                    z01 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va[(u) & 1][1][0], (u) == 0 ? z16 : z01, 0, 0, 0); \
                } else if ((JJ) == 5) {                                                                          \
                    z30 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va[(u) & 1][0][0], z30, 0, 0, 0);           \
                    z31 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va[(u) & 1][1][0], z31, 0, 0, 0);           \
                } else {                                                                                         \
                    m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va[(u) & 1][0][0], (u) == 0 ? z16 : m0, 0, 0, 0); \
                    m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va[(u) & 1][1][0], (u) == 0 ? z16 : m1, 0, 0, 0); \
                }                                                                                                \
                if (!(LP) && (DMA) >= 0 && (DMA) < ND) dma_chunk(nsrc, (BUF) ^ 1, (DMA));                        \
                if ((JJ) == 5 && (u) >= 8 && (u) < 24) {      /* Z_0, Z_1 of this row */                          \
                    if ((u) < 16) W4Z_STORE1(z00, z01, 0, zcur, (u) - 8)                                         \
                    else W4Z_STORE1(z10, z11, 1, zcur, (u) - 16)                                                 \
                }                                                                                                \
                if (!SPLIT && (JJ) == 0 && (u) >= 8 && (u) < 24) {      /* Z_2, Z_3 of the previous row (SPLIT: one row, no previous one) */ \
                    if ((u) < 16) W4Z_STORE1(z20, z21, 2, zprev, (u) - 8)                                        \
                    else W4Z_STORE1(z30, z31, 3, zprev, (u) - 16)                                                \
                }                                                                                                \
                _Pragma("unroll") for (int r = 1; r < 4; r++) {                                                  \
                    if ((JJ) == 0) {                                                                             \
                        z00 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va[(u) & 1][0][r], z00, 0, 0, 0);       \
                        z01 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va[(u) & 1][1][r], z01, 0, 0, 0);       \
                    } else if ((JJ) == 5) {                                                                      \
                        z30 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va[(u) & 1][0][r], z30, 0, 0, 0);       \
                        z31 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va[(u) & 1][1][r], z31, 0, 0, 0);       \
                    } else {                                                                                     \
                        m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va[(u) & 1][0][r], m0, 0, 0, 0);         \
                        m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va[(u) & 1][1][r], m1, 0, 0, 0);         \
                    }                                                                                            \
                }                                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }
#define WGZ_(i, JJ, BUF) W4Z_GROUP((i) & RM, (i), (SPLIT && (JJ) == 5) ? w4_wait_last((i)) : w4_wait((i), (!SPLIT && (JJ) == 0) || (JJ) == 5, ND), (i), JJ, BUF, SPLIT && (JJ) == 5)
    // one position (row i_, column JJ): slab in LDS buffer JJ & 1
#define W4Z_POSITION(i_, JJ)                                                                                     \
        {                                                                                                        \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF) : "memory");                                            \
            __builtin_amdgcn_s_barrier();                                                                        \
            const float *nsrc;                                                                                   \
            {                                                                                                    \
                const int pp = 6 * (i_) + (JJ);                                                                  \
                if (SPLIT) nsrc = V + (size_t)b * G::BLOCK_FLOATS + (size_t)(pp + 1) * G::POS_STRIDE;      /* (not used by the job's last position) */ \
                else {                                                                                           \
                    int nb = pp < 6 * R - 1 ? b : b + (int)gridDim.x;                                            \
                    nb = nb < nblocks ? nb : b;                                                                  \
                    nsrc = V + (size_t)nb * G::BLOCK_FLOATS + (size_t)(pp < 6 * R - 1 ? pp + 1 : 0) * G::POS_STRIDE; \
                }                                                                                                \
            }                                                                                                    \
            const float *vb = wsm + ((JJ) & 1) * SLAB + j * CP + 4 * h;                                          \
            f32x4 va[2][2];                                                                                      \
            va[0][0] = *reinterpret_cast<const f32x4 *>(vb);                                                     \
            va[0][1] = *reinterpret_cast<const f32x4 *>(vb + 32 * CP);                                           \
            WGZ_(0, JJ, (JJ) & 1) WGZ_(1, JJ, (JJ) & 1) WGZ_(2, JJ, (JJ) & 1) WGZ_(3, JJ, (JJ) & 1)                     \
            WGZ_(4, JJ, (JJ) & 1) WGZ_(5, JJ, (JJ) & 1) WGZ_(6, JJ, (JJ) & 1) WGZ_(7, JJ, (JJ) & 1)                     \
            WGZ_(8, JJ, (JJ) & 1) WGZ_(9, JJ, (JJ) & 1) WGZ_(10, JJ, (JJ) & 1) WGZ_(11, JJ, (JJ) & 1)                   \
            WGZ_(12, JJ, (JJ) & 1) WGZ_(13, JJ, (JJ) & 1) WGZ_(14, JJ, (JJ) & 1) WGZ_(15, JJ, (JJ) & 1)                 \
            WGZ_(16, JJ, (JJ) & 1) WGZ_(17, JJ, (JJ) & 1) WGZ_(18, JJ, (JJ) & 1) WGZ_(19, JJ, (JJ) & 1)                 \
            WGZ_(20, JJ, (JJ) & 1) WGZ_(21, JJ, (JJ) & 1) WGZ_(22, JJ, (JJ) & 1) WGZ_(23, JJ, (JJ) & 1)                 \
            WGZ_(24, JJ, (JJ) & 1) WGZ_(25, JJ, (JJ) & 1) WGZ_(26, JJ, (JJ) & 1) WGZ_(27, JJ, (JJ) & 1)                 \
            WGZ_(28, JJ, (JJ) & 1) WGZ_(29, JJ, (JJ) & 1) WGZ_(30, JJ, (JJ) & 1) WGZ_(31, JJ, (JJ) & 1)                 \
        }
    // the vector part: Z_b += A^T[b][JJ] * M for the columns that accumulate aside (A^T[.][1..4] = (1,1,1,1), (1,-1,1,-1), (1,2,4,8), (1,-2,4,-8))
#define W4Z_UPDATE(JJ)                                                                                           \
        _Pragma("unroll") for (int v = 0; v < 16; v++) {                                                         \
            z00[v] += m0[v]; z01[v] += m1[v];                                                                    \
            if ((JJ) == 1) { z10[v] = m0[v]; z11[v] = m1[v]; z20[v] = m0[v]; z21[v] = m1[v]; z30[v] = m0[v]; z31[v] = m1[v]; } \
            if ((JJ) == 2) { z10[v] -= m0[v]; z11[v] -= m1[v]; z20[v] += m0[v]; z21[v] += m1[v]; z30[v] -= m0[v]; z31[v] -= m1[v]; } \
            if ((JJ) == 3) {                                                                                     \
                z10[v] = __builtin_fmaf(2.f, m0[v], z10[v]); z11[v] = __builtin_fmaf(2.f, m1[v], z11[v]);        \
                z20[v] = __builtin_fmaf(4.f, m0[v], z20[v]); z21[v] = __builtin_fmaf(4.f, m1[v], z21[v]);        \
                z30[v] = __builtin_fmaf(8.f, m0[v], z30[v]); z31[v] = __builtin_fmaf(8.f, m1[v], z31[v]);        \
            }                                                                                                    \
            if ((JJ) == 4) {                                                                                     \
                z10[v] = __builtin_fmaf(-2.f, m0[v], z10[v]); z11[v] = __builtin_fmaf(-2.f, m1[v], z11[v]);      \
                z20[v] = __builtin_fmaf(4.f, m0[v], z20[v]); z21[v] = __builtin_fmaf(4.f, m1[v], z21[v]);        \
                z30[v] = __builtin_fmaf(-8.f, m0[v], z30[v]); z31[v] = __builtin_fmaf(-8.f, m1[v], z31[v]);      \
            }                                                                                                    \
        }

#pragma unroll 1
    for (; b < nblocks; b += gridDim.x) {
        const int R = rows_of(b);
        if (SPLIT) {
            if (row0 >= R) continue;      // (an edge class has five rows: uniform over the workgroup)
            W4_PROLOGUE(row0)             // (what the previous job left in flight -- its last Z stores -- is older than these requests and retires before them)
        }
#pragma unroll 1
        for (int i = row0; i < (SPLIT ? row0 + 1 : R); i++) {
            zcur = Zo + (size_t)b * ZBLOCK + (size_t)i * (4 * C);
            W4Z_POSITION(i, 0)            // (stores Z_2, Z_3 of the previous row to zprev)
            W4Z_POSITION(i, 1) W4Z_UPDATE(1)
            W4Z_POSITION(i, 2) W4Z_UPDATE(2)
            W4Z_POSITION(i, 3) W4Z_UPDATE(3)
            W4Z_POSITION(i, 4) W4Z_UPDATE(4)
            W4Z_POSITION(i, 5)            // (stores Z_0, Z_1 of this row to zcur)
            zprev = zcur;
        }
        if (SPLIT) W4Z_LAST_ROW()
    }
    // the last row's Z_2, Z_3 (first the weight stream's last seven requests land: nobody waits for them -- conv_wino6.hip, r5)
    if (!SPLIT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W4Z_LAST_ROW()
    }
#undef W4Z_LAST_ROW
#undef W4_PROLOGUE
#undef W4Z_UPDATE
#undef W4Z_POSITION
#undef WGZ_
#undef W4Z_GROUP
#undef W4Z_STORE1
#undef W4_UNEXT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- output transform, the half the GEMM kernel left: A^T Z, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1], + bias (+ residual), ReLU ----
// One wave per tile at a time, lane = 4 channels: the tile's 24 (edge tiles: 20) x 1 KiB of Z are one contiguous run; the 4x4 pixels leave as
// 1-KiB NHWC runs (pixels beyond the board edge are dropped).  HEADS (the tower's last layer): the activations are consumed here --
// the heads' three 1x1 filters (policy conv C -> 2 + folded BN + ReLU, value conv C -> 1 + BatchNorm1d over cells + ReLU;
// model.hpp:80-92, c6_head_conv_kernel) are applied to every pixel (a wave sum per filter) and nothing else is written: 12 GB of
// activations neither stored nor read back.
// the 4x4 outputs of a tile (before bias) for this lane's 4 channels from the GEMM kernel's row sums Z_i[b]: the other half of the
// output transform; zb = the lane's corner of the tile in Z.  Class A: A^T Z with F(4)'s A^T over 6 rows.  Class B': F(3)'s
// A^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1] over 5 rows -> output rows 0 .. 2.  Class B (stored transposed: Z's row index is the patch
// COLUMN j, b the output ROW): the same over j -> output columns 0 .. 2.  The fourth row / column of an edge tile lies beyond the board.
#define W4_ZTILE(C) (24 * (C))
template <int C>
__device__ __forceinline__ void w4_tile_outputs(const float *zb, f32x4 (&y)[4][4], int cls) {
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cls == W4_CLS_A) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            f32x4 z[6];
#pragma unroll
            for (int i = 0; i < 6; i++) z[i] = *reinterpret_cast<const f32x4 *>(zb + (size_t)(i * 4 + b) * C);
            const f32x4 s12 = z[1] + z[2], d12 = z[1] - z[2], s34 = z[3] + z[4], d34 = z[3] - z[4];
            y[0][b] = z[0] + s12 + s34;
            y[1][b] = w4_fma(2.f, d34, d12);
            y[2][b] = w4_fma(4.f, s34, s12);
            y[3][b] = w4_fma(8.f, d34, d12) + z[5];
        }
    } else {
        f32x4 e[3][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const f32x4 z0 = *reinterpret_cast<const f32x4 *>(zb + (size_t)(0 * 4 + b) * C), z1 = *reinterpret_cast<const f32x4 *>(zb + (size_t)(1 * 4 + b) * C),
                        z2 = *reinterpret_cast<const f32x4 *>(zb + (size_t)(2 * 4 + b) * C), z3 = *reinterpret_cast<const f32x4 *>(zb + (size_t)(3 * 4 + b) * C),
                        z4 = *reinterpret_cast<const f32x4 *>(zb + (size_t)(4 * 4 + b) * C);
            const f32x4 s12 = z1 + z2, d12 = z1 - z2;
            e[0][b] = z0 + s12 + z3;
            e[1][b] = w4_fma(2.f, z3, d12);
            e[2][b] = w4_fma(4.f, z3, s12) + z4;
        }
        const bool tr = cls == W4_CLS_B;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const f32x4 rowwise = a < 3 ? e[a < 3 ? a : 0][b] : zero4, colwise = b < 3 ? e[b < 3 ? b : 0][a] : zero4;
                y[a][b] = tr ? colwise : rowwise;
            }
    }
}
// STATS (r6, the trainer's forward pass): the per-channel sums of the outputs and of their squares -- the batch statistics of the BatchNorm behind the
// convolution (model.hpp:61-78 in train mode) -- accumulated on the way out: partial[blockIdx.x][2][C], summed in block order by c6t_colstats_finalize_kernel.
// A wave adds its tiles' pixels in the order it visits them, the four waves of a workgroup are added in wave order: a fixed order for a fixed grid, so a
// step stays bit-reproducible.  (Before: a pass of its own over the stored outputs, 0.11 ms per layer at the reference trainer's shape.)
// STATS = 2 (the trainer's input-gradient pass): the outputs are the gradient `dout` that reaches a BatchNorm + ReLU unit; the two sums its backward pass
// needs -- s0 = sum dz, s1 = sum dz * xhat with dz = dout * (aout > 0), xhat = (y - mean) * invstd (c6t_colstats mode 1) -- are taken here from the unit's
// stored output `aout` and pre-norm input `y`: that pass over three tensors (0.25 ms per layer) is gone.
struct C6BnBwdSrc { const float *aout, *y, *mean, *var; };
template <int N, bool HEADS, int STATS = 0>
__global__ __launch_bounds__(256) void c6_wino4_out_kernel(const float *__restrict__ M, const float *__restrict__ bias, const float *__restrict__ res,
                                                           float *__restrict__ out, long total_tiles, int relu, C6HeadFuse hd, float *__restrict__ stat_partial = nullptr,
                                                           C6BnBwdSrc bn = C6BnBwdSrc{nullptr, nullptr, nullptr, nullptr}) {
    using G = W4Geom<N>;
    f32x4 st0 = f32x4{0.f, 0.f, 0.f, 0.f}, st1 = st0, bm = st0, bis = st0;
    if (STATS == 2) {
        bm = *reinterpret_cast<const f32x4 *>(bn.mean + (threadIdx.x & 63) * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bn.var + (threadIdx.x & 63) * 4);
#pragma unroll
        for (int r = 0; r < 4; r++) bis[r] = 1.0f / sqrtf(bv[r] + 1e-5f);      // (BN_EPS of train_kernels.hip: torch's BatchNorm default)
    }
    constexpr int C = G::C, T = G::T, TW = G::TW;
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const f32x4 bi = bias ? *reinterpret_cast<const f32x4 *>(bias + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float floor_ = relu ? 0.f : -INFINITY;
    f32x4 hw0 = bi, hw1 = bi, hw2 = bi;
    if (HEADS) {
        hw0 = *reinterpret_cast<const f32x4 *>(hd.w + lane * 4);
        hw1 = *reinterpret_cast<const f32x4 *>(hd.w + C + lane * 4);
        hw2 = *reinterpret_cast<const f32x4 *>(hd.w + 2 * C + lane * 4);
    }
    const W4Map<N> map(total_tiles / T);
#pragma unroll 1
    for (long t = wave0; t < total_tiles; t += nwaves) {
        const long bd = t / T;
        const int tt = (int)(t - bd * T), ty = tt / TW, tx = tt - ty * TW;
        const float *mb = M + (size_t)map.slot(bd, ty, tx) * W4_ZTILE(C) + lane * 4;
        f32x4 yv[4][4];
        w4_tile_outputs<C>(mb, yv, W4Map<N>::cls(ty, tx));
        float hs[4][4][3];      // HEADS: this lane's share of the three filters' sums at the tile's 16 pixels
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int bb = 0; bb < 4; bb++) { hs[a][bb][0] = 0.f; hs[a][bb][1] = 0.f; hs[a][bb][2] = 0.f; }
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int py = 4 * ty + a;
            if (py >= N) continue;
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
                const int px = 4 * tx + bb;
                if (px >= N) continue;
                const int p = py * N + px;
                const size_t o = ((size_t)bd * (N * N) + (size_t)p) * C + lane * 4;
                f32x4 v = yv[a][bb] + bi;
                if (res) v = v + *reinterpret_cast<const f32x4 *>(res + o);
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = v[r] > floor_ ? v[r] : floor_;
                if (STATS == 1) {
                    st0 = st0 + v;
#pragma unroll
                    for (int r = 0; r < 4; r++) st1[r] = __builtin_fmaf(v[r], v[r], st1[r]);
                }
                if (STATS == 2) {
                    const f32x4 av = *reinterpret_cast<const f32x4 *>(bn.aout + o), yy = *reinterpret_cast<const f32x4 *>(bn.y + o);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float dz = av[r] > 0.f ? v[r] : 0.f;
                        st0[r] += dz;
                        st1[r] = __builtin_fmaf(dz, (yy[r] - bm[r]) * bis[r], st1[r]);
                    }
                }
                if (!HEADS) {
                    *reinterpret_cast<f32x4 *>(out + o) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        hs[a][bb][0] = __builtin_fmaf(v[r], hw0[r], hs[a][bb][0]);
                        hs[a][bb][1] = __builtin_fmaf(v[r], hw1[r], hs[a][bb][1]);
                        hs[a][bb][2] = __builtin_fmaf(v[r], hw2[r], hs[a][bb][2]);
                    }
                }
            }
        }
        if (HEADS) {
            // 48 independent wave sums (16 pixels x 3 filters), issued together so that their dependent steps interleave
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int bb = 0; bb < 4; bb++)
#pragma unroll
                    for (int f = 0; f < 3; f++) hs[a][bb][f] = w4_wave_sum_row3(hs[a][bb][f]);
            // lanes 48 .. 63 have all 48 sums: lane 48 + l writes pixel l of the tile (three 16-lane stores instead of 48 one-lane ones)
            float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int bb = 0; bb < 4; bb++)
                    if ((lane & 15) == a * 4 + bb) { m0 = hs[a][bb][0]; m1 = hs[a][bb][1]; m2 = hs[a][bb][2]; }
            const int py = 4 * ty + ((lane & 15) >> 2), px = 4 * tx + (lane & 3);
            if (lane >= 48 && py < N && px < N) {
                const int p = py * N + px;
                const float y0 = m0 + hd.pc_b[0], y1 = m1 + hd.pc_b[1];
                hd.ph[(size_t)bd * (2 * N * N) + p] = y0 > 0.f ? y0 : 0.f;             // flatten is channel-major (NCHW view)
                hd.ph[(size_t)bd * (2 * N * N) + N * N + p] = y1 > 0.f ? y1 : 0.f;
                const float vv = (m2 + hd.vc_b) * hd.vn_scale[p] + hd.vn_shift[p];
                hd.vh[(size_t)bd * (N * N) + p] = vv > 0.f ? vv : 0.f;
            }
        }
    }
    if (STATS != 0) {
        __shared__ f32x4 sred[2][4][64];
        const int wv = threadIdx.x >> 6;
        sred[0][wv][lane] = st0;
        sred[1][wv][lane] = st1;
        __syncthreads();
        if (wv == 0) {
            for (int w2 = 1; w2 < 4; w2++) { st0 = st0 + sred[0][w2][lane]; st1 = st1 + sred[1][w2][lane]; }
            *reinterpret_cast<f32x4 *>(stat_partial + ((size_t)blockIdx.x * 2 + 0) * C + lane * 4) = st0;
            *reinterpret_cast<f32x4 *>(stat_partial + ((size_t)blockIdx.x * 2 + 1) * C + lane * 4) = st1;
        }
    }
}

// ---- output transform of one layer + input transform of the next, in one pass ------------------------------------------
// Between the two convolutions of a residual block the activations have one reader: written by the first layer's output transform
// (12 GB per 32 768 boards) only to be read back by the second layer's input transform.  This kernel goes from Z of layer l (the GEMM
// kernel's row sums) to V of layer l + 1 with the activations passing through LDS: one workgroup per board at a time, wave = tile column,
// lane = 4 channels.  Step t: (a) the wave turns its tile (t, tx) of Z into 4x4 pixels (A^T Z + bias, ReLU) and puts the FIRST of their rows, board
// row 4t, into the rolling 6-row window of c6_wino4_in_kernel; (b) the window now holds rows 4t - 5 .. 4t: the patch rows of tile
// row t - 1, whose B^T d B goes out as the next layer's V; (c) pixel rows 4t + 1 .. 4t + 3, kept in registers meanwhile, take the
// slots of the rows (b) read last.  HBM: Z in, V out -- 47 GB instead of 72 for the pair of kernels it replaces.
// RS: the residual is the STEM's output, computed here from the board (w4_stem_pixel; weights and board plane in LDS behind the window)
template <int N, bool RS>
__global__ __launch_bounds__(64 * W4Geom<N>::TW) void c6_wino4_mid_kernel(const float *__restrict__ M, const float *__restrict__ bias,
                                                                          const float *__restrict__ res, float *__restrict__ act,
                                                                          float *__restrict__ V, int nboards, C6StemSrc S) {
    using G = W4Geom<N>;
    constexpr int C = G::C, T = G::T, TW = G::TW, COLS = 4 * TW + 2, ROWF = COLS * C, NT = 64 * TW;
    extern __shared__ __attribute__((aligned(16))) float win[];                // [6][COLS][C]
    const int tid = threadIdx.x, lane = tid & 63, tx = tid >> 6;
    const W4Map<N> map(nboards);
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 bi = bias ? *reinterpret_cast<const f32x4 *>(bias + lane * 4) : zero4;
    for (int k = tid; k < 6 * ROWF / 4; k += NT) reinterpret_cast<f32x4 *>(win)[k] = zero4;      // halo columns stay zero
    float *wpix = win + (4 * tx + 1) * C + lane * 4;       // window column of this wave's first pixel column (board column 4 tx)
    float *swl = win + 6 * ROWF, *s0 = swl + 9 * C, *bdp = s0 + 9 * C;     // RS: the stem's board-plane weights [9][C], its player-plane sums [9][C], the board plane [(N+2)^2]
    f32x4 sb = zero4;
    float pv = 0.f;
    if (RS) {
        for (int k = tid; k < 9 * C / 4; k += NT) reinterpret_cast<f32x4 *>(swl)[k] = *reinterpret_cast<const f32x4 *>(S.w + ((k / (C / 4)) * 2 + 1) * C + (k % (C / 4)) * 4);
        if (tid < C / 4) w4_stem_player_sums(S.w, C, tid * 4, s0);
        for (int k = tid; k < (N + 2) * (N + 2); k += NT) bdp[k] = 0.f;
        sb = *reinterpret_cast<const f32x4 *>(S.bias + lane * 4);
    }
#pragma unroll 1
    for (int bd = blockIdx.x; bd < nboards; bd += gridDim.x) {
        __syncthreads();                                    // the previous board's last window reads (and the zero fill)
        for (int k = tid; k < ROWF / 4; k += NT) reinterpret_cast<f32x4 *>(win)[k] = zero4;      // slot 0 = board row -1
        if (RS) {
            pv = w4_stem_plane<N>(S, bd, bdp, tid, NT);
            __syncthreads();
        }
#pragma unroll 1
        for (int t = 0; t <= TW; t++) {
            f32x4 keep[3][4];
            if (t < TW) {
                const float *mb = M + (size_t)map.slot(bd, t, tx) * W4_ZTILE(C) + lane * 4;
                f32x4 yv[4][4];
                w4_tile_outputs<C>(mb, yv, W4Map<N>::cls(t, tx));
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const int py = 4 * t + a;
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) {
                        f32x4 v = yv[a][bb] + bi;
                        const bool inb = py < N && 4 * tx + bb < N;
                        const size_t o = ((size_t)bd * (N * N) + (size_t)((inb ? py : 0) * N + (inb ? 4 * tx + bb : 0))) * C + lane * 4;
                        if (RS) {
                            if (inb) v = v + w4_stem_pixel<N>(bdp, w4_fma(pv, *reinterpret_cast<const f32x4 *>(s0 + w4_border_class(py, 4 * tx + bb, N) * C + lane * 4), zero4), py, 4 * tx + bb, sb,
                                                              [&](int t) { return *reinterpret_cast<const f32x4 *>(swl + t * C + lane * 4); });
                        } else if (res && inb) v = v + *reinterpret_cast<const f32x4 *>(res + o);
#pragma unroll
                        for (int r = 0; r < 4; r++) v[r] = v[r] > 0.f ? v[r] : 0.f;
                        if (!inb) v = zero4;                             // beyond the board: the next layer's zero padding
                        else if (act) *reinterpret_cast<f32x4 *>(act + o) = v;      // a block's output is also the next block's residual
                        if (a == 0) *reinterpret_cast<f32x4 *>(wpix + ((py + 1) % 6) * ROWF + bb * C) = v;
                        else keep[a - 1][bb] = v;
                    }
                }
            } else {
#pragma unroll
                for (int bb = 0; bb < 4; bb++) *reinterpret_cast<f32x4 *>(wpix + ((4 * t + 1) % 6) * ROWF + bb * C) = zero4;      // row 4 TW: below the board
            }
            __syncthreads();                                // board rows 4t - 5 .. 4t are in the window
            if (t >= 1) {
                const int ty = t - 1;
                const long tt = map.slot(bd, ty, tx);
                float *vb = V + (size_t)(tt / W4_BT) * G::BLOCK_FLOATS + (size_t)(tt % W4_BT) * G::TILE_STRIDE;
                const float *wc = win + (4 * tx) * C;
                int slot[6];
#pragma unroll
                for (int a = 0; a < 6; a++) slot[a] = ((4 * ty + a) % 6) * ROWF;
                w4_patch_transform<C>(wc, slot, vb, lane, W4Map<N>::cls(ty, tx));
            }
            __syncthreads();                                // rows 4t - 5 .. 4t - 3 are read no more: their slots take rows 4t + 1 .. 4t + 3
            if (t < TW) {
#pragma unroll
                for (int a = 1; a < 4; a++)
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) *reinterpret_cast<f32x4 *>(wpix + ((4 * t + a + 1) % 6) * ROWF + bb * C) = keep[a - 1][bb];
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
size_t c6_conv_packed_wino4_floats(int C) { return (size_t)(W4_NP + 2 * W4_NP_EDGE) * C * C; }

// Transformed weights U = G_v g G_h^T (double, rounded once), three sets one behind the other (W4Geom): class A (36 positions p = 6 i + j:
// F(4)'s G both ways), class B (30 positions p = 6 j + i: G4 vertically, F(3)'s G horizontally), class B' (30 positions p = 6 i + j: G3
// vertically, G4 horizontally).  A set is packed [wave][position][group u][lane][r]: U_p[cout = 32 wave + (lane & 31)][cin = 8u + 4(lane >> 5) + r].
// g: the 9 taps of (cout, cin); flip: the input-gradient layer's taps w'[2 - a][2 - b]
__host__ __device__ inline double w4_u_element(int set, int pos, const float *g, bool flip) {
    const double G4[6][3] = {{1.0 / 4, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const double G3[5][3] = {{1.0 / 2, 0.0, 0.0}, {-1.0 / 2, -1.0 / 2, -1.0 / 2}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0.0, 0.0, 1.0}};
    const int p = pos / 6, q = pos % 6;
    const double *gv = set == 0 ? G4[p] : set == 1 ? G4[q] : G3[p];      // the row of G on the vertical side
    const double *gh = set == 0 ? G4[q] : set == 1 ? G3[p] : G4[q];      // ... and on the horizontal side
    double sum = 0.0;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) sum += gv[a] * (double)g[flip ? 8 - (a * 3 + b) : a * 3 + b] * gh[b];
    return sum;
}
// element e of the packed array -> (set, position, cout, cin)
__host__ __device__ inline void w4_u_index(int C, size_t e, int &set, int &pos, int &co, int &ci) {
    const size_t cc = (size_t)C * C;
    set = e < W4_NP * cc ? 0 : e < (W4_NP + W4_NP_EDGE) * cc ? 1 : 2;
    const int npos = set == 0 ? W4_NP : W4_NP_EDGE, NG = C / 8;
    size_t q = e - (set == 0 ? 0 : set == 1 ? W4_NP * cc : (W4_NP + W4_NP_EDGE) * cc);
    const int r = (int)(q % 4); q /= 4;
    const int lane = (int)(q % 64); q /= 64;
    const int u = (int)(q % NG); q /= NG;
    pos = (int)(q % npos); q /= npos;
    co = (int)q * 32 + (lane & 31);
    ci = 8 * u + 4 * (lane >> 5) + r;
}
// w: folded OIHW [C][C][3][3]
void c6_pack_conv_weights_wino4(int C, const float *w, float *packed) {
    const size_t total = c6_conv_packed_wino4_floats(C);
    for (size_t e = 0; e < total; e++) {
        int set, pos, co, ci;
        w4_u_index(C, e, set, pos, co, ci);
        packed[e] = (float)w4_u_element(set, pos, w + ((size_t)co * C + ci) * 9, false);
    }
}

// the same packing on the device, from weights that change every training step (trainer.cpp); transpose_flip: the layer of the
// input gradient, w'[ci][co][2 - a][2 - b]
__global__ void c6_pack_wino4_kernel(int C, const float *__restrict__ w, float *__restrict__ packed, int transpose_flip) {
    const size_t total = (size_t)(W4_NP + 2 * W4_NP_EDGE) * C * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int set, pos, co, ci;
        w4_u_index(C, e, set, pos, co, ci);
        const float *g = transpose_flip ? w + ((size_t)ci * C + co) * 9 : w + ((size_t)co * C + ci) * 9;
        packed[e] = (float)w4_u_element(set, pos, g, transpose_flip != 0);
    }
}
void c6_pack_conv_weights_wino4_dev(int C, const float *w_oihw_dev, float *packed_dev, int transpose_flip, hipStream_t s) {
    hipLaunchKernelGGL(c6_pack_wino4_kernel, dim3(1024), dim3(256), 0, s, C, w_oihw_dev, packed_dev, transpose_flip);
}

// floats of the scratch for `nboards` boards: V and Z, one behind the other (Z uses 24 / 36 of its half)
static long w4_nblocks(int N, int nboards) { return N == 19 ? W4Map<19>(nboards).nblocks : N == 15 ? W4Map<15>(nboards).nblocks : W4Map<9>(nboards).nblocks; }
static size_t w4_half_floats(int N, int nboards) { return (size_t)w4_nblocks(N, nboards) * W4_BT * W4_NP * W4_C; }
size_t c6_conv_wino4_vbuf_floats(int C, int N, int nboards) { return C == W4_C ? 2 * w4_half_floats(N, nboards) : 0; }

template <int N>
static void launch_w4(int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out, float *vbuf, int num_cus,
                      int relu, int steps, hipStream_t s, const C6HeadFuse *heads, const C6StemSrc *stem, float *zsep = nullptr, float *stat_partial = nullptr,
                      const float *const *bn_bwd = nullptr) {
    using G = W4Geom<N>;
    const long tiles = (long)nboards * G::T;
    const W4Map<N> map(nboards);
    const int nblocks = (int)map.nblocks;
    const bool tail = w4_has_tail<N>(nboards);
    float *Vb = vbuf, *Mb = zsep ? zsep : vbuf + w4_half_floats(N, nboards);      // zsep: V alone in `vbuf` (a caller that keeps V), Z elsewhere
    if ((steps & 1) && stem) {      // the first layer: window rows computed from the boards
        constexpr size_t in_smem = (size_t)6 * (4 * G::TW + 2) * W4_C * 4 + (size_t)9 * W4_C * 4 + (size_t)(N + 2) * (N + 2) * 4;
        const int grid_in = nboards < num_cus ? nboards : num_cus;
        w4_smem_once(reinterpret_cast<const void *>(&c6_wino4_stem_in_kernel<N>), (int)in_smem);
        hipLaunchKernelGGL((c6_wino4_stem_in_kernel<N>), dim3(grid_in), dim3(W4_IN_THREADS), in_smem, s, *stem, Vb, out, nboards);
        if (tail) hipLaunchKernelGGL((c6_wino4_tail_kernel<N>), dim3(256), dim3(256), 0, s, Vb, nboards);
    } else if (steps & 1) {
        constexpr size_t in_smem = (size_t)6 * (4 * G::TW + 2) * W4_C * 4;
        const int grid_in = nboards < num_cus ? nboards : num_cus;
        w4_smem_once(reinterpret_cast<const void *>(&c6_wino4_in_kernel<N>), (int)in_smem);
        hipLaunchKernelGGL((c6_wino4_in_kernel<N>), dim3(grid_in), dim3(W4_IN_THREADS), in_smem, s, in, Vb, nboards);
        if (tail) hipLaunchKernelGGL((c6_wino4_tail_kernel<N>), dim3(256), dim3(256), 0, s, Vb, nboards);
    }
    if (steps & 2) {
        const size_t smem = G::LDS_BYTES;
        const int grid = nblocks < num_cus ? nblocks : num_cus;
        // r5 -- rows instead of blocks as the unit of work while a CU gets fewer than eight blocks (conv_wino6.hip's c6_w6_dispatch);
        // C6_W4_SPLIT = 0 / 1: a test hook that forces one form (tests/test_gpu_variants.py compares the two bit for bit)
        static const int force = [] { const char *e = getenv("C6_W4_SPLIT"); return e ? atoi(e) : -1; }();
        static const int force_hw = [] { const char *e = getenv("C6_W4_HALF"); return e ? atoi(e) : -1; }();
        const bool split = force >= 0 ? force != 0 : nblocks < 8 * num_cus;
        if (split && (force_hw >= 0 ? force_hw != 0 : 2 * 6 * nblocks <= num_cus)) {      // the tiniest batches: half-width jobs of four waves
            w4_smem_once(reinterpret_cast<const void *>(&c6_wino4_gemm_kernel<N, true, true>), (int)smem);
            hipLaunchKernelGGL((c6_wino4_gemm_kernel<N, true, true>), dim3(grid, 6, 2), dim3(256), smem, s, Vb, wp, Mb, nblocks, (int)map.blkB, (int)map.blkBP);
        } else if (split) {
            w4_smem_once(reinterpret_cast<const void *>(&c6_wino4_gemm_kernel<N, true>), (int)smem);
            hipLaunchKernelGGL((c6_wino4_gemm_kernel<N, true>), dim3(grid, 6), dim3(G::THREADS), smem, s, Vb, wp, Mb, nblocks, (int)map.blkB, (int)map.blkBP);
        } else {
            w4_smem_once(reinterpret_cast<const void *>(&c6_wino4_gemm_kernel<N, false>), (int)smem);
            hipLaunchKernelGGL((c6_wino4_gemm_kernel<N, false>), dim3(grid), dim3(G::THREADS), smem, s, Vb, wp, Mb, nblocks, (int)map.blkB, (int)map.blkBP);
        }
    }
    if (steps & 16) {      // output transform of this layer (bias, residual if any, ReLU; activations stored if `out`) + input transform of the next: M -> V
        constexpr size_t mid_smem = (size_t)6 * (4 * G::TW + 2) * W4_C * 4;
        constexpr size_t rs_smem = mid_smem + (size_t)18 * W4_C * 4 + (size_t)(N + 2) * (N + 2) * 4;
        const int grid_mid = nboards < num_cus ? nboards : num_cus;
        if (stem) {      // the residual is the stem's output, recomputed from the boards
            w4_smem_once(reinterpret_cast<const void *>(&c6_wino4_mid_kernel<N, true>), (int)rs_smem);
            hipLaunchKernelGGL((c6_wino4_mid_kernel<N, true>), dim3(grid_mid), dim3(64 * G::TW), rs_smem, s, Mb, bias, res, out, Vb, nboards, *stem);
        } else {
            w4_smem_once(reinterpret_cast<const void *>(&c6_wino4_mid_kernel<N, false>), (int)mid_smem);
            hipLaunchKernelGGL((c6_wino4_mid_kernel<N, false>), dim3(grid_mid), dim3(64 * G::TW), mid_smem, s, Mb, bias, res, out, Vb, nboards, C6StemSrc{});
        }
        // (no tail kernel here, as in conv_wino6.hip: this kernel writes the batch's tiles only, and every chain that reaches it began with a step-1 launch
        //  into this V for this batch, which zeroed the tail)
    }
    if (steps & 4) {
        const long want = (tiles + 3) / 4;
        const int grid_out = (int)(want < (long)num_cus * 8 ? want : (long)num_cus * 8);
        if (heads) hipLaunchKernelGGL((c6_wino4_out_kernel<N, true>), dim3(grid_out), dim3(256), 0, s, Mb, bias, res, out, tiles, relu, *heads, (float *)nullptr);
        else if (stat_partial && bn_bwd)
            hipLaunchKernelGGL((c6_wino4_out_kernel<N, false, 2>), dim3(grid_out), dim3(256), 0, s, Mb, bias, res, out, tiles, relu, C6HeadFuse{}, stat_partial,
                               C6BnBwdSrc{bn_bwd[0], bn_bwd[1], bn_bwd[2], bn_bwd[3]});
        else if (stat_partial) hipLaunchKernelGGL((c6_wino4_out_kernel<N, false, 1>), dim3(grid_out), dim3(256), 0, s, Mb, bias, res, out, tiles, relu, C6HeadFuse{}, stat_partial);
        else hipLaunchKernelGGL((c6_wino4_out_kernel<N, false>), dim3(grid_out), dim3(256), 0, s, Mb, bias, res, out, tiles, relu, C6HeadFuse{}, (float *)nullptr);
    }
}

// steps: bit 0 input transform, bit 1 GEMMs, bit 2 output transform (7 = the layer); bit 4: output transform fused with the NEXT layer's input transform
bool c6_conv_wino4_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out,
                            float *vbuf, int num_cus, hipStream_t s, int relu, int steps, const C6HeadFuse *heads, const C6StemSrc *stem, float *zsep, float *stat_partial, const float *const *bn_bwd) {
    if (C != W4_C) return false;
    if (nboards <= 0) return true;
    if (N == 19) { launch_w4<19>(nboards, in, wp, bias, res, out, vbuf, num_cus, relu, steps, s, heads, stem, zsep, stat_partial, bn_bwd); return true; }
    if (N == 15) { launch_w4<15>(nboards, in, wp, bias, res, out, vbuf, num_cus, relu, steps, s, heads, stem, zsep, stat_partial, bn_bwd); return true; }
    if (N == 9) { launch_w4<9>(nboards, in, wp, bias, res, out, vbuf, num_cus, relu, steps, s, heads, stem, zsep, stat_partial, bn_bwd); return true; }
    return false;
}
// workgroups of the output transform's launch (= the chunks of `stat_partial`: [chunks][2][C] floats)
int c6_conv_wino4_out_grid(int N, int nboards, int num_cus) {
    const long tiles = (long)nboards * (N == 19 ? W4Geom<19>::T : N == 15 ? W4Geom<15>::T : W4Geom<9>::T), want = (tiles + 3) / 4;
    return (int)(want < (long)num_cus * 8 ? want : (long)num_cus * 8);
}
size_t c6_conv_wino4_v_floats(int C, int N, int nboards) { return C == W4_C ? w4_half_floats(N, nboards) : 0; }
// matrix FLOPs the GEMM kernel executes for `nboards` boards (whole 64-tile blocks x 36 or 30 positions x C x C MACs)
template <int N>
static double w4_exec_flops(int C, int nboards) {
    const W4Map<N> m(nboards);
    return 2.0 * ((double)W4_NP * m.blkB + (double)W4_NP_EDGE * (m.nblocks - m.blkB)) * W4_BT * (double)C * C;
}
double c6_conv_wino4_exec_flops(int C, int N, int nboards) {
    return N == 19 ? w4_exec_flops<19>(C, nboards) : N == 15 ? w4_exec_flops<15>(C, nboards) : w4_exec_flops<9>(C, nboards);
}

// ---- weight gradient in the Winograd domain (the trainer's FP32 path, trainer.cpp) -----------------------------------------------
// Y = A^T [U .* V] A per tile, so  dU_p[co][ci] = sum over tiles (A dY A^T)_p[co] V_p[ci]  and  dg = G_v^T dU G_h: the 3x3 filter's gradient from
// 846 (19x19) / 529 (15x15) products per board and channel pair instead of 9 N^2 -- 3.8x fewer matrix FLOPs than the direct
// form (c6_wgrad_kernel, train_kernels.hip), in f32 like it.  Three kernels behind the input transform of x (c6_wino4_in_kernel):
//   c6_wino4_dyt_kernel    dY (NHWC) -> D = A dY A^T in V's layout (regions, classes and the transposed class B as for V)
//   c6_wino4_wgrad_kernel  per (class, position p, chunk of blocks):  partial[co][ci] = sum_tiles D_p[tile][co] V_p[tile][ci]
//   c6_wino4_wgrad_sum_kernel / _finish_kernel   sum the chunks; apply G_v^T . G_h of each class: dW (OIHW)
// A (6x4) = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1],  F(3): (5x3) = [1 0 0; 1 1 1; 1 -1 1; 1 2 4; 0 0 1]
#define W4_A4_ROW(i, y0, y1, y2, y3)                                                                  \
    ((i) == 0 ? (y0)                                                                                  \
   : (i) == 1 ? ((y0) + (y2)) + ((y1) + (y3))                                                         \
   : (i) == 2 ? ((y0) + (y2)) - ((y1) + (y3))                                                         \
   : (i) == 3 ? w4_fma(4.f, y2, y0) + w4_fma(8.f, y3, (y1) + (y1))                                    \
   : (i) == 4 ? w4_fma(4.f, y2, y0) - w4_fma(8.f, y3, (y1) + (y1))                                    \
              : (y3))
#define W4_A3_ROW(i, y0, y1, y2)                                                                      \
    ((i) == 0 ? (y0) : (i) == 1 ? ((y0) + (y2)) + (y1) : (i) == 2 ? ((y0) + (y2)) - (y1) : (i) == 3 ? w4_fma(4.f, y2, y0) + ((y1) + (y1)) : (y2))
template <int N>
__global__ __launch_bounds__(256) void c6_wino4_dyt_kernel(const float *__restrict__ dy, float *__restrict__ D, long total_tiles) {
    using G = W4Geom<N>;
    constexpr int C = G::C, T = G::T, TW = G::TW;
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const W4Map<N> map(total_tiles / T);
#pragma unroll 1
    for (long t = wave0; t < total_tiles; t += nwaves) {
        const long bd = t / T;
        const int tt = (int)(t - bd * T), ty = tt / TW, tx = tt - ty * TW, cls = W4Map<N>::cls(ty, tx);
        const long sl = map.slot(bd, ty, tx);
        float *db = D + (size_t)(sl / W4_BT) * G::BLOCK_FLOATS + (size_t)(sl % W4_BT) * G::TILE_STRIDE + lane * 4;
        f32x4 y[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int py = 4 * ty + a, px = 4 * tx + b;
                const bool inb = py < N && px < N;
                y[a][b] = inb ? *reinterpret_cast<const f32x4 *>(dy + ((size_t)bd * (N * N) + (size_t)(py * N + px)) * C + lane * 4) : zero4;
            }
        if (cls == W4_CLS_A) {
            f32x4 r[6][4];      // vertical: r[i][b] = sum_a A[i][a] y[a][b]
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int i = 0; i < 6; i++) r[i][b] = W4_A4_ROW(i, y[0][b], y[1][b], y[2][b], y[3][b]);
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) *reinterpret_cast<f32x4 *>(db + (size_t)(i * 6 + j) * C) = W4_A4_ROW(j, r[i][0], r[i][1], r[i][2], r[i][3]);
        } else if (cls == W4_CLS_BP) {      // 3 output rows: F(3) vertically, F(4) horizontally
            f32x4 r[5][4];
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int i = 0; i < 5; i++) r[i][b] = W4_A3_ROW(i, y[0][b], y[1][b], y[2][b]);
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) *reinterpret_cast<f32x4 *>(db + (size_t)(i * 6 + j) * C) = W4_A4_ROW(j, r[i][0], r[i][1], r[i][2], r[i][3]);
        } else {                            // 3 output columns: F(4) vertically, F(3) horizontally, stored transposed
            f32x4 r[6][3];
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int i = 0; i < 6; i++) r[i][b] = W4_A4_ROW(i, y[0][b], y[1][b], y[2][b], y[3][b]);
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 5; j++) *reinterpret_cast<f32x4 *>(db + (size_t)(j * 6 + i) * C) = W4_A3_ROW(j, r[i][0], r[i][1], r[i][2]);
        }
    }
}

// the work of the GEMM kernel: per class its blocks cut into chunks of `per`; item = (class, position, chunk)
struct W4WgPlan {
    int first_blk[3], nblk[3], chunks[3], item0[3], per, items;
};
template <int N>
static W4WgPlan w4_wg_plan(int nboards, int num_cus) {
    const W4Map<N> m(nboards);
    W4WgPlan p;
    p.first_blk[0] = 0; p.first_blk[1] = (int)m.blkB; p.first_blk[2] = (int)m.blkBP;
    p.nblk[0] = (int)m.blkB; p.nblk[1] = (int)(m.blkBP - m.blkB); p.nblk[2] = (int)(m.nblocks - m.blkBP);
    const long units = (long)W4_NP * p.nblk[0] + (long)W4_NP_EDGE * (p.nblk[1] + p.nblk[2]);
    // two rounds of workgroups over the CUs, and not one item more: the smallest chunk size whose items fit into 2 x num_cus (a 513th item
    // would run alone for a third round)
    p.per = (int)((units + 2L * num_cus - 1) / (2L * num_cus));
    if (p.per < 1) p.per = 1;
    for (;; p.per++) {
        p.items = 0;
        for (int s = 0; s < 3; s++) {
            p.chunks[s] = (p.nblk[s] + p.per - 1) / p.per;
            p.item0[s] = p.items;
            p.items += (s == 0 ? W4_NP : W4_NP_EDGE) * p.chunks[s];
        }
        if (p.items <= 2 * num_cus || (p.chunks[0] <= 1 && p.chunks[1] <= 1 && p.chunks[2] <= 1)) break;
    }
    return p;
}
// Workgroup of 8 waves: wave w owns couts [32w, 32w + 32) x all 256 cins (8 accumulator tiles); K = the tiles of the item's blocks, 32
// at a time: the 32 rows of D_p and of V_p (1 KiB each, stride one tile) land in LDS by LDS-DMA one stage ahead (rows padded to C + 32
// floats: the two K rows of a 32x32x2 operand fall into different halves of the banks); per K step of 2 tiles one LDS read for the D
// fragment and eight for the V fragments feed eight MFMAs.
#define W4G_CP (W4_C + 32)
#define W4G_ROWS 32
template <int N>
__global__ __launch_bounds__(512) void c6_wino4_wgrad_kernel(const float *__restrict__ D, const float *__restrict__ V, float *__restrict__ partial, W4WgPlan P) {
    using G = W4Geom<N>;
    constexpr int C = G::C, CP = W4G_CP, STG = W4G_ROWS * CP;      // floats of one operand's stage
    extern __shared__ __attribute__((aligned(16))) float wsm[];      // [2 stages][D, V][32][CP]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = blockIdx.x;
    const int set = item >= P.item0[2] ? 2 : item >= P.item0[1] ? 1 : 0;
    const int local = item - P.item0[set], pos = local / P.chunks[set], chunk = local - pos * P.chunks[set];
    const int b0 = P.first_blk[set] + chunk * P.per;
    const int b1 = b0 + P.per < P.first_blk[set] + P.nblk[set] ? b0 + P.per : P.first_blk[set] + P.nblk[set];
    const int nst = 2 * (b1 - b0);                                   // stages of 32 tiles
    const unsigned lane16 = (unsigned)lane * 16u;
    // stage st -> buffer st & 1: wave w copies rows 4w .. 4w + 3 of D and of V
    auto dma = [&](int st) {
        const int blk = b0 + (st >> 1), half = st & 1, buf = st & 1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = wave * 4 + r;
            const size_t src = (size_t)blk * G::BLOCK_FLOATS + (size_t)(half * W4G_ROWS + row) * G::TILE_STRIDE + (size_t)pos * C;
            const float *gd = D + src, *gv = V + src;
            const int ld = ((buf * 2 + 0) * STG + row * CP) * 4, lv = ((buf * 2 + 1) * STG + row * CP) * 4;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(gd), "s"(ld) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(gv), "s"(lv) : "memory");
        }
    };
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[j][v] = 0.f;
    const int i = lane & 31, kk = lane >> 5;
    dma(0);
#pragma unroll 1
    for (int st = 0; st < nst; st++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                             // stage st has landed; everyone is done with the other buffer
        if (st + 1 < nst) dma(st + 1);
        const float *dl = wsm + ((st & 1) * 2 + 0) * STG + kk * CP + wave * 32 + i;
        const float *vl = wsm + ((st & 1) * 2 + 1) * STG + kk * CP + i;
#pragma unroll 4
        for (int s = 0; s < W4G_ROWS / 2; s++) {
            const float a = dl[2 * s * CP];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vl[2 * s * CP + 32 * j], acc[j], 0, 0, 0);
        }
    }
    // partial[item][co][ci]: acc[j][v] = (co = 32 wave + 8 (v / 4) + 4 kk + v % 4, ci = 32 j + i)
    float *pb = partial + (size_t)item * C * C;
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int v = 0; v < 16; v++) pb[(size_t)(32 * wave + 8 * (v / 4) + 4 * kk + (v % 4)) * C + 32 * j + i] = acc[j][v];
}
// dU[96 positions][co][ci] = the sum of an item's chunks in chunk order (positions of the three classes one behind the other)
__global__ void c6_wino4_wgrad_sum_kernel(int C, const float *__restrict__ partial, float *__restrict__ du, W4WgPlan P) {
    const size_t cc = (size_t)C * C, total = (size_t)(W4_NP + 2 * W4_NP_EDGE) * cc;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
        const int gp = (int)(k / cc);
        const size_t e = k - (size_t)gp * cc;
        const int set = gp < W4_NP ? 0 : gp < W4_NP + W4_NP_EDGE ? 1 : 2, pos = gp - (set == 0 ? 0 : set == 1 ? W4_NP : W4_NP + W4_NP_EDGE);
        float v = 0.f;
        for (int c = 0; c < P.chunks[set]; c++) v += partial[(size_t)(P.item0[set] + pos * P.chunks[set] + c) * cc + e];
        du[k] = v;
    }
}
// dW[co][ci][a][b] (OIHW) = sum over classes and positions  G_v[.][a] dU_p G_h[.][b]
__global__ void c6_wino4_wgrad_finish_kernel(int C, const float *__restrict__ du, float *__restrict__ dw) {
    const double G4[6][3] = {{1.0 / 4, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const double G3[5][3] = {{1.0 / 2, 0.0, 0.0}, {-1.0 / 2, -1.0 / 2, -1.0 / 2}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0.0, 0.0, 1.0}};
    const size_t cc = (size_t)C * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < cc; e += (size_t)gridDim.x * blockDim.x) {
        double g[9];
#pragma unroll
        for (int k = 0; k < 9; k++) g[k] = 0.0;
#pragma unroll
        for (int gp = 0; gp < W4_NP + 2 * W4_NP_EDGE; gp++) {
            const int set = gp < W4_NP ? 0 : gp < W4_NP + W4_NP_EDGE ? 1 : 2, pos = gp - (set == 0 ? 0 : set == 1 ? W4_NP : W4_NP + W4_NP_EDGE);
            const double v = (double)du[(size_t)gp * cc + e];
            const int p = pos / 6, q = pos % 6;
            const double *gv = set == 0 ? G4[p] : set == 1 ? G4[q] : G3[p];
            const double *gh = set == 0 ? G4[q] : set == 1 ? G3[p] : G4[q];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) g[a * 3 + b] += gv[a] * v * gh[b];
        }
#pragma unroll
        for (int k = 0; k < 9; k++) dw[e * 9 + k] = (float)g[k];
    }
}

template <int N>
static void launch_w4_wgrad(int nboards, const float *dy, const float *x, float *vbuf, float *partial, float *dw, int num_cus, hipStream_t s, const float *vkept) {
    using G = W4Geom<N>;
    const long tiles = (long)nboards * G::T;
    const float *Vb = vkept ? vkept : vbuf;      // the forward pass's B^T x B of this layer, if the caller kept it
    float *Db = vbuf + w4_half_floats(N, nboards);
    if (!vkept) launch_w4<N>(nboards, x, nullptr, nullptr, nullptr, nullptr, vbuf, num_cus, 0, 1, s, nullptr, nullptr);      // V = B^T x B
    const long want = (tiles + 3) / 4;
    const int grid_t = (int)(want < (long)num_cus * 8 ? want : (long)num_cus * 8);
    hipLaunchKernelGGL((c6_wino4_dyt_kernel<N>), dim3(grid_t), dim3(256), 0, s, dy, Db, tiles);
    if (w4_has_tail<N>(nboards)) hipLaunchKernelGGL((c6_wino4_tail_kernel<N>), dim3(256), dim3(256), 0, s, Db, nboards);
    const W4WgPlan P = w4_wg_plan<N>(nboards, num_cus);
    constexpr size_t smem = (size_t)4 * W4G_ROWS * W4G_CP * 4;
    w4_smem_once(reinterpret_cast<const void *>(&c6_wino4_wgrad_kernel<N>), (int)smem);
    hipLaunchKernelGGL((c6_wino4_wgrad_kernel<N>), dim3(P.items), dim3(512), smem, s, Db, Vb, partial, P);
    float *du = partial + (size_t)P.items * W4_C * W4_C;
    hipLaunchKernelGGL(c6_wino4_wgrad_sum_kernel, dim3(2048), dim3(256), 0, s, W4_C, partial, du, P);
    hipLaunchKernelGGL(c6_wino4_wgrad_finish_kernel, dim3(256), dim3(256), 0, s, W4_C, du, dw);
}
size_t c6_conv_wino4_wgrad_partial_floats(int C, int N, int nboards, int num_cus) {
    if (C != W4_C) return 0;
    const W4WgPlan P = N == 19 ? w4_wg_plan<19>(nboards, num_cus) : N == 15 ? w4_wg_plan<15>(nboards, num_cus) : w4_wg_plan<9>(nboards, num_cus);
    return (size_t)(P.items + W4_NP + 2 * W4_NP_EDGE) * C * C;      // the items' partial sums + dU
}
// dW (OIHW, overwritten) of the 3x3 convolution y = conv(x, w) from dy; vbuf: c6_conv_wino4_vbuf_floats, partial: the size above
bool c6_conv_wino4_wgrad(int C, int N, int nboards, const float *dy, const float *x, float *vbuf, float *partial, float *dw, int num_cus, hipStream_t s,
                         const float *vkept) {
    if (C != W4_C || nboards <= 0) return false;
    if (N == 19) { launch_w4_wgrad<19>(nboards, dy, x, vbuf, partial, dw, num_cus, s, vkept); return true; }
    if (N == 15) { launch_w4_wgrad<15>(nboards, dy, x, vbuf, partial, dw, num_cus, s, vkept); return true; }
    if (N == 9) { launch_w4_wgrad<9>(nboards, dy, x, vbuf, partial, dw, num_cus, s, vkept); return true; }
    return false;
}
