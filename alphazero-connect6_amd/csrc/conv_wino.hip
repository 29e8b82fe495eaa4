// conv_wino.hip -- FP32 3x3 convolution by Winograd F(2x2, 3x3) on the FP32 matrix cores (v_mfma_f32_32x32x2_f32), for towers of
// 64 and 128 channels; 256 channels (the benchmark's tower) run the F(4x4, 3x3) form of conv_wino4.hip, to which the host
// functions at the end of this file forward.  (F(2x2) at 256 channels, with the transformed input tile-major as in conv_wino4.hip:
// transform 12.0 ms + GEMM 53.0 ms = 64.3 ms per 32768-board layer against 46 ms, profiles/r02_wino_f2_prof.json.)
//
// Same layer as c6_conv3x3_f32n_kernel (conv3x3 C -> C, pad 1, + folded-BN bias (+ residual) + ReLU;
// AlphaZero/cpp_weighted/model.hpp:61-78), same f32 arithmetic (f32 operands, exact f32 products, f32 accumulate), 2.03x fewer
// matrix FLOPs at N = 19 (100 tiles x 16 positions = 1600 products per (cin, cout) instead of 361 x 9 = 3249):
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A,   d = 4x4 input patch of a tile, Y = its 2x2 outputs, summed over cin.
// B^T / A^T hold only 0, +-1 (adds), G's 1/2, 1/4 are folded into the weights on the host (double, rounded once).
//
//  K-in   c6_wino_in_kernel:  activations NHWC -> V[block][pos 16][tile 64][C + 4]  (B^T d B per tile and channel; a block = 64
//         consecutive tiles of the batch, tiles of a board in row-major order, 4 pad floats per row so that the GEMM kernel's
//         LDS image IS this memory image).  HBM bound: reads the activations once (rolling row window in LDS), writes 16/4 x
//         their size.
//  K-gemm c6_wino_gemm_kernel: per block and position p the GEMM  M_p[cout][tile] = sum_cin U_p[cout][cin] V_p[cin][tile]
//         on 32x32x2 MFMAs, the output transform  Y += A^T-coefficient(p) * M_p  accumulated in registers position by
//         position (position-outer order: 2 + 8 accumulator tiles per wave instead of 32), epilogue = bias + residual + ReLU +
//         16-byte NHWC stores of the 2x2 pixels of every tile.
//         Schedule = the f32n one: persistent workgroups; wave w owns output channels [32w, 32w + 32) for all 64 tiles of the
//         block, so its transformed weights are private and stream L2 -> VGPR three 8-channel groups ahead (no weight staging
//         in LDS); V_p (64 x (C + 4) floats) lands in LDS by LDS-DMA as a linear copy one position ahead, double-buffered;
//         ONE barrier per position.
//         Weights are the MFMA A operand (D = U x V): a lane's accumulators are 16-byte NHWC runs of one tile.
// MFMA operand mapping (as in conv_f32n.hip): one ds_read_b128 gives lane (tile j, h) channels 8u + 4h .. + 3 of group u; MFMA r
// of the group contracts the channel pair (8u + r, 8u + 4 + r); weight lane (cout i, h) holds U_p[i][8u + 4h + r].
#include "c6_device.h"
#include "c6_engine.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WINO_BT 64      // tiles per block
#define WINO_PF 7       // weight groups in flight ahead of the one being multiplied (ring of 8)

template <int C, int N>
struct WinoGeom {
    static constexpr int TW = (N + 1) / 2;          // tiles per board side
    static constexpr int T = TW * TW;               // tiles per board (100 / 64 / 25)
    static constexpr int CP = C + 4;                // padded row: conflict-free ds_read_b128 with one tile per lane
    static constexpr int SLAB = WINO_BT * CP;       // floats of one (block, position) slab: V in memory is [block][position 16][tile 64][C + 4],
    static constexpr int NCH = SLAB / 256;          // the LDS image itself, copied linearly in 1-KiB chunks
    static constexpr int POS_STRIDE = SLAB;         // floats between consecutive positions of one tile
    static constexpr int TILE_STRIDE = CP;          // floats between consecutive tiles of one position
    static constexpr size_t BLOCK_FLOATS = (size_t)16 * SLAB;
    static constexpr int WAVES = C / 32;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int NG = C / 8;                // 8-channel groups per position
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 4;
};

// ---- input transform ------------------------------------------------------------------------------------------------
// One workgroup walks over boards; a board's rows pass through a rolling 4-row window in LDS (row y in slot (y + 1) & 3, columns
// x = -1 .. 2 TW with a zero halo), so every activation is read from HBM ONCE although each pixel belongs to up to four 4x4
// patches (the first form of this kernel gathered the patches straight from memory: 35 GB fetched for 12 GB of activations at
// 32 768 boards, profiles/r02_wino_prof_v1.json).  Per tile row: the two new rows are prefetched into registers while the tiles
// of the current row are transformed (C / 4 lanes per tile, each 4 channels: B^T d B is 32 + 32 adds on f32x4) and stored --
// 16 positions x one (C + 4)-float row per tile.
#define WINO_IN_THREADS 320
template <int C>
struct WinoInGeom {
    static constexpr int CS = C;                      // channels per workgroup (halving it to 128 -- three workgroups per CU -- was slower:
    static constexpr int NS = C / CS;                 // 19.3 vs 16.8 ms, the 512-byte runs cost more than the extra waves bring)
};
template <int C, int N>
__global__ __launch_bounds__(WINO_IN_THREADS) void c6_wino_in_kernel(const float *__restrict__ in, float *__restrict__ V, int nboards) {
    using G = WinoGeom<C, N>;
    constexpr int CS = WinoInGeom<C>::CS, NS = WinoInGeom<C>::NS;
    constexpr int LPT = CS / 4, TPI = WINO_IN_THREADS / LPT, CP = G::CP, T = G::T, TW = G::TW, COLS = 2 * TW + 2;
    constexpr int ROWF = COLS * CS;                                 // floats of one window row
    constexpr int NPF = (2 * N * LPT + WINO_IN_THREADS - 1) / WINO_IN_THREADS;      // prefetch items (f32x4) per thread and step
    extern __shared__ __attribute__((aligned(16))) float win[];    // [4][COLS][CS]
    const int tid = threadIdx.x, cq = tid % LPT, tsub = tid / LPT;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < 4 * ROWF / 4; k += WINO_IN_THREADS) reinterpret_cast<f32x4 *>(win)[k] = zero4;      // halo columns stay zero
    // item k of a 2-row load: row y0 + k / (N LPT), pixel (k / LPT) % N, channel quad k % LPT; rows outside the board are zeros
    auto item_src = [&](const float *ib, int y0, int k, bool &ok) -> const float * {
        const int r = k / (N * LPT), px = (k / LPT) % N, q = k % LPT, y = y0 + r;
        ok = k < 2 * N * LPT && y >= 0 && y < N;
        return ib + (size_t)((ok ? y : 0) * N + px) * C + q * 4;
    };
    auto item_dst = [&](int y0, int k) -> float * {
        const int r = k / (N * LPT), px = (k / LPT) % N, q = k % LPT, y = y0 + r;
        return win + ((y + 1) & 3) * ROWF + (px + 1) * CS + q * 4;
    };
#pragma unroll 1
    for (int item = blockIdx.x; item < nboards * NS; item += gridDim.x) {
        const int bd = item / NS, c0 = (item % NS) * CS;           // board and first channel of this work item
        const float *ib = in + (size_t)bd * (N * N) * C + c0;
        __syncthreads();                                            // the previous item's last reads (and the zero fill)
#pragma unroll
        for (int half = 0; half < 2; half++)                        // rows -1, 0 and 1, 2
#pragma unroll
            for (int i = 0; i < NPF; i++) {
                const int k = tid + i * WINO_IN_THREADS;
                bool ok;
                const float *sp = item_src(ib, 2 * half - 1, k, ok);
                const f32x4 v = ok ? *reinterpret_cast<const f32x4 *>(sp) : zero4;
                if (k < 2 * N * LPT) *reinterpret_cast<f32x4 *>(item_dst(2 * half - 1, k)) = v;
            }
#pragma unroll 1
        for (int ty = 0; ty < TW; ty++) {
            __syncthreads();                                        // rows 2 ty - 1 .. 2 ty + 2 are in the window
            f32x4 pf[NPF];
            const bool more = ty + 1 < TW;
            if (more) {
#pragma unroll
                for (int i = 0; i < NPF; i++) {
                    bool ok;
                    const float *sp = item_src(ib, 2 * ty + 3, tid + i * WINO_IN_THREADS, ok);
                    pf[i] = ok ? *reinterpret_cast<const f32x4 *>(sp) : zero4;
                }
            }
#pragma unroll 1
            for (int tx = tsub; tx < TW; tx += TPI) {
                const long t = (long)bd * T + ty * TW + tx;
                float *vb = V + (size_t)(t / WINO_BT) * G::BLOCK_FLOATS + (size_t)(t % WINO_BT) * G::TILE_STRIDE + c0 + cq * 4;
                f32x4 d[4][4];
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        d[r][c] = *reinterpret_cast<const f32x4 *>(win + ((2 * ty + r) & 3) * ROWF + (2 * tx + c) * CS + cq * 4);      // row 2 ty - 1 + r, column 2 tx - 1 + c
                f32x4 w[4][4];
#pragma unroll
                for (int c = 0; c < 4; c++) {                       // B^T d: rows
                    w[0][c] = d[0][c] - d[2][c];
                    w[1][c] = d[1][c] + d[2][c];
                    w[2][c] = d[2][c] - d[1][c];
                    w[3][c] = d[1][c] - d[3][c];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {                       // (B^T d) B: columns
                    const f32x4 v0 = w[i][0] - w[i][2], v1 = w[i][1] + w[i][2], v2 = w[i][2] - w[i][1], v3 = w[i][1] - w[i][3];
                    float *o = vb + (size_t)(i * 4) * G::POS_STRIDE;
                    *reinterpret_cast<f32x4 *>(o) = v0;
                    *reinterpret_cast<f32x4 *>(o + G::POS_STRIDE) = v1;
                    *reinterpret_cast<f32x4 *>(o + 2 * G::POS_STRIDE) = v2;
                    *reinterpret_cast<f32x4 *>(o + 3 * G::POS_STRIDE) = v3;
                }
            }
            if (more) {
                __syncthreads();                                    // rows 2 ty - 1, 2 ty are read no more: their slots take rows 2 ty + 3, 2 ty + 4
#pragma unroll
                for (int i = 0; i < NPF; i++) {
                    const int k = tid + i * WINO_IN_THREADS;
                    if (k < 2 * N * LPT) *reinterpret_cast<f32x4 *>(item_dst(2 * ty + 3, k)) = pf[i];
                }
            }
        }
    }
}

// the tail of the last block (tiles beyond the batch) is never written by the kernel above: zero it once per launch so that the GEMM
// multiplies zeros, not stale bits (their products only reach accumulators that are never stored, but NaNs would still be ugly)
template <int C, int N>
__global__ void c6_wino_tail_kernel(float *__restrict__ V, long total_tiles) {
    using G = WinoGeom<C, N>;
    const long nblocks = (total_tiles + WINO_BT - 1) / WINO_BT;
    const int first = (int)(total_tiles - (nblocks - 1) * WINO_BT);      // valid tiles of the last block
    float *blk = V + (size_t)(nblocks - 1) * G::BLOCK_FLOATS;
    const int row = G::CP, n = (WINO_BT - first) * 16 * row;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int t = first + i / (16 * row), r = i % (16 * row), p = r / row, c = r % row;
        blk[(size_t)t * G::TILE_STRIDE + (size_t)p * G::POS_STRIDE + c] = 0.f;
    }
}

// ---- GEMM + output transform ---------------------------------------------------------------------------------------
// Every vector-memory operation of the main loop is issued by hand (inline asm) because vmcnt retires IN ORDER: a weight load
// issued behind the slab DMAs cannot be waited for without waiting for them.  With the compiler's own loads its s_waitcnt
// counts do not see the DMAs and wait for far too much; here each wait names exactly the operations that are younger than
// the group it needs: the WINO_PF weight groups behind it, plus the ND slab DMAs while those are younger too.
// operations younger than the weights of group i when they are waited for, SPREAD schedule (see the kernel)
constexpr int wino_wait(int i, int nd) {
    const int lo = i - WINO_PF > 0 ? i - WINO_PF : 0, hi = (nd - 1 < i - 1) ? nd - 1 : i - 1;
    return WINO_PF + (hi >= lo ? hi - lo + 1 : 0);
}
#define WINO_ULOAD(dst, voff, base) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#define WINO_UWAIT(dst, n) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(dst) : "n"(n) : "memory")

template <int C, int N>
__global__ __launch_bounds__(64 * (C / 32)) void c6_wino_gemm_kernel(const float *__restrict__ V, const float *__restrict__ U,
                                                                     const float *__restrict__ bias, const float *__restrict__ res,
                                                                     float *__restrict__ out, long total_tiles, int nblocks, int relu) {
    using G = WinoGeom<C, N>;
    constexpr int CP = G::CP, SLAB = G::SLAB, NCH = G::NCH, WAVES = G::WAVES, NG = G::NG, T = G::T, TW = G::TW, PF = WINO_PF;
    constexpr int NGT = 16 * NG;                     // weight groups of a whole block (all 16 positions)
    constexpr int ND = (NCH + WAVES - 1) / WAVES;    // slab DMAs per wave and position (the same count in every wave)
    static_assert(PF == 7 && NG % 8 == 0, "ring of 8 weight groups, 7 in flight");
    constexpr bool SPREAD = NG >= 16 && ND <= 16;    // one slab DMA per group instead of all of them behind the barrier (C >= 128)
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n4 = wave * 32 + 4 * h;
    int b = blockIdx.x;
    if (b >= nblocks) return;

    // this wave's transformed weights: [wave][pos][group][lane][4] floats, one linear 1-KiB-per-group stream per block
    const float *ubase = U + (size_t)wave * NGT * 256;
    const unsigned lane16 = (unsigned)lane * 16u;

    // LDS-DMA of one (block, position) slab: NCH linear 1-KiB chunks, chunk c by wave c mod WAVES; a wave whose last turn
    // falls behind the slab copies the last chunk again (same bytes, same place) so that every wave issues ND operations
    // (scalar chunk address + the lane's constant 16-byte offset: no address VGPRs)
    auto dma_chunk = [&](const float *src, int buf, int c0) {
        int c = c0 * WAVES + wave;
        c = c < NCH ? c : NCH - 1;
        const float *gp = src + (size_t)c * 256;
        const int lds_off = (buf * SLAB + c * 256) * 4;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(gp), "s"(lds_off) : "memory");
    };
    auto dma_slab = [&](const float *src, int buf) {
#pragma unroll
        for (int c0 = 0; c0 < ND; c0++) dma_chunk(src, buf, c0);
    };

    dma_slab(V + (size_t)b * G::BLOCK_FLOATS, 0);
    f32x4 ring[8];
#pragma unroll
    for (int q = 0; q < PF; q++) WINO_ULOAD(ring[q], lane16 + (unsigned)q * 1024u, ubase);
    int gi = PF;                                     // next weight group to request, 0 .. NGT-1 (wraps: the next block's)

#pragma unroll 1
    for (; b < nblocks; b += gridDim.x) {
        // y[a][b][hf][4g + r] = channel 32 wave + 8g + 4h + r of pixel (2 ty + a, 2 tx + b) of tile 32 hf + j.  The accumulators start
        // as bias (+ residual): all the block's residual loads are in flight at once and land a whole position before they are used
        f32x16 y[2][2][2];
        // element offset of pixel (a, bb) of tile half hf for this lane, ok = inside the batch and the board
        auto pix = [&](int hf, int a, int bb, bool &ok) -> size_t {
            const long t = (long)b * WINO_BT + hf * 32 + j;
            const bool tv = t < total_tiles;
            const long bd = tv ? t / T : 0;
            const int tt = tv ? (int)(t - bd * T) : 0, ty = tt / TW, tx = tt - ty * TW;
            const int py = 2 * ty + a, px = 2 * tx + bb;
            ok = tv && py < N && px < N;
            return ok ? ((size_t)bd * (N * N) + (size_t)(py * N + px)) * C + n4 : (size_t)n4;      // clamped: loads stay unconditional
        };
        {
            f32x4 bi[4];
#pragma unroll
            for (int g = 0; g < 4; g++) bi[g] = bias ? *reinterpret_cast<const f32x4 *>(bias + n4 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (res) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int bb = 0; bb < 2; bb++) {
                            bool ok;
                            const float *rp = res + pix(hf, a, bb, ok);
#pragma unroll
                            for (int g = 0; g < 4; g++) {
                                const f32x4 rv = *reinterpret_cast<const f32x4 *>(rp + 8 * g);
#pragma unroll
                                for (int r = 0; r < 4; r++) y[a][bb][hf][4 * g + r] = rv[r] + bi[g][r];
                            }
                        }
            } else {
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int bb = 0; bb < 2; bb++)
#pragma unroll
                            for (int v = 0; v < 16; v++) y[a][bb][hf][v] = bi[v >> 2][v & 3];
            }
        }

        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), as a builtin: the compiler's own wait bookkeeping sees the residual loads retired
#pragma unroll 1
        for (int p = 0; p < 16; p++) {
            // the slab of position p (issued one position ago) has landed: only the PF youngest operations -- weight
            // requests, all younger than those DMAs -- may still be in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF) : "memory");
            __syncthreads();
            const float *nsrc;
            {
                int nb = p < 15 ? b : b + (int)gridDim.x;
                nb = nb < nblocks ? nb : b;          // no next block: copy this block's slab 0 again (never read)
                nsrc = V + (size_t)nb * G::BLOCK_FLOATS + (size_t)(p < 15 ? p + 1 : 0) * G::POS_STRIDE;
            }
            if (!SPREAD) dma_slab(nsrc, (p + 1) & 1);
            const float *vb = wsm + (p & 1) * SLAB + j * CP + 4 * h;
            f32x16 m0, m1;
#pragma unroll
            for (int v = 0; v < 16; v++) { m0[v] = 0.f; m1[v] = 0.f; }
            // one 8-channel group: request group + PF, fetch the next group's tile fragments, wait for this group's weights,
            // (SPREAD: issue one slab DMA in the shadow of the MFMAs), 2 x 4 MFMAs.
            // (u) + 1 == NG reads one group into the row padding / the next row: loaded, never used.
            f32x4 va0 = *reinterpret_cast<const f32x4 *>(vb), va1 = *reinterpret_cast<const f32x4 *>(vb + 32 * CP), vn0, vn1;
#define WINO_GROUP(q, u, NWAIT, DMA)                                                                             \
            {                                                                                                    \
                WINO_ULOAD(ring[((q) + PF) & 7], lane16 + (unsigned)gi * 1024u, ubase);                          \
                gi = gi + 1 == NGT ? 0 : gi + 1;                                                                 \
                vn0 = *reinterpret_cast<const f32x4 *>(vb + 8 * ((u) + 1 < NG ? (u) + 1 : (u)));                 \
                vn1 = *reinterpret_cast<const f32x4 *>(vb + 32 * CP + 8 * ((u) + 1 < NG ? (u) + 1 : (u)));       \
                WINO_UWAIT(ring[(q)], NWAIT);                                                                    \
                const f32x4 w = ring[(q)];                                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va0[0], m0, 0, 0, 0);                            \
                m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va1[0], m1, 0, 0, 0);                            \
                if ((DMA) >= 0 && (DMA) < ND) dma_chunk(nsrc, (p + 1) & 1, (DMA));                               \
                _Pragma("unroll") for (int r = 1; r < 4; r++) {                                                  \
                    m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va0[r], m0, 0, 0, 0);                        \
                    m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va1[r], m1, 0, 0, 0);                        \
                }                                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                va0 = vn0; va1 = vn1;                                                                            \
            }
            if (SPREAD) {
                // DMA k is issued in group k, behind that group's weight wait: younger than the weights of group i are its 7
                // followers plus the DMAs k with max(0, i - 7) <= k <= min(ND - 1, i - 1)   (wino_wait)
#define WG_(i) WINO_GROUP((i) & 7, (i), wino_wait((i), ND), (i))
                WG_(0) WG_(1) WG_(2) WG_(3) WG_(4) WG_(5) WG_(6) WG_(7) WG_(8) WG_(9) WG_(10) WG_(11) WG_(12) WG_(13) WG_(14) WG_(15)
#undef WG_
#pragma unroll 1
                for (int u0 = 16; u0 < NG; u0 += 8) {
                    WINO_GROUP(0, u0, PF, -1) WINO_GROUP(1, u0 + 1, PF, -1) WINO_GROUP(2, u0 + 2, PF, -1) WINO_GROUP(3, u0 + 3, PF, -1)
                    WINO_GROUP(4, u0 + 4, PF, -1) WINO_GROUP(5, u0 + 5, PF, -1) WINO_GROUP(6, u0 + 6, PF, -1) WINO_GROUP(7, u0 + 7, PF, -1)
                }
            } else {
                // all ND DMAs went out before group 0; groups 0..6 were requested before them: those DMAs are younger too
                WINO_GROUP(0, 0, PF + ND, -1) WINO_GROUP(1, 1, PF + ND, -1) WINO_GROUP(2, 2, PF + ND, -1) WINO_GROUP(3, 3, PF + ND, -1)
                WINO_GROUP(4, 4, PF + ND, -1) WINO_GROUP(5, 5, PF + ND, -1) WINO_GROUP(6, 6, PF + ND, -1) WINO_GROUP(7, 7, PF, -1)
#pragma unroll 1
                for (int u0 = 8; u0 < NG; u0 += 8) {
                    WINO_GROUP(0, u0, PF, -1) WINO_GROUP(1, u0 + 1, PF, -1) WINO_GROUP(2, u0 + 2, PF, -1) WINO_GROUP(3, u0 + 3, PF, -1)
                    WINO_GROUP(4, u0 + 4, PF, -1) WINO_GROUP(5, u0 + 5, PF, -1) WINO_GROUP(6, u0 + 6, PF, -1) WINO_GROUP(7, u0 + 7, PF, -1)
                }
            }
#undef WINO_GROUP
            // output transform, one position's share: Y[a][b] += A^T[a][i] A^T[b][j] M_(i,j)   (A^T = [1 1 1 0; 0 1 -1 -1]).  One fma
            // per accumulator register: the coefficients are 0 / +-1, so c * m is exact and the result is the plain sum.  (Branch-free
            // on purpose: a switch over the 16 positions that skips the zero coefficients costs 24 more VGPRs -- spills.)
            const int pi = p >> 2, pj = p & 3;
            const float ai0 = pi < 3 ? 1.f : 0.f, ai1 = pi == 0 ? 0.f : (pi == 1 ? 1.f : -1.f);
            const float bj0 = pj < 3 ? 1.f : 0.f, bj1 = pj == 0 ? 0.f : (pj == 1 ? 1.f : -1.f);
            const float c00 = ai0 * bj0, c01 = ai0 * bj1, c10 = ai1 * bj0, c11 = ai1 * bj1;
#pragma unroll
            for (int v = 0; v < 16; v++) {
                y[0][0][0][v] = __builtin_fmaf(c00, m0[v], y[0][0][0][v]); y[0][0][1][v] = __builtin_fmaf(c00, m1[v], y[0][0][1][v]);
                y[0][1][0][v] = __builtin_fmaf(c01, m0[v], y[0][1][0][v]); y[0][1][1][v] = __builtin_fmaf(c01, m1[v], y[0][1][1][v]);
                y[1][0][0][v] = __builtin_fmaf(c10, m0[v], y[1][0][0][v]); y[1][0][1][v] = __builtin_fmaf(c10, m1[v], y[1][0][1][v]);
                y[1][1][0][v] = __builtin_fmaf(c11, m0[v], y[1][1][0][v]); y[1][1][1][v] = __builtin_fmaf(c11, m1[v], y[1][1][1][v]);
            }
        }

        // epilogue: ReLU + 16-byte NHWC stores
        const float floor_ = relu ? 0.f : -INFINITY;
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int bb = 0; bb < 2; bb++)
                {
                    bool ok;
                    float *op = out + pix(hf, a, bb, ok);
                    if (ok) {
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            f32x4 yv;
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const float tv = y[a][bb][hf][4 * g + r];
                                yv[r] = tv > floor_ ? tv : floor_;
                            }
                            *reinterpret_cast<f32x4 *>(op + 8 * g) = yv;
                        }
                    }
                }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last requests (next block's weights, a spare slab) retire before the wave ends
}

// ---- host side -------------------------------------------------------------------------------------------------------
// C = 256 runs the F(4x4, 3x3) form (conv_wino4.hip)
static bool wino4(int C) { return C == 256; }
size_t c6_conv_packed_wino_floats(int C) { return wino4(C) ? c6_conv_packed_wino4_floats(C) : (size_t)16 * C * C; }

// w: folded OIHW [C][C][3][3] -> U = G g G^T (double, rounded once) packed [wave][pos = 4i + j][group u][lane][r]:
//    U_pos[cout = 32 wave + (lane & 31)][cin = 8u + 4(lane >> 5) + r]
void c6_pack_conv_weights_wino(int C, const float *w, float *packed) {
    if (wino4(C)) { c6_pack_conv_weights_wino4(C, w, packed); return; }
    static const double Gm[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int WAVES = C / 32, NG = C / 8;
    for (int wv = 0; wv < WAVES; wv++)
        for (int lane = 0; lane < 64; lane++)
            for (int u = 0; u < NG; u++)
                for (int r = 0; r < 4; r++) {
                    const int co = wv * 32 + (lane & 31), ci = 8 * u + 4 * (lane >> 5) + r;
                    const float *g = w + ((size_t)co * C + ci) * 9;
                    for (int i = 0; i < 4; i++)
                        for (int jj = 0; jj < 4; jj++) {
                            double s = 0.0;
                            for (int a = 0; a < 3; a++)
                                for (int b = 0; b < 3; b++) s += Gm[i][a] * (double)g[a * 3 + b] * Gm[jj][b];
                            packed[((((size_t)wv * 16 + (i * 4 + jj)) * NG + u) * 64 + lane) * 4 + r] = (float)s;
                        }
                }
}

// floats of the transformed-input scratch for `nboards` boards
size_t c6_conv_wino_vbuf_floats(int C, int N, int nboards) {
    if (wino4(C)) return c6_conv_wino4_vbuf_floats(C, N, nboards);
    const long TW = (N + 1) / 2, T = TW * TW, tiles = (long)nboards * T, nblocks = (tiles + WINO_BT - 1) / WINO_BT;
    return (size_t)nblocks * 16 * WINO_BT * (C + 4);
}

template <int C, int N>
static void launch_wino_in(int nboards, const float *in, float *vbuf, int num_cus, hipStream_t s) {
    using G = WinoGeom<C, N>;
    const long tiles = (long)nboards * G::T;
    constexpr size_t in_smem = (size_t)4 * (2 * G::TW + 2) * WinoInGeom<C>::CS * 4;
    int per_cu_in = (int)(((size_t)160 * 1024) / in_smem);
    per_cu_in = per_cu_in < 1 ? 1 : (per_cu_in > 4 ? 4 : per_cu_in);
    const int items = nboards * WinoInGeom<C>::NS;
    const int grid_in = items < num_cus * per_cu_in ? items : num_cus * per_cu_in;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_wino_in_kernel<C, N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)in_smem);
    hipLaunchKernelGGL((c6_wino_in_kernel<C, N>), dim3(grid_in), dim3(WINO_IN_THREADS), in_smem, s, in, vbuf, nboards);
    if (tiles % WINO_BT) hipLaunchKernelGGL((c6_wino_tail_kernel<C, N>), dim3(64), dim3(256), 0, s, vbuf, tiles);
}

template <int C, int N>
static void launch_wino_gemm(int nboards, const float *vbuf, const float *wp, const float *bias, const float *res, float *out, int num_cus, int relu,
                             hipStream_t s) {
    using G = WinoGeom<C, N>;
    const long tiles = (long)nboards * G::T;
    const int nblocks = (int)((tiles + WINO_BT - 1) / WINO_BT);
    const size_t smem = G::LDS_BYTES;
    const int per_cu = (int)(((size_t)160 * 1024) / smem) < 1 ? 1 : (int)(((size_t)160 * 1024) / smem);
    const int wg_per_cu = per_cu > (16 / G::WAVES) ? (16 / G::WAVES) : per_cu;      // two waves per SIMD
    const int grid = nblocks < num_cus * wg_per_cu ? nblocks : num_cus * wg_per_cu;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_wino_gemm_kernel<C, N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((c6_wino_gemm_kernel<C, N>), dim3(grid), dim3(G::THREADS), smem, s, vbuf, wp, bias, res, out, tiles, nblocks, relu);
}

#define WINO_ALL(M) M(128, 19) M(128, 15) M(128, 9) M(64, 19) M(64, 15) M(64, 9)
// step 1 of a layer: activations -> transformed tiles in vbuf
bool c6_conv_wino_transform(int C, int N, int nboards, const float *in, float *vbuf, int num_cus, hipStream_t s) {
    if (nboards <= 0) return true;
    if (wino4(C)) return c6_conv_wino4_dispatch(C, N, nboards, in, nullptr, nullptr, nullptr, nullptr, vbuf, num_cus, s, 1, 1);
#define WINOCASE(CC, NN) if (C == CC && N == NN) { launch_wino_in<CC, NN>(nboards, in, vbuf, num_cus, s); return true; }
    WINO_ALL(WINOCASE)
#undef WINOCASE
    return false;
}
// step 2: the 16 GEMMs + output transform + bias / residual / ReLU (F(4x4): the 36 GEMMs, their results into vbuf's second half)
bool c6_conv_wino_gemm(int C, int N, int nboards, const float *vbuf, const float *wp, const float *bias, const float *res, float *out, int num_cus,
                       hipStream_t s, int relu) {
    if (nboards <= 0) return true;
    if (wino4(C)) return c6_conv_wino4_dispatch(C, N, nboards, nullptr, wp, nullptr, nullptr, nullptr, const_cast<float *>(vbuf), num_cus, s, relu, 2);
#define WINOCASE(CC, NN) if (C == CC && N == NN) { launch_wino_gemm<CC, NN>(nboards, vbuf, wp, bias, res, out, num_cus, relu, s); return true; }
    WINO_ALL(WINOCASE)
#undef WINOCASE
    return false;
}
// step 3, F(4x4) only (0: the form in use has none): output transform + bias / residual / ReLU (2: and the heads' 1x1 filters)
int c6_conv_wino_out(int C, int N, int nboards, float *vbuf, const float *bias, const float *res, float *out, int num_cus, hipStream_t s, int relu,
                     const C6HeadFuse *heads) {
    if (!wino4(C)) return 0;
    if (nboards > 0) c6_conv_wino4_dispatch(C, N, nboards, nullptr, nullptr, bias, res, out, vbuf, num_cus, s, relu, 4, heads);
    return heads ? 2 : 1;
}
// F(4x4) only: the output transform of a layer + the input transform of the layer that reads it, as one kernel; the activations between
// them are stored only if `act` is given (a block's output, the next block's residual); false: not available for this shape
bool c6_conv_wino_mid(int C, int N, int nboards, float *vbuf, const float *bias, const float *res, float *act, int num_cus, hipStream_t s,
                      const C6StemSrc *res_stem) {
    if (!wino4(C)) return false;
    if (nboards <= 0) return true;
    return c6_conv_wino4_dispatch(C, N, nboards, nullptr, nullptr, bias, res, act, vbuf, num_cus, s, 1, 16, nullptr, res_stem);
}
bool c6_conv_wino_stem_transform(int C, int N, int nboards, const C6StemSrc &stem, float *vbuf, float *act, int num_cus, hipStream_t s) {
    if (!wino4(C)) return false;
    if (nboards <= 0) return true;
    return c6_conv_wino4_dispatch(C, N, nboards, nullptr, nullptr, nullptr, nullptr, act, vbuf, num_cus, s, 1, 1, nullptr, &stem);
}
bool c6_conv_wino_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out,
                           float *vbuf, int num_cus, hipStream_t s, int relu) {
    if (!c6_conv_wino_transform(C, N, nboards, in, vbuf, num_cus, s) || !c6_conv_wino_gemm(C, N, nboards, vbuf, wp, bias, res, out, num_cus, s, relu)) return false;
    c6_conv_wino_out(C, N, nboards, vbuf, bias, res, out, num_cus, s, relu);
    return true;
}
// matrix FLOPs the GEMM kernel executes for `nboards` boards (whole 64-tile blocks x 16 positions x C x C MACs)
double c6_conv_wino_exec_flops(int C, int N, int nboards) {
    if (wino4(C)) return c6_conv_wino4_exec_flops(C, N, nboards);
    const long TW = (N + 1) / 2, tiles = (long)nboards * TW * TW, nblocks = (tiles + WINO_BT - 1) / WINO_BT;
    return 2.0 * 16.0 * (double)nblocks * WINO_BT * (double)C * C;
}
