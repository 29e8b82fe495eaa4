// conv_wino4.hip -- FP32 3x3 convolution by Winograd F(4x4, 3x3) on the FP32 matrix cores, C = 256 (the benchmark's tower).
//
// Same layer and the same f32 arithmetic as conv_wino.hip (conv3x3 C -> C, pad 1, + folded-BN bias (+ residual) + ReLU;
// AlphaZero/cpp_weighted/model.hpp:61-78; f32 operands, exact f32 products, f32 accumulate), with 6x6 patches / 4x4 outputs:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A      (points 0, +-1, +-2, inf: Lavin & Gray's matrices)
// 25 tiles x 36 positions = 900 products per (cin, cout) and 19x19 board instead of 1600 for F(2x2) and 3249 direct: 3.6x fewer
// matrix FLOPs than the direct form.  The price: a tile has 16 outputs per (tile, cout), so the output transform cannot be
// accumulated in registers position by position as conv_wino.hip does (16 x 8192 accumulators per wave) -- the 36 products go
// to memory and a third kernel applies A^T . A.  Three launches per layer:
//   c6_wino4_in_kernel    activations NHWC -> V[block][tile 64][pos 36][C]      HBM bound: reads 1x, writes 36/16 x the activations
//   c6_wino4_gemm_kernel  M[block][tile][pos][cout] = sum_cin U_pos[cout][cin] V[block][tile][pos][cin]: conv_wino.hip's schedule
//                         (persistent workgroups, wave w owns couts [32w, 32w+32) x 64 tiles, weights L2 -> VGPR through a ring of
//                         8 groups, V slabs by LDS-DMA one position ahead, one barrier per position) without the accumulators: the
//                         32 result registers of a position are copied aside and stored during the next one
//   c6_wino4_out_kernel   A^T M A + bias (+ residual), ReLU -> activations NHWC    HBM bound: reads 36/16 x + 1x, writes 1x
// Rounding: a layer's error is ~6x that of the direct f32 kernel (2.4e-6 rms of activations of order 1); over the benchmark's net
// the outputs move by < 1e-7 (tests/test_gpu_net.py holds every arithmetic to 1e-5 against the oracle and a float64 PyTorch model).
#include "c6_device.h"
#include "c6_engine.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define W4_BT 64        // tiles per block
#define W4_NP 36        // positions of a 6x6 patch
#define W4_PF 7         // weight groups in flight ahead of the one being multiplied (ring of 8)
#define W4_C 256

template <int N>
struct W4Geom {
    static constexpr int C = W4_C;
    static constexpr int TW = (N + 3) / 4;          // tiles per board side (5 / 4 / 3)
    static constexpr int T = TW * TW;
    static constexpr int CP = C + 4;                // LDS row: conflict-free ds_read_b128 with one tile per lane
    static constexpr int SLAB = W4_BT * CP;         // floats of one (block, position) slab in LDS
    static constexpr int POS_STRIDE = C;            // V / M in memory: [block][tile][pos][C]
    static constexpr int TILE_STRIDE = W4_NP * C;
    static constexpr size_t BLOCK_FLOATS = (size_t)W4_BT * W4_NP * C;
    static constexpr int WAVES = C / 32;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int NG = C / 8;
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 4;
};

__device__ __forceinline__ f32x4 w4_fma(float a, f32x4 x, f32x4 y) { return __builtin_elementwise_fma(f32x4{a, a, a, a}, x, y); }

// ---- input transform: B^T d B, B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1] ----
// One workgroup per board at a time; the board's rows pass through a rolling 6-row window in LDS (row y in slot (y + 1) mod 6,
// columns x = -1 .. 4 TW with a zero halo): every activation is read from HBM once.  Per tile row the four new rows are prefetched
// into registers while the tiles of the current row are transformed -- 64 lanes per tile, each 4 channels; a result row i of the
// patch is formed from the window (B^T's rows have 3-4 non-zeros) and leaves as six 1-KiB runs.
#define W4_IN_THREADS 320
// row i of B^T applied along the first index of x[6]
#define W4_BT_ROW(i, x0, x1, x2, x3, x4, x5)                                                          \
    ((i) == 0 ? w4_fma(4.f, x0, w4_fma(-5.f, x2, x4))                                                 \
   : (i) == 1 ? w4_fma(-4.f, (x1) + (x2), (x3) + (x4))                                               \
   : (i) == 2 ? w4_fma(4.f, (x1) - (x2), (x4) - (x3))                                                \
   : (i) == 3 ? w4_fma(2.f, (x3) - (x1), (x4) - (x2))                                                \
   : (i) == 4 ? w4_fma(2.f, (x1) - (x3), (x4) - (x2))                                                \
              : w4_fma(4.f, x1, w4_fma(-5.f, x3, x5)))
template <int N>
__global__ __launch_bounds__(W4_IN_THREADS) void c6_wino4_in_kernel(const float *__restrict__ in, float *__restrict__ V, int nboards) {
    using G = W4Geom<N>;
    constexpr int C = G::C, LPT = C / 4, TPI = W4_IN_THREADS / LPT, T = G::T, TW = G::TW, COLS = 4 * TW + 2;
    constexpr int ROWF = COLS * C;
    constexpr int NPF = (4 * N * LPT + W4_IN_THREADS - 1) / W4_IN_THREADS;      // prefetch items (f32x4) per thread and step
    extern __shared__ __attribute__((aligned(16))) float win[];                // [6][COLS][C]
    const int tid = threadIdx.x, cq = tid % LPT, tsub = tid / LPT;
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
                const long t = (long)bd * T + ty * TW + tx;
                float *vb = V + (size_t)(t / W4_BT) * G::BLOCK_FLOATS + (size_t)(t % W4_BT) * G::TILE_STRIDE + cq * 4;
                // window slot of patch row a (board row 4 ty - 1 + a) and the patch's first column
                const float *wc = win + (4 * tx) * C + cq * 4;
                int slot[6];
#pragma unroll
                for (int a = 0; a < 6; a++) slot[a] = ((4 * ty + a) % 6) * ROWF;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    // w[b] = sum_a B^T[i][a] d[a][b]
                    f32x4 w[6];
#pragma unroll
                    for (int b = 0; b < 6; b++) {
                        const float *col = wc + b * C;
#define W4_D(a) (*reinterpret_cast<const f32x4 *>(col + slot[a]))
                        if (i == 0) w[b] = w4_fma(4.f, W4_D(0), w4_fma(-5.f, W4_D(2), W4_D(4)));
                        else if (i == 5) w[b] = w4_fma(4.f, W4_D(1), w4_fma(-5.f, W4_D(3), W4_D(5)));
                        else {
                            const f32x4 d1 = W4_D(1), d2 = W4_D(2), d3 = W4_D(3), d4 = W4_D(4);
                            w[b] = W4_BT_ROW(i, zero4, d1, d2, d3, d4, zero4);
                        }
#undef W4_D
                    }
                    float *o = vb + (size_t)(i * 6) * G::POS_STRIDE;
#pragma unroll
                    for (int jj = 0; jj < 6; jj++)
                        *reinterpret_cast<f32x4 *>(o + jj * G::POS_STRIDE) = W4_BT_ROW(jj, w[0], w[1], w[2], w[3], w[4], w[5]);
                    __builtin_amdgcn_sched_barrier(0);      // one patch row at a time: interleaved, the six rows' window reads spill
                }
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

// the tail of the last block (tiles beyond the batch) is never written by the kernel above: zero it once per launch
template <int N>
__global__ void c6_wino4_tail_kernel(float *__restrict__ V, long total_tiles) {
    using G = W4Geom<N>;
    const long nblocks = (total_tiles + W4_BT - 1) / W4_BT;
    const int first = (int)(total_tiles - (nblocks - 1) * W4_BT);
    float *blk = V + (size_t)(nblocks - 1) * G::BLOCK_FLOATS + (size_t)first * G::TILE_STRIDE;
    const size_t n = (size_t)(W4_BT - first) * G::TILE_STRIDE;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) blk[i] = 0.f;
}

// ---- the 36 GEMMs of a block -------------------------------------------------------------------------------------------
// Vector-memory operations of the main loop retire IN ORDER on one counter (loads, LDS-DMAs and stores alike), so they are
// issued by hand and every wait names exactly the operations younger than the weight group it needs (conv_wino.hip).  Schedule of
// a position (NG = 32 groups of 8 input channels): group k < 8 issues slab DMA k of the next position behind its weight wait;
// group 8 issues the 8 stores of the PREVIOUS position's results (kept aside in 32 registers); every group requests the weights of
// group + 7 first.  Younger than the weights of group i when they are waited for:
constexpr int w4_wait(int i) {
    const int nd = 8;
    const int lo = i - W4_PF > 0 ? i - W4_PF : 0, hi = (nd - 1 < i - 1) ? nd - 1 : i - 1;
    const int dmas = hi >= lo ? hi - lo + 1 : 0;
    const int stores = (i > 8 && i - W4_PF <= 8) ? 8 : 0;      // the stores went out in group 8, behind the request of group 8 + 7
    return W4_PF + dmas + stores;
}
#define W4_ULOAD(dst, voff, base) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#define W4_UWAIT(dst, n) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(dst) : "n"(n) : "memory")
#define W4_STORE(voff, data, base, imm) asm volatile("global_store_dwordx4 %0, %1, %2 offset:" #imm ::"v"(voff), "v"(data), "s"(base) : "memory")

template <int N>
__global__ __launch_bounds__(64 * (W4_C / 32)) void c6_wino4_gemm_kernel(const float *__restrict__ V, const float *__restrict__ U,
                                                                        float *__restrict__ Mo, int nblocks) {
    using G = W4Geom<N>;
    constexpr int C = G::C, CP = G::CP, SLAB = G::SLAB, WAVES = G::WAVES, NG = G::NG, PF = W4_PF;
    constexpr int NGT = W4_NP * NG;                  // weight groups of a whole block
    constexpr int ND = W4_BT / WAVES;                // slab DMAs per wave and position
    static_assert(PF == 7 && NG == 32 && ND == 8, "ring of 8 weight groups, 7 in flight; 8 DMAs and 8 stores per position");
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n4 = wave * 32 + 4 * h;
    int b = blockIdx.x;
    if (b >= nblocks) return;

    const float *ubase = U + (size_t)wave * NGT * 256;
    const unsigned lane16 = (unsigned)lane * 16u;
    // the (tile j, position 0, couts n4 ..) corner of this lane in a block of M, bytes; tile half 1 is 32 tiles further
    const unsigned moff0 = (unsigned)(((size_t)j * G::TILE_STRIDE + n4) * 4), moff1 = moff0 + (unsigned)((size_t)32 * G::TILE_STRIDE * 4);

    auto dma_chunk = [&](const float *src, int buf, int c0) {
        const int c = c0 * WAVES + wave;            // tile c of the block: its row of this position
        const float *gp = src + (size_t)c * G::TILE_STRIDE;
        const int lds_off = (buf * SLAB + c * CP) * 4;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(gp), "s"(lds_off) : "memory");
    };

#pragma unroll
    for (int c0 = 0; c0 < ND; c0++) dma_chunk(V + (size_t)b * G::BLOCK_FLOATS, 0, c0);
    f32x4 ring[8];
#pragma unroll
    for (int q = 0; q < PF; q++) W4_ULOAD(ring[q], lane16 + (unsigned)q * 1024u, ubase);
    int gi = PF;

    // results of the previous position, on their way to memory; the very first "previous position" is this block's position 0
    // itself (zeros now, overwritten by the real results one position later: same wave, same addresses, in order)
    f32x16 s0, s1;
#pragma unroll
    for (int v = 0; v < 16; v++) { s0[v] = 0.f; s1[v] = 0.f; }
    const float *sdst = Mo + (size_t)b * G::BLOCK_FLOATS;

#pragma unroll 1
    for (; b < nblocks; b += gridDim.x) {
#pragma unroll 1
        for (int p = 0; p < W4_NP; p++) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF) : "memory");      // slab p has landed (and the stores of two positions ago)
            __builtin_amdgcn_s_barrier();
            const float *nsrc;
            {
                int nb = p < W4_NP - 1 ? b : b + (int)gridDim.x;
                nb = nb < nblocks ? nb : b;          // no next block: copy this block's slab 0 again (never read)
                nsrc = V + (size_t)nb * G::BLOCK_FLOATS + (size_t)(p < W4_NP - 1 ? p + 1 : 0) * G::POS_STRIDE;
            }
            const float *vb = wsm + (p & 1) * SLAB + j * CP + 4 * h;
            f32x16 m0, m1;
#pragma unroll
            for (int v = 0; v < 16; v++) { m0[v] = 0.f; m1[v] = 0.f; }
            f32x4 va0 = *reinterpret_cast<const f32x4 *>(vb), va1 = *reinterpret_cast<const f32x4 *>(vb + 32 * CP), vn0, vn1;
#define W4_GROUP(q, u, NWAIT, DMA, ST)                                                                           \
            {                                                                                                    \
                W4_ULOAD(ring[((q) + PF) & 7], lane16 + (unsigned)gi * 1024u, ubase);                            \
                gi = gi + 1 == NGT ? 0 : gi + 1;                                                                 \
                vn0 = *reinterpret_cast<const f32x4 *>(vb + 8 * ((u) + 1 < NG ? (u) + 1 : (u)));                 \
                vn1 = *reinterpret_cast<const f32x4 *>(vb + 32 * CP + 8 * ((u) + 1 < NG ? (u) + 1 : (u)));       \
                W4_UWAIT(ring[(q)], NWAIT);                                                                      \
                const f32x4 w = ring[(q)];                                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va0[0], m0, 0, 0, 0);                            \
                m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], va1[0], m1, 0, 0, 0);                            \
                if ((DMA) >= 0 && (DMA) < ND) dma_chunk(nsrc, (p + 1) & 1, (DMA));                               \
                if (ST) {                                                                                        \
                    const f32x4 a0 = {s0[0], s0[1], s0[2], s0[3]}, a1 = {s0[4], s0[5], s0[6], s0[7]};            \
                    const f32x4 a2 = {s0[8], s0[9], s0[10], s0[11]}, a3 = {s0[12], s0[13], s0[14], s0[15]};      \
                    const f32x4 b0 = {s1[0], s1[1], s1[2], s1[3]}, b1 = {s1[4], s1[5], s1[6], s1[7]};            \
                    const f32x4 b2 = {s1[8], s1[9], s1[10], s1[11]}, b3 = {s1[12], s1[13], s1[14], s1[15]};      \
                    W4_STORE(moff0, a0, sdst, 0); W4_STORE(moff0, a1, sdst, 32);                                 \
                    W4_STORE(moff0, a2, sdst, 64); W4_STORE(moff0, a3, sdst, 96);                                \
                    W4_STORE(moff1, b0, sdst, 0); W4_STORE(moff1, b1, sdst, 32);                                 \
                    W4_STORE(moff1, b2, sdst, 64); W4_STORE(moff1, b3, sdst, 96);                                \
                    asm volatile("s_nop 1" ::: "memory");                                                        \
                }                                                                                                \
                _Pragma("unroll") for (int r = 1; r < 4; r++) {                                                  \
                    m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va0[r], m0, 0, 0, 0);                        \
                    m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], va1[r], m1, 0, 0, 0);                        \
                }                                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                va0 = vn0; va1 = vn1;                                                                            \
            }
#define WG_(i) W4_GROUP((i) & 7, (i), w4_wait(i), (i), (i) == 8)
            WG_(0) WG_(1) WG_(2) WG_(3) WG_(4) WG_(5) WG_(6) WG_(7) WG_(8) WG_(9) WG_(10) WG_(11) WG_(12) WG_(13) WG_(14) WG_(15)
#undef WG_
#pragma unroll 1
            for (int u0 = 16; u0 < NG; u0 += 8) {
                W4_GROUP(0, u0, PF, -1, false) W4_GROUP(1, u0 + 1, PF, -1, false) W4_GROUP(2, u0 + 2, PF, -1, false) W4_GROUP(3, u0 + 3, PF, -1, false)
                W4_GROUP(4, u0 + 4, PF, -1, false) W4_GROUP(5, u0 + 5, PF, -1, false) W4_GROUP(6, u0 + 6, PF, -1, false) W4_GROUP(7, u0 + 7, PF, -1, false)
            }
#undef W4_GROUP
            // this position's results aside: stored in group 8 of the next position
            s0 = m0; s1 = m1;
            sdst = Mo + (size_t)b * G::BLOCK_FLOATS + (size_t)p * G::POS_STRIDE;
        }
    }
    {
        const f32x4 a0 = {s0[0], s0[1], s0[2], s0[3]}, a1 = {s0[4], s0[5], s0[6], s0[7]};
        const f32x4 a2 = {s0[8], s0[9], s0[10], s0[11]}, a3 = {s0[12], s0[13], s0[14], s0[15]};
        const f32x4 b0 = {s1[0], s1[1], s1[2], s1[3]}, b1 = {s1[4], s1[5], s1[6], s1[7]};
        const f32x4 b2 = {s1[8], s1[9], s1[10], s1[11]}, b3 = {s1[12], s1[13], s1[14], s1[15]};
        W4_STORE(moff0, a0, sdst, 0); W4_STORE(moff0, a1, sdst, 32); W4_STORE(moff0, a2, sdst, 64); W4_STORE(moff0, a3, sdst, 96);
        W4_STORE(moff1, b0, sdst, 0); W4_STORE(moff1, b1, sdst, 32); W4_STORE(moff1, b2, sdst, 64); W4_STORE(moff1, b3, sdst, 96);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- output transform: A^T M A, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1], + bias (+ residual), ReLU ----
// One wave per tile at a time, lane = 4 channels: the tile's 36 x 1 KiB of M are one contiguous run; the 4x4 pixels leave as
// 1-KiB NHWC runs (pixels beyond the board edge are dropped).
template <int N>
__global__ __launch_bounds__(256) void c6_wino4_out_kernel(const float *__restrict__ M, const float *__restrict__ bias, const float *__restrict__ res,
                                                           float *__restrict__ out, long total_tiles, int relu) {
    using G = W4Geom<N>;
    constexpr int C = G::C, T = G::T, TW = G::TW;
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const f32x4 bi = bias ? *reinterpret_cast<const f32x4 *>(bias + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float floor_ = relu ? 0.f : -INFINITY;
#pragma unroll 1
    for (long t = wave0; t < total_tiles; t += nwaves) {
        const float *mb = M + (size_t)(t / W4_BT) * G::BLOCK_FLOATS + (size_t)(t % W4_BT) * G::TILE_STRIDE + lane * 4;
        const long bd = t / T;
        const int tt = (int)(t - bd * T), ty = tt / TW, tx = tt - ty * TW;
        // tcol[a][jj] = sum_i A^T[a][i] M[i][jj]
        f32x4 tc[4][6];
#pragma unroll
        for (int jj = 0; jj < 6; jj++) {
            f32x4 m[6];
#pragma unroll
            for (int i = 0; i < 6; i++) m[i] = *reinterpret_cast<const f32x4 *>(mb + (size_t)(i * 6 + jj) * C);
            const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
            tc[0][jj] = m[0] + s12 + s34;
            tc[1][jj] = w4_fma(2.f, d34, d12);
            tc[2][jj] = w4_fma(4.f, s34, s12);
            tc[3][jj] = w4_fma(8.f, d34, d12) + m[5];
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int py = 4 * ty + a;
            if (py >= N) continue;
            const f32x4 s12 = tc[a][1] + tc[a][2], d12 = tc[a][1] - tc[a][2], s34 = tc[a][3] + tc[a][4], d34 = tc[a][3] - tc[a][4];
            f32x4 y[4];
            y[0] = tc[a][0] + s12 + s34;
            y[1] = w4_fma(2.f, d34, d12);
            y[2] = w4_fma(4.f, s34, s12);
            y[3] = w4_fma(8.f, d34, d12) + tc[a][5];
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
                const int px = 4 * tx + bb;
                if (px >= N) continue;
                const size_t o = ((size_t)bd * (N * N) + (size_t)(py * N + px)) * C + lane * 4;
                f32x4 v = y[bb] + bi;
                if (res) v = v + *reinterpret_cast<const f32x4 *>(res + o);
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = v[r] > floor_ ? v[r] : floor_;
                *reinterpret_cast<f32x4 *>(out + o) = v;
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
size_t c6_conv_packed_wino4_floats(int C) { return (size_t)W4_NP * C * C; }

// w: folded OIHW [C][C][3][3] -> U = G g G^T (double, rounded once) packed [wave][pos = 6i + j][group u][lane][r]:
//    U_pos[cout = 32 wave + (lane & 31)][cin = 8u + 4(lane >> 5) + r]
void c6_pack_conv_weights_wino4(int C, const float *w, float *packed) {
    static const double Gm[6][3] = {{1.0 / 4, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                    {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int WAVES = C / 32, NG = C / 8;
    for (int wv = 0; wv < WAVES; wv++)
        for (int lane = 0; lane < 64; lane++)
            for (int u = 0; u < NG; u++)
                for (int r = 0; r < 4; r++) {
                    const int co = wv * 32 + (lane & 31), ci = 8 * u + 4 * (lane >> 5) + r;
                    const float *g = w + ((size_t)co * C + ci) * 9;
                    for (int i = 0; i < 6; i++)
                        for (int jj = 0; jj < 6; jj++) {
                            double s = 0.0;
                            for (int a = 0; a < 3; a++)
                                for (int b = 0; b < 3; b++) s += Gm[i][a] * (double)g[a * 3 + b] * Gm[jj][b];
                            packed[((((size_t)wv * W4_NP + (i * 6 + jj)) * NG + u) * 64 + lane) * 4 + r] = (float)s;
                        }
                }
}

// floats of the scratch for `nboards` boards: V and M, one behind the other
static size_t w4_half_floats(int N, int nboards) {
    const long TW = (N + 3) / 4, tiles = (long)nboards * TW * TW, nblocks = (tiles + W4_BT - 1) / W4_BT;
    return (size_t)nblocks * W4_BT * W4_NP * W4_C;
}
size_t c6_conv_wino4_vbuf_floats(int C, int N, int nboards) { return C == W4_C ? 2 * w4_half_floats(N, nboards) : 0; }

template <int N>
static void launch_w4(int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out, float *vbuf, int num_cus,
                      int relu, int steps, hipStream_t s) {
    using G = W4Geom<N>;
    const long tiles = (long)nboards * G::T;
    const int nblocks = (int)((tiles + W4_BT - 1) / W4_BT);
    float *Vb = vbuf, *Mb = vbuf + w4_half_floats(N, nboards);
    if (steps & 1) {
        constexpr size_t in_smem = (size_t)6 * (4 * G::TW + 2) * W4_C * 4;
        const int grid_in = nboards < num_cus ? nboards : num_cus;
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_wino4_in_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)in_smem);
        hipLaunchKernelGGL((c6_wino4_in_kernel<N>), dim3(grid_in), dim3(W4_IN_THREADS), in_smem, s, in, Vb, nboards);
        if (tiles % W4_BT) hipLaunchKernelGGL((c6_wino4_tail_kernel<N>), dim3(256), dim3(256), 0, s, Vb, tiles);
    }
    if (steps & 2) {
        const size_t smem = G::LDS_BYTES;
        const int grid = nblocks < num_cus ? nblocks : num_cus;
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_wino4_gemm_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((c6_wino4_gemm_kernel<N>), dim3(grid), dim3(G::THREADS), smem, s, Vb, wp, Mb, nblocks);
    }
    if (steps & 4) {
        const long want = (tiles + 3) / 4;
        const int grid_out = (int)(want < (long)num_cus * 8 ? want : (long)num_cus * 8);
        hipLaunchKernelGGL((c6_wino4_out_kernel<N>), dim3(grid_out), dim3(256), 0, s, Mb, bias, res, out, tiles, relu);
    }
}

// steps: bit 0 input transform, bit 1 GEMMs, bit 2 output transform (7 = the layer)
bool c6_conv_wino4_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out,
                            float *vbuf, int num_cus, hipStream_t s, int relu, int steps) {
    if (C != W4_C) return false;
    if (nboards <= 0) return true;
    if (N == 19) { launch_w4<19>(nboards, in, wp, bias, res, out, vbuf, num_cus, relu, steps, s); return true; }
    if (N == 15) { launch_w4<15>(nboards, in, wp, bias, res, out, vbuf, num_cus, relu, steps, s); return true; }
    if (N == 9) { launch_w4<9>(nboards, in, wp, bias, res, out, vbuf, num_cus, relu, steps, s); return true; }
    return false;
}
// matrix FLOPs the GEMM kernel executes for `nboards` boards (whole 64-tile blocks x 36 positions x C x C MACs)
double c6_conv_wino4_exec_flops(int C, int N, int nboards) {
    const long TW = (N + 3) / 4, tiles = (long)nboards * TW * TW, nblocks = (tiles + W4_BT - 1) / W4_BT;
    return 2.0 * W4_NP * (double)nblocks * W4_BT * (double)C * C;
}
