// c6_wino.h -- device helpers shared by the FP32 Winograd kernels (conv_wino4.hip: F(4x4) / F(3) tilings, all board sizes and the trainer;
// conv_wino6.hip: the evaluator's 19x19 tiling with F(6) across): vector types, the 1-D transform rows, the stem computed in place
// (model.hpp:56-59), wave sums, and the hand-issued memory operations of the GEMM kernels.
#pragma once
#include "c6_device.h"
#include "c6_engine.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 w4_fma(float a, f32x4 x, f32x4 y) { return __builtin_elementwise_fma(f32x4{a, a, a, a}, x, y); }
__device__ __forceinline__ f32x2 w4_fma(float a, f32x2 x, f32x2 y) { return __builtin_elementwise_fma(f32x2{a, a}, x, y); }

// row i of B^T applied along the first index of x[6]
#define W4_BT_ROW(i, x0, x1, x2, x3, x4, x5)                                                          \
    ((i) == 0 ? w4_fma(4.f, x0, w4_fma(-5.f, x2, x4))                                                 \
   : (i) == 1 ? w4_fma(-4.f, (x1) + (x2), (x3) + (x4))                                               \
   : (i) == 2 ? w4_fma(4.f, (x1) - (x2), (x4) - (x3))                                                \
   : (i) == 3 ? w4_fma(2.f, (x3) - (x1), (x4) - (x2))                                                \
   : (i) == 4 ? w4_fma(2.f, (x1) - (x3), (x4) - (x2))                                                \
              : w4_fma(4.f, x1, w4_fma(-5.f, x3, x5)))
// row i of F(3,3)'s B^T = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1] applied along the first index of x[5]
#define W4_BT3_ROW(i, x0, x1, x2, x3, x4)                                                             \
    ((i) == 0 ? w4_fma(2.f, (x0) - (x2), (x3) - (x1))                                                 \
   : (i) == 1 ? w4_fma(-2.f, x1, (x3) - (x2))                                                         \
   : (i) == 2 ? w4_fma(2.f, x1, w4_fma(-3.f, x2, x3))                                                 \
   : (i) == 3 ? (x3) - (x1)                                                                           \
              : w4_fma(2.f, (x1) - (x3), (x4) - (x2)))

// ---- the stem as an input ---------------------------------------------------------------------------------------------
// conv 2 -> C (+ folded BN, ReLU) on the planes [player broadcast, board image] (model.hpp:56-59, c6_stem_kernel) at pixel (y, x)
// for 4 channels: 9 multiply-adds per channel for the board plane + one for the player plane (below).  bd = the board image as floats on
// the (N+2)-wide zero-haloed grid, w1(t) = the lane's 4 board-plane weights of tap t, pl_term = pv * S0[border class].  Cheap enough to be COMPUTED wherever the stem's activations are needed
// (the first layer's input transform, the residual of the first block) instead of being written and read back: 36 GB per forward pass.
template <int N, typename WT>
__device__ __forceinline__ f32x4 w4_stem_pixel(const float *bd, f32x4 pl_term, int y, int x, f32x4 bias, WT w1) {
    constexpr int P = N + 2;
    f32x4 acc = bias + pl_term;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const float bdv = bd[(y + t / 3) * P + x + t % 3];      // the zero-haloed grid: tap (t / 3 - 1, t % 3 - 1)
        const f32x4 w = w1(t);
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = __builtin_fmaf(w[r], bdv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = acc[r] > 0.f ? acc[r] : 0.f;
    return acc;
}
// The player plane is a constant inside the board and zero outside, so its nine taps collapse to  pv * (sum of the plane-0 weights of
// the taps that lie on the board)  -- one of 9 sums per channel, by the pixel's border class 3 * (top, middle, bottom) + (left, middle,
// right): this thread's 4 channels of the table S0[9][C]
__device__ __forceinline__ int w4_border_class(int y, int x, int N) { return 3 * (y == 0 ? 0 : y == N - 1 ? 2 : 1) + (x == 0 ? 0 : x == N - 1 ? 2 : 1); }
__device__ __forceinline__ void w4_stem_player_sums(const float *w /* [9][2][C] */, int C, int c4, float *s0 /* [9][C] */) {
#pragma unroll
    for (int cls = 0; cls < 9; cls++) {
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int dy = t / 3 - 1, dx = t % 3 - 1, ry = cls / 3, rx = cls % 3;
            const bool in = !(ry == 0 && dy < 0) && !(ry == 2 && dy > 0) && !(rx == 0 && dx < 0) && !(rx == 2 && dx > 0);
            if (in) sum = sum + *reinterpret_cast<const f32x4 *>(w + (t * 2 + 0) * C + c4);
        }
        *reinterpret_cast<f32x4 *>(s0 + cls * C + c4) = sum;
    }
}
// board image k of leaf b / images as floats on the zero-haloed grid (augment_way8, augment.rs:60-73: the stem's gather)
template <int N>
__device__ __forceinline__ float w4_stem_plane(const C6StemSrc &S, long b, float *bd, int tid, int nthreads) {
    constexpr int P = N + 2, cells = N * N;
    const long leaf = b / S.images;
    const int k = S.images == 8 ? (int)(b % 8) : 6;      // image 6 is the identity
    const int8_t *board = S.boards + leaf * (size_t)S.board_stride;
    for (int c = tid; c < cells; c += nthreads) bd[(c / N + 1) * P + c % N + 1] = (float)board[S.aug_map[k * cells + c]];
    return (float)S.player[leaf * (size_t)S.player_stride];
}


#define W4_ULOAD(dst, voff, base) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#define W4_UWAIT(dst, n) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(dst) : "n"(n) : "memory")
// The store reads its four data registers a few cycles AFTER it issues, and nothing interlocks a vector instruction that overwrites them meanwhile (the
// compiler knows the hazard for its own stores, not for inline asm): the two wait states are part of the SAME asm statement, so that nothing can be scheduled
// between the store and them.  (r5: as a statement of their own -- `asm volatile("s_nop 1")` behind the store -- they let the compiler put the NEXT store's data
// computation, into the same registers, in between: the four-wave jobs' third row sum came out wrong in 8 of its 16 registers.)
#define W4_STORE(voff, data, base, imm) asm volatile("global_store_dwordx4 %0, %1, %2 offset:" #imm "\n\ts_nop 1" ::"v"(voff), "v"(data), "s"(base) : "memory")

// sum over the wave, valid in lanes 48 .. 63: four butterfly steps inside the rows of 16 lanes, then row 0 -> 1 and 2 -> 3 (row_bcast:15) and
// rows 0 + 1 -> 3 (row_bcast:31) -- six DPP additions
__device__ __forceinline__ float w4_wave_sum_row3(float v) {
    v += c6_dpp_f32<C6_DPP_XOR1>(v);
    v += c6_dpp_f32<C6_DPP_XOR2>(v);
    v += c6_dpp_f32<C6_DPP_HALF_MIRROR>(v);
    v += c6_dpp_f32<C6_DPP_MIRROR>(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return v;
}

// the dynamic-LDS limit of a kernel, raised once per (kernel, device) instead of at every launch
#include <mutex>
#include <set>
#include <utility>
static inline void w4_smem_once(const void *fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert({fn, dev}).second) c6_note_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) of a Winograd kernel");
}
