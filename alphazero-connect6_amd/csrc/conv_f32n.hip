// conv_f32n.hip -- exact-FP32 3x3 convolution, N-split persistent schedule (c6_conv3x3_f32n_kernel).
//
// The schedule of the split-bf16 kernels (conv_bf16x3.hip) on v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate):
//  * one 512-thread workgroup per CU walks over boards; wave w owns output channels [32w, 32w+32) for EVERY pixel
//    (12 tiles of 32x32 at N = 19: 192 accumulator VGPRs);
//  * its weights are private: 16 floats per lane and (slice, tap) stream L2 -> VGPR one step ahead -- no weight stages in
//    LDS, one barrier per 32-channel slice instead of one per tap (c6_conv3x3_kernel: 72 per board);
//  * LDS holds two zero-haloed FP32 input buffers on the (N+1)-wide row grid (slot = 36 floats: 32 channels + 4 pad, the
//    conflict-free stride for ds_read_b128 with one pixel per lane); the next slice leaves HBM by LDS-DMA into a
//    wave-private landing area and is moved to its slots two sub-stages later (asm DMA: see conv_bf16x3.hip);
//  * weights are the MFMA A operand (D = W x X^T): a lane's accumulators are 16-byte NHWC runs, the residual is the
//    accumulators' initial value, the epilogue is bias + ReLU + 16-byte stores.
// MFMA operand mapping: one ds_read_b128 gives lane (pixel j, h) channels 8u + 4h .. + 3 of an 8-channel group u; MFMA r
// of the group contracts the channel pair (8u + r, 8u + 4 + r), and weight lane (cout i, h) holds W[i][8u + 4h + r].
#include "c6_device.h"
#include "c6_engine.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define F3N_SLOT 36     // floats per LDS pixel slot

template <int C, int N>
struct F3NGeom {
    static constexpr int W = N + 1;
    static constexpr int MROWS = N * W;
    static constexpr int MT = (MROWS + 31) / 32;
    static constexpr int SLOTS = ((MT * 32 + 2 * W + 2 + 1) + 7) & ~7;
    static constexpr int IN_FLOATS = SLOTS * F3N_SLOT;
    static constexpr int WAVES = C / 32;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int NITEM = (N * N * 8 + THREADS - 1) / THREADS;     // 16-byte staging items per thread and slice
    static constexpr int IPS = (NITEM + 5) / 6;
    static constexpr size_t LDS_BYTES = (size_t)2 * IN_FLOATS * 4 + (size_t)THREADS * IPS * 16;
};

template <int C, int N>
__global__ __launch_bounds__(64 * (C / 32)) void c6_conv3x3_f32n_kernel(const float *__restrict__ in, const float *__restrict__ wp,
                                                                        const float *__restrict__ bias, const float *__restrict__ res,
                                                                        float *__restrict__ out, int nboards, int relu) {
    using G = F3NGeom<C, N>;
    constexpr int W = G::W, MT = G::MT, KS = C / 32, THREADS = G::THREADS, WAVES = G::WAVES, CELLS = N * N, NITEM = G::NITEM, IPS = G::IPS;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float *lds0 = fsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *stg = fsm + 2 * G::IN_FLOATS + (size_t)wave * IPS * 256;
    const int b0 = blockIdx.x, bstep = gridDim.x;
    const int nlocal = b0 < nboards ? (nboards - b0 + bstep - 1) / bstep : 0;
    const int total = nlocal * KS;

    for (int k = tid; k < 2 * G::IN_FLOATS; k += THREADS) lds0[k] = 0.f;   // both buffers, halo = 0

    auto slot_of = [&](int item) -> int {
        const int p = item >> 3, q = item & 7;
        return ((p / N + 1) * W + (p % N + 1)) * F3N_SLOT + q * 4;
    };
    // weights of this wave: [stage = ks*9 + tap][wave][u][lane][4]
    auto wptr = [&](int stage, int u) -> const f32x4 * {
        return reinterpret_cast<const f32x4 *>(wp + (((size_t)stage * WAVES + wave) * 4 + u) * 256) + lane;
    };

    if (total == 0) return;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < NITEM; k++) {      // very first slice, synchronously
        const int item = tid + k * THREADS;
        if (item < CELLS * 8)
            *reinterpret_cast<f32x4 *>(lds0 + slot_of(item)) =
                *reinterpret_cast<const f32x4 *>(in + (size_t)b0 * (size_t)(CELLS * C) + (size_t)(item >> 3) * C + (item & 7) * 4);
    }
    f32x4 wc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) wc[u] = *wptr(0, u);

#pragma unroll 1
    for (int lb = 0; lb < nlocal; lb++) {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));      // opaque per board: keeps the row offsets out of (spilled) registers across the MFMA loop
        const int pi = lane_o & 31, n4 = wave * 32 + 4 * (lane_o >> 5);
        auto row_off = [&](int mt, bool &ok) -> size_t {
            const int m = mt * 32 + pi, y = m / W, x = m % W;
            ok = x < N && y < N;
            return (size_t)((ok ? y : 0) * N + (ok ? x : 0)) * C + n4;
        };
        const size_t b = (size_t)(b0 + lb * bstep);
        f32x16 acc[MT];
        if (res) {
            const float *resb = res + b * (size_t)CELLS * C;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                bool ok;
                const float *rp = resb + row_off(mt, ok);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const f32x4 t = *reinterpret_cast<const f32x4 *>(rp + 8 * g);
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[mt][4 * g + r] = t[r];
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int v = 0; v < 16; v++) acc[mt][v] = 0.f;
        }
#pragma unroll 1
        for (int ks = 0; ks < KS; ks++) {
            const int gs = lb * KS + ks;
            __syncthreads();
            const float *lds_in = lds0 + (gs & 1) * G::IN_FLOATS;
            float *lds_nx = lds0 + ((gs + 1) & 1) * G::IN_FLOATS;
            const bool more = gs + 1 < total;
            const float *nslice = in + (size_t)(b0 + (ks + 1 < KS ? lb : lb + 1) * bstep) * (size_t)(CELLS * C) + (ks + 1 < KS ? (ks + 1) * 32 : 0);
            int abase = (lane & 31) * F3N_SLOT + 4 * (lane >> 5);     // tap 0 = top-left neighbour of row m = j; channels 4h.. of a group
#pragma unroll 1
            for (int tap = 0; tap < 9; tap++) {
                f32x4 wn[4];
                {
                    int ntap = tap + 1, nks = ks;
                    if (ntap == 9) { ntap = 0; nks++; }
                    if (nks == KS) nks = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) wn[u] = *wptr(nks * 9 + ntap, u);
                }
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    // trickle: item group k leaves HBM at sub-stage 3k (LDS-DMA, no VGPRs) and moves to its slots at 3k + 2;
                    // exactly the four weight loads of one step are younger than the DMA by then (vmcnt completes in order)
                    if (more) {
                        const int sidx = tap * 2 + half, k = sidx / 3, ph = sidx % 3;
                        if (ph == 0) {
#pragma unroll
                            for (int q = 0; q < IPS; q++)
                                if (k * IPS + q < NITEM) {
                                    int item = tid + (k * IPS + q) * THREADS;
                                    item = item < CELLS * 8 ? item : CELLS * 8 - 1;
                                    const float *gp = nslice + (size_t)(item >> 3) * C + (item & 7) * 4;
                                    const int lds_off = __builtin_amdgcn_readfirstlane((int)(2 * G::IN_FLOATS * 4 + (wave * IPS + q) * 1024));
                                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lds_off) : "memory", "m0");
                                }
                        } else if (ph == 2) {
                            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
                            for (int q = 0; q < IPS; q++) {
                                const int item = tid + (k * IPS + q) * THREADS;
                                if (k * IPS + q < NITEM && item < CELLS * 8)
                                    *reinterpret_cast<f32x4 *>(lds_nx + slot_of(item)) = *reinterpret_cast<const f32x4 *>(stg + q * 256 + lane * 4);
                            }
                        }
                    }
                    // two 8-channel groups per half; pixel fragments double-buffered by hand
#pragma unroll
                    for (int uu = 0; uu < 2; uu++) {
                        const int u = half * 2 + uu;
                        f32x4 px[2];
                        auto load_p = [&](int slot, int mt) { px[slot] = *reinterpret_cast<const f32x4 *>(lds_in + abase + (mt * 32) * F3N_SLOT + u * 8); };
                        load_p(0, 0);
#pragma unroll
                        for (int mt = 0, sl = 0; mt < MT; mt++, sl ^= 1) {
                            if (mt + 1 < MT) load_p(sl ^ 1, mt + 1);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int r = 0; r < 4; r++) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[u][r], px[sl][r], acc[mt], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) wc[u] = wn[u];
                abase += (tap % 3 == 2 ? W - 2 : 1) * F3N_SLOT;
            }
        }   // ks

        // epilogue: acc[mt][4g + r] = channel 32 wave + 8g + 4h + r of pixel row m = 32 mt + i
        float *outb = out + b * (size_t)CELLS * C;
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 bi[4];
#pragma unroll
        for (int g = 0; g < 4; g++) bi[g] = bias ? *reinterpret_cast<const f32x4 *>(bias + n4 + 8 * g) : zero4;
        const float floor_ = relu ? 0.f : -INFINITY;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            bool ok;
            float *op = outb + row_off(mt, ok);
            if (ok) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 yv;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float t = acc[mt][4 * g + r] + bi[g][r];
                        yv[r] = t > floor_ ? t : floor_;
                    }
                    *reinterpret_cast<f32x4 *>(op + 8 * g) = yv;
                }
            }
        }
    }
}

size_t c6_conv_packed_f32n_floats(int C) { return (size_t)9 * C * C; }

// w: folded OIHW [C][C][3][3] -> [stage = ks*9 + tap][wave][u][lane][r]:  W[n = wave*32 + (lane & 31)][k = ks*32 + 8u + 4(lane >> 5) + r][tap]
void c6_pack_conv_weights_f32n(int C, const float *w, float *packed) {
    const int KS = C / 32, WAVES = C / 32;
    for (int ks = 0; ks < KS; ks++)
        for (int t = 0; t < 9; t++)
            for (int wv = 0; wv < WAVES; wv++)
                for (int u = 0; u < 4; u++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int r = 0; r < 4; r++) {
                            const int n = wv * 32 + (lane & 31), k = ks * 32 + 8 * u + 4 * (lane >> 5) + r;
                            packed[((((size_t)(ks * 9 + t) * WAVES + wv) * 4 + u) * 64 + lane) * 4 + r] = w[((size_t)n * C + k) * 9 + t];
                        }
}

template <int C, int N>
static void launch_f32n(int nboards, int grid, const float *in, const float *wp, const float *bias, const float *res, float *out, int relu, hipStream_t s) {
    using G = F3NGeom<C, N>;
    const size_t smem = G::LDS_BYTES;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_f32n_kernel<C, N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((c6_conv3x3_f32n_kernel<C, N>), dim3(grid), dim3(G::THREADS), smem, s, in, wp, bias, res, out, nboards, relu);
}

bool c6_conv_f32n_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out,
                           int num_cus, hipStream_t s, int relu) {
    if (nboards <= 0) return true;
    const int per_cu = C == 256 ? 1 : (C == 128 ? 2 : 4);
    const int grid = nboards < num_cus * per_cu ? nboards : num_cus * per_cu;
#define F3NCASE(CC, NN) if (C == CC && N == NN) { launch_f32n<CC, NN>(nboards, grid, in, wp, bias, res, out, relu, s); return true; }
    F3NCASE(256, 19) F3NCASE(256, 15) F3NCASE(256, 9)
    F3NCASE(128, 19) F3NCASE(128, 15) F3NCASE(128, 9)
    F3NCASE(64, 19) F3NCASE(64, 15) F3NCASE(64, 9)
#undef F3NCASE
    return false;
}
