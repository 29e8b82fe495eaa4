// conv_bf16x3.hip -- K4, "split" variant of the 3x3 C->C convolution: FP32 operands are split into two bf16 numbers
// (x = hi + mid + O(2^-17 x)) and every product is evaluated as  x_hi*w_hi + x_hi*w_mid + x_mid*w_hi  on the bf16 matrix
// cores (v_mfma_f32_32x32x16_bf16) with FP32 accumulation.  bf16 x bf16 products are exact in FP32, so the only error is
// the dropped cross terms: <= 3 * 2^-16 |x w| per product (tests hold the whole net to the same 1e-5 as the FP32-MFMA
// kernel; measured 2e-7 on the value, 4e-9 on the policy).  Three bf16 MFMAs replace one FP32 MFMA at 16x the rate.
//
// Arithmetic: AlphaZero/cpp_weighted/model.hpp:64-74 (conv3x3 pad 1 + folded BatchNorm + residual + ReLU), same NHWC FP32
// activations in HBM as c6_conv3x3_kernel; the split happens while a 32-channel slice is staged into LDS.
// Structure: one 512-thread workgroup per board, 8 waves = 4 (M) x 2 (N), 3 x 4 tiles of 32x32 per wave, zero-haloed
// (N+2)^2 grid in LDS (taps = constant offsets), weight stages (pre-split on the host) streamed by LDS-DMA one stage
// ahead, LDS-transposed epilogue with 16-byte residual loads / output stores.
#include "c6_device.h"
#include "c6_engine.h"

#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define B3_THREADS 512
#define B3_SLOT 72          // ushort (2-byte) units per LDS pixel slot: 32 hi + 32 mid + 8 pad  (144 bytes)

// ABL: ablation switch used only by c6_bench_conv (0 = the product kernel): 1 no weight DMA, 2 no input staging,
//      3 neither and no barriers, 5 no epilogue stores (accumulators kept alive)
template <int C, int MTW, int ABL = 0>
__global__ __launch_bounds__(B3_THREADS) void c6_conv3x3_bf16x3_kernel(const float *__restrict__ in, const unsigned short *__restrict__ wp,
                                                                       const float *__restrict__ bias, const float *__restrict__ res,
                                                                       float *__restrict__ out, int N) {
    constexpr int NTW = C / 64;
    constexpr int KS = C / 32;
    constexpr int WSTAGE = 2 * 2 * 2 * C * 8;     // ushorts per weight stage: [plane][g][h][n][8]  (32 KiB at C = 256)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    const int cells = N * N, P = N + 2;
    const int in_units = P * P * B3_SLOT;
    unsigned short *lds_in = smem16;
    unsigned short *lds_w = smem16 + ((in_units + 7) & ~7);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const size_t b = blockIdx.x;
    const float *inb = in + b * (size_t)cells * C;

    int aoff[MTW];   // ushort units
#pragma unroll
    for (int mt = 0; mt < MTW; mt++) {
        int p = (wm * MTW + mt) * 32 + i;
        if (p >= cells) p = 0;
        aoff[mt] = ((p / N + 1) * P + (p % N + 1)) * B3_SLOT + 8 * h;
    }

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[mt][nt][v] = 0.f;

    for (int k = tid; k < in_units / 2; k += B3_THREADS) reinterpret_cast<uint32_t *>(lds_in)[k] = 0u;   // zero halo

    constexpr int WPIECES = WSTAGE * 2 / 1024;     // 1-KiB LDS-DMA pieces per stage
    auto dma_w = [&](int stage, int buf) {
        const unsigned short *src = wp + (size_t)stage * WSTAGE;
        unsigned short *dst = lds_w + buf * WSTAGE;
        for (int pc = wave; pc < WPIECES; pc += B3_THREADS / 64)
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)(src + pc * 512 + lane * 8),
                                             (__attribute__((address_space(3))) void *)(dst + pc * 512), 16, 0, 0);
    };

    if (ABL != 1 && ABL != 3) dma_w(0, 0);
    for (int ks = 0; ks < KS; ks++) {
        if (ABL != 3) __syncthreads();
        // stage + split the 32-channel input slice: x -> hi = bf16(x), mid = bf16(x - hi).  The loads of a thread go out in
        // batches of SB before the first conversion of the batch: ceil(6 / SB) HBM round trips per slice instead of 6.
        if (ABL != 2 && ABL != 3) {
            constexpr int SB = 3;
#pragma unroll 1
            for (int k0 = 0; k0 < 6; k0 += SB) {
                f32x4 v[SB];
#pragma unroll
                for (int k = 0; k < SB; k++) {
                    // unconditional loads (clamped): a branch around each load would make hipcc wait vmcnt(0) per element
                    int item = tid + (k0 + k) * B3_THREADS;
                    item = item < cells * 8 ? item : cells * 8 - 1;
                    v[k] = *reinterpret_cast<const f32x4 *>(inb + (size_t)(item >> 3) * C + ks * 32 + (item & 7) * 4);
                }
#pragma unroll
                for (int k = 0; k < SB; k++) {
                    const int item = tid + (k0 + k) * B3_THREADS;
                    if (item < cells * 8) {
                        const int p = item >> 3, q = item & 7;
                        bf16x4 hi, mid;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            hi[r] = (__bf16)v[k][r];
                            mid[r] = (__bf16)(v[k][r] - (float)hi[r]);
                        }
                        unsigned short *dst = lds_in + ((p / N + 1) * P + (p % N + 1)) * B3_SLOT + q * 4;
                        *reinterpret_cast<bf16x4 *>(dst) = hi;
                        *reinterpret_cast<bf16x4 *>(dst + 32) = mid;
                    }
                }
            }
        }
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
            const int stage = ks * 9 + tap, buf = stage & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ABL != 3) __syncthreads();
            if (ABL != 1 && ABL != 3) if (stage + 1 < KS * 9) dma_w(stage + 1, buf ^ 1);
            const int toff = ((tap / 3 - 1) * P + (tap % 3 - 1)) * B3_SLOT;
            const unsigned short *wb = lds_w + buf * WSTAGE;
#pragma unroll
            for (int g = 0; g < 2; g++) {       // two 16-deep k steps per 32-channel slice
                bf16x8 ahi[MTW], amid[MTW];
#pragma unroll
                for (int mt = 0; mt < MTW; mt++) {
                    const unsigned short *ap = lds_in + aoff[mt] + toff + g * 16;
                    ahi[mt] = *reinterpret_cast<const bf16x8 *>(ap);
                    amid[mt] = *reinterpret_cast<const bf16x8 *>(ap + 32);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) {
                    const int n = (wn * NTW + nt) * 32 + i;
                    const bf16x8 bhi = *reinterpret_cast<const bf16x8 *>(wb + (((0 * 2 + g) * 2 + h) * C + n) * 8);
                    const bf16x8 bmid = *reinterpret_cast<const bf16x8 *>(wb + (((1 * 2 + g) * 2 + h) * C + n) * 8);
#pragma unroll
                    for (int mt = 0; mt < MTW; mt++) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid[mt], bhi, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[mt], bmid, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[mt], bhi, acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    if (ABL == 5) {   // keep the accumulators alive without the store traffic
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < MTW; mt++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++)
#pragma unroll
                for (int v = 0; v < 16; v++) sum += acc[mt][nt][v];
        if (sum == 123.456f) out[tid] = sum;
        return;
    }
    // epilogue: transpose each 32x32 tile through a wave-private LDS patch -> 16-byte residual loads / stores
    __syncthreads();
    constexpr int PSTR = 36;
    float *patch = reinterpret_cast<float *>(smem16) + wave * (32 * PSTR);
    float *outb = out + b * (size_t)cells * C;
    const float *resb = res ? res + b * (size_t)cells * C : nullptr;
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const int n4 = (wn * NTW + nt) * 32 + c4;
        const f32x4 bi = *reinterpret_cast<const f32x4 *>(bias + n4);
#pragma unroll
        for (int mt = 0; mt < MTW; mt++) {
#pragma unroll
            for (int v = 0; v < 16; v++) patch[((v & 3) + 8 * (v >> 2) + 4 * h) * PSTR + i] = acc[mt][nt][v];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = j * 8 + r8;
                const int p = (wm * MTW + mt) * 32 + row;
                f32x4 y = *reinterpret_cast<const f32x4 *>(patch + row * PSTR + c4);
                if (p < cells) {
                    y = y + bi;
                    if (resb) y = y + *reinterpret_cast<const f32x4 *>(resb + (size_t)p * C + n4);
#pragma unroll
                    for (int q = 0; q < 4; q++) y[q] = y[q] > 0.f ? y[q] : 0.f;
                    *reinterpret_cast<f32x4 *>(outb + (size_t)p * C + n4) = y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

static inline unsigned short f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf16_to_f32(unsigned short h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

size_t c6_conv_packed_b3_units(int C) { return (size_t)9 * (C / 32) * (2 * 2 * 2 * C * 8); }

// w: folded OIHW [C][C][3][3] -> [stage = ks*9 + tap][plane][g][h][n][j], k = ks*32 + g*16 + h*8 + j, bf16 bits
void c6_pack_conv_weights_b3(int C, const float *w, unsigned short *packed) {
    const int KS = C / 32;
    const size_t WSTAGE = (size_t)2 * 2 * 2 * C * 8;
    for (int ks = 0; ks < KS; ks++)
        for (int t = 0; t < 9; t++)
            for (int g = 0; g < 2; g++)
                for (int h = 0; h < 2; h++)
                    for (int n = 0; n < C; n++)
                        for (int j = 0; j < 8; j++) {
                            const int k = ks * 32 + g * 16 + h * 8 + j;
                            const float x = w[((size_t)n * C + k) * 9 + t];
                            const unsigned short hi = f32_to_bf16_rne(x);
                            const unsigned short mid = f32_to_bf16_rne(x - bf16_to_f32(hi));
                            const size_t base = (size_t)(ks * 9 + t) * WSTAGE;
                            packed[base + ((((size_t)0 * 2 + g) * 2 + h) * C + n) * 8 + j] = hi;
                            packed[base + ((((size_t)1 * 2 + g) * 2 + h) * C + n) * 8 + j] = mid;
                        }
}

template <int C>
static void launch_b3(int MTW, int nboards, size_t smem, const float *in, const unsigned short *wp, const float *bias, const float *res,
                      float *out, int N, hipStream_t s) {
#define B3CASE(M)                                                                                                                          \
    case M:                                                                                                                                \
        hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_bf16x3_kernel<C, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((c6_conv3x3_bf16x3_kernel<C, M>), dim3(nboards), dim3(B3_THREADS), smem, s, in, wp, bias, res, out, N);          \
        break;
    switch (MTW) { B3CASE(1) B3CASE(2) default: B3CASE(3) }
#undef B3CASE
}

void c6_conv_b3_ablate(int ablate, int nboards, const float *in, const unsigned short *wp, const float *bias, const float *res,
                       float *out, hipStream_t s) {
    const int N = 19, P = N + 2, C = 256;
    const size_t smem = ((size_t)((P * P * B3_SLOT + 7) & ~7) + 2 * (size_t)(2 * 2 * 2 * C * 8)) * sizeof(unsigned short);
#define B3ABL(A) case A: hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_conv3x3_bf16x3_kernel<256, 3, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((c6_conv3x3_bf16x3_kernel<256, 3, A>), dim3(nboards), dim3(B3_THREADS), smem, s, in, wp, bias, res, out, N); break;
    switch (ablate) { B3ABL(1) B3ABL(2) B3ABL(3) B3ABL(5) default: break; }
#undef B3ABL
}

bool c6_conv_b3_dispatch(int C, int N, int nboards, const float *in, const unsigned short *wp, const float *bias, const float *res,
                         float *out, hipStream_t s) {
    const int cells = N * N, MT = (cells + 31) / 32, MTW = (MT + 3) / 4, P = N + 2;
    size_t smem = ((size_t)((P * P * B3_SLOT + 7) & ~7) + 2 * (size_t)(2 * 2 * 2 * C * 8)) * sizeof(unsigned short);
    if (smem < (size_t)8 * 32 * 36 * sizeof(float)) smem = (size_t)8 * 32 * 36 * sizeof(float);
    switch (C) {
    case 64: launch_b3<64>(MTW, nboards, smem, in, wp, bias, res, out, N, s); return true;
    case 128: launch_b3<128>(MTW, nboards, smem, in, wp, bias, res, out, N, s); return true;
    case 256: launch_b3<256>(MTW, nboards, smem, in, wp, bias, res, out, N, s); return true;
    default: return false;
    }
}
