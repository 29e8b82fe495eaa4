// Layout probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, E8M0 block scales) and v_mfma_f32_16x16x32_f16:
// (found with mfma_fp8_discover.hip / mfma_fp8_scale.hip)  A[i][k]: lane = i + 16 * g, byte t with k = 16 g + (t & 15) + 64 (t >> 4);
// B[k][j] likewise with lane = j + 16 g;  D[i][j]: lane = j + 16 * (i / 4), register i % 4;  the E8M0 scale of (row, K block
// kb = k / 32) is read from lane row + 16 * kb.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k8(const uint8_t *A, const uint8_t *B, const uint8_t *sa, const uint8_t *sb, float *D) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int v = 0; v < 8; v++) {
        uint32_t wa = 0, wb = 0;
        for (int q = 0; q < 4; q++) {
            const int t = 4 * v + q, k = 16 * (l >> 4) + (t & 15) + 64 * (t >> 4);      // K index of byte t of lane group l >> 4
            wa |= (uint32_t)A[(l & 15) * 128 + k] << (8 * q);                             // A row-major [16][128]
            wb |= (uint32_t)B[k * 16 + (l & 15)] << (8 * q);                              // B [128][16]
        }
        a[v] = (int)wa; b[v] = (int)wb;
    }
    // the scale of (row, K block kb = k / 32) is taken from lane row + 16 * kb
    const int scale_a = sa[(l & 15) * 4 + (l >> 4)], scale_b = sb[(l & 15) * 4 + (l >> 4)];
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 4; r++) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
__global__ void k16(const _Float16 *A, const _Float16 *B, float *D) {      // A [16][32], B [32][16]
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int q = 0; q < 8; q++) { a[q] = A[(l & 15) * 32 + 8 * (l >> 4) + q]; b[q] = B[(8 * (l >> 4) + q) * 16 + (l & 15)]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
static float e4m3(uint8_t v) {      // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(8.0f + m, e - 10);
    return s ? -f : f;
}
int main() {
    srand(1);
    std::vector<uint8_t> A(16 * 128), B(128 * 16), sa(64), sb(64);
    for (auto &x : A) { x = rand() & 0xff; if ((x & 0x7f) == 0x7f) x &= 0xfe; }      // no NaN
    for (auto &x : B) { x = rand() & 0xff; if ((x & 0x7f) == 0x7f) x &= 0xfe; }
    for (auto &x : sa) x = 120 + rand() % 10;
    for (auto &x : sb) x = 110 + rand() % 20;
    uint8_t *dA, *dB, *dsa, *dsb; float *dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k8, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double worst = 0, mag = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double s = 0;
            for (int k = 0; k < 128; k++)
                s += (double)e4m3(A[i * 128 + k]) * ldexp(1.0, sa[i * 4 + k / 32] - 127) * (double)e4m3(B[k * 16 + j]) * ldexp(1.0, sb[j * 4 + k / 32] - 127);
            worst = fmax(worst, fabs(s - D[i * 16 + j])); mag = fmax(mag, fabs(s));
        }
    printf("fp8 16x16x128 scaled: max |ref - gpu| = %.3e (max |ref| %.3e) -> %s\n", worst, mag, worst <= 1e-5 * mag ? "layout confirmed" : "MISMATCH");
    std::vector<_Float16> A16(16 * 32), B16(32 * 16);
    for (auto &x : A16) x = (_Float16)((rand() % 2001 - 1000) / 512.0f);
    for (auto &x : B16) x = (_Float16)((rand() % 2001 - 1000) / 512.0f);
    _Float16 *dA16, *dB16;
    hipMalloc(&dA16, A16.size() * 2); hipMalloc(&dB16, B16.size() * 2);
    hipMemcpy(dA16, A16.data(), A16.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB16, B16.data(), B16.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA16, dB16, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    worst = 0; mag = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double s = 0;
            for (int k = 0; k < 32; k++) s += (double)(float)A16[i * 32 + k] * (double)(float)B16[k * 16 + j];
            worst = fmax(worst, fabs(s - D[i * 16 + j])); mag = fmax(mag, fabs(s));
        }
    printf("f16 16x16x32: max |ref - gpu| = %.3e (max |ref| %.3e) -> %s\n", worst, mag, worst <= 1e-5 * mag ? "layout confirmed" : "MISMATCH");
    return 0;
}
