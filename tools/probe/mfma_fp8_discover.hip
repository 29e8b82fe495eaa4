// one-hot discovery of the K mapping of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8, neutral scales)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
// block = (Lb, tb); A one-hot at (La, ta).  out[block] = (row i, col j) of the nonzero output or -1
__global__ void disc(int La, int ta, int *out) {
    const int l = threadIdx.x, Lb = blockIdx.x >> 5, tb = blockIdx.x & 31;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l == La) a[ta >> 2] = 0x38 << (8 * (ta & 3));
    if (l == Lb) b[tb >> 2] = 0x38 << (8 * (tb & 3));
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 127, 0, 127);
    for (int r = 0; r < 4; r++)
        if (acc[r] != 0.f) out[blockIdx.x] = ((4 * (l >> 4) + r) << 8) | (l & 15) | ((int)(acc[r] * 16) << 16);
}
int main() {
    int *d; hipMalloc(&d, 2048 * 4);
    const int tests[][2] = {{0, 0}, {0, 1}, {0, 4}, {0, 15}, {0, 16}, {0, 31}, {16, 0}, {16, 17}, {32, 0}, {48, 5}, {3, 0}, {19, 20}};
    for (auto &t : tests) {
        hipMemset(d, 0xff, 2048 * 4);
        hipLaunchKernelGGL(disc, dim3(2048), dim3(64), 0, 0, t[0], t[1], d);
        std::vector<int> h(2048); hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost);
        printf("A one-hot lane %2d byte %2d matches B at:", t[0], t[1]);
        int cnt = 0;
        for (int k = 0; k < 2048; k++) if (h[k] != -1) { if (cnt < 6) printf("  (lane %d byte %d -> D[%d][%d] val %.3f)", k >> 5, k & 31, (h[k] >> 8) & 0xff, h[k] & 0xff, (h[k] >> 16) / 16.0); cnt++; }
        printf("  [%d matches]\n", cnt);
    }
    return 0;
}
