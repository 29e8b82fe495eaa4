// which lane's E8M0 scale does v_mfma_scale_f32_16x16x128_f8f6f4 apply to an element?  A one-hot (lane La, byte ta) = 1.0,
// B all ones; scale_a of lane l = 127 + code(l), scale_b = 127 (then the roles swapped).  D[i][j] = 2^code(applied lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ void sc(int La, int ta, int swap, float *out) {
    const int l = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b;
    for (int v = 0; v < 8; v++) b[v] = 0x38383838;
    if (l == La) a[ta >> 2] = 0x38 << (8 * (ta & 3));
    const int code = 127 - 32 + l;      // distinct per lane: result 2^(l - 32)
    f32x4 acc = {0, 0, 0, 0};
    if (!swap) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, code, 0, 127);
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, acc, 0, 0, 0, 127, 0, code);
    for (int r = 0; r < 4; r++) out[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
int main() {
    float *d; hipMalloc(&d, 1024);
    const int tests[][2] = {{0, 0}, {0, 20}, {5, 3}, {16, 0}, {21, 9}, {32, 0}, {37, 31}, {48, 0}, {63, 7}};
    for (int swap = 0; swap < 2; swap++)
        for (auto &t : tests) {
            hipLaunchKernelGGL(sc, dim3(1), dim3(64), 0, 0, t[0], t[1], swap, d);
            std::vector<float> h(256); hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
            printf("%s one-hot lane %2d byte %2d:", swap ? "B" : "A", t[0], t[1]);
            int shown = 0;
            for (int k = 0; k < 256 && shown < 3; k++) if (h[k] != 0.f) { printf("  D[%d][%d] = 2^%.0f -> scale of lane %d", k / 16, k % 16, log2f(h[k]), (int)log2f(h[k]) + 32); shown++; }
            printf("\n");
        }
    return 0;
}
