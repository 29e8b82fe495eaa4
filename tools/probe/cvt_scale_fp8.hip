// semantics of v_cvt_scalef32_pk_fp8_f32 (gfx950): does it convert src * scale or src / scale?  prints the e4m3 bytes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(float a, float b, float scale, unsigned *o) {
    s16x2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, a, b, scale, false);
    o[0] = *(unsigned *)&r;
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
    o[1] = (unsigned)p;
}
int main() {
    unsigned *d, h[2];
    hipMalloc(&d, 8);
    const float tests[][3] = {{1.0f, 2.0f, 1.0f}, {1.0f, 2.0f, 8.0f}, {1.0f, 2.0f, 0.125f}, {1000.0f, -3.0f, 8.0f}, {1000.0f, -3.0f, 1.0f}, {0.3f, 0.011f, 0.125f}};
    for (auto &t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, t[0], t[1], t[2], d);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g b=%g scale=%g: scaled cvt bytes %02x %02x   plain cvt bytes %02x %02x\n", t[0], t[1], t[2], h[0] & 0xff, (h[0] >> 8) & 0xff, h[1] & 0xff, (h[1] >> 8) & 0xff);
    }
    return 0;
}
