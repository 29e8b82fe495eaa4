// What a write-only / read-only / copy kernel reaches on this part (the yardsticks of conv_wino6.hip's boundary kernels: the first one only
// writes V, the last one only reads Z and the residual, the ones between read and write):
//   hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_write.hip -o /tmp/hbm_write && /tmp/hbm_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void wr(f4 *p, size_t n, float v) {
    const f4 x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = x;
}
__global__ void rd(const f4 *p, size_t n, float *out) {
    f4 a = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) out[0] = a[0];
}
__global__ void cp(const f4 *p, f4 *q, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) q[i] = p[i];
}
int main() {
    const size_t bytes = (size_t)12 << 30, n = bytes / 16;
    f4 *a, *b; float *o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256 * 4, 256 * 8, 256 * 16}) {
        for (int k = 0; k < 3; k++) {
            float ms[3];
            for (int rep = 0; rep < 2; rep++) {      // second repetition counts
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, a, n, 1.0f);
                if (k == 1) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, a, n, o);
                if (k == 2) hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, a, b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[k], e0, e1);
            }
            const double moved = (k == 2 ? 2.0 : 1.0) * (double)bytes;
            printf("grid %5d  %-5s %7.3f ms  %6.2f TB/s\n", grid, k == 0 ? "write" : k == 1 ? "read" : "copy", ms[k], moved / ms[k] / 1e9);
        }
    }
    return 0;
}
