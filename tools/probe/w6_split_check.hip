// Check (round 5): c6_w6_gemm_kernel's SPLIT form (rows as work items) against the block form on the same V / U: Z must be bit-identical
// (the same instruction stream per position).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/probe/_w6_split_check tools/probe/w6_split_check.hip
#include "../../alphazero-connect6_amd/csrc/conv_wino6.hip"
#include <cstdio>
#include <vector>
#include <cstring>
void c6_note_hip(hipError_t, const char *) {}
__global__ void fill_kernel(float *p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((int)(x & 0xffff) - 32768) * (1.f / 65536.f);
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    const int nboards = argc > 1 ? atoi(argv[1]) : 13;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const W6Map map(nboards);
    const int nblocks = (int)map.nblocks, n6 = (int)map.blkDP, n5 = nblocks - n6;
    const size_t vfl = (size_t)nblocks * W6_VBLOCK, zfl = (size_t)nblocks * W6_ZBLOCK;
    float *V, *Z0, *Z1, *U;
    CK(hipMalloc(&V, vfl * 4)); CK(hipMalloc(&Z0, zfl * 4)); CK(hipMalloc(&Z1, zfl * 4)); CK(hipMalloc(&U, c6_w6_packed_floats() * 4));
    fill_kernel<<<1024, 256>>>(V, vfl, 1u); fill_kernel<<<256, 256>>>(U, c6_w6_packed_floats(), 3u);
    CK(hipMemset(Z0, 0xff, zfl * 4)); CK(hipMemset(Z1, 0xff, zfl * 4));
    CK(hipDeviceSynchronize());
    constexpr size_t smem = (size_t)2 * W6_BT * (W6_C + 4) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    printf("V %p .. %p\nU %p .. %p\nZ0 %p .. %p\nZ1 %p .. %p\n", V, V + vfl, U, U + c6_w6_packed_floats(), Z0, Z0 + zfl, Z1, Z1 + zfl);
    printf("%d boards: %d blocks (%d + %d of class D'), blkB %ld blkBP %ld blkD %ld\n", nboards, nblocks, n6, n5, map.blkB, map.blkBP, map.blkD);
    hipLaunchKernelGGL((c6_w6_gemm_kernel<false, false>), dim3(n6 < cus ? n6 : cus), dim3(512), smem, 0, V, U, Z0, 0, n6, (int)map.blkB, (int)map.blkBP, (int)map.blkD);
    hipLaunchKernelGGL((c6_w6_gemm_kernel<true, false>), dim3(n5 < cus ? n5 : cus), dim3(512), smem, 0, V, U, Z0, n6, nblocks, (int)map.blkB, (int)map.blkBP, (int)map.blkD);
    CK(hipDeviceSynchronize());
    printf("block form done\n"); fflush(stdout);
    const long it6 = (long)map.blkB * 6 + (long)(map.blkD - map.blkB) * 5 + (long)(n6 - map.blkD) * 8;
    const int which = argc > 2 ? atoi(argv[2]) : 3;
    if (which & 1) {
        hipLaunchKernelGGL((c6_w6_gemm_kernel<false, true>), dim3(n6 < cus ? n6 : cus, 8), dim3(512), smem, 0, V, U, Z1, 0, n6, (int)map.blkB, (int)map.blkBP, (int)map.blkD);
        CK(hipDeviceSynchronize());
        printf("split form <false> done (%ld items)\n", it6); fflush(stdout);
    }
    if (which & 2) {
        hipLaunchKernelGGL((c6_w6_gemm_kernel<true, true>), dim3(n5 < cus ? n5 : cus, 8), dim3(512), smem, 0, V, U, Z1, n6, nblocks, (int)map.blkB, (int)map.blkBP, (int)map.blkD);
        CK(hipDeviceSynchronize());
        printf("split form <true> done (%d items)\n", 4 * n5); fflush(stdout);
    }
    if (which & 4) {      // the half-width jobs (four waves, 128 output channels each): both kernels again, into Z1
        CK(hipMemset(Z1, 0xff, zfl * 4));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_w6_gemm_kernel<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((c6_w6_gemm_kernel<false, true, true>), dim3(n6 < cus ? n6 : cus, 8, 2), dim3(256), smem, 0, V, U, Z1, 0, n6, (int)map.blkB, (int)map.blkBP, (int)map.blkD);
        hipLaunchKernelGGL((c6_w6_gemm_kernel<true, true, true>), dim3(n5 < cus ? n5 : cus, 8, 2), dim3(256), smem, 0, V, U, Z1, n6, nblocks, (int)map.blkB, (int)map.blkBP, (int)map.blkD);
        CK(hipDeviceSynchronize());
        printf("half-width split form done\n"); fflush(stdout);
    }
    if (which & 8) {      // the same jobs, both row forms in ONE launch (c6_w6_gemm_half_pair_kernel: what the library launches for a handful of games)
        CK(hipMemset(Z1, 0xff, zfl * 4));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_w6_gemm_half_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int g6 = n6 < cus ? n6 : cus, g5 = n5 < cus ? n5 : cus;
        hipLaunchKernelGGL(c6_w6_gemm_half_pair_kernel, dim3(g6 + g5, 8, 2), dim3(256), smem, 0, V, U, Z1, n6, nblocks, (int)map.blkB, (int)map.blkBP, (int)map.blkD, g6);
        CK(hipDeviceSynchronize());
        printf("half-width jobs, one launch: done\n"); fflush(stdout);
    }
    if (which & 16) {      // ... and the eight-wave split jobs in one launch (c6_w6_gemm_split_pair_kernel)
        CK(hipMemset(Z1, 0xff, zfl * 4));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&c6_w6_gemm_split_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int g6 = n6 < cus ? n6 : cus, g5 = n5 < cus ? n5 : cus;
        hipLaunchKernelGGL(c6_w6_gemm_split_pair_kernel, dim3(g6 + g5, 8), dim3(512), smem, 0, V, U, Z1, n6, nblocks, (int)map.blkB, (int)map.blkBP, (int)map.blkD, g6);
        CK(hipDeviceSynchronize());
        printf("split jobs, one launch: done\n"); fflush(stdout);
    }
    std::vector<float> h0(zfl), h1(zfl);
    CK(hipMemcpy(h0.data(), Z0, zfl * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), Z1, zfl * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, first = 0;
    const size_t lo = (which & 29) ? 0 : (size_t)n6 * W6_ZBLOCK, hi = (which & 30) ? zfl : (size_t)n6 * W6_ZBLOCK;
    for (size_t i = lo; i < hi; i++) if (memcmp(&h0[i], &h1[i], 4)) { if (!diff) first = i; diff++; }
    {
        std::vector<size_t> byblk(nblocks, 0); size_t byb[4] = {0, 0, 0, 0}, byhalf[2] = {0, 0}, byrow[9] = {0};
        for (size_t i = lo; i < hi; i++) if (memcmp(&h0[i], &h1[i], 4)) { byblk[i / W6_ZBLOCK]++; const size_t r = i % W6_ZTILE / W6_C; byb[r & 3]++; byrow[r >> 2]++; byhalf[(i % W6_ZBLOCK / W6_ZTILE) >> 5]++; }
        size_t bych[256] = {0};
        for (size_t i = lo; i < hi; i++) if (memcmp(&h0[i], &h1[i], 4)) bych[i % W6_C]++;
        printf("by output channel (nonzero):"); for (int k = 0; k < 256; k++) if (bych[k]) printf(" %d:%zu", k, bych[k]);
        printf("\n");
        printf("by block:"); for (int k = 0; k < nblocks; k++) printf(" %zu", byblk[k]);
        printf("\nby b: %zu %zu %zu %zu; by tile half: %zu %zu; by row:", byb[0], byb[1], byb[2], byb[3], byhalf[0], byhalf[1]);
        for (int k = 0; k < 9; k++) printf(" %zu", byrow[k]);
        printf("\n");
    }
    {   // host reference of a few row sums of block 0 (class A: positions p = 6 i + j, weight set at 0): Z_b = sum_j A4^T[b][j] * sum_cin U_p[cout][cin] V_p[tile][cin]
        std::vector<float> hv(W6_VBLOCK), hu((size_t)36 * 256 * 256);
        CK(hipMemcpy(hv.data(), V, W6_VBLOCK * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hu.data(), U, hu.size() * 4, hipMemcpyDeviceToHost));
        const double AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
        for (int tile : {0, 40}) for (int row : {0, 3}) for (int bb : {2, 3}) {
            const int cout = 5;
            double z = 0;
            for (int jj = 0; jj < 6; jj++) {
                const int pos = 6 * row + jj;
                double m = 0;
                for (int cin = 0; cin < 256; cin++) {
                    const int wv = cout / 32, ln = (cout & 31) + 32 * ((cin & 7) >> 2), u = cin >> 3, r = cin & 3;
                    m += (double)hu[((((size_t)wv * 36 + pos) * 32 + u) * 64 + ln) * 4 + r] * (double)hv[(size_t)tile * W6_VTILE + (size_t)pos * 256 + cin];
                }
                z += AT[bb][jj] * m;
            }
            const size_t idx = (size_t)tile * W6_ZTILE + (size_t)(row * 4 + bb) * 256 + cout;
            printf("block 0 tile %2d row %d b %d cout %d: host %.6f  block form %.6f  split form %.6f\n", tile, row, bb, cout, z, h0[idx], h1[idx]);
        }
    }
    printf("floats that differ: %zu of %zu", diff, zfl);
    if (diff) printf(" (first at block %zu tile %zu row-sum %zu channel %zu: %g vs %g)", first / W6_ZBLOCK, first % W6_ZBLOCK / W6_ZTILE, first % W6_ZTILE / W6_C, first % W6_C, h0[first], h1[first]);
    printf("\n");
    return diff != 0;
}
