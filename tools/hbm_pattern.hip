// hbm_pattern.hip -- what HBM delivers for K1's access pattern, without K1's arithmetic (load + xor only).
//   pattern 0: SoA planes   addr = cell * G + 4 * lane_global            (the engine's board_soa layout)
//   pattern 1: tiled AoSoA  addr = (tile * CELLS + cell) * TILE + 4 * lane_in_tile,  TILE = 1024 games (one workgroup)
//   pattern 2: linear sweep (each wave reads CELLS consecutive 256-byte chunks: upper bound)
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_pattern.hip -o /tmp/hbm_pattern ; run: /tmp/hbm_pattern [cells] [games]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int PATTERN, int UNR>
__global__ __launch_bounds__(256) void probe(const uint8_t *__restrict__ p, uint32_t *__restrict__ out, int cells, size_t G) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;      // 4 games per lane
    if (q * 4 >= G) return;
    uint32_t acc = 0;
    for (int c0 = 0; c0 < cells; c0 += UNR) {
        uint32_t v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int c = c0 + u < cells ? c0 + u : cells - 1;
            size_t a;
            if (PATTERN == 0) a = (size_t)c * G + q * 4;
            else if (PATTERN == 1) a = ((size_t)blockIdx.x * cells + c) * 1024 + threadIdx.x * 4;
            else a = ((size_t)(q >> 6) * cells + c) * 256 + (q & 63) * 4;
            v[u] = *reinterpret_cast<const uint32_t *>(p + a);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) acc ^= v[u];
    }
    if (acc == 0x12345678u) out[q & 1023] = acc;      // never true on zeroed memory; keeps the loads alive
}

template <int PATTERN>
static void run(const char *name, const uint8_t *d, uint32_t *o, int cells, size_t G) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (int)((G / 4 + 255) / 256);
    float best = 1e9f, sum = 0;
    for (int i = -1; i < 10; i++) {
        hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL((probe<PATTERN, 19>), dim3(grid), dim3(256), 0, nullptr, d, o, cells, G);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 0) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-14s cells=%d G=%zu: avg %.1f us, best %.1f us -> %.0f GB/s\n", name, cells, G, sum * 100, best * 1000, (double)cells * G / (sum / 10) / 1e6);
}

int main(int argc, char **argv) {
    const int cells = argc > 1 ? atoi(argv[1]) : 361;
    const size_t G = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)4 << 20;
    uint8_t *d; uint32_t *o;
    if (hipMalloc(&d, (size_t)cells * G + 4096) != hipSuccess || hipMalloc(&o, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, (size_t)cells * G + 4096);
    run<0>("SoA planes", d, o, cells, G);
    run<1>("AoSoA 1024", d, o, cells, G);
    run<2>("linear", d, o, cells, G);
    hipFree(d); hipFree(o);
    return 0;
}
