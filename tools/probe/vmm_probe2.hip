// Probe (round 3): which (offset, size) patterns hipMemMap / hipMemSetAccess accept when a range is grown by chunks of different sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(unsigned *p, size_t n, unsigned v) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + (unsigned)i; }
__global__ void check(const unsigned *p, size_t n, unsigned v, unsigned *bad) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n && p[i] != v + (unsigned)i) atomicAdd(bad, 1u); }
static hipMemAllocationProp prop;
static int try_pattern(const char *name, const std::vector<size_t> &sizes, int mode) {
    // mode 0: SetAccess per chunk; 1: SetAccess over the whole mapped prefix after each map; 2: per chunk, flags on a desc array of 1
    const size_t MB2 = (size_t)2 << 20;
    void *base = nullptr;
    const size_t reserve = (size_t)64 << 30;
    CK(hipMemAddressReserve(&base, reserve, 0, nullptr, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t off = 0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    int ok = 1;
    for (size_t k = 0; k < sizes.size() && ok; k++) {
        const size_t sz = sizes[k] * MB2;
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, sz, &prop, 0);
        if (e != hipSuccess) { printf("  %s: chunk %zu create %s\n", name, k, hipGetErrorString(e)); ok = 0; break; }
        hs.push_back(h);
        e = hipMemMap((char *)base + off, sz, 0, h, 0);
        if (e != hipSuccess) { printf("  %s: chunk %zu (off %zu x 2MB, size %zu x 2MB) map %s\n", name, k, off / MB2, sizes[k], hipGetErrorString(e)); ok = 0; break; }
        e = mode == 1 ? hipMemSetAccess(base, off + sz, &acc, 1) : hipMemSetAccess((char *)base + off, sz, &acc, 1);
        if (e != hipSuccess) { printf("  %s: chunk %zu (off %zu x 2MB, size %zu x 2MB) setaccess %s\n", name, k, off / MB2, sizes[k], hipGetErrorString(e)); ok = 0; (void)hipGetLastError(); break; }
        off += sz;
    }
    if (ok) {
        unsigned *bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
        const size_t n = off / 4;
        fill<<<(unsigned)((n + 255) / 256), 256>>>((unsigned *)base, n, 7u);
        check<<<(unsigned)((n + 255) / 256), 256>>>((unsigned *)base, n, 7u, bad);
        hipError_t e = hipDeviceSynchronize();
        unsigned hb = 1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("  %s: mapped %zu x 2MB in %zu chunks, kernel %s, mismatches %u\n", name, off / MB2, sizes.size(), hipGetErrorString(e), hb);
        hipFree(bad);
    }
    size_t o2 = 0;
    for (size_t k = 0; k < hs.size(); k++) { if (o2 < off + 1) hipMemUnmap((char *)base + o2, sizes[k] * MB2); hipMemRelease(hs[k]); o2 += sizes[k] * MB2; }
    hipMemAddressFree(base, reserve);
    (void)hipGetLastError();
    return 0;
}
int main() {
    CK(hipSetDevice(0));
    prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    hipError_t e1 = hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    hipError_t e2 = hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("granularity min %zu (%s) recommended %zu (%s)\n", gmin, hipGetErrorString(e1), grec, hipGetErrorString(e2));
    (void)hipGetLastError();
    for (int mode = 0; mode < 2; mode++) {
        printf("mode %d (%s)\n", mode, mode == 0 ? "SetAccess per chunk" : "SetAccess over the mapped prefix");
        try_pattern("equal 512,512,512", {512, 512, 512}, mode);
        try_pattern("81 then 567", {81, 567}, mode);
        try_pattern("81 then 81", {81, 81}, mode);
        try_pattern("567 then 81", {567, 81}, mode);
        try_pattern("1,1,1,1", {1, 1, 1, 1}, mode);
        try_pattern("3 then 5 then 64", {3, 5, 64}, mode);
        try_pattern("64 then 3", {64, 3}, mode);
    }
    return 0;
}
