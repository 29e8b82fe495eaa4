// Probe (round 3): HIP virtual memory management on this pool -- reserve a large address range, map physical chunks into it as they are
// needed (a record pool that grows without copying).   hipcc -O2 -o /tmp/vmm tools/probe/vmm_probe.hip && /tmp/vmm
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(unsigned *p, size_t n, unsigned v) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + (unsigned)i; }
__global__ void check(const unsigned *p, size_t n, unsigned v, unsigned *bad) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n && p[i] != v + (unsigned)i) atomicAdd(bad, 1u); }
int main() {
    int dev = 0, vmm = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("VMM supported attribute: %d\n", vmm);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity %zu\n", gran);
    const size_t reserve = (size_t)200 << 30, chunk = (size_t)1 << 30;
    void *base = nullptr;
    CK(hipMemAddressReserve(&base, reserve, 0, nullptr, 0));
    printf("reserved 200 GiB at %p\n", base);
    unsigned *bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    hipMemGenericAllocationHandle_t h[4];
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int k = 0; k < 4; k++) {
        CK(hipMemCreate(&h[k], chunk, &prop, 0));
        CK(hipMemMap((char *)base + k * chunk, chunk, 0, h[k], 0));
        CK(hipMemSetAccess((char *)base + k * chunk, chunk, &acc, 1));
        const size_t n = chunk / 4;
        fill<<<(unsigned)((n + 255) / 256), 256>>>((unsigned *)((char *)base + k * chunk), n, 1000u * k);
        CK(hipDeviceSynchronize());
        // everything mapped so far is still intact, through ONE pointer
        for (int j = 0; j <= k; j++) check<<<(unsigned)((n + 255) / 256), 256>>>((unsigned *)((char *)base + j * chunk), n, 1000u * j, bad);
        CK(hipDeviceSynchronize());
        unsigned hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("chunk %d mapped, mismatches so far %u\n", k, hb);
    }
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot)); printf("free %.1f GiB of %.1f\n", fr / 1073741824.0, tot / 1073741824.0);
    for (int k = 0; k < 4; k++) { CK(hipMemUnmap((char *)base + k * chunk, chunk)); CK(hipMemRelease(h[k])); }
    CK(hipMemAddressFree(base, reserve));
    printf("VMM OK\n");
    return 0;
}
