// HIP virtual memory management as the base of a device arena: one reserved address range, physical chunks created and mapped as the arena grows,
// never given back -- does the driver's hipMalloc-after-hipFree stall (malloc_probe.hip) have a counterpart here?
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/vmm_probe.hip -o scripts/probe/bin/vmm_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(double *p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1.0; }
int main() {
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu bytes\n", gran);
    const size_t GB = (size_t)1 << 30, VA = 256 * GB;
    void *base = nullptr;
    double t0 = now(); CK(hipMemAddressReserve(&base, VA, 0, nullptr, 0)); printf("reserve 256 GiB of address space: %.2f ms\n", now() - t0);
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t chunk : {1 * GB, 8 * GB}) {
        const int n = (int)(64 * GB / chunk);
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (int rep = 0; rep < 2; ++rep) {
            double tc = 0, tm = 0, ta = 0;
            for (int i = 0; i < n; ++i) {
                double a = now(); CK(hipMemCreate(&h[i], chunk, &prop, 0)); double b = now();
                CK(hipMemMap((char *)base + (size_t)i * chunk, chunk, 0, h[i], 0)); double c = now();
                CK(hipMemSetAccess((char *)base + (size_t)i * chunk, chunk, &acc, 1)); double d = now();
                tc += b - a; tm += c - b; ta += d - c;
            }
            double tk = now(); hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (double *)base, (size_t)64 * GB / 8); CK(hipDeviceSynchronize()); tk = now() - tk;
            double tu = now();
            for (int i = 0; i < n; ++i) { CK(hipMemUnmap((char *)base + (size_t)i * chunk, chunk)); CK(hipMemRelease(h[i])); }
            tu = now() - tu;
            printf("64 GiB in %d chunks of %zu GiB, rep %d: create %.1f ms, map %.1f ms, set access %.1f ms, first touch of all of it %.1f ms, unmap + release %.1f ms\n",
                   n, chunk / GB, rep, tc, tm, ta, tk, tu);
        }
    }
    // the pattern that stalls hipMalloc: grow to 128 GiB, shrink to 0, grow again
    {
        const size_t chunk = 2 * GB; const int n = 64;
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (int rep = 0; rep < 3; ++rep) {
            double t = now();
            for (int i = 0; i < n; ++i) { CK(hipMemCreate(&h[i], chunk, &prop, 0)); CK(hipMemMap((char *)base + (size_t)i * chunk, chunk, 0, h[i], 0)); CK(hipMemSetAccess((char *)base + (size_t)i * chunk, chunk, &acc, 1)); }
            double t1 = now();
            for (int i = 0; i < n; ++i) { CK(hipMemUnmap((char *)base + (size_t)i * chunk, chunk)); CK(hipMemRelease(h[i])); }
            printf("128 GiB in 2 GiB chunks, rep %d: grow %.1f ms, shrink %.1f ms\n", rep, t1 - t, now() - t1);
        }
    }
    // for comparison in the same process: hipMalloc after those releases
    { void *p; double t = now(); CK(hipMalloc(&p, 32 * GB)); printf("hipMalloc 32 GiB after the releases: %.1f ms\n", now() - t); CK(hipFree(p)); t = now(); CK(hipMalloc(&p, 32 * GB)); printf("hipMalloc 32 GiB again after hipFree: %.1f ms\n", now() - t); CK(hipFree(p)); }
    CK(hipMemAddressFree(base, VA));
    return 0;
}
