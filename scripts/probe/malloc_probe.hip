// Cost of device allocations on this box: hipMalloc / first touch / hipFree / re-allocation, and the stream-ordered pool.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/malloc_probe.hip -o scripts/probe/bin/malloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreate(&s));
    void *w; CK(hipMalloc(&w, 1 << 20)); CK(hipFree(w));
    for (double gb : {1.0, 8.0, 32.0, 64.0}) {
        const size_t n = (size_t)(gb * 1e9);
        void *p = nullptr;
        double t0 = now(); CK(hipMalloc(&p, n)); double t1 = now();
        CK(hipMemsetAsync(p, 0, n, s)); CK(hipStreamSynchronize(s)); double t2 = now();
        CK(hipMemsetAsync(p, 1, n, s)); CK(hipStreamSynchronize(s)); double t3 = now();
        CK(hipFree(p)); double t4 = now();
        CK(hipMalloc(&p, n)); double t5 = now();
        CK(hipMemsetAsync(p, 0, n, s)); CK(hipStreamSynchronize(s)); double t6 = now();
        CK(hipFree(p)); double t7 = now();
        printf("%5.0f GB: hipMalloc %8.2f ms, first memset %8.2f ms, second memset %8.2f ms, hipFree %8.2f ms | again: hipMalloc %8.2f ms, memset %8.2f ms, hipFree %8.2f ms\n",
               gb, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6);
    }
    // many medium allocations in sequence (the symbolic phase's pattern): 16 x 8 GB alloc / free
    { double t0 = now(); for (int i = 0; i < 16; ++i) { void *p; CK(hipMalloc(&p, (size_t)8e9)); CK(hipFree(p)); } printf("16 x (hipMalloc 8 GB + hipFree): %.2f ms\n", now() - t0); }
    // stream-ordered pool that keeps what it has
    hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
    uint64_t thr = UINT64_MAX; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < 16; ++i) { void *p; CK(hipMallocAsync(&p, (size_t)8e9, s)); CK(hipMemsetAsync(p, 0, 1 << 20, s)); CK(hipFreeAsync(p, s)); }
        CK(hipStreamSynchronize(s));
        printf("pool, rep %d: 16 x (hipMallocAsync 8 GB + hipFreeAsync): %.2f ms\n", rep, now() - t0);
    }
    { void *a, *b, *c2; double t0 = now(); CK(hipMallocAsync(&a, (size_t)16e9, s)); CK(hipMallocAsync(&b, (size_t)16e9, s)); CK(hipStreamSynchronize(s)); double t1 = now();
      CK(hipFreeAsync(a, s)); CK(hipFreeAsync(b, s)); CK(hipMallocAsync(&c2, (size_t)30e9, s)); CK(hipStreamSynchronize(s)); double t2 = now();
      printf("pool: 2 x 16 GB %.2f ms; free both + 30 GB %.2f ms\n", t1 - t0, t2 - t1); CK(hipFreeAsync(c2, s)); CK(hipStreamSynchronize(s)); }
    // one big arena
    { void *p; double t0 = now(); CK(hipMalloc(&p, (size_t)120e9)); double t1 = now(); CK(hipMemsetAsync(p, 0, (size_t)120e9, s)); CK(hipStreamSynchronize(s)); double t2 = now(); CK(hipFree(p));
      printf("120 GB arena: hipMalloc %.2f ms, first memset %.2f ms, hipFree %.2f ms\n", t1 - t0, t2 - t1, now() - t2); }
    return 0;
}
