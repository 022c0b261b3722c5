// Experiment helper (scripts/vmm_probe.py): a device buffer built with the HIP virtual-memory API -- physical chunks of a chosen size mapped into a
// reserved address range with a chosen alignment, in order or shuffled -- to see what about "where the K values lie" sets the assembly kernel's time.
// hipcc -shared -fPIC -o vmm_alloc.so vmm_alloc.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
extern "C" void *vmm_alloc(size_t bytes, size_t chunkBytes, size_t vaAlign, int shuffle, int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("granularity failed\n"); return nullptr; }
    if (chunkBytes == 0) chunkBytes = bytes;
    chunkBytes = (chunkBytes + gran - 1) / gran * gran;
    const size_t nChunk = (bytes + chunkBytes - 1) / chunkBytes;
    const size_t total = nChunk * chunkBytes;
    void *base = nullptr;
    if (hipMemAddressReserve(&base, total, vaAlign, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return nullptr; }
    std::vector<size_t> order(nChunk);
    for (size_t i = 0; i < nChunk; ++i) order[i] = i;
    // the handles are created in index order (the driver hands out physical memory in that order), mapped at shuffled positions
    if (shuffle) { std::mt19937_64 g(12345); std::shuffle(order.begin(), order.end(), g); }
    for (size_t i = 0; i < nChunk; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunkBytes, &prop, 0) != hipSuccess) { printf("create failed at %zu\n", i); return nullptr; }
        if (hipMemMap((char *)base + order[i] * chunkBytes, chunkBytes, 0, h, 0) != hipSuccess) { printf("map failed at %zu\n", i); return nullptr; }
        (void)hipMemRelease(h);      // the mapping keeps the memory alive
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(base, total, &acc, 1) != hipSuccess) { printf("set access failed\n"); return nullptr; }
    printf("vmm_alloc: %zu chunks of %zu MiB (granularity %zu KiB), base %p\n", nChunk, chunkBytes >> 20, gran >> 10, base);
    return base;
}
