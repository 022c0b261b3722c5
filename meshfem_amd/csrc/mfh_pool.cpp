// Device allocations of the library (DBuf): a per-process cache of released blocks.
//
// Why: on this driver stack a hipMalloc that follows large hipFree calls of the same process can take SECONDS (measured on MI355X / ROCm 7.2,
// profiles/r04_malloc_probe.txt: 32 GB again after a free 1.75 s, 16 x (8 GB hipMalloc + hipFree) 3.65 s, 120 GB 5.7 s -- against 0.3 ms for
// the first 32 GB of the process; the stream-ordered pool, which keeps its memory, serves the same sequence in 7 ms). The setup phases of a
// context allocate and release about 3.5x the memory they end up holding (sort keys, scan storage, lists), so the first assembly of a
// 40 M-element mesh in a process that had released memory before took 5.4 - 7.4 s, "almost all of it hipMalloc / hipFree" (VERDICT r3).
// The reference reserves its triplet storage once (LinearElasticity.hh:1441-1443).
//
// What: a released block goes to a size-ordered free list instead of back to the driver; an allocation takes the smallest cached block
// that fits without wasting more than a quarter (+1 MiB), else asks the driver; when the driver is out of memory the cache is flushed
// and the request repeated. The cache is bounded (MFH_DEVICE_CACHE_MB, default FOUR FIFTHS of the device's memory): beyond that the largest blocks
// go back to the driver -- and the next large hipMalloc pays for it (the stall grows with the bytes freed, ~18 GB/s): with a bound of a third, and
// still with half, a released 129 GB context (configs[4]) on top of the setup's cached temporaries lost its 58 GB value array and the next
// context waited 1 - 3 s for a new one (scripts/setup_probe.py 119, round 4). What the library itself cannot allocate any more flushes the
// cache and is retried; OTHER allocators of the process (torch, RCCL) do not know about it: mfh_device_cache_trim() before handing them the device. MFH_DEVICE_CACHE_MB=0 restores plain hipMalloc / hipFree. mfh_device_cache_trim() empties it on request.
#include "mfh_internal.hh"
#include <map>
#include <mutex>
#include <unordered_map>

namespace mfh {

namespace {
struct Block { size_t bytes; int dev; };
struct DevCache {
    std::multimap<size_t, void *> free;      // released blocks by size
    size_t cachedBytes = 0, limit = 0;
    bool init = false, enabled = true;
    int64_t hits = 0, misses = 0, flushes = 0, bypassed = 0;
    size_t evictedBytes = 0;
};
std::mutex g_mu;
std::map<int, DevCache> g_cache;                 // one free list per device
std::unordered_map<void *, Block> g_blocks;      // every live or cached block handed out by device_alloc, with the device it lives on
thread_local hipStream_t t_streams[2] = {nullptr, nullptr};
thread_local int t_mode = 0;

void wait_for_users(int blockDev, int currentDev) {
    if (t_mode == 2) return;
    if (t_mode == 1) {
        (void)hipStreamSynchronize(t_streams[0]);
        if (t_streams[1]) (void)hipStreamSynchronize(t_streams[1]);
        return;
    }
    (void)blockDev; (void)currentDev;                             // (mode 0 never reaches here: device_free hands such blocks to hipFree)
}

DevCache &cache_of(int dev) {
    DevCache &C = g_cache[dev];
    if (!C.init) {
        C.init = true;
        size_t fr = 0, total = 0;
        if (hipMemGetInfo(&fr, &total) != hipSuccess) { (void)hipGetLastError(); total = (size_t)64 << 30; }
        C.limit = total / 5 * 4;
        if (const char *e = getenv("MFH_DEVICE_CACHE_MB")) {
            const long long mb = atoll(e);
            if (mb <= 0) C.enabled = false;
            else C.limit = (size_t)mb << 20;
        }
    }
    return C;
}

void flush_locked(DevCache &C) {
    for (auto &kv : C.free) { g_blocks.erase(kv.second); (void)hipFree(kv.second); }
    C.free.clear();
    C.cachedBytes = 0;
    ++C.flushes;
}
}   // namespace

void *device_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    int dev = 0;
    MFH_HIP(hipGetDevice(&dev));
    bytes = (bytes + 255) & ~(size_t)255;
    std::lock_guard<std::mutex> lock(g_mu);
    DevCache &C = cache_of(dev);
    if (C.enabled) {
        auto it = C.free.lower_bound(bytes);
        if (it != C.free.end() && it->first <= bytes + bytes / 4 + ((size_t)1 << 20)) {
            void *p = it->second;
            C.cachedBytes -= it->first;
            C.free.erase(it);
            ++C.hits;
            return p;
        }
    }
    void *p = nullptr;
    static const bool trace = getenv("MFH_POOL_TRACE") != nullptr;
    const double tTrace = trace ? now_ms() : 0.0;
    hipError_t e = hipMalloc(&p, bytes);
    if (trace && (now_ms() - tTrace > 20.0 || bytes >= ((size_t)4 << 30))) {
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        fprintf(stderr, "[pool] hipMalloc %.1f MB: %.1f ms; cached %.1f GB in %zu blocks, device free %.1f GB, returned to the driver so far %.1f GB (%lld blocks outside an API scope)\n",
                bytes / 1e6, now_ms() - tTrace, C.cachedBytes / 1e9, C.free.size(), fr / 1e9, C.evictedBytes / 1e9, (long long)C.bypassed);
    }
    if (e != hipSuccess && C.enabled && !C.free.empty()) {
        (void)hipGetLastError();
        flush_locked(C);
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        throw Error(MFH_ERR_HIP, std::string("hipMalloc of ") + std::to_string(bytes >> 20) + " MiB: " + hipGetErrorString(e));
    }
    if (C.enabled) g_blocks[p] = Block{bytes, dev};
    ++C.misses;
    return p;
}

void device_free(void *p) {
    if (!p) return;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return; }
    std::unique_lock<std::mutex> lock(g_mu);
    auto it = g_blocks.find(p);
    if (it == g_blocks.end()) {            // allocated with the cache disabled
        lock.unlock();
        (void)hipFree(p);
        return;
    }
    const Block blk = it->second;          // the block goes back to the list of the device it lives on, whatever the caller's current device
    if (t_mode == 0) {
        // released outside any API entry (no PoolScope: the streams that may still use the block are unknown): plain hipFree, which waits for
        // the block's users by itself -- NOT a device-wide wait, which would invalidate the stream captures of other host threads
        DevCache &C0 = cache_of(blk.dev);
        ++C0.bypassed;
        C0.evictedBytes += blk.bytes;
        g_blocks.erase(it);
        lock.unlock();
        (void)hipFree(p);
        return;
    }
    lock.unlock();
    wait_for_users(blk.dev, cur);          // what hipFree did implicitly: nothing in flight uses the block when somebody else gets it
    lock.lock();
    DevCache &C = cache_of(blk.dev);
    C.free.emplace(blk.bytes, p);
    C.cachedBytes += blk.bytes;
    while (C.cachedBytes > C.limit && !C.free.empty()) {      // over the bound: the largest blocks go back to the driver
        auto last = std::prev(C.free.end());
        C.cachedBytes -= last->first;
        C.evictedBytes += last->first;
        g_blocks.erase(last->second);
        (void)hipFree(last->second);
        C.free.erase(last);
    }
}

PoolScope::PoolScope(hipStream_t a, hipStream_t b, int mode) {
    saved[0] = t_streams[0]; saved[1] = t_streams[1]; savedMode = t_mode;
    t_streams[0] = a; t_streams[1] = b; t_mode = mode;
}
PoolScope::~PoolScope() { t_streams[0] = saved[0]; t_streams[1] = saved[1]; t_mode = savedMode; }

void device_cache_trim() {
    std::lock_guard<std::mutex> lock(g_mu);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    for (auto &kv : g_cache) {
        if (kv.second.free.empty()) continue;
        (void)hipSetDevice(kv.first);
        (void)hipDeviceSynchronize();
        flush_locked(kv.second);
    }
    if (cur >= 0) (void)hipSetDevice(cur);     // the caller's current device is left as it was
}

void device_cache_stats(int dev, int64_t *cachedBytes, int64_t *blocks, int64_t *hits, int64_t *misses, int64_t *flushes) {
    std::lock_guard<std::mutex> lock(g_mu);
    DevCache &C = g_cache[dev];
    if (cachedBytes) *cachedBytes = (int64_t)C.cachedBytes;
    if (blocks) *blocks = (int64_t)C.free.size();
    if (hits) *hits = C.hits;
    if (misses) *misses = C.misses;
    if (flushes) *flushes = C.flushes;
}

}   // namespace mfh
