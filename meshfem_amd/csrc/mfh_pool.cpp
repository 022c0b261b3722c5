// Device allocations of the library (DBuf): a per-process ARENA that sub-allocates inside the segments it holds.
//
// Why an arena at all: on this driver stack a hipMalloc that follows large hipFree calls of the same process can take SECONDS (measured on
// MI355X / ROCm 7.2, profiles/r04_malloc_probe.txt: 32 GB again after a free 1.75 s, 16 x (8 GB hipMalloc + hipFree) 3.65 s, 120 GB 5.7 s --
// against 0.3 ms for the first 32 GB of the process: memory the process never held is handed out lazily, memory it released is cleared
// first, at ~18 GB/s). The setup phases of a context allocate and release about 3.5x the memory they end up holding (sort keys, scan
// storage, lists). The reference reserves its triplet storage once (LinearElasticity.hh:1441-1443).
//
// Why it splits and coalesces (round 5; rounds 3-4 had a size-bucket list that reused a released block only for a request of nearly the
// same size): with buckets a process that had run other meshes first held its memory in blocks of the WRONG sizes, the next large context
// missed all of them, and on a device that the cache had filled the driver was asked again -- 4.5 s in the symbolic phase of the 119^3 cube
// on the round-4 driver box against 0.36 s in a fresh process. Here a released chunk merges with its free neighbours (address order, inside
// its segment) and any request is cut from the smallest free chunk that holds it (best fit); the driver is asked for a new segment only
// when nothing fits. Requests below 1 MiB share 64 MiB segments; larger ones are rounded up to 2 MiB, get a segment of their own size when
// nothing fits (which later requests may split at 2 MiB boundaries) and never share a segment with the small ones.
//
// Bounds. (1) Free bytes the arena keeps while contexts are alive: MFH_DEVICE_CACHE_MB, default HALF the device; beyond it whole free
// segments go back to the driver, largest first. (2) When the LAST context of a device closes (mfh_destroy ->
// device_arena_context_closed) whole free segments go back until at most a quarter of the device stays (MFH_DEVICE_CACHE_IDLE_MB):
// other allocators of the process (torch, RCCL) then find the memory. mfh_device_cache_trim() gives back every free segment;
// MFH_DEVICE_CACHE_MB=0 restores plain hipMalloc / hipFree. A request the driver cannot serve releases the free segments and is repeated.
#include "mfh_internal.hh"
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <atomic>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace mfh {

namespace {
// Two size classes that never share a segment. SMALL: requests below 1 MiB, multiples of 256 B, cut from 64 MiB segments -- the few
// hundred long-lived tables of a context must not pin (or cut notches into) the multi-GB segments. LARGE: everything else, rounded up to
// 2 MiB and cut at 2 MiB boundaries: a large buffer then starts on the same kind of boundary that a hipMalloc of its own would give it.
constexpr size_t ALIGN = 256;
constexpr size_t SMALL_REQUEST = (size_t)1 << 20;
constexpr size_t SMALL_SEGMENT = (size_t)64 << 20;
constexpr size_t SEGMENT_ROUND = (size_t)2 << 20;
// Experiment knob: where the segments come from. MFH_ARENA_ALLOC = "plain" (hipMalloc, the default), "contiguous" (hipExtMallocWithFlags, hipDeviceMallocContiguous),
// "vmm" (hipMemCreate + hipMemMap: one physical allocation per segment mapped into a reserved address range), for segments of MFH_ARENA_ALLOC_MIN_MB ... _MAX_MB. Measured in docs/design/04_2_k_assemble_gather.md (xi):
// contiguous memory is the WORST home for the K values and for the solver's vectors; nothing in the product path sets these.
int alloc_kind() {
    static const int k = [] {
        const char *e = getenv("MFH_ARENA_ALLOC");
        if (!e) return 0;
        if (!strcmp(e, "contiguous")) return 1;
        if (!strcmp(e, "vmm")) return 2;
        return 0;
    }();
    return k;
}
std::unordered_map<void *, size_t> g_vmm;     // segments mapped through the virtual-memory API: base -> mapped bytes (under g_mu or single-threaded use)
std::mutex g_vmmMu;
hipError_t seg_malloc(void **p, size_t bytes) {
    static const size_t minBytes = [] { const char *e = getenv("MFH_ARENA_ALLOC_MIN_MB"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 64) << 20; }();
    static const size_t maxBytes = [] { const char *e = getenv("MFH_ARENA_ALLOC_MAX_MB"); const long v = e ? atol(e) : 0; return v > 0 ? (size_t)v << 20 : ~(size_t)0; }();
    const int kind = (bytes >= minBytes && bytes <= maxBytes) ? alloc_kind() : 0;
    if (kind == 1) {
        const hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
        return hipMalloc(p, bytes);
    }
    if (kind == 2) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran) {
            const size_t total = (bytes + gran - 1) / gran * gran;
            void *base = nullptr;
            hipMemGenericAllocationHandle_t h;
            if (hipMemAddressReserve(&base, total, (size_t)2 << 20, nullptr, 0) == hipSuccess) {
                if (hipMemCreate(&h, total, &prop, 0) == hipSuccess) {
                    hipMemAccessDesc acc = {};
                    acc.location.type = hipMemLocationTypeDevice;
                    acc.location.id = dev;
                    acc.flags = hipMemAccessFlagsProtReadWrite;
                    if (hipMemMap(base, total, 0, h, 0) == hipSuccess && hipMemSetAccess(base, total, &acc, 1) == hipSuccess) {
                        (void)hipMemRelease(h);
                        std::lock_guard<std::mutex> lk(g_vmmMu);
                        g_vmm[base] = total;
                        *p = base;
                        return hipSuccess;
                    }
                    (void)hipMemRelease(h);
                }
                (void)hipMemAddressFree(base, total);
            }
        }
        (void)hipGetLastError();
        return hipMalloc(p, bytes);
    }
    return hipMalloc(p, bytes);
}
hipError_t seg_free(void *p) {
    size_t total = 0;
    {
        std::lock_guard<std::mutex> lk(g_vmmMu);
        auto it = g_vmm.find(p);
        if (it != g_vmm.end()) { total = it->second; g_vmm.erase(it); }
    }
    if (!total) return hipFree(p);
    (void)hipDeviceSynchronize();
    hipError_t e = hipMemUnmap(p, total);
    if (e == hipSuccess) e = hipMemAddressFree(p, total);
    return e;
}
// Experiment knobs (scripts/layout_probe.sh; no effect measured, docs/design/04_2 (xi)): granularity of the LARGE class inside a segment and a stagger added to successive large requests,
// so that the buffers of a context cut from ONE segment do not all start on 2 MiB boundaries of the same physical run.
size_t large_gran() { static const size_t g = [] { const char *e = getenv("MFH_ARENA_GRAN_KB"); const long v = e ? atol(e) : 0; return v > 0 ? (size_t)v << 10 : SEGMENT_ROUND; }(); return g; }
size_t large_stagger() { static const size_t g = [] { const char *e = getenv("MFH_ARENA_STAGGER_KB"); const long v = e ? atol(e) : 0; return v > 0 ? (size_t)v << 10 : (size_t)0; }(); return g; }

struct Chunk {
    size_t bytes;
    char *segment;      // base address of the segment it lies in
    bool free;
};
struct Segment { size_t bytes, freeBytes; int cls; };
struct Arena {
    std::map<char *, Chunk> chunks;                       // every chunk of every segment, by address
    std::set<std::pair<size_t, char *>> freeBySize[3];    // the free ones of every class (0 small, 1 large, 2 K values), by (size, address): best fit = lower_bound
    std::map<char *, Segment> segments;
    std::vector<std::pair<char *, size_t>> quarantine;    // released outside an API scope: users unknown until the next device-wide wait
    size_t held = 0, live = 0, liveHigh = 0, limit = 0, idleLimit = 0, quarantined = 0;
    bool init = false, enabled = true;
    int contexts = 0;
    int pendingReserve = 0;                               // reservations under way on other threads (device_arena_reserve)
    size_t pendingBytes = 0;                              // ... and their bytes
    int64_t hits = 0, misses = 0, flushes = 0, bypassed = 0;
    size_t returnedBytes = 0;
    size_t free_bytes() const { return held - live - quarantined; }
};
std::mutex g_mu;
std::condition_variable g_cv;                            // a reservation has arrived
std::map<int, Arena> g_arena;                            // one arena per device
std::unordered_map<void *, int> g_owner;                 // live pointer handed out by device_alloc -> device
thread_local hipStream_t t_streams[2] = {nullptr, nullptr};
thread_local int t_mode = 0;
thread_local int t_tag = 0;                              // PoolTag: 1 = the request is the value array of K
// Class 2, "K values" (round 6; docs/design/04_2 (xi)): the assembly kernel's time follows WHERE the value array of K lies -- in one physical run with the
// context's other buffers (one large hipMalloc cut into pieces) it sits at the slow end of its spread on every box (3.27 against 2.9-3.1 ms at 5 M
// quadratic tets). The value array therefore never shares a segment with anything else: requests made under PoolTag(1) are served from segments of
// their own (reserved by the values share of mfh_device_reserve, or taken from the driver on demand).
constexpr size_t VALUES_MIN = (size_t)64 << 20;

void wait_for_users() {
    if (t_mode != 1) return;
    // a failed wait (e.g. a stream the caller has destroyed meanwhile) must not surface at an unrelated launch later: clear it
    if (hipStreamSynchronize(t_streams[0]) != hipSuccess) (void)hipGetLastError();
    if (t_streams[1] && hipStreamSynchronize(t_streams[1]) != hipSuccess) (void)hipGetLastError();
}

Arena &arena_of(int dev) {
    Arena &A = g_arena[dev];
    if (!A.init) {
        A.init = true;
        size_t fr = 0, total = 0;
        if (hipMemGetInfo(&fr, &total) != hipSuccess) { (void)hipGetLastError(); total = (size_t)64 << 30; }
        // several processes on one device (forced multi-rank runs, tests): every one of them has an arena, and the out-of-memory path of one cannot
        // reclaim what another holds -- the bounds are shares of the device (MFH_DEVICE_SHARERS, set by meshfem_amd.distributed from the local world size)
        size_t sharers = 1;
        if (const char *e = getenv("MFH_DEVICE_SHARERS")) sharers = (size_t)std::max(1L, atol(e));
        A.limit = total / 2 / sharers;
        A.idleLimit = total / 4 / sharers;
        if (const char *e = getenv("MFH_DEVICE_CACHE_MB")) {
            const long long mb = atoll(e);
            if (mb <= 0) A.enabled = false;
            else A.limit = (size_t)mb << 20;
        }
        if (const char *e = getenv("MFH_DEVICE_CACHE_IDLE_MB")) A.idleLimit = (size_t)std::max(0LL, atoll(e)) << 20;
        A.idleLimit = std::min(A.idleLimit, A.limit);
    }
    return A;
}

void insert_free(Arena &A, char *p, size_t bytes, char *seg) {
    // merge with the free neighbours of the same segment
    auto &freeBySize = A.freeBySize[A.segments[seg].cls];
    auto it = A.chunks.find(p);
    if (it != A.chunks.begin()) {
        auto prev = std::prev(it);
        if (prev->second.free && prev->second.segment == seg && prev->first + prev->second.bytes == p) {
            freeBySize.erase({prev->second.bytes, prev->first});
            bytes += prev->second.bytes;
            p = prev->first;
            A.chunks.erase(it);
            it = prev;
        }
    }
    auto next = std::next(it);
    if (next != A.chunks.end() && next->second.free && next->second.segment == seg && p + bytes == next->first) {
        freeBySize.erase({next->second.bytes, next->first});
        bytes += next->second.bytes;
        A.chunks.erase(next);
    }
    it->second = Chunk{bytes, seg, true};
    freeBySize.insert({bytes, p});
}

// give whole free segments back to the driver, largest first, while pred() holds; returns the bytes returned
template <class Pred> size_t release_free_segments(Arena &A, Pred keepGoing) {
    size_t done = 0;
    while (keepGoing()) {
        char *best = nullptr;
        size_t bestBytes = 0;
        for (auto &kv : A.segments)
            if (kv.second.freeBytes == kv.second.bytes && kv.second.bytes > bestBytes) { best = kv.first; bestBytes = kv.second.bytes; }
        if (!best) break;
        A.freeBySize[A.segments[best].cls].erase({bestBytes, best});
        A.chunks.erase(best);
        A.segments.erase(best);
        A.held -= bestBytes;
        A.returnedBytes += bestBytes;
        done += bestBytes;
        (void)seg_free(best);
    }
    return done;
}

// chunks released outside an API scope become reusable after a device-wide wait (the caller has made one)
void drain_quarantine(Arena &A) {
    for (auto &q : A.quarantine) {
        auto it = A.chunks.find(q.first);
        if (it == A.chunks.end()) continue;
        char *seg = it->second.segment;
        A.segments[seg].freeBytes += q.second;
        insert_free(A, q.first, q.second, seg);
    }
    A.quarantine.clear();
    A.quarantined = 0;
}

char *carve(Arena &A, int cls, std::set<std::pair<size_t, char *>>::iterator fit, size_t bytes) {
    char *p = fit->second;
    const size_t have = fit->first;
    A.freeBySize[cls].erase(fit);
    Chunk &c = A.chunks[p];
    char *seg = c.segment;
    if (have > bytes) {                    // the rest stays free (a neighbour of p in address order; its other neighbour is not free: they were merged)
        A.chunks[p + bytes] = Chunk{have - bytes, seg, true};
        A.freeBySize[cls].insert({have - bytes, p + bytes});
    }
    c = Chunk{bytes, seg, false};
    A.segments[seg].freeBytes -= bytes;
    return p;
}
}   // namespace

// Debug aid (MFH_ARENA_GUARD=1): every request is followed by GUARD bytes of a pattern that device_free checks -- a kernel that writes past the end of
// its buffer (the batch's work vectors of round 6 did, into slack they happened to own) is reported with the buffer's size instead of corrupting a
// neighbour some day. Synchronous fills and read-backs: for test runs, not for timing.
constexpr size_t GUARD = 512;
bool guard_on() { static const bool g = getenv("MFH_ARENA_GUARD") != nullptr && atoi(getenv("MFH_ARENA_GUARD")) != 0; return g; }
std::unordered_map<void *, size_t> g_guarded;           // pointer -> requested bytes (under g_mu)
static void guard_check(void *p, size_t requested) {
    unsigned char h[GUARD];
    (void)hipDeviceSynchronize();
    if (hipMemcpy(h, (char *)p + requested, GUARD, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
    size_t first = GUARD, count = 0;
    for (size_t i = 0; i < GUARD; ++i) if (h[i] != 0xA5) { if (first == GUARD) first = i; ++count; }
    if (count) {
        fprintf(stderr, "[arena guard] OVERRUN: %zu of the %zu guard bytes behind a buffer of %zu bytes at %p were written (first at +%zu)\n", count, GUARD, requested, p, first);
        abort();
    }
}

void *device_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    int dev = 0;
    MFH_HIP(hipGetDevice(&dev));
    const size_t requested = bytes;
    if (guard_on()) bytes += GUARD;
    const int cls = bytes < SMALL_REQUEST ? 0 : ((t_tag == 1 && bytes >= VALUES_MIN) ? 2 : 1);
    if (cls == 1 && large_stagger()) { static std::atomic<unsigned> turn{0}; bytes += (size_t)(turn.fetch_add(1) % 8u) * large_stagger(); }
    bytes = cls == 0 ? (bytes + ALIGN - 1) & ~(ALIGN - 1) : (bytes + large_gran() - 1) / large_gran() * large_gran();
    std::unique_lock<std::mutex> lock(g_mu);
    Arena &A = arena_of(dev);
    if (!A.enabled) {
        void *p = nullptr;
        const hipError_t e = seg_malloc(&p, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); throw Error(MFH_ERR_HIP, std::string("hipMalloc of ") + std::to_string(bytes >> 20) + " MiB: " + hipGetErrorString(e)); }
        return p;
    }
    auto &freeBySize = A.freeBySize[cls];
    auto fit = freeBySize.lower_bound({bytes, nullptr});
    // a reservation under way on another thread (mfh_device_reserve, asynchronous) may bring what this request needs: wait for it rather
    // than ask the driver for the same memory a second time
    while (fit == freeBySize.end() && cls >= 1 && A.pendingReserve > 0) {
        g_cv.wait(lock);
        fit = freeBySize.lower_bound({bytes, nullptr});
    }
    if (fit != freeBySize.end()) ++A.hits;
    else {
        // nothing the arena holds fits: one new segment from the driver
        ++A.misses;
        size_t segBytes = cls == 0 ? SMALL_SEGMENT : bytes;
        static const bool trace = getenv("MFH_POOL_TRACE") != nullptr;
        const double tTrace = trace ? now_ms() : 0.0;
        void *p = nullptr;
        hipError_t e = seg_malloc(&p, segBytes);
        if (e != hipSuccess && segBytes > bytes) { (void)hipGetLastError(); segBytes = bytes; e = seg_malloc(&p, segBytes); }
        if (e != hipSuccess) {
            // the driver is out of memory: what the arena holds but does not use goes back, then once more
            (void)hipGetLastError();
            if (!A.quarantine.empty()) { (void)hipDeviceSynchronize(); drain_quarantine(A); }
            ++A.flushes;
            release_free_segments(A, [] { return true; });
            fit = freeBySize.lower_bound({bytes, nullptr});      // (the quarantine may have completed a chunk that fits)
            if (fit == freeBySize.end()) e = seg_malloc(&p, segBytes);
            else e = hipSuccess, p = nullptr;
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            throw Error(MFH_ERR_HIP, std::string("hipMalloc of ") + std::to_string(segBytes >> 20) + " MiB: " + hipGetErrorString(e));
        }
        if (trace && (now_ms() - tTrace > 20.0 || segBytes >= ((size_t)4 << 30))) {
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            fprintf(stderr, "[arena] new segment %.1f MB: %.1f ms; held %.1f GB in %zu segments (live %.1f GB), device free %.1f GB, returned to the driver so far %.1f GB\n",
                    segBytes / 1e6, now_ms() - tTrace, A.held / 1e9, A.segments.size(), A.live / 1e9, fr / 1e9, A.returnedBytes / 1e9);
        }
        if (p) {
            char *seg = (char *)p;
            A.segments[seg] = Segment{segBytes, segBytes, cls};
            A.chunks[seg] = Chunk{segBytes, seg, true};
            A.held += segBytes;
            fit = freeBySize.insert({segBytes, seg}).first;
        }
    }
    char *p = carve(A, cls, fit, bytes);
    A.live += bytes;
    A.liveHigh = std::max(A.liveHigh, A.live);
    g_owner[p] = dev;
    if (guard_on()) {
        g_guarded[p] = requested;
        lock.unlock();
        (void)hipDeviceSynchronize();
        MFH_HIP(hipMemset(p + requested, 0xA5, GUARD));
    }
    return p;
}

void device_free(void *p) {
    if (!p) return;
    std::unique_lock<std::mutex> lock(g_mu);
    if (guard_on()) {
        auto gi = g_guarded.find(p);
        if (gi != g_guarded.end()) {
            const size_t requested = gi->second;
            g_guarded.erase(gi);
            lock.unlock();
            guard_check(p, requested);
            lock.lock();
        }
    }
    auto own = g_owner.find(p);
    if (own == g_owner.end()) {            // allocated with the arena disabled
        lock.unlock();
        (void)seg_free(p);
        return;
    }
    const int dev = own->second;           // the chunk goes back to the arena of the device it lives on, whatever the caller's current device
    const int mode = t_mode;
    if (mode == 1) {
        lock.unlock();
        wait_for_users();                  // what hipFree did implicitly: nothing in flight uses the chunk when somebody else gets it
        lock.lock();
    }
    g_owner.erase(p);
    Arena &A = arena_of(dev);
    auto it = A.chunks.find((char *)p);
    if (it == A.chunks.end() || it->second.free) return;     // (cannot happen: every owned pointer is a live chunk)
    const size_t bytes = it->second.bytes;
    char *seg = it->second.segment;
    A.live -= bytes;
    if (mode == 0) {
        // released outside any API entry (no PoolScope: the streams that may still use the chunk are unknown). A chunk that is a whole
        // segment goes to hipFree, which waits for its users by itself -- NOT a device-wide wait here, which would invalidate the stream
        // captures of other host threads; a part of a segment waits in quarantine for the next device-wide wait the arena makes anyway
        ++A.bypassed;
        if (bytes == A.segments[seg].bytes) {
            A.chunks.erase(it);
            A.segments.erase(seg);
            A.held -= bytes;
            A.returnedBytes += bytes;
            lock.unlock();
            (void)seg_free(p);
            return;
        }
        A.quarantine.emplace_back((char *)p, bytes);
        A.quarantined += bytes;
        return;
    }
    A.segments[seg].freeBytes += bytes;
    insert_free(A, (char *)p, bytes, seg);
    if (A.free_bytes() > A.limit) release_free_segments(A, [&] { return A.free_bytes() > A.limit; });
}

PoolScope::PoolScope(hipStream_t a, hipStream_t b, int mode) {
    saved[0] = t_streams[0]; saved[1] = t_streams[1]; savedMode = t_mode;
    t_streams[0] = a; t_streams[1] = b; t_mode = mode;
}
PoolScope::~PoolScope() { t_streams[0] = saved[0]; t_streams[1] = saved[1]; t_mode = savedMode; }

// One segment of `bytes` (rounded up to 2 MiB) taken from the driver NOW and put into the arena as free space: what the contexts created
// afterwards need is then cut from it -- no call to the driver during their setup, and the temporaries of the setup phases merge back
// into one chunk. The reference's counterpart is the single reserve of its triplet storage (LinearElasticity.hh:1441-1443). Why a caller
// would ask: on a box whose device nobody has used since boot the driver clears whatever a process takes beyond the first ~66 GB WHILE it
// is being allocated, at 25-40 GB/s (profiles/r05_large_allocation_trace_119.log) -- 2-3 s inside the first assembly of a 40 M-element
// mesh. Asynchronous: the call returns at once and the allocation proceeds on a thread of its own, e.g. while the caller reads its mesh;
// an allocation of the library that finds nothing waits for it. Free space the arena already holds counts: nothing happens if a free
// chunk of that size exists. The bound on the free bytes is raised to the reservation while contexts are alive.
// cls 2: a segment for the value array of K (see VALUES_MIN). Returns false when a synchronous reservation could not get its memory.
bool device_arena_reserve(int dev, size_t bytes, bool async, int cls) {
    bytes = (bytes + SEGMENT_ROUND - 1) & ~(SEGMENT_ROUND - 1);
    if (bytes == 0) return true;
    if (cls != 2 || bytes < VALUES_MIN) cls = 1;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        int cur = 0;
        const bool haveCur = hipGetDevice(&cur) == hipSuccess;
        if (!haveCur) (void)hipGetLastError();
        (void)hipSetDevice(dev);
        Arena &A = arena_of(dev);            // (reads the device's memory size the first time)
        if (haveCur) (void)hipSetDevice(cur);
        if (!A.enabled) return true;
        if (!A.freeBySize[cls].empty() && std::prev(A.freeBySize[cls].end())->first >= bytes) return true;
        ++A.pendingReserve;
        // The reservation itself must not count as "too much free memory" when something else is released -- neither this one nor the ones made just before it
        // that have not been used yet (round 6: the two segments of one mfh_device_reserve_for, 95 + 63 GB, on top of 70 GB of free memory from earlier meshes:
        // the bound of 154 + ... GB was passed, and the first release of a small buffer gave the LARGEST free segments -- the reservation -- back to the driver;
        // the 119^3 context then waited 2.6 + 2.1 s for the driver to clear them again). The bound follows what is free now plus everything under way.
        A.pendingBytes += bytes;
        A.limit = std::max(A.limit, A.free_bytes() + A.pendingBytes + A.idleLimit / 4);
    }
    static std::once_flag once;
    std::call_once(once, [] {
        std::atexit([] {                        // a reservation still under way when the process ends is waited for (never abandoned inside the driver)
            std::unique_lock<std::mutex> lock(g_mu);
            g_cv.wait(lock, [] { for (auto &kv : g_arena) if (kv.second.pendingReserve > 0) return false; return true; });
        });
    });
    auto ok = std::make_shared<std::atomic<bool>>(true);
    auto work = [dev, bytes, cls, ok]() {
        void *p = nullptr;
        const double t0 = now_ms();
        hipError_t e = hipSetDevice(dev);
        if (e == hipSuccess) e = seg_malloc(&p, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        std::lock_guard<std::mutex> lock(g_mu);
        Arena &A = g_arena[dev];
        if (p) {
            char *seg = (char *)p;
            A.segments[seg] = Segment{bytes, bytes, cls};
            A.chunks[seg] = Chunk{bytes, seg, true};
            A.held += bytes;
            A.freeBySize[cls].insert({bytes, seg});
        } else *ok = false;
        if (getenv("MFH_POOL_TRACE")) fprintf(stderr, "[arena] reservation of %.1f MB: %s after %.1f ms\n", bytes / 1e6, p ? "arrived" : "FAILED", now_ms() - t0);
        --A.pendingReserve;
        A.pendingBytes -= std::min(A.pendingBytes, bytes);
        g_cv.notify_all();
    };
    if (async) { std::thread(work).detach(); return true; }
    work();
    return *ok;
}

PoolTag::PoolTag(int tag) : saved(t_tag) { t_tag = tag; }
PoolTag::~PoolTag() { t_tag = saved; }

void device_arena_context_opened(int dev) {
    std::lock_guard<std::mutex> lock(g_mu);
    ++arena_of(dev).contexts;
}

void device_arena_context_closed(int dev) {
    std::unique_lock<std::mutex> lock(g_mu);
    Arena &A = arena_of(dev);
    g_cv.wait(lock, [&] { return A.pendingReserve == 0; });
    if (A.contexts > 0) --A.contexts;
    if (!A.enabled) return;
    // chunks released outside an API scope wait in the quarantine for a device-wide synchronisation: with the last context gone nothing of this
    // library is running on the device, so this is the place for one -- otherwise they would pin their segments against the trim below
    if (A.contexts == 0 && !A.quarantine.empty()) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
        (void)hipSetDevice(dev);
        if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
        drain_quarantine(A);
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    // With no context left on the device only a reserve for the next one stays: the rest is for the process's other allocators (torch,
    // RCCL). NOT trimmed while other contexts live, and not to the high-water mark of the live bytes either (first version of this
    // arena, profiles/r05_arena_probe_119_warm_with_high_water_trim.log): memory this process has returned is what the driver hands out
    // -- after clearing it -- for the next large request; 28 GB returned by the closes of two 60^3 contexts cost the 58.5 GB value array
    // of the next 119^3 context 0.97 s, 167 GB returned 1.34 s (never-held memory: 0.3 ms).
    if (A.contexts == 0) release_free_segments(A, [&] { return A.free_bytes() > A.idleLimit; });
}

void device_cache_trim() {
    std::unique_lock<std::mutex> lock(g_mu);
    g_cv.wait(lock, [] { for (auto &kv : g_arena) if (kv.second.pendingReserve > 0) return false; return true; });
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    for (auto &kv : g_arena) {
        Arena &A = kv.second;
        if (A.held == A.live) continue;
        (void)hipSetDevice(kv.first);
        if (!A.quarantine.empty()) {
            if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
            drain_quarantine(A);
        }
        ++A.flushes;
        release_free_segments(A, [] { return true; });
    }
    if (cur >= 0) (void)hipSetDevice(cur);     // the caller's current device is left as it was
}

void device_cache_stats(int dev, int64_t *cachedBytes, int64_t *blocks, int64_t *hits, int64_t *misses, int64_t *flushes) {
    std::lock_guard<std::mutex> lock(g_mu);
    Arena &A = g_arena[dev];
    if (cachedBytes) *cachedBytes = (int64_t)(A.held - A.live);
    if (blocks) *blocks = (int64_t)(A.freeBySize[0].size() + A.freeBySize[1].size() + A.freeBySize[2].size());
    if (hits) *hits = A.hits;
    if (misses) *misses = A.misses;
    if (flushes) *flushes = A.flushes;
}

size_t device_arena_largest_free(int dev) {          // the largest free chunk of the large and K-values classes
    std::lock_guard<std::mutex> lock(g_mu);
    Arena &A = g_arena[dev];
    size_t best = 0;
    for (int cls = 1; cls <= 2; ++cls)
        if (!A.freeBySize[cls].empty()) best = std::max(best, std::prev(A.freeBySize[cls].end())->first);
    return best;
}

void device_arena_stats(int dev, int64_t out[8]) {
    std::lock_guard<std::mutex> lock(g_mu);
    Arena &A = g_arena[dev];
    out[0] = (int64_t)A.held; out[1] = (int64_t)A.live; out[2] = (int64_t)A.liveHigh; out[3] = (int64_t)A.segments.size();
    out[4] = (int64_t)(A.freeBySize[0].size() + A.freeBySize[1].size() + A.freeBySize[2].size()); out[5] = (int64_t)A.returnedBytes; out[6] = (int64_t)A.quarantined; out[7] = (int64_t)A.limit;
}

}   // namespace mfh
