"""Write bandwidth of windows of a physically CONTIGUOUS buffer (hipExtMallocWithFlags, hipDeviceMallocContiguous) against a plain hipMalloc:
hipMemset over windows of 8 MB ... 8 GB.   python scripts/fill_probe2.py"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
MB, GB = 1 << 20, 1 << 30
def rate(p, nbytes, reps=5):
    best = 0
    n = max(1, int(2 * GB // nbytes))          # several windows' worth per timing, so that small windows are not launch-bound
    for _ in range(reps):
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            hip.hipMemset(C.c_void_p(p), 0, nbytes)
        hip.hipDeviceSynchronize(); dt = time.perf_counter() - t0
        best = max(best, n * nbytes / dt / 1e12)
    return best
for name, flag in (("plain hipMalloc", None), ("contiguous", 4), ("plain hipMalloc", None), ("contiguous", 4)):
    p = C.c_void_p()
    e = hip.hipMalloc(C.byref(p), 8 * GB) if flag is None else hip.hipExtMallocWithFlags(C.byref(p), 8 * GB, flag)
    if e: print(name, "allocation failed", e); continue
    base = p.value
    out = []
    for w in (8 * GB, 2 * GB, 512 * MB, 128 * MB, 32 * MB):
        out.append("%4d MB: %s" % (w // MB, " ".join("%.2f" % rate(base + k * (8 * GB // 4), w) for k in range(4 if w < 8 * GB else 1))))
    print("%-16s %s" % (name, " | ".join(out)), flush=True)
