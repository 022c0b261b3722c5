"""Write bandwidth (hipMemsetD8 over the whole buffer) of 8 GB buffers the arena obtained in different ways: is the driver's physical placement
visible to a plain fill?   python scripts/fill_probe.py"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
hip = C.CDLL("libamdhip64.so")
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
GB = 1 << 30
keep = M.Context(0)
def alloc(nbytes):
    p = C.c_void_p(); keep._ck(keep.lib.mfh_debug_arena_alloc(keep.h, int(nbytes), C.byref(p))); return p.value
def free(p):
    keep._ck(keep.lib.mfh_debug_arena_free(keep.h, C.c_void_p(p)))
def rate(p, nbytes, reps=6):
    best = 0
    for _ in range(reps):
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter(); hip.hipMemset(C.c_void_p(p), 0, nbytes); hip.hipDeviceSynchronize(); dt = time.perf_counter() - t0
        best = max(best, nbytes / dt / 1e12)
    return best
# (a) eight separate 8 GB hipMallocs
ps = [alloc(8 * GB) for _ in range(8)]
print("separate 8 GB buffers:", " ".join("%.2f" % rate(p, 8 * GB) for p in ps), "TB/s", flush=True)
# 1 GB windows of the first one
print("1 GB windows of buffer 0:", " ".join("%.2f" % rate(ps[0] + k * GB, GB) for k in range(8)), flush=True)
for p in ps: free(p)
# (b) one 64 GB segment, 8 GB windows
big = alloc(64 * GB)
print("8 GB windows of one 64 GB buffer:", " ".join("%.2f" % rate(big + k * 8 * GB, 8 * GB) for k in range(8)), "TB/s", flush=True)
print("1 GB windows of its first 8 GB:", " ".join("%.2f" % rate(big + k * GB, GB) for k in range(8)), flush=True)
print("256 MB windows:", " ".join("%.2f" % rate(big + k * (GB // 4), GB // 4) for k in range(16)), flush=True)
free(big)
