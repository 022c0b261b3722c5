"""Assembly kernel time against HOW the arena got its memory from the driver: nothing in advance (one hipMalloc per large buffer), one
20 GB segment, several segments. One process per variant (argument).   python scripts/segment_probe.py <variant> [grid]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
variant = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
GB = 1 << 30
plans = {"none": [], "one20": [20], "8+8+4": [8, 8, 4], "10x2+8": [2] * 10 + [8], "40x0.5+8": [0.5] * 40 + [8], "one8+rest": [8]}
keep = M.Context(0)
def alloc(nbytes):
    p = C.c_void_p(); keep._ck(keep.lib.mfh_debug_arena_alloc(keep.h, int(nbytes), C.byref(p))); return p.value
ps = [alloc(int(g * GB)) for g in plans[variant]]
for p in ps:
    keep._ck(keep.lib.mfh_debug_arena_free(keep.h, C.c_void_p(p)))
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1]); T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(200.0, 0.35); c.symbolic(False); c.assemble(); c.dev_sync()
k = [c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20) for _ in range(3)]
a = M.device_arena_stats(0)
print("%-10s kernel %.3f %.3f %.3f ms  segments %d held %.1f GB" % (variant, *k, a["segments"], a["held_bytes"] / 1e9), flush=True)
