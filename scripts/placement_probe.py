"""Does the assembly kernel's time depend on WHERE its arrays lie relative to each other? One process, one reserved arena segment; before every
context a padding buffer of d x 2 MiB shifts everything the context allocates. If the time follows d the virtual layout matters; if it does not (and still
differs between processes) it is the physical placement.        python scripts/placement_probe.py [grid]"""
import ctypes as C
import sys
sys.path.insert(0, ".")
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
M.device_reserve(64 << 30, 0, wait=True)
keep = M.Context(0)


def alloc(nbytes):
    p = C.c_void_p()
    keep._ck(keep.lib.mfh_debug_arena_alloc(keep.h, int(nbytes), C.byref(p)))
    return p.value


for d in (0, 1, 2, 3, 4, 7, 8, 16, 31, 32, 64, 100, 128, 256, 512, 1024, 0, 1):
    pad = alloc(d * (2 << 20)) if d else None
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200.0, 0.35)
    c.symbolic(False)
    c.assemble(); c.dev_sync()
    t = [c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20) for _ in range(2)]
    st = M.device_arena_stats(0)
    print("pad %5d x 2 MiB at %s: kernel %.3f %.3f ms   (arena: %d segments, held %.1f GB)" % (d, hex(pad) if pad else "-", t[0], t[1], st["segments"], st["held_bytes"] / 1e9), flush=True)
    c.close()
    if pad:
        keep._ck(keep.lib.mfh_debug_arena_free(keep.h, C.c_void_p(pad)))
