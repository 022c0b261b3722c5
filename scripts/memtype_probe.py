"""The K values in memory of another type: hipExtMallocWithFlags fine-grained (0x1), uncached (0x3), contiguous (0x4) against plain hipMalloc;
the assembly kernel's time on each (hook mfh_debug_adopt_vals), K checked by a checksum.   python scripts/memtype_probe.py [grid]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1]); T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(200.0, 0.35); c.symbolic(False); c.assemble(); c.dev_sync()
nbytes = c.matrix_storage()[1] * 72 + (4 << 20)
t = lambda: min(c.time_assembly_kernel(M.ASSEMBLE_GATHER, 10) for _ in range(2))
ref = float(np.abs(c.export_upper_triplets()[2]).sum())
print("arena buffer: %.3f ms, checksum %.6f" % (t(), ref), flush=True)
for label, flag in (("plain hipMalloc", None), ("fine-grained", 1), ("uncached", 3), ("contiguous", 4), ("plain hipMalloc", None), ("fine-grained", 1), ("uncached", 3)):
    p = C.c_void_p()
    e = hip.hipMalloc(C.byref(p), nbytes) if flag is None else hip.hipExtMallocWithFlags(C.byref(p), nbytes, flag)
    if e:
        print(label, "allocation failed", e); continue
    c._ck(c.lib.mfh_debug_adopt_vals(c.h, p))
    c.assemble(); c.dev_sync()
    ms = t()
    chk = float(np.abs(c.export_upper_triplets()[2]).sum())
    print("%-16s %.3f ms   checksum difference %.1e" % (label, ms, abs(chk - ref) / ref), flush=True)
