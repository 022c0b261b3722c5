"""What about the memory of the K values sets the assembly kernel's time? The values buffer is replaced by buffers built with the HIP
virtual-memory API (scripts/probe/vmm_alloc.cpp): physical chunks of 2 MiB ... the whole buffer, mapped in order or shuffled, address range
aligned to 2 MiB or 1 GiB.   python scripts/vmm_probe.py [grid]"""
import ctypes as C, sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
so = "/tmp/vmm_alloc.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "scripts", "probe", "vmm_alloc.cpp")])
vmm = C.CDLL(so)
vmm.vmm_alloc.restype = C.c_void_p
vmm.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1]); T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(200.0, 0.35); c.symbolic(False); c.assemble(); c.dev_sync()
nbytes = c.matrix_storage()[1] * 72 + (4 << 20)
def t():
    return min(c.time_assembly_kernel(M.ASSEMBLE_GATHER, 10) for _ in range(2))
print("arena buffer (one hipMalloc): %.3f ms" % t(), flush=True)
MB, GB = 1 << 20, 1 << 30
for label, chunk, align, shuffle in (("whole buffer, VA 2 MiB", 0, 2 * MB, 0), ("whole buffer, VA 1 GiB", 0, GB, 0), ("2 MiB chunks in order", 2 * MB, 2 * MB, 0),
                                     ("2 MiB chunks shuffled", 2 * MB, 2 * MB, 1), ("64 MiB chunks in order", 64 * MB, 64 * MB, 0), ("64 MiB chunks shuffled", 64 * MB, 64 * MB, 1),
                                     ("1 GiB chunks, VA 1 GiB", GB, GB, 0), ("whole buffer, VA 2 MiB", 0, 2 * MB, 0), ("2 MiB chunks shuffled", 2 * MB, 2 * MB, 1),
                                     ("whole buffer, VA 1 GiB", 0, GB, 0), ("2 MiB chunks in order", 2 * MB, 2 * MB, 0)):
    p = vmm.vmm_alloc(nbytes, chunk, align, shuffle, 0)
    if not p:
        print(label, "allocation failed"); continue
    c._ck(c.lib.mfh_debug_adopt_vals(c.h, C.c_void_p(p)))
    c.assemble()
    print("%-28s %.3f ms" % (label, t()), flush=True)
v = c.export_upper_triplets()[2]
print("checksum of K", float(np.abs(v).sum()))
