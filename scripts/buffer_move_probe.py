"""Does the assembly kernel's time follow WHERE individual buffers lie? One process: after every timing one buffer moves into a newly
allocated one (mfh_debug_move_buffer; the old memory is held, so the new buffer is other memory).   python scripts/buffer_move_probe.py [grid] [reserve]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
if len(sys.argv) > 2 and sys.argv[2] == "reserve":
    M.device_reserve(int(3.6e3 * 24 * n ** 3), 0, wait=True)
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1]); T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(200.0, 0.35); c.symbolic(False); c.assemble(); c.dev_sync()
names = ["K values", "gather codes", "gather slots", "element records", "column indices"]
def t():
    return min(c.time_assembly_kernel(M.ASSEMBLE_GATHER, 10) for _ in range(2))
print("start: %.3f ms" % t(), flush=True)
for which in (0, 0, 0, 1, 1, 1, 2, 2, 3, 3, 0, 0, 1, 1, 0, 1, 0, 1):
    c._ck(c.lib.mfh_debug_move_buffer(c.h, which))
    print("moved %-16s -> %.3f ms" % (names[which], t()), flush=True)
K0 = c.export_upper_triplets()[2]
print("checksum of K", float(np.abs(K0).sum()))
