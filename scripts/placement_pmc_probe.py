"""One counter, one process: the K values move eight times (hook mfh_debug_move_buffer), three assembly launches after every move; the script prints the
kernel time of every group, rocprofv3 --pmc <counter> records the counter of every dispatch in the same order.
    rocprofv3 --pmc TCC_TAG_STALL_sum --output-format csv -d <dir> -- python scripts/placement_pmc_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, meshfem_amd as M
from meshfem_amd import grid
n = 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1]); T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(200.0, 0.35); c.symbolic(False); c.assemble(); c.dev_sync()
for k in range(9):
    if k:
        c._ck(c.lib.mfh_debug_move_buffer(c.h, 0))
    print("group %d kernel_ms %.3f" % (k, c.time_assembly_kernel(M.ASSEMBLE_GATHER, 3)), flush=True)
