"""Linear tets, matrix-free cluster operator: sweep of the block size.   python scripts/p1_block_sweep.py [grid] [sizes...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
for be in [int(a) for a in sys.argv[2:]] or (256, 512, 1024, 2048, 4096):
    c = M.Context(0); c.mesh_build(T, V, 1); c.material_isotropic(200., 0.35)
    c.set_option("mf_block_elems", be); c.set_option("matrix_free", 1); c.assemble()
    ms = min(c.time_spmv_kernel(20) for _ in range(3))
    print(n, be, "ms %.4f" % ms, c.matrix_free_info(), flush=True)
    c.close()
