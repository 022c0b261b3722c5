"""Workload for rocprofv3: assembled SpMV vs the matrix-free operator (two-pass) at the bench size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(200., 0.35); c.assemble()
c.set_option("matrix_free", 0)
print("assembled ms", c.time_spmv_kernel(10))
c.set_option("matrix_free", 1)
for mode in (1, 2, 3, 4):
    c.set_option("matrix_free_mode", mode)
    print("matrix-free mode", mode, "ms", c.time_spmv_kernel(10))
