"""Option mg_coarse_fp32 (the multigrid preconditioner reads the linear level's K and the aggregate stencils from FP32 copies) against full FP64
storage, same process, alternating: iterations, solve time, difference of the solutions.    python scripts/mg_fp32_ab.py [grid] [degree] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T.astype("int32"), V, deg)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.set_preconditioner(M.PRECOND_MULTIGRID)
sol = {}
for rep in range(reps):
    for fp32 in (1, 0):
        c.set_option("mg_coarse_fp32", fp32)
        u = c.sim_solve(rtol=1e-8)          # builds the hierarchy of this setting
        u = c.sim_solve(rtol=1e-8)
        i, g = dict(c.last_info), c.multigrid_info()
        sol[fp32] = u
        print("rep %d mg_coarse_fp32=%d: %d iterations, solve %.2f ms (%.3f ms / iteration), true residual %.2e, hierarchy setup %.1f ms"
              % (rep, fp32, i["iterations"], i["solve_ms"], i["solve_ms"] / max(1, i["iterations"]), i["true_rel_residual"], g["setup_ms"]), flush=True)
print("rel-L2 difference of the two solutions: %.2e" % (np.linalg.norm(sol[1] - sol[0]) / np.linalg.norm(sol[0])))
