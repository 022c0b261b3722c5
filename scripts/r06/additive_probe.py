"""EXPERIMENT: additive (one operator application per iteration) against multiplicative V(1,1) on the quadratic level.  MFH_MG_ADDITIVE=<w> python scripts/r06/additive_probe.py [grid]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.assemble()
c.set_preconditioner(M.PRECOND_MULTIGRID)
u = c.sim_solve(rtol=1e-8, maxit=800)
for rep in range(2):
    u = c.sim_solve(rtol=1e-8, maxit=800)
    i = c.last_info
    print("MFH_MG_ADDITIVE=%s grid %d: %d iterations, solve %.1f ms (%.2f ms / iteration), true residual %.2e, max|u| %.10g"
          % (os.environ.get("MFH_MG_ADDITIVE", "-"), n, i["iterations"], i["solve_ms"], i["solve_ms"] / max(1, i["iterations"]), i["true_rel_residual"], np.abs(u).max()), flush=True)
