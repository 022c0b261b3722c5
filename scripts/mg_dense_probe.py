"""Size of the level that is inverted densely (option mg_dense_max): hierarchy setup, iterations and solve time.  python scripts/mg_dense_probe.py [grid] [values...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
vals = [int(v) for v in sys.argv[2:]] or [1200, 400, 100]
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.assemble()
for rep in range(2):
    for v in vals:
        c.set_option("mg_dense_max", v)
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        t0 = time.perf_counter()
        u = c.sim_solve(rtol=1e-8, maxit=500)
        wall = time.perf_counter() - t0
        u2 = c.sim_solve(rtol=1e-8, maxit=500)
        print("mg_dense_max %5d: first solve wall %.1f ms (hierarchy %.1f ms), %d iterations, solve %.1f ms; levels %s; max|u| %.10g"
              % (v, wall * 1e3, c.multigrid_info()["setup_ms"], c.last_info["iterations"], c.last_info["solve_ms"], [L["aggregates"] for L in c.multigrid_levels()], np.abs(u).max()), flush=True)
