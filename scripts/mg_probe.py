"""p-multigrid preconditioner: iteration counts and times over smoother settings, against the two-level preconditioner.
    python scripts/mg_probe.py [grid] [k0,k1,ratio0,ratio1 ...]        (MG_DEGREE=1: linear elements, BASELINE configs[1] at grid 55)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sets = [tuple(float(v) for v in a.split(",")) for a in sys.argv[2:]] or [(2, 4, 0.25, 0.05, 1)]
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, int(os.environ.get("MG_DEGREE", "2")))
print("%d elements, %d DoFs" % (len(T), 3 * c.n_dof), flush=True)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.set_preconditioner(M.PRECOND_TWO_LEVEL)
u_tl = c.sim_solve(rtol=1e-8)
i = c.last_info
print("two-level: %d iterations, %.1f ms, true residual %.2e, setup %.1f ms" % (i["iterations"], i["solve_ms"], i["true_rel_residual"], c.precond_info()["setup_ms"]), flush=True)
c.set_preconditioner(M.PRECOND_MULTIGRID)
for st in sets:
    k0, k1, r0, r1 = st[:4]
    cyc = st[4] if len(st) > 4 else 1
    c.set_option("mg_coarse_cycles", cyc)
    if len(st) > 5:
        c.set_option("mg_steps_agg", st[5])
    if len(st) > 6:
        c.set_option("mg_ratio_agg", st[6])
    c.set_option("mg_steps_fine", k0); c.set_option("mg_steps_coarse", k1); c.set_option("mg_ratio_fine", r0); c.set_option("mg_ratio_coarse", r1)
    for rep in range(2):
        t0 = time.time()
        try:
            u = c.sim_solve(rtol=1e-8, maxit=300)
        except M.MeshFEMHipError as e:
            print("  ", (k0, k1, r0, r1), "FAILED:", e, c.last_info, flush=True)
            break
        i = c.last_info
    else:
        print("multigrid k0=%d k1=%d ratio0=%.3f ratio1=%.3f cycles=%d: %d iterations, %.1f ms (%.2f ms / iteration), true residual %.2e, rel-L2 vs two-level %.1e, %s %s"
              % (k0, k1, r0, r1, cyc, i["iterations"], i["solve_ms"], i["solve_ms"] / max(1, i["iterations"]), i["true_rel_residual"],
                 np.linalg.norm(u - u_tl) / np.linalg.norm(u_tl), c.multigrid_info(), c.precond_info()), flush=True)
