"""Multigrid preconditioner parameters around their defaults at BASELINE configs[2] (with the FP32 copies of the coarse operators the linear level is
cheaper than when the defaults were chosen): iterations and solve time.    python scripts/mg_param_sweep.py [grid]
End of round 4, one MI355X, 60^3: defaults (1 step per level, one cycle) 35 iterations / 128 ms; mg_steps_coarse 2 / 3 / 4 / 5: 34 / 31 / 29 / 29 iterations,
137 / 139 / 142 / 153 ms; mg_coarse_cycles 2: 26 iterations / 161 ms (27 / 149 with two steps); mg_steps_fine 2: 30 iterations / 191 ms -- the defaults stay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T.astype("int32"), V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.set_preconditioner(M.PRECOND_MULTIGRID)
defaults = dict(mg_steps_fine=1, mg_steps_coarse=1, mg_ratio_fine=0.3, mg_ratio_coarse=0.3, mg_coarse_cycles=1)       # mfh_ctx.hh
variants = [dict(), dict(mg_steps_coarse=2), dict(mg_steps_coarse=3), dict(mg_steps_coarse=4), dict(mg_coarse_cycles=2), dict(mg_coarse_cycles=2, mg_steps_coarse=2),
            dict(mg_ratio_coarse=0.1), dict(mg_ratio_coarse=0.5), dict(mg_ratio_fine=0.2), dict(mg_ratio_fine=0.4), dict(mg_steps_fine=2)]
for v in variants:
    for k, d in defaults.items():
        c.set_option(k, v.get(k, d))
    c.sim_solve(rtol=1e-8)
    c.sim_solve(rtol=1e-8)
    i = c.last_info
    print("%-50s %3d iterations, %7.2f ms (%.3f ms / iteration)" % (v or "defaults", i["iterations"], i["solve_ms"], i["solve_ms"] / max(1, i["iterations"])), flush=True)
