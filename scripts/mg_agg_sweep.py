"""Smoothing steps and first-level size of the AGGREGATE levels of the multigrid hierarchy (the part that is replicated on every rank of a partitioned
solve, so what bounds its strong scaling): iterations and solve time in one context.    python scripts/mg_agg_sweep.py [grid]"""
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
for steps, target in [(2, 32), (1, 32), (3, 32), (2, 64), (1, 64), (2, 128), (1, 128)]:
    c.set_option("mg_steps_agg", steps)
    c.set_option("mg_agg_target", target)
    c.sim_solve(rtol=1e-8)
    c.sim_solve(rtol=1e-8)
    i, p = c.last_info, c.precond_info()
    print("mg_steps_agg %d mg_agg_target %3d: %3d iterations, %8.2f ms (%.3f ms / iteration), first aggregate level %s" % (steps, target, i["iterations"], i["solve_ms"], i["solve_ms"] / max(1, i["iterations"]), p.get("aggregates")), flush=True)
