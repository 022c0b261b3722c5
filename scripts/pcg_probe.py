"""On-GPU probe of PCG variants: iterations / time vs preconditioner and aggregate size."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
aggs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200, 0.35)
c.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9], [0, -1, 0])
c.assemble()
u0 = c.sim_solve(rtol=1e-8)
i0 = dict(c.last_info)
print("block-jacobi", json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in i0.items()}), flush=True)
c.set_preconditioner(M.PRECOND_TWO_LEVEL)
for a in aggs:
    c.set_option("agg_nodes", a)
    t = time.time()
    u = c.sim_solve(rtol=1e-8)
    wall = time.time() - t
    i1 = dict(c.last_info)
    err = np.linalg.norm(u - u0) / np.linalg.norm(u0)
    print("two-level agg_nodes", a, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in i1.items()}),
          json.dumps(c.precond_info()), "wall %.2f" % wall, "rel diff vs BJ %.2e" % err, flush=True)
