"""A/B of the PCG variants and of batched right-hand sides on one MI355X (config 3 / config 4 shapes).
    python scripts/pcg_ab.py [grid] [out.json]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out_path = sys.argv[2] if len(sys.argv) > 2 else None
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.assemble()
res = dict(grid=n, elements=int(c.n_elem), dof=int(3 * c.n_dof))
f = c.neumann_load().ravel()
vars_, vals = c.bc_dirichlet_vars()
c.fix_variables(vars_, vals)
for pre, pname in ((M.PRECOND_BLOCK_JACOBI, "block_jacobi"), (M.PRECOND_TWO_LEVEL, "two_level")):
    c.set_preconditioner(pre)
    for variant in (0, 1):
        c.set_option("pcg_variant", variant)
        u = c.solve(f, rtol=1e-8, maxit=20000)
        u = c.solve(f, rtol=1e-8, maxit=20000)
        i = c.last_info
        res["%s_variant%d" % (pname, variant)] = dict(iterations=i["iterations"], solve_ms=i["solve_ms"], ms_per_it=i["solve_ms"] / max(1, i["iterations"]),
                                                      true_rel_residual=i["true_rel_residual"], graph=i["used_graph"])
        print(pname, variant, res["%s_variant%d" % (pname, variant)], flush=True)
    # six right-hand sides: batched vs one at a time
    rng = np.random.default_rng(0)
    F = np.stack([f * (1 + k) + 1e-3 * np.abs(f).max() * rng.standard_normal(len(f)) for k in range(6)])
    c.set_option("pcg_variant", 1)
    for batch in (1, 0):
        c.set_option("batch_rhs", batch)
        t0 = time.time()
        U, infos = c.solve_batch(F, rtol=1e-8, maxit=20000)
        wall = time.time() - t0
        ms = sum(i["solve_ms"] for i in infos) if not batch else infos[0]["solve_ms"]
        res["%s_six_rhs_batch%d" % (pname, batch)] = dict(iterations=[i["iterations"] for i in infos], device_ms=ms, wall_s=wall,
                                                           sizes=[i["reserved"] for i in infos])
        print(pname, "six rhs batch", batch, res["%s_six_rhs_batch%d" % (pname, batch)], flush=True)
    c.set_option("batch_rhs", 1)
print(json.dumps(res))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
