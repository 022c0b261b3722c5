"""Batched right-hand sides (NR = 2, 3, 6 interleaved vectors per operator pass) against one solve at a time.
    python scripts/batch_ab.py [grid]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.assemble()
f = c.neumann_load().ravel()
vars_, vals = c.bc_dirichlet_vars()
c.fix_variables(vars_, vals)
c.set_preconditioner(M.PRECOND_TWO_LEVEL)
rng = np.random.default_rng(0)
import os
for nr in [int(v) for v in os.environ.get("BATCH_NR", "1,2,3,6").split(",")]:
    F = np.stack([f * (1 + k) + 1e-3 * np.abs(f).max() * rng.standard_normal(len(f)) for k in range(nr)])
    for batch in [int(v) for v in os.environ.get("BATCH_MODES", "0,1").split(",")]:
        if nr == 1 and batch:
            continue
        c.set_option("batch_rhs", batch)
        for rep in range(2):
            t0 = time.time()
            U, infos = c.solve_batch(F, rtol=1e-8, maxit=20000)
            wall = time.time() - t0
        ms = sum(i["solve_ms"] for i in infos) if not batch else infos[0]["solve_ms"]
        its = [i["iterations"] for i in infos]
        print("nrhs %d batch %d: device %.1f ms (%.1f ms per rhs), wall %.3f s, iterations %s, batch sizes %s" %
              (nr, batch, ms, ms / nr, wall, its, [i["reserved"] for i in infos]), flush=True)
