"""A free body under the rigid-motion constraint rows (`no_rigid_motion` in .bc files): K singular on the free variables, one consistent singular
PCG solve -- block-Jacobi against the multigrid hierarchy with its pinned dense level.      python scripts/free_body_probe.py [grid]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
sim = M.Simulator(T, V, 2)
sim.setIsotropicMaterial(200.0, 0.35)
sim.rtol = 1e-8
big = 1e9
sim.applyNeumannBox([1 - 1e-9, -big, -big], [1 + 1e-9, big, big], [1.0, 0.3, 0.0])
sim.applyNeumannBox([-1e-9, -big, -big], [1e-9, big, big], [-1.0, 0.1, 0.0])
sim.applyNoRigidMotionConstraint()
res = {}
for name, pc in (("multigrid", M.PRECOND_MULTIGRID), ("block_jacobi", M.PRECOND_BLOCK_JACOBI)):
    sim.ctx.set_preconditioner(pc)
    for rep in range(2):
        t0 = time.time()
        u = sim.solve()
        wall = time.time() - t0
    res[name] = u
    print("%s: %d elements, %d iterations, solve %.1f ms, wall %.2f s, note '%s'" % (name, len(T), sim.info["iterations"], sim.info["solve_ms"], wall,
                                                                                  sim.ctx.precond_info()["note"]), flush=True)
print("rel-L2 difference %.2e" % (np.linalg.norm(res["multigrid"] - res["block_jacobi"]) / np.linalg.norm(res["block_jacobi"])))
