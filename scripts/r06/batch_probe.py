"""configs[3] operator, one vector at a time against NR = 6 interleaved vectors (block-Jacobi PCG, a fixed number of iterations: for a
rocprofv3 kernel trace).     python scripts/r06/batch_probe.py [grid] [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid, homogenization as H
from meshfem_amd.linear_elasticity import Simulator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
its = int(sys.argv[2]) if len(sys.argv) > 2 else 30
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
P = grid.synthetic_orthotropic_field(len(T), 3, 0)
sim = Simulator(np.ascontiguousarray(T, dtype=np.int32), V, 2, 0)
sim.setOrthotropicField(P)
sim.ctx.set_preconditioner(M.PRECOND_JACOBI)
sim.applyPeriodicConditions(1e-7)
sim.applyNoRigidMotionConstraint()
sim.setUsePinNoRigidTranslationConstraint(True)
F = np.stack([sim.constantStrainLoad(-H.canonical_strain_flat(3, k)).ravel() for k in range(6)])
c = sim.ctx
sim.maxit = 3
try:
    sim.solve(F[0])                 # sets the pin (fixed variables stay in the context)
except Exception as e:
    print("(first solve: %s)" % str(e)[:60])
for batch in (0, 1):
    c.set_option("batch_rhs", batch)
    for rep in range(2):
        t0 = time.time()
        try:
            U, infos = c.solve_batch(F, rtol=1e-30, maxit=its)
        except Exception as e:        # not converged by construction
            infos = None
            print("  (%s)" % str(e)[:80])
        print("batch %d rep %d: wall %.3f s %s" % (batch, rep, time.time() - t0, [round(i["solve_ms"], 1) for i in infos] if infos else ""), flush=True)
