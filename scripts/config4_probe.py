"""Where the time of BASELINE configs[3] (periodic homogenization, per-element orthotropic field) goes: per-solve device
time, per-iteration time, operator mode, host-side phases."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid, homogenization as H
from meshfem_amd.linear_elasticity import Simulator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
t0 = time.time()
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
rng = np.random.default_rng(0)
nE = len(T)
P = grid.synthetic_orthotropic_field(nE, 3, 0)   # seed 0; the one non-PD draw in 2 M is repaired
t1 = time.time(); print("mesh + field generation %.2f s" % (t1 - t0))
sim = Simulator(T, V, 2); t2 = time.time(); print("Simulator (mesh_build) %.2f s" % (t2 - t1))
sim.setOrthotropicField(P); sim.ctx.set_preconditioner(M.PRECOND_TWO_LEVEL)
sim.applyPeriodicConditions(); t3 = time.time(); print("material + periodic conditions %.2f s" % (t3 - t2))
sim.applyNoRigidMotionConstraint(); sim.setUsePinNoRigidTranslationConstraint(True)
for k in range(6):
    ta = time.time()
    rhs = sim.constantStrainLoad(-H.canonical_strain_flat(3, k)); tb = time.time()
    w = sim.solve(rhs); tc = time.time()
    i = sim.info
    print("cell problem %d: load %.3f s, solve wall %.3f s, device solve %.1f ms, %d iterations, %.3f ms/it, graph %s" %
          (k, tb - ta, tc - tb, i["solve_ms"], i["iterations"], i["solve_ms"] / max(i["iterations"], 1), i.get("used_graph")))
print("matrix_free_info", sim.ctx.matrix_free_info())
print("precond", sim.ctx.precond_info(), "timing", sim.ctx.timing())
print("operator kernel ms", sim.ctx.time_spmv_kernel(10))
print("total %.2f s" % (time.time() - t0))
