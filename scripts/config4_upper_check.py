"""BASELINE configs[3] (periodic homogenization, orthotropic field, two-level PCG) on the upper-triangle storage against the
default storage: homogenized tensors and iteration counts must agree.   python scripts/config4_upper_check.py [grid]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid, homogenization as H
from meshfem_amd.linear_elasticity import Simulator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
P = grid.synthetic_orthotropic_field(len(T), 3, 0)
res = {}
for storage in (0, 1):
    sim = Simulator(T, V, 2)
    sim.ctx.set_option("matrix_storage", storage)
    sim.setOrthotropicField(P)
    sim.ctx.set_preconditioner(M.PRECOND_TWO_LEVEL)
    sim.applyPeriodicConditions()
    sim.applyNoRigidMotionConstraint(); sim.setUsePinNoRigidTranslationConstraint(True)
    t0 = time.time()
    its, ws = [], []
    for k in range(6):
        w = sim.solve(sim.constantStrainLoad(-H.canonical_strain_flat(3, k)))
        its.append(sim.info["iterations"]); ws.append(w)
    res[storage] = (its, ws, sim.ctx.timing()["assemble_ms"], sim.ctx.matrix_info()[2], time.time() - t0)
    print("storage %d: iterations %s, assembly %.2f ms, stored blocks %d, six solves %.2f s" % (storage, its, res[storage][2], res[storage][3], res[storage][4]), flush=True)
d = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(res[1][1], res[0][1]))
print("max rel-L2 difference of the fluctuation displacements: %.2e" % d)
