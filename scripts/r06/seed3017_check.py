"""Which side of free-body seed 3017 is off? KKT residuals of the library's solution and of the oracle's sparse-LU solution, both measured with the ORACLE's K and constraint rows."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import meshfem_amd as M
from oracle import meshfem_oracle as O
from fuzz_unstructured_util import random_mesh
from scipy.spatial import Delaunay
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3017
rng, dim, deg, E, V, mat = random_mesh(seed)
if len(E) > 9000:
    keep = rng.random(len(V)) < 0.35
    P = V[keep]; E = Delaunay(P).simplices.astype(np.int32)
    vol = np.linalg.det(P[E[:, 1:]] - P[E[:, :1]]); fl = vol < 0
    E[fl, 0], E[fl, 1] = E[fl, 1].copy(), E[fl, 0].copy()
    E = np.ascontiguousarray(E[np.abs(vol) > 1e-6 * np.abs(vol).mean()]); used = np.unique(E)
    rm = -np.ones(len(P), np.int64); rm[used] = np.arange(len(used)); E = rm[E].astype(np.int32); V = np.ascontiguousarray(P[used])
pin = bool(seed % 2)
ref = O.Simulator(E, V, deg)
ref.set_material_constant(O.ElasticityTensor.isotropic(dim, 200.0, 0.35))
sim = M.Simulator(E, V, deg)
sim.setIsotropicMaterial(200.0, 0.35)
sim.rtol = 1e-10
f = rng.standard_normal((sim.ctx.n_node, dim))
sim.applyNoRigidMotionConstraint(); sim.setUsePinNoRigidTranslationConstraint(pin)
sim.ctx.set_preconditioner(M.PRECOND_MULTIGRID)
u_gpu = sim.solve(f).ravel()
u_ref = O.solve_constrained(ref, f=f.ravel(), use_pin=pin, no_rigid_motion=True).ravel()
K = ref.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper().tocsr()
C = O.rotation_rows(ref)
m = ref.mesh
interior = np.flatnonzero(~m.is_bdry_node)
node = int(interior[0]) if len(interior) else 0
fv = [dim * ref.DoF(node) + c for c in range(dim)] if pin else []
if not pin: C = np.vstack([C, O.translation_rows(ref, list(range(dim)))])
free = np.ones(K.shape[0], bool); free[fv] = False
Cf = C[:, free]
Q, _ = np.linalg.qr(Cf.T)                 # orthonormal basis of the constraint rows' span on the free variables
for name, u in (("library (multigrid PCG)", u_gpu), ("oracle (sparse LU of the KKT system)", u_ref)):
    r = (f.ravel() - K @ u)[free]
    rperp = r - Q @ (Q.T @ r)             # K u + C^T lambda = f  <=>  the residual lies in span(C^T)
    print("%-40s |C u| %.2e   |pinned u| %.2e   residual outside span(C^T) / |f| %.2e" % (name, np.abs(C @ u).max(), np.abs(u[fv]).max() if fv else 0.0, np.linalg.norm(rperp) / np.linalg.norm(f.ravel()[free])))
print("difference of the two solutions / |u|: %.2e" % (np.linalg.norm(u_gpu - u_ref) / np.linalg.norm(u_ref)))
