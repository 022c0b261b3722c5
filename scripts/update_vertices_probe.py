"""Cost of one shape-optimisation step at the bench size: vertex update on the same connectivity versus a rebuild."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
rng = np.random.default_rng(0)
dV = rng.uniform(-1, 1, size=V.shape) * (0.02 / n)
dV[(V[:, 0] < 1e-12) | (V[:, 0] > 1 - 1e-12)] = 0.0          # keep the loaded / clamped faces where the box regions look for them
def setup(c):
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.set_preconditioner(M.PRECOND_TWO_LEVEL)
c = M.Context(0)
t0 = time.time(); c.mesh_build(T, V, 2); setup(c); c.assemble(); c.dev_sync(); t1 = time.time()
print("first build + assemble (includes HIP start-up): %.3f s" % (t1 - t0))
u = c.sim_solve(); print("first solve wall %.3f s (%d iterations)" % (time.time() - t1, c.last_info["iterations"]))
for k in range(2):
    t0 = time.time(); c.mesh_update_vertices(V + (k + 1) * dV); c.assemble(); c.dev_sync(); t1 = time.time()
    u = c.sim_solve(); t2 = time.time()
    print("update step %d: vertices + re-embed + assemble %.3f s, solve wall %.3f s (%d iterations)" % (k, t1 - t0, t2 - t1, c.last_info["iterations"]))
c.close()
t0 = time.time(); c2 = M.Context(0); c2.mesh_build(T, V + 2 * dV, 2); setup(c2); c2.assemble(); c2.dev_sync(); t1 = time.time()
u2 = c2.sim_solve(); t2 = time.time()
print("rebuild from scratch: mesh + symbolic + assemble %.3f s, solve wall %.3f s" % (t1 - t0, t2 - t1))
print("solutions agree to %.2e" % (np.linalg.norm(u - u2) / np.linalg.norm(u2)))
