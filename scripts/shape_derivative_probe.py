"""Workload for rocprofv3: the forward-mode shape-derivative kernels at the bench size (60^3 grid, P2 tets)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(200., 0.35)
rng = np.random.default_rng(0)
u = rng.normal(size=(c.n_node, 3)); du = rng.normal(size=(c.n_node, 3)); dp = rng.normal(size=V.shape) * 1e-3
w = rng.normal(size=(6, c.n_node, 3))
for name, fn in [("apply_delta_K", lambda: c.apply_delta_K(u, dp)),
                 ("delta_constant_strain_load", lambda: c.delta_constant_strain_load([1, 0, 0, 0, 0, 0.5], dp)),
                 ("delta_average_strain", lambda: c.delta_average_strain(u, du, dp)),
                 ("mutual_energies", lambda: c.mutual_energies(w)),
                 ("delta_mutual_energies", lambda: c.mutual_energies(w, dp))]:
    fn()
    t0 = time.time(); fn(); print("%-28s wall %.1f ms (host <-> device copies included)" % (name, 1e3 * (time.time() - t0)))
