"""Matrix-free operator per material flavour (isotropic constant / isotropic field / orthotropic field / general tensor field), 44^3 grid -> 2.04 M quadratic tets.
    python scripts/r06/op_materials.py [grid]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
nE = len(T)
P = grid.synthetic_orthotropic_field(nE, 3, 0)
c0 = M.Context(0)
c0.mesh_build(T, V, 2)
c0.material_ortho_field(P)
D = np.stack([c0.material_get(e) for e in range(0, nE, max(1, nE // 4096))])
c0.close()
Dfull = np.repeat(D, int(np.ceil(nE / len(D))), axis=0)[:nE].copy()
rng = np.random.default_rng(0)
for name, setup in (("isotropic constant", lambda c: c.material_isotropic(200.0, 0.3)),
                    ("isotropic field", lambda c: c.material_iso_field(rng.uniform(100, 300, nE), rng.uniform(0.2, 0.35, nE))),
                    ("orthotropic field", lambda c: c.material_ortho_field(P)),
                    ("general tensor field", lambda c: c.material_tensor_field(Dfull))):
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    setup(c)
    c.assemble()
    t = [c.time_spmv_kernel(20) for _ in range(3)]
    print("%-22s operator %s ms" % (name, " ".join("%.4f" % v for v in t)), flush=True)
    c.close()
