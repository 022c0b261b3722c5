"""Assembly kernel time per material flavour (isotropic constant, orthotropic field, full tensor field).
    python scripts/asm_materials.py [grid] [deg]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
rng = np.random.default_rng(0)
nE = len(T)
c = M.Context(0)
c.mesh_build(T, V, deg)
c.symbolic(False)
for name in ("iso", "ortho_field", "tensor_field"):
    if name == "iso":
        c.material_isotropic(200.0, 0.35)
    elif name == "ortho_field":
        p = np.empty((nE, 9))
        p[:, 0:3] = rng.uniform(100, 300, (nE, 3))
        p[:, 3:6] = rng.uniform(0.1, 0.3, (nE, 3))
        p[:, 6:9] = rng.uniform(40, 120, (nE, 3))
        c.material_ortho_field(p)
    else:
        A = rng.standard_normal((nE, 6, 6))
        c.material_tensor_field(np.einsum("eij,ekj->eik", A, A) + 6 * np.eye(6))
    c.assemble()
    t = [c.time_assembly_kernel(M.ASSEMBLE_GATHER, 10) for _ in range(3)]
    print("%-13s %s ms" % (name, " ".join("%.3f" % x for x in t)), flush=True)
