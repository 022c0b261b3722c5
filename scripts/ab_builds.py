"""One sample of the headline measurements with ONE build of the library (A/B of builds on one box: scripts/ab_builds.sh alternates
the trees, every sample in a process of its own). The tree is a directory holding a `meshfem_amd` package with its own libmeshfem_hip.so
(the current tree, or an older commit exported under _ab_old/<name>/); only API that exists since round 3 is used.
    python scripts/ab_builds.py <tree dir> <label> [grid]      -> one JSON line"""
import json
import os
import sys
import time

tree, label = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
sys.path.insert(0, os.path.abspath(tree))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

assert os.path.abspath(M.__file__).startswith(os.path.abspath(tree)), M.__file__
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.symbolic(False)
c.set_option("reembed", 1)
c.assemble(); c.dev_sync()
for _ in range(5):
    c.assemble()
c.dev_sync()
steps = 20
passes = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps):
        c.assemble()
    c.dev_sync()
    passes.append((time.perf_counter() - t0) / steps * 1e3)
kern = [c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20) for _ in range(3)]
f = c.neumann_load().ravel()
vars_, vals = c.bc_dirichlet_vars()
c.fix_variables(vars_, vals)
c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
ops = [c.time_spmv_kernel(50) for _ in range(3)]
its = []
for rep in range(2):
    try:
        c.solve(f, rtol=1e-30, maxit=300)
    except M.MeshFEMHipError:
        pass
    its.append(c.last_info["solve_ms"] / 300)
c.set_preconditioner(M.PRECOND_MULTIGRID)
mg = []
for rep in range(3):
    u = c.solve(f, rtol=1e-8, maxit=500)
    mg.append(dict(iterations=c.last_info["iterations"], solve_ms=c.last_info["solve_ms"], setup_ms=c.last_info["setup_ms"]))
print(json.dumps(dict(label=label, grid=n, elements=int(c.n_elem), pass_ms=passes, kernel_ms=kern, operator_ms=ops, pcg_bj_iteration_ms=its,
                      multigrid=mg, lib=M.LIB_PATH if hasattr(M, "LIB_PATH") else None)), flush=True)
