"""Where the multigrid hierarchy setup goes at BASELINE configs[4]'s size (one context): laps of ensure_multigrid (MFH_MG_TIMING), of the
linear-level child context's mesh build (MFH_MESH_TIMING) and of the device block cache (MFH_POOL_TRACE=1 optional).
    python scripts/mg_setup_probe.py [grid] [reps]"""
import os
import sys
import time

os.environ.setdefault("MFH_MG_TIMING", "1")
os.environ.setdefault("MFH_MESH_TIMING", "1")
os.environ.setdefault("MFH_SOLVE_TIMING", "1")
os.environ.setdefault("MFH_SYM_TIMING", "1")
os.environ.setdefault("MFH_TL_TIMING", "1")
os.environ.setdefault("MFH_MFC_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 119
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
for rep in range(reps):
    c = M.Context(0)
    t0 = time.perf_counter()
    c.mesh_build(T, V, 2)
    ta = time.perf_counter()
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    tb = time.perf_counter()
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    tc = time.perf_counter()
    c.symbolic(False)
    td = time.perf_counter()
    c.assemble(); c.dev_sync()
    t1 = time.perf_counter()
    print("rep %d: build %.1f + dirichlet box %.1f + neumann box %.1f + symbolic %.1f + first assembly %.1f = %.1f ms"
          % (rep, 1e3 * (ta - t0), 1e3 * (tb - ta), 1e3 * (tc - tb), 1e3 * (td - tc), 1e3 * (t1 - td), 1e3 * (t1 - t0)), flush=True)
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    t0 = time.perf_counter()
    u = c.sim_solve(rtol=1e-8, maxit=2000)
    t1 = time.perf_counter()
    i, g = dict(c.last_info), c.multigrid_info()
    print("rep %d: sim_solve wall %.1f ms: hierarchy setup %.1f ms, solve %.1f ms (%d iterations), rest %.1f ms"
          % (rep, 1e3 * (t1 - t0), g["setup_ms"], i["solve_ms"], i["iterations"], 1e3 * (t1 - t0) - g["setup_ms"] - i["solve_ms"]), flush=True)
    t0 = time.perf_counter()
    u = c.sim_solve(rtol=1e-8, maxit=2000)
    t1 = time.perf_counter()
    print("rep %d: second sim_solve wall %.1f ms (solve %.1f ms)" % (rep, 1e3 * (t1 - t0), c.last_info["solve_ms"]), flush=True)
    c.close()
