"""Same-process A/B of one operator option (default: mf_stage_positions 0 / 1): operator alone, block-Jacobi PCG iteration, multigrid solve.
    python scripts/mf_stage_ab.py [grid] [option] [values ...]"""
import json
import sys

sys.path.insert(0, ".")
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
opt = sys.argv[2] if len(sys.argv) > 2 else "mf_stage_positions"
vals = [float(v) for v in sys.argv[3:]] or [0, 1]
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.assemble()
res = {v: dict(op=[], bj=[], mg=[], it=[], umax=[]) for v in vals}
for rep in range(3):
    for v in vals:
        c.set_option(opt, v)
        c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
        try:
            c.sim_solve(rtol=1e-30, maxit=200)
        except M.MeshFEMHipError:
            pass
        res[v]["bj"].append(c.last_info["solve_ms"] / 200)
        res[v]["op"].append(min(c.time_spmv_kernel(50) for _ in range(2)))
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u = c.sim_solve(rtol=1e-8, maxit=500)
        res[v]["mg"].append(c.last_info["solve_ms"]); res[v]["it"].append(c.last_info["iterations"]); res[v]["umax"].append(float(np.abs(u).max()))
for v in vals:
    r = res[v]
    print("%s = %g: operator %s ms | PCG block-Jacobi iteration %s ms | multigrid solve %s ms (%s iterations), max|u| %.12g"
          % (opt, v, " ".join("%.4f" % x for x in r["op"]), " ".join("%.4f" % x for x in r["bj"]), " ".join("%.1f" % x for x in r["mg"]), r["it"][0], r["umax"][0]), flush=True)
print(json.dumps({str(k): v for k, v in res.items()}))
