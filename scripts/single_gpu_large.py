"""The largest problem one MI355X holds: BASELINE configs[4]'s 119^3 grid (40.4 M P2 tets, 172.9 M DOF -- sized for 8 GPUs) on ONE
device. Assembly kernel time, matrix-free operator, two-level PCG to 1e-8; writes gpurun_out/single_gpu_large_<n>.json.
    python scripts/single_gpu_large.py [grid]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 119
out = dict(grid=n)
t0 = time.time(); V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1]); out["mesh_gen_s"] = time.time() - t0
out["elements"] = int(len(T))
c = M.Context(0)
t0 = time.time(); c.mesh_build(T, V, 2); out["femmesh_build_s"] = time.time() - t0
del V, T
out.update(nodes=int(c.n_node), dof=int(3 * c.n_dof))
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
t0 = time.time(); c.symbolic(False); out["symbolic_s"] = time.time() - t0
c.assemble()
nr, nc, nnzb = c.matrix_info()
upper, stored = c.matrix_storage()
out.update(nnz_blocks=int(nnzb), stored_blocks=int(stored), matrix_storage="upper" if upper else "full", K_GB=stored * 72 / 1e9)
ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, 5)
out.update(assembly_kernel_ms=ms, elements_per_s=out["elements"] / ms * 1e3, alg_frac=(4316 if upper else 7736) * out["elements"] / ms / 1e6 / 8000.0)
print(json.dumps(out), flush=True)
out["matrix_free_operator_ms"] = c.time_spmv_kernel(10)
c.set_preconditioner(M.PRECOND_TWO_LEVEL)
t0 = time.time()
u = c.sim_solve(rtol=1e-8, maxit=20000)
out["solve_wall_s"] = time.time() - t0
i, p = c.last_info, c.precond_info()
out.update(iterations=i["iterations"], converged=bool(i["converged"]), solve_ms=i["solve_ms"], true_rel_residual=i["true_rel_residual"],
           ms_per_iteration=i["solve_ms"] / max(1, i["iterations"]), coarse_setup_ms=p["setup_ms"], aggregates=p["aggregates"],
           max_abs_u=float(np.abs(u).max()))
free, total = torch.cuda.mem_get_info(0)
out.update(device_memory_in_use_GB=(total - free) / 1e9, device_memory_total_GB=total / 1e9)
print(json.dumps(out), flush=True)
json.dump(out, open("gpurun_out/single_gpu_large_%d.json" % n, "w"), indent=1)
