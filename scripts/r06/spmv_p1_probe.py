"""Block-CSR SpMV of a LINEAR-element K the size of configs[2]'s linear multigrid level, with the matrix in the memory-side cache and out of it
(run under rocprofv3 --kernel-trace; the k_spmv durations are read from the trace).  python scripts/r06/spmv_p1_probe.py [grid]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
big = torch.empty(1 << 28, dtype=torch.float64, device="cuda:0")      # 2 GiB: a fill of it empties the 256 MiB memory-side cache
for slots in (0, 256, 1024, 2048):
    c = M.Context(0)
    c.mesh_build(T, V, 1)
    c.material_isotropic(200.0, 0.35)
    c.set_option("matrix_free", 0)
    if slots: c.set_option("chunk_slots", slots)
    c.assemble()
    u = np.random.default_rng(0).standard_normal(c.n_dof * 3)
    print("chunk_slots", slots or "default", "rows", c.n_dof, flush=True)
    for rep in range(3): c.apply_K(u)            # matrix resident in the memory-side cache
    for rep in range(3):
        big.fill_(float(rep)); torch.cuda.synchronize()
        c.apply_K(u)                             # matrix from HBM
    big.fill_(-1.0); torch.cuda.synchronize()    # separator in the trace
    del c
