"""How much do INDEPENDENT PCG solves gain from running concurrently on one GPU (separate contexts = separate HIP streams,
one host thread each)? Decides whether interleaving the six cell problems on several streams is worth building.
    python scripts/concurrent_probe.py [grid] [threads] [precond]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 3
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
ctxs, fs = [], []
for t in range(nt):
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.assemble()
    f = c.neumann_load().ravel() * (1 + t)
    vars_, vals = c.bc_dirichlet_vars()
    c.fix_variables(vars_, vals)
    c.set_preconditioner(pre)
    c.solve(f, rtol=1e-8)          # warm-up (lists, coarse setup)
    ctxs.append(c); fs.append(f)
t0 = time.time()
for c, f in zip(ctxs, fs):
    c.solve(f, rtol=1e-8)
seq = time.time() - t0
its = [c.last_info["iterations"] for c in ctxs]
dev = [c.last_info["solve_ms"] for c in ctxs]
def work(c, f):
    c.solve(f, rtol=1e-8)
t0 = time.time()
th = [threading.Thread(target=work, args=(c, f)) for c, f in zip(ctxs, fs)]
[t.start() for t in th]; [t.join() for t in th]
con = time.time() - t0
print("grid %d, %d solves, iterations %s: one after the other %.3f s (device %.0f ms each), concurrently %.3f s => %.2fx" % (n, nt, its, seq, dev[0], con, seq / con))
