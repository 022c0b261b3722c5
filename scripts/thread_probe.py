"""Two host threads, one context each, at the same time (ctypes releases the GIL): mesh build, assembly, hierarchy, solves -- the results must be
those of the same work done one after the other.     python scripts/thread_probe.py"""
import sys, os, threading, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
from meshfem_amd import grid

def work(n, deg, pre, out, key):
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    c = M.Context(0)
    c.mesh_build(T, V, deg)
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.set_preconditioner(pre)
    us = [c.sim_solve(rtol=1e-9) for _ in range(3)]
    i, j, v = c.export_upper_triplets()
    out[key] = (us[-1], c.last_info["iterations"], float(np.abs(v).sum()))
    c.close()

jobs = [(14, 2, M.PRECOND_MULTIGRID), (11, 2, M.PRECOND_TWO_LEVEL), (17, 1, M.PRECOND_MULTIGRID), (12, 2, M.PRECOND_BLOCK_JACOBI)]
seq, par = {}, {}
for k, j in enumerate(jobs):
    work(*j, seq, k)
for rep in range(3):
    th = [threading.Thread(target=work, args=(*j, par, k)) for k, j in enumerate(jobs)]
    [t.start() for t in th]; [t.join() for t in th]
    for k in range(len(jobs)):
        du = np.linalg.norm(par[k][0] - seq[k][0]) / np.linalg.norm(seq[k][0])
        print("rep %d job %d: iterations %d vs %d, |K| %.12e vs %.12e, rel diff of u %.1e" % (rep, k, par[k][1], seq[k][1], par[k][2], seq[k][2], du), flush=True)
        assert du < 1e-7 and abs(par[k][2] - seq[k][2]) <= 1e-12 * seq[k][2]
print("OK")
