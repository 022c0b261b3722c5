"""CPU direct-solve baseline (oracle/direct_solve.py) alone: time per phase, optional cProfile, BLAS thread count.
    python scripts/ds_probe.py [grid] [deg] [blas threads of the top fronts, 0 = default] [profile 0/1] [subtree workers, default 1]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from threadpoolctl import threadpool_limits

from oracle import c_oracle as CO, direct_solve as DS, meshfem_oracle as O
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prof = len(sys.argv) > 4 and sys.argv[4] == "1"
workers = int(sys.argv[5]) if len(sys.argv) > 5 else 1
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
h = M.Context(-1); h.mesh_build(T, V, deg); en, nn, pos = h.elem_nodes(), h.n_node, h.node_positions(); h.close()
D = O.ElasticityTensor.isotropic(3, 200.0, 0.35).D
Ap, Ai, Ax, t = CO.assemble_csc(3, deg, en, V, D, nn)
N = 3 * nn
U = sp.csc_matrix((Ax, Ai, Ap), shape=(N, N))
K = (U + sp.triu(U, 1).T).tocsr()
free_nodes = np.flatnonzero(np.abs(pos[:, 0]) >= 1e-9)
free = (3 * free_nodes[:, None] + np.arange(3)[None, :]).ravel()
Kr = K[free][:, free]
f = np.random.default_rng(0).standard_normal(len(free))


def run():
    mf = DS.MultifrontalCholesky(Kr, pos[free_nodes], 3)
    mf.factor(workers=workers, blas_threads=nth or None)
    x = mf.solve(f)
    print("dof %d, %d subtrees on %d workers, blas threads %d: order %.2fs, %d supernodes, factor %.2fs (subtrees %.2fs; %.1f GF/s, nnz(L) %.3g), solve %.2fs, residual %.1e"
          % (len(free), mf.subtrees, workers, DS.blas_threads(), mf.t_order, len(mf.kids), mf.t_factor, mf.t_subtrees, mf.flops / mf.t_factor / 1e9, mf.factor_nnz, mf.t_solve,
             np.linalg.norm(Kr @ x - f) / np.linalg.norm(f)), flush=True)


ctx = threadpool_limits(limits=nth, user_api="blas") if nth else None
if prof:
    cProfile.run("run()", "/tmp/ds_prof.out")
    pstats.Stats("/tmp/ds_prof.out").sort_stats("tottime").print_stats(12)
else:
    run()
