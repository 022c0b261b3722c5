"""Contexts are independent: several host threads, one context each, running mesh build / assembly / hierarchy / solves at the same time
(the reference's Simulator objects are used that way from TBB tasks) give what the same work gives one after the other."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _work(n, deg, pre, out, key):
    import meshfem_amd as M
    from meshfem_amd import grid
    try:
        V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
        c = M.Context(0)
        c.mesh_build(T, V, deg)
        c.material_isotropic(200.0, 0.35)
        c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
        c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
        c.set_preconditioner(pre)
        u = [c.sim_solve(rtol=1e-9) for _ in range(2)][-1]
        _, _, v = c.export_upper_triplets()
        out[key] = (u, c.last_info["iterations"], float(np.abs(v).sum()))
        c.close()
    except Exception as e:   # noqa: BLE001 -- reported by the asserting thread
        out[key] = e


def test_contexts_on_concurrent_host_threads():
    import meshfem_amd as M
    jobs = [(10, 2, M.PRECOND_MULTIGRID), (8, 2, M.PRECOND_TWO_LEVEL), (12, 1, M.PRECOND_MULTIGRID), (9, 2, M.PRECOND_BLOCK_JACOBI)]
    seq, par = {}, {}
    for k, j in enumerate(jobs):
        _work(*j, seq, k)
        assert not isinstance(seq[k], Exception), seq[k]
    for _ in range(2):
        th = [threading.Thread(target=_work, args=(*j, par, k)) for k, j in enumerate(jobs)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k in range(len(jobs)):
            assert not isinstance(par[k], Exception), par[k]
            assert par[k][1] == seq[k][1]
            assert abs(par[k][2] - seq[k][2]) <= 1e-12 * seq[k][2]
            assert np.linalg.norm(par[k][0] - seq[k][0]) <= 1e-8 * np.linalg.norm(seq[k][0])
