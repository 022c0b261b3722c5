"""The CPU baseline's sparse direct solver (oracle/direct_solve.py: multifrontal Cholesky on a nested-dissection tree, the method of
the reference's CHOLMOD factorisation, SparseMatrices.hh:1984-2296) against scipy's SuperLU and against the oracle's own solve."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import direct_solve as DS
from oracle import meshfem_oracle as O
from meshfem_amd import grid


def _cantilever(n, deg):
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    sim = O.Simulator(T, V, deg)
    sim.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
    sim.apply_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0])
    sim.apply_neumann_box([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9], [0, -1, 0], "traction")
    K = sim.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper().tocsr()
    return sim, K, sim.neumannLoad().ravel()


def test_multifrontal_cholesky_matches_superlu_and_the_oracle_solve():
    sim, K, f = _cantilever(2, 2)
    pos = sim.mesh.node_pos
    free_nodes = np.flatnonzero(np.abs(pos[:, 0]) >= 1e-9)
    free = (3 * free_nodes[:, None] + np.arange(3)[None, :]).ravel()
    Kr = K[free][:, free]
    mf = DS.MultifrontalCholesky(Kr, pos[free_nodes], block=3, leaf=8).factor()
    assert len(mf.kids) > 7                                   # a real tree, not one dense front
    x = mf.solve(f[free])
    assert np.linalg.norm(Kr @ x - f[free]) <= 1e-12 * np.linalg.norm(f[free])
    x2 = spla.splu(Kr.tocsc()).solve(f[free])
    assert np.linalg.norm(x - x2) <= 1e-10 * np.linalg.norm(x2)
    u = np.zeros(K.shape[0]); u[free] = x
    assert np.linalg.norm(u - sim.solve().ravel()) <= 1e-10 * np.linalg.norm(u)


def test_multifrontal_cholesky_scalar_unknowns_and_every_variable_eliminated_once():
    rng = np.random.default_rng(0)
    n = 9
    idx = np.arange(n ** 3).reshape(n, n, n)
    rows, cols = [], []
    for ax in range(3):
        a, b = np.take(idx, range(n - 1), axis=ax).ravel(), np.take(idx, range(1, n), axis=ax).ravel()
        rows += [a, b]; cols += [b, a]
    A = sp.csr_matrix((-np.ones(sum(map(len, rows))), (np.concatenate(rows), np.concatenate(cols))), shape=(n ** 3, n ** 3))
    A = A + sp.diags(-np.asarray(A.sum(axis=1)).ravel() + 0.1)
    P = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(float)
    mf = DS.MultifrontalCholesky(A, P, block=1, leaf=20).factor()
    assert np.array_equal(np.sort(mf.perm), np.arange(n ** 3))
    b = rng.standard_normal(n ** 3)
    x = mf.solve(b)
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)
    assert mf.factor_nnz < 0.2 * (n ** 3) ** 2 / 2            # sparse factor: far below the dense triangle


def test_multifrontal_cholesky_subtree_parallel_factorisation_is_the_same_factor():
    """workers > 1: independent subtrees on threads with single-threaded BLAS, the fronts above them afterwards -- the same
    fronts in another order, so the factor and the solution agree to rounding."""
    sim, K, f = _cantilever(3, 1)
    pos = sim.mesh.node_pos
    free_nodes = np.flatnonzero(np.abs(pos[:, 0]) >= 1e-9)
    free = (3 * free_nodes[:, None] + np.arange(3)[None, :]).ravel()
    Kr = K[free][:, free]
    a = DS.MultifrontalCholesky(Kr, pos[free_nodes], block=3, leaf=4).factor()
    b = DS.MultifrontalCholesky(Kr, pos[free_nodes], block=3, leaf=4).factor(workers=3, blas_threads=2)
    assert b.subtrees >= 6 and a.subtrees == 1 and a.factor_nnz == b.factor_nnz and a.flops == b.flops
    for k in range(len(a.kids)):
        assert np.array_equal(a.bnd[k], b.bnd[k]) and np.allclose(np.tril(a.L11[k]), np.tril(b.L11[k]), rtol=1e-12, atol=1e-14)
        assert np.allclose(a.L21[k], b.L21[k], rtol=1e-11, atol=1e-13)
    xa, xb = a.solve(f[free]), b.solve(f[free])
    assert np.linalg.norm(xa - xb) <= 1e-12 * np.linalg.norm(xa)
    assert np.linalg.norm(Kr @ xb - f[free]) <= 1e-12 * np.linalg.norm(f[free])
