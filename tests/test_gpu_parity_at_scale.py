"""K parity against the oracle at 10^5 - 10^6 elements: the assembled stiffness matrix of the HIP path,
exported through mfh_export_upper_triplets (== TripletMatrix::dumpBinary after sumRepeated, SparseMatrices.hh:280-374,629-645),
against the plain-C oracle's CSC of the reference loop structure (threaded perElementStiffness -> serial upper-triplet push ->
sumRepeated -> CSC; LinearElasticity.hh:165-232,1408-1466) on the same mesh: pattern bit-exact and in the same order, values to
1e-12 relative to max |K| (FP64; the two sides sum an entry's <= ~30 contributions in different orders).
The small-size tests of test_gpu_parity.py compare the same things at 10^2 - 10^3 elements; the bench line carries the same check
on its CPU-baseline sample (cpu_baseline.k_parity)."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import meshfem_oracle as O
from oracle import parity
import meshfem_amd as M
from meshfem_amd import grid

pytestmark = pytest.mark.gpu

K_RTOL = 1e-12


def _check(kp):
    assert kp["order_is_sumRepeated"] and kp["upper_only"], kp
    assert kp["pattern_identical"], kp
    assert kp["max_rel_err"] <= K_RTOL, kp


@pytest.mark.timeout(900)
@pytest.mark.parametrize("deg,n", [(2, 24), (1, 40)])
def test_isotropic_K_matches_oracle_at_scale(deg, n):
    """24^3 grid -> 331 776 P2 tets (1.45 M DOF, ~41 M upper entries); 40^3 -> 1 536 000 P1 tets (BASELINE configs[1] is 35^3)."""
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    c = M.Context(0)
    c.mesh_build(T, V, deg)
    c.material_isotropic(200.0, 0.35)
    c.assemble()
    i, j, v = c.export_upper_triplets()
    en, nn = c.elem_nodes(), c.n_node
    c.close()
    D = O.ElasticityTensor.isotropic(3, 200.0, 0.35).D
    Ap, Ai, Ax, _ = CO.assemble_csc(3, deg, en, V, D, nn)
    kp = parity.compare_upper_triplets_with_csc(i, j, v, Ap, Ai, Ax)
    assert kp["nnz_compared"] > (4e7 if deg == 2 else 9e6)
    _check(kp)


@pytest.mark.timeout(900)
def test_orthotropic_field_periodic_K_matches_oracle_at_scale():
    """BASELINE configs[3] at 16^3 (98 304 P2 tets): per-element orthotropic field (the generator's seeded draw), periodic DoF
    map. The device inverts each element's compliance matrix itself (ElasticityTensor.hh:136-152); the oracle gets the 6x6 tensors
    from numpy's inverse of the same matrices."""
    n = 16
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    P = grid.synthetic_orthotropic_field(len(T), 3, seed=0)
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_ortho_field(P)
    ndof = c.apply_periodic_conditions()
    c.assemble()
    i, j, v = c.export_upper_triplets()
    en, nn = c.elem_nodes(), c.n_node
    dof, _ = c.get_dof_map()
    c.close()
    assert ndof < nn
    S = np.zeros((len(T), 6, 6))
    S[:, 0, 0], S[:, 1, 1], S[:, 2, 2] = 1 / P[:, 0], 1 / P[:, 1], 1 / P[:, 2]
    S[:, 0, 1] = S[:, 1, 0] = -P[:, 3] / P[:, 1]
    S[:, 0, 2] = S[:, 2, 0] = -P[:, 4] / P[:, 2]
    S[:, 1, 2] = S[:, 2, 1] = -P[:, 5] / P[:, 2]
    S[:, 3, 3], S[:, 4, 4], S[:, 5, 5] = 1 / P[:, 6], 1 / P[:, 7], 1 / P[:, 8]
    D = np.linalg.inv(S)
    assert np.allclose(D[7], O.ElasticityTensor.orthotropic3d(*P[7]).D, rtol=1e-13, atol=0)
    Ap, Ai, Ax, _ = CO.assemble_csc(3, 2, en, V, D, ndof, dof)
    kp = parity.compare_upper_triplets_with_csc(i, j, v, Ap, Ai, Ax)
    assert kp["n"] == 3 * ndof and kp["nnz_compared"] > 1e7
    _check(kp)
