"""CPU tests of the host-side logic of libmeshfem_hip (host-only context, device = -1): FEMMesh
numbering, boundary extraction, periodic DoF map, symbolic phase (pattern + gather lists) and
the Simulator-level boundary-condition helpers, all against the oracle. Index results are bit-exact."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import meshfem_oracle as O
import meshfem_amd as M
from meshfem_amd import grid


def test_grid_generator_matches_reference_loops():
    for dims in [(1, 1, 1), (3, 2, 2), (2, 4, 3)]:
        V, T = O.grid_tet_mesh(*dims)
        V2, T2 = grid.grid_tet_mesh(*dims)
        assert np.array_equal(T, T2) and np.array_equal(V, V2)
        assert len(T) == 24 * np.prod(dims)
    V, T = grid.grid_tet_mesh(4, 2, 2, [0, 0, 0], [2, 1, 1])
    assert np.allclose(V.max(axis=0), [2, 1, 1])
    sim = O.Simulator(T, V, 1)
    assert sim.vol.min() > 0 and abs(sim.vol.sum() - 2.0) < 1e-12


@pytest.mark.parametrize("deg", [1, 2])
def test_femmesh_numbering_and_boundary(deg):
    V, T = grid.grid_tet_mesh(4, 3, 2)
    m = O.FEMMesh(T, V, deg)
    c = M.Context(-1)
    c.mesh_build(T, V, deg)
    assert c.n_node == m.num_nodes and c.n_bdry_elem == len(m.bdry_elem_nodes)
    assert np.array_equal(c.elem_nodes(), m.elem_nodes)
    assert np.array_equal(c.node_positions(), m.node_pos)
    assert np.array_equal(c.boundary_elem_nodes(), m.bdry_elem_nodes)
    assert np.array_equal(c.boundary_nodes(), m.bdry_nodes)
    bv, bn = m.bdry_elem_geometry()
    v2, n2 = c.boundary_elem_geometry()
    assert np.abs(bv - v2).max() < 1e-15 and np.abs(bn - n2).max() < 1e-15
    # outward normals: boundary of a box
    ctr = V[m.bdry_elem_verts].mean(axis=1) - V.mean(axis=0)
    assert (np.einsum("ij,ij->i", ctr, n2) > 0).all()
    assert c.pin_node() == int(np.flatnonzero(~m.is_bdry_node)[0])


@pytest.mark.parametrize("deg", [1, 2])
def test_femmesh_2d(deg):
    V, T = grid.grid_tri_mesh(3, 2)
    m = O.FEMMesh(T, V, deg)
    c = M.Context(-1)
    c.mesh_build(T, V, deg)
    assert np.array_equal(c.elem_nodes(), m.elem_nodes)
    assert np.array_equal(c.boundary_elem_nodes(), m.bdry_elem_nodes)
    assert np.array_equal(c.boundary_nodes(), m.bdry_nodes)
    bv, bn = m.bdry_elem_geometry()
    v2, n2 = c.boundary_elem_geometry()
    assert np.abs(bv - v2).max() < 1e-15 and np.abs(bn - n2).max() < 1e-15


def test_unstructured_disjoint_tets_and_bad_input():
    rng = np.random.default_rng(1)
    n = 300
    V = rng.random((4 * n, 3)) + np.repeat(np.arange(n), 4)[:, None] * 2.0
    T = np.arange(4 * n).reshape(n, 4)
    c = M.Context(-1)
    c.mesh_build(T, V, 2)          # 6 edge nodes per element: hash table growth path
    m = O.FEMMesh(T, V, 2)
    assert np.array_equal(c.elem_nodes(), m.elem_nodes) and c.n_node == 10 * n
    with pytest.raises(M.MeshFEMHipError):
        c.mesh_build(np.array([[0, 1, 2, 99999]]), V, 1)      # "Bad vertex index encountered."
    with pytest.raises(M.MeshFEMHipError):
        c.mesh_build(np.array([[0, 1, 2, 3], [0, 1, 2, 4], [0, 1, 2, 5]]), V, 1)   # non-manifold face
    # a build that fails leaves a context WITHOUT a mesh (not with half of the new one or the tables of the old one) ...
    with pytest.raises(M.MeshFEMHipError, match="no mesh"):
        c.fix_variables(np.array([0]))
    # ... and the context takes the next mesh as if nothing had happened
    c.mesh_build(T, V, 2)
    assert np.array_equal(c.elem_nodes(), m.elem_nodes)


def _emulate_gather(c, Ke, dim):
    S = c.symbolic_get(True)
    npe = c.npe
    nr = len(S["rowPtr"]) - 1
    vals = np.zeros((len(S["colIdx"]), dim, dim))
    KeB = Ke.reshape(len(Ke), npe, dim, npe, dim)
    for ch in range(S["n_chunk"]):
        r0 = S["chunkRow"][ch]
        s0 = S["rowPtr"][r0]
        b, e = S["contribPtr"][ch], S["contribPtr"][ch + 1]
        code = S["contribCode"][b:e].astype(np.int64)
        ls = S["contribSlot"][b:e].astype(np.int64)
        assert ls.max() < S["chunk_slots"] and s0 + ls.max() < S["rowPtr"][S["chunkRow"][ch + 1]]
        el, ij = code // (npe * npe), code % (npe * npe)
        np.add.at(vals, s0 + ls, KeB[el, ij // npe, :, ij % npe, :])
        assert np.array_equal(S["scatterSlot"][code], s0 + ls)
    return sp.bsr_matrix((vals, S["colIdx"], S["rowPtr"]), shape=(nr * dim, c.matrix_info()[1] * dim)).tocsr(), S


@pytest.mark.parametrize("deg", [1, 2])
@pytest.mark.parametrize("order", [0, 1, 2])
def test_symbolic_phase_reproduces_oracle_matrix(deg, order):
    """The gather lists, applied to the oracle's Ke on the CPU, give the oracle's K: validates the
    block-CSR pattern, chunking, contribution codes/slots and the scatter map (a6, a9)."""
    V, T = grid.grid_tet_mesh(3, 2, 2)
    sim = O.Simulator(T, V, deg)
    sim.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
    Ke = sim.per_element_stiffness()
    Kref = sim.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper()
    c = M.Context(-1)
    c.mesh_build(T, V, deg)
    c.set_option("chunk_slots", 128 if deg == 1 else 256)
    c.set_option("contrib_order", order)
    c.symbolic(True)
    A, S = _emulate_gather(c, Ke, 3)
    assert abs(A - Kref).max() < 1e-14 * abs(Kref).max()
    # pattern: sorted unique columns per row, every contribution accounted for once
    assert S["n_contrib"] == len(T) * c.npe ** 2
    for r in range(len(S["rowPtr"]) - 1):
        cols = S["colIdx"][S["rowPtr"][r]:S["rowPtr"][r + 1]]
        assert np.all(np.diff(cols) > 0)
    assert np.all(np.diff(S["chunkRow"]) > 0)
    assert np.all(S["rowPtr"][S["chunkRow"][1:]] - S["rowPtr"][S["chunkRow"][:-1]] <= S["chunk_slots"])


def test_periodic_dof_map_and_partitioned_rows():
    V, T = grid.grid_tet_mesh(3, 2, 2)
    sim = O.Simulator(T, V, 2)
    sim.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
    Ke = sim.per_element_stiffness()
    c = M.Context(-1)
    c.mesh_build(T, V, 2)
    nd = c.apply_periodic_conditions(1e-7)
    sim.applyPeriodicConditions()
    dm, nd2 = c.get_dof_map()
    assert nd == nd2 == sim.numDoFs() and np.array_equal(dm, sim.dofForNode)
    c.symbolic(True)
    A, _ = _emulate_gather(c, Ke, 3)
    Kp = sim.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper()
    assert abs(A - Kp).max() < 1e-14 * abs(Kp).max()
    # row partition: only the first nOwned block rows are assembled (multi-GPU local problem)
    c2 = M.Context(-1)
    m = sim.mesh
    n_owned = m.num_nodes // 2
    c2.mesh_set(3, 2, m.elem_nodes, m.node_pos, n_owned)
    c2.symbolic(True)
    B, _ = _emulate_gather(c2, Ke, 3)
    sim2 = O.Simulator(T, V, 2, mesh=m)
    sim2.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
    Kfull = sim2.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper().tocsr()
    assert B.shape == (3 * n_owned, 3 * m.num_nodes)
    assert abs(B - Kfull[:3 * n_owned]).max() < 1e-14 * abs(Kfull).max()


@pytest.mark.parametrize("deg", [1, 2])
def test_boundary_condition_helpers(deg):
    """cantilever.bc semantics: box% Dirichlet on boundary nodes, total force / region area."""
    V, T = grid.grid_tet_mesh(5, 2, 2)
    c = M.Context(-1)
    c.mesh_build(T, V, deg)
    sim = O.Simulator(T, V, deg)
    c.bc_dirichlet_box([-1e-4] * 3, [1e-4, 1.0001, 1.0001], [0, 0, 0], relative=True)
    c.bc_neumann_box([0.9999, -1e-4, -1e-4], [1.0001, 1.0001, 1.0001], [0, -10, 0], kind=M.NEUMANN_FORCE, relative=True)
    mn, mx = sim.box_percent([-1e-4] * 3, [1e-4, 1.0001, 1.0001]); sim.apply_dirichlet_box(mn, mx, [0, 0, 0])
    mn, mx = sim.box_percent([0.9999, -1e-4, -1e-4], [1.0001, 1.0001, 1.0001]); sim.apply_neumann_box(mn, mx, [0, -10, 0], "force")
    dv, dx = c.bc_dirichlet_vars()
    rv, rx = sim.dirichlet_vars_and_values()
    assert list(dv) == list(rv) and np.array_equal(dx, rx)
    f, fr = c.neumann_load(), sim.neumannLoad()
    assert np.abs(f - fr).max() < 1e-15 and abs(f[:, 1].sum() + 10) < 1e-12
    # pressure: traction = -p n
    c.bc_clear()
    c.bc_neumann_box([0.9999, -1e-4, -1e-4], [1.0001, 1.0001, 1.0001], [2.5], kind=M.NEUMANN_PRESSURE, relative=True)
    f = c.neumann_load()
    assert abs(f[:, 0].sum() + 2.5 * 4.0) < 1e-12          # face area 2x2, normal +x
    with pytest.raises(M.MeshFEMHipError, match="unmatched"):
        c.bc_neumann_box([5, 5, 5], [6, 6, 6], [0, 0, 1])
    c.bc_dirichlet_box([-1e-4] * 3, [1e-4, 1.0001, 1.0001], [0, 0, 0], relative=True)
    with pytest.raises(M.MeshFEMHipError, match="Conflicting"):
        c.bc_dirichlet_box([-1e-4] * 3, [1e-4, 1.0001, 1.0001], [1, 0, 0], relative=True)
    with pytest.raises(M.MeshFEMHipError, match="already fixed"):
        c.fix_variables([3, 3])


def test_reorder_mesh_is_a_relabelling():
    """grid.reorder_mesh (bench ordering variants): same elements up to numbering, orientation kept."""
    from meshfem_amd import grid
    V, T = grid.grid_tet_mesh(3, 2, 2)
    ref = sorted(map(tuple, V[T].reshape(len(T), -1).round(12).tolist()))
    for mode in ("shuffle", "morton"):
        V2, T2 = grid.reorder_mesh(V, T, mode)
        assert sorted(map(tuple, V2[T2].reshape(len(T), -1).round(12).tolist())) == ref
        assert not np.array_equal(T2, T)
    with pytest.raises(ValueError):
        grid.reorder_mesh(V, T, "hilbert")


def test_mesh_update_vertices_keeps_topology_and_recomputes_geometry():
    """mfh_mesh_update_vertices == updateMeshNodePositions (LinearElasticity.hh:1279-1284): node positions (P2 edge
    midpoints) and boundary areas / normals follow the new vertices; numbering and boundary lists are untouched."""
    import meshfem_amd as M
    from meshfem_amd import grid
    rng = np.random.default_rng(0)
    V, T = grid.grid_tet_mesh(3, 2, 2, [0, 0, 0], [1, 1, 1])
    V2 = V + 0.02 * rng.normal(size=V.shape)
    a, b = M.Context(-1), M.Context(-1)
    a.mesh_build(T, V, 2)
    en, ben, bn = a.elem_nodes(), a.boundary_elem_nodes(), a.boundary_nodes()
    a.mesh_update_vertices(V2)
    b.mesh_build(T, V2, 2)
    assert np.array_equal(a.elem_nodes(), en) and np.array_equal(a.boundary_elem_nodes(), ben) and np.array_equal(a.boundary_nodes(), bn)
    assert np.array_equal(a.node_positions(), b.node_positions())
    (va, na), (vb, nb) = a.boundary_elem_geometry(), b.boundary_elem_geometry()
    assert np.array_equal(va, vb) and np.array_equal(na, nb)
    with pytest.raises(ValueError):
        a.mesh_update_vertices(V2[:-1])
    a.close(); b.close()


@pytest.mark.parametrize("order", [0, 1, 2])
def test_upper_only_storage_symbolic_phase(order):
    """Option matrix_storage 1 (the triangle the reference's TripletMatrix holds): the gather lists cover exactly the blocks
    (r, c >= r); applied to the oracle's Ke they give the block upper triangle of the oracle's K with 55 of every
    element's 100 contributions."""
    V, T = grid.grid_tet_mesh(3, 2, 2)
    sim = O.Simulator(T, V, 2)
    sim.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
    Ke = sim.per_element_stiffness()
    Kref = sim.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper()
    c = M.Context(-1)
    c.mesh_build(T, V, 2)
    c.set_option("matrix_storage", 1)
    c.set_option("contrib_order", order)
    c.symbolic(True)
    A, S = _emulate_gather(c, Ke, 3)
    nb = len(S["rowPtr"]) - 1
    B = sp.bsr_matrix(Kref.tocsr(), blocksize=(3, 3))
    rows = np.repeat(np.arange(nb), np.diff(B.indptr))
    keep = B.indices >= rows
    U = sp.bsr_matrix((B.data[keep], B.indices[keep], np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=nb))])), shape=B.shape).tocsr()
    assert abs(A - U).max() < 1e-14 * abs(Kref).max()
    assert S["n_contrib"] == 55 * len(T) and len(S["colIdx"]) == (B.nnz // 9 + nb) // 2
    srows = np.repeat(np.arange(nb), np.diff(S["rowPtr"]))
    assert np.all(S["colIdx"] >= srows) and np.array_equal(S["colIdx"][S["rowPtr"][:-1]], np.arange(nb))   # the diagonal block leads every row
