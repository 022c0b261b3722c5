"""PeriodicCondition (BoundaryConditions.hh:452-561, PeriodicBoundaryMatcher.hh:127-360): DoF identification on the
bounding-box cell -- strict matching (a mismatch throws), ignoreMismatch and ignoreDims. Index work: the host library's
DoF map equals the oracle's bit for bit (no device needed: host-only context)."""
import numpy as np
import pytest

import meshfem_amd as M
from oracle import meshfem_oracle as O


def _grid(dim, n=3):
    if dim == 3:
        V, T = O.grid_tet_mesh(n, n, n)
        return V / n, T
    V, Q = O.gen_grid_2d(n, n)
    V, T = O.quad_tri_subdiv(V, Q)
    return V[:, :2] / n, T


def _host_dofs(V, T, deg, ignore_mismatch=False, ignore_dims=()):
    c = M.Context(-1)
    c.mesh_build(T, V, deg)
    c.set_option("periodic_ignore_mismatch", 1 if ignore_mismatch else 0)
    c.set_option("periodic_ignore_dims", sum(1 << d for d in ignore_dims))
    n = c.apply_periodic_conditions(1e-7)
    dofs = c.get_dof_map()[0]
    c.close()
    return dofs, n


@pytest.mark.parametrize("dim,deg", [(2, 1), (2, 2), (3, 1), (3, 2)])
def test_strict_matching_equals_oracle_and_counts(dim, deg):
    V, T = _grid(dim)
    dofs, n = _host_dofs(V, T, deg)
    mesh = O.FEMMesh(T, V, deg)
    od, on, _ = O.periodic_dofs_for_nodes(mesh)
    assert n == on and np.array_equal(dofs, od)
    # a periodic grid of m^dim cells has exactly as many DoFs as the torus has nodes
    m = 3 * (1 if deg == 1 else 2)
    verts_torus = {2: 3 * 3 + 9, 3: None}[dim]                     # 2D: 9 corners + 9 cell centres (quad_tri_subdiv)
    if dim == 2 and deg == 1:
        assert n == verts_torus
    assert n < mesh.num_nodes and dofs.max() == n - 1 and len(np.unique(dofs)) == n


@pytest.mark.parametrize("dim", [2, 3])
def test_mismatch_throws_unless_ignored(dim):
    V, T = _grid(dim)
    V = V.copy()
    # slide one node along the max-x face: it stays on the face but loses its partner on the min-x face
    on = np.flatnonzero((np.abs(V[:, 0] - 1.0) < 1e-12) & (V[:, 1] > 0.2) & (V[:, 1] < 0.5) & ((V[:, 2] > 0.2) & (V[:, 2] < 0.5) if dim == 3 else True))
    V[on[0], 1] += 0.01
    with pytest.raises(M.MeshFEMHipError, match="periodic-identified node"):
        _host_dofs(V, T, 1)
    with pytest.raises(RuntimeError, match="periodic-identified node|Unmatched non-minimal"):
        O.periodic_dofs_for_nodes(O.FEMMesh(T, V, 1))
    dofs, n = _host_dofs(V, T, 1, ignore_mismatch=True)
    od, on_, _ = O.periodic_dofs_for_nodes(O.FEMMesh(T, V, 1), ignore_mismatch=True)
    assert n == on_ and np.array_equal(dofs, od)
    ref, nref = _host_dofs(_grid(dim)[0], T, 1)
    assert n == nref + 1                                              # the two unmatched nodes keep their own DoFs


@pytest.mark.parametrize("deg", [1, 2])
def test_ignore_dims_keeps_the_cell_periodic_in_the_other_directions(deg):
    V, T = _grid(3)
    dofs, n = _host_dofs(V, T, deg, ignore_dims=(2,))
    mesh = O.FEMMesh(T, V, deg)
    od, on, internal = O.periodic_dofs_for_nodes(mesh, ignore_dims=(2,))
    assert n == on and np.array_equal(dofs, od)
    P = mesh.node_pos
    top, bot = np.abs(P[:, 2] - 1) < 1e-12, np.abs(P[:, 2]) < 1e-12
    assert not set(dofs[top]) & set(dofs[bot])                       # z faces are not identified
    full, nfull = _host_dofs(V, T, deg)
    assert n > nfull
    # boundary elements on the z faces are ordinary boundary (traction may act there); x / y faces are internal
    be = mesh.bdry_elem_nodes
    on_z = np.array([(np.abs(P[b, 2] - 1) < 1e-12).all() or (np.abs(P[b, 2]) < 1e-12).all() for b in be])
    assert not internal[on_z].any() and internal[~on_z].all()


def test_manual_periodic_vertices_file_reproduces_the_detected_map(tmp_path):
    """PeriodicCondition(mesh, pcFile): pairs of identified nodes -> connected components -> DoFs in node order."""
    from meshfem_amd import homogenization as H
    V, T = _grid(2, 4)
    dofs, n = _host_dofs(V, T, 2)
    groups = {}
    for node, d in enumerate(dofs):
        groups.setdefault(int(d), []).append(node)
    lines = []
    for g in groups.values():                                  # a chain per identified set (corners: 4 nodes)
        lines += ["%d %d" % (a, b) for a, b in zip(g[1:], g[:-1])]
    path = tmp_path / "pairs.txt"
    path.write_text("\n".join(lines) + "\n")
    fd, fn = H.periodic_dofs_from_file(str(path), len(dofs))
    assert fn == n and np.array_equal(fd, dofs)
    (tmp_path / "bad.txt").write_text("0 99999\n")
    with pytest.raises(RuntimeError, match="out of range"):
        H.periodic_dofs_from_file(str(tmp_path / "bad.txt"), len(dofs))


@pytest.mark.gpu
def test_manual_periodic_vertices_file_homogenization_equals_detected(tmp_path):
    import os
    from meshfem_amd import homogenization as H, mesh_io
    from meshfem_amd.linear_elasticity import Simulator
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    V, E, _ = mesh_io.load_msh(os.path.join(gold, "meshes", "2D_microstructure.msh"))
    V = V[:, :2].copy()
    res = []
    for manual in (False, True):
        sim = Simulator(E, V, 2); sim.rtol = 1e-11; sim.setIsotropicMaterial(200.0, 0.35)
        if manual:
            w, _ = H.solve_cell_problems(sim, manual_periodic_vertices_file=str(tmp_path / "pairs.txt"))
        else:
            w, _ = H.solve_cell_problems(sim)
            dofs = sim.ctx.get_dof_map()[0]
            groups = {}
            for node, d in enumerate(dofs):
                groups.setdefault(int(d), []).append(node)
            (tmp_path / "pairs.txt").write_text("\n".join("%d %d" % (a, b) for g in groups.values() for a, b in zip(g[1:], g[:-1])) + "\n")
        res.append(H.homogenized_elasticity_tensor(sim, w))
    assert np.abs(res[0] - res[1]).max() < 1e-8 * np.abs(res[0]).max()


@pytest.mark.parametrize("ignore_mismatch", [False, True])
def test_matching_is_by_distance_not_by_quantised_keys(ignore_mismatch):
    """The reference matches with CollisionGrid::getClosestPoint(query, eps) (PeriodicBoundaryMatcher.hh:222, :300;
    CollisionGrid.hh:58-88): partners within eps of the translated position are identified wherever they fall relative
    to any quantisation grid. Coordinate noise 10x below eps on a 12^3 grid, ten seeds (a key-equality matcher fails
    this in about four of ten), plus a half-cell coordinate offset by 4e-10."""
    V0, T = O.grid_tet_mesh(12, 12, 12)
    # spacing 0.0833335: every odd grid coordinate is an odd multiple of 0.5e-6, i.e. it sits on a rounding boundary of a
    # key grid with cells of 10 eps = 1e-6
    L = 12 * 0.0833335
    V0 = V0 * 0.0833335
    ref, nref = _host_dofs(V0, T, 1)
    for seed in range(10):
        rng = np.random.default_rng(seed)
        V = V0 + rng.uniform(-1e-8, 1e-8, V0.shape)
        # the bounding box itself must stay the unit cube: keep the eight corners exact
        corner = np.all((V0 == 0.0) | (np.abs(V0 - L) < 1e-12), axis=1)
        V[corner] = V0[corner]
        dofs, n = _host_dofs(V, T, 1, ignore_mismatch=ignore_mismatch)
        assert n == nref and np.array_equal(dofs, ref), "seed %d" % seed
    V = V0.copy()
    half = np.flatnonzero((np.abs(V0[:, 0] - L) < 1e-12) & (np.abs(V0[:, 1] - L / 2) < 1e-12))
    V[half, 1] += 4e-10
    dofs, n = _host_dofs(V, T, 1, ignore_mismatch=ignore_mismatch)
    assert n == nref and np.array_equal(dofs, ref)
    od, on_, _ = O.periodic_dofs_for_nodes(O.FEMMesh(T, V, 1), ignore_mismatch=ignore_mismatch)
    assert on_ == n and np.array_equal(od, dofs)


def test_distinct_nodes_closer_than_ten_eps_are_not_merged():
    """Nodes of one face that are further apart than eps keep different DoFs even when they are closer than 10 eps
    (a key grid of cell size 10 eps would merge them): eps = 0.02 on a grid whose nodes are 1/6 apart."""
    V, T = O.grid_tet_mesh(3, 3, 3)
    V = V / 3.0
    c = M.Context(-1)
    c.mesh_build(T, V, 1)
    n = c.apply_periodic_conditions(0.02)
    dofs = c.get_dof_map()[0]
    c.close()
    ref, nref = _host_dofs(V, T, 1)
    assert n == nref and np.array_equal(dofs, ref)
    od, on_, _ = O.periodic_dofs_for_nodes(O.FEMMesh(T, V, 1), eps=0.02)
    assert on_ == n and np.array_equal(od, dofs)
