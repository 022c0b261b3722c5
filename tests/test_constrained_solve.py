"""KKT branch of Simulator::solve: pin / translation / rotation constraints of
assembleConstrainedSystem (LinearElasticity.hh:1201-1249). The reference solves [[K, C^T], [C, 0]] with UMFPACK;
the oracle does the same with a sparse LU; the HIP path eliminates the <= 6 rows around SPD PCG solves.
Tolerance: U_RTOL rel-L2 on displacements (north-star tolerance), PCG rtol 1e-11."""
import numpy as np
import pytest

from oracle import meshfem_oracle as O

U_RTOL = 1e-6
BIG = 1e9


def _grid(dim, deg):
    if dim == 3:
        V, T = O.grid_tet_mesh(3, 2, 2)
        V = V / np.array([3.0, 2.0, 2.0])
    else:
        V, Q = O.gen_grid_2d(5, 4)
        V, T = O.quad_tri_subdiv(V, Q)
        V = V[:, :2] / np.array([5.0, 4.0])
    return V, T


def _face(dim, axis, at):
    lo = [-BIG] * dim; hi = [BIG] * dim
    lo[axis], hi[axis] = at - 1e-9, at + 1e-9
    return lo, hi


def _oracle_sim(V, T, deg, dim):
    sim = O.Simulator(T, V, deg)
    sim.set_material_constant(O.ElasticityTensor.isotropic(dim, 200.0, 0.35))
    return sim


# ------------------------------------------------------------------------------------------------ CPU
def test_oracle_kkt_reduces_to_spd_cases():
    """With a well-posed Dirichlet problem the constrained path equals the SPD path; the rigid-motion
    constraint on a free body returns a solution orthogonal to the six rigid modes with f - K u in their span."""
    V, T = _grid(3, 1)
    sim = _oracle_sim(V, T, 1, 3)
    sim.apply_dirichlet_box(*_face(3, 0, 0.0), [0, 0, 0])
    sim.apply_neumann_box(*_face(3, 0, 1.0), [0, -1, 0], "traction")
    assert np.abs(O.solve_constrained(sim) - sim.solve()).max() < 1e-12
    free = _oracle_sim(V, T, 1, 3)
    free.apply_neumann_box(*_face(3, 0, 1.0), [1, 0.3, 0], "traction")
    free.apply_neumann_box(*_face(3, 0, 0.0), [-1, 0.1, 0], "traction")      # not self-equilibrated: multipliers act
    with pytest.raises(RuntimeError, match="Unimplemented"):
        O.solve_constrained(free)
    u = O.solve_constrained(free, no_rigid_motion=True)
    R = np.vstack([O.rotation_rows(free), O.translation_rows(free, [0, 1, 2])])
    assert np.abs(R @ u.ravel()).max() < 1e-10
    K = free.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper()
    res = free.neumannLoad().ravel() - K @ u.ravel()
    coef = np.linalg.lstsq(R.T, res, rcond=None)[0]
    assert np.abs(res - R.T @ coef).max() < 1e-9 * np.abs(res).max()
    with pytest.raises(RuntimeError, match="Invalid rigid motion RHS"):
        O.solve_constrained(free, no_rigid_motion=True, rm_rhs=[0.0] * 5)


# ------------------------------------------------------------------------------------------------ GPU
def _gpu_sim(V, T, deg):
    import meshfem_amd as M
    sim = M.Simulator(T, V, deg)
    sim.setIsotropicMaterial(200.0, 0.35)
    sim.rtol = 1e-11
    return sim


def _err(u, ref):
    return np.linalg.norm(u - ref) / np.linalg.norm(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,deg", [(3, 1), (3, 2), (2, 2)])
@pytest.mark.parametrize("pin", [False, True])
def test_free_body_no_rigid_motion(dim, deg, pin):
    """No Dirichlet condition at all: rotation rows + translation rows (or pin). One consistent singular solve."""
    import meshfem_amd as M
    V, T = _grid(dim, deg)
    t1 = [1.0, 0.3, 0.0][:dim]; t0 = [-1.0, 0.1, 0.0][:dim]                  # net force and moment: multipliers act
    ref = _oracle_sim(V, T, deg, dim)
    ref.apply_neumann_box(*_face(dim, 0, 1.0), t1, "traction"); ref.apply_neumann_box(*_face(dim, 0, 0.0), t0, "traction")
    sim = _gpu_sim(V, T, deg)
    sim.applyNeumannBox(*_face(dim, 0, 1.0), t1); sim.applyNeumannBox(*_face(dim, 0, 0.0), t0)
    with pytest.raises(M.MeshFEMHipError, match="Unimplemented"):
        sim.solve()
    sim.applyNoRigidMotionConstraint()
    sim.setUsePinNoRigidTranslationConstraint(pin)
    u = sim.solve()
    u_ref = O.solve_constrained(ref, use_pin=pin, no_rigid_motion=True)
    assert _err(u, u_ref) < U_RTOL
    if not pin:
        # prescribed rigid motion: the constraint right-hand side (m_rigidMotionConstraintRHS)
        nrot = 3 if dim == 3 else 1
        rhs = np.concatenate([np.full(nrot, 0.02), np.arange(1, dim + 1) * 0.1])
        sim.setRigidMotionConstraintRHS(rhs)
        assert _err(sim.solve(), O.solve_constrained(ref, no_rigid_motion=True, rm_rhs=rhs)) < U_RTOL
        sim.setRigidMotionConstraintRHS(rhs[:-1])
        with pytest.raises(M.MeshFEMHipError, match="Invalid rigid motion RHS"):
            sim.solve()


@pytest.mark.gpu
@pytest.mark.parametrize("pin", [False, True])
def test_partial_dirichlet_posedness_analysis(pin):
    """(y, z) fixed on the faces x = 0 and x = 1, nothing fixes x: analyzeDirichletPosedness adds the x translation
    constraint (row, or pinned component)."""
    V, T = _grid(3, 2)
    ref = _oracle_sim(V, T, 2, 3)
    sim = _gpu_sim(V, T, 2)
    for at in (0.0, 1.0):
        ref.apply_dirichlet_box(*_face(3, 0, at), [0, 0.01 * at, 0], (False, True, True))
        sim.applyDirichletBox(*_face(3, 0, at), [0, 0.01 * at, 0], components=[False, True, True])
    ref.apply_neumann_box(*_face(3, 1, 1.0), [0.5, -1, 0], "traction")
    sim.applyNeumannBox(*_face(3, 1, 1.0), [0.5, -1, 0])
    sim.setUsePinNoRigidTranslationConstraint(pin)
    assert _err(sim.solve(), O.solve_constrained(ref, use_pin=pin)) < U_RTOL


@pytest.mark.gpu
def test_dirichlet_plus_rigid_motion_rows_schur_path():
    """A clamped face AND the six rigid-motion rows: K is regular on the free variables, the rows are genuine extra
    constraints (Schur complement, 7 PCG solves)."""
    V, T = _grid(3, 1)
    ref = _oracle_sim(V, T, 1, 3)
    sim = _gpu_sim(V, T, 1)
    ref.apply_dirichlet_box(*_face(3, 0, 0.0), [0, 0, 0]); sim.applyDirichletBox(*_face(3, 0, 0.0), [0, 0, 0])
    ref.apply_neumann_box(*_face(3, 0, 1.0), [0, -1, 0.2], "traction"); sim.applyNeumannBox(*_face(3, 0, 1.0), [0, -1, 0.2])
    sim.applyNoRigidMotionConstraint()
    u = sim.solve()
    assert sim.info["iterations"] > 0
    assert _err(u, O.solve_constrained(ref, no_rigid_motion=True)) < U_RTOL


@pytest.mark.gpu
def test_periodic_cell_translation_rows_instead_of_pin():
    """solveCellProblems' system with the translation rows instead of the pinned node: same strains, fluctuation
    displacements with zero mean over the DoFs."""
    from meshfem_amd import homogenization as H
    V, T = O.grid_tet_mesh(3, 3, 3)
    V = V / 3.0
    keep = ~np.all((V[T].mean(axis=1) > 1 / 3) & (V[T].mean(axis=1) < 2 / 3), axis=1)   # a void in the middle
    T = T[keep]
    used = np.unique(T)                                   # drop the void's centre vertex (it would float)
    remap = np.full(len(V), -1); remap[used] = np.arange(len(used))
    V, T = V[used], remap[T]
    ref = _oracle_sim(V, T, 2, 3)
    ref.applyPeriodicConditions()
    rhs_ref = ref.constantStrainLoad(-O.canonical_strain(3, 0))
    u_ref = O.solve_constrained(ref, f=rhs_ref, no_rigid_motion=True)
    sim = _gpu_sim(V, T, 2)
    sim.applyPeriodicConditions()
    sim.applyNoRigidMotionConstraint()
    u = sim.solve(sim.constantStrainLoad(-H.canonical_strain_flat(3, 0)))
    assert _err(u, u_ref) < U_RTOL
    dm, nd = sim.ctx.get_dof_map()
    first = np.unique(dm, return_index=True)[1]
    assert np.abs(u[first].sum(axis=0)).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("deg", [1, 2])
def test_mixed_case_fixed_component_plus_rigid_motion_rows(deg):
    """only the x component is fixed on the face x = 0, so the translations along y, z AND the rotation about x
    survive (3 null modes), while `no_rigid_motion` adds 6 rows: neither "rows match the null space" (q == k) nor "K
    regular" (q == 0). Round 1 rejected this; the general elimination (lambda = lambda0 + N mu, (k - q) + 1 consistent
    singular solves) must reproduce the oracle's sparse-LU KKT solve."""
    import meshfem_amd as M
    V, T = _grid(3, deg)
    ref = _oracle_sim(V, T, deg, 3)
    sim = _gpu_sim(V, T, deg)
    ref.apply_dirichlet_box(*_face(3, 0, 0.0), [0.01, 0, 0], (True, False, False))
    sim.applyDirichletBox(*_face(3, 0, 0.0), [0.01, 0, 0], components=[True, False, False])
    ref.apply_neumann_box(*_face(3, 0, 1.0), [0.3, -1, 0.2], "traction")
    sim.applyNeumannBox(*_face(3, 0, 1.0), [0.3, -1, 0.2])
    ref.useRigidMotionConstraint = True
    sim.applyNoRigidMotionConstraint()
    sim.setUsePinNoRigidTranslationConstraint(False)
    u_ref = O.solve_constrained(ref, no_rigid_motion=True)
    assert _err(sim.solve(), u_ref) < U_RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("dim,deg,n", [(3, 2, 6), (3, 1, 10), (2, 2, 16)])
def test_free_body_with_the_multigrid_preconditioner(dim, deg, n):
    """K singular on the free variables (no Dirichlet condition; six / three rigid-motion rows): the multigrid hierarchy is rebuilt with its
    dense last level pinned and stays in use -- same displacements as the block-Jacobi solve of the same singular system (which test_free_body_no_rigid_motion checks against the
    oracle's KKT solve), a fraction of its iterations; a
    regular solve on the same context afterwards switches the hierarchy back."""
    import meshfem_amd as M
    from meshfem_amd import grid
    if dim == 3:
        V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    else:
        V, T = grid.grid_tri_mesh(n, n)
        V = V / float(n)
    t1 = [1.0, 0.3, 0.0][:dim]; t0 = [-1.0, 0.1, 0.0][:dim]
    sim = _gpu_sim(V, T, deg)
    sim.rtol = 1e-10
    sim.applyNeumannBox(*_face(dim, 0, 1.0), t1); sim.applyNeumannBox(*_face(dim, 0, 0.0), t0)
    sim.applyNoRigidMotionConstraint()
    sim.ctx.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
    u_bj = sim.solve()
    it_bj = sim.info["iterations"]
    sim.ctx.set_preconditioner(M.PRECOND_MULTIGRID)
    u = sim.solve()
    it_mg, note = sim.info["iterations"], sim.ctx.precond_info()["note"]
    assert sim.info["converged"] and "pinned" in note, note
    assert _err(u, u_bj) < 1e-7 and it_mg < 0.35 * it_bj, (it_mg, it_bj)
    # a regular system on the same context: clamp a face, drop the rows
    sim.removeNoRigidMotionConstraint()
    sim.applyDirichletBox(*_face(dim, 0, 0.0), [0.0] * dim)
    u2 = sim.solve()
    assert sim.info["converged"] and sim.info["iterations"] < 70 and "pinned" not in sim.ctx.precond_info()["note"]
    sim.ctx.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
    assert _err(u2, sim.solve()) < 1e-7
