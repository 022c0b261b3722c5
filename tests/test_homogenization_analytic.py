"""Fixtures that come from NEITHER the oracle NOR the HIP path ("add at least one fixture from an independent
source"): the closed-form homogenized tensor of a two-phase laminate, and BASELINE configs[3] at full size through
size-independent properties.

Laminate: isotropic layers normal to z, volume fractions f_a / f_b. Periodic cell problems are solved exactly by fields
that are piecewise linear in z, which both P1 and P2 elements on a layer-aligned grid reproduce, so Ch equals the
classical laminate formulas to rounding (Backus 1962 / Milton, "The Theory of Composites", ch. 9):
    with <.> the volume average, L = lambda, M = mu, P = L + 2M:
    C33 = <1/P>^-1                    C44 = C55 = <1/M>^-1              C66 = <M>
    C13 = C23 = <L/P> C33             C11 = C22 = <4M(L+M)/P> + <L/P>^2 C33       C12 = C11 - 2 C66
(Voigt order of the reference, Flattening.hh:47-60: xx, yy, zz, yz, xz, xy; tensor shear entries.)"""
import numpy as np
import pytest

from oracle import meshfem_oracle as O


def laminate_closed_form(Ea, nua, Eb, nub, fa):
    def lame(E, nu):
        return nu * E / ((1 + nu) * (1 - 2 * nu)), E / (2 + 2 * nu)
    (La, Ma), (Lb, Mb) = lame(Ea, nua), lame(Eb, nub)
    fb = 1 - fa
    avg = lambda a, b: fa * a + fb * b                                       # noqa: E731
    Pa, Pb = La + 2 * Ma, Lb + 2 * Mb
    C33 = 1 / avg(1 / Pa, 1 / Pb)
    C44 = 1 / avg(1 / Ma, 1 / Mb)
    C66 = avg(Ma, Mb)
    C13 = avg(La / Pa, Lb / Pb) * C33
    C11 = avg(4 * Ma * (La + Ma) / Pa, 4 * Mb * (Lb + Mb) / Pb) + avg(La / Pa, Lb / Pb) ** 2 * C33
    C12 = C11 - 2 * C66
    C = np.zeros((6, 6))
    C[:3, :3] = [[C11, C12, C13], [C12, C11, C13], [C13, C13, C33]]
    C[3, 3] = C[4, 4] = C44
    C[5, 5] = C66
    return C


def _laminate_mesh(n=4, layers_a=1):
    V, T = O.grid_tet_mesh(n, n, n)
    V = V / n
    zc = V[T].mean(axis=1)[:, 2]
    in_a = zc < layers_a / n                     # the lowest `layers_a` hex layers are phase a
    return V, T, in_a


@pytest.mark.parametrize("deg", [1, 2])
def test_oracle_laminate_homogenization_matches_closed_form(deg):
    """CPU: the oracle's periodic homogenization (direct solves) against the analytic laminate tensor."""
    V, T, in_a = _laminate_mesh(3, 1)
    E = np.where(in_a, 300.0, 90.0)
    nu = np.where(in_a, 0.2, 0.35)
    sim = O.Simulator(T, V, deg)
    sim.set_material_field([O.ElasticityTensor.isotropic(3, float(e), float(v)) for e, v in zip(E, nu)])
    w = O.solve_cell_problems(sim)
    Ch = O.homogenized_elasticity_tensor(sim, w)
    ref = laminate_closed_form(300.0, 0.2, 90.0, 0.35, 1.0 / 3.0)
    assert np.abs(Ch - ref).max() <= 1e-10 * np.abs(ref).max(), np.abs(Ch - ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("deg", [1, 2])
@pytest.mark.parametrize("precond", ["block_jacobi", "two_level"])
def test_hip_laminate_homogenization_matches_closed_form(deg, precond):
    """GPU: periodic DoFs, per-element isotropic field, six cell problems, stress-form Ch against the analytic tensor.
    Tolerance: the PCG solves to rtol 1e-12; Ch depends on the fluctuations through volume averages (1e-9)."""
    import meshfem_amd as M
    from meshfem_amd import homogenization as H
    V, T, in_a = _laminate_mesh(6, 2)
    E = np.where(in_a, 300.0, 90.0)
    nu = np.where(in_a, 0.2, 0.35)
    r = H.homogenize(V, T, deg, E=E, nu=nu, rtol=1e-12,
                     preconditioner=M.PRECOND_TWO_LEVEL if precond == "two_level" else M.PRECOND_BLOCK_JACOBI)
    ref = laminate_closed_form(300.0, 0.2, 90.0, 0.35, 1.0 / 3.0)
    assert np.abs(r["Ch"] - ref).max() <= 1e-9 * np.abs(ref).max(), np.abs(r["Ch"] - ref).max()
    r["sim"].ctx.close()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_configs3_full_size_properties():
    """BASELINE configs[3]: 44^3 grid -> 2,044,416 P2 tets, per-element orthotropic field (SURVEY 8d ranges, seed 0), periodic
    DoFs, 6 cell problems (PeriodicHomogenization.hh:34-54). No direct solve exists at this size; size-independent
    properties instead: every solve converged with a TRUE residual <= 2 rtol, Ch has the major symmetry to 1e-10 and is
    positive definite, lies between the Reuss and Voigt bounds of the field (diagonal entries), and the two preconditioners
    (independent Krylov histories) give the same Ch to 1e-9, the same element strains to 1e-6 and the same centred
    fluctuations to 1e-5."""
    import meshfem_amd as M
    from meshfem_amd import grid, homogenization as H
    n = 44
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    P = grid.synthetic_orthotropic_field(len(T), 3, 0)
    res = {}
    for name, pc in (("two_level", M.PRECOND_TWO_LEVEL), ("block_jacobi", M.PRECOND_BLOCK_JACOBI)):
        r = H.homogenize(V, T, 2, ortho_params=P, rtol=1e-9, preconditioner=pc)
        sim = r["sim"]
        assert 3 * sim.numDoFs() > 8.5e6
        Ch = r["Ch"]
        assert np.abs(Ch - Ch.T).max() <= 1e-10 * np.abs(Ch).max()
        assert np.linalg.eigvalsh(0.5 * (Ch + Ch.T)).min() > 10.0
        for info in r["infos"]:
            assert info["converged"] and info["true_rel_residual"] <= 2e-9
        res[name] = (Ch, [w.copy() for w in r["w_ij"]], r["iterations"], [e.copy() for e in r["strain_w_ij"]])
        if name == "two_level":
            # Voigt (volume-average stiffness) / Reuss (volume-average compliance) bounds on the diagonal; equal volumes
            D = np.stack([sim.ctx.material_get(e) for e in range(0, len(T), 997)])
            voigt = D.mean(axis=0)
            reuss = np.linalg.inv(np.linalg.inv(D).mean(axis=0))
            for i in range(6):
                assert reuss[i, i] * 0.97 <= Ch[i, i] <= voigt[i, i] * 1.03, (i, reuss[i, i], Ch[i, i], voigt[i, i])
        sim.ctx.close()
    (Ca, wa, ia, ea), (Cb, wb, ib, eb) = res["two_level"], res["block_jacobi"]
    assert np.abs(Ca - Cb).max() <= 1e-9 * np.abs(Cb).max()
    # The cell problems are pinned at ONE node, so a near-translation of the whole cell costs almost no energy: a residual
    # of 1e-9 leaves that mode loose at the 1e-4 level (measured 2.7e-4) without touching any strain. Compare what the
    # homogenization consumes -- the element strains -- and the displacements after removing the mean translation.
    for x, y in zip(ea, eb):
        assert np.linalg.norm(x - y) <= 1e-6 * np.linalg.norm(y)
    for x, y in zip(wa, wb):
        xc, yc = x - x.mean(axis=0), y - y.mean(axis=0)
        assert np.linalg.norm(xc - yc) <= 1e-5 * np.linalg.norm(yc)
    assert max(ia) < 0.4 * max(ib)                      # the coarse space pays: 5-10x fewer iterations at this size
