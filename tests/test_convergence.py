"""Mesh-refinement convergence of the whole path (assembly -> non-zero Dirichlet elimination -> PCG), the role the
reference's experiments/elasticity_convergence/run.sh plays (grid sweeps through Simulate_cli). Manufactured solution:
u = grad(phi) with phi harmonic satisfies the homogeneous Navier equations for every isotropic material
(div u = lap phi = 0, lap u = grad lap phi = 0), so prescribing u on the whole boundary and solving with zero load must
converge to u: nodal max-norm error O(h^2) for P1 and O(h^3) for P2 (observed rates asserted with slack)."""
import numpy as np
import pytest

from oracle import meshfem_oracle as O


def _exact(X):
    if X.shape[1] == 2:                       # phi = exp(x) sin(y)
        ex, s, c = np.exp(X[:, 0]), np.sin(X[:, 1]), np.cos(X[:, 1])
        return np.column_stack([ex * s, ex * c])
    a, b = 0.6, 0.8                           # phi = exp(z) sin(a x) cos(b y), a^2 + b^2 = 1
    ez = np.exp(X[:, 2])
    sx, cx, sy, cy = np.sin(a * X[:, 0]), np.cos(a * X[:, 0]), np.sin(b * X[:, 1]), np.cos(b * X[:, 1])
    return np.column_stack([a * ez * cx * cy, -b * ez * sx * sy, ez * sx * cy])


def _mesh(dim, n):
    if dim == 3:
        V, T = O.grid_tet_mesh(n, n, n)
        return V / n, T
    V, Q = O.gen_grid_2d(n, n)
    V, T = O.quad_tri_subdiv(V, Q)
    return V[:, :2] / n, T


def _boundary_vars(pos, dim):
    on = (np.abs(pos) < 1e-12).any(axis=1) | (np.abs(pos - 1.0) < 1e-12).any(axis=1)
    nodes = np.flatnonzero(on)
    vars_ = (dim * nodes[:, None] + np.arange(dim)[None, :]).ravel()
    return vars_, _exact(pos[nodes]).ravel()


def _oracle_error(dim, deg, n):
    V, T = _mesh(dim, n)
    sim = O.Simulator(T, V, deg)
    sim.set_material_constant(O.ElasticityTensor.isotropic(dim, 200.0, 0.35 if dim == 3 else 0.3))
    K = sim.assembleStiffnessMatrix()
    sysm = O.SPSDSystem(K)
    vars_, vals = _boundary_vars(sim.mesh.node_pos, dim)
    sysm.fix_variables(vars_, vals)
    u = sysm.solve(np.zeros(dim * sim.mesh.num_nodes)).reshape(-1, dim)
    return np.abs(u - _exact(sim.mesh.node_pos)).max()


def test_oracle_converges_at_the_expected_rates_2d():
    # plane stress is not the 3D Navier operator, but u = grad(harmonic) is divergence-free with harmonic components, which
    # solves the plane-stress equations as well (they have the same form with lambda replaced by lambda*)
    e1 = [_oracle_error(2, 1, n) for n in (4, 8, 16)]
    e2 = [_oracle_error(2, 2, n) for n in (2, 4, 8)]
    assert e1[0] / e1[1] > 3.0 and e1[1] / e1[2] > 3.3
    assert e2[0] / e2[1] > 6.0 and e2[1] / e2[2] > 6.5
    assert e2[2] < 1e-4 and e1[2] < 1e-2


def _hip_error(dim, deg, n):
    import meshfem_amd as M
    V, T = _mesh(dim, n)
    c = M.Context(0)
    c.mesh_build(T, V, deg)
    c.material_isotropic(200.0, 0.35 if dim == 3 else 0.3)
    c.assemble()
    pos = c.node_positions()
    vars_, vals = _boundary_vars(pos, dim)
    c.fix_variables(vars_, vals)
    u = c.solve(np.zeros(dim * c.n_node), rtol=1e-12).reshape(-1, dim)
    assert c.last_info["converged"]
    return np.abs(u - _exact(pos)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_hip_path_converges_at_the_expected_rates(dim):
    n1, n2 = ((8, 16, 32), (4, 8, 16)) if dim == 3 else ((16, 32, 64), (8, 16, 32))
    e1 = [_hip_error(dim, 1, n) for n in n1]
    e2 = [_hip_error(dim, 2, n) for n in n2]
    assert e1[0] / e1[1] > 3.3 and e1[1] / e1[2] > 3.6, e1          # O(h^2)
    assert e2[0] / e2[1] > 6.5 and e2[1] / e2[2] > 7.0, e2          # O(h^3) at the nodes (superconvergent on these grids: >= 8)
    # same discretisation as the oracle at a size it solves directly
    assert abs(_hip_error(dim, 2, 4) - _oracle_error(dim, 2, 4)) < 1e-9
