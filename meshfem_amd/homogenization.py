"""Periodic homogenization on the GPU path: mirror of `periodic_homogenization.homogenize`
(src/python_bindings/periodic_homogenization.cc:36-90,159-171) = solveCellProblems
(src/lib/MeshFEM/PeriodicHomogenization.hh:34-54) + homogenizedElasticityTensor (:72-103)."""
import numpy as np

from .core import flat_len
from .linear_elasticity import Simulator


def canonical_strain_flat(dim, k):
    """SymmetricMatrix CanonicalBasis(k) flattened: 1 on the diagonal entries, 1/2 for shear
    (SymmetricMatrix.hh:405-413)."""
    v = np.zeros(flat_len(dim))
    v[k] = 1.0 if k < dim else 0.5
    return v


def _unflatten(dim, v):
    M = np.zeros((dim, dim))
    idx = {2: [(0, 0), (1, 1), (0, 1)], 3: [(0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1)]}[dim]
    for k, (i, j) in enumerate(idx):
        M[i, j] = M[j, i] = v[k]
    return M


def periodic_dofs_from_file(path, num_nodes):
    """PeriodicCondition(mesh, pcFile) (BoundaryConditions.hh:565-613): the file lists pairs of identified NODE indices; every
    connected component of that graph shares one DoF, numbered in node order. Returns (dofForNode, numDoFs)."""
    adj = [[] for _ in range(num_nodes)]
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 2:
                continue
            a, b = int(t[0]), int(t[1])
            if not (0 <= a < num_nodes and 0 <= b < num_nodes):
                raise RuntimeError("periodic vertex file: node index out of range")
            adj[a].append(b); adj[b].append(a)
    dof = np.full(num_nodes, -1, dtype=np.int32)
    nd = 0
    for n in range(num_nodes):
        if dof[n] >= 0:
            continue
        dof[n] = nd
        queue = [n]
        while queue:
            u = queue.pop()
            for v in adj[u]:
                if dof[v] < 0:
                    dof[v] = nd
                    queue.append(v)
        nd += 1
    return dof, nd


def solve_cell_problems(sim: Simulator, cell_epsilon=1e-7, ignore_periodic_mismatch=False, manual_periodic_vertices_file=""):
    """== solveCellProblems: periodic DoFs + pinned node, one solve per canonical strain with
    rhs = constantStrainLoad(-e_ij). The matrix is assembled once; the 3/6 right-hand sides reuse it."""
    if manual_periodic_vertices_file:                       # PeriodicHomogenization_cli.cc:101-102
        sim.ctx.dof_map(*periodic_dofs_from_file(manual_periodic_vertices_file, sim.numNodes()))
    else:
        sim.applyPeriodicConditions(cell_epsilon, ignoreMismatch=ignore_periodic_mismatch)
    sim.applyNoRigidMotionConstraint()
    sim.setUsePinNoRigidTranslationConstraint(True)
    # one Simulator, one system, 3 / 6 right-hand sides (the reference factors once and back-substitutes per load, :47-53): handed over together,
    # so that the preconditioner's coarse levels serve all of them at once and the load vectors are formed where they are used
    w = sim.solveConstantStrainLoads([-canonical_strain_flat(sim.N, k) for k in range(flat_len(sim.N))])
    return w, list(sim.infos)


def homogenized_elasticity_tensor(sim: Simulator, w_ij, base_cell_volume=0.0):
    """== homogenizedElasticityTensor (stress-like form, PeriodicHomogenization.hh:72-100): Eh.DRow(i) = 1/|Y| sum_e vol_e
    [E_e : avg strain(w_i) + E_e.DRow(i)] = 1/|Y| sum_e vol_e E_e : (avg strain(w_i) + e_i): one device reduction per probe
    strain (mfh_integrated_stress); no per-element field and no affine displacement field visit the host."""
    N, fl = sim.N, flat_len(sim.N)
    if base_cell_volume == 0.0:
        mn, mx = sim.boundingBox()
        base_cell_volume = float(np.prod(mx - mn))
    Eh = np.zeros((fl, fl))
    for i in range(fl):
        Eh[i, :] = sim.ctx.integrated_stress(w_ij[i], canonical_strain_flat(N, i)) / base_cell_volume
    return Eh


def _bdry_shape_integrals(K, deg):
    """Integrals of the boundary element's nodal shape functions over a unit-volume (K-1)-simplex
    (Interpolant::integrate, Functions.hh:246-318): P1 equal weights; P2 edge 1/6,1/6,4/6; P2 triangle 0 at
    the vertices, 1/3 at the edge nodes."""
    if deg == 1:
        return np.full(K, 1.0 / K)
    return np.array([1 / 6.0, 1 / 6.0, 4 / 6.0]) if K == 2 else np.array([0, 0, 0, 1 / 3.0, 1 / 3.0, 1 / 3.0])


def homogenized_elasticity_tensor_displacement_form(sim: Simulator, w_ij, base_cell_volume=0.0):
    """== homogenizedElasticityTensorDisplacementForm (PeriodicHomogenization.hh:146-186), the form the
    reference's Python binding returns: Eh.DRow(i) = 1/|Y| [E : sum_be sym(int_be w_i (x) n) + E vol(omega)].
    Assumes a constant base tensor (element 0's), like the reference. A boundary sum over tens of thousands of
    faces at most: evaluated on the host from the device solution."""
    N, fl = sim.N, flat_len(sim.N)
    c = sim.ctx
    D = c.material_get(0)
    ben = c.boundary_elem_nodes()
    area, nrm = c.boundary_elem_geometry()
    wts = _bdry_shape_integrals(N, sim.degree)
    pos = sim.nodes()
    if base_cell_volume == 0.0:
        base_cell_volume = float(np.prod(sim.boundingBox()[1] - sim.boundingBox()[0]))
    idx = {2: [(0, 0), (1, 1), (0, 1)], 3: [(0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1)]}[N]
    dbl = np.ones(fl)
    dbl[N:] = 2.0
    Eh = np.zeros((fl, fl))
    for i in range(fl):
        w_int = np.einsum("k,bkc->bc", wts, w_ij[i][ben]) * area[:, None]          # int_be w
        nw = 0.5 * (w_int[:, :, None] * nrm[:, None, :] + nrm[:, :, None] * w_int[:, None, :]).sum(axis=0)
        nw_flat = np.array([nw[a, b] for a, b in idx])
        Eh[i, :] = D @ (nw_flat * dbl)                                               # doubleContract (ElasticityTensor.hh:437-449)
    Eh += D * float(c.elem_volumes().sum())
    return Eh / base_cell_volume


# ---- orthotropic base cell (OrthotropicHomogenization.hh): 1/4 (2D) or 1/8 (3D) of the period cell
def ortho_cell_fixed_vars(sim: Simulator, cell_epsilon=1e-7):
    """Fixed-variable sets of Orthotropic::solveCellProblems (OrthotropicHomogenization.hh:84-136): element 0 for the N
    stretch probes (w_c = 0 on the reflection planes with normal e_c), then one per shear probe s (3D: always the
    component perpendicular to the shear plane; for planes c != s also the component that is neither c nor s; 2D: the
    component other than c). Variables are N * node + c."""
    N, fl = sim.N, flat_len(sim.N)
    pos = sim.nodes()
    mn, mx = sim.boundingBox()
    on = (np.abs(pos - mn) <= cell_epsilon) | (np.abs(pos - mx) <= cell_epsilon)     # FaceMembership::onMinOrMaxFace
    sets = [np.flatnonzero(on.ravel())]
    for s in range(fl - N):
        fix = np.zeros_like(on)
        for c in range(N):
            if N == 3:
                fix[:, s] |= on[:, c]
                if c != s:
                    fix[:, N - (c + s)] |= on[:, c]
            else:
                fix[:, 1 if c == 0 else 0] |= on[:, c]
        sets.append(np.flatnonzero(fix.ravel()))
    return sets


def solve_cell_problems_orthotropic(sim: Simulator, cell_epsilon=1e-7):
    """== Orthotropic::solveCellProblems (OrthotropicHomogenization.hh:44-153): no periodicity, no rigid-motion rows;
    symmetry planes fix components instead. The reference builds 1 + (flatLen - N) copies of K in separate SPSDSystems;
    here the operator is assembled once and only the fixed-variable mask (and the diagonal blocks derived from it)
    changes between the probes."""
    sim.removePeriodicConditions()
    sim.removeNoRigidMotionConstraint()
    N, c = sim.N, sim.ctx
    c.assemble()
    loads = [sim.constantStrainLoad(-canonical_strain_flat(N, k)) for k in range(flat_len(N))]
    sets = ortho_cell_fixed_vars(sim, cell_epsilon)
    w, infos = [None] * flat_len(N), [None] * flat_len(N)
    for si, fv in enumerate(sets):
        probes = range(N) if si == 0 else [N + si - 1]
        c.clear_fixed()
        c.fix_variables(fv, np.zeros(len(fv)))
        for k in probes:
            w[k] = c.solve(loads[k].ravel(), rtol=sim.rtol, maxit=sim.maxit).reshape(-1, N)
            infos[k] = dict(c.last_info)
    c.clear_fixed()
    return w, infos


def fluctuation_displacement_sign(N, ij, r):
    """OrthotropicHomogenization.hh:161-174: sign of probe ij's fluctuation under the reflection with bit mask r."""
    if ij < N:
        return 1.0
    bits = [(r >> b) & 1 for b in range(N)]
    if N == 3:
        bits[ij - N] = 0
    return -1.0 if sum(bits) == 1 else 1.0


def homogenized_tensor_from_ortho_cell_quantity(N, EhO):
    """OrthotropicHomogenization.hh:183-198: average of the sub-cell quantity over the 2^N reflections."""
    fl = flat_len(N)
    sg = np.array([[fluctuation_displacement_sign(N, ij, r) for ij in range(fl)] for r in range(1 << N)])
    Eh = np.triu(np.einsum("ri,rj->ij", sg, sg) * np.asarray(EhO)) / (1 << N)
    return Eh + np.triu(Eh, 1).T


def homogenize_orthotropic_cell(sim: Simulator, base_cell_volume=0.0, form="displacement"):
    """Orthotropic::homogenizedElasticityTensor{DisplacementForm} (:200-216) on the simulator of the orthotropic base
    cell. Returns (Ch, w_ij, infos)."""
    w, infos = solve_cell_problems_orthotropic(sim)
    f = homogenized_elasticity_tensor_displacement_form if form == "displacement" else homogenized_elasticity_tensor
    return homogenized_tensor_from_ortho_cell_quantity(sim.N, f(sim, w, base_cell_volume)), w, infos


def homogenized_elasticity_tensor_energy_form(sim: Simulator, w_ij, base_cell_volume=0.0):
    """Ch_ijkl = 1/|Y| int (e^ij + eps(w^ij)) : C : (e^kl + eps(w^kl)) dV, one device reduction per tensor entry.
    Equal to the stress-like and displacement forms at the cell-problem solutions."""
    if base_cell_volume == 0.0:
        mn, mx = sim.boundingBox()
        base_cell_volume = float(np.prod(mx - mn))
    return sim.ctx.mutual_energies(w_ij) / base_cell_volume


def delta_fluctuation_displacements(sim: Simulator, w_ij, delta_p):
    """== deltaFluctuationDisplacements (PeriodicHomogenization.hh:527-544): change of the cell-problem solutions
    under the vertex perturbation delta_p: K dw = deltaConstantStrainLoad(-e_ij) - (delta K) w_ij, solved with the
    constraints of the cell problems (the simulator left by solve_cell_problems)."""
    out = []
    for k in range(flat_len(sim.N)):
        rhs = sim.deltaConstantStrainLoad(-canonical_strain_flat(sim.N, k), delta_p)
        rhs -= sim.applyDeltaStiffnessMatrix(w_ij[k], delta_p)
        out.append(sim.solve(rhs))
    return out


def delta_homogenized_elasticity_tensor(sim: Simulator, w_ij, delta_p, base_cell_volume=0.0):
    """Change of Ch under delta_p in the volume form quoted at PeriodicHomogenization.hh:484-491 -- the exact derivative of
    the discrete Ch; like the reference, the periodic cell volume |Y| is held fixed. The reference's own
    deltaHomogenizedElasticityTensor (:492-514) evaluates the continuous boundary form instead:
    `delta_homogenized_elasticity_tensor_boundary_form` reproduces that value."""
    if base_cell_volume == 0.0:
        mn, mx = sim.boundingBox()
        base_cell_volume = float(np.prod(mx - mn))
    return sim.ctx.mutual_energies(w_ij, delta_p) / base_cell_volume


def _bdry_gradient_weights(N, deg):
    """W[a, n] = int over the unit-volume boundary simplex of lambda_a phi_n, phi the nodal basis of the degree 2 (deg - 1)
    interpolant homogenizedElasticityTensorGradient returns (exact; the reference integrates with a quadrature rule of
    degree 1 + GDeg, PeriodicHomogenization.hh:508-511)."""
    from math import factorial

    def mono(alpha):
        num = factorial(N - 1)
        for x in alpha:
            num *= factorial(x)
        return num / factorial(sum(alpha) + N - 1)
    if deg == 1:
        return np.full((N, 1), 1.0 / N)
    edges = [(0, 1)] if N == 2 else [(0, 1), (1, 2), (2, 0)]
    W = np.zeros((N, N + len(edges)))
    for a in range(N):
        for n in range(N):
            a2 = [0] * N; a2[n] += 2; a2[a] += 1
            a1 = [0] * N; a1[n] += 1; a1[a] += 1
            W[a, n] = 2 * mono(a2) - mono(a1)
        for k, (s, t) in enumerate(edges):
            al = [0] * N; al[s] += 1; al[t] += 1; al[a] += 1
            W[a, N + k] = 4 * mono(al)
    return W


def homogenized_elasticity_tensor_gradient(sim: Simulator, w_ij):
    """== homogenizedElasticityTensorGradient (PeriodicHomogenization.hh:226-288): the steepest-ascent normal velocity of
    every component of Ch as a per-boundary-element interpolant of degree 2 (Deg - 1): the nodal values
    G_ijkl = 1/|bbox| (e_ij + eps(w_ij)) : E : (e_kl + eps(w_kl)) at the boundary element's nodes, zero on the periodic
    (internal) boundary. Returns [nBE, 1 | npbe, flatLen, flatLen]. The strains and stresses at the boundary corners come
    from the device (mfh_boundary_strain_field); the flatLen^2 contraction over a few 10^4 faces is done here."""
    N, fl = sim.N, flat_len(sim.N)
    c = sim.ctx
    pos = sim.nodes()
    bbox_vol = float(np.prod(sim.boundingBox()[1] - sim.boundingBox()[0]))
    dbl = np.ones(fl)
    dbl[N:] = 2.0
    G = np.stack([c.boundary_strain_field(w_ij[k]) + canonical_strain_flat(N, k)[None, None, :] for k in range(fl)])      # [fl, nBE, nq, fl]
    # stress(w) + E : e  ==  E : G (the device applies each element's own tensor)
    S = np.stack([c.boundary_strain_field(w_ij[k] + pos @ _unflatten(N, canonical_strain_flat(N, k)).T, stress=True)
                  for k in range(fl)])
    if sim.degree == 2:                  # the linear interpolants at the boundary edge midpoints (0,1),(1,2),(2,0)
        edges = [(0, 1)] if N == 2 else [(0, 1), (1, 2), (2, 0)]
        mid = lambda X: np.concatenate([X] + [0.5 * (X[:, :, [s]] + X[:, :, [t]]) for s, t in edges], axis=2)
        G, S = mid(G), mid(S)
    out = np.einsum("ibnc,c,kbnc->bnik", S, dbl, G) / bbox_vol
    out = 0.5 * (out + out.transpose(0, 1, 3, 2))      # major symmetry; the reference fills the upper triangle only
    out[c.boundary_elem_internal().astype(bool)] = 0.0
    return out


def delta_homogenized_elasticity_tensor_boundary_form(sim: Simulator, w_ij, delta_p):
    """== deltaHomogenizedElasticityTensor (PeriodicHomogenization.hh:492-514) exactly as the reference evaluates it: the
    linear normal velocity n . delta_p of every boundary element integrated against homogenizedElasticityTensorGradient
    (the continuous, Eulerian shape derivative; it differs from the exact discrete derivative of
    `delta_homogenized_elasticity_tensor` by the discretisation error)."""
    c = sim.ctx
    sd = homogenized_elasticity_tensor_gradient(sim, w_ij)
    area, nrm = c.boundary_elem_geometry()
    ben = c.boundary_elem_nodes()[:, :sim.N]
    nsv = np.einsum("bc,bac->ba", nrm, np.asarray(delta_p, dtype=np.float64)[ben])
    W = _bdry_gradient_weights(sim.N, sim.degree)
    return np.einsum("b,ba,an,bnik->ik", area, nsv, W, sd)


def normal_shape_velocity(sim: Simulator, delta_p):
    """The linear normal velocity n . delta_p of every boundary element at its vertices (PeriodicHomogenization.hh:499-505): [nBE, N]"""
    _, nrm = sim.ctx.boundary_elem_geometry()
    ben = sim.ctx.boundary_elem_nodes()[:, :sim.N]
    return np.einsum("bc,bac->ba", nrm, np.asarray(delta_p, dtype=np.float64)[ben])


def _simplex_moments(d, order):
    """int over the unit-volume d-simplex of mu_a mu_b (order 2) / mu_a mu_b mu_c (order 3)"""
    from math import factorial
    n = d + 1
    out = np.zeros((n,) * order)
    for idx in np.ndindex(*out.shape):
        al = np.bincount(idx, minlength=n)
        out[idx] = factorial(d) * np.prod([factorial(int(x)) for x in al]) / factorial(order + d)
    return out


def change_in_div_tensor_load(sim: Simulator, vn, t, ignore_periodic_bdry=True):
    """== Simulator::changeInDivTensorLoad (LinearElasticity.hh:590-650): the per-DoF load -int_bdry vn strain(phi) : t dA
    for boundary tensor interpolants t [nBE, 1 | N, flatLen] and the linear normal velocity vn [nBE, N]. A sum over the
    boundary elements only (host): strain(phi_n e_c) : t = (t grad phi_n)_c with grad phi_n linear in the face's barycentric
    coordinates, integrated exactly through the simplex moments instead of a quadrature rule."""
    N, deg, c = sim.N, sim.degree, sim.ctx
    K = N
    area, _ = c.boundary_elem_geometry()
    ben = c.boundary_elem_nodes()[:, :N]
    parent = c.boundary_elem_parents().astype(np.int64)
    en = sim.elements()[parent]                                              # parent element nodes [nBE, npe]
    pos = sim.nodes()
    P = pos[en[:, :K + 1]]
    Minv = np.linalg.inv(np.transpose(P[:, :K] - P[:, [K]], (0, 2, 1)))     # rows: grad lambda_0..K-1
    gl = np.concatenate([Minv, -Minv.sum(axis=1, keepdims=True)], axis=1)   # [nBE, K+1, N]
    vn = np.asarray(vn, dtype=np.float64)
    tm = np.stack([np.stack([_unflatten(N, x) for x in row]) for row in np.asarray(t)])     # [nBE, nq, N, N]
    w = area.copy()
    if ignore_periodic_bdry:
        w[c.boundary_elem_internal().astype(bool)] = 0.0
    if deg == 1:
        contrib = np.einsum("e,e,eij,enj->eni", w, vn.mean(axis=1), tm[:, 0], gl)
    else:
        npe = en.shape[1]
        L = (en[:, :K + 1, None] == ben[:, None, :]).astype(np.float64)      # lambda_k = sum_c L[k, c] mu_c on the face
        A = np.zeros((len(en), npe, N))
        B = np.zeros((len(en), npe, K + 1, N))
        A[:, :K + 1] = -gl                                                   # grad phi_k = (4 lambda_k - 1) grad lambda_k
        for k in range(K + 1):
            B[:, k, k] = 4 * gl[:, k]
        es, et = (0, 1, 2, 0, 2, 1), (1, 2, 0, 3, 3, 3)                      # Simplex.hh:43-44
        for ei in range(npe - K - 1):                                        # grad phi_(s,t) = 4 (lambda_t grad lambda_s + lambda_s grad lambda_t)
            B[:, K + 1 + ei, et[ei]] = 4 * gl[:, es[ei]]
            B[:, K + 1 + ei, es[ei]] = 4 * gl[:, et[ei]]
        Bm = np.einsum("enkj,ekc->encj", B, L)
        M2, T3 = _simplex_moments(N - 1, 2), _simplex_moments(N - 1, 3)
        contrib = np.einsum("e,ea,ebij,ab,enj->eni", w, vn, tm, M2, A, optimize=True) \
            + np.einsum("e,ea,ebij,abc,encj->eni", w, vn, tm, T3, Bm, optimize=True)
    dm, nd = c.get_dof_map()
    load = np.zeros((nd, N))
    np.subtract.at(load, dm[en], contrib)
    return load


def fluctuation_displacement_shape_derivatives(sim: Simulator, w_ij, vn, project_out_normal_stress=False):
    """== fluctuationDisplacementShapeDerivatives (PeriodicHomogenization.hh:301-370): the Eulerian shape derivative of
    every fluctuation displacement under the normal boundary velocity vn [nBE, N] (`normal_shape_velocity`), the "direct"
    approach: cell problems loaded with -int_bdry vn strain(phi) : C : [strain(w^kl) + e^kl] dA. The boundary stresses come
    from the device (mfh_boundary_strain_field), the solves reuse the assembled operator and its preconditioner."""
    N, c = sim.N, sim.ctx
    pos = sim.nodes()
    _, nrm = c.boundary_elem_geometry()
    internal = c.boundary_elem_internal().astype(bool)
    out = []
    for k in range(len(w_ij)):
        st = c.boundary_strain_field(w_ij[k] + pos @ _unflatten(N, canonical_strain_flat(N, k)).T, stress=True)
        st[internal] = 0.0
        if project_out_normal_stress:                                        # s - (sn) n^T - n (sn)^T + (n^T s n) n n^T (:343-355)
            sm = np.stack([np.stack([_unflatten(N, x) for x in row]) for row in st])
            sn = np.einsum("bqij,bj->bqi", sm, nrm)
            nsn = np.einsum("bqi,bi->bq", sn, nrm)
            sm = sm - sn[..., :, None] * nrm[:, None, None, :] - nrm[:, None, :, None] * sn[..., None, :] \
                + nsn[..., None, None] * (nrm[:, None, :, None] * nrm[:, None, None, :])
            idx = {2: [(0, 0), (1, 1), (0, 1)], 3: [(0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1)]}[N]
            st = np.stack([sm[..., a, b] for a, b in idx], axis=-1)
            st[internal] = 0.0
        out.append(sim.solve(change_in_div_tensor_load(sim, vn, st, True)))
    return out


def homogenized_elasticity_tensor_discrete_differential(sim: Simulator, w_ij, base_cell_volume=0.0, full=False):
    """== homogenizedElasticityTensorDiscreteDifferential (PeriodicHomogenization.hh:372-480): the exact derivative of Ch
    with respect to every vertex coordinate (|Y| held fixed), all directions in one element sweep per tensor entry.
    Returns [nPairs, nVert, N] over the upper triangle ij <= kl (row-major), or with full=True the reference's
    OneForm layout [nVert, N, flatLen, flatLen]."""
    if base_cell_volume == 0.0:
        mn, mx = sim.boundingBox()
        base_cell_volume = float(np.prod(mx - mn))
    d = sim.ctx.mutual_energy_differential(w_ij) / base_cell_volume
    if not full:
        return d
    fl = flat_len(sim.N)
    out = np.empty((d.shape[1], sim.N, fl, fl))
    iu = np.triu_indices(fl)
    out[:, :, iu[0], iu[1]] = np.transpose(d, (1, 2, 0))
    out[:, :, iu[1], iu[0]] = np.transpose(d, (1, 2, 0))
    return out


def delta_homogenized_compliance_tensor(sim: Simulator, w_ij, delta_p, base_cell_volume=0.0):
    """== deltaHomogenizedComplianceTensor (:516-524): -Sh : dCh : Sh as flattened matrices with the shear-doubling
    of doubleDoubleContract."""
    Ch = homogenized_elasticity_tensor_energy_form(sim, w_ij, base_cell_volume)
    dCh = delta_homogenized_elasticity_tensor(sim, w_ij, delta_p, base_cell_volume)
    fl = flat_len(sim.N)
    dbl = np.ones(fl)
    dbl[sim.N:] = 2.0
    # compliance in the same flattened-tensor storage: Sh = inverse of the rank-4 map  (ElasticityTensor.hh inverse())
    Sh = np.linalg.inv(Ch * dbl[None, :]) / dbl[None, :]
    return -(Sh * dbl[None, :]) @ dCh @ (dbl[:, None] * Sh)


def delta_macro_strain_to_micro_strain_tensors(sim: Simulator, w_ij, delta_w_ij, delta_p):
    """== deltaMacroStrainToMicroStrainTensors (:549-563): per element, column kl of delta G is
    deltaAverageStrainField(w_kl, delta w_kl, delta_p). Returns [nElem, flatLen (ij), flatLen (kl)]."""
    cols = [sim.deltaAverageStrainField(w_ij[k], delta_w_ij[k], delta_p) for k in range(flat_len(sim.N))]
    return np.stack(cols, axis=2)


def homogenize(vertices, elements, degree=2, Cbase=None, E=None, nu=None, ortho_params=None, device=0, rtol=1e-8,
              preconditioner=None):
    """Returns dict(Ch, w_ij, strain_w_ij, iterations) like the reference's `homogenize`.
    Material: `Cbase` (flattened D or an object with .D) for a homogeneous base material, or
    per-element `E`/`nu`, or per-element orthotropic parameters."""
    sim = Simulator(elements, vertices, degree, device)
    sim.rtol = rtol
    if preconditioner is not None:
        sim.ctx.set_preconditioner(preconditioner)
    if Cbase is not None:
        sim.setMaterial(Cbase)
    elif ortho_params is not None:
        sim.setOrthotropicField(ortho_params)
    elif E is not None:
        if np.ndim(E) == 0:
            sim.setIsotropicMaterial(E, nu)
        else:
            sim.setIsotropicField(E, nu)
    w, infos = solve_cell_problems(sim)
    Ch = homogenized_elasticity_tensor(sim, w)
    return dict(Ch=Ch, w_ij=w, strain_w_ij=[sim.averageStrainField(x) for x in w],
                iterations=[i["iterations"] for i in infos], infos=infos, sim=sim)
