"""Generates the committed golden fixtures (run from the repo root: python tests/golden/make_goldens.py).

1. quadrature_monomials.json -- the exact integrals of the monomial bases over the unit-volume
   K-simplex that the reference's tests/test_quadrature.cc:59-60,83-89,127-132 holds (data only:
   values K! a! b! c! / (a+b+c+K)!, same ordering as the reference's function tables).
2. ke_exact.json -- exact (sympy rational arithmetic) element stiffness matrices for P1/P2 tets and
   tris with rational vertices and rational isotropic / orthotropic D: independent of any
   quadrature rule, anchors A6 of SURVEY.md.
3. cantilever_small.npz -- inputs and oracle outputs (upper K triplets after sumRepeated, load,
   Dirichlet variables, direct-solve displacements) of a 5x2x2 cantilever, P1 and P2, produced by
   oracle/meshfem_oracle.py. The reference itself cannot be run here (Eigen/SuiteSparse absent), so
   this fixture pins the ORACLE's output across refactors; it is not captured reference output.
4. example_meshes.npz -- the same for the reference's unstructured example meshes (data files copied from
   examples/meshes into tests/golden/meshes: cube_cross.msh, ball.msh, 2D_microstructure.msh, and the orthotropic base
   cells 2D_microstructure_orthocell.msh, 3D_microstructure_orthocell.msh):
   direct-solve displacements under box Dirichlet / traction conditions (P1, P2) and periodic
   homogenization results Ch, w_ij (cube_cross 3D, 2D_microstructure 2D). Oracle output, not reference output.
"""
import json
import os
import sys
from fractions import Fraction
from math import factorial

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def monomial_tables():
    # exponent tuples in the order of the reference's function tables (u, v, w = first barycentric coords)
    t1 = [[(0,)], [(1,)], [(2,)], [(3,)], [(4,)]]
    t2 = [[(0, 0)], [(0, 1), (1, 0)], [(0, 2), (1, 1), (2, 0)], [(0, 3), (1, 2), (2, 1), (3, 0)],
          [(0, 4), (1, 3), (2, 2), (3, 1), (4, 0)], [(0, 5), (1, 4), (2, 3), (3, 2), (4, 1), (5, 0)]]
    # 3D: (u,v,w) exponents; reference order per degree: w^d ... u^d  (test_quadrature.cc:91-126)
    t3 = [[(0, 0, 0)],
          [(0, 0, 1), (0, 1, 0), (1, 0, 0)],
          [(0, 0, 2), (0, 1, 1), (0, 2, 0), (1, 0, 1), (1, 1, 0), (2, 0, 0)],
          [(0, 0, 3), (0, 1, 2), (0, 2, 1), (0, 3, 0), (1, 0, 2), (1, 1, 1), (1, 2, 0), (2, 0, 1), (2, 1, 0), (3, 0, 0)],
          [(0, 0, 4), (0, 1, 3), (0, 2, 2), (0, 3, 1), (0, 4, 0), (1, 0, 3), (1, 1, 2), (1, 2, 1), (1, 3, 0), (2, 0, 2),
           (2, 1, 1), (2, 2, 0), (3, 0, 1), (3, 1, 0), (4, 0, 0)]]
    out = {}
    for K, tab in ((1, t1), (2, t2), (3, t3)):
        rows = []
        for deg_row in tab:
            r = []
            for ex in deg_row:
                num = factorial(K)
                for a in ex:
                    num *= factorial(a)
                val = Fraction(num, factorial(sum(ex) + K))
                r.append(dict(exponents=list(ex), value=str(val)))
            rows.append(r)
        out[str(K)] = rows
    return out


def exact_ke(K, deg, verts, D):
    """Exact Ke via sympy: shape functions in barycentric coords, exact simplex integration."""
    import sympy as sp
    from oracle import meshfem_oracle as O
    N = K
    nv = K + 1
    lam = sp.symbols("l0:%d" % nv)
    P = sp.Matrix(verts)
    # gradients of barycentric coordinates: solve [1 x^T] system
    A = sp.Matrix([[1] + list(P.row(k)) for k in range(nv)])
    Ainv = A.inv()            # column k of Ainv: coefficients of lambda_k = a + b.x
    gl = [sp.Matrix([Ainv[1 + a, k] for a in range(N)]) for k in range(nv)]
    vol = abs(A.det()) / factorial(K)
    if deg == 1:
        phi = list(lam)
    else:
        phi = [2 * l * (l - sp.Rational(1, 2)) for l in lam]
        phi += [4 * lam[O.EDGE_START[e]] * lam[O.EDGE_END[e]] for e in range(O.num_edges(K))]
    grads = [sum((sp.diff(p, lam[k]) * gl[k] for k in range(nv)), sp.zeros(N, 1)) for p in phi]

    def integrate(poly):
        poly = sp.Poly(sp.expand(poly), *lam)
        tot = 0
        for mon, coef in poly.terms():
            num = factorial(K)
            for a in mon:
                num *= factorial(a)
            tot += coef * sp.Rational(num, factorial(sum(mon) + K))
        return tot * vol
    Dm = sp.Matrix(D)
    n = len(phi)
    Ke = sp.zeros(n * N, n * N)
    for i in range(n):
        for j in range(n):
            for c in range(N):
                for d in range(N):
                    expr = 0
                    for a in range(N):
                        for b in range(N):
                            expr += grads[i][a] * Dm[O.flatten_indices(N, a, c), O.flatten_indices(N, d, b)] * grads[j][b]
                    Ke[i * N + c, j * N + d] = integrate(expr)
    return np.array(Ke.tolist(), dtype=object), vol


def ke_fixtures():
    import sympy as sp
    R = sp.Rational
    lam, mu = R(7, 3), R(5, 4)
    iso3 = sp.zeros(6, 6)
    for i in range(3):
        for j in range(3):
            iso3[i, j] = lam
        iso3[i, i] = lam + 2 * mu
    for k in range(3, 6):
        iso3[k, k] = mu
    ort3 = sp.Matrix([[R(11, 2), R(3, 2), R(1), 0, 0, 0], [R(3, 2), R(7), R(2), 0, 0, 0], [R(1), R(2), R(9, 2), 0, 0, 0],
                      [0, 0, 0, R(5, 3), 0, 0], [0, 0, 0, 0, R(2), 0], [0, 0, 0, 0, 0, R(7, 4)]])
    iso2 = sp.Matrix([[lam + 2 * mu, lam, 0], [lam, lam + 2 * mu, 0], [0, 0, mu]])
    ort2 = sp.Matrix([[R(11, 2), R(3, 2), 0], [R(3, 2), R(7), 0], [0, 0, R(5, 3)]])
    tets = [[[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]],
            [[R(1, 3), R(1, 5), 0], [R(7, 4), R(1, 2), R(1, 7)], [R(2, 3), R(9, 5), R(1, 3)], [R(1, 2), R(3, 4), R(11, 6)]]]
    tris = [[[0, 0], [1, 0], [0, 1]], [[R(1, 3), R(1, 5)], [R(7, 4), R(1, 2)], [R(2, 3), R(9, 5)]]]
    cases = []
    for K, geoms, mats in ((3, tets, (("iso", iso3), ("ortho", ort3))), (2, tris, (("iso", iso2), ("ortho", ort2)))):
        for gi, g in enumerate(geoms):
            for deg in (1, 2):
                for mname, Dm in mats:
                    Ke, vol = exact_ke(K, deg, g, Dm)
                    cases.append(dict(K=K, deg=deg, material=mname, geom=gi,
                                      verts=[[float(x) for x in row] for row in g],
                                      D=[[float(x) for x in row] for row in Dm.tolist()],
                                      vol=float(vol), Ke=[[float(x) for x in row] for row in Ke]))
                    print("exact Ke: K=%d deg=%d %s geom %d" % (K, deg, mname, gi), flush=True)
    return cases


def cantilever_fixture():
    from oracle import meshfem_oracle as O
    V, T = O.grid_tet_mesh(5, 2, 2)
    out = dict(V=V, T=T)
    for deg in (1, 2):
        sim = O.Simulator(T, V, deg)
        sim.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
        mn, mx = sim.box_percent([-1e-4] * 3, [1e-4, 1.0001, 1.0001]); sim.apply_dirichlet_box(mn, mx, [0, 0, 0])
        mn, mx = sim.box_percent([0.9999, -1e-4, -1e-4], [1.0001, 1.0001, 1.0001]); sim.apply_neumann_box(mn, mx, [0, -10, 0], "force")
        Kt = sim.assembleStiffnessMatrix().sum_repeated()
        fv, fx = sim.dirichlet_vars_and_values()
        out.update({"p%d_elem_nodes" % deg: sim.mesh.elem_nodes, "p%d_bdry_elem_nodes" % deg: sim.mesh.bdry_elem_nodes,
                    "p%d_bdry_nodes" % deg: sim.mesh.bdry_nodes, "p%d_K_i" % deg: Kt.i, "p%d_K_j" % deg: Kt.j, "p%d_K_v" % deg: Kt.v,
                    "p%d_load" % deg: sim.neumannLoad(), "p%d_fixed_vars" % deg: np.array(fv), "p%d_u" % deg: sim.solve()})
    return out


def example_mesh_fixture():
    from oracle import meshfem_oracle as O
    from meshfem_amd import mesh_io            # host-only MSH reader (no device code involved)
    out = {}
    for name, lo_box, hi_box, trac in (("cube_cross", ([-1e-3] * 3, [0.02, 1.001, 1.001]), ([0.98, -1e-3, -1e-3], [1.001] * 3), [0, -1, 0]),
                                       ("ball", ([-1e-3] * 3, [1.001, 1.001, 0.12]), ([-1e-3, -1e-3, 0.88], [1.001] * 3), [0.3, 0, -1])):
        V, E, _ = mesh_io.load_msh(os.path.join(HERE, "meshes", name + ".msh"))
        for deg in (1, 2):
            sim = O.Simulator(E, V, deg)
            sim.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
            mn, mx = sim.box_percent(*lo_box); sim.apply_dirichlet_box(mn, mx, [0, 0, 0])
            mn, mx = sim.box_percent(*hi_box); sim.apply_neumann_box(mn, mx, trac, "traction")
            fv, _ = sim.dirichlet_vars_and_values()
            load = sim.neumannLoad()
            assert len(fv) >= 9 and np.abs(load).sum() > 0, (name, len(fv))
            Kt = sim.assembleStiffnessMatrix().sum_repeated()
            out.update({"%s_p%d_u" % (name, deg): sim.solve(), "%s_p%d_load" % (name, deg): load,
                        "%s_p%d_fixed_vars" % (name, deg): np.array(fv), "%s_p%d_K_nnz" % (name, deg): np.array([Kt.nnz(), int((np.abs(Kt.v) > 1e-12 * np.abs(Kt.v).max()).sum())]),
                        "%s_p%d_K_checksum" % (name, deg): np.array([Kt.v.sum(), np.abs(Kt.v).sum(), (Kt.v * (1 + Kt.i % 7) * (1 + Kt.j % 5)).sum()])})
            print(name, deg, "fixed", len(fv), "nnz", Kt.nnz(), flush=True)
    for name, dim in (("cube_cross", 3), ("2D_microstructure", 2)):
        V, E, _ = mesh_io.load_msh(os.path.join(HERE, "meshes", name + ".msh"))
        for deg in (1, 2):
            sim = O.Simulator(E, V[:, :dim], deg)
            sim.set_material_constant(O.ElasticityTensor.isotropic(dim, 200.0, 0.35))
            w = O.solve_cell_problems(sim)
            Ch = O.homogenized_elasticity_tensor(sim, w)
            out.update({"%s_hom_p%d_Ch" % (name, deg): Ch, "%s_hom_p%d_w" % (name, deg): np.array(w),
                        "%s_hom_p%d_ndof" % (name, deg): np.array([sim.numDoFs()])})
            print(name, "homogenization", deg, np.round(np.diag(Ch), 4), flush=True)
    # orthotropic base cells of the reference (OrthotropicHomogenization.hh route, displacement form like the binding)
    for name, dim, degs in (("2D_microstructure_orthocell", 2, (1, 2)), ("3D_microstructure_orthocell", 3, (1,))):
        V, E, _ = mesh_io.load_msh(os.path.join(HERE, "meshes", name + ".msh"))
        for deg in degs:
            sim = O.Simulator(E, V[:, :dim], deg)
            sim.set_material_constant(O.ElasticityTensor.isotropic(dim, 200.0, 0.35))
            w = O.solve_cell_problems_orthotropic(sim)
            Ch = O.homogenized_tensor_from_ortho_cell_quantity(dim, O.homogenized_elasticity_tensor_displacement_form(sim, w))
            out.update({"%s_hom_p%d_Ch" % (name, deg): Ch, "%s_hom_p%d_w_norms" % (name, deg): np.array([np.linalg.norm(x) for x in w])})
            print(name, "orthotropic-cell homogenization", deg, np.round(np.diag(Ch), 4), flush=True)
    return out


if __name__ == "__main__":
    if "--examples-only" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "example_meshes.npz"), **example_mesh_fixture())
        sys.exit(0)
    with open(os.path.join(HERE, "quadrature_monomials.json"), "w") as f:
        json.dump(monomial_tables(), f, indent=0)
    np.savez_compressed(os.path.join(HERE, "cantilever_small.npz"), **cantilever_fixture())
    np.savez_compressed(os.path.join(HERE, "example_meshes.npz"), **example_mesh_fixture())
    with open(os.path.join(HERE, "ke_exact.json"), "w") as f:
        json.dump(ke_fixtures(), f)
    print("done")
