"""PeriodicHomogenization_cli-compatible driver on the GPU path (mirror of src/bin/PeriodicHomogenization_cli.cc):

    python -m meshfem_amd.periodic_homogenization_cli mesh.msh [-m material.json] [-d 2] [-O]
        [-M m2mstress.txt] [-o fields.msh] [-c] [--distanceToIsotropy] [--distanceToMaterial mat.json]
        [--device 0] [--rtol 1e-10] [--preconditioner auto|multigrid|two_level|block_jacobi]

Solves the cell problems of a periodic tri/tet cell (or of its orthotropic base cell with -O,
OrthotropicHomogenization.hh) for a homogeneous base material and prints, in the reference's order
(PeriodicHomogenization_cli.cc:120-172): the homogenized elasticity tensor in the DISPLACEMENT form
(PeriodicHomogenization.hh:146-186), its extreme eigenstrains, the compliance tensor, approximate Young / shear
moduli and Poisson ratios, and the anisotropy. -M dumps the per-element macro-stress-to-micro-stress tensors
(and gtensors.txt) (:174-188), -o the fields `load_ij k`, `w_ij k`, `strain w_ij k` (:190-228, piecewise-linear
subsample; -D full-degree nodal fields and per-element strain interpolants as $ElementNodeData)."""
import argparse
import sys

import numpy as np

from . import _lib as L
from . import homogenization as H
from .core import flat_len
from .linear_elasticity import Simulator
from .mesh_io import MSHFieldWriter, load_mesh, upsample_interpolant
from .simulate_cli import parse_material
from .tensors import ElasticityTensor, closest_isotropic_tensor


def _fmt_matrix(M):
    return "\n".join(" ".join("%.16g" % x for x in row) for row in np.asarray(M))


def _macro_to_micro(sim, w, fl):
    """macroStrainToMicroStrainTensors (PeriodicHomogenization.hh:195-209): G_e, column kl = avg strain(w_kl) + e_kl,
    as flattened [nElem, fl, fl] WITHOUT major symmetry."""
    N = sim.N
    G = np.stack([sim.averageStrainField(w[k]) + H.canonical_strain_flat(N, k)[None, :] for k in range(fl)], axis=2)
    return G


def _unflattened(N, Dm):
    """ElasticityTensor::writeUnflattened (ElasticityTensor.hh:613-633) for a flattened tensor without major symmetry."""
    from .tensors import _flat
    r = range(N)
    return "{" + ", ".join("{" + ", ".join("{" + ", ".join("{" + ", ".join(repr(float(Dm[_flat(N, i, j), _flat(N, k, l)])) for l in r) + "}"
                                                            for k in r) + "}" for j in r) + "}" for i in r) + "}"


def main(argv=None, out=sys.stdout):
    ap = argparse.ArgumentParser(prog="periodic_homogenization_cli")
    ap.add_argument("mesh")
    ap.add_argument("-m", "--material", default="")
    ap.add_argument("-d", "--degree", type=int, default=2)
    ap.add_argument("-M", "--m2mstress", default="")
    ap.add_argument("-o", "--fieldOutput", default="")
    ap.add_argument("-c", "--centerFluctuationDisplacements", action="store_true")
    ap.add_argument("-D", "--fullDegreeFieldOutput", action="store_true")
    ap.add_argument("--distanceToIsotropy", action="store_true")
    ap.add_argument("--distanceToMaterial", default="")
    ap.add_argument("--ignorePeriodicMismatch", action="store_true")
    ap.add_argument("--manualPeriodicVertices", default="")
    ap.add_argument("-O", "--orthotropicCell", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--preconditioner", default="auto", choices=["auto", "multigrid", "two_level", "block_jacobi", "jacobi"],
                    help="auto (default): the multigrid V-cycle, or two_level on a mesh stretched past the measured crossover (MFH_PRECOND_AUTO); where a preconditioner does not apply the library falls back and says so")
    ap.add_argument("--ascii", action="store_true", help="write an ASCII .msh (default binary like the reference)")
    a = ap.parse_args(argv)
    if a.degree not in (1, 2):
        ap.error("FEM Degree must be 1 or 2")
    V, E, _ = load_mesh(a.mesh)
    K = E.shape[1] - 1
    if K not in (2, 3):
        raise RuntimeError("Mesh must be triangle or tet.")
    N = K
    V = np.ascontiguousarray(V[:, :N])
    fl = flat_len(N)
    mat = parse_material(a.material, N) if a.material else ElasticityTensor(N)      # Materials.hh default E = 1, nu = 0.3
    sim = Simulator(E, V, degree=a.degree, device=a.device)
    sim.rtol = a.rtol
    sim.ctx.set_preconditioner({"auto": L.PRECOND_AUTO, "multigrid": L.PRECOND_MULTIGRID, "two_level": L.PRECOND_TWO_LEVEL, "block_jacobi": L.PRECOND_BLOCK_JACOBI, "jacobi": L.PRECOND_JACOBI}[a.preconditioner])
    sim.setMaterial(mat)
    if a.orthotropicCell:
        Ch, w, infos = H.homogenize_orthotropic_cell(sim)
    else:
        w, infos = H.solve_cell_problems(sim, ignore_periodic_mismatch=a.ignorePeriodicMismatch,
                                         manual_periodic_vertices_file=a.manualPeriodicVertices)
        Ch = H.homogenized_elasticity_tensor_displacement_form(sim, w)
    Eh = ElasticityTensor(N)
    Eh.D = 0.5 * (Ch + Ch.T)                         # the reference stores the upper triangle of a major-symmetric tensor
    p = lambda *x: print(*x, file=out)
    p("Homogenized elasticity tensor:")
    p(_fmt_matrix(Eh.D)); p()
    lam, Q = Eh.computeEigenstrains()
    for name, k in (("Minimum", 0), ("Intermediate", 1), ("Max", 2)):        # :125-133 prints eigenpairs 0, 1, 2
        p("%s Eh eigenvalue %.16g for eigenstrain: %s" % (name, lam[k], " ".join("%.16g" % x for x in Q[:, k])))
    S = Eh.inverse()
    p("Homogenized compliance tensor:")
    p(_fmt_matrix(S.D))
    moduli = [(1.0 if i < N else 0.25) / S.D[i, i] for i in range(fl)]        # shear moduli are multiplied by 4 (:139-141)
    if N == 2:
        poisson = [-S.D[0, 1] / S.D[1, 1], -S.D[1, 0] / S.D[0, 0]]
        p("Approximate Young moduli:\t%.16g\t%.16g" % tuple(moduli[:2]))
        p("Approximate shear modulus:\t%.16g" % moduli[2])
        p("v_yx, v_xy:\t%.16g\t%.16g" % tuple(poisson))
    else:
        poisson = [-S.D[0, 1] / S.D[1, 1], -S.D[0, 2] / S.D[2, 2], -S.D[1, 2] / S.D[2, 2],
                   -S.D[1, 0] / S.D[0, 0], -S.D[2, 0] / S.D[0, 0], -S.D[2, 1] / S.D[1, 1]]
        p("Approximate Young moduli:\t%.16g\t%.16g\t%.16g" % tuple(moduli[:3]))
        p("Approximate shear moduli:\t%.16g\t%.16g\t%.16g" % tuple(moduli[3:]))
        p("v_yx, v_zx, v_zy:\t%.16g\t%.16g\t%.16g" % tuple(poisson[:3]))
        p("v_xy, v_xz, v_yz:\t%.16g\t%.16g\t%.16g" % tuple(poisson[3:]))
    p("Anisotropy:\t%.16g" % Eh.anisotropy())
    p("PCG iterations per cell problem: %s" % " ".join(str(i["iterations"]) for i in infos))

    if a.m2mstress:                                   # F_e = E_base : (G_e : S)   (:174-188)
        G = _macro_to_micro(sim, w, fl)
        dbl = np.ones(fl); dbl[N:] = 2.0
        with open(a.m2mstress, "w") as mf, open("gtensors.txt", "w") as gf:
            for e in range(G.shape[0]):
                gf.write(_unflattened(N, G[e]) + "\n")
                F = mat.D @ (dbl[:, None] * (G[e] @ (dbl[:, None] * S.D)))        # doubleContract: F(A:B) = F(A) S F(B)
                mf.write(_unflattened(N, F) + "\n")

    if a.fieldOutput:                                 # :190-228
        if a.centerFluctuationDisplacements:
            w = [x - x.mean(axis=0) for x in w]
        nodes, elems = sim.nodes(), sim.elements()
        dm, _ = sim.ctx.get_dof_map()
        nv = len(V)
        if a.fullDegreeFieldOutput:
            wr = MSHFieldWriter(a.fieldOutput, nodes, elems, binary=not a.ascii)
            cut = slice(None)
        else:
            wr = MSHFieldWriter(a.fieldOutput, nodes[:nv], elems[:, :K + 1], binary=not a.ascii)
            cut = slice(0, nv)
        for k in range(fl):
            load = sim.constantStrainLoad(-H.canonical_strain_flat(N, k))[dm]          # dofToNodeField
            wr.addField("load_ij %d" % k, load[cut], "node")
            wr.addField("w_ij %d" % k, w[k][cut], "node")
            if a.fullDegreeFieldOutput and a.degree == 2:  # :214-226
                wr.addElementNodeField("strain w_ij %d" % k, upsample_interpolant(sim.strainField(w[k]), N))
            else:
                wr.addField("strain w_ij %d" % k, sim.averageStrainField(w[k]), "element")
        wr.close()

    if a.distanceToIsotropy:                          # :230-236
        iso = closest_isotropic_tensor(Eh)
        p()
        p("(Sq Rel Frob) Distance to Isotropy:\t%.16g" % ((iso - Eh).frobeniusNormSq() / iso.frobeniusNormSq()))
        p("Closest isotropic tensor:")
        p(_fmt_matrix(iso.D)); p()
    if a.distanceToMaterial:                          # :238-242
        tgt = parse_material(a.distanceToMaterial, N)
        p("(Sq Rel Frob) Distance to Specified Tensor:\t%.16g" % ((Eh - tgt).frobeniusNormSq() / tgt.frobeniusNormSq()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
