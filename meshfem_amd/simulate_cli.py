"""Simulate_cli-compatible driver on the GPU path (mirror of src/bin/Simulate_cli.cc):

    python -m meshfem_amd.simulate_cli mesh.msh -m material.json -b conditions.bc -d 2 -o out.msh
        [--dumpMatrix K.bin] [-D] [--device 0] [--rtol 1e-8] [--preconditioner two_level|block_jacobi]

Reads a Gmsh 2.2 tri/tet mesh, a `.material` JSON (isotropic / orthotropic / symmetric_material,
Materials.cc:178-311) or per-element material fields stored in an .msh (`E`,`nu` or the 9 (3D) /
4 (2D) orthotropic fields, Simulate_cli.cc:116-163), and a `.bc` JSON with box / box% regions of
type dirichlet[xyz] / force / traction / pressure / delta force (BoundaryConditions.cc:227-389).
Writes the fields u, load, strain, stress, Ku like Simulate_cli.cc:207-242."""
import argparse
import json
import sys

import numpy as np

from . import _lib as L
from .linear_elasticity import Simulator
from .mesh_io import MSHFieldWriter, load_msh
from .tensors import ElasticityTensor


def parse_material(path, dim):
    """Materials::Constant JSON (Materials.cc:301-311) -> ElasticityTensor."""
    if isinstance(path, dict):
        cfg = path
    else:
        with open(path) as f:
            cfg = json.load(f)
    t = cfg["type"]
    ten = ElasticityTensor(dim)
    if t in ("isotropic_material", "isotropic"):
        return ten.setIsotropic(float(cfg["young"]), float(cfg["poisson"]))
    if t in ("orthotropic_material", "orthotropic"):
        y, p, s = cfg["young"], cfg["poisson"], cfg["shear"]
        if dim == 2:
            if abs(p[1] / y[1] - p[0] / y[0]) > 1e-10:
                raise RuntimeError("Orthotopic parameters violate symmetry")
            return ten.setOrthotropic(y[0], y[1], p[1], s[0])
        if (abs(p[5] / y[1] - p[4] / y[0]) > 1e-10 or abs(p[0] / y[1] - p[1] / y[2]) > 1e-10 or abs(p[2] / y[2] - p[3] / y[0]) > 1e-10):
            raise RuntimeError("Orthotopic parameters violate symmetry")
        return ten.setOrthotropic(y[0], y[1], y[2], p[5], p[2], p[1], s[0], s[1], s[2])
    if t in ("symmetric_material", "anisotropic"):
        D = np.array(cfg["material_matrix"], dtype=np.float64)
        if np.abs(D - D.T).max() > 1e-10 * max(1.0, np.abs(D).max()):
            raise RuntimeError("Asymmetric material_matrix")
        ten.D = D
        return ten
    raise RuntimeError("Invalid type.")


def material_json(tensor):
    """Materials::Constant::getJson (Materials.cc:314-328): always the anisotropic form."""
    return {"type": "anisotropic", "material_matrix": np.asarray(tensor.D).tolist()}


def _vec(v, dim):
    out = [float(x) for x in v]          # expression strings are not supported (numbers only)
    return np.array((out + [0.0] * 3)[:dim])


def apply_boundary_conditions(sim, path):
    """readBoundaryConditions + applyBoundaryConditions for box / box% regions."""
    with open(path) as f:
        cfg = json.load(f)
    if cfg.get("no_rigid_motion", False):                    # BoundaryConditions.cc:236-239 -> applyNoRigidMotionConstraint
        sim.applyNoRigidMotionConstraint()
    N = sim.N
    for r in cfg["regions"]:
        t = r["type"]
        comps = None
        if t.startswith("dirichlet"):
            rest = t[9:]
            k = 0
            while k < len(rest) and rest[k] in "xyz":
                k += 1
            if k:
                comps = ["xyz"[c] in rest[:k] for c in range(N)]
            t = "dirichlet" + rest[k:]
        if "box" in r:
            mn, mx, rel = _vec(r["box"]["minCorner"], N), _vec(r["box"]["maxCorner"], N), False
        elif "box%" in r:
            mn, mx, rel = _vec(r["box%"]["minCorner"], N), _vec(r["box%"]["maxCorner"], N), True
        else:
            raise RuntimeError("only box / box% regions are supported")
        val = r["value"]
        if t == "dirichlet":
            sim.applyDirichletBox(mn, mx, _vec(val, N), relative=rel, components=comps)
        elif t == "force":
            sim.applyNeumannBox(mn, mx, _vec(val, N), kind=L.NEUMANN_FORCE, relative=rel)
        elif t == "traction":
            sim.applyNeumannBox(mn, mx, _vec(val, N), kind=L.NEUMANN_TRACTION, relative=rel)
        elif t == "pressure":
            sim.applyNeumannBox(mn, mx, [float(val if np.isscalar(val) else val[0])], kind=L.NEUMANN_PRESSURE, relative=rel)
        else:
            raise RuntimeError("Invalid type '%s'" % r["type"])


def main(argv=None):
    ap = argparse.ArgumentParser(prog="simulate_cli")
    ap.add_argument("mesh")
    ap.add_argument("-m", "--material", default="")
    ap.add_argument("-b", "--boundaryConditions")
    ap.add_argument("-o", "--outputMSH")
    ap.add_argument("--dumpMatrix", default="")
    ap.add_argument("-d", "--degree", type=int, default=2)
    ap.add_argument("-D", "--fullDegreeFieldOutput", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--preconditioner", default="two_level", choices=["two_level", "block_jacobi", "jacobi"])
    ap.add_argument("--ascii", action="store_true", help="write an ASCII .msh (default binary like the reference)")
    a = ap.parse_args(argv)
    if not a.dumpMatrix and not a.outputMSH:
        ap.error("must specify output msh file (unless dumping a stiffness matrix)")
    if a.outputMSH and not a.boundaryConditions:
        ap.error("must specify boundary conditions to run a simulation")
    V, E, _ = load_msh(a.mesh)
    K = E.shape[1] - 1
    if K not in (2, 3):
        raise RuntimeError("only triangle and tetrahedron meshes are supported")
    N = K
    if N == 2:
        if np.abs(V[:, 2]).max() > 0:
            raise RuntimeError("2D simulation needs a planar (z = 0) triangle mesh")
        V = V[:, :2]
    sim = Simulator(E, np.ascontiguousarray(V), degree=a.degree, device=a.device)
    sim.rtol = a.rtol
    sim.ctx.set_preconditioner({"two_level": L.PRECOND_TWO_LEVEL, "block_jacobi": L.PRECOND_BLOCK_JACOBI, "jacobi": L.PRECOND_JACOBI}[a.preconditioner])
    if a.material.endswith(".msh"):
        _, _, fields = load_msh(a.material)
        if "E" in fields and "nu" in fields:
            sim.setIsotropicField(fields["E"][1][:, 0], fields["nu"][1][:, 0])
        else:
            names = ["E_x", "E_y", "E_z", "nu_yx", "nu_zx", "nu_zy", "mu_yz", "mu_zx", "mu_xy"] if N == 3 else ["E_x", "E_y", "nu_yx", "mu"]
            sim.setOrthotropicField(np.column_stack([fields[k][1][:, 0] for k in names]))
    elif a.material:
        sim.setMaterial(parse_material(a.material, N))
    if a.dumpMatrix and not a.boundaryConditions:          # Simulate_cli.cc:178-184
        i, j, v = sim.assembleStiffnessMatrix()
        with open(a.dumpMatrix, "wb") as f:
            np.array([len(v)], dtype=np.uint64).tofile(f); i.tofile(f); j.tofile(f); v.tofile(f)
        return 0
    apply_boundary_conditions(sim, a.boundaryConditions)
    u = sim.solve()
    e, s = sim.averageStrainField(u), sim.averageStressField(u)
    dm, _ = sim.ctx.get_dof_map()
    f = sim.neumannLoad()[dm]
    Ku = sim.applyStiffnessMatrix(u)[dm] if sim.numDoFs() == sim.numNodes() else None
    print("PCG: %d iterations, relative residual %.3e, %.1f ms" % (sim.info["iterations"], sim.info["true_rel_residual"], sim.info["solve_ms"]))
    nodes, elems = sim.nodes(), sim.elements()
    if a.fullDegreeFieldOutput:
        w = MSHFieldWriter(a.outputMSH, nodes, elems, binary=not a.ascii)
        w.addField("u", u, "node"); w.addField("load", f, "node")
    else:                                                   # piecewise-linear subsample (MSHFieldWriter.hh:74-83)
        nv = len(V)
        w = MSHFieldWriter(a.outputMSH, nodes[:nv], elems[:, :K + 1], binary=not a.ascii)
        w.addField("u", u[:nv], "node"); w.addField("load", f[:nv], "node")
    w.addField("strain", e, "element"); w.addField("stress", s, "element")
    if Ku is not None:
        w.addField("Ku", Ku if a.fullDegreeFieldOutput else Ku[:len(V)], "node")
    w.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
