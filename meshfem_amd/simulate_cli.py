"""Simulate_cli-compatible driver on the GPU path (mirror of src/bin/Simulate_cli.cc):

    python -m meshfem_amd.simulate_cli mesh.msh -m material.json -b conditions.bc -d 2 -o out.msh
        [-f fieldPrefix] [-e extra.msh] [--dumpMatrix K.bin] [-D] [--device 0] [--rtol 1e-8]
        [--preconditioner auto|multigrid|two_level|block_jacobi]

Reads a Gmsh 2.2 tri/tet mesh, a `.material` JSON (isotropic / orthotropic / symmetric_material,
Materials.cc:178-311) or per-element material fields stored in an .msh (`E`,`nu` or the 9 (3D) /
4 (2D) orthotropic fields, Simulate_cli.cc:116-163), and a `.bc` JSON with box / box% regions of
type dirichlet[xyz] / force / traction / pressure / delta force (BoundaryConditions.cc:227-389), numeric or
expression-valued (`"sin(pi * x)"`, tinyexpr grammar: meshfem_amd/expressions.py), `dirichlet nodes` /
`delta force nodes` lists, `traction | pressure | force elements` lists, `dirichlet elements`, and path / polygon
regions (Geometry.hh:68-191); top-level `no_rigid_motion`, `pin_translation` and `fix_periodic_pair_<c>`.
target conditions are skipped with the reference's warning, contact / fracture conditions are "Illegal BC type"
(LinearElasticity.hh:933-938,:1024).
Writes the fields u, load, strain, stress, Ku like Simulate_cli.cc:207-242."""
import argparse
import json
import sys

import numpy as np

from . import _lib as L
from .linear_elasticity import Simulator
from .mesh_io import MSHFieldWriter, load_mesh, load_msh, upsample_interpolant
from .tensors import ElasticityTensor
from .expressions import ExpressionVector, environment


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
    out = [float(x) for x in v]
    return np.array((out + [0.0] * 3)[:dim])


def _is_expression_vector(v):
    return not np.isscalar(v) and any(isinstance(x, str) for x in v)


def _expression_vector(v, dim):
    """parseExpressionVector + the 2D truncation of BoundaryConditions.cc:343-350."""
    comps = list(v)
    if dim == 2 and len(comps) == 3 and float(comps[2]) == 0:
        comps.pop()
    if len(comps) != dim:
        raise RuntimeError("Incorrect expression vector size")
    return ExpressionVector(comps)


class _PathRegion:
    """PathRegion (Geometry.hh:68-124): points closer than 1e-5 to the polyline. minCorner / maxCorner stay zero like the
    reference's default Region (they only feed the region_* expression variables)."""

    def __init__(self, pts, dim):
        self.pts = np.array([_vec(p, dim) for p in pts])
        self.mn = self.mx = np.zeros(dim)

    def contains(self, P):
        P = np.asarray(P, dtype=np.float64)
        inside = np.zeros(len(P), dtype=bool)
        for e1, e2 in zip(self.pts[:-1], self.pts[1:]):
            v = e2 - e1
            t = np.clip(((P - e1) @ v) / (v @ v), 0.0, 1.0)
            inside |= np.linalg.norm(P - (e1 + t[:, None] * v), axis=1) < 1e-5
        return inside


class _PolygonalRegion:
    """PolygonalRegion (Geometry.hh:126-191): odd number of crossings of the segment from (min x - 1, 1.90588) to the point
    with the polygon's edges, with the reference's intersection predicate (eps 1e-10). 2D only, like the reference."""

    def __init__(self, pts, dim):
        self.pts = np.array([_vec(p, dim)[:2] for p in pts])
        self.outside = np.array([self.pts[:, 0].min() - 1.0, 1.90588])
        self.mn = self.mx = np.zeros(dim)

    def contains(self, P):
        P = np.asarray(P, dtype=np.float64)[:, :2]
        det = lambda u, v: u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]
        count = np.zeros(len(P), dtype=np.int64)
        c = self.outside
        for k in range(len(self.pts)):
            a, b = self.pts[k], self.pts[(k + 1) % len(self.pts)]
            d = P
            x, y, z = det(c - a, d - c), det(b - a, a - c), det(b - a, d - c)
            miss = (np.abs(z) < 1e-10) | (x * z < 0) | (x * z > z * z) | (y * z < 0) | (y * z > z * z)
            count += ~miss
        return count % 2 == 1


class _BoxRegion:
    def __init__(self, mn, mx):
        self.mn, self.mx = np.asarray(mn, dtype=np.float64), np.asarray(mx, dtype=np.float64)

    def contains(self, P):                                   # BBox::containsPoint, inclusive (Geometry.hh:276-279)
        return np.all((np.asarray(P) >= self.mn) & (np.asarray(P) <= self.mx), axis=1)


def _node_values(values, dim):
    """parseNodeConditionValues (BoundaryConditions.cc:62-80): [[value, [node, ...]], ...]."""
    idx, val = [], []
    for value, nodes in values:
        for nd in nodes:
            idx.append(int(nd)); val.append(_vec(value, dim))
    return np.array(idx, dtype=np.int64), np.array(val).reshape(len(idx), dim)


def _boundary_element_lookup(sim):
    """corner-index set -> boundary element (UnorderedTriplet matching of NeumannElementsCondition)."""
    K = sim.N
    ben = sim.ctx.boundary_elem_nodes()[:, :K]
    return {tuple(sorted(int(x) for x in row)): b for b, row in enumerate(ben)}


def apply_boundary_conditions(sim, path):
    """readBoundaryConditions + applyBoundaryConditions (BoundaryConditions.cc:227-389, LinearElasticity.hh:881-1027):
    box / box% regions of type dirichlet[xyz] / force / traction / pressure / delta force with numeric or EXPRESSION
    values (traction, dirichlet, delta force), node lists (dirichlet nodes, delta force nodes) and boundary-element lists
    (traction / pressure / force elements)."""
    if isinstance(path, dict):
        cfg = path
    else:
        with open(path) as f:
            cfg = json.load(f)
    if cfg.get("no_rigid_motion", False):                    # BoundaryConditions.cc:236-239 -> applyNoRigidMotionConstraint
        sim.applyNoRigidMotionConstraint()
    N = sim.N
    be_lookup = None
    geo = {}
    if cfg.get("pin_translation") or any(("fix_periodic_pair_" + a) in cfg for a in "xyz"[:N]):
        # Simulate_cli.cc:191-193: applyTranslationPins before, applyPeriodicPairDirichletConditions after the regions
        pos_ = sim.nodes()
        bn_ = sim.ctx.boundary_nodes()
        for d in range(N):                                   # LinearElasticity.hh:1095-1111: pin component d of the
            if "xyz"[d] in cfg.get("pin_translation", ""):   # boundary node with the smallest coordinate d (first one found)
                node = bn_[int(np.argmin(pos_[bn_, d]))]
                sim.applyDirichletNodes([int(node)], np.zeros((1, N)), [a == d for a in range(N)])
    region_of_node = {}                                      # BoundaryNode::dirichletRegionIdx (setDirichletRegion, :946,:962)
    n_dirichlet_regions = 0

    def mesh_geometry():                                     # only list / expression / delta-force regions need it
        if not geo:
            p = sim.nodes()
            nv = sim.ctx.n_vert
            geo.update(pos=p, mn=p[:nv].min(axis=0), mx=p[:nv].max(axis=0))
        return geo["pos"], geo["mn"], geo["mx"]
    for r in cfg["regions"]:
        t = r["type"]
        comps = None
        if t.startswith("dirichlet"):
            rest = t[9:]
            k = 0
            while k < len(rest) and rest[k] in "xyz":
                k += 1
            if k > 3:
                raise RuntimeError("invalid mask")
            if k:
                comps = ["xyz"[a] in rest[:k] for a in range(N)]
            t = "dirichlet" + rest[k:]
        if t.startswith("target"):                            # TargetCondition / TargetNodesCondition are not Dirichlet
            print("WARNING: ignoring target boundary conditions.", file=sys.stderr)     # conditions (:933-938)
            continue
        if t in ("contact", "fracture"):
            raise RuntimeError("Illegal BC type")              # :1024
        if t == "dirichlet" and hasattr(sim, "ctx"):        # region index of the boundary nodes inside (:939-949)
            n_dirichlet_regions += 1
            if ("box" in r or "box%" in r) and not _is_expression_vector(r["value"]):
                pos_, mn_, mx_ = mesh_geometry()
                key = "box" if "box" in r else "box%"
                lo_, hi_ = _vec(r[key]["minCorner"], N), _vec(r[key]["maxCorner"], N)
                if key == "box%":
                    lo_, hi_ = mn_ + lo_ * (mx_ - mn_), mn_ + hi_ * (mx_ - mn_)
                bn_ = sim.ctx.boundary_nodes()
                for nd in bn_[np.all((pos_[bn_] >= lo_) & (pos_[bn_] <= hi_), axis=1)]:
                    region_of_node[int(nd)] = n_dirichlet_regions
        if t == "dirichlet nodes":
            idx, val = _node_values(r["values"], N)
            sim.applyDirichletNodes(idx, val, comps)
            continue
        if t == "delta force nodes":
            idx, val = _node_values(r["values"], N)
            for ni, f in zip(idx, val):
                if ni > sim.numNodes():
                    raise RuntimeError("DeltaForceNodesCondition node index out of bounds: %d" % ni)
                sim.ctx.bc_delta_force(int(ni), f)
            continue
        if t in ("traction elements", "pressure elements", "force elements"):
            if be_lookup is None:
                be_lookup = _boundary_element_lookup(sim)
            area, nrm = sim.ctx.boundary_elem_geometry()
            bes, vals = [], []
            for value, elems in r["values"]:
                v = _vec(value, N)
                for corners in elems:
                    key = tuple(sorted(int(x) for x in corners))[:N] if len(corners) == N else None
                    if key is None or key not in be_lookup:
                        raise RuntimeError("Some element boundary conditions weren't matched.")
                    bes.append(be_lookup[key]); vals.append(v)
            bes, vals = np.array(bes, dtype=np.int64), np.array(vals).reshape(len(bes), N)
            if t == "pressure elements":
                vals = -vals[:, :1] * nrm[bes]
            elif t == "force elements":                        # total force / area of the listed elements (:984-989)
                vals = vals / area[bes].sum()
            sim.applyNeumannElements(bes, vals)
            continue
        region = None
        if "box" in r:
            mn, mx, rel = _vec(r["box"]["minCorner"], N), _vec(r["box"]["maxCorner"], N), False
        elif "box%" in r:
            mn, mx, rel = _vec(r["box%"]["minCorner"], N), _vec(r["box%"]["maxCorner"], N), True
        elif "path" in r:
            region = _PathRegion(r["path"], N)
        elif "polygon" in r:
            region = _PolygonalRegion(r["polygon"], N)
        elif t == "dirichlet elements" and "element vertices" in r:
            region = "elements"
        else:
            raise RuntimeError("regions are box, box%, path, polygon, node lists or element lists")
        val = r["value"]
        needs_host = _is_expression_vector(val) or t == "delta force" or region is not None
        if needs_host:
            c = sim.ctx
            pos, mesh_min, mesh_max = mesh_geometry()
            if region is None:
                amn, amx = (mesh_min + mn * (mesh_max - mesh_min), mesh_min + mx * (mesh_max - mesh_min)) if rel else (mn, mx)
                region = _BoxRegion(amn, amx)
            ev = _expression_vector(val, N) if _is_expression_vector(val) else None

            def values_at(P, k):                              # numeric value or the expression at the k selected points
                if ev is None:
                    return np.tile(_vec(val, N) if not np.isscalar(val) else [float(val)] + [0.0] * (N - 1), (k, 1))
                return ev.eval(environment(N, mesh_min, mesh_max, region.mn, region.mx, P), k)
            if t == "dirichlet elements":                   # DirichletElementsCondition (:951-965): all nodes of the listed boundary elements
                if be_lookup is None:
                    be_lookup = _boundary_element_lookup(sim)
                ben_all = c.boundary_elem_nodes()
                n_dirichlet_regions += 1
                sel = []
                for corners in r["element vertices"]:
                    key = tuple(sorted(int(x) for x in corners))
                    if len(key) != N:
                        raise RuntimeError("Error parsing element vertices.")
                    if key in be_lookup:
                        sel.extend(int(x) for x in ben_all[be_lookup[key]])
                sel = np.array(sorted(set(sel)), dtype=np.int64)
                region = _BoxRegion(np.zeros(N), np.zeros(N))
                sim.applyDirichletNodes(sel, values_at(pos[sel], len(sel)), comps)
                for nd in sel:
                    region_of_node[int(nd)] = n_dirichlet_regions
            elif t == "dirichlet":                          # :939-949: boundary NODES inside the region
                bn = c.boundary_nodes()
                sel = bn[region.contains(pos[bn])]
                sim.applyDirichletNodes(sel, values_at(pos[sel], len(sel)), comps)
                for nd in sel:
                    region_of_node[int(nd)] = n_dirichlet_regions
            elif t in ("traction", "force", "pressure"):    # :897-933: boundary ELEMENTS by vertex barycentre
                if ev is not None and t != "traction":
                    raise RuntimeError("Only region-based traction, dirichlet, target, and delta force support expression vectors")
                ben = c.boundary_elem_nodes()[:, :N]
                ctr = pos[ben].mean(axis=1)
                sel = np.flatnonzero(region.contains(ctr))
                if len(sel) == 0:
                    raise RuntimeError("Neumann region unmatched")
                area, nrm = c.boundary_elem_geometry()
                v = values_at(ctr[sel], len(sel))
                if t == "pressure":
                    v = -v[:, :1] * nrm[sel]
                elif t == "force":
                    v = v / area[sel].sum()
                sim.applyNeumannElements(sel, v)
            elif t == "delta force":                        # :1009-1015: every NODE inside the region
                sel = np.flatnonzero(region.contains(pos))
                for ni, fv in zip(sel, values_at(pos[sel], len(sel))):
                    c.bc_delta_force(int(ni), fv)
            else:
                raise RuntimeError("Invalid type '%s'" % r["type"])
            continue
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
    for c_ in range(N):                                      # "fix_periodic_pair_<component>": "<orthogonal axis>"
        key = "fix_periodic_pair_" + "xyz"[c_]               # (BoundaryConditions.cc:229-245, BoundaryConditions.hh:54-100,
        if key not in cfg:                                   #  LinearElasticity.hh:1087-1093): component c of ONE matching node
            continue                                         #  pair on the min / max faces of the axis is fixed to zero
        face = "xyz"[:N].find(cfg[key]) if cfg[key] in ("x", "y", "z")[:N] and cfg[key] != "xyz"[c_] else -1
        if face < 0:
            raise RuntimeError("invalid " + key)
        lo, hi = pos_[:sim.ctx.n_vert].min(axis=0), pos_[:sim.ctx.n_vert].max(axis=0)
        first = next((int(n) for n in bn_ if abs(pos_[n, face] - lo[face]) <= 1e-5), None)
        if first is None:
            raise RuntimeError("No vertices on the periodic pair face.")
        target = pos_[first].copy(); target[face] = hi[face]
        second = next((int(n) for n in bn_ if np.linalg.norm(pos_[n] - target) <= 1e-5), None)
        if second is None:
            raise RuntimeError("Couldn't match vertex in periodic pair Dirichlet condition")
        sim.applyDirichletNodes([first, second], np.zeros((2, N)), [a == c_ for a in range(N)])
    return region_of_node


def region_surface_forces(sim, u, region_of_node):
    """reportRegionSurfaceForces (LinearElasticity.hh:1251-1270): K u summed over the boundary nodes of every Dirichlet
    region (index 0 collects the boundary nodes outside all of them)."""
    f = sim.applyStiffnessMatrix(u)
    bn = sim.ctx.boundary_nodes()
    idx = np.array([region_of_node.get(int(n), 0) for n in bn], dtype=np.int64)
    forces = np.zeros((idx.max() + 1 if len(idx) else 1, sim.N))
    np.add.at(forces, idx, f[bn])
    return forces


def main(argv=None):
    ap = argparse.ArgumentParser(prog="simulate_cli")
    ap.add_argument("mesh")
    ap.add_argument("-m", "--material", default="")
    ap.add_argument("-f", "--matFieldName", default="", help="name prefix of the material fields in the .msh passed as --material")
    ap.add_argument("-e", "--extraMesh", default="", help="adds another independent input mesh to the problem")
    ap.add_argument("-b", "--boundaryConditions")
    ap.add_argument("-o", "--outputMSH")
    ap.add_argument("--dumpMatrix", default="")
    ap.add_argument("-d", "--degree", type=int, default=2)
    ap.add_argument("-D", "--fullDegreeFieldOutput", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--preconditioner", default="auto", choices=["auto", "multigrid", "two_level", "block_jacobi", "jacobi"],
                    help="auto (default): the multigrid V-cycle, or two_level on a mesh stretched past the measured crossover (MFH_PRECOND_AUTO); where a preconditioner does not apply the library falls back and says so")
    ap.add_argument("--ascii", action="store_true", help="write an ASCII .msh (default binary like the reference)")
    a = ap.parse_args(argv)
    if not a.dumpMatrix and not a.outputMSH:
        ap.error("must specify output msh file (unless dumping a stiffness matrix)")
    if a.outputMSH and not a.boundaryConditions:
        ap.error("must specify boundary conditions to run a simulation")
    V, E, _ = load_mesh(a.mesh)                            # .msh / .off / .obj / .mesh (MeshIO::load)
    if a.extraMesh:                                         # a second, independent mesh in the same problem (Simulate_cli.cc:270-310)
        V2, E2, _ = load_mesh(a.extraMesh)
        if E2.shape[1] != E.shape[1]:
            raise RuntimeError("Extra mesh of different type.")
        E = np.vstack([E, E2 + len(V)])
        V = np.vstack([V, V2])
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
    sim.ctx.set_preconditioner({"auto": L.PRECOND_AUTO, "multigrid": L.PRECOND_MULTIGRID, "two_level": L.PRECOND_TWO_LEVEL, "block_jacobi": L.PRECOND_BLOCK_JACOBI, "jacobi": L.PRECOND_JACOBI}[a.preconditioner])
    if a.material.endswith(".msh"):                        # heterogeneous material fields (Simulate_cli.cc:104-163)
        _, _, fields = load_msh(a.material)
        pre = a.matFieldName

        def elem_field(name):
            kind, vals = fields[pre + name]
            if kind != "element" or len(vals) != sim.numElements():
                raise RuntimeError("Material parameter fields of incorrect size.")
            return vals[:, 0]
        if pre + "E" in fields and pre + "nu" in fields:
            sim.setIsotropicField(elem_field("E"), elem_field("nu"))
            print("Loaded %dD isotropic material" % N)
        else:
            names = ["E_x", "E_y", "E_z", "nu_yx", "nu_zx", "nu_zy", "mu_yz", "mu_zx", "mu_xy"] if N == 3 else ["E_x", "E_y", "nu_yx", "mu"]
            if any(pre + k not in fields for k in names):
                raise RuntimeError("No complete material parameter field was found.")
            sim.setOrthotropicField(np.column_stack([elem_field(k) for k in names]))
            print("Loaded %dD Orthotropic material" % N)
    elif a.material:
        sim.setMaterial(parse_material(a.material, N))
    if a.dumpMatrix and not a.boundaryConditions:          # Simulate_cli.cc:178-184
        i, j, v = sim.assembleStiffnessMatrix()
        with open(a.dumpMatrix, "wb") as f:
            np.array([len(v)], dtype=np.uint64).tofile(f); i.tofile(f); j.tofile(f); v.tofile(f)
        return 0
    regions = apply_boundary_conditions(sim, a.boundaryConditions)
    if a.dumpMatrix:                                        # Simulate_cli.cc:195 sim.dumpSystem -> SPSDSystem::sumAndDumpUpper:
        i, j, v = sim.assembleStiffnessMatrix()             # the REDUCED system, fixed variables removed and renumbered
        fv, _ = sim.ctx.bc_dirichlet_vars()
        keep = np.ones(N * sim.numDoFs(), dtype=bool)
        keep[fv] = False
        new = (np.cumsum(keep) - 1).astype(np.uint64)
        sel = keep[i.astype(np.int64)] & keep[j.astype(np.int64)]
        with open(a.dumpMatrix, "wb") as f:
            np.array([int(sel.sum())], dtype=np.uint64).tofile(f)
            new[i[sel].astype(np.int64)].tofile(f); new[j[sel].astype(np.int64)].tofile(f); v[sel].tofile(f)
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
    if a.fullDegreeFieldOutput and a.degree == 2:          # full-degree per-element strain / stress (Simulate_cli.cc:216-229)
        w.addElementNodeField("strain", upsample_interpolant(sim.strainField(u), N))
        w.addElementNodeField("stress", upsample_interpolant(sim.stressField(u), N))
    else:
        w.addField("strain", e, "element"); w.addField("stress", s, "element")
    if Ku is not None:
        for ri, fr in enumerate(region_surface_forces(sim, u, regions)):     # Simulate_cli.cc:239
            print("region %d force:" % ri + "".join("\t%g" % x for x in fr))
        w.addField("Ku", Ku if a.fullDegreeFieldOutput else Ku[:len(V)], "node")
    w.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
