"""Gmsh MSH 2.2 reader/writer compatible with the reference's MeshIO_MSH and MSHFieldWriter
(src/lib/MeshFEM/MeshIO.cc:525-760, MSHFieldWriter.hh): ASCII and binary, consecutively numbered
1-indexed nodes, a single element type (tri = 2, tet = 4, tri6 = 9, tet10 = 11), per-node and
per-element scalar / vector / symmetric-matrix fields (2-vectors padded to 3, matrices to 3x3)."""
import struct

import numpy as np

ELEM_NODES = {2: 3, 4: 4, 9: 6, 11: 10, 3: 4, 5: 8, 1: 2, 8: 3}     # Gmsh elm-type -> nodes (MeshIO.cc:527-531)
TYPE_FOR_NODES = {3: 2, 4: 4, 6: 9, 10: 11}                           # "guess" of MESH_GUESS: 4 nodes = tet


def _readline(f):
    line = f.readline()
    while line and not line.strip():
        line = f.readline()
    return line.decode().strip()


def load_msh(path):
    """Returns (vertices [n,3] f64, elements [m,k] i64, fields dict name -> (domain, array))."""
    with open(path, "rb") as f:
        if _readline(f) != "$MeshFormat":
            raise RuntimeError("Bad MSH file format")
        version, file_type, data_size = _readline(f).split()
        binary = int(file_type) == 1
        if int(file_type) > 1 or int(data_size) != 8:
            raise RuntimeError("Unsupported MSH file format")
        if binary:
            (one,) = struct.unpack("<i", f.read(4))
            if one != 1:
                raise RuntimeError("Unsupported MSH file format")
        if _readline(f) != "$EndMeshFormat" or _readline(f) != "$Nodes":
            raise RuntimeError("Bad MSH file format")
        nn = int(_readline(f))
        if binary:
            rec = np.frombuffer(f.read(nn * 28), dtype=np.dtype([("i", "<i4"), ("p", "<f8", 3)]))
            if not np.array_equal(rec["i"], np.arange(1, nn + 1)):
                raise RuntimeError("Unsupported MSH file format")       # nodes must be consecutive, 1-indexed
            V = rec["p"].astype(np.float64)
        else:
            V = np.empty((nn, 3))
            for i in range(nn):
                t = _readline(f).split()
                if int(t[0]) != i + 1:
                    raise RuntimeError("Unsupported MSH file format")
                V[i] = [float(x) for x in t[1:4]]
        if _readline(f) != "$EndNodes" or _readline(f) != "$Elements":
            raise RuntimeError("Bad MSH file format")
        ne = int(_readline(f))
        elems, etype = [], None
        if binary:
            read = 0
            while read < ne:
                t, cnt, ntags = struct.unpack("<3i", f.read(12))
                etype = t if etype is None else etype
                if t != etype:
                    raise RuntimeError("Bad MSH file format")
                k = ELEM_NODES[t]
                data = np.frombuffer(f.read(cnt * 4 * (1 + ntags + k)), dtype="<i4").reshape(cnt, 1 + ntags + k)
                elems.append(data[:, 1 + ntags:] - 1)
                read += cnt
            E = np.concatenate(elems).astype(np.int64) if elems else np.zeros((0, 3), np.int64)
        else:
            for _ in range(ne):
                t = [int(x) for x in _readline(f).split()]
                etype = t[1] if etype is None else etype
                if t[1] != etype:
                    raise RuntimeError("Bad MSH file format")
                elems.append([x - 1 for x in t[3 + t[2]:3 + t[2] + ELEM_NODES[t[1]]]])
            E = np.array(elems, dtype=np.int64)
        if _readline(f) != "$EndElements":
            raise RuntimeError("Bad MSH file format")
        fields = {}
        while True:
            hdr = _readline(f)
            if not hdr:
                break
            if hdr not in ("$NodeData", "$ElementData", "$ElementNodeData"):
                continue
            nstr = int(_readline(f))
            name = [_readline(f).strip('"') for _ in range(nstr)][0]
            for _ in range(int(_readline(f))):
                _readline(f)
            ntag = int(_readline(f))
            itags = [int(_readline(f)) for _ in range(ntag)]
            dim, cnt = itags[1], itags[2]
            if hdr == "$ElementNodeData":                    # elem_idx nodesPerElem values (MSHFieldWriter.hh:284-303)
                rows = []
                for _ in range(cnt):
                    if binary:
                        _, npe_ = struct.unpack("<2i", f.read(8))
                        rows.append(np.frombuffer(f.read(8 * npe_ * dim), dtype="<f8").reshape(npe_, dim))
                    else:
                        t = _readline(f).split()
                        rows.append(np.array([float(x) for x in t[2:]]).reshape(int(t[1]), dim))
                fields[name] = ("element node", np.stack(rows))
                _readline(f)
                continue
            if binary:
                rec = np.frombuffer(f.read(cnt * (4 + 8 * dim)), dtype=np.dtype([("i", "<i4"), ("v", "<f8", dim)]))
                vals = rec["v"].reshape(cnt, dim).astype(np.float64)
            else:
                vals = np.array([[float(x) for x in _readline(f).split()[1:]] for _ in range(cnt)]).reshape(cnt, dim)
            fields[name] = ("node" if hdr == "$NodeData" else "element", vals)
            _readline(f)
    return V, E, fields


def load_off(path):
    """Object File Format surface meshes (MeshIO_OFF, MeshIO.cc): 'OFF', counts, vertices, polygons.
    Returns (vertices [n,3], elements [m,k])."""
    with open(path) as f:
        tok = f.read().split()
    if tok[0] != "OFF":
        raise RuntimeError("Bad OFF file format")
    nv, nf = int(tok[1]), int(tok[2])
    pos = 4
    V = np.array(tok[pos:pos + 3 * nv], dtype=np.float64).reshape(nv, 3)
    pos += 3 * nv
    E = []
    for _ in range(nf):
        k = int(tok[pos])
        E.append([int(x) for x in tok[pos + 1:pos + 1 + k]])
        pos += 1 + k
    return V, np.array(E, dtype=np.int64)


def load_obj(path):
    """Wavefront OBJ surface meshes (MeshIO_OBJ, MeshIO.cc): `v x y z` and `f a b c ...` records (1-based, `a/t/n` forms
    accepted, negative = relative indices); everything else is skipped."""
    V, E = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                V.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = [int(x.split("/")[0]) for x in t[1:]]
                E.append([i - 1 if i > 0 else len(V) + i for i in idx])
    return np.array(V, dtype=np.float64).reshape(-1, 3), np.array(E, dtype=np.int64)


def load_medit(path):
    """MEDIT .mesh files (MeshIO_MEDIT): `Vertices` (x y [z] ref) and `Triangles` / `Tetrahedra` (1-based indices + ref).
    Tetrahedra win when both are present (the triangles are then the boundary)."""
    with open(path) as f:
        tok = f.read().split()
    dim, pos = 3, 0
    V, tris, tets = None, None, None
    while pos < len(tok):
        key = tok[pos].lower()
        if key == "dimension":
            dim = int(tok[pos + 1]); pos += 2
        elif key == "vertices":
            n = int(tok[pos + 1]); pos += 2
            V = np.array(tok[pos:pos + n * (dim + 1)], dtype=np.float64).reshape(n, dim + 1)[:, :dim]; pos += n * (dim + 1)
        elif key in ("triangles", "tetrahedra"):
            k = 3 if key == "triangles" else 4
            n = int(tok[pos + 1]); pos += 2
            E = np.array(tok[pos:pos + n * (k + 1)], dtype=np.int64).reshape(n, k + 1)[:, :k] - 1; pos += n * (k + 1)
            if k == 3:
                tris = E
            else:
                tets = E
        elif key == "end":
            break
        else:
            pos += 1
    if V is None or (tris is None and tets is None):
        raise RuntimeError("Bad MEDIT file format")
    if dim == 2:
        V = np.column_stack([V, np.zeros(len(V))])
    return V, (tets if tets is not None else tris)


def load_mesh(path):
    """MeshIO::load dispatch on the extension (.msh / .off / .obj / .mesh)."""
    low = path.lower()
    if low.endswith(".off"):
        V, E = load_off(path)
        return V, E, {}
    if low.endswith(".obj"):
        V, E = load_obj(path)
        return V, E, {}
    if low.endswith(".mesh"):
        V, E = load_medit(path)
        return V, E, {}
    return load_msh(path)


def upsample_interpolant(values, dim):
    """Degree-1 per-element interpolant values [nElem, dim+1, k] -> degree-2 nodal values [nElem, nodesPerElem, k]
    (SymmetricMatrixInterpolant upsampling in Simulate_cli.cc:216-229): corners, then the edge midpoints in the
    reference's edge order (Simplex.hh:43-47)."""
    edges = [(0, 1), (1, 2), (2, 0)] if dim == 2 else [(0, 1), (1, 2), (2, 0), (0, 3), (2, 3), (1, 3)]
    values = np.asarray(values)
    mids = np.stack([0.5 * (values[:, a] + values[:, b]) for a, b in edges], axis=1)
    return np.concatenate([values, mids], axis=1)


class MSHFieldWriter:
    """== MSHFieldWriter (binary by default, like the reference)."""

    def __init__(self, path, nodes, elements, binary=True):
        self.f = open(path, "wb")
        self.binary = binary
        nodes = np.asarray(nodes, dtype=np.float64)
        if nodes.shape[1] == 2:
            nodes = np.column_stack([nodes, np.zeros(len(nodes))])
        elements = np.asarray(elements, dtype=np.int64)
        self.n_nodes, self.n_elems = len(nodes), len(elements)
        etype = TYPE_FOR_NODES[elements.shape[1]]
        w = self.f.write
        w(b"$MeshFormat\n2.2 %d 8\n" % (1 if binary else 0))
        if binary:
            w(struct.pack("<i", 1) + b"\n")
        w(b"$EndMeshFormat\n$Nodes\n%d\n" % len(nodes))
        if binary:
            rec = np.empty(len(nodes), dtype=np.dtype([("i", "<i4"), ("p", "<f8", 3)]))
            rec["i"], rec["p"] = np.arange(1, len(nodes) + 1), nodes
            w(rec.tobytes() + b"\n")
        else:
            for i, p in enumerate(nodes):
                w(("%d %.17g %.17g %.17g\n" % (i + 1, p[0], p[1], p[2])).encode())
        w(b"$EndNodes\n$Elements\n%d\n" % len(elements))
        if binary:
            if len(elements):
                w(struct.pack("<3i", etype, len(elements), 0))
                data = np.empty((len(elements), 1 + elements.shape[1]), dtype="<i4")
                data[:, 0], data[:, 1:] = np.arange(1, len(elements) + 1), elements + 1
                w(data.tobytes())
            w(b"\n")
        else:
            for i, e in enumerate(elements):
                w(("%d %d 0 %s\n" % (i + 1, etype, " ".join(str(int(x) + 1) for x in e))).encode())
        w(b"$EndElements\n")

    def addField(self, name, values, domain):
        """values: [n] scalar, [n, 2|3] vector, or [n, flatLen] symmetric matrices in the reference's
        flattened order (written as padded 3x3 scanline, MSHFieldWriter.hh:160-170)."""
        values = np.asarray(values, dtype=np.float64)
        n = self.n_nodes if domain == "node" else self.n_elems
        if values.ndim == 1:
            values = values[:, None]
        if len(values) != n:
            raise RuntimeError("Invalid field domain size.")
        k = values.shape[1]
        if domain == "element" and k in (3, 6) and name in ("strain", "stress") or k == 6:
            N = 2 if k == 3 else 3
            idx = {2: [(0, 0), (1, 1), (0, 1)], 3: [(0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1)]}[N]
            out = np.zeros((n, 3, 3))
            for q, (a, b) in enumerate(idx):
                out[:, a, b] = out[:, b, a] = values[:, q]
            out = out.reshape(n, 9)
        elif k == 2:
            out = np.column_stack([values, np.zeros(n)])
        elif k in (1, 3):
            out = values
        else:
            raise RuntimeError("Invalid field dimension.")
        w = self.f.write
        w(("$%s\n1\n\"%s\"\n0\n3\n0\n%d\n%d\n" % ("NodeData" if domain == "node" else "ElementData", name, out.shape[1], n)).encode())
        if self.binary:
            rec = np.empty(n, dtype=np.dtype([("i", "<i4"), ("v", "<f8", out.shape[1])]))
            rec["i"], rec["v"] = np.arange(1, n + 1), out.reshape(n, -1) if out.shape[1] > 1 else out
            w(rec.tobytes() + b"\n")
        else:
            for i in range(n):
                w(("%d %s\n" % (i + 1, " ".join("%.17g" % x for x in out[i]))).encode())
        w(("$End%s\n" % ("NodeData" if domain == "node" else "ElementData")).encode())

    def addElementNodeField(self, name, values):
        """Per-element interpolants as $ElementNodeData (MSHFieldWriter.hh:262-306): values [nElem, nodesPerElem, k] with
        k = 1 | 2 | 3 (vectors padded to 3) or flatLen (symmetric matrices, padded 3x3 scanline)."""
        values = np.asarray(values, dtype=np.float64)
        if values.ndim != 3 or len(values) != self.n_elems:
            raise RuntimeError("Vector-of-interpolants must be per-element.")
        n, npe, k = values.shape
        if k == 6 or (k == 3 and (name.startswith("strain") or name.startswith("stress"))):
            N = 2 if k == 3 else 3
            idx = {2: [(0, 0), (1, 1), (0, 1)], 3: [(0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1)]}[N]
            out = np.zeros((n, npe, 3, 3))
            for q, (a, b) in enumerate(idx):
                out[:, :, a, b] = out[:, :, b, a] = values[:, :, q]
            out = out.reshape(n, npe, 9)
        elif k == 2:
            out = np.concatenate([values, np.zeros((n, npe, 1))], axis=2)
        elif k in (1, 3):
            out = values
        else:
            raise RuntimeError("Invalid field dimension.")
        w = self.f.write
        w(("$ElementNodeData\n1\n\"%s\"\n0\n3\n0\n%d\n%d\n" % (name, out.shape[2], n)).encode())
        if self.binary:
            rec = np.empty(n, dtype=np.dtype([("i", "<i4"), ("n", "<i4"), ("v", "<f8", npe * out.shape[2])]))
            rec["i"], rec["n"], rec["v"] = np.arange(1, n + 1), npe, out.reshape(n, -1)
            w(rec.tobytes() + b"\n")
        else:
            for i in range(n):
                w(("%d %d %s\n" % (i + 1, npe, " ".join("%.17g" % x for x in out[i].ravel()))).encode())
        w(b"$EndElementNodeData\n")

    def close(self):
        self.f.close()
