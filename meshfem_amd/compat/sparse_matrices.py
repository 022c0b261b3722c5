"""`sparse_matrices` (src/python_bindings/sparse_matrices.cc:15-66): Triplet, TripletMatrix and SPSDSystem.
SPSDSystem hands the matrix to the HIP library (`mfh_matrix_set_upper_triplets`): full symmetric CSR in HBM,
masked Jacobi-preconditioned CG instead of CHOLMOD, same fixVariables / solve semantics
(SparseMatrices.hh:2389-2500,2515-2606)."""
import struct

import numpy as np

from meshfem_amd.core import Context


class Triplet:
    def __init__(self, i=0, j=0, v=0.0):
        self.i, self.j, self.v = int(i), int(j), float(v)

    def __repr__(self):
        return "%d\t%d\t%s" % (self.i, self.j, repr(self.v))


class TripletMatrix:
    """Sparse matrix in triplet (COO) format; symmetry_mode 'NONE' or 'UPPER_TRIANGLE' (SparseMatrices.hh:211-234)."""

    def __init__(self, m=0, n=0):
        self.m, self.n = int(m), int(n)
        self._i, self._j, self._v = [], [], []
        self.symmetry_mode = "NONE"

    @staticmethod
    def fromArrays(m, n, i, j, v, symmetry_mode="NONE"):
        A = TripletMatrix(m, n)
        A._i, A._j, A._v = list(map(int, i)), list(map(int, j)), list(map(float, v))
        A.symmetry_mode = symmetry_mode
        return A

    @property
    def nnz(self):
        return len(self._v)

    def entries(self):
        return iter(Triplet(i, j, v) for i, j, v in zip(self._i, self._j, self._v))

    def addNZ(self, i, j, v):
        if i >= self.m or j >= self.n:
            raise RuntimeError("Index out of bounds")
        self._i.append(int(i)); self._j.append(int(j)); self._v.append(float(v))

    def arrays(self):
        return np.array(self._i, dtype=np.int64), np.array(self._j, dtype=np.int64), np.array(self._v, dtype=np.float64)

    def sumRepeated(self):
        """Sort by (column, row), sum duplicates, drop exact zeros (SparseMatrices.hh:280-374)."""
        i, j, v = self.arrays()
        order = np.lexsort((i, j))
        i, j, v = i[order], j[order], v[order]
        if len(v):
            head = np.concatenate([[True], (i[1:] != i[:-1]) | (j[1:] != j[:-1])])
            seg = np.cumsum(head) - 1
            vs = np.zeros(seg[-1] + 1)
            np.add.at(vs, seg, v)
            i, j, v = i[head], j[head], vs
            keep = v != 0.0
            i, j, v = i[keep], j[keep], v[keep]
        self._i, self._j, self._v = i.tolist(), j.tolist(), v.tolist()

    def reflectUpperTriangle(self):
        i, j, v = self.arrays()
        up = i <= j
        i, j, v = i[up], j[up], v[up]
        off = i != j
        self._i = np.concatenate([i, j[off]]).tolist()
        self._j = np.concatenate([j, i[off]]).tolist()
        self._v = np.concatenate([v, v[off]]).tolist()
        self.symmetry_mode = "NONE"

    def diag(self):
        d = np.zeros(min(self.m, self.n))
        i, j, v = self.arrays()
        on = i == j
        np.add.at(d, i[on], v[on])
        return d

    def apply(self, x):
        x = np.asarray(x, dtype=np.float64)
        i, j, v = self.arrays()
        y = np.zeros(self.m)
        np.add.at(y, i, v * x[j])
        if self.symmetry_mode == "UPPER_TRIANGLE":
            off = i != j
            np.add.at(y, j[off], v[off] * x[i[off]])
        return y

    def toSciPy(self):
        import scipy.sparse as sp
        i, j, v = self.arrays()
        return sp.coo_matrix((v, (i, j)), shape=(self.m, self.n)).tocsc()

    def dumpBinary(self, path):
        """uint64 nnz, then all row indices, all column indices (uint64 each), all values (double)
        (SparseMatrices.hh:623-645)."""
        i, j, v = self.arrays()
        with open(path, "wb") as f:
            np.array([len(v)], dtype="<u8").tofile(f)
            i.astype("<u8").tofile(f)
            j.astype("<u8").tofile(f)
            v.astype("<f8").tofile(f)

    def readBinary(self, path):
        """Sizes are inferred from the largest indices, like the reference (:647-670)."""
        with open(path, "rb") as f:
            (nnz,) = struct.unpack("<Q", f.read(8))
            i = np.frombuffer(f.read(8 * nnz), dtype="<u8")
            j = np.frombuffer(f.read(8 * nnz), dtype="<u8")
            v = np.frombuffer(f.read(8 * nnz), dtype="<f8")
        self.m = int(i.max()) + 1 if nnz else 0
        self.n = int(j.max()) + 1 if nnz else 0
        self._i, self._j, self._v = i.astype(np.int64).tolist(), j.astype(np.int64).tolist(), v.tolist()


class SPSDSystem:
    """A (constrained) SPSD system that can be solved for several right-hand sides. Constraint rows `C x = C_rhs`
    (the Lagrange-multiplier branch the reference hands to UMFPACK, SparseMatrices.hh:2340-2348,2572-2590) are eliminated
    with a Schur complement around SPD solves: x = x0 - Y lambda with K x0 = b, K Y = C^T, (C Y) lambda = C x0 - C_rhs,
    i.e. k + 1 PCG solves for k rows (the second and later right-hand sides reuse Y). K must be non-singular on the free
    variables; singular elasticity systems with rigid-motion rows are handled at the Simulator level instead."""

    def __init__(self, K, C=None, C_rhs=None, device=0):
        self._C = self._Crhs = self._Y = None
        if C is not None:
            if C.n != K.n:
                raise RuntimeError("Constraint matrix has the wrong number of columns")
            ci, cj, cv = C.arrays()
            self._C = np.zeros((C.m, C.n))
            np.add.at(self._C, (ci.astype(np.int64), cj.astype(np.int64)), cv)
            self._Crhs = np.zeros(C.m) if C_rhs is None else np.asarray(C_rhs, dtype=np.float64).copy()
            if self._Crhs.shape != (C.m,):
                raise RuntimeError("Constraint rhs has the wrong size")
        if K.m != K.n:
            raise RuntimeError("K must be square")
        i, j, v = K.arrays()
        if K.symmetry_mode != "UPPER_TRIANGLE":            # SPSDSystem::set keeps the upper triangle (:2337)
            up = i <= j
            i, j, v = i[up], j[up], v[up]
        elif (i > j).any():
            raise RuntimeError("entry below the diagonal in an UPPER_TRIANGLE matrix")
        self.ctx = Context(device)
        self.ctx.matrix_set_upper_triplets(K.m, i, j, v)
        self.n = K.m
        self.rtol, self.maxit = 1e-10, 100000
        self.info = None

    def fixVariables(self, fixedVars, fixedVarValues, keepFactorization=False):
        if len(fixedVars) != len(fixedVarValues):
            raise RuntimeError("Fixed variable index and value arrays must be the same size.")
        if len(fixedVars):
            self.ctx.fix_variables(np.asarray(fixedVars, dtype=np.int64), np.asarray(fixedVarValues, dtype=np.float64))
            self._fixed = np.concatenate([getattr(self, "_fixed", np.zeros(0, np.int64)), np.asarray(fixedVars, dtype=np.int64)])
            self._Y = None

    def setForceSupernodal(self, force):
        pass                                               # a CHOLMOD tuning knob; meaningless for PCG

    def solve(self, b):
        b = np.asarray(b, dtype=np.float64)
        if b.shape != (self.n,):
            raise RuntimeError("Bad rhs size")
        x = self.ctx.solve(b, rtol=self.rtol, maxit=self.maxit)
        self.info = self.ctx.last_info
        if self._C is None:
            return x
        Cf = self._C.copy()
        Cf[:, getattr(self, "_fixed", np.zeros(0, np.int64))] = 0.0       # rows restricted to the free variables
        if self._Y is None:                                               # K Y = C^T with the fixed variables at zero
            self.ctx.set_option("solve_homogeneous", 1)
            try:
                self._Y = np.stack([self.ctx.solve(Cf[r], rtol=self.rtol, maxit=self.maxit) for r in range(len(Cf))])
            finally:
                self.ctx.set_option("solve_homogeneous", 0)
        S = Cf @ self._Y.T
        try:
            lam = np.linalg.solve(S, self._C @ x - self._Crhs)
        except np.linalg.LinAlgError:
            raise RuntimeError("constraint rows are linearly dependent on the free variables")
        return x - lam @ self._Y


_SYM_CODES = {"NONE": 0, "UPPER_TRIANGLE": 1, "LOWER_TRIANGLE": 2}      # SparseMatrices.hh SymmetryMode


class SuiteSparseMatrix:
    """Sparse matrix in compressed-column format (CSCMatrix<SuiteSparse_long, double>, SparseMatrices.hh:1300-1790;
    binding sparse_matrices.cc:67-141): m, n, nz, Ap, Ai, Ax, symmetry_mode. `solve` hands an UPPER_TRIANGLE matrix to the
    HIP PCG (the binding builds a CholmodFactorizer); plain Python attributes, so instances pickle like the binding's."""

    def __init__(self, arg=None):
        self.m = self.n = self.nz = 0
        self.Ap, self.Ai, self.Ax = np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(0)
        self.symmetry_mode = "NONE"
        if isinstance(arg, str):
            self.readBinary(arg)
        elif arg is not None:
            self.setFromTMatrix(arg)

    def setFromTMatrix(self, tm):                            # CSCMatrix::setFromTMatrix (:422-447): sorted, repeats summed
        import scipy.sparse as sp
        i, j, v = tm.arrays()
        A = sp.coo_matrix((v, (i.astype(np.int64), j.astype(np.int64))), shape=(tm.m, tm.n)).tocsc()
        A.sum_duplicates(); A.sort_indices()
        self.m, self.n, self.nz = tm.m, tm.n, int(A.nnz)
        self.Ap, self.Ai, self.Ax = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)
        self.symmetry_mode = tm.symmetry_mode

    def getTripletMatrix(self):
        cols = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.Ap))
        return TripletMatrix.fromArrays(self.m, self.n, self.Ai, cols, self.Ax, symmetry_mode=self.symmetry_mode)

    def toSciPy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.Ax, self.Ai, self.Ap), shape=(self.m, self.n))

    def setZero(self): self.Ax[:] = 0.0
    def fill(self, v): self.Ax[:] = v
    def trace(self): return float(self.toSciPy().diagonal().sum())

    def apply(self, vec, transpose=False):                   # symmetric storage expands like CSCMatrix::apply
        x = np.asarray(vec, dtype=np.float64)
        A = self.toSciPy()
        if self.symmetry_mode != "NONE":
            import scipy.sparse as sp
            A = A + (sp.triu(A, 1) if self.symmetry_mode == "UPPER_TRIANGLE" else sp.tril(A, -1)).T
        return (A.T if transpose else A) @ x

    def solve(self, b, device=0, rtol=1e-10, maxit=100000):
        if self.symmetry_mode != "UPPER_TRIANGLE":
            raise RuntimeError("Only symmetric matrices are currently supported")
        b = np.asarray(b, dtype=np.float64)
        c = Context(device)
        tm = self.getTripletMatrix()
        i, j, v = tm.arrays()
        c.matrix_set_upper_triplets(self.m, i, j, v)
        x = c.solve(b, rtol=rtol, maxit=maxit)
        c.close()
        return x

    def dumpBinary(self, path):                              # SparseMatrices.hh:1448-1471
        with open(path, "wb") as f:
            np.array([self.m, self.n, self.nz], dtype=np.int64).tofile(f)
            np.array([_SYM_CODES[self.symmetry_mode]], dtype=np.uint32).tofile(f)
            np.asarray(self.Ap, dtype=np.int64).tofile(f); np.asarray(self.Ai, dtype=np.int64).tofile(f)
            np.asarray(self.Ax, dtype=np.float64).tofile(f)

    def readBinary(self, path):                              # :1473-1495
        with open(path, "rb") as f:
            self.m, self.n, self.nz = (int(x) for x in np.fromfile(f, dtype=np.int64, count=3))
            code = int(np.fromfile(f, dtype=np.uint32, count=1)[0])
            inv = {v: k for k, v in _SYM_CODES.items()}
            if code not in inv:
                raise RuntimeError("Invalid symmetry_mode")
            self.symmetry_mode = inv[code]
            self.Ap = np.fromfile(f, dtype=np.int64, count=self.n + 1)
            self.Ai = np.fromfile(f, dtype=np.int64, count=self.nz)
            self.Ax = np.fromfile(f, dtype=np.float64, count=self.nz)
