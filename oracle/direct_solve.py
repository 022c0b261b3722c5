"""Sparse direct solve for the CPU baseline: multifrontal Cholesky with a nested-dissection ordering. TEST INFRASTRUCTURE
ONLY (imported by tests/ and bench.py's cpu_baseline leg, never by meshfem_amd/).

What it stands in for: the reference factors the reduced stiffness matrix with CHOLMOD -- `CholmodFactorizer`,
SparseMatrices.hh:1984-2296: supernodal LL^T (`CHOLMOD_AUTO` picks supernodal for 3D problems), NESDIS / METIS nested
dissection among the orderings tried (:2243-2295), `cholmod_l_solve2` for the triangular solves. SuiteSparse is not installed in
this image (and un-fetchable), so the same METHOD is restated on numpy / LAPACK: a nested-dissection elimination tree whose nodes
are the supernodes, one dense frontal matrix per node (potrf + trsm + syrk, the BLAS-3 kernels CHOLMOD's supernodal
factorisation runs), extend-add of the children's update matrices. The dissection is geometric -- recursive bisection of the node
coordinates at the median, the separator being the nodes of the smaller-coordinate side that touch the other side in the graph
of K -- where NESDIS uses METIS' graph bisection; the fill it produces has the same O(n^(4/3)) growth. LAPACK runs on the
threads of the process' BLAS (threadpoolctl reports how many): that is the "TBB/CHOLMOD on the host cores" role.

`scipy.sparse.linalg.splu` (SuperLU, sequential, scalar updates) stays the oracle's direct solver at the sizes of the parity
tests; it cannot factor 5e5 - 1e6 unknowns of a 3D quadratic mesh in a benchmark's time budget, this can."""
import time
from contextlib import nullcontext as _nullcontext

import numpy as np
import scipy.linalg as sl
import scipy.sparse as sp

from .c_oracle import extend_add as _extend_add, zeros as _zeros          # threaded C loops


def _dissect(ids, pos, indptr, indices, leaf, out, depth=0):
    """Recursive bisection of the unknowns `ids`. Appends tree nodes (own unknowns, list of children) to `out` in
    post-order and returns the index of the subtree's root."""
    if len(ids) <= leaf:
        out.append((ids, []))
        return len(out) - 1
    p = pos[ids]
    ax = int(np.argmax(p.max(axis=0) - p.min(axis=0)))
    x = p[:, ax]
    # candidate cuts: distinct coordinate values around the median (a mesh whose nodes sit on planes -- the reference's grid
    # generator -- is cut best exactly ON such a plane: the separator is then one layer of nodes); the smallest separator wins
    qs = np.unique(np.quantile(x, [0.5, 0.42, 0.58, 0.46, 0.54], method="nearest"))
    side = _dissect.side
    cnt_all = np.diff(indptr)
    best = None
    for cut in qs:
        left = x <= cut
        if left.all() or not left.any():
            continue
        side[ids[left]] = 1
        side[ids[~left]] = 2
        L = ids[left]
        cnt = cnt_all[L]
        nb = indices[_ranges(indptr[L], cnt)]
        starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        hit = np.add.reduceat(np.concatenate([(side[nb] == 2).astype(np.int64), [0]]), np.minimum(starts, len(nb)))[:len(L)]
        touches = (hit > 0) & (cnt > 0)
        side[ids] = 0
        score = int(touches.sum()) + abs(int(left.sum()) - len(ids) // 2) // 8     # small separator, halves not too uneven
        if best is None or score < best[0]:
            best = (score, left, L, touches)
    if best is None:
        out.append((ids, []))
        return len(out) - 1
    _, left, L, touches = best
    sep = L[touches]
    Lr = L[~touches]
    R = ids[~left]
    kids = []
    for part in (Lr, R):
        if len(part):
            kids.append(_dissect(part, pos, indptr, indices, leaf, out, depth + 1))
    out.append((sep, kids))
    return len(out) - 1


def _ranges(starts, counts):
    """Concatenation of arange(s, s + c) for every (s, c)."""
    total = int(counts.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    rep = np.repeat(starts - np.concatenate([[0], np.cumsum(counts)[:-1]]), counts)
    return rep + np.arange(total)


class MultifrontalCholesky:
    """K = L L^T for a symmetric positive definite scipy matrix K (both triangles stored) whose unknowns come in groups of
    `block` per point of `coords` (3 displacement components per node). factor() then solve(b)."""

    def __init__(self, K, coords, block=1, leaf=None):
        K = sp.csr_matrix(K)
        n = K.shape[0]
        assert K.shape[0] == K.shape[1] and n == block * len(coords)
        self.n, self.block = n, block
        t0 = time.perf_counter()
        # dissect the POINT graph (one vertex per node), then expand to the scalar unknowns
        if block > 1:
            r = np.repeat(np.arange(n) // block, np.diff(K.indptr))
            G = sp.csr_matrix((np.ones(K.nnz, dtype=np.int8), (r, K.indices // block)), shape=(n // block, n // block))
            G.sum_duplicates()
        else:
            G = K
        npt = n // block
        leaf = leaf or max(32, 384 // block)
        _dissect.side = np.zeros(npt, dtype=np.int8)
        tree = []
        import sys
        sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
        root = _dissect(np.arange(npt, dtype=np.int64), np.asarray(coords, dtype=np.float64), G.indptr.astype(np.int64),
                        G.indices.astype(np.int64), leaf, tree)
        assert root == len(tree) - 1
        # elimination order = post-order of the tree; perm[new] = old (scalar unknowns)
        own = [np.sort(t[0]) for t in tree]
        order_pts = np.concatenate(own) if own else np.zeros(0, np.int64)
        assert len(order_pts) == npt
        perm = (order_pts[:, None] * block + np.arange(block)[None, :]).ravel()
        self.perm = perm
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        self.first = np.concatenate([[0], np.cumsum([len(o) * block for o in own])]).astype(np.int64)
        self.kids = [t[1] for t in tree]
        self.parent = np.full(len(tree), -1, dtype=np.int64)
        for k, ch in enumerate(self.kids):
            for c in ch:
                self.parent[c] = k
        # permuted matrix, lower triangle by columns: column j holds the rows >= j
        Kp = K[perm][:, perm].tocsc()
        Kp.sort_indices()
        self.Kp = Kp
        self.t_order = time.perf_counter() - t0
        self.L11, self.L21, self.bnd = [None] * len(tree), [None] * len(tree), [None] * len(tree)
        self.factor_nnz = 0
        self.flops = 0.0

    def _front(self, k, upd, dsyrk, threads=1):
        """Assembles and eliminates the front of tree node k; the children's update matrices are consumed from `upd`."""
        Kp, first = self.Kp, self.first
        indptr, indices, data = Kp.indptr, Kp.indices, Kp.data
        s0, s1 = int(first[k]), int(first[k + 1])
        ns = s1 - s0
        # boundary = rows > own range in the own columns, plus the children's boundaries beyond the own range
        lo, hi = indptr[s0], indptr[s1]
        rows = indices[lo:hi]
        cols = np.repeat(np.arange(s0, s1), np.diff(indptr[s0:s1 + 1]))
        below = rows >= s1
        parts = [rows[below]]
        for c in self.kids[k]:
            b = self.bnd[c]
            parts.append(b[b >= s1])
        bnd = np.unique(np.concatenate(parts)) if parts else np.zeros(0, np.int64)
        nb = len(bnd)
        # frontal matrix as a panel P = [F11; F21] ((ns + nb) x ns, lower part of F11) and the Schur block S = F22 (nb x nb, lower
        # part). K only feeds the panel: its entries among boundary unknowns are assembled where one of them is eliminated
        P = _zeros((ns + nb, ns), threads)
        S = _zeros((nb, nb), threads)
        inown = (rows >= s0) & (rows < s1)
        P[rows[inown] - s0, cols[inown] - s0] = data[lo:hi][inown]
        if nb:
            P[ns + np.searchsorted(bnd, rows[below]), cols[below] - s0] = data[lo:hi][below]
        for c in self.kids[k]:                      # extend-add of the children's update matrices (lower triangles)
            b, U = self.bnd[c], upd[c]
            if U is None or not len(b):
                continue
            loc = np.where(b < s1, b - s0, ns + np.searchsorted(bnd, b))
            _extend_add(P, S, U, loc)                # on the calling thread: an OpenMP team here fights the BLAS threads (measured 2x slower)
            upd[c] = None
        flops = 0.0
        if ns:
            L11 = sl.cholesky(P[:ns], lower=True, overwrite_a=True, check_finite=False)
            flops += ns ** 3 / 3.0
        else:
            L11 = np.zeros((0, 0))
        if nb and ns:
            L21 = sl.solve_triangular(L11, P[ns:].T, lower=True, check_finite=False).T      # nb x ns
            # S -= L21 L21^T on the lower triangle, in place: the row-major buffer of S read column-major is S^T, whose UPPER
            # triangle is the lower triangle of S
            L21f = np.asfortranarray(L21)
            dsyrk(alpha=-1.0, a=L21f, beta=1.0, c=S.T, lower=0, overwrite_c=True)
            flops += ns * ns * nb + ns * nb * nb
            L21 = np.ascontiguousarray(L21)
        else:
            L21 = np.zeros((nb, ns))
        self.L11[k], self.L21[k], self.bnd[k] = L11, L21, bnd
        upd[k] = S if nb else None
        self._node_flops[k] = flops
        self._node_nnz[k] = ns * (ns + 1) // 2 + ns * nb

    def factor(self, workers=1, blas_threads=None):
        """Numeric factorisation. workers > 1: independent subtrees of the elimination tree are factored concurrently by that many
        threads with single-threaded BLAS (LAPACK releases the interpreter lock), the fronts above them one after the other on
        `blas_threads` BLAS threads -- the two levels of parallelism a supernodal solver uses."""
        t0 = time.perf_counter()
        n_nodes = len(self.kids)
        upd = [None] * n_nodes
        dsyrk = sl.get_blas_funcs("syrk", dtype=np.float64)
        self._node_flops, self._node_nnz = np.zeros(n_nodes), np.zeros(n_nodes, dtype=np.int64)
        top = np.zeros(n_nodes, dtype=bool)             # nodes handled after the subtrees
        roots = [n_nodes - 1]
        if workers > 1:
            # split the tree from the root down until there are enough subtrees to keep the workers busy
            size = np.diff(self.first).astype(np.float64)
            for k in range(n_nodes):                     # post-order: children before parents
                for c in self.kids[k]:
                    size[k] += size[c]
            while len(roots) < 4 * workers:
                big = max(roots, key=lambda r: size[r])
                if not self.kids[big]:
                    break
                roots.remove(big)
                top[big] = True
                roots.extend(self.kids[big])
        self.subtrees = len(roots)

        def subtree(r):
            stack, order = [r], []
            while stack:                                 # nodes of the subtree, then reversed pre-order = a valid elimination order
                k = stack.pop()
                order.append(k)
                stack.extend(self.kids[k])
            for k in reversed(order):
                self._front(k, upd, dsyrk)
        if workers > 1 and len(roots) > 1:
            from concurrent.futures import ThreadPoolExecutor
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=1, user_api="blas"):
                with ThreadPoolExecutor(max_workers=workers) as ex:
                    list(ex.map(subtree, sorted(roots, key=lambda r: -size[r])))
            self.t_subtrees = time.perf_counter() - t0
            with threadpool_limits(limits=blas_threads, user_api="blas") if blas_threads else _nullcontext():
                for k in range(n_nodes):
                    if top[k]:
                        self._front(k, upd, dsyrk, blas_threads or 1)
        else:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=blas_threads, user_api="blas") if blas_threads else _nullcontext():
                subtree(n_nodes - 1)
            self.t_subtrees = time.perf_counter() - t0
        self.flops = float(self._node_flops.sum())
        self.factor_nnz = int(self._node_nnz.sum())
        self.t_factor = time.perf_counter() - t0
        return self

    def solve(self, b):
        t0 = time.perf_counter()
        y = np.asarray(b, dtype=np.float64)[self.perm].copy()
        first = self.first
        for k in range(len(self.kids)):                  # forward: L y = b
            s0, s1 = int(first[k]), int(first[k + 1])
            if s1 > s0:
                y[s0:s1] = sl.solve_triangular(self.L11[k], y[s0:s1], lower=True, check_finite=False)
                if len(self.bnd[k]):
                    y[self.bnd[k]] -= self.L21[k] @ y[s0:s1]
        for k in range(len(self.kids) - 1, -1, -1):      # backward: L^T x = y
            s0, s1 = int(first[k]), int(first[k + 1])
            if s1 > s0:
                r = y[s0:s1]
                if len(self.bnd[k]):
                    r = r - self.L21[k].T @ y[self.bnd[k]]
                y[s0:s1] = sl.solve_triangular(self.L11[k], r, lower=True, trans="T", check_finite=False)
        x = np.empty_like(y)
        x[self.perm] = y
        self.t_solve = time.perf_counter() - t0
        return x


def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([t.get("num_threads", 1) for t in threadpool_info() if t.get("user_api") == "blas"] or [1])
    except Exception:
        return 1
