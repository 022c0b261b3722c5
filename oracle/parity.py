"""At-scale comparison of an assembled stiffness matrix with the oracle's. TEST INFRASTRUCTURE ONLY (imported by tests/ and
bench.py's cpu_baseline leg, never by meshfem_amd/).

Both sides hold the upper triangle of K in the order of TripletMatrix::sumRepeated (column-major, rows ascending inside a
column; SparseMatrices.hh:280-374) -- the library's mfh_export_upper_triplets (== dumpBinary content, :629-645) and the plain-C
oracle's CSC (oracle/c/meshfem_oracle.c: threaded Ke -> serial upper-triplet push -> sumRepeated -> CSC,
LinearElasticity.hh:1408-1466) -- so the comparison is a linear pass over two sorted key arrays."""
import numpy as np


def compare_upper_triplets_with_csc(i, j, v, Ap, Ai, Ax, prune_rel=1e-13):
    """(i, j, v): exported triplets; (Ap, Ai, Ax): oracle CSC of the upper triangle, n = len(Ap) - 1 scalar variables.
    Both sides prune EXACT zeros only; an entry whose contributions cancel to rounding noise (|v| <= prune_rel max|K|) may
    survive on one side and vanish on the other, so the pattern is compared after dropping those from both.
    Returns a dict: n, nnz of both sides, `pattern_identical` (the (row, col) sequences are equal element by element),
    `max_abs_err / max|K|` over the common pattern, and the largest magnitude among the entries only one side holds."""
    n = len(Ap) - 1
    i = np.asarray(i, np.int64); j = np.asarray(j, np.int64); v = np.asarray(v, np.float64)
    Ai = np.asarray(Ai, np.int64); Ax = np.asarray(Ax, np.float64)
    col = np.repeat(np.arange(n, dtype=np.int64), np.diff(np.asarray(Ap, np.int64)))
    kmax = float(max(np.abs(Ax).max(), np.abs(v).max()))
    thr = prune_rel * kmax
    key_g = j * n + i
    key_o = col * n + Ai
    sorted_g = bool(np.all(np.diff(key_g) > 0))           # strictly ascending: sumRepeated's order, no duplicates
    sorted_o = bool(np.all(np.diff(key_o) > 0))
    big_g, big_o = np.abs(v) > thr, np.abs(Ax) > thr
    kg, ko = key_g[big_g], key_o[big_o]
    identical = len(kg) == len(ko) and bool(np.array_equal(kg, ko))
    if identical:
        err = float(np.abs(v[big_g] - Ax[big_o]).max()) if len(kg) else 0.0
        only = 0.0
    else:
        common_g = np.isin(key_g, key_o, assume_unique=True)
        common_o = np.isin(key_o, key_g, assume_unique=True)
        err = float(np.abs(v[common_g] - Ax[common_o]).max()) if common_g.any() else float("inf")
        only = float(max(np.abs(v[~common_g]).max(initial=0.0), np.abs(Ax[~common_o]).max(initial=0.0)))
    return dict(n=int(n), nnz_hip=int(len(v)), nnz_oracle=int(len(Ax)), nnz_compared=int(len(kg)),
                order_is_sumRepeated=sorted_g and sorted_o, pattern_identical=identical,
                max_rel_err=err / kmax, max_unmatched_rel=only / kmax, upper_only=bool(np.all(i <= j)),
                prune_rel=prune_rel)
