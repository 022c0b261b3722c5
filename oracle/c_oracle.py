"""ctypes loader of the plain-C oracle (oracle/c/meshfem_oracle.c). TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__ and bench.py's cpu_baseline leg, never by meshfem_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmeshfem_oracle.so")


def build(force=False):
    src = os.path.join(HERE, "c", "meshfem_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(src) > os.path.getmtime(LIB):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        # -march=native is resolved on the machine that builds; the GPU box rebuilds if the
        # shipped .so is older than the source, otherwise uses the shipped one (x86-64-v3 safe flags)
        subprocess.check_call(["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-fPIC", "-std=c11", "-shared",
                               "-o", LIB, src, "-lm"])
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.oracle_num_threads.restype = C.c_int
        _lib.oracle_assemble_csc.restype = C.c_int64
        _lib.oracle_assemble_fused.restype = C.c_double
    return _lib


def num_threads():
    return load().oracle_num_threads()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def element_stiffness(dim, deg, elem_nodes, vert_pos, D):
    """Upper triangle of every Ke (lower = 0), LinearElasticity.hh:165-232 loop structure."""
    lib = load()
    en = np.ascontiguousarray(elem_nodes, np.int32)
    vp = np.ascontiguousarray(vert_pos, np.float64)
    D = np.ascontiguousarray(D, np.float64)
    fl = dim * (dim + 1) // 2
    nD = D.size // (fl * fl)
    nE, npe = en.shape
    ks = npe * dim
    Ke = np.empty((nE, ks, ks))
    vol = np.empty(nE)
    lib.oracle_element_stiffness(C.c_int(dim), C.c_int(deg), C.c_int64(nE), _p(en), C.c_int(npe), _p(vp), _p(D),
                                 C.c_int64(nD), _p(Ke), _p(vol))
    return Ke, vol


def assemble_csc(dim, deg, elem_nodes, vert_pos, D, n_dof, dof_for_node=None):
    """Reference-style assembly (threaded Ke -> serial upper-triplet push -> sumRepeated -> CSC).
    Returns (Ap, Ai, Ax, times{ke,push,compress,total} in seconds)."""
    lib = load()
    en = np.ascontiguousarray(elem_nodes, np.int32)
    vp = np.ascontiguousarray(vert_pos, np.float64)
    D = np.ascontiguousarray(D, np.float64)
    fl = dim * (dim + 1) // 2
    nD = D.size // (fl * fl)
    nE, npe = en.shape
    n = dim * n_dof
    cap = nE * (npe * dim) ** 2 // 2 + n
    Ap = np.empty(n + 1, np.int64)
    Ai = np.empty(cap, np.int64)
    Ax = np.empty(cap)
    times = np.zeros(4)
    dm = None if dof_for_node is None else np.ascontiguousarray(dof_for_node, np.int32)
    nnz = lib.oracle_assemble_csc(C.c_int(dim), C.c_int(deg), C.c_int64(nE), _p(en), C.c_int(npe), _p(vp),
                                  None if dm is None else _p(dm), C.c_int64(n_dof), _p(D), C.c_int64(nD),
                                  _p(Ap), _p(Ai), _p(Ax), C.c_int64(cap), _p(times))
    if nnz < 0:
        raise RuntimeError("oracle_assemble_csc failed (%d)" % nnz)
    return Ap, Ai[:nnz].copy(), Ax[:nnz].copy(), dict(ke=times[0], push=times[1], compress=times[2], total=times[3])


def assemble_fused(dim, deg, elem_nodes, vert_pos, D, n_dof, Ap, Ai, dof_for_node=None):
    """The tuned host assembly (oracle_assemble_fused): Ke on the thread's stack, entries added atomically into the known CSC pattern
    (Ap, Ai of assemble_csc). Returns (Ax, seconds, entries not in the pattern = exact zeros the port pruned)."""
    lib = load()
    en = np.ascontiguousarray(elem_nodes, np.int32)
    vp = np.ascontiguousarray(vert_pos, np.float64)
    D = np.ascontiguousarray(D, np.float64)
    fl = dim * (dim + 1) // 2
    nD = D.size // (fl * fl)
    nE, npe = en.shape
    Ap = np.ascontiguousarray(Ap, np.int64)
    Ai = np.ascontiguousarray(Ai, np.int64)
    Ax = np.empty(len(Ai))
    missed = C.c_int64(0)
    dm = None if dof_for_node is None else np.ascontiguousarray(dof_for_node, np.int32)
    t = lib.oracle_assemble_fused(C.c_int(dim), C.c_int(deg), C.c_int64(nE), _p(en), C.c_int(npe), _p(vp), None if dm is None else _p(dm),
                                  _p(D), C.c_int64(nD), _p(Ap), _p(Ai), _p(Ax), C.c_int64(dim * n_dof), C.byref(missed))
    return Ax, t, missed.value


def extend_add(P, S, U, loc, threads=1):
    """Multifrontal extend-add (oracle/direct_solve.py): the LOWER triangle of the child's update matrix U lands in the parent's
    panel P ((ns + nb) x ns) and Schur block S (nb x nb) at the increasing positions loc; all arrays C-contiguous float64."""
    lib = load()
    loc = np.ascontiguousarray(loc, np.int64)
    assert P.flags.c_contiguous and S.flags.c_contiguous and U.flags.c_contiguous
    lib.oracle_extend_add(_p(P), _p(S), C.c_int64(P.shape[1]), C.c_int64(S.shape[0]), _p(U), C.c_int64(U.shape[0]), _p(loc), C.c_int(int(threads)))


def zeros(shape, threads=0):
    """np.zeros; big arrays are filled by all threads (threads != 1: the caller is not itself one of many worker threads)."""
    if threads == 1 or int(np.prod(shape)) < (1 << 22):
        return np.zeros(shape)
    a = np.empty(shape)
    load().oracle_zero(_p(a), C.c_int64(a.size))
    return a
