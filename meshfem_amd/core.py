"""Thin numpy-facing wrapper over the C ABI (one `Context` == one mfh_ctx)."""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from ._lib import as_f64, as_i32, as_i64, ptr


def flat_len(dim):
    return dim * (dim + 1) // 2


class Context:
    def __init__(self, device=0):
        self.lib = L.load()
        h = C.c_void_p()
        st = self.lib.mfh_create(int(device), C.byref(h))
        if st != L.OK:
            raise L.MeshFEMHipError(st, "mfh_create(device=%d) failed with status %d: no usable HIP device "
                                        "(libmeshfem_hip has no CPU fallback)" % (device, st))
        self.h = h
        self.host_only = device == -1
        self.dim = self.deg = None
        self.op = L.OP_ELASTICITY
        self.external = False
        # experiments: MFH_OPTIONS="name=value,name=value" presets mfh_set_option on every new context (A/B runs of the
        # measurement scripts without editing them); unset in normal use
        for kv in filter(None, os.environ.get("MFH_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            self.set_option(k.strip(), float(v))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mfh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        if st != L.OK:
            raise L.MeshFEMHipError(st, self.lib.mfh_last_error(self.h).decode())

    # ---------------------------------------------------------------- mesh
    def mesh_build(self, elems, verts, deg):
        elems, verts = as_i32(elems), as_f64(verts)
        self.dim, self.deg = verts.shape[1], deg
        self._ck(self.lib.mfh_mesh_build(self.h, self.dim, deg, len(elems), len(verts), ptr(elems), ptr(verts)))
        self._sizes()

    def mesh_set(self, dim, deg, elem_nodes, node_pos, n_owned=None):
        """Explicit node table (any numbering); rows of K = the first n_owned nodes (halo nodes last)."""
        elem_nodes, node_pos = as_i32(elem_nodes), as_f64(node_pos)
        self.dim, self.deg = dim, deg
        n_node = len(node_pos)
        n_owned = n_node if n_owned is None else n_owned
        self._ck(self.lib.mfh_mesh_set(self.h, dim, deg, len(elem_nodes), n_node, n_owned, ptr(elem_nodes), ptr(node_pos)))
        self._sizes()

    def _sizes(self):
        v = [C.c_int64() for _ in range(5)]
        a, b = C.c_int32(), C.c_int32()
        self._ck(self.lib.mfh_mesh_sizes(self.h, *[C.byref(x) for x in v], C.byref(a), C.byref(b)))
        self.n_elem, self.n_node, self.n_vert, self.n_bdry_elem, self.n_bdry_node = [x.value for x in v]
        self.npe, self.npbe = a.value, b.value
        self.n_dof = self.n_node
        self.external = False

    def elem_nodes(self):
        out = np.empty((self.n_elem, self.npe), dtype=np.int32)
        self._ck(self.lib.mfh_mesh_get_elem_nodes(self.h, ptr(out)))
        return out

    def node_positions(self):
        out = np.empty((self.n_node, self.dim))
        self._ck(self.lib.mfh_mesh_get_node_positions(self.h, ptr(out)))
        return out

    def boundary_elem_nodes(self):
        out = np.empty((self.n_bdry_elem, self.npbe), dtype=np.int32)
        self._ck(self.lib.mfh_mesh_get_boundary_elem_nodes(self.h, ptr(out)))
        return out

    def boundary_elem_parents(self):
        out = np.empty(self.n_bdry_elem, dtype=np.int32)
        self._ck(self.lib.mfh_mesh_get_boundary_elem_parents(self.h, ptr(out)))
        return out

    def boundary_elem_internal(self):
        """1 for boundary elements lying on the periodic cell boundary (BoundaryElementData::isInternal)"""
        out = np.empty(self.n_bdry_elem, dtype=np.uint8)
        self._ck(self.lib.mfh_mesh_get_boundary_elem_internal(self.h, ptr(out)))
        return out

    def boundary_nodes(self):
        out = np.empty(self.n_bdry_node, dtype=np.int32)
        self._ck(self.lib.mfh_mesh_get_boundary_nodes(self.h, ptr(out)))
        return out

    def boundary_elem_geometry(self):
        vol = np.empty(self.n_bdry_elem)
        nrm = np.empty((self.n_bdry_elem, self.dim))
        self._ck(self.lib.mfh_mesh_get_boundary_elem_geometry(self.h, ptr(vol), ptr(nrm)))
        return vol, nrm

    def elem_volumes(self):
        out = np.empty(self.n_elem)
        self._ck(self.lib.mfh_mesh_get_elem_volumes(self.h, ptr(out)))
        return out

    def mesh_update_vertices(self, verts):
        """New vertex positions on the same connectivity (Simulator::updateMeshNodePositions): keeps every setup phase."""
        v = as_f64(np.asarray(verts, dtype=np.float64)[:, :self.dim])
        if v.shape != (self.n_vert, self.dim):
            raise ValueError("expected %d x %d vertex positions" % (self.n_vert, self.dim))
        self._ck(self.lib.mfh_mesh_update_vertices(self.h, ptr(v)))

    # ---------------------------------------------------------------- materials
    def material_isotropic(self, E, nu):
        self._ck(self.lib.mfh_material_isotropic(self.h, float(E), float(nu)))

    def material_const(self, D):
        D = as_f64(D)
        assert D.shape == (flat_len(self.dim),) * 2
        self._ck(self.lib.mfh_material_const(self.h, ptr(D)))

    def material_iso_field(self, E, nu):
        E, nu = as_f64(E), as_f64(nu)
        assert len(E) == len(nu) == self.n_elem
        self._ck(self.lib.mfh_material_iso_field(self.h, ptr(E), ptr(nu)))

    def material_ortho_field(self, params):
        params = as_f64(params)
        assert params.shape == (self.n_elem, 9 if self.dim == 3 else 4)
        self._ck(self.lib.mfh_material_ortho_field(self.h, ptr(params)))

    def material_tensor_field(self, D):
        D = as_f64(D)
        assert D.shape == (self.n_elem,) + (flat_len(self.dim),) * 2
        self._ck(self.lib.mfh_material_tensor_field(self.h, ptr(D)))

    def material_get(self, e):
        out = np.empty((flat_len(self.dim),) * 2)
        self._ck(self.lib.mfh_material_get(self.h, int(e), ptr(out)))
        return out

    # ---------------------------------------------------------------- DoF map
    def dof_map(self, dof_for_node, n_dof):
        if dof_for_node is None:
            self._ck(self.lib.mfh_dof_map(self.h, None, 0))
            self.n_dof = self.n_node
        else:
            d = as_i32(dof_for_node)
            self._ck(self.lib.mfh_dof_map(self.h, ptr(d), int(n_dof)))
            self.n_dof = int(n_dof)

    def apply_periodic_conditions(self, eps=1e-7):
        n = C.c_int64()
        self._ck(self.lib.mfh_apply_periodic_conditions(self.h, float(eps), C.byref(n)))
        self.n_dof = n.value
        return n.value

    def dof_map_partitioned(self, dof_for_node, n_dof, n_owned_dof):
        """DoF map of a row-partitioned context (mfh_mesh_set): local DoFs owned-first, the rows of K are the first n_owned_dof."""
        dm = as_i32(dof_for_node)
        self._ck(self.lib.mfh_dof_map_partitioned(self.h, ptr(dm), int(n_dof), int(n_owned_dof)))
        self.n_dof = int(n_dof)

    def get_dof_map(self):
        out = np.empty(self.n_node, dtype=np.int32)
        n = C.c_int64()
        self._ck(self.lib.mfh_get_dof_map(self.h, ptr(out), C.byref(n)))
        return out, n.value

    # ---------------------------------------------------------------- assembly
    def symbolic(self, with_scatter=False):
        self._ck(self.lib.mfh_symbolic(self.h, int(with_scatter)))

    def symbolic_sizes(self):
        a, b = C.c_int64(), C.c_int64()
        c, d = C.c_int32(), C.c_int32()
        self._ck(self.lib.mfh_symbolic_sizes(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(n_chunk=a.value, n_contrib=b.value, chunk_slots=c.value, max_row_len=d.value)

    def matrix_info(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.lib.mfh_matrix_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def matrix_storage(self):
        """(upper_only, stored_blocks): option "matrix_storage". matrix_info / export_bsr describe K itself either way; the
        symbolic phase (symbolic_get) describes what is stored."""
        u, n = C.c_int32(), C.c_int64()
        self._ck(self.lib.mfh_matrix_storage(self.h, C.byref(u), C.byref(n)))
        return bool(u.value), n.value

    def symbolic_get(self, with_scatter=False):
        nr, nc, _ = self.matrix_info()
        nnzb = self.matrix_storage()[1]
        sz = self.symbolic_sizes()
        out = dict(rowPtr=np.empty(nr + 1, np.int32), colIdx=np.empty(nnzb, np.int32),
                   chunkRow=np.empty(sz["n_chunk"] + 1, np.int32), contribPtr=np.empty(sz["n_chunk"] + 1, np.int64),
                   contribCode=np.empty(sz["n_contrib"], np.uint32), contribSlot=np.empty(sz["n_contrib"], np.uint16))
        sc = np.empty(self.n_elem * self.npe * self.npe, np.int32) if with_scatter else None
        self._ck(self.lib.mfh_symbolic_get(self.h, ptr(out["rowPtr"]), ptr(out["colIdx"]), ptr(out["chunkRow"]),
                                           ptr(out["contribPtr"]), ptr(out["contribCode"]), ptr(out["contribSlot"]), ptr(sc)))
        if with_scatter:
            out["scatterSlot"] = sc
        out.update(sz)
        return out

    @property
    def bs(self):
        """Variables per DoF = block edge of K: dim for elasticity, 1 for the scalar operators and for
        caller-supplied matrices."""
        return self.dim if (self.op == L.OP_ELASTICITY and not self.external) else 1

    def matrix_set_upper_triplets(self, n, i, j, v):
        """SPSDSystem(K) for a caller-supplied SPD matrix (upper-triangle triplets, repeats summed)."""
        i, j = np.ascontiguousarray(i, dtype=np.uint64), np.ascontiguousarray(j, dtype=np.uint64)
        v = as_f64(v)
        assert len(i) == len(j) == len(v)
        self._ck(self.lib.mfh_matrix_set_upper_triplets(self.h, int(n), len(v), ptr(i), ptr(j), ptr(v)))
        self.external, self.op = True, L.OP_ELASTICITY
        self.n_dof = self.n_node = int(n)
        self.dim = 1

    def set_operator(self, op):
        """OP_ELASTICITY (default) | OP_LAPLACIAN | OP_MASS: same mesh, pattern and kernels, 1x1 blocks
        for the scalar operators (Laplacian.hh, MassMatrix.hh, Poisson.hh)."""
        self._ck(self.lib.mfh_set_operator(self.h, int(op)))
        self.op = int(op)

    def precond_choice(self):
        """(kind, chosen automatically?, stretch of the mesh the choice looked at): mfh_precond_choice. With PRECOND_AUTO the choice is made now."""
        k, a, st = C.c_int32(), C.c_int32(), C.c_double()
        self._ck(self.lib.mfh_precond_choice(self.h, C.byref(k), C.byref(a), C.byref(st)))
        return k.value, bool(a.value), st.value

    def matrix_free_info(self):
        a, mo, mb = C.c_int32(), C.c_int32(), C.c_int32()
        nb, ne, ni = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.lib.mfh_matrix_free_info(self.h, C.byref(a), C.byref(mo), C.byref(nb), C.byref(ne), C.byref(ni), C.byref(mb)))
        return dict(active=bool(a.value), mode=mo.value, blocks=nb.value, block_rows=ne.value, interface_partials=ni.value,
                    max_block_rows=mb.value)

    def average_gradient(self, u_nodes):
        u = as_f64(u_nodes)
        out = np.empty((self.n_elem, self.dim))
        self._ck(self.lib.mfh_average_gradient(self.h, ptr(u), ptr(out)))
        return out

    def assemble(self, mode=L.ASSEMBLE_GATHER):
        self._ck(self.lib.mfh_assemble(self.h, int(mode)))

    def export_bsr(self):
        nr, nc, nnzb = self.matrix_info()
        rp, ci = np.empty(nr + 1, np.int32), np.empty(nnzb, np.int32)
        vals = np.empty((nnzb, self.bs, self.bs))
        self._ck(self.lib.mfh_export_bsr(self.h, ptr(rp), ptr(ci), ptr(vals)))
        return rp, ci, vals

    def export_scipy(self):
        import scipy.sparse as sp
        rp, ci, vals = self.export_bsr()
        nr, nc, _ = self.matrix_info()
        return sp.bsr_matrix((vals, ci, rp), shape=(nr * self.bs, nc * self.bs)).tocsr()

    def debug_device_node_tables(self):
        """(elem_nodes, node_pos) as the kernels see them (device copies; test hook)."""
        en, pos = np.empty((self.n_elem, self.npe), dtype=np.int32), np.empty((self.n_node, self.dim))
        self._ck(self.lib.mfh_debug_device_node_tables(self.h, ptr(en), ptr(pos)))
        return en, pos

    def export_upper_triplets(self):
        n = C.c_uint64(0)
        self._ck(self.lib.mfh_export_upper_triplets(self.h, None, None, None, C.byref(n)))
        i, j, v = np.empty(n.value, np.uint64), np.empty(n.value, np.uint64), np.empty(n.value)
        self._ck(self.lib.mfh_export_upper_triplets(self.h, ptr(i), ptr(j), ptr(v), C.byref(n)))   # n: entries written (zeros pruned)
        return i[:n.value], j[:n.value], v[:n.value]

    def element_stiffness(self, first=0, count=None):
        count = self.n_elem - first if count is None else count
        ks = self.npe * self.bs
        out = np.empty((count, ks, ks))
        self._ck(self.lib.mfh_element_stiffness(self.h, int(first), int(count), ptr(out)))
        return out

    # ---------------------------------------------------------------- solve
    def clear_fixed(self):
        self._ck(self.lib.mfh_clear_fixed(self.h))

    def fix_variables(self, vars_, vals=None):
        v = as_i64(vars_)
        x = None if vals is None else as_f64(vals)
        self._ck(self.lib.mfh_fix_variables(self.h, len(v), ptr(v), ptr(x)))

    def set_preconditioner(self, kind):
        self._ck(self.lib.mfh_set_preconditioner(self.h, int(kind)))

    def precond_info(self):
        a, m, t, note = C.c_int32(), C.c_int64(), C.c_double(), C.c_char_p()
        self._ck(self.lib.mfh_precond_info(self.h, C.byref(a), C.byref(m), C.byref(t), C.byref(note)))
        return dict(aggregates=a.value, coarse_dim=m.value, setup_ms=t.value, note=(note.value or b"").decode())

    def multigrid_info(self):
        a, b = C.c_int64(), C.c_int64()
        l0, l1, t = C.c_double(), C.c_double(), C.c_double()
        self._ck(self.lib.mfh_multigrid_info(self.h, C.byref(a), C.byref(b), C.byref(l0), C.byref(l1), C.byref(t)))
        return dict(fine_dof=a.value, coarse_dof=b.value, lambda_max_fine=l0.value, lambda_max_coarse=l1.value, setup_ms=t.value)

    def placement_info(self):
        """Kernel time (ms) of every candidate of the last placement trials (option placement_trials); [] when none ran."""
        out = (C.c_double * 16)()
        n = C.c_int32()
        self._ck(self.lib.mfh_placement_info(self.h, 16, out, C.byref(n)))
        return [float(out[k]) for k in range(min(16, n.value))]

    def multigrid_levels(self):
        """Aggregate levels of the multigrid hierarchy, finest first (mfh_multigrid_level_info)."""
        out = (C.c_int64 * (7 * 16))()
        n = C.c_int32()
        self._ck(self.lib.mfh_multigrid_level_info(self.h, 16, out, C.byref(n)))
        keys = ("aggregates", "rows", "entries", "partitioned", "peers", "halo_received", "owned_sent")
        return [dict(zip(keys, [int(out[7 * l + k]) for k in range(7)])) for l in range(min(n.value, 16))]

    def solve(self, f, rtol=1e-8, maxit=100000):
        f = as_f64(f)
        n = self.bs * self.n_dof
        nrhs = f.size // n
        assert f.size == nrhs * n
        u = np.empty_like(f)
        info = L.SolveInfo()
        st = self.lib.mfh_solve(self.h, nrhs, ptr(f), ptr(u), float(rtol), int(maxit), C.byref(info))
        self.last_info = info.as_dict()
        self._ck(st)
        return u

    def solve_batch(self, f, rtol=1e-8, maxit=100000):
        """All right-hand sides (rows of f) in batches sharing the operator passes; returns (u, [info per rhs])."""
        f = as_f64(f)
        n = self.bs * self.n_dof
        nrhs = f.size // n
        assert f.size == nrhs * n
        u = np.empty((nrhs, n))
        infos = (L.SolveInfo * nrhs)()
        st = self.lib.mfh_solve_batch(self.h, nrhs, ptr(f), ptr(u), float(rtol), int(maxit), infos)
        self.last_infos = [i.as_dict() for i in infos]
        self.last_info = self.last_infos[-1]
        self._ck(st)
        return u, self.last_infos

    def apply_K(self, u):
        u = as_f64(u)
        nr, nc, _ = self.matrix_info() if self._assembled_info() else (self.n_dof, self.n_dof, 0)
        out = np.empty(nr * self.bs)
        self._ck(self.lib.mfh_apply_K(self.h, ptr(u), ptr(out)))
        return out

    def _assembled_info(self):
        a = C.c_int64()
        return self.lib.mfh_matrix_info(self.h, C.byref(a), None, None) == L.OK

    # ---------------------------------------------------------------- Simulator-level helpers
    def bc_clear(self):
        self._ck(self.lib.mfh_bc_clear(self.h))

    def bc_dirichlet_box(self, mn, mx, value, relative=False, components=None):
        mn, mx, value = as_f64(mn), as_f64(mx), as_f64(value)
        mask = (1 << self.dim) - 1 if components is None else sum(1 << c for c in range(self.dim) if components[c])
        self._ck(self.lib.mfh_bc_dirichlet_box(self.h, ptr(mn), ptr(mx), int(relative), ptr(value), mask))

    def bc_neumann_box(self, mn, mx, value, kind=L.NEUMANN_TRACTION, relative=False):
        mn, mx = as_f64(mn), as_f64(mx)
        value = as_f64(np.atleast_1d(value))
        if len(value) < self.dim:
            value = np.concatenate([value, np.zeros(self.dim - len(value))])
        self._ck(self.lib.mfh_bc_neumann_box(self.h, ptr(mn), ptr(mx), int(relative), ptr(value), int(kind)))

    def bc_dirichlet_nodes(self, nodes, values, components=None):
        nodes = as_i64(nodes)
        values = as_f64(np.asarray(values, dtype=np.float64).reshape(len(nodes), self.dim))
        mask = (1 << self.dim) - 1 if components is None else sum(1 << c for c in range(self.dim) if components[c])
        self._ck(self.lib.mfh_bc_dirichlet_nodes(self.h, len(nodes), ptr(nodes), ptr(values), mask))

    def bc_neumann_elements(self, bdry_elems, tractions):
        be = as_i64(bdry_elems)
        t = as_f64(np.asarray(tractions, dtype=np.float64).reshape(len(be), self.dim))
        self._ck(self.lib.mfh_bc_neumann_elements(self.h, len(be), ptr(be), ptr(t)))

    def bc_delta_force(self, node, force):
        force = as_f64(force)
        self._ck(self.lib.mfh_bc_delta_force(self.h, int(node), ptr(force)))

    def bc_dirichlet_vars(self):
        n = C.c_int64(0)
        self._ck(self.lib.mfh_bc_dirichlet_vars(self.h, None, None, C.byref(n)))
        v, x = np.empty(n.value, np.int64), np.empty(n.value)
        self._ck(self.lib.mfh_bc_dirichlet_vars(self.h, ptr(v), ptr(x), C.byref(n)))
        return v, x

    def pin_node(self):
        n = C.c_int64()
        self._ck(self.lib.mfh_pin_node(self.h, C.byref(n)))
        return n.value

    def neumann_load(self):
        out = np.empty((self.n_dof, self.dim))
        self._ck(self.lib.mfh_neumann_load(self.h, ptr(out)))
        return out

    def constant_strain_load(self, cstrain_flat):
        e = as_f64(cstrain_flat)
        out = np.empty((self.n_dof, self.dim))
        self._ck(self.lib.mfh_constant_strain_load(self.h, ptr(e), ptr(out)))
        return out

    def sim_solve(self, f=None, use_pin=False, rtol=1e-8, maxit=100000):
        fp = None if f is None else as_f64(f)
        u = np.empty((self.n_node, self.bs))
        info = L.SolveInfo()
        st = self.lib.mfh_sim_solve(self.h, ptr(fp), int(use_pin), ptr(u), float(rtol), int(maxit), C.byref(info))
        self.last_info = info.as_dict()
        self._ck(st)
        return u

    def sim_solve_constrained(self, f=None, flags=0, rigid_motion_rhs=None, rtol=1e-8, maxit=100000):
        """Simulator::solve with the pin / translation / rotation constraints of assembleConstrainedSystem
        (flags: SOLVE_PIN | SOLVE_NO_RIGID_MOTION | SOLVE_ALLOW_ILL_POSED; 0 = posedness analysis)."""
        fp = None if f is None else as_f64(f)
        rr = None if rigid_motion_rhs is None else as_f64(rigid_motion_rhs)
        u = np.empty((self.n_node, self.bs))
        info = L.SolveInfo()
        st = self.lib.mfh_sim_solve_constrained(self.h, ptr(fp), int(flags), ptr(rr), 0 if rr is None else len(rr), ptr(u),
                                                float(rtol), int(maxit), C.byref(info))
        self.last_info = info.as_dict()
        self._ck(st)
        return u

    def sim_solve_batch(self, f, flags=0, rtol=1e-8, maxit=100000):
        """Simulator::solve for several load vectors (rows of f, dim*nDoF each) on one constrained system: returns (u [nrhs, nNode, dim],
        [info per load]). Positive definite systems go through the batches of mfh_solve_batch (the multigrid V-cycle's coarse levels serve the
        whole batch), systems with constraint rows are solved one load after the other."""
        f = as_f64(f)
        n = self.bs * self.n_dof
        nrhs = f.size // n
        assert f.size == nrhs * n
        u = np.empty((nrhs, self.n_node, self.bs))
        infos = (L.SolveInfo * nrhs)()
        st = self.lib.mfh_sim_solve_batch(self.h, nrhs, ptr(f), int(flags), ptr(u), float(rtol), int(maxit), infos)
        self.last_infos = [i.as_dict() for i in infos]
        self.last_info = self.last_infos[-1]
        self._ck(st)
        return u, self.last_infos

    def solve_cell_problems(self, cstrains, flags=0, rtol=1e-8, maxit=100000):
        """w[k] = Simulator::solve(constantStrainLoad(cstrains[k])) for every row of cstrains (flattened, tensor shear) on the constrained system
        of the context (mfh_solve_cell_problems): returns (w [nStrains, nNode, dim], [info per strain])."""
        cs = as_f64(cstrains)
        fl = self.dim * (self.dim + 1) // 2
        ns = cs.size // fl
        assert cs.size == ns * fl
        w = np.empty((ns, self.n_node, self.bs))
        infos = (L.SolveInfo * ns)()
        st = self.lib.mfh_solve_cell_problems(self.h, ns, ptr(cs), int(flags), ptr(w), float(rtol), int(maxit), infos)
        self.last_infos = [i.as_dict() for i in infos]
        self.last_info = self.last_infos[-1]
        self._ck(st)
        return w, self.last_infos

    def average_strain(self, u_nodes):
        u = as_f64(u_nodes)
        out = np.empty((self.n_elem, flat_len(self.dim)))
        self._ck(self.lib.mfh_average_strain(self.h, ptr(u), ptr(out)))
        return out

    def strain_field(self, u_nodes, stress=False):
        """per-element strain (stress) interpolant values: [nElem, 1 | dim+1, flatLen]"""
        u = as_f64(u_nodes)
        out = np.empty((self.n_elem, 1 if self.deg == 1 else self.dim + 1, flat_len(self.dim)))
        self._ck(self.lib.mfh_strain_field(self.h, ptr(u), int(bool(stress)), ptr(out)))
        return out

    def boundary_strain_field(self, u_nodes, stress=False):
        """the parent element's strain (stress) interpolant at the corners of every boundary element: [nBE, 1 | dim, flatLen]"""
        u = as_f64(u_nodes)
        out = np.empty((self.n_bdry_elem, 1 if self.deg == 1 else self.dim, flat_len(self.dim)))
        self._ck(self.lib.mfh_boundary_strain_field(self.h, ptr(u), int(bool(stress)), ptr(out)))
        return out

    def average_stress(self, u_nodes):
        u = as_f64(u_nodes)
        out = np.empty((self.n_elem, flat_len(self.dim)))
        self._ck(self.lib.mfh_average_stress(self.h, ptr(u), ptr(out)))
        return out

    def integrated_stress(self, u_nodes, cstrain_flat=None):
        """sum_e vol_e C_e : (average strain_e(u) + cstrain) reduced on the device (flatLen values): the element loop of
        homogenizedElasticityTensor (PeriodicHomogenization.hh:72-100) without a per-element field on the host."""
        u = as_f64(u_nodes)
        cs = None if cstrain_flat is None else as_f64(cstrain_flat)
        out = np.empty(flat_len(self.dim))
        self._ck(self.lib.mfh_integrated_stress(self.h, ptr(u), None if cs is None else ptr(cs), ptr(out)))
        return out

    # ---------------------------------------------------------------- discrete shape derivatives (forward mode)
    def _delta_p(self, delta_p):
        dp = as_f64(delta_p)
        if dp.size != self.n_vert * self.dim:
            raise ValueError("deltaP must be a per-vertex field [nVert x dim]")
        return dp

    def apply_delta_K(self, u_nodes, delta_p):
        """(delta K) u for a per-node field u under the vertex perturbation delta_p; per-DoF result."""
        u, dp = as_f64(u_nodes), self._delta_p(delta_p)
        out = np.empty((self.n_dof, self.dim))
        self._ck(self.lib.mfh_apply_delta_K(self.h, ptr(u), ptr(dp), ptr(out)))
        return out

    def delta_constant_strain_load(self, cstrain_flat, delta_p):
        e, dp = as_f64(cstrain_flat), self._delta_p(delta_p)
        out = np.empty((self.n_dof, self.dim))
        self._ck(self.lib.mfh_delta_constant_strain_load(self.h, ptr(e), ptr(dp), ptr(out)))
        return out

    def delta_average_strain(self, u_nodes, delta_u, delta_p, stress=False):
        u, du, dp = as_f64(u_nodes), as_f64(delta_u), self._delta_p(delta_p)
        out = np.empty((self.n_elem, flat_len(self.dim)))
        self._ck(self.lib.mfh_delta_average_strain(self.h, ptr(u), ptr(du), ptr(dp), int(bool(stress)), ptr(out)))
        return out

    def mutual_energies(self, w, delta_p=None):
        """flatLen x flatLen matrix of sum_e int (e^ij + eps(w^ij)) : C : (e^kl + eps(w^kl)) dV, or its discrete shape
        derivative under delta_p. w: flatLen per-node fields."""
        fl = flat_len(self.dim)
        wa = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.float64).reshape(self.n_node, self.dim) for x in w]))
        if wa.shape[0] != fl:
            raise ValueError("need one fluctuation displacement per canonical strain")
        dp = None if delta_p is None else self._delta_p(delta_p)
        out = np.empty((fl, fl))
        self._ck(self.lib.mfh_mutual_energies(self.h, ptr(wa), ptr(dp), ptr(out)))
        return out

    def mutual_energy_differential(self, w):
        """d(mutual energies)/d(vertex positions): [nPairs, nVert, dim], pairs = upper triangle ij <= kl row-major."""
        fl = flat_len(self.dim)
        wa = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.float64).reshape(self.n_node, self.dim) for x in w]))
        if wa.shape[0] != fl:
            raise ValueError("need one fluctuation displacement per canonical strain")
        out = np.empty((fl * (fl + 1) // 2, self.n_vert, self.dim))
        self._ck(self.lib.mfh_mutual_energy_differential(self.h, ptr(wa), ptr(out)))
        return out

    # ---------------------------------------------------------------- device pointers (torch interop)
    def stream(self):
        return self.lib.mfh_stream(self.h)

    def set_stream(self, hip_stream):
        """Adopt a caller-owned hipStream_t (int handle, e.g. torch.cuda.current_stream().cuda_stream)."""
        self._ck(self.lib.mfh_set_stream(self.h, C.c_void_p(hip_stream)))

    def dev_spmv(self, x_ptr, y_ptr):
        self._ck(self.lib.mfh_dev_spmv(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def dev_precond(self, r_ptr, z_ptr):
        self._ck(self.lib.mfh_dev_precond(self.h, C.c_void_p(r_ptr), C.c_void_p(z_ptr)))

    def tl_partitioned_begin(self, n_agg, agg_of_node, rel_pos, ac_ptr):
        """Caller-supplied (global) aggregates: this context's Galerkin contribution into the device
        buffer at ac_ptr ((n_agg*modes)^2 doubles)."""
        agg = np.ascontiguousarray(agg_of_node, dtype=np.int32)
        rp = np.ascontiguousarray(rel_pos, dtype=np.float64)
        n_local = self.matrix_info()[1]
        if agg.shape != (n_local,) or rp.shape != (n_local, 3):
            raise ValueError("agg_of_node / rel_pos must cover every local node")
        self._ck(self.lib.mfh_tl_partitioned_begin(self.h, int(n_agg), agg.ctypes.data_as(C.c_void_p),
                                                   rp.ctypes.data_as(C.c_void_p), C.c_void_p(ac_ptr)))

    def tl_partitioned_finish(self, ac_ptr):
        self._ck(self.lib.mfh_tl_partitioned_finish(self.h, C.c_void_p(ac_ptr)))

    def dev_tl_restrict(self, r_ptr, rc_ptr):
        self._ck(self.lib.mfh_dev_tl_restrict(self.h, C.c_void_p(r_ptr), C.c_void_p(rc_ptr)))

    def dev_tl_apply(self, r_ptr, rc_ptr, z_ptr):
        self._ck(self.lib.mfh_dev_tl_apply(self.h, C.c_void_p(r_ptr), C.c_void_p(rc_ptr), C.c_void_p(z_ptr)))

    def dev_pcg_update_xr(self, num_ptr, den_ptr, p_ptr, ap_ptr, x_ptr, r_ptr):
        self._ck(self.lib.mfh_dev_pcg_update_xr(self.h, *[C.c_void_p(q) for q in (num_ptr, den_ptr, p_ptr, ap_ptr, x_ptr, r_ptr)]))

    def dev_pcg_direction(self, num_ptr, den_ptr, z_ptr, p_ptr):
        self._ck(self.lib.mfh_dev_pcg_direction(self.h, *[C.c_void_p(q) for q in (num_ptr, den_ptr, z_ptr, p_ptr)]))

    def dev_dots(self, r_ptr, z_ptr, out_ptr):
        self._ck(self.lib.mfh_dev_dots(self.h, C.c_void_p(r_ptr), C.c_void_p(z_ptr), C.c_void_p(out_ptr)))

    def dev_mask_fixed(self, r_ptr):
        self._ck(self.lib.mfh_dev_mask_fixed(self.h, C.c_void_p(r_ptr)))

    def dev_set_fixed_values(self, u_ptr):
        self._ck(self.lib.mfh_dev_set_fixed_values(self.h, C.c_void_p(u_ptr)))

    # ---- row-partitioned solve through an mfh_comm
    def dist_setup(self, comm, peers, send_ptr, send_nodes, recv_ptr):
        peers, send_ptr, send_nodes, recv_ptr = as_i32(peers), as_i64(send_ptr), as_i32(send_nodes), as_i64(recv_ptr)
        self._ck(self.lib.mfh_dist_setup(self.h, comm.h, len(peers), ptr(peers), ptr(send_ptr), ptr(send_nodes), ptr(recv_ptr)))
        self._comm = comm          # keep the communicator (and its callbacks) alive as long as the context uses it

    def dist_two_level(self, n_agg, agg_of_node, rel_pos):
        agg_of_node, rel_pos = as_i32(agg_of_node), as_f64(rel_pos)
        self._ck(self.lib.mfh_dist_two_level(self.h, int(n_agg), ptr(agg_of_node), ptr(rel_pos)))

    def dist_solve(self, f_owned, rtol=1e-8, maxit=100000):
        """f_owned: [nrhs, dim * nOwned] (or flat for one right-hand side); returns (u_owned, [info per rhs])."""
        f = as_f64(f_owned)
        self.symbolic(False)                       # (the row count comes from the pattern; a no-op once it exists)
        nr, _, _ = self.matrix_info()
        n = self.bs * nr
        nrhs = f.size // n
        assert f.size == nrhs * n
        u = np.empty((nrhs, n))
        infos = (L.SolveInfo * nrhs)()
        st = self.lib.mfh_dist_solve(self.h, nrhs, ptr(f), ptr(u), float(rtol), int(maxit), infos)
        self.last_infos = [i.as_dict() for i in infos]
        self.last_info = self.last_infos[-1]
        self._ck(st)
        return u, self.last_infos

    def dist_apply_K(self, u_owned):
        u = as_f64(u_owned)
        out = np.empty_like(u)
        self._ck(self.lib.mfh_dist_apply_K(self.h, ptr(u), ptr(out)))
        return out

    def dist_stats(self):
        st = L.DistStats()
        self._ck(self.lib.mfh_dist_get_stats(self.h, C.byref(st)))
        return st.as_dict()

    def dev_memcpy(self, dst, src, nbytes, kind, stream=None):
        self._ck(self.lib.mfh_dev_memcpy(self.h, dst, src, int(nbytes), int(kind), stream))

    def dev_sync(self):
        self._ck(self.lib.mfh_dev_sync(self.h))

    # ---------------------------------------------------------------- measurement
    def timing(self):
        t = L.Timing()
        self._ck(self.lib.mfh_get_timing(self.h, C.byref(t)))
        return t.as_dict()

    def time_assembly_kernel(self, mode=L.ASSEMBLE_GATHER, reps=5):
        v = C.c_double()
        self._ck(self.lib.mfh_time_assembly_kernel(self.h, int(mode), int(reps), C.byref(v)))
        return v.value

    def time_spmv_kernel(self, reps=10):
        v = C.c_double()
        self._ck(self.lib.mfh_time_spmv_kernel(self.h, int(reps), C.byref(v)))
        return v.value

    def set_option(self, key, value):
        self._ck(self.lib.mfh_set_option(self.h, key.encode(), float(value)))


def device_cache_trim():
    """Return every cached device block of this process to the driver (mfh_device_cache_trim)."""
    L.load().mfh_device_cache_trim()


def device_cache_stats(device=0):
    lib = L.load()
    v = [C.c_int64() for _ in range(5)]
    lib.mfh_device_cache_stats(int(device), *[C.byref(x) for x in v])
    return dict(zip(("cached_bytes", "blocks", "hits", "misses", "flushes"), [x.value for x in v]))


def device_reserve(nbytes, device=0, wait=False):
    """One free segment of `nbytes` for the library's device arena, taken from the driver now (mfh_device_reserve) -- on a thread of its own
    unless `wait`: call it before reading / generating the mesh."""
    st = L.load().mfh_device_reserve(int(device), int(nbytes), 0 if wait else 1)
    if st != L.OK:
        raise L.MeshFEMHipError(st, "mfh_device_reserve failed")


def device_reserve_for(dim, deg, n_elem, device=0, wait=False):
    """mfh_device_reserve_for: the reservation for a context on a mesh of n_elem simplices of that kind -- the value array of K in a segment of
    its own, the rest in another. Returns the estimated total (bytes)."""
    lib = L.load()
    st = lib.mfh_device_reserve_for(int(device), int(dim), int(deg), int(n_elem), 0 if wait else 1)
    if st != L.OK:
        raise L.MeshFEMHipError(st, "mfh_device_reserve_for failed")
    return context_bytes_estimate(dim, deg, n_elem)[0]


def context_bytes_estimate(dim, deg, n_elem):
    """(total bytes, bytes of K's value array) a context on such a mesh holds at its peak (mfh_context_bytes_estimate)."""
    t, v = C.c_int64(), C.c_int64()
    st = L.load().mfh_context_bytes_estimate(int(dim), int(deg), int(n_elem), C.byref(t), C.byref(v))
    if st != L.OK:
        raise L.MeshFEMHipError(st, "mfh_context_bytes_estimate failed")
    return t.value, v.value


def device_arena_stats(device=0):
    """State of the library's device arena (mfh_device_arena_stats): bytes held / live / live high-water mark, segments, free chunks, ..."""
    lib = L.load()
    v = (C.c_int64 * 8)()
    lib.mfh_device_arena_stats(int(device), v)
    return dict(zip(("held_bytes", "live_bytes", "live_high_water_bytes", "segments", "free_chunks", "returned_to_driver_bytes",
                     "quarantined_bytes", "free_bound_bytes"), list(v)))
