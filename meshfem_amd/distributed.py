"""Row-partitioned multi-GPU assembly + PCG.

One process per GPU. Nodes (block rows of K) are partitioned; every rank keeps the elements
incident to its owned nodes, numbers its nodes owned-first / halo-last and assembles the
`nOwned x nLocal` block rows it owns with NO communication (owner computes). The PCG needs, per
iteration, one halo exchange of the search direction (point-to-point with the neighbouring ranks)
and two small all-reduces of the dot products -- `torch.distributed` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests). The local kernels are the library's device-pointer
entry points (`HipLocalOps`); the tests substitute their own local operator, the product never
falls back to one.

The reference has no distributed code: this module has no counterpart to cite
beyond the serial path it parallelises (LinearElasticity.hh:479-487, SparseMatrices.hh:2515-2606).
"""
import time

import numpy as np
import torch
import torch.distributed as dist

TET_FACES = ((1, 3, 2), (0, 2, 3), (0, 3, 1), (0, 1, 2))          # TetMesh.hh:221-226
EDGE_OF = {frozenset(p): 4 + e for e, p in enumerate(((0, 1), (1, 2), (2, 0), (0, 3), (2, 3), (1, 3)))}  # Simplex.hh:43-44


# ------------------------------------------------------------------------------------------------
# partitioning (pure numpy; identical on GPU and in the gloo tests)
# ------------------------------------------------------------------------------------------------
class LocalMesh:
    """A rank's share: elements with >= 1 owned node, nodes renumbered owned-first (sorted by global
    key), then halo nodes grouped by owner rank (sorted by key inside a group)."""

    def __init__(self, elem_nodes, node_pos, keys, n_owned, halo_ranges, kept_elems):
        self.elem_nodes = elem_nodes          # [nElemLocal, npe] local node ids
        self.node_pos = node_pos              # [nLocal, dim]
        self.keys = keys                      # [nLocal] global int64 key of every local node
        self.n_owned = n_owned
        self.halo_ranges = halo_ranges        # {owner rank: (start, end)} in local node numbering
        self.kept_elems = kept_elems          # indices into the candidate element array
        self.n_local = len(keys)


def partition(elem_nodes, node_pos, node_keys, node_owner, rank):
    """elem_nodes: candidate elements (any superset of the elements touching this rank's nodes) in
    some local numbering; node_keys: globally unique int64 per node; node_owner: rank per node."""
    elem_nodes = np.asarray(elem_nodes)
    owned_mask = node_owner == rank
    keep = owned_mask[elem_nodes].any(axis=1)
    en = elem_nodes[keep]
    used = np.unique(en)
    owner_u, keys_u = node_owner[used], node_keys[used]
    # owned first; halo grouped by owner; key-sorted inside each group
    group = np.where(owner_u == rank, -1, owner_u)
    order = np.lexsort((keys_u, group))
    new_nodes = used[order]
    remap = np.full(len(node_keys), -1, dtype=np.int64)
    remap[new_nodes] = np.arange(len(new_nodes))
    n_owned = int((group == -1).sum())
    halo_ranges = {}
    g_sorted = group[order]
    for q in np.unique(g_sorted[n_owned:]):
        idx = np.flatnonzero(g_sorted == q)
        halo_ranges[int(q)] = (int(idx[0]), int(idx[-1]) + 1)
    return LocalMesh(remap[en].astype(np.int32), node_pos[new_nodes], node_keys[new_nodes], n_owned, halo_ranges,
                     np.flatnonzero(keep))


def _needs_staging(device, group=None):
    """gloo moves host memory: device tensors are staged through the host (2-ranks-on-1-GPU tests).
    With nccl (= RCCL) buffers go GPU to GPU over xGMI."""
    return torch.device(device).type == "cuda" and dist.is_initialized() and dist.get_backend(group) == "gloo"


def rcb_node_owner(node_pos, world):
    """Recursive coordinate bisection of the nodes ("general meshes: RCB"): split the longest axis of
    the bounding box at the weighted median, recursively, into `world` parts of (almost) equal node counts. Deterministic
    (stable sort on (coordinate, node index)), so every rank computes the same owners from the same global mesh."""
    pos = np.asarray(node_pos, dtype=np.float64)
    owner = np.zeros(len(pos), dtype=np.int64)

    def split(idx, first, parts):
        if parts == 1 or len(idx) == 0:
            owner[idx] = first
            return
        left_parts = parts // 2
        ext = pos[idx].max(axis=0) - pos[idx].min(axis=0)
        ax = int(np.argmax(ext))
        order = idx[np.lexsort((idx, pos[idx, ax]))]
        cut = (len(idx) * left_parts) // parts
        split(order[:cut], first, left_parts)
        split(order[cut:], first + left_parts, parts - left_parts)

    split(np.arange(len(pos)), 0, int(world))
    return owner


def distribute_mesh(vertices, elements, degree, rank, world):
    """A rank's LocalMesh of a GLOBAL mesh that every rank holds (general unstructured meshes of moderate size; the
    synthetic slabs below never materialise the global mesh): FEM node numbering on the host, RCB ownership of the
    nodes, then the generic `partition`. Keys are the global node ids."""
    from .core import Context
    h = Context(-1)                                        # host-only: edge-node numbering (FEMMesh.inl:22-36)
    h.mesh_build(np.asarray(elements), np.asarray(vertices, dtype=np.float64), degree)
    en, pos = h.elem_nodes().astype(np.int64), h.node_positions()
    bnodes = h.boundary_nodes()
    h.close()
    owner = rcb_node_owner(pos, world)
    lm = partition(en, pos, np.arange(len(pos), dtype=np.int64), owner, rank)
    lm.global_is_boundary = np.zeros(len(pos), dtype=bool)
    lm.global_is_boundary[bnodes] = True
    lm.n_global = len(pos)
    return lm


class HaloExchange:
    """Point-to-point exchange lists. Every rank asks the owners for its halo nodes by key."""

    def __init__(self, lm: LocalMesh, rank, world, device, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.n_owned, self.n_local = lm.n_owned, lm.n_local
        requests = [None] * world
        for q, (s, e) in lm.halo_ranges.items():
            requests[q] = lm.keys[s:e]
        gathered = [None] * world
        dist.all_gather_object(gathered, requests, group=group)     # setup only (python objects)
        owned_keys = lm.keys[:lm.n_owned]
        sorter = np.argsort(owned_keys)
        self.send_idx = {}
        for q in range(world):
            req = gathered[q][rank] if gathered[q] is not None else None
            if req is None or len(req) == 0:
                continue
            pos = np.searchsorted(owned_keys, req, sorter=sorter)
            idx = sorter[np.clip(pos, 0, len(sorter) - 1)]
            if not np.array_equal(owned_keys[idx], req):
                raise RuntimeError("halo request for a node this rank does not own")
            self.send_idx[q] = torch.as_tensor(idx, dtype=torch.long, device=device)
        self.recv_range = dict(lm.halo_ranges)
        self.stage = _needs_staging(device, group)

    def exchange(self, v, dim):
        """v: flat tensor of n_local*dim; fills the halo part from the owners."""
        if self.world == 1:
            return
        v2 = v.view(-1, dim)
        ops, keep, staged = [], [], []
        for q, idx in sorted(self.send_idx.items()):
            buf = v2[idx].contiguous()
            if self.stage:
                buf = buf.cpu()
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, q, group=self.group))
        for q, (s, e) in sorted(self.recv_range.items()):
            dst = v2[s:e]
            if self.stage:
                dst = torch.empty(dst.shape, dtype=v.dtype)
                staged.append((s, e, dst))
            ops.append(dist.P2POp(dist.irecv, dst, q, group=self.group))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        for s, e, t in staged:
            v2[s:e] = t.to(v.device)


# ------------------------------------------------------------------------------------------------
# local operators
# ------------------------------------------------------------------------------------------------
class HipLocalOps:
    """The rank's block rows on its GPU, through the C ABI's device-pointer entry points."""

    def __init__(self, ctx, dim):
        self.ctx, self.dim = ctx, dim
        nr, nc, _ = ctx.matrix_info()
        self.n_rows, self.n_cols = nr * dim, nc * dim
        self.device = torch.device("cuda", torch.cuda.current_device())
        # run the library's kernels on torch's current stream: ordered with torch ops and RCCL
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64, device=self.device)

    def spmv(self, x_local, out):
        self.ctx.dev_spmv(x_local.data_ptr(), out.data_ptr())

    def precond(self, r, out):
        self.ctx.dev_precond(r.data_ptr(), out.data_ptr())

    def mask_fixed(self, v):
        self.ctx.dev_mask_fixed(v.data_ptr())

    def set_fixed_values(self, u):
        self.ctx.dev_set_fixed_values(u.data_ptr())

    # fused vector updates; the scalars stay on the device (they come out of all-reduces)
    def update_xr(self, num, den, p, Ap, x, r):
        self.ctx.dev_pcg_update_xr(num.data_ptr(), den.data_ptr(), p.data_ptr(), Ap.data_ptr(), x.data_ptr(), r.data_ptr())

    def direction(self, num, den, z, p):
        self.ctx.dev_pcg_direction(num.data_ptr(), den.data_ptr(), z.data_ptr(), p.data_ptr())

    def dots(self, r, z, out):
        self.ctx.dev_dots(r.data_ptr(), z.data_ptr(), out.data_ptr())

    # two-level preconditioner building blocks (global aggregates; DistributedTwoLevel reduces over ranks)
    def tl_begin(self, n_agg, agg_of_node, rel_pos):
        m = n_agg * (6 if self.dim == 3 else 3)
        Ac = torch.empty((m, m), dtype=torch.float64, device=self.device)
        self.ctx.tl_partitioned_begin(n_agg, agg_of_node, rel_pos, Ac.data_ptr())
        return Ac

    def tl_finish(self, Ac):
        self.ctx.tl_partitioned_finish(Ac.data_ptr())

    def tl_restrict(self, r, rc):
        self.ctx.dev_tl_restrict(r.data_ptr(), rc.data_ptr())

    def tl_apply(self, r, rc, z):
        self.ctx.dev_tl_apply(r.data_ptr(), rc.data_ptr(), z.data_ptr())


# ------------------------------------------------------------------------------------------------
# distributed PCG (classic, two all-reduces per iteration)
# ------------------------------------------------------------------------------------------------
def _allreduce(vals, device, group, op=None):
    t = torch.stack(vals) if isinstance(vals, (list, tuple)) else vals
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        op = dist.ReduceOp.SUM if op is None else op
        if _needs_staging(t.device, group):
            h = t.cpu()
            dist.all_reduce(h, op=op, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=group)
    return t


def extend_fixed_to_halo(halo, n_local, dim, owned_vars, owned_vals, device):
    """Fixed variables are known to the owner of a node; the Galerkin coarse operator needs the mask
    on halo columns too. Returns (vars, vals) over all local variables (owned + halo)."""
    m = torch.zeros(n_local * dim, dtype=torch.float64, device=device)
    v = torch.zeros(n_local * dim, dtype=torch.float64, device=device)
    idx = torch.as_tensor(np.asarray(owned_vars, dtype=np.int64), device=device)
    m[idx] = 1.0
    v[idx] = torch.as_tensor(np.asarray(owned_vals, dtype=np.float64) if owned_vals is not None else np.zeros(len(idx)), device=device)
    halo.exchange(m, dim)
    halo.exchange(v, dim)
    sel = torch.nonzero(m > 0.5).ravel()
    return sel.cpu().numpy(), v[sel].cpu().numpy()


class DistributedTwoLevel:
    """M^-1 = D^-1 + Z (Z^T K Z)^-1 Z^T with GLOBAL aggregates: geometric bins over the global bounding
    box, six rigid-body modes per bin (three in 2D). Every rank contributes the Galerkin product of its
    own rows; one all-reduce of the (small, dense) coarse operator at setup, every rank inverts it
    redundantly; per application one extra all-reduce of the restricted residual (6 * bins doubles).
    The local kernels are `ops.tl_*` (HipLocalOps: the library's device kernels)."""

    def __init__(self, ops, halo, node_pos, n_owned, target_aggregates, group=None):
        self.ops, self.group = ops, group
        dim = ops.dim
        dev = ops.zeros(1).device
        pos = np.asarray(node_pos, dtype=np.float64)
        lo = _allreduce(torch.as_tensor(pos.min(axis=0), device=dev), dev, group, dist.ReduceOp.MIN).cpu().numpy()
        hi = _allreduce(torch.as_tensor(pos.max(axis=0), device=dev), dev, group, dist.ReduceOp.MAX).cpu().numpy()
        ext = np.maximum(hi - lo, 1e-300)
        H = (np.prod(ext) / max(1, target_aggregates)) ** (1.0 / dim)
        nb = np.maximum(1, np.rint(ext / H)).astype(np.int64)
        hb = ext / nb
        cell = np.minimum(np.floor((pos - lo) / hb).astype(np.int64), nb - 1)
        cell = np.maximum(cell, 0)
        agg = cell[:, 0]
        for a in range(1, dim):
            agg = agg * nb[a] + cell[:, a]
        # the owner's binning is authoritative (a node on a bin face must not depend on rounding)
        t = torch.zeros(len(pos), dtype=torch.float64, device=dev)
        t[:n_owned] = torch.as_tensor(agg[:n_owned].astype(np.float64), device=dev)
        halo.exchange(t, 1)
        agg = np.rint(t.cpu().numpy()).astype(np.int64)
        cell = np.zeros((len(pos), dim), dtype=np.int64)
        rem = agg.copy()
        for a in range(dim - 1, -1, -1):
            cell[:, a] = rem % nb[a]
            rem //= nb[a]
        centre = lo + (cell + 0.5) * hb
        rel = np.zeros((len(pos), 3))
        rel[:, :dim] = (pos - centre) / H
        self.n_agg = int(np.prod(nb))
        self.bins, self.H = nb, H
        t0 = time.perf_counter()
        Ac = ops.tl_begin(self.n_agg, agg.astype(np.int32), rel)
        _allreduce(Ac, dev, group)
        ops.tl_finish(Ac)
        del Ac
        self.m = self.n_agg * (6 if dim == 3 else 3)
        self.rc = ops.zeros(self.m)
        self.setup_s = time.perf_counter() - t0

    def __call__(self, r, z):
        self.ops.tl_restrict(r, self.rc)
        _allreduce(self.rc, self.rc.device, self.group)
        self.ops.tl_apply(r, self.rc, z)


def distributed_pcg(ops, halo, f_owned, rtol=1e-8, maxit=20000, group=None, check_every=25, precond=None):
    """Solve K u = f on the free variables; f_owned: this rank's dim*nOwned right-hand side.
    Returns (u_owned, info). All vectors are torch tensors on ops' device."""
    dim = ops.dim
    nr, nc = ops.n_rows, ops.n_cols
    dev = f_owned.device
    precond = ops.precond if precond is None else precond
    # b = P (f - K ubar): ubar = fixed values on every rank's owned nodes, halo part by exchange
    u0 = ops.zeros(nc)
    ops.set_fixed_values(u0)
    halo.exchange(u0, dim)
    Ku0 = ops.zeros(nr)
    ops.spmv(u0, Ku0)
    b = f_owned - Ku0
    ops.mask_fixed(b)
    x = ops.zeros(nr)
    r = b.clone()
    z = ops.zeros(nr)
    precond(r, z)
    p = ops.zeros(nc)
    p[:nr] = z
    Ap = ops.zeros(nr)
    # local kernels: the ops' fused ones (HipLocalOps) or plain torch (the CPU tests' operator)
    def dots(rv, zv):
        out = torch.empty(2, dtype=torch.float64, device=dev)
        if hasattr(ops, "dots"):
            ops.dots(rv, zv, out)
        else:
            out[0], out[1] = torch.dot(rv, zv), torch.dot(rv, rv)
        return out

    def update_xr(num, den):
        if hasattr(ops, "update_xr"):
            ops.update_xr(num, den, p, Ap, x, r)
        else:
            a = num / den
            x.add_(p[:nr] * a)
            r.sub_(Ap * a)

    def direction(num, den):
        if hasattr(ops, "direction"):
            ops.direction(num, den, z, p)
        else:
            p[:nr].mul_(num / den).add_(z)

    red = _allreduce(dots(r, z), dev, group)
    rz, bb = red[0:1], red[1].item()
    stop = rtol * rtol * bb
    info = dict(iterations=0, converged=bb == 0.0, rel_residual=0.0)
    it = 0
    rr = bb
    t0 = time.perf_counter()
    hist = []
    while it < maxit and not info["converged"]:
        halo.exchange(p, dim)
        ops.spmv(p, Ap)
        ops.mask_fixed(Ap)
        pAp = _allreduce(torch.dot(p[:nr], Ap).reshape(1), dev, group)
        update_xr(rz, pAp)                                   # x += (rz/pAp) p ; r -= (rz/pAp) Ap
        precond(r, z)
        red = _allreduce(dots(r, z), dev, group)
        direction(red[0:1], rz)                              # p = z + (rz_new/rz) p
        rz = red[0:1]
        hist.append(red[1])
        it += 1
        if it % check_every == 0 or it == maxit:
            h = torch.stack(hist).cpu().numpy()        # one host sync per check_every iterations
            hit = np.flatnonzero(h <= stop)
            if len(hit):
                info["converged"] = True
                rr = float(h[hit[0]])
                info["iterations"] = it - len(h) + int(hit[0]) + 1
            else:
                rr = float(h[-1])
                info["iterations"] = it
            hist = []
    if dev.type == "cuda":
        torch.cuda.synchronize()
    info["solve_s"] = time.perf_counter() - t0
    info["loop_iterations"] = it
    info["rel_residual"] = float(np.sqrt(rr / bb)) if bb > 0 else 0.0
    u = x.clone()
    full = ops.zeros(nc)
    full[:nr] = u
    ops.set_fixed_values(full)
    return full[:nr].clone(), info


# ------------------------------------------------------------------------------------------------
# synthetic z-slab meshes (weak scaling): the reference generator restricted to a z-range
# ------------------------------------------------------------------------------------------------
def slab_local_mesh(n, rank, world, deg=2, layers=None, device=-1):
    """Grid n x n x (layers*world) of cubic cells of size 1/n; rank owns hex layers [layers*rank, layers*(rank+1))
    (layers defaults to n: a cube per rank). Returns the LocalMesh plus integer lattice coordinates (units of
    1/(4n)) of every local node."""
    from . import grid
    from .core import Context
    layers = n if layers is None else int(layers)
    z0, z1 = layers * rank, layers * (rank + 1)
    top = 1 if rank < world - 1 else 0                 # one halo hex layer above the owned interface plane
    V, H = grid.gen_grid_3d(n, n, z1 - z0 + top, z0=z0)
    V, T = grid.hex_tet_subdiv(V, H)
    h = Context(device)                                # P2 node numbering (FEMMesh.inl:22-36): host (-1) or device radix sorts
    h.mesh_build(T, V, deg)
    en, pos = h.elem_nodes().astype(np.int64), h.node_positions()
    h.close()
    lat = np.rint(pos * 4).astype(np.int64)            # vertices/centres/midpoints live on the quarter lattice
    M = 4 * n + 1
    keys = (lat[:, 0] * M + lat[:, 1]) * (4 * layers * world + 1) + lat[:, 2]
    # rank r owns lattice z in (4 L r, 4 L (r+1)]; the global bottom plane belongs to rank 0
    owner = np.clip((lat[:, 2] + 4 * layers - 1) // (4 * layers) - 1, 0, world - 1)
    lm = partition(en, pos / n, keys, owner, rank)
    lm.lattice = np.rint(lm.node_pos * 4 * n).astype(np.int64)
    return lm


def slab_traction_load(lm, n, traction):
    """neumannLoad (LinearElasticity.hh:703-717) for a constant traction on the plane x = 1, P2 tets:
    each boundary face contributes t*A/3 to its three edge nodes and nothing to its vertices
    (Functions.hh:263-274). Every rank has all faces incident to its owned nodes."""
    en, pos = lm.elem_nodes, lm.node_pos
    lat_x = lm.lattice[:, 0]
    load = np.zeros((lm.n_owned, 3))
    on = lat_x[en[:, :4]] == 4 * n
    for f, (a, b, c) in enumerate(TET_FACES):
        sel = np.flatnonzero(on[:, a] & on[:, b] & on[:, c])
        if not len(sel):
            continue
        pa, pb, pc = pos[en[sel, a]], pos[en[sel, b]], pos[en[sel, c]]
        area = 0.5 * np.linalg.norm(np.cross(pb - pa, pc - pa), axis=1)
        for u, v in ((a, b), (b, c), (c, a)):
            node = en[sel, EDGE_OF[frozenset((u, v))]]
            own = node < lm.n_owned
            np.add.at(load, node[own], (area[own] / 3.0)[:, None] * np.asarray(traction)[None, :])
    return load


def _bench_slabs_solve(args, rank, world, dev, c, lm, n, deg, out):
        ops = HipLocalOps(c, 3)
        halo = HaloExchange(lm, rank, world, dev)
        # u = 0 on x = 0: the mask covers the halo nodes too (needed by the Galerkin coarse operator)
        fixed_nodes = np.flatnonzero(lm.lattice[:, 0] == 0)
        c.fix_variables((3 * fixed_nodes[:, None] + np.arange(3)[None, :]).ravel())
        f = torch.as_tensor(slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel(), device=dev)
        n_coarse = getattr(args, "coarse_aggregates", 0)
        if n_coarse == 0:
            n_coarse = min(1000 * world, 2048)
        pre, pre_desc, tl_info = None, "3x3 block-Jacobi", None
        maxit = args.maxit
        if n_coarse > 0:
            pre = DistributedTwoLevel(ops, halo, lm.node_pos, lm.n_owned, n_coarse)
            pre_desc = "two-level: 3x3 block-Jacobi + rigid-body modes of %d global bins" % pre.n_agg
            tl_info = dict(bins=[int(b) for b in pre.bins], coarse_dim=int(pre.m), setup_s=pre.setup_s)
        elif world > 1:
            # the bar gets longer with N (weak scaling), so block-Jacobi alone needs O(N) more iterations:
            # measure the per-iteration rate on a bounded number of iterations
            maxit = min(args.maxit, 3000)
        dist.barrier()
        u, info = distributed_pcg(ops, halo, f, rtol=args.rtol, maxit=maxit, precond=pre)
        nd = torch.tensor([3.0 * lm.n_owned], dtype=torch.float64, device=dev)
        dist.all_reduce(nd)
        ts = torch.tensor([info["solve_s"]], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        # true residual of the returned displacement, over all ranks
        full = ops.zeros(ops.n_cols)
        full[:ops.n_rows] = u
        halo.exchange(full, 3)
        Ku = ops.zeros(ops.n_rows)
        ops.spmv(full, Ku)
        res = f - Ku
        ops.mask_fixed(res)
        fm = f.clone()
        ops.mask_fixed(fm)
        nrm = torch.stack([torch.dot(res, res), torch.dot(fm, fm), u.abs().max()])
        dist.all_reduce(nrm[:2])
        dist.all_reduce(nrm[2:], op=dist.ReduceOp.MAX)
        out["pcg"] = dict(iterations=info["iterations"], converged=bool(info["converged"]), rtol=args.rtol,
                          rel_residual=info["rel_residual"], true_rel_residual=float(torch.sqrt(nrm[0] / nrm[1]).item()),
                          max_abs_u=float(nrm[2].item()), dof=int(nd.item()), solve_s=ts.item(),
                          dof_per_s=nd.item() * info["loop_iterations"] / ts.item(),
                          ms_per_iteration=ts.item() / max(1, info["loop_iterations"]) * 1e3,
                          maxit=maxit, preconditioner=pre_desc, two_level=tl_info,
                          operator="matrix-free (k_mf_cluster + k_mf_rows)" if deg == 2 else "assembled block-CSR (k_spmv)",
                          comm="halo P2P + %d all-reduce / iteration (torch.distributed nccl=RCCL)" % (3 if pre else 2))


def bench_slabs(args, rank, world, local_rank):
    """bench.py --gpus N>1: weak scaling over z-slabs; returns the JSON dict on every rank."""
    import meshfem_amd as M
    dev = torch.device("cuda", local_rank)
    deg = args.deg
    # weak scaling towards BASELINE configs[4] (a ~40 M-tet CUBE in 8 z-slabs): the global grid is
    # n x n x (layers * world) with n ~ grid * world^(1/3), and every rank keeps ~24 grid^3 elements
    n = int(round(args.grid * world ** (1.0 / 3.0)))
    layers = max(1, int(round(args.grid ** 3 / float(n * n))))
    t0 = time.time()
    lm = slab_local_mesh(n, rank, world, deg, layers, device=local_rank)
    t_mesh = time.time() - t0
    c = M.Context(local_rank)
    c.mesh_set(3, deg, lm.elem_nodes, lm.node_pos, lm.n_owned)
    c.material_isotropic(200.0, 0.35)
    t0 = time.time(); c.symbolic(False); t_sym = time.time() - t0
    c.set_option("reembed", 1)
    # elements are counted once globally: a rank "owns" the elements of its own hex layers
    n_elem_global = 24 * n * n * layers * world
    for _ in range(args.warmup):
        c.assemble()
    c.dev_sync(); torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        c.assemble()
    c.dev_sync(); torch.cuda.synchronize(); dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = dt.item()
    k_ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, max(3, args.steps))
    alg = 7736 if deg == 2 else 1328
    out = dict(metric="stiffness_assembly_elements_per_s", value=n_elem_global * args.steps / dt, unit="elements/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload="%d x %d x %d grid -> %d P%d tets, z-slabs of %d hex layers per GPU (configs[4] shape: 120^3 at 8 GPUs)"
                                    % (n, n, layers * world, n_elem_global, deg, layers), elements=n_elem_global,
                           local_elements=int(len(lm.elem_nodes)), local_nodes=int(lm.n_local), owned_nodes=int(lm.n_owned),
                           parallelism="row/element partition x%d, owner computes" % world),
               roofline=dict(bound="hbm", kernel="k_assemble_gather", achieved=alg * len(lm.elem_nodes) / k_ms / 1e6, peak=8000.0,
                             unit="GB/s", frac=alg * len(lm.elem_nodes) / k_ms / 1e6 / 8000.0, traffic=None, kernel_ms=k_ms,
                             note="rank 0's local launch (its elements incl. the halo layer)"),
               setup=dict(local_mesh_s=t_mesh, symbolic_s=t_sym))
    if not args.no_solve:
        # The assembly figures above are the headline metric; a failure of the solver leg (it is the only part that
        # depends on the interconnect) must not lose them: it is reported inside the JSON line instead.
        try:
            _bench_slabs_solve(args, rank, world, dev, c, lm, n, deg, out)
        except Exception as e:   # noqa: BLE001
            out["pcg"] = dict(error="%s: %s" % (type(e).__name__, e))
    return out
