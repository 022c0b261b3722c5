"""Randomised check of the general-mesh distributed path (distributed.scatter_mesh): see scripts/fuzz_scatter.py. run(seed) spawns 2-3 ranks
sharing the GPU and returns (ok, one-line report)."""
import os
import socket

import numpy as np


def load_of(pos):
    return np.stack([np.sin(7 * pos[:, 0] + 3 * pos[:, 1] + pos[:, 2]), np.cos(5 * pos[:, 1] - 2 * pos[:, 0]), -1.0 + 0 * pos[:, 0]], axis=1)


def worker(rank, world, port, seed, ret):
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D
    from scipy.spatial import cKDTree
    from fuzz_unstructured_util import random_mesh
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng, dim, deg, E, V, mat = random_mesh(seed)
        lm = D.scatter_mesh(V if rank == 0 else None, E if rank == 0 else None, deg, rank, world)
        ext = V.max(axis=0)
        c = M.Context(0)
        c.mesh_set(dim, deg, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.3)
        comm = D.make_comm(c, rank, world)
        solver = D.DistSolver(c, lm, rank, world, comm)
        fixed = np.flatnonzero(lm.node_pos[:, 0] <= 0.12 * ext[0])
        c.fix_variables((dim * fixed[:, None] + np.arange(dim)).ravel())
        f = load_of(np.pad(lm.node_pos[:lm.n_owned], ((0, 0), (0, 3 - dim))))[:, :dim].ravel().copy()
        out = {}
        for name, pre in (("mg", M.PRECOND_MULTIGRID), ("bj", M.PRECOND_BLOCK_JACOBI)):
            c.set_preconditioner(pre)
            u, infos = solver.solve(f, rtol=1e-10, maxit=200000)
            out[name] = (u[0].reshape(-1, dim).copy(), infos[0]["iterations"], bool(infos[0]["converged"]))
        pos_owned = lm.node_pos[:lm.n_owned].copy()
        comm.close(); c.close()
        # the whole mesh in one context (every rank does its own: the meshes are small)
        g = M.Context(0)
        g.mesh_build(E, V, deg); g.material_isotropic(200.0, 0.3)
        P = g.node_positions()
        gf = np.flatnonzero(P[:, 0] <= 0.12 * ext[0])
        g.fix_variables((dim * gf[:, None] + np.arange(dim)).ravel())
        fg = load_of(np.pad(P, ((0, 0), (0, 3 - dim))))[:, :dim]
        g.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
        ug = g.solve(fg.ravel(), rtol=1e-11, maxit=400000).reshape(-1, dim)
        g.close()
        d, idx = cKDTree(P).query(pos_owned)
        errs = {k: float(np.linalg.norm(out[k][0] - ug[idx]) / np.linalg.norm(ug)) for k in out}
        ret[rank] = dict(seed=seed, dim=dim, deg=deg, verts=len(V), elems=len(E), owned=int(lm.n_owned), match=float(d.max()), err=errs,
                         its={k: out[k][1] for k in out}, conv=all(out[k][2] for k in out), n_glob=len(P))
    finally:
        dist.destroy_process_group()


def run(seed):
    import torch.multiprocessing as mp
    world = 2 + seed % 2
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(world, port, seed, ret), nprocs=world, join=True)
    r = [ret[k] for k in range(world)]
    ok = all(x["conv"] and x["match"] < 1e-9 and max(x["err"].values()) < 1e-6 for x in r) and sum(x["owned"] for x in r) == r[0]["n_glob"]
    return ok, "seed %d world %d dim %d deg %d verts %d elems %d  err %s its %s owned %s" % (
        seed, world, r[0]["dim"], r[0]["deg"], r[0]["verts"], r[0]["elems"], {k: "%.1e" % max(x["err"][k] for x in r) for k in r[0]["err"]}, r[0]["its"], [x["owned"] for x in r])
