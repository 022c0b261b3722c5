"""Two ranks sharing ONE MI355X (gloo rendezvous, buffers staged through the host): the row-partitioned HIP
path -- local assembly of owned rows, halo exchange, distributed PCG with block-Jacobi and with the
two-level preconditioner on global aggregates -- against the single-context solve of the same problem.
The 8-GPU box runs the same code with backend nccl (= RCCL); this test covers everything but the transport."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ret):
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        lm = D.slab_local_mesh(n, rank, world, 2)
        c = M.Context(0)
        c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        c.assemble()
        ops = D.HipLocalOps(c, 3)
        halo = D.HaloExchange(lm, rank, world, dev)
        owned_fixed = np.flatnonzero(lm.lattice[:lm.n_owned, 0] == 0)
        ov = (3 * owned_fixed[:, None] + np.arange(3)[None, :]).ravel()
        ev, evals = D.extend_fixed_to_halo(halo, lm.n_local, 3, ov, None, dev)
        all_fixed = np.flatnonzero(lm.lattice[:, 0] == 0)
        assert np.array_equal(ev, (3 * all_fixed[:, None] + np.arange(3)[None, :]).ravel())
        c.fix_variables(ev, evals)
        f = torch.as_tensor(D.slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel(), device=dev)
        u1, i1 = D.distributed_pcg(ops, halo, f, rtol=1e-10, maxit=20000)
        pre = D.DistributedTwoLevel(ops, halo, lm.node_pos, lm.n_owned, 16 * world)
        u2, i2 = D.distributed_pcg(ops, halo, f, rtol=1e-10, maxit=20000, precond=pre)
        ret[rank] = dict(keys=lm.keys[:lm.n_owned].copy(), u_bj=u1.cpu().numpy().reshape(-1, 3), u_tl=u2.cpu().numpy().reshape(-1, 3),
                         it_bj=i1["iterations"], it_tl=i2["iterations"], conv=(i1["converged"], i2["converged"]), n_agg=pre.n_agg)
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_partitioned_solve_matches_single_context():
    import torch.multiprocessing as mp
    import meshfem_amd as M
    from meshfem_amd import grid
    world, n = 2, 6
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    # the same problem in one context
    V, T = grid.grid_tet_mesh(n, n, n * world, [0, 0, 0], [1, 1, world])
    sim = M.Simulator(T, V, 2)
    sim.setIsotropicMaterial(200.0, 0.35)
    sim.ctx.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9 + world], [0, 0, 0])
    sim.ctx.bc_neumann_box([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9 + world], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    sim.rtol = 1e-10
    u_ref = sim.solve()
    lat = np.rint(sim.nodes() * 4 * n).astype(np.int64)
    keys = (lat[:, 0] * (4 * n + 1) + lat[:, 1]) * (4 * n * world + 1) + lat[:, 2]
    order = np.argsort(keys)
    seen = 0
    for r in range(world):
        d = ret[r]
        assert all(d["conv"])
        idx = order[np.searchsorted(keys[order], d["keys"])]
        assert np.array_equal(keys[idx], d["keys"])
        for name in ("u_bj", "u_tl"):
            # tolerance: both sides are PCG solutions to rtol 1e-10 of the same SPD system
            assert np.linalg.norm(d[name] - u_ref[idx]) / np.linalg.norm(u_ref) < 1e-7, (r, name)
        seen += len(idx)
    assert seen == len(keys)                                   # every node owned exactly once
    assert ret[0]["it_bj"] == ret[1]["it_bj"] and ret[0]["it_tl"] == ret[1]["it_tl"]
    assert ret[0]["it_tl"] < 0.6 * ret[0]["it_bj"], (ret[0]["it_tl"], ret[0]["it_bj"])
