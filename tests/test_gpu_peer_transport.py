"""Peer transfers of the row-partitioned solve (mfh_comm_enable_peer, mfh_peer.hip): the halo exchange and the small all-reduces as
direct device-to-device writes through HIP IPC, no library call and no host code per iteration. Ranks sharing ONE MI355X (same-device
IPC handles): every solve is run twice on the same contexts -- through the callback transport (gloo, staged through the host) and
through the peer transfers -- and must agree: same iteration counts, displacements to 1e-12 (the all-reduce sums in rank order on
both transports here: gloo's sum of two or three terms is order-independent up to the last bit, hence a tolerance instead of
array_equal). The 8-GPU box runs the same code across devices over xGMI; bench.py selects it after mfh_comm_selftest."""
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


def _worker(rank, world, port, n, deg, ret):
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    os.environ.setdefault("MFH_PEER_TIMEOUT_S", "10")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        lm = D.slab_local_mesh(n, rank, world, deg)
        c = M.Context(0)
        c.mesh_set(3, deg, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        c.set_option("dist_profile", 1)
        comm = D.Comm.callbacks(c, rank, world)
        comm.selftest()
        solver = D.DistSolver(c, lm, rank, world, comm)
        all_fixed = np.flatnonzero(lm.lattice[:, 0] == 0)
        c.fix_variables((3 * all_fixed[:, None] + np.arange(3)[None, :]).ravel())
        f = D.slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel()

        def solves():
            out = {}
            c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
            for variant, name in ((1, "bj_cg"), (0, "bj_classic")):
                c.set_option("dist_pcg_variant", variant)
                u, i = solver.solve(f, rtol=1e-10, maxit=20000)
                out[name] = (u[0].copy(), i[0]["iterations"], i[0]["converged"], c.dist_stats())
            c.set_option("dist_pcg_variant", 1)
            solver.two_level(16 * world)
            u, i = solver.solve(f, rtol=1e-10, maxit=20000)
            out["two_level"] = (u[0].copy(), i[0]["iterations"], i[0]["converged"], c.dist_stats())
            c.set_preconditioner(M.PRECOND_MULTIGRID)
            u, i = solver.solve(f, rtol=1e-10, maxit=2000)
            out["multigrid"] = (u[0].copy(), i[0]["iterations"], i[0]["converged"], c.dist_stats())
            return out

        a = solves()
        pre = [k for k, v in a.items() if not (v[3]["transport"] == 3 and v[3]["peer_enabled"] == 0)]
        comm.enable_peer()
        assert "HIP IPC" in comm.describe()
        comm.selftest()                                    # six rounds of ring shift + short and long all-reduce through the peer path
        solver2 = D.DistSolver(c, lm, rank, world, comm)   # mfh_dist_setup again: sizes the staging for this mesh's halos (collective)
        solver = solver2
        b = solves()
        # Collect, do not assert here: a rank that leaves the collective sequence on its own would leave the others waiting.
        # Two ranks: a sum of two terms has one value whatever the order, the two transports give the same bits and the same iteration
        # counts. Three ranks: gloo's ring may add in another order than rank order, the last bit of a dot product differs now and then, and
        # two CG runs that differ in rounding agree to about the tolerance they were stopped at (1e-10), not to 1e-12.
        problems = [("before enable", k) for k in pre]
        tol, slack = (1e-12, 0) if world == 2 else (5e-8, 3)
        for name in a:
            ua, ita, ca, _ = a[name]
            ub, itb, cb, st = b[name]
            if not (ca and cb): problems.append((name, "not converged"))
            if not (st["transport"] == 2 and st["peer_enabled"] == 1 and st["fallback_exchanges"] == 0): problems.append((name, "transport", st))
            if not (st["profiled_applications"] > 0 and st["exchange_ms"] > 0 and st["operator_ms"] >= st["interior_ms"] > 0): problems.append((name, "profile", st))
            if st["halo_bytes_per_exchange"] != 24 * (st["halo_nodes_sent"] + st["halo_nodes_received"]): problems.append((name, "bytes", st))
            if abs(ita - itb) > slack: problems.append((name, "iterations", ita, itb))
            err = np.linalg.norm(ua - ub) / np.linalg.norm(ua)
            if not err <= tol: problems.append((name, "u", err))
        st = b["multigrid"][3]
        if not (st["allreduces_small"] > 0 and st["fallback_allreduces"] == 0): problems.append(("multigrid", "all-reduces", st))
        # and back: the transport underneath serves again
        comm.disable_peer()
        c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
        u, i = solver.solve(f, rtol=1e-10, maxit=20000)
        if not (abs(i[0]["iterations"] - a["bj_cg"][1]) <= slack and c.dist_stats()["transport"] == 3): problems.append(("after disable", i[0]["iterations"], c.dist_stats()))
        ret[rank] = dict(iterations={k: v[1] for k, v in b.items()}, stats=b["multigrid"][3], desc=comm.describe(), problems=problems)
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,n,deg", [(2, 6, 2), (3, 4, 2), (2, 10, 1)])
def test_peer_transfers_equal_the_callback_transport(world, n, deg):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, deg, ret), nprocs=world, join=True)
    assert set(ret.keys()) == set(range(world))
    for r in range(world):
        assert not ret[r]["problems"], (r, ret[r]["problems"])
        assert ret[r]["iterations"] == ret[0]["iterations"]
    assert ret[0]["iterations"]["multigrid"] < 0.5 * ret[0]["iterations"]["two_level"] or deg == 1


def _worker_timeout(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D
    from meshfem_amd import _lib as L
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MFH_PEER_TIMEOUT_S"] = "2"
    torch.cuda.set_device(0)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        lm = D.slab_local_mesh(3, rank, world, 1)
        c = M.Context(0)
        c.mesh_set(3, 1, lm.elem_nodes, lm.node_pos, lm.n_owned)
        comm = D.Comm.callbacks(c, rank, world)
        comm.enable_peer()
        comm.selftest()
        x = torch.ones(4, dtype=torch.float64, device="cuda")
        if rank == 0:
            # rank 1 never joins this all-reduce: the wait gives up after MFH_PEER_TIMEOUT_S, the call reports it, the device is idle again
            st = c.lib.mfh_comm_allreduce(c.h, comm.h, x.data_ptr(), 4)
            ret["status"] = int(st)
            ret["message"] = c.lib.mfh_last_error(c.h).decode()
            torch.cuda.synchronize()
            ret["alive"] = float((x * 0 + 1).sum().item())
        dist.barrier()
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_missing_rank_times_out_instead_of_hanging_the_device():
    import torch.multiprocessing as mp
    from meshfem_amd import _lib as L
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_timeout, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["status"] == L.ERR_HIP and "waited more than" in ret["message"], dict(ret)
    assert ret["alive"] == 4.0


def _worker_fault(rank, world, port, ret):
    """bench.py's chain on two ranks: the solves on the transport underneath, then the peer transfers -- with rank 1 opening a handle whose
    bytes were overwritten (MFH_PEER_FAULT_RANK). Every rank must back out of the peer transfers, stay on the transport underneath, and
    solve to the same u as before; the preflight record says which slabs were mapped when the peer path does work."""
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MFH_PEER_TIMEOUT_S", "10")
    torch.cuda.set_device(0)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        n, deg = 6, 2
        lm = D.slab_local_mesh(n, rank, world, deg)
        c = M.Context(0)
        c.mesh_set(3, deg, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        c.set_option("deterministic", 1)       # run-to-run identical bits: "the same u" can then be checked with array_equal
        dev = torch.device("cuda", 0)
        comm, group, tried = D.robust_comm(c, rank, world, dev)
        pf0 = comm.preflight(world, 1 << 20)
        solver = D.DistSolver(c, lm, rank, world, comm, group)
        all_fixed = np.flatnonzero(lm.lattice[:, 0] == 0)
        c.fix_variables((3 * all_fixed[:, None] + np.arange(3)[None, :]).ravel())
        f = D.slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel()
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u0, i0 = solver.solve(f, rtol=1e-10, maxit=2000)
        # the peer leg with the fault: unavailable on EVERY rank, nobody left waiting
        os.environ["MFH_PEER_FAULT_RANK"] = "1"
        ok, outcome = D.try_enable_peer(comm, rank, world, dev, group)
        del os.environ["MFH_PEER_FAULT_RANK"]
        solver = D.DistSolver(c, lm, rank, world, comm, group)
        u1, i1 = solver.solve(f, rtol=1e-10, maxit=2000)
        st1 = c.dist_stats()
        # ... and without the fault the same chain lands on the peer transfers
        ok2, outcome2 = D.try_enable_peer(comm, rank, world, dev, group)
        pf2 = comm.preflight(world, 1 << 20) if ok2 else None
        solver = D.DistSolver(c, lm, rank, world, comm, group)
        u2, i2 = solver.solve(f, rtol=1e-10, maxit=2000)
        st2 = c.dist_stats()
        ret[rank] = dict(tried=tried, pf0=pf0, pf2=pf2, fault=(ok, outcome), clean=(ok2, outcome2), it=(i0[0]["iterations"], i1[0]["iterations"], i2[0]["iterations"]),
                         same_u_after_fault=bool(np.array_equal(u0[0], u1[0])), err_peer=float(np.linalg.norm(u2[0] - u0[0]) / np.linalg.norm(u0[0])),
                         transport_after_fault=st1["transport"], peer_after_fault=st1["peer_enabled"], transport_clean=st2["transport"])
        if ok2:
            comm.disable_peer()
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_failing_ipc_handle_lands_on_the_next_transport_with_identical_u():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_fault, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert set(ret.keys()) == {0, 1}
    for r in (0, 1):
        rec = ret[r]
        assert rec["fault"][0] is False and "unavailable" in rec["fault"][1], rec["fault"]
        assert rec["same_u_after_fault"] and rec["it"][0] == rec["it"][1]          # same transport, same bits
        assert rec["transport_after_fault"] == 3 and rec["peer_after_fault"] == 0  # callbacks (the ranks share one device: gloo)
        assert rec["clean"][0] is True and rec["transport_clean"] == 2
        assert rec["err_peer"] <= 1e-12 and rec["it"][2] == rec["it"][0]
        pf0, pf2 = rec["pf0"], rec["pf2"]
        assert pf0["allreduce_of_ones"] == 2.0 and pf0["ring_GBs_transport_underneath"] > 0 and pf0["peer_transfers_enabled"] is False
        assert pf0["can_access_peer"] == ["same device", "same device"]
        assert pf2["peer_transfers_enabled"] and pf2["ipc_slab_mapped"] == [True, True] and pf2["ring_GBs_peer_transfers"] > 0
        assert pf2["device_free_GB"] > 0 and pf2["arena_held_GB"] >= pf2["arena_live_GB"] > 0
