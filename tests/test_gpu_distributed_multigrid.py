"""MFH_PRECOND_MULTIGRID on row-partitioned contexts: ranks sharing ONE MI355X (gloo rendezvous, the communicator's collectives are
host-staged callbacks). The nodal levels (quadratic, linear) are partitioned like the mesh -- halo exchanges inside the Chebyshev
smoothers, before the restriction and before the prolongation --, the aggregate levels are replicated (one all-reduce of the
restricted residual per V-cycle). Checked against the single-context solve of the same problem: same displacements, (about) the
same mesh-independent iteration count as the unpartitioned V-cycle, far fewer iterations than the partitioned two-level solve."""
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
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lm = D.slab_local_mesh(n, rank, world, deg)
        c = M.Context(0)
        c.mesh_set(3, deg, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        comm = D.make_comm(c, rank, world)
        solver = D.DistSolver(c, lm, rank, world, comm)
        all_fixed = np.flatnonzero(lm.lattice[:, 0] == 0)
        c.fix_variables((3 * all_fixed[:, None] + np.arange(3)[None, :]).ravel())
        f = D.slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel()
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u_mg, i_mg = solver.solve(f, rtol=1e-10, maxit=2000)
        g, p = c.multigrid_info(), c.precond_info()
        # a second solve reuses the hierarchy; other smoother settings keep the answer
        c.set_option("mg_steps_fine", 2); c.set_option("mg_steps_coarse", 2)
        u_mg2, i_mg2 = solver.solve(f, rtol=1e-10, maxit=2000)
        c.set_option("mg_steps_fine", 1); c.set_option("mg_steps_coarse", 1)
        tl = solver.two_level(16 * world)                       # switches the context to the two-level preconditioner
        u_tl, i_tl = solver.solve(f, rtol=1e-10, maxit=20000)
        # the whole mesh dilated by 2 on every rank (mfh_mesh_update_vertices on a partitioned context): K scales with length^(dim - 2),
        # the same load then moves the nodes half as far; the multigrid hierarchy is rebuilt on the new geometry
        c.mesh_update_vertices(2.0 * np.asarray(lm.node_pos))
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u_dil, i_dil = solver.solve(f, rtol=1e-10, maxit=2000)
        assert i_dil[0]["converged"] and np.linalg.norm(u_dil[0] - 0.5 * u_mg[0]) <= 1e-7 * np.linalg.norm(u_mg[0])
        ret[rank] = dict(keys=lm.keys[:lm.n_owned].copy(), u_mg=u_mg[0].reshape(-1, 3), u_mg2=u_mg2[0].reshape(-1, 3), u_tl=u_tl[0].reshape(-1, 3),
                         it_mg=i_mg[0]["iterations"], it_mg2=i_mg2[0]["iterations"], it_tl=i_tl[0]["iterations"],
                         conv=(i_mg[0]["converged"], i_mg2[0]["converged"], i_tl[0]["converged"]), res=i_mg[0]["true_rel_residual"],
                         info=g, pinfo=p, owned=lm.n_owned)
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,n,deg", [(2, 6, 2), (3, 4, 2), (2, 10, 1)])
def test_partitioned_multigrid_matches_the_single_context_solve(world, n, deg):
    import torch.multiprocessing as mp
    import meshfem_amd as M
    from meshfem_amd import grid
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, deg, ret), nprocs=world, join=True)
    assert set(ret.keys()) == set(range(world))
    V, T = grid.grid_tet_mesh(n, n, n * world, [0, 0, 0], [1, 1, world])
    sim = M.Simulator(T, V, deg)
    sim.setIsotropicMaterial(200.0, 0.35)
    sim.ctx.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9 + world], [0, 0, 0])
    sim.ctx.bc_neumann_box([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9 + world], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    sim.ctx.set_preconditioner(M.PRECOND_MULTIGRID)
    sim.rtol = 1e-10
    u_ref = sim.solve()
    it_single = sim.info["iterations"]
    m = 4 * n                                                 # quarter lattice of the node keys (slab_local_mesh)
    lat = np.rint(sim.nodes() * m).astype(np.int64)
    keys = (lat[:, 0] * (m + 1) + lat[:, 1]) * (m * world + 1) + lat[:, 2]
    order = np.argsort(keys)
    seen = 0
    for r in range(world):
        d = ret[r]
        assert all(d["conv"]) and d["res"] < 2e-10, (r, d["conv"], d["res"])
        idx = order[np.searchsorted(keys[order], d["keys"])]
        assert np.array_equal(keys[idx], d["keys"])
        for name in ("u_mg", "u_mg2", "u_tl"):
            assert np.linalg.norm(d[name] - u_ref[idx]) / np.linalg.norm(u_ref) < 1e-7, (r, name)
        assert d["info"]["fine_dof"] == d["owned"] and d["pinfo"]["aggregates"] > 0 and d["pinfo"]["note"] == ""
        seen += len(idx)
    assert seen == len(keys)
    its = [ret[r]["it_mg"] for r in range(world)]
    assert len(set(its)) == 1                                  # every rank counts the same iterations
    # the partitioned V-cycle is the unpartitioned one up to the aggregate lattice (global bins instead of occupied bins)
    assert its[0] <= 1.3 * it_single + 4, (its, it_single)
    assert its[0] < 0.5 * ret[0]["it_tl"], (its, ret[0]["it_tl"])


def _worker_ball(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D, mesh_io
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
        V, E, _ = mesh_io.load_msh(os.path.join(gold, "meshes", "ball.msh"))
        g = np.load(os.path.join(gold, "example_meshes.npz"))
        lm = D.distribute_mesh(V, E, 2, rank, world)
        c = M.Context(0)
        c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        comm = D.make_comm(c, rank, world)
        solver = D.DistSolver(c, lm, rank, world, comm)
        gfixed = np.zeros(3 * lm.n_global, bool)
        gfixed[g["ball_p2_fixed_vars"]] = True
        c.fix_variables(np.flatnonzero(gfixed[(3 * lm.keys[:, None] + np.arange(3)).ravel()]))
        f = g["ball_p2_load"][lm.keys[:lm.n_owned]].ravel().copy()
        out = {}
        for name, opts in (("default", {}), ("small_bins", {"mg_agg_target": 8, "mg_dense_max": 40})):
            for k, v in opts.items():
                c.set_option(k, v)
            c.set_preconditioner(M.PRECOND_MULTIGRID)
            u, infos = solver.solve(f, rtol=1e-10, maxit=2000)
            u_ref = g["ball_p2_u"]
            out[name] = (float(np.linalg.norm(u[0].reshape(-1, 3) - u_ref[lm.keys[:lm.n_owned]]) / np.linalg.norm(u_ref)), bool(infos[0]["converged"]),
                         infos[0]["iterations"], c.precond_info())
        c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
        _, infos = solver.solve(f, rtol=1e-10, maxit=20000)
        out["block_jacobi_iterations"] = infos[0]["iterations"]
        out["owned"] = lm.n_owned
        ret[rank] = out
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_partitioned_multigrid_on_an_unstructured_rcb_split():
    """The reference's ball.msh (P2) cut by recursive coordinate bisection over three ranks: irregular halos, several peers per rank, and a
    global lattice whose corner bins are EMPTY (a ball in its bounding box) -- aggregates without a node keep zero blocks and drop out of the
    smoothers. Against the committed direct solve."""
    import torch.multiprocessing as mp
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_ball, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert set(ret.keys()) == set(range(world))
    n_nodes = len(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_meshes.npz"))["ball_p2_u"])
    assert sum(ret[r]["owned"] for r in range(world)) == n_nodes
    for name in ("default", "small_bins"):
        assert all(ret[r][name][1] for r in range(world)), name
        assert max(ret[r][name][0] for r in range(world)) < 1e-6, (name, [ret[r][name][0] for r in range(world)])
        assert len({ret[r][name][2] for r in range(world)}) == 1
        assert ret[0][name][2] < 0.5 * ret[0]["block_jacobi_iterations"], (name, ret[0][name][2], ret[0]["block_jacobi_iterations"])
    assert ret[0]["small_bins"][3]["aggregates"] > ret[0]["default"][3]["aggregates"]


def _worker_2d(rank, world, port, nx, deg, ret):
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D, grid
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, T = grid.grid_tri_mesh(nx, nx)
        V = V / float(nx)
        lm = D.distribute_mesh(V, T, deg, rank, world)
        c = M.Context(0)
        c.mesh_set(2, deg, lm.elem_nodes, lm.node_pos[:, :2], lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        comm = D.make_comm(c, rank, world)
        solver = D.DistSolver(c, lm, rank, world, comm)
        pos = np.asarray(lm.node_pos)[:, :2]
        fixed = np.flatnonzero(np.abs(pos[:, 0]) < 1e-12)
        c.fix_variables((2 * fixed[:, None] + np.arange(2)[None, :]).ravel())
        f = np.zeros((lm.n_owned, 2))
        f[np.abs(pos[:lm.n_owned, 0] - 1.0) < 1e-12, 1] = -1e-2                      # nodal forces on the right edge
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u, infos = solver.solve(f.ravel(), rtol=1e-10, maxit=2000)
        it_mg = infos[0]["iterations"]
        c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
        u2, infos2 = solver.solve(f.ravel(), rtol=1e-10, maxit=50000)
        ret[rank] = dict(keys=lm.keys[:lm.n_owned].copy(), u=u[0].reshape(-1, 2), u_bj=u2[0].reshape(-1, 2), it_mg=it_mg, it_bj=infos2[0]["iterations"],
                         conv=bool(infos[0]["converged"]) and bool(infos2[0]["converged"]), pinfo=c.precond_info())
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("deg", [2, 1])
def test_partitioned_multigrid_in_two_dimensions(deg):
    """Triangles, RCB over two ranks (the 2D instances of the transfer, stencil and aggregate kernels on a partitioned context)."""
    import torch.multiprocessing as mp
    import meshfem_amd as M
    from meshfem_amd import grid
    world, nx = 2, 24
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_2d, args=(world, _free_port(), nx, deg, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    V, T = grid.grid_tri_mesh(nx, nx)
    c = M.Context(0)
    c.mesh_build(T, V / float(nx), deg)
    c.material_isotropic(200.0, 0.35)
    pos = c.node_positions()
    fixed = np.flatnonzero(np.abs(pos[:, 0]) < 1e-12)
    c.fix_variables((2 * fixed[:, None] + np.arange(2)[None, :]).ravel())
    f = np.zeros((len(pos), 2)); f[np.abs(pos[:, 0] - 1.0) < 1e-12, 1] = -1e-2
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    u_ref = c.solve(f.ravel(), rtol=1e-10).reshape(-1, 2)
    it_single = c.last_info["iterations"]
    c.close()
    seen = 0
    for r in range(world):
        d = ret[r]
        assert d["conv"] and d["pinfo"]["aggregates"] >= 0
        for name in ("u", "u_bj"):
            assert np.linalg.norm(d[name] - u_ref[d["keys"]]) / np.linalg.norm(u_ref) < 1e-7, (r, name)
        seen += len(d["keys"])
    assert seen == len(pos)
    assert ret[0]["it_mg"] == ret[1]["it_mg"] and ret[0]["it_mg"] <= 1.3 * it_single + 4 and ret[0]["it_mg"] < 0.3 * ret[0]["it_bj"], (dict(ret[0]), it_single)


def _worker_partitioned_levels(rank, world, port, n, deg, ret):
    """The same solve with the aggregate levels replicated on every rank (mg_replicate_max 0: rounds 3-4) and PARTITIONED (every level
    with more than 10 aggregates; several of them, forced by a small dense level): the V-cycle is the same operator, only the sums of the
    shared aggregates are formed in another order -- same iteration count, u to rounding."""
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MFH_PEER_TIMEOUT_S", "20")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lm = D.slab_local_mesh(n, rank, world, deg)
        c = M.Context(0)
        c.mesh_set(3, deg, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        comm = D.make_comm(c, rank, world)
        solver = D.DistSolver(c, lm, rank, world, comm)
        all_fixed = np.flatnonzero(lm.lattice[:, 0] == 0)
        c.fix_variables((3 * all_fixed[:, None] + np.arange(3)[None, :]).ravel())
        f = D.slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel()
        c.set_option("mg_dense_max", 5)
        out, levels = {}, {}
        for name, rmax in (("replicated", 0), ("partitioned", 10)):
            c.set_option("mg_replicate_max", rmax)
            c.set_preconditioner(M.PRECOND_MULTIGRID)
            u, i = solver.solve(f, rtol=1e-10, maxit=2000)
            out[name] = (u[0].copy(), i[0]["iterations"], bool(i[0]["converged"]), i[0]["true_rel_residual"], c.precond_info())
            levels[name] = c.multigrid_levels()
        # ... and through the peer transfers (the level exchanges ride on the same lists): the same again
        comm.enable_peer()
        solver = D.DistSolver(c, lm, rank, world, comm)
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u, i = solver.solve(f, rtol=1e-10, maxit=2000)
        st = c.dist_stats()
        out["partitioned_peer"] = (u[0].copy(), i[0]["iterations"], bool(i[0]["converged"]), i[0]["true_rel_residual"], c.precond_info())
        comm.disable_peer()
        ur = out["replicated"][0]
        rec = {k: dict(it=v[1], conv=v[2], res=v[3], err=float(np.linalg.norm(v[0] - ur) / np.linalg.norm(ur)), aggregates=v[4]["aggregates"]) for k, v in out.items()}
        rec["levels"] = levels
        rec["peer_stats"] = dict(transport=st["transport"], fallback_exchanges=st["fallback_exchanges"], fallback_allreduces=st["fallback_allreduces"])
        ret[rank] = rec
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,n,deg", [(2, 8, 2), (3, 6, 2), (2, 12, 1)])
def test_partitioned_aggregate_levels_equal_the_replicated_ones(world, n, deg):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_partitioned_levels, args=(world, _free_port(), n, deg, ret), nprocs=world, join=True)
    assert set(ret.keys()) == set(range(world))
    for r in range(world):
        d = ret[r]
        for k in ("replicated", "partitioned", "partitioned_peer"):
            assert d[k]["conv"] and d[k]["res"] < 2e-10, (r, k, d[k])
            assert d[k]["it"] == ret[0]["replicated"]["it"], (r, k, d[k]["it"], ret[0]["replicated"]["it"])
            assert d[k]["err"] <= 1e-9, (r, k, d[k]["err"])
        assert d["peer_stats"]["transport"] == 2 and d["peer_stats"]["fallback_exchanges"] == 0, d["peer_stats"]
        rep, par = d["levels"]["replicated"], d["levels"]["partitioned"]
        assert len(rep) == len(par) >= 3 and not any(L["partitioned"] for L in rep)
        assert [L["aggregates"] for L in rep] == [L["aggregates"] for L in par]
        for L in par:           # partitioned exactly where the level has more than 10 aggregates, never the dense last one
            assert L["partitioned"] == (1 if (L["aggregates"] > 10 and L is not par[-1]) else 0), par
            if L["partitioned"]:
                assert L["rows"] < L["aggregates"] and L["peers"] >= 1 and L["halo_received"] > 0 and L["entries"] == L["rows"] + L["halo_received"]
    # every aggregate of a partitioned level is owned by exactly one rank
    for l, L in enumerate(ret[0]["levels"]["partitioned"]):
        if L["partitioned"]:
            assert sum(ret[r]["levels"]["partitioned"][l]["rows"] for r in range(world)) <= L["aggregates"]       # (< : empty bins of the lattice have no owner)
            assert sum(ret[r]["levels"]["partitioned"][l]["rows"] for r in range(world)) >= 0.5 * L["aggregates"]
