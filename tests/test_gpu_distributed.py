"""Two ranks sharing ONE MI355X (gloo rendezvous; the communicator's two collectives are callbacks that stage through the
host): the row-partitioned HIP path inside the library -- local assembly of owned rows, mfh_dist_setup, mfh_dist_solve
(Chronopoulos-Gear PCG, packed halo buffers, exchange overlapped with the interior element blocks, one fused
all-reduce per iteration) with block-Jacobi and with the two-level preconditioner on global aggregates -- against the
single-context solve of the same problem. The 8-GPU box runs the same code over the library's RCCL communicator; that
transport is covered at world size 1 (self send/receive + all-reduce) by test_rccl_communicator_world_1."""
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


def _worker(rank, world, port, n, ret, storage=-1):
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
        c.set_option("matrix_storage", storage)
        c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        c.assemble()
        # quadratic elasticity on the matrix-free operator: the automatic choice stores the upper triangle of the owned rows
        assert c.matrix_storage()[0] == (storage != 0)
        comm = D.make_comm(c, rank, world)            # gloo process group -> callbacks staged through the host
        assert comm.kind == "callbacks"
        comm.selftest()
        solver = D.DistSolver(c, lm, rank, world, comm)
        halo = D.HaloExchange(lm, rank, world, torch.device("cpu"))
        owned_fixed = np.flatnonzero(lm.lattice[:lm.n_owned, 0] == 0)
        ov = (3 * owned_fixed[:, None] + np.arange(3)[None, :]).ravel()
        ev, evals = D.extend_fixed_to_halo(halo, lm.n_local, 3, ov, None, torch.device("cpu"))
        all_fixed = np.flatnonzero(lm.lattice[:, 0] == 0)
        assert np.array_equal(ev, (3 * all_fixed[:, None] + np.arange(3)[None, :]).ravel())
        c.fix_variables(ev, evals)
        f = D.slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel()
        u1, i1 = solver.solve(f, rtol=1e-10, maxit=20000)
        # the operator alone: K applied to the solution reproduces the load on the free owned variables
        Ku = solver.apply_K(u1[0])
        free = np.ones(3 * lm.n_owned, bool); free[ov] = False
        assert np.linalg.norm((Ku - f)[free]) <= 1e-8 * np.linalg.norm(f)
        # the classic loop (two all-reduces per iteration) gives the same solution as Chronopoulos-Gear (one)
        c.set_option("dist_pcg_variant", 0)
        u1c, i1c = solver.solve(f, rtol=1e-10, maxit=20000)
        assert i1c[0]["converged"] and np.linalg.norm(u1c[0] - u1[0]) <= 1e-8 * np.linalg.norm(u1[0])
        assert abs(i1c[0]["iterations"] - i1[0]["iterations"]) <= max(3, 0.05 * i1[0]["iterations"])
        c.set_option("dist_pcg_variant", 1)
        tl = solver.two_level(16 * world)
        u2, i2 = solver.solve(f, rtol=1e-10, maxit=20000)
        c.set_option("dist_pcg_variant", 0)
        u2c, i2c = solver.solve(f, rtol=1e-10, maxit=20000)
        assert i2c[0]["converged"] and np.linalg.norm(u2c[0] - u2[0]) <= 1e-8 * np.linalg.norm(u2[0])
        c.set_option("dist_pcg_variant", 1)
        # three right-hand sides at once (batch of 2 + 1): scaled copies of the load
        F3 = np.stack([f, -2.0 * f, 0.5 * f])
        c.set_option("batch_rhs", 1)
        u3, i3 = solver.solve(F3, rtol=1e-10, maxit=20000)
        assert [i["reserved"] for i in i3] == [2, 2, 1]
        for k, sc in enumerate((1.0, -2.0, 0.5)):
            assert np.linalg.norm(u3[k] - sc * u2[0]) <= 1e-7 * np.linalg.norm(u2[0])
        ret[rank] = dict(keys=lm.keys[:lm.n_owned].copy(), u_bj=u1[0].reshape(-1, 3), u_tl=u2[0].reshape(-1, 3),
                         it_bj=i1[0]["iterations"], it_tl=i2[0]["iterations"], conv=(i1[0]["converged"], i2[0]["converged"]),
                         n_agg=int(np.prod(tl["bins"])))
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("storage", [-1, 0])
def test_two_ranks_one_gpu_partitioned_solve_matches_single_context(storage):
    """storage: option matrix_storage of the ranks' contexts (-1 automatic = upper triangle of the owned rows, the blocks towards
    halo columns kept by both ranks and counted half in the Galerkin product; 0 = both triangles): same solutions, same
    iteration counts."""
    import torch.multiprocessing as mp
    import meshfem_amd as M
    from meshfem_amd import grid
    world, n = 2, 6
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, ret, storage), nprocs=world, join=True)
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
    _ITERATIONS[storage] = (ret[0]["it_bj"], ret[0]["it_tl"])
    if len(_ITERATIONS) == 2:                                  # the coarse operator is the same whatever the storage
        assert abs(_ITERATIONS[-1][1] - _ITERATIONS[0][1]) <= 1 and abs(_ITERATIONS[-1][0] - _ITERATIONS[0][0]) <= 1, _ITERATIONS


_ITERATIONS = {}


def _worker_nonzero_dirichlet(rank, world, port, n, ret):
    """Non-zero Dirichlet data (a) on the x = 0 face, which crosses the inter-rank plane -- the fixed list of a rank then holds
    halo variables, whose indices lie past the end of the owned-row vectors (out-of-bounds scatter) -- and (b) on the
    top z face, which only the last rank sees (the lift b = f - K ubar is a collective and must not depend on a
    rank-local flag)."""
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lm = D.slab_local_mesh(n, rank, world, 2)
        c = M.Context(0)
        c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        c.assemble()
        comm = D.make_comm(c, rank, world)
        solver = D.DistSolver(c, lm, rank, world, comm)
        halo = D.HaloExchange(lm, rank, world, torch.device("cpu"))
        lat = lm.lattice[:lm.n_owned]
        zmax = 4 * n * world
        on_x0 = lat[:, 0] == 0
        on_top = (lat[:, 2] == zmax) & ~on_x0
        nodes = np.flatnonzero(on_x0 | on_top)
        pos = lm.node_pos[nodes]
        vals = np.where(on_x0[nodes][:, None], np.stack([0 * pos[:, 2], 0.01 * pos[:, 1], 0.02 * pos[:, 2]], 1),
                        np.array([[0.03, 0.0, -0.01]]))
        ov = (3 * nodes[:, None] + np.arange(3)[None, :]).ravel()
        ev, evals = D.extend_fixed_to_halo(halo, lm.n_local, 3, ov, vals.ravel(), torch.device("cpu"))
        if rank == 0:
            assert ev.max() >= 3 * lm.n_owned            # halo variables in the fixed list of rank 0 ...
            assert not np.any(on_top)                    # ... and the top face is invisible to it
        c.fix_variables(ev, evals)
        f = np.zeros(3 * lm.n_owned)
        out = {}
        for variant in (1, 0):
            c.set_option("dist_pcg_variant", variant)
            u, info = solver.solve(f, rtol=1e-10, maxit=20000)
            assert info[0]["converged"]
            out[variant] = u[0].reshape(-1, 3).copy()
        assert np.linalg.norm(out[0] - out[1]) <= 1e-8 * np.linalg.norm(out[1])
        assert np.allclose(out[1].ravel()[ov], vals.ravel(), rtol=0, atol=1e-15)     # u = x + ubar on the owned rows
        ret[rank] = dict(keys=lm.keys[:lm.n_owned].copy(), u=out[1])
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_nonzero_dirichlet_values_across_the_rank_interface():
    import torch.multiprocessing as mp
    import meshfem_amd as M
    from meshfem_amd import grid
    world, n = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_nonzero_dirichlet, args=(world, _free_port(), n, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    V, T = grid.grid_tet_mesh(n, n, n * world, [0, 0, 0], [1, 1, world])
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200.0, 0.35)
    pos = c.node_positions()
    on_x0 = np.abs(pos[:, 0]) < 1e-12
    on_top = (np.abs(pos[:, 2] - world) < 1e-12) & ~on_x0
    nodes = np.flatnonzero(on_x0 | on_top)
    p = pos[nodes]
    vals = np.where(on_x0[nodes][:, None], np.stack([0 * p[:, 2], 0.01 * p[:, 1], 0.02 * p[:, 2]], 1), np.array([[0.03, 0.0, -0.01]]))
    c.fix_variables((3 * nodes[:, None] + np.arange(3)[None, :]).ravel(), vals.ravel())
    u_ref = c.solve(np.zeros(3 * len(pos)), rtol=1e-10).reshape(-1, 3)
    c.close()
    lat = np.rint(pos * 4 * n).astype(np.int64)
    keys = (lat[:, 0] * (4 * n + 1) + lat[:, 1]) * (4 * n * world + 1) + lat[:, 2]
    order = np.argsort(keys)
    for r in range(world):
        d = ret[r]
        idx = order[np.searchsorted(keys[order], d["keys"])]
        assert np.array_equal(keys[idx], d["keys"])
        # both sides are PCG solutions to rtol 1e-10 of the same SPD system
        assert np.linalg.norm(d["u"] - u_ref[idx]) / np.linalg.norm(u_ref) < 1e-7, r


def _worker_general(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D, mesh_io
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
        V, E, _ = mesh_io.load_msh(os.path.join(gold, "meshes", "ball.msh"))
        g = np.load(os.path.join(gold, "example_meshes.npz"))
        lm = D.distribute_mesh(V, E, 2, rank, world)
        c = M.Context(0)
        c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        c.assemble()
        comm = D.make_comm(c, rank, world)
        solver = D.DistSolver(c, lm, rank, world, comm)
        gfixed = np.zeros(3 * lm.n_global, bool)
        gfixed[g["ball_p2_fixed_vars"]] = True                      # Dirichlet variables of the committed golden problem
        lvars = np.flatnonzero(gfixed[(3 * lm.keys[:, None] + np.arange(3)).ravel()])
        c.fix_variables(lvars)
        f = g["ball_p2_load"][lm.keys[:lm.n_owned]].ravel().copy()
        solver.two_level(8)
        u, infos = solver.solve(f, rtol=1e-10, maxit=20000)
        info = infos[0]
        u_ref = g["ball_p2_u"]
        err = np.linalg.norm(u[0].reshape(-1, 3) - u_ref[lm.keys[:lm.n_owned]]) / np.linalg.norm(u_ref)
        ret[rank] = (err, bool(info["converged"]), lm.n_owned, info["iterations"])
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_rcb_unstructured_mesh_matches_golden():
    """ball.msh (P2) split by recursive coordinate bisection over two ranks sharing the GPU: partitioned HIP assembly,
    matrix-free operator on the owned rows, global two-level preconditioner -- against the committed direct solve."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_general, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    assert all(ret[r][1] for r in range(world))
    assert max(ret[r][0] for r in range(world)) < 1e-6, dict(ret)          # north-star tolerance on displacements
    assert ret[0][2] + ret[1][2] == len(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_meshes.npz"))["ball_p2_u"])   # every P2 node owned once
    assert ret[0][3] == ret[1][3]


def _worker_scatter(rank, world, port, ret):
    """scatter_mesh (only rank 0 reads the mesh, every rank numbers its own share) feeding the HIP path: the golden problem of ball.msh, with
    the golden's global vectors indexed through the node keys (vertex id / edge_node_key of the end vertices)."""
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
        V = E = None
        if rank == 0:
            V, E, _ = mesh_io.load_msh(os.path.join(gold, "meshes", "ball.msh"))
        lm = D.scatter_mesh(V, E, 2, rank, world)
        # the test's own reference numbering (global FEM node ids <-> keys)
        Vg, Eg, _ = mesh_io.load_msh(os.path.join(gold, "meshes", "ball.msh"))
        h = M.Context(-1)
        h.mesh_build(Eg, Vg, 2)
        en_g = h.elem_nodes().astype(np.int64)
        n_glob = h.n_node
        h.close()
        key_g = np.arange(n_glob, dtype=np.int64)
        for k, (a, b) in enumerate(D._LOCAL_EDGES[3]):
            key_g[en_g[:, 4 + k]] = D.edge_node_key(en_g[:, a], en_g[:, b], len(Vg))
        order = np.argsort(key_g)
        gid = order[np.searchsorted(key_g[order], lm.keys)]
        g = np.load(os.path.join(gold, "example_meshes.npz"))
        c = M.Context(0)
        c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        comm = D.make_comm(c, rank, world)
        solver = D.DistSolver(c, lm, rank, world, comm)
        gfixed = np.zeros(3 * n_glob, bool)
        gfixed[g["ball_p2_fixed_vars"]] = True
        c.fix_variables(np.flatnonzero(gfixed[(3 * gid[:, None] + np.arange(3)).ravel()]))
        f = g["ball_p2_load"][gid[:lm.n_owned]].ravel().copy()
        # an unstructured RCB partition with the aggregate levels PARTITIONED too (forced: the mesh is far below the default threshold)
        c.set_option("mg_replicate_max", 6)
        c.set_option("mg_dense_max", 4)
        c.set_option("mg_agg_target", 8)
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u, infos = solver.solve(f, rtol=1e-10, maxit=5000)
        levels = c.multigrid_levels()
        u_ref = g["ball_p2_u"]
        err = np.linalg.norm(u[0].reshape(-1, 3) - u_ref[gid[:lm.n_owned]]) / np.linalg.norm(u_ref)
        ret[rank] = (err, bool(infos[0]["converged"]), lm.n_owned, infos[0]["iterations"], int(lm.n_local), int(n_glob), levels)
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_scatter_mesh_three_ranks_matches_golden():
    """ball.msh (P2) scattered from rank 0 over three ranks sharing the GPU, multigrid PCG on the partitioned contexts, against the committed
    direct solve (north-star tolerance 1e-6 on the displacements)."""
    import torch.multiprocessing as mp
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_scatter, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1, 2}
    assert all(ret[r][1] for r in range(world))
    assert max(ret[r][0] for r in range(world)) < 1e-6, dict(ret)
    assert sum(ret[r][2] for r in range(world)) == ret[0][5]             # every P2 node owned once
    assert len({ret[r][3] for r in range(world)}) == 1
    assert max(ret[r][4] for r in range(world)) < 0.8 * ret[0][5]         # no rank holds the whole mesh
    lv = ret[0][6]
    assert len(lv) >= 2 and lv[0]["partitioned"] == 1 and lv[-1]["partitioned"] == 0, lv
    assert sum(ret[r][6][0]["rows"] for r in range(world)) <= lv[0]["aggregates"]


def test_rccl_communicator_world_1():
    """The library's RCCL communicator (looked up with dlopen) on this GPU: unique id, ncclCommInitRank, grouped
    ncclSend / ncclRecv to itself and ncclAllReduce with known answers (mfh_comm_selftest), then the distributed entry
    points with that communicator at world size 1 against the single-context solve."""
    import meshfem_amd as M
    from meshfem_amd import distributed as D, grid
    n = 5
    lm = D.slab_local_mesh(n, 0, 1, 2)
    c = M.Context(0)
    c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned)
    c.material_isotropic(200.0, 0.35)
    c.assemble()
    comm = D.Comm.rccl(c, 0, 1)
    assert "RCCL" in comm.describe()
    comm.selftest()
    solver = D.DistSolver(c, lm, 0, 1, comm)
    fixed = np.flatnonzero(lm.lattice[:, 0] == 0)
    c.fix_variables((3 * fixed[:, None] + np.arange(3)[None, :]).ravel())
    f = D.slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel()
    u, infos = solver.solve(f, rtol=1e-10)
    assert infos[0]["converged"] and infos[0]["true_rel_residual"] < 1e-9
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    sim = M.Simulator(T, V, 2)
    sim.setIsotropicMaterial(200.0, 0.35)
    sim.ctx.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0])
    sim.ctx.bc_neumann_box([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    sim.rtol = 1e-10
    u_ref = sim.solve()
    lat = np.rint(sim.nodes() * 4 * n).astype(np.int64)
    keys = (lat[:, 0] * (4 * n + 1) + lat[:, 1]) * (4 * n + 1) + lat[:, 2]
    order = np.argsort(keys)
    idx = order[np.searchsorted(keys[order], lm.keys[:lm.n_owned])]
    assert np.linalg.norm(u[0].reshape(-1, 3) - u_ref[idx]) / np.linalg.norm(u_ref) < 1e-7
    comm.close()
    c.close()


def test_callback_communicator_over_a_nccl_process_group_world_1():
    """The fallback transport when torch.distributed runs on nccl (= RCCL): the callbacks stage the library's device buffers
    through DEVICE tensors (a nccl group moves nothing else). World size 1: the all-reduce path of a partitioned solve."""
    import socket
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D
    for attempt in range(5):       # a port found free may be taken again before the store binds it (seen once on a busy box: EADDRINUSE)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        try:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
            break
        except dist.DistNetworkError:
            if attempt == 4:
                raise
    try:
        n = 4
        lm = D.slab_local_mesh(n, 0, 1, 2)
        c = M.Context(0)
        c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned)
        c.material_isotropic(200.0, 0.35)
        c.assemble()
        comm = D.make_comm(c, 0, 1, prefer="callbacks")
        assert "callbacks" in comm.describe()
        comm.selftest()
        solver = D.DistSolver(c, lm, 0, 1, comm)
        fixed = np.flatnonzero(lm.lattice[:, 0] == 0)
        c.fix_variables((3 * fixed[:, None] + np.arange(3)[None, :]).ravel())
        f = D.slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel()
        u, infos = solver.solve(f, rtol=1e-10)
        assert infos[0]["converged"] and infos[0]["true_rel_residual"] < 1e-9
        c.set_option("dist_pcg_variant", 0)
        u2, infos2 = solver.solve(f, rtol=1e-10)
        assert np.linalg.norm(u2 - u) < 1e-8 * np.linalg.norm(u)
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


def _worker_periodic(rank, world, port, n, ret):
    import torch
    import torch.distributed as dist
    import meshfem_amd as M
    from meshfem_amd import distributed as D, grid
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
        E, nu = _periodic_cell_material(V, T)
        lm = D.distribute_periodic_mesh(V, T, 2, rank, world)
        c = M.Context(0)
        c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos)
        c.material_iso_field(E[lm.kept_elems], nu[lm.kept_elems])
        c.dof_map_partitioned(lm.dof_for_node, lm.n_local, lm.n_owned)
        comm = D.make_comm(c, rank, world)
        solver = D.DistSolver(c, lm, rank, world, comm)          # exchange lists in DoF numbers
        pinned = np.flatnonzero(lm.keys == 0)                     # global DoF 0, on every rank that sees it (owner or halo)
        c.fix_variables((3 * pinned[:, None] + np.arange(3)[None, :]).ravel())
        out = {}
        for k in (0, 5):                                          # a stretch and a shear probe
            cs = np.zeros(6); cs[k] = -1.0 if k < 3 else -0.5
            f = c.constant_strain_load(cs)[:lm.n_owned].ravel().copy()
            u, infos = solver.solve(f, rtol=1e-10, maxit=20000)
            out[k] = (u[0].reshape(-1, 3), bool(infos[0]["converged"]), infos[0]["iterations"])
            Ku = solver.apply_K(u[0])
            free = np.ones(3 * lm.n_owned, bool)
            free[(3 * pinned[pinned < lm.n_owned][:, None] + np.arange(3)[None, :]).ravel()] = False
            assert np.linalg.norm((Ku - f)[free]) <= 1e-8 * np.linalg.norm(f)
        # the two-level preconditioner on global aggregates of the DoFs: same solution, fewer iterations
        solver.two_level(8 * world)
        u_tl, infos = solver.solve(f, rtol=1e-10, maxit=20000)
        assert infos[0]["converged"], infos[0]
        assert np.linalg.norm(u_tl[0] - u[0]) <= 1e-7 * np.linalg.norm(u[0]), np.linalg.norm(u_tl[0] - u[0]) / np.linalg.norm(u[0])
        assert infos[0]["iterations"] <= out[5][2], (infos[0]["iterations"], out[5][2])
        # ... and the multigrid V-cycle: coarse DoFs in the order of the fine ones, aggregates on a periodic global lattice (positions wrapped to
        # the minimal faces, so that every rank bins a DoF alike whichever of its periodic images it holds)
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u_mg, infos = solver.solve(f, rtol=1e-10, maxit=2000)
        assert infos[0]["converged"] and c.precond_info()["note"] == "", (infos[0], c.precond_info())
        assert np.linalg.norm(u_mg[0] - u[0]) <= 1e-7 * np.linalg.norm(u[0]), np.linalg.norm(u_mg[0] - u[0]) / np.linalg.norm(u[0])
        assert infos[0]["iterations"] < 0.7 * out[5][2], (infos[0]["iterations"], out[5][2])
        ret[rank] = dict(keys=lm.keys[:lm.n_owned].copy(), out=out)
        comm.close()
        c.close()
    finally:
        dist.destroy_process_group()


def _periodic_cell_material(V, T):
    """a smooth periodic stiffness field (an inclusion in the middle of the cell), the same on every rank"""
    ctr = V[T].mean(axis=1)
    r2 = ((ctr - 0.5) ** 2).sum(axis=1)
    return 100.0 + 200.0 * np.exp(-r2 / 0.05), np.full(len(T), 0.3)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_periodic_cell_problem_with_a_partitioned_dof_map(world):
    """mfh_dof_map_partitioned: the periodic DoF map of a homogenization cell on a row-partitioned context -- rows and exchange lists in DoF
    numbers, a DoF's nodes on opposite cell faces possibly on different ranks' local meshes. Against the single-context solve."""
    import torch.multiprocessing as mp
    import meshfem_amd as M
    from meshfem_amd import grid
    n = 6
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_periodic, args=(world, _free_port(), n, ret), nprocs=world, join=True)
    assert set(ret.keys()) == set(range(world))
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    E, nu = _periodic_cell_material(V, T)
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_iso_field(E, nu)
    nd = c.apply_periodic_conditions()
    c.fix_variables(np.arange(3))
    seen = sum(len(ret[r]["keys"]) for r in range(world))
    assert seen == nd                                           # every DoF owned exactly once
    for k in (0, 5):
        cs = np.zeros(6); cs[k] = -1.0 if k < 3 else -0.5
        f = c.constant_strain_load(cs)
        u_ref = c.solve(f.ravel(), rtol=1e-10).reshape(-1, 3)
        assert np.abs(u_ref).max() > 1e-3                       # the inclusion makes the fluctuation non-trivial
        for r in range(world):
            u, conv, _ = ret[r]["out"][k]
            assert conv and np.linalg.norm(u - u_ref[ret[r]["keys"]]) / np.linalg.norm(u_ref) < 1e-7, (r, k)
        assert len({ret[r]["out"][k][2] for r in range(world)}) == 1
    c.close()
