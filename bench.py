#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native MeshFEM hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: the script launches its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

Metric (BASELINE.json): stiffness-assembly elements/s (+ PCG DOF/s), quadratic tets.
  * workload at N=1: BASELINE.json configs[2] -- synthetic 60^3 grid -> 5,184,000 P2 tets
    (22,292,283 DOF), isotropic E=200 nu=0.35, u=0 on x=0, traction (0,-1,0) on x=1.
  * a STEP = one numeric pass of the hot path over the whole mesh: element embedding (a1) +
    per-element stiffness blocks (a2-a5) + global assembly into the device block-CSR (a6,a9).
    Inputs (connectivity, coordinates, material, gather lists) are resident in HBM.
    The symbolic phase (pattern + gather lists) is once-per-mesh setup and reported separately.
  * `value` = elements * K / (wall time of the K timed steps), max over ranks.
  * N>1: weak scaling towards BASELINE configs[4] -- the global grid is n x n x (L N) with n ~ 60 N^(1/3) and
    L ~ 60^3 / n^2 hex layers per rank (N = 8: the 120^3 cube, 41.5 M P2 tets, 15-layer z-slabs), so every rank
    keeps ~5.2 M elements (row/element partition, owner computes, no assembly communication); the PCG leg
    runs the distributed solver (halo exchange + all-reduces over RCCL, global two-level preconditioner).
Every N = 1 line also carries BASELINE configs[1] (`variants.config1_p1`), configs[3] (`variants.config3_homogenization`: six cell problems end to end, Ch
checked in-line) and configs[4]'s cube in one context (`strong_scaling_n1`: run FIRST, in a process of its own -- `python bench.py --leg strong_n1` --, and
again inside this process at the end as `warm_process`); `--leg config{1,2,3}|strong_n1` runs one workload alone (one rocprofv3 trace per workload).
Extra objects on the JSON line: `roofline` (assembly kernel: the contract fraction on SURVEY's algorithmic bytes AND the
fraction on the bytes the counters / the design actually move), `pcg` (solve to 1e-8 incl. its own roofline for the
operator kernels), `cpu_baseline` (plain-C port of the reference loop structure timed on the host cores, bounded sample,
plus a DIRECT SOLVE of a bounded sample next to the PCG at the same size), `variants` (atomic-scatter assembly).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 achievable
PMC_PROFILE = "r06_pmc_traffic_n60.json"      # scripts/pmc_collect.py at this round's kernels, both triangles of K stored (k_spmv, k_mf_*)
PMC_PROFILE_UPPER = "r06_pmc_traffic_n60_upper_storage.json"   # the same with the upper-triangle storage (PMC_UPPER_STORAGE=1): the default of configs[2]
PMC_KERNEL_KEY = "k_assemble_gather"
ALG_BYTES = {(3, 2): 7736, (3, 1): 1328, (2, 2): 1368, (2, 1): 0}   # SURVEY.md section 8(d), const material, both triangles
# SURVEY.md section 8(d), row "upper-only variant, matches reference storage": 40 + 96 + 55 x 4 + 55 x 72 (P2 tet); P2 triangle alike
ALG_BYTES_UPPER = {(3, 2): 4316, (3, 1): 872, (2, 2): 24 + 48 + 21 * 4 + 21 * 32}   # (3, 1): 112 + 10 blocks x 76 B (1 328 - 6 x 76)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=0, help="hexes per side (0 = 60: configs[2] / the per-rank volume of the weak-scaling run; "
                                                        "119 with --scaling strong: configs[4])")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("MFH_BENCH_SCALING") or None,
                    help="default: N = 1 runs configs[2] (60^3) and adds the 119^3 cube in one context as `strong_scaling_n1`; N > 1 runs the "
                         "STRONG-scaling line -- ONE grid^3 cube (119^3 = configs[4], 40.4 M P2 tets) dealt out over the N ranks (north_star: "
                         ">= 6x at 8 GPUs) -- and adds the weak-scaling run (~60^3 hexes per rank; N = 8: the 120^3 cube) as `weak_scaling`. "
                         "--scaling weak: the weak line alone; --scaling strong with N = 1: the 119^3 cube as the line itself")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the secondary weak-scaling object")
    ap.add_argument("--no-strong-n1", action="store_true", help="N = 1: skip the strong_scaling_n1 object (the 119^3 cube in one context)")
    ap.add_argument("--deg", type=int, default=2)
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--maxit", type=int, default=20000)
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--coarse-aggregates", type=int, default=0,
                    help="N>1: global aggregates of the two-level preconditioner (0 = min(1000 N, 2048); -1 = block-Jacobi only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-orderings", action="store_true", help="skip the shuffled / Morton-ordered mesh variants")
    ap.add_argument("--cpu-grid", type=int, default=0, help="grid size of the CPU-baseline sample (0 = auto)")
    ap.add_argument("--cpu-solve-grid", type=int, default=0, help="grid size of the CPU direct-solve sample (0 = auto)")
    ap.add_argument("--leg", default="", help="internal: run ONE leg of the N = 1 line in this (fresh) process and print its JSON object "
                                              "(strong_n1: the 119^3 cube in one context, what a one-shot caller of that size sees)")
    ap.add_argument("--no-forced-ranks", action="store_true", help="N = 1: skip variants.forced_2_ranks_one_gpu (48^3 cube over two processes on the one device)")
    ap.add_argument("--no-config3", action="store_true", help="N = 1: skip variants.config3_homogenization (44^3 periodic cell, 6 cell problems)")
    ap.add_argument("--ranks-per-gpu-ok", action="store_true",
                    help="N > visible GPUs: share the GPUs (ranks on one GPU talk through gloo, staged through the host)")
    a = ap.parse_args()
    if a.grid <= 0:
        a.grid = 119 if (a.scaling == "strong" or (a.scaling is None and a.gpus > 1)) else 60
    return a


def pmc_traffic(kernel_key, n, deg, profile=None):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in their
    own runs, calibrated on a known byte count: profiles/r01_pmc_traffic_n60.json). PMC counters cannot
    be collected from inside this process; the figure is only attached when the workload matches."""
    name = profile or PMC_PROFILE
    for cand in (name, name.replace("r06_", "r05_")):      # this round's passes; the previous round's (same kernels) until they are committed
        path = os.path.join(ROOT, "profiles", cand)
        try:
            with open(path) as f:
                d = json.load(f)
            if d["meta"]["n"] == n and deg == 2:
                return d[kernel_key]["traffic_bytes"], os.path.relpath(path, ROOT)
        except Exception:
            pass
    return None, None


PROFILE_ROUND = "r06"     # prefix of the committed per-workload traces under profiles/ (scripts/r06/final_run.sh writes them)


def profile_kernel_avg_ms(csv_name, kernel_prefix):
    """Average duration (ms) of a kernel in a committed rocprofv3 --kernel-trace --stats summary under profiles/ (None if the file or the kernel is
    missing): the live figure of a bench line is one box's draw of the placement lottery, the committed trace is another -- both are printed."""
    import csv
    for cand in (csv_name, csv_name.replace("r06_", "r05_")):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", cand)
        try:
            with open(path) as f:
                for row in csv.DictReader(f):
                    if kernel_prefix in row.get("Name", ""):
                        return float(row["AverageNs"]) * 1e-6, int(row["Calls"])
        except Exception:   # noqa: BLE001
            pass
    return None, 0


# FP64 flops of the matrix-free operator per element, counted in the ISA of the build (profiles/r06_operator_isa_counts.txt)
MF_FLOPS_PER_ELEMENT = {"iso": 976, "ortho": 1028}
FP64_PEAK_TFS = 78.6


def compulsory_assembly_bytes(c, n_elem, nnzb, geo_bytes=128):
    """HBM bytes the owner-computes design must move per launch: K written once (72 B per block), the element records
    (read at least once), the gather lists (4 B code + 2 B slot per contribution) and the chunk / row tables."""
    sizes = c.symbolic_sizes()
    nr, _, _ = c.matrix_info()
    return int(nnzb * 72 + n_elem * geo_bytes + sizes["n_contrib"] * 6 + sizes["n_chunk"] * 12 + nr * 4)


def cpu_direct_solve(deg, n, rtol):
    """The reference's dominant cost is the sparse direct solve (CholmodFactorizer, SparseMatrices.hh:2002-2024,2106-2124,
    settings :2243-2295: supernodal LL^T, nested dissection among the orderings). CHOLMOD / SuiteSparse is not installed in
    this image (ctypes.util.find_library finds none), so the same METHOD runs restated on LAPACK: a multifrontal Cholesky on a
    geometric nested-dissection tree (oracle/direct_solve.py; potrf / trsm / gemm on the BLAS threads of the process) on a
    bounded sample of the same workload (same generator, boundary conditions and material), next to the library's PCG at that
    size; the two displacement fields are compared."""
    import ctypes.util
    import scipy.sparse as sp
    from oracle import c_oracle as CO
    from oracle import direct_solve as DS
    import meshfem_amd as M
    from meshfem_amd import grid
    D = np.zeros((6, 6))
    lam, mu = 0.35 * 200 / (1.35 * 0.3), 200 / 2.7
    D[:3, :3] = lam
    D[np.arange(3), np.arange(3)] = lam + 2 * mu
    D[np.arange(3, 6), np.arange(3, 6)] = mu
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    c = M.Context(0)
    c.mesh_build(T, V, deg)
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.set_preconditioner(M.PRECOND_TWO_LEVEL)
    t0 = time.perf_counter()
    u_gpu = c.sim_solve(rtol=rtol)
    gpu_wall = time.perf_counter() - t0            # symbolic + assembly + coarse setup + PCG + transfers, cold context
    gi = dict(c.last_info)
    en, nn, pos, f = c.elem_nodes(), c.n_node, c.node_positions(), c.neumann_load().ravel()
    c.close()
    Ap, Ai, Ax, t = CO.assemble_csc(3, deg, en, V, D, nn)
    N = 3 * nn
    t0 = time.perf_counter()
    U = sp.csc_matrix((Ax, Ai, Ap), shape=(N, N))
    K = (U + sp.triu(U, 1).T).tocsr()
    free_nodes = np.flatnonzero(np.abs(pos[:, 0]) >= 1e-9)
    free = (3 * free_nodes[:, None] + np.arange(3)[None, :]).ravel()
    Kr = K[free][:, free]                          # SPSDSystem::fixVariables: eliminated rows / columns (:2389-2500)
    t_elim = time.perf_counter() - t0
    # the fronts one after the other on 16 BLAS threads: measured on this pool's 128-core hosts, 515 k DOF factor in 13.3 s
    # (570 GFlop/s); more OpenBLAS threads lose on fronts of this size (64: 1.5x slower), and factoring independent subtrees on
    # worker threads with single-threaded BLAS first (factor(workers=32)) ends slower too (19.9 s): the top fronts dominate
    ncpu = os.cpu_count() or 1
    workers, top_threads = 1, max(1, min(16, ncpu))
    threads = top_threads
    mf = DS.MultifrontalCholesky(Kr, pos[free_nodes], block=3)
    mf.factor(workers=workers, blas_threads=top_threads)
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=top_threads, user_api="blas"):
        x = mf.solve(f[free])
    u = np.zeros(N)
    u[free] = x
    err = float(np.linalg.norm(u_gpu.ravel() - u) / np.linalg.norm(u))
    res = float(np.linalg.norm(Kr @ x - f[free]) / np.linalg.norm(f[free]))
    have_cholmod = ctypes.util.find_library("cholmod") is not None
    return dict(kind="port of the method: multifrontal (supernodal) Cholesky, geometric nested dissection, LAPACK potrf/trsm/syrk on %d BLAS threads "
                     "(oracle/direct_solve.py); CHOLMOD %s"
                     % (top_threads, "found but not bound" if have_cholmod else "not installed on this box"),
                sample="%d^3 grid -> %d P%d tets" % (n, len(T), deg), dof=int(N), free_dof=int(len(free)), cores=threads,
                assemble_s=float(t["total"]), eliminate_s=t_elim, ordering_s=mf.t_order, factor_s=mf.t_factor, factor_subtrees_s=mf.t_subtrees, backsolve_s=mf.t_solve,
                solve_s=mf.t_order + mf.t_factor + mf.t_solve,
                time_to_solution_s=float(t["total"]) + t_elim + mf.t_order + mf.t_factor + mf.t_solve,
                factor_nnz=int(mf.factor_nnz), factor_gflops=mf.flops / 1e9, supernodes=len(mf.kids), rel_residual=res,
                gpu_pcg=dict(iterations=gi["iterations"], solve_s=gi["solve_ms"] * 1e-3, wall_s_cold_context=gpu_wall,
                             preconditioner="two-level", rel_l2_vs_direct=err, rtol=rtol))


def cpu_baseline(deg, cpu_grid):
    """Reference loop structure (threaded Ke -> serial triplet push -> sumRepeated -> CSC) in plain C
    on the host cores, on a bounded sample of the same workload (same generator, smaller grid)."""
    from oracle import c_oracle as CO
    import meshfem_amd as M
    from meshfem_amd import grid
    D = np.zeros((6, 6))
    lam, mu = 0.35 * 200 / (1.35 * 0.3), 200 / 2.7
    D[:3, :3] = lam
    D[np.arange(3), np.arange(3)] = lam + 2 * mu
    D[np.arange(3, 6), np.arange(3, 6)] = mu
    best = None
    sizes = [cpu_grid] if cpu_grid else [12, 24, 32]      # the last size that starts is reported: ~10 s of CPU work
    for n in sizes:
        V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
        h = M.Context(-1)                      # host-only: node numbering only
        h.mesh_build(T, V, deg)
        en, nn = h.elem_nodes(), h.n_node
        h.close()
        Ap, Ai, Ax, t = CO.assemble_csc(3, deg, en, V, D, nn)
        best = dict(value=len(T) / t["total"], unit="elements/s", cores=CO.num_threads(), kind="port",
                    sample="%d^3 grid -> %d P%d tets; threaded Ke %.2fs + serial triplet push %.2fs + sumRepeated/CSC %.2fs"
                           % (n, len(T), deg, t["ke"], t["push"], t["compress"]))
        last = (n, V, T, Ap, Ai, Ax)
        # the tuned host assembly beside the port (same Ke routine; no Ke array, no triplets, no sort: atomic adds into the known pattern):
        # what the GPU figure should be read against if the CPU code were free to restructure the reference's loop too
        try:
            CO.assemble_fused(3, deg, en, V, D, nn, Ap, Ai)                    # (first touch of Ax, thread start-up)
            Ax2, t2, missed = CO.assemble_fused(3, deg, en, V, D, nn, Ap, Ai)
            best["tuned"] = dict(value=len(T) / t2, unit="elements/s", cores=CO.num_threads(), seconds=t2, kind="port, restructured",
                                 max_rel_diff_vs_port=float(np.abs(Ax2 - Ax).max() / np.abs(Ax).max()), pruned_exact_zeros=int(missed),
                                 note="Ke on the thread's stack (the port's routine), binary search + atomic add into the CSC pattern of the port's result")
            del Ax2
        except Exception as e:   # noqa: BLE001
            best["tuned"] = dict(error="%s: %s" % (type(e).__name__, e))
        if t["total"] > 8.0:
            break
        del Ap, Ai, Ax
    # the matrix the timing above produced is not thrown away: the HIP path assembles the SAME sample and its
    # mfh_export_upper_triplets (== TripletMatrix::dumpBinary after sumRepeated) is compared entry by entry with the oracle's CSC
    try:
        from oracle import parity
        n, V, T, Ap, Ai, Ax = last
        c = M.Context(0)
        c.mesh_build(T, V, deg)
        c.material_isotropic(200.0, 0.35)
        c.assemble()
        i, j, v = c.export_upper_triplets()
        c.close()
        kp = parity.compare_upper_triplets_with_csc(i, j, v, Ap, Ai, Ax)
        kp["sample"] = "%d^3 grid -> %d P%d tets" % (n, len(T), deg)
        kp["ok"] = bool(kp["pattern_identical"] and kp["order_is_sumRepeated"] and kp["max_rel_err"] <= 1e-12)
        best["k_parity"] = kp
    except Exception as e:   # noqa: BLE001 -- reported inside the line, never silently dropped
        best["k_parity"] = dict(ok=False, error="%s: %s" % (type(e).__name__, e))
    return best


def cpu_baseline_with_solve(args):
    out = cpu_baseline(args.deg, args.cpu_grid)
    try:
        ds = None
        # grow the sample while the factorisation time (~ N^2) leaves room: 289 k, 515 k DOF for quadratic tets (~5 s, ~15 s on 16 threads)
        for n in ([args.cpu_solve_grid] if args.cpu_solve_grid else [14, 17]):
            ds = cpu_direct_solve(args.deg, n, args.rtol)
            if ds["solve_s"] > 8.0:      # the next size costs ~3x as much
                break
        out["direct_solve"] = ds
        out.update(solve_s=ds["solve_s"], dof=ds["dof"], solve_kind=ds["kind"])
    except Exception as e:   # noqa: BLE001 -- the baseline must not lose the GPU numbers
        out["direct_solve"] = dict(error="%s: %s" % (type(e).__name__, e))
    return out


def hbm_stream_probe(torch, n_doubles=1 << 27, reps=10):
    """Measured HBM rates of this box (peaks must be checked on the box): copy (1 read + 1 write), triad
    (2 reads + 1 write), fill (write only) and a sum reduction (read only) over 1 GiB vectors with torch's kernels; context
    for the roofline fractions, which are still quoted against the 8 TB/s spec."""
    a = torch.ones(n_doubles, dtype=torch.float64, device="cuda")
    b = torch.full((n_doubles,), 2.0, dtype=torch.float64, device="cuda")
    c = torch.empty_like(a)
    res = {}
    for name, fn, passes in (("copy", lambda: c.copy_(a), 2), ("triad", lambda: torch.add(a, b, alpha=3.0, out=c), 3),
                             ("fill", lambda: c.fill_(1.5), 1), ("read", lambda: a.sum(), 1)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[name + "_GBs"] = passes * n_doubles * 8 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b, c
    # (no torch.cuda.empty_cache(): a hipFree of gigabytes makes the next large hipMalloc of the process stall for up to seconds on this
    # driver stack -- profiles/r04_malloc_probe.txt; the 3 GiB stay in torch's allocator)
    res["note"] = "1 GiB f64 vectors, torch elementwise kernels, HIP events"
    return res


def reserve_for(M, n, deg):
    """mfh_device_reserve_for for a context on the n^3 grid, asynchronous: the library knows what such a context holds (mfh_context_bytes_estimate)
    and reserves the value array of K in a segment of its own (round 5 did the split here by hand: the placement policy lives in the arena now,
    docs/design/04_2_k_assemble_gather.md (xi), (xii))."""
    return M.device_reserve_for(3, deg, 24 * n ** 3)


def forced_2_ranks_one_gpu(args, n=48, timeout_s=240):
    """something multi-rank in the N = 1 driver line -- the n^3 cube dealt over TWO processes that share the one device
    (`bench.py --gpus 2 --ranks-per-gpu-ok --scaling strong`): the library's preflight, the transport the halo exchange landed on (peer copies
    over HIP IPC first, then the callbacks), multigrid iterations and two global figures of the solution against the SAME cube solved in one
    context of this process. It measures no scaling (two ranks time-slice one GPU); it is the RCCL -> peer -> callbacks chain and the partitioned
    hierarchy run on a box the builder never touched."""
    import subprocess
    import meshfem_amd as M
    from meshfem_amd import grid
    t0 = time.time()
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "2", "--ranks-per-gpu-ok", "--scaling", "strong", "--grid", str(n), "--no-cpu", "--no-weak",
           "--steps", "5", "--warmup", "1", "--rtol", str(args.rtol)]
    env = dict(os.environ)
    env.pop("MFH_BENCH_CHILD", None)
    env["MFH_DEVICE_SHARERS"] = "2"
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return dict(error="no answer within %d s" % timeout_s)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return dict(error="exit code %d" % p.returncode, stderr_tail=p.stderr[-600:])
    d = json.loads(lines[-1])
    mg = d.get("pcg_multigrid") if isinstance(d.get("pcg_multigrid"), dict) else {}
    res = dict(workload=d.get("config", {}).get("workload"), wall_s=time.time() - t0, devices=d.get("devices"),
               preflight=dict(transports_tried=d.get("preflight", {}).get("transports_tried"), communicator=d.get("preflight", {}).get("communicator"),
                              peer_transfers=(d.get("preflight", {}).get("peer_transfers") or {}).get("outcome")),
               dist={k: d.get("dist", {}).get(k) for k in ("communicator", "halo_transport_last_solve", "halo_bytes_per_exchange", "interior_items", "boundary_items", "peer_counters")},
               assembly=dict(value=d.get("value"), ms_per_step=d.get("ms_per_step")),
               pcg_multigrid={k: mg.get(k) for k in ("iterations", "converged", "true_rel_residual", "solve_s", "transport", "overlap", "hierarchy_setup_ms", "max_abs_u", "u_l2")},
               pcg_two_level={k: (d.get("pcg") or {}).get(k) for k in ("iterations", "converged", "solve_s", "transport")})
    # the same cube in ONE context of this process
    try:
        V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
        c = M.Context(0)
        c.mesh_build(np.ascontiguousarray(T, dtype=np.int32), V, args.deg)
        c.material_isotropic(200.0, 0.35)
        c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
        c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u = c.sim_solve(rtol=args.rtol, maxit=2000)
        one = dict(iterations=int(c.last_info["iterations"]), max_abs_u=float(np.abs(u).max()), u_l2=float(np.sqrt(np.sum(u ** 2))))
        c.close()
        res["one_context"] = one
        if mg.get("u_l2"):
            res["against_one_context"] = dict(iterations_equal=bool(one["iterations"] == mg.get("iterations")),
                                              u_l2_rel_diff=abs(mg["u_l2"] - one["u_l2"]) / one["u_l2"],
                                              max_abs_u_rel_diff=abs(mg["max_abs_u"] - one["max_abs_u"]) / one["max_abs_u"])
    except Exception as e:   # noqa: BLE001
        res["one_context"] = dict(error="%s: %s" % (type(e).__name__, e))
    return res


def one_shot_fresh_context(M, T, V, deg, rtol, maxit):
    """What a Simulate_cli caller pays end to end: a FRESH context, mesh_build -> boundary conditions -> Simulator::solve with
    the drivers' default preconditioner (symbolic phase, first assembly, multigrid hierarchy, PCG, download of u), wall clock around all of it,
    plus the library's own laps. The process is warm (HIP initialised, device memory in the library's arena): what a second mesh of a process pays."""
    c = M.Context(0)
    t0 = time.perf_counter()
    c.mesh_build(T, V, deg)
    t1 = time.perf_counter()
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    t2 = time.perf_counter()
    u = c.sim_solve(rtol=rtol, maxit=maxit)
    t3 = time.perf_counter()
    i, g, tm = dict(c.last_info), c.multigrid_info(), c.timing()
    nE = int(c.n_elem)
    res = dict(wall_s=t3 - t0, elements_per_s=nE / (t3 - t0),
               breakdown_s=dict(femmesh_build=t1 - t0, boundary_conditions=t2 - t1, solve_call=t3 - t2,
                                symbolic=tm["symbolic_ms"] * 1e-3, embedding_and_assembly=(tm["geometry_ms"] + tm["assemble_ms"]) * 1e-3,
                                hierarchy=g["setup_ms"] * 1e-3, pcg=i["solve_ms"] * 1e-3,
                                rest_of_solve_call=(t3 - t2) - (tm["symbolic_ms"] + tm["geometry_ms"] + tm["assemble_ms"] + g["setup_ms"] + i["solve_ms"]) * 1e-3),
               iterations=i["iterations"], converged=bool(i["converged"]), max_abs_u=float(np.abs(u).max()),
               note="fresh context, warm process; from mesh_build to u on the host; the timed step of the headline is %.1f %% of it" % (100.0 * (tm["geometry_ms"] + tm["assemble_ms"]) * 1e-3 / (t3 - t0)))
    c.close()
    return res


def run_single(args):
    import torch
    import meshfem_amd as M
    from meshfem_amd import grid
    n, deg = args.grid, args.deg
    strong = args.scaling == "strong"
    # configs[4]'s cube in ONE context as a one-shot caller sees it: a process of its own, run BEFORE this one touches the device
    # (inside a process that had run the other legs the same code took 4.8 s instead of 0.6 s for the first assembly)
    strong_first = None
    if not strong and not args.no_strong_n1 and deg == 2:
        strong_first = run_leg_subprocess(args, "strong_n1")
    # "reserve once", as in the strong_n1 leg: the memory of this leg (3.6 kB per quadratic tet) is asked for before anything else happens in
    # the process, on a thread of the library: what the driver has to clear first -- the process before this one returned 146 GB -- is cleared
    # while torch starts, the bandwidth probe runs and the mesh is generated (2.85 s inside the symbolic phase's hipMalloc otherwise,
    # profiles/r05_bench_n1_without_reservation_after_the_119_leg.json)
    reserve_main = 0
    if not os.environ.get("MFH_BENCH_NO_RESERVE"):
        reserve_main = reserve_for(M, n, deg)
    torch.cuda.set_device(0)
    hbm_measured = hbm_stream_probe(torch)
    if strong:      # one 40 M-element context: no room (and no point) for the storage / ordering variants next to it
        args.no_orderings = True
    t0 = time.time()
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    T = np.ascontiguousarray(T, dtype=np.int32)            # the C ABI's index type: the conversion belongs to the generator, not to the FEMMesh build
    t_gen = time.time() - t0
    c = M.Context(0)
    t0 = time.time(); c.mesh_build(T, V, deg); t_build = time.time() - t0
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    t0 = time.time(); c.symbolic(False); t_sym = time.time() - t0
    c.set_option("reembed", 1)               # every step re-runs the embedding kernel too
    nE = c.n_elem
    t0 = time.time(); c.assemble(); c.dev_sync(); t_first = time.time() - t0     # the first pass: what a one-shot caller pays on top of mesh build + symbolic phase
    for _ in range(max(0, args.warmup - 1)):
        c.assemble()
    torch.cuda.synchronize(); c.dev_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        c.assemble()
    c.dev_sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_step = dt / args.steps * 1e3
    value = nE * args.steps / dt
    # dominant kernel alone, HIP events on the context's stream
    k_ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, max(3, args.steps))
    # Storage of K: automatic (option matrix_storage -1). Elasticity solves on the matrix-free operator, nothing multiplies by the stored K,
    # and the context stores / assembles the triangle the reference's TripletMatrix holds (SURVEY 8d's "upper-only variant, matches
    # reference storage", 4 316 B per P2 tet, 872 B per P1 tet); both triangles (7 736 / 1 328 B) with option matrix_free 0.
    upper, stored_blocks = c.matrix_storage()
    bytes_per_element = ALG_BYTES_UPPER[(3, deg)] if upper else ALG_BYTES[(3, deg)]
    alg = bytes_per_element * nE
    tb, tsrc = pmc_traffic(PMC_KERNEL_KEY, n, deg, PMC_PROFILE_UPPER if upper else PMC_PROFILE)
    nr, nc, nnzb = c.matrix_info()
    comp = compulsory_assembly_bytes(c, nE, stored_blocks)
    roof = dict(bound="hbm", kernel="k_assemble_gather", achieved=alg / k_ms / 1e6, peak=HBM_PEAK_GBS, unit="GB/s",
                frac=alg / k_ms / 1e6 / HBM_PEAK_GBS, traffic=None if tb is None else tb / k_ms / 1e6,
                # the contract fraction above counts SURVEY 8(d)'s 100 Ke blocks per element, which an owner-computes kernel
                # never moves; the two fractions below are on bytes that DO cross the HBM interface
                frac_traffic=None if tb is None else tb / k_ms / 1e6 / HBM_PEAK_GBS,
                compulsory_bytes=comp, frac_compulsory=comp / k_ms / 1e6 / HBM_PEAK_GBS,
                traffic_bytes_per_launch_from_profile=tb, traffic_from_profile=tsrc, alg_bytes_per_launch=alg, kernel_ms=k_ms,
                kernel_trace="profiles/r06_config2_kernel_stats.csv: rocprofv3 --kernel-trace --stats of `python bench.py --leg config2` (this workload alone: "
                             "k_assemble_gather<3, 2, 0, true, false>, no deterministic and no 119^3 launches in the file)",
                bytes_per_element=bytes_per_element,
                matrix_storage="upper triangle (blocks (r, c >= r): what the reference assembles)" if upper else "both triangles",
                note="frac = SURVEY 8(d) algorithmic bytes (the row of the storage in use) / time / 8 TB/s (contract); frac_traffic = "
                     "rocprofv3 PMC bytes of the committed profile / time; frac_compulsory = bytes the design must move (stored K "
                     "once + records + lists) / time")
    # the committed trace's average of the same kernel (another box, another draw of where the driver put the K values): both on the line
    if (n, deg) == (60, 2) and upper:
        pavg, pcalls = profile_kernel_avg_ms(PROFILE_ROUND + "_config2_kernel_stats.csv", "k_assemble_gather<3, 2, 0, true, false>")
        if pavg:
            roof.update(kernel_ms_profile_avg=pavg, profile_launches=pcalls, frac_profile_avg=alg / pavg / 1e6 / HBM_PEAK_GBS,
                        kernel_ms_spread="2.88-3.30 ms by process and box -- where the driver puts the K values (profiles/r06_config2_profile_spread.txt: five profiled "
                                         "processes per box; the profile average and the process's own HIP events agree to 0.9 % inside every process)",
                        frac_traffic_profile_avg=None if tb is None else tb / pavg / 1e6 / HBM_PEAK_GBS,
                        frac_compulsory_profile_avg=comp / pavg / 1e6 / HBM_PEAK_GBS)
    # context: the kernel's MEASURED traffic rate against the triad rate measured on this box a minute ago
    if roof["traffic"] is not None:
        roof["measured_triad_GBs"] = hbm_measured["triad_GBs"]
        roof["traffic_frac_of_measured_triad"] = roof["traffic"] / hbm_measured["triad_GBs"]
    out = dict(metric="stiffness_assembly_elements_per_s", value=value, unit="elements/s", n_gpus=1, steps=args.steps,
               warmup=args.warmup, ms_per_step=ms_step, higher_is_better=True, scaling=args.scaling or "strong", vs_baseline=None,
               dtype="f64", data="synthetic",
               config=dict(workload="%s: %d^3 grid -> %d P%d tets, isotropic E=200 nu=0.35, Dirichlet x=0, traction x=1"
                                    % (("configs[4] on one GPU" if n == 119 else "the %d^3 cube of the strong-scaling run on one GPU" % n) + " (the N = 1 point of the strong-scaling curve)"
                                       if strong else ("configs[2]" if (n, deg) == (60, 2) else "configs[2] generator at another size"), n, nE, deg),
                           elements=nE, nodes=c.n_node, dof=3 * c.n_dof, nnz_blocks=nnzb,
                           stored_blocks=stored_blocks, matrix_storage="upper" if upper else "full",
                           parallelism="1 GPU", step="embed + Ke blocks + assembly into block-CSR (gather/owner-computes)"),
               roofline=roof, hbm_measured=hbm_measured,
               setup=dict(mesh_gen_s=t_gen, femmesh_build_s=t_build, symbolic_s=t_sym, first_assemble_call_s=t_first,
                          # the timed step excludes what a one-shot caller pays once per mesh -- FEMMesh build (edge numbering,
                          # boundary extraction, node table; includes the upload), symbolic phase (pattern + gather lists), first pass
                          first_assembly_ms=(t_build + t_sym + t_first) * 1e3, first_assembly_elements_per_s=nE / (t_build + t_sym + t_first),
                          reservation="mfh_device_reserve(%.1f GB in two segments, asynchronous) at process start" % (reserve_main / 1e9) if reserve_main else "none",
                          **c.timing(), **c.symbolic_sizes()))
    # comparison variant: element-major global-atomic scatter (north_star: "colored or atomic ... by evidence")
    try:
        a_ms = c.time_assembly_kernel(M.ASSEMBLE_ATOMIC, 3)
        out["variants"] = dict(atomic_scatter=dict(kernel_ms=a_ms, elements_per_s=nE / a_ms * 1e3, alg_GBs=alg / a_ms / 1e6))
        c.assemble()
    except M.MeshFEMHipError as e:
        out["variants"] = dict(atomic_scatter=str(e))
    # option placement_trials (off by default): the same workload on a second context that auditions three more buffers for the K values at its
    # first assembly -- what a caller that assembles hundreds of times can buy, and what it costs the first assembly
    # (docs/design/04_2_k_assemble_gather.md (xi): the kernel's time follows where the driver put the values)
    if not strong and not args.no_orderings:
        try:
            cp = M.Context(0)
            cp.set_option("placement_trials", 3)
            t0 = time.time(); cp.mesh_build(T, V, deg); tbp = time.time() - t0
            cp.material_isotropic(200.0, 0.35)
            t0 = time.time(); cp.symbolic(False); ts = time.time() - t0
            cp.set_option("reembed", 1)
            t0 = time.time(); cp.assemble(); cp.dev_sync(); tf = time.time() - t0
            for _ in range(args.warmup):
                cp.assemble()
            cp.dev_sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                cp.assemble()
            cp.dev_sync()
            dtp = (time.perf_counter() - t0) / args.steps
            kp = cp.time_assembly_kernel(M.ASSEMBLE_GATHER, max(3, args.steps))
            out["variants"]["placement_trials_3"] = dict(kernel_ms=kp, ms_per_step=dtp * 1e3, elements_per_s=nE / dtp, candidates_kernel_ms=cp.placement_info(),
                                                         first_assembly_ms=(tbp + ts + tf) * 1e3, first_assemble_call_ms=tf * 1e3,
                                                         frac_traffic=None if tb is None else tb / kp / 1e6 / HBM_PEAK_GBS,
                                                         note="option placement_trials 3 (library default 0): the headline above is the default")
            cp.close()
        except Exception as e:   # noqa: BLE001 -- never lose the line over a variant
            out["variants"]["placement_trials_3"] = dict(error="%s: %s" % (type(e).__name__, e))
    if not args.no_solve and strong:
        # the N = 1 point of the strong-scaling curve: the solver the N > 1 runs use (two-level PCG), time to rtol
        c.set_preconditioner(M.PRECOND_TWO_LEVEL)
        t0 = time.time()
        u = c.sim_solve(rtol=args.rtol, maxit=args.maxit)
        i2, p2 = dict(c.last_info), c.precond_info()
        ndof = 3 * c.n_dof
        out["pcg"] = dict(iterations=i2["iterations"], converged=bool(i2["converged"]), rtol=args.rtol, rel_residual=i2["rel_residual"],
                          true_rel_residual=i2["true_rel_residual"], solve_s=i2["solve_ms"] * 1e-3, dof=ndof,
                          dof_per_s=ndof * i2["iterations"] / (i2["solve_ms"] * 1e-3), ms_per_iteration=i2["solve_ms"] / max(1, i2["iterations"]),
                          preconditioner="two-level: 3x3 block-Jacobi + rigid-body modes of %d aggregates" % p2["aggregates"],
                          coarse_setup_ms=p2["setup_ms"], coarse_dim=p2["coarse_dim"], wall_s=time.time() - t0, max_abs_u=float(np.abs(u).max()),
                          operator="matrix-free (k_mf_cluster + k_mf_rows)", ranks=1)
        # ... and the multigrid V-cycle, which the N > 1 runs report as pcg_multigrid too (partitioned nodal levels there)
        try:
            c.set_preconditioner(M.PRECOND_MULTIGRID)
            t0 = time.time()
            u3 = c.sim_solve(rtol=args.rtol, maxit=min(args.maxit, 2000))
            i3, p3, g3 = dict(c.last_info), c.precond_info(), c.multigrid_info()
            out["pcg_multigrid"] = dict(iterations=i3["iterations"], converged=bool(i3["converged"]), true_rel_residual=i3["true_rel_residual"],
                                        solve_s=i3["solve_ms"] * 1e-3, ms_per_iteration=i3["solve_ms"] / max(1, i3["iterations"]),
                                        hierarchy_setup_ms=g3["setup_ms"], wall_s_with_setup=time.time() - t0, aggregates=p3["aggregates"],
                                        dense_level_dim=p3["coarse_dim"], note=p3["note"],
                                        speedup_solve_vs_two_level=i2["solve_ms"] / max(i3["solve_ms"], 1e-30),
                                        rel_l2_vs_two_level=float(np.linalg.norm(u3 - u) / np.linalg.norm(u)))
            del u3
        except M.MeshFEMHipError as e:
            out["pcg_multigrid"] = str(e)
    elif not args.no_solve:
        t0 = time.time()
        u = c.sim_solve(rtol=args.rtol, maxit=args.maxit)
        info = dict(c.last_info)
        ndof = 3 * c.n_dof
        c.set_option("matrix_free", 0)
        sp_ms = c.time_spmv_kernel(20)           # assembled block-CSR SpMV (k_spmv)
        c.set_option("matrix_free", 1)
        mf_ms = c.time_spmv_kernel(20)           # matrix-free operator (k_mf_forces + k_mf_rows), the PCG default for P2
        mf_info = c.matrix_free_info()
        c.set_option("matrix_free", -1)
        npe = 10 if deg == 2 else 4
        # compulsory bytes of the matrix-free operator (cluster variant): element record + connectivity + local row
        # indices per element; x gather + y + x (dot) per node; interface partials written and read back + their lists
        # (constant material, round 2: the gradients are recomputed from the corner positions, so the 128-B record is not read;
        # the vertex positions are counted once)
        mf_bytes = (nE * (4 * npe + 2 * npe) + len(V) * 24 + c.n_dof * 3 * 24 + mf_info["block_rows"] * 8
                    + mf_info["interface_partials"] * (2 * 24 + 2))
        t1, _ = pmc_traffic("k_mf_cluster", n, deg)
        t2, _ = pmc_traffic("k_mf_rows", n, deg)
        mf_traffic = None if t1 is None or t2 is None else t1 + t2
        sp_bytes = nnzb * 76 + nr * 3 * 16 + nr * 4
        it_bytes = sp_bytes + ndof * 112
        stb, stsrc = pmc_traffic("k_spmv", n, deg)
        out["pcg"] = dict(iterations=info["iterations"], converged=info["converged"], rtol=args.rtol,
                          rel_residual=info["rel_residual"], true_rel_residual=info["true_rel_residual"],
                          solve_ms=info["solve_ms"], dof=ndof,
                          dof_per_s=ndof * info["iterations"] / (info["solve_ms"] * 1e-3),
                          ms_per_iteration=info["solve_ms"] / max(1, info["iterations"]),
                          preconditioner="3x3 block-Jacobi (north_star baseline preconditioner)", wall_s=time.time() - t0,
                          operator="matrix-free (k_mf_cluster + k_mf_rows)",
                          algorithm="classic PCG (two reduction points; the default for one right-hand side on one GPU)",
                          matrix_free=dict(kernels_ms=mf_ms, speedup_vs_assembled_spmv=sp_ms / mf_ms, bytes_per_application=mf_bytes, lists=mf_info,
                                           **(dict(fp64_flops_per_element=MF_FLOPS_PER_ELEMENT["iso"], fp64_TFs=MF_FLOPS_PER_ELEMENT["iso"] * nE / mf_ms * 1e-9,
                                                   fp64_frac_of_peak=MF_FLOPS_PER_ELEMENT["iso"] * nE / mf_ms * 1e-9 / FP64_PEAK_TFS,
                                                   flop_per_byte=MF_FLOPS_PER_ELEMENT["iso"] * nE / mf_bytes, ridge_flop_per_byte=FP64_PEAK_TFS * 1e3 / HBM_PEAK_GBS,
                                                   fp64_note="flops counted in the ISA of this build (profiles/r06_operator_isa_counts.txt): the operator is on the "
                                                             "bandwidth side of the ridge") if deg == 2 else {}),
                                           traffic_bytes_per_application=mf_traffic, traffic=None if mf_traffic is None else mf_traffic / mf_ms / 1e6,
                                           achieved=mf_bytes / mf_ms / 1e6, frac=mf_bytes / mf_ms / 1e6 / HBM_PEAK_GBS, unit="GB/s",
                                           note="same operator as the assembled K to rounding; trades 72 B/block of matrix traffic for FP64 flops"),
                          max_abs_u=float(np.abs(u).max()),
                          roofline=dict(bound="hbm", kernel="k_spmv", note="assembled block-CSR SpMV (option matrix_free 0; the solves use the matrix-free operator below)",
                                        achieved=sp_bytes / sp_ms / 1e6, peak=HBM_PEAK_GBS,
                                        unit="GB/s", frac=sp_bytes / sp_ms / 1e6 / HBM_PEAK_GBS, kernel_ms=sp_ms,
                                        traffic=None if stb is None else stb / sp_ms / 1e6, traffic_bytes_per_launch=stb,
                                        traffic_source=stsrc,
                                        alg_bytes_per_launch=sp_bytes,
                                        iteration_achieved=it_bytes / (info["solve_ms"] / max(1, info["iterations"])) / 1e6))
        # the loop the multi-GPU solve runs (Chronopoulos-Gear: one reduction point), here at world size 1
        try:
            c.set_option("pcg_variant", 1)
            c.sim_solve(rtol=args.rtol, maxit=args.maxit)
            icg = dict(c.last_info)
            out["pcg"]["chronopoulos_gear"] = dict(iterations=icg["iterations"], solve_ms=icg["solve_ms"],
                                                   ms_per_iteration=icg["solve_ms"] / max(1, icg["iterations"]),
                                                   true_rel_residual=icg["true_rel_residual"],
                                                   note="mfh_dist_solve's loop (one fused reduction per iteration) on the unpartitioned mesh")
        except M.MeshFEMHipError as e:
            out["pcg"]["chronopoulos_gear"] = str(e)
        c.set_option("pcg_variant", -1)
        # same system with the two-level preconditioner (block-Jacobi + rigid-body-mode coarse space)
        try:
            c.set_preconditioner(M.PRECOND_TWO_LEVEL)
            t0 = time.time()
            u2 = c.sim_solve(rtol=args.rtol, maxit=args.maxit)
            i2, p2 = dict(c.last_info), c.precond_info()
            out["pcg_two_level"] = dict(iterations=i2["iterations"], converged=i2["converged"], true_rel_residual=i2["true_rel_residual"],
                                        solve_ms=i2["solve_ms"], coarse_setup_ms=p2["setup_ms"], aggregates=p2["aggregates"],
                                        coarse_dim=p2["coarse_dim"], note=p2["note"], wall_s=time.time() - t0,
                                        ms_per_iteration=i2["solve_ms"] / max(1, i2["iterations"]),
                                        dof_per_s=ndof * i2["iterations"] / (i2["solve_ms"] * 1e-3),
                                        speedup_time_to_solution=info["solve_ms"] / (i2["solve_ms"] + p2["setup_ms"]),
                                        rel_l2_vs_block_jacobi=float(np.linalg.norm(u2 - u) / np.linalg.norm(u)))
        except M.MeshFEMHipError as e:
            out["pcg_two_level"] = str(e)
        # same system with the p-multigrid preconditioner (quadratic level -> linear level -> rigid-body coarse level)
        if deg == 2:
            try:
                c.set_preconditioner(M.PRECOND_MULTIGRID)
                t0 = time.time()
                u3 = c.sim_solve(rtol=args.rtol, maxit=args.maxit)
                i3, p3, g3 = dict(c.last_info), c.precond_info(), c.multigrid_info()
                out["pcg_multigrid"] = dict(iterations=i3["iterations"], converged=i3["converged"], true_rel_residual=i3["true_rel_residual"],
                                            solve_ms=i3["solve_ms"], hierarchy_setup_ms=g3["setup_ms"], levels=dict(quadratic_dof=3 * g3["fine_dof"], linear_dof=3 * g3["coarse_dof"],
                                                                                                            rigid_body_dim=p3["coarse_dim"]),
                                            lambda_max=[g3["lambda_max_fine"], g3["lambda_max_coarse"]], note=p3["note"], wall_s=time.time() - t0,
                                            ms_per_iteration=i3["solve_ms"] / max(1, i3["iterations"]),
                                            speedup_time_to_solution_vs_block_jacobi=info["solve_ms"] / (i3["solve_ms"] + g3["setup_ms"]),
                                            rel_l2_vs_block_jacobi=float(np.linalg.norm(u3 - u) / np.linalg.norm(u)),
                                            coarse_storage="FP32 copies of the linear level's K and of the aggregate stencils inside the preconditioner "
                                                           "(option mg_coarse_fp32; products and sums FP64)")
                # the same solve with the preconditioner's matrices kept in FP64 (what the option buys; the hierarchy is rebuilt for it and back)
                c.set_option("mg_coarse_fp32", 0)
                u4 = c.sim_solve(rtol=args.rtol, maxit=args.maxit)
                i4 = dict(c.last_info)
                out["pcg_multigrid"]["fp64_coarse_storage"] = dict(iterations=i4["iterations"], solve_ms=i4["solve_ms"],
                                                                   ms_per_iteration=i4["solve_ms"] / max(1, i4["iterations"]),
                                                                   rel_l2_vs_default=float(np.linalg.norm(u4 - u3) / np.linalg.norm(u3)))
                c.set_option("mg_coarse_fp32", 1)
            except M.MeshFEMHipError as e:
                out["pcg_multigrid"] = str(e) if not isinstance(out.get("pcg_multigrid"), dict) else out["pcg_multigrid"]
    # Option "deterministic" (bit-reproducible assembly, operator and PCG): the same timed step and the block-Jacobi PCG
    # with the waves adding in order and the dot products through the fixed tree, next to the default figures of this line
    if not strong and isinstance(out.get("variants"), dict):
        try:
            c.set_option("deterministic", 1)
            c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
            for _ in range(3):
                c.assemble()
            c.dev_sync()
            # every pass timed on its own (a pass ends with a stream synchronisation anyway), the MEDIAN reported: late in the line's sequence one
            # pass in twenty or so takes tens of milliseconds (a host-side one-off, not the kernels: the event-timed kernel stays put), which a
            # mean over 20 passes turns into "2x"
            ts = []
            for _ in range(args.steps):
                t0 = time.perf_counter(); c.assemble(); c.dev_sync(); ts.append(time.perf_counter() - t0)
            dtd = float(np.median(ts))
            kd = c.time_assembly_kernel(M.ASSEMBLE_GATHER, max(3, args.steps))
            det = dict(ms_per_step=dtd * 1e3, ms_per_step_max=max(ts) * 1e3, elements_per_s=nE / dtd, kernel_ms=kd, step_cost_vs_default=dtd * 1e3 / ms_step, kernel_cost_vs_default=kd / k_ms,
                       pass_timers=c.timing(), storage="upper" if c.matrix_storage()[0] else "full",
                       note="option deterministic 1: two runs give identical bits (tests/test_gpu_deterministic.py)")
            if not args.no_solve and "pcg" in out:
                ud = c.sim_solve(rtol=args.rtol, maxit=args.maxit)
                idt = dict(c.last_info)
                det["pcg_block_jacobi"] = dict(iterations=idt["iterations"], converged=bool(idt["converged"]), solve_ms=idt["solve_ms"],
                                               ms_per_iteration=idt["solve_ms"] / max(1, idt["iterations"]),
                                               cost_per_iteration_vs_default=(idt["solve_ms"] / max(1, idt["iterations"])) / out["pcg"]["ms_per_iteration"],
                                               rel_l2_vs_default=float(np.linalg.norm(ud - u) / np.linalg.norm(u)))
                del ud
            out["variants"]["deterministic"] = det
        except M.MeshFEMHipError as e:
            out["variants"]["deterministic"] = str(e)
        finally:
            c.set_option("deterministic", 0)
    # The same pass with BOTH triangles of K stored (option matrix_storage 0: the round-1 definition of this benchmark, and what a
    # context does on its own when something multiplies by the stored K). Same timed region (embedding + blocks + assembly).
    if upper and not strong:
        try:
            c.close()
            cu = M.Context(0)
            cu.set_option("matrix_storage", 0)
            cu.mesh_build(T, V, deg)
            cu.material_isotropic(200.0, 0.35)
            cu.symbolic(False)
            cu.set_option("reembed", 1)
            for _ in range(args.warmup):
                cu.assemble()
            cu.dev_sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                cu.assemble()
            cu.dev_sync()
            dtu = (time.perf_counter() - t0) / args.steps
            ku = cu.time_assembly_kernel(M.ASSEMBLE_GATHER, max(3, args.steps))
            tbf, tsf = pmc_traffic(PMC_KERNEL_KEY, n, deg, PMC_PROFILE)
            out["variants"]["both_triangles_storage"] = dict(
                ms_per_step=dtu * 1e3, elements_per_s=nE / dtu, kernel_ms=ku, stored_blocks=int(cu.matrix_storage()[1]),
                K_bytes=int(cu.matrix_storage()[1]) * 72, bytes_per_element=ALG_BYTES[(3, deg)],
                frac=ALG_BYTES[(3, deg)] * nE / ku / 1e6 / HBM_PEAK_GBS, frac_traffic=None if tbf is None else tbf / ku / 1e6 / HBM_PEAK_GBS,
                traffic_from_profile=tsf,
                note="option matrix_storage 0; SURVEY 8(d) 7 736 B per P2 tet")
            cu.close()
        except M.MeshFEMHipError as e:
            out["variants"]["both_triangles_storage"] = str(e)
    if not args.no_orderings:
        # the same mesh with shuffled / space-filling-curve numbering (gather locality)
        c.close()
        out["variants"]["orderings"] = {"generator": dict(assembly_kernel_ms=k_ms, spmv_kernel_ms=out.get("pcg", {}).get("roofline", {}).get("kernel_ms"),
                                                          matrix_free_kernels_ms=out.get("pcg", {}).get("matrix_free", {}).get("kernels_ms"))}
        for mode in ("morton", "shuffle"):
            try:
                V2, T2 = grid.reorder_mesh(V, T, mode)
                c2 = M.Context(0)
                c2.mesh_build(T2, V2, deg)
                c2.material_isotropic(200.0, 0.35)
                c2.assemble()
                ak = c2.time_assembly_kernel(M.ASSEMBLE_GATHER, 5)
                c2.set_option("matrix_free", 0)
                sk = c2.time_spmv_kernel(10)
                c2.set_option("matrix_free", 1)
                out["variants"]["orderings"][mode] = dict(assembly_kernel_ms=ak, spmv_kernel_ms=sk, matrix_free_kernels_ms=c2.time_spmv_kernel(10))
                c2.close()
                del V2, T2
            except M.MeshFEMHipError as e:
                out["variants"]["orderings"][mode] = str(e)
    if not strong and (n, deg) == (60, 2) and isinstance(out.get("variants"), dict):
        # BASELINE configs[1] (35^3 grid -> 1,029,000 LINEAR tets, the same boundary-value problem) in every driver line: the assembly step, the
        # assembled block-CSR SpMV (the PCG operator of linear meshes) against its own algorithmic bytes, and the multigrid PCG to rtol
        try:
            c.close()
        except Exception:   # noqa: BLE001
            pass
        try:
            out["variants"]["config1_p1"] = config1_p1(args)
        except Exception as e:   # noqa: BLE001
            out["variants"]["config1_p1"] = dict(error="%s: %s" % (type(e).__name__, e))
    if not strong and (n, deg) == (60, 2) and not args.no_config3 and not args.no_solve and isinstance(out.get("variants"), dict):
        # BASELINE configs[3] in every driver line
        try:
            out["variants"]["config3_homogenization"] = config3_homogenization(args)
        except Exception as e:   # noqa: BLE001
            out["variants"]["config3_homogenization"] = dict(error="%s: %s" % (type(e).__name__, e))
    if not strong and (n, deg) == (60, 2) and not args.no_solve:
        # the one-shot sum of configs[2]: a fresh context of the same mesh, from mesh_build to u on the host
        try:
            c.close()
        except Exception:   # noqa: BLE001
            pass
        try:
            out["one_shot"] = dict(workload=out["config"]["workload"], **one_shot_fresh_context(M, T, V, deg, args.rtol, min(args.maxit, 2000)))
        except Exception as e:   # noqa: BLE001
            out["one_shot"] = dict(error="%s: %s" % (type(e).__name__, e))
    if not strong and (n, deg) == (60, 2) and not args.no_solve and not args.no_forced_ranks and isinstance(out.get("variants"), dict):
        try:
            out["variants"]["forced_2_ranks_one_gpu"] = forced_2_ranks_one_gpu(args)
        except Exception as e:   # noqa: BLE001
            out["variants"]["forced_2_ranks_one_gpu"] = dict(error="%s: %s" % (type(e).__name__, e))
    if not strong and not args.no_strong_n1 and deg == 2:
        # the N = 1 point of the strong-scaling curve the N > 1 runs measure by default (configs[4]'s 119^3 cube in ONE context), so that
        # one driver pass over N = 1, 2, 4, 8 holds the whole curve
        try:
            c.close()
        except Exception:   # noqa: BLE001 -- already closed by the variants above
            pass
        out["strong_scaling_n1"] = strong_first if strong_first is not None else dict(error="leg not run")
        try:      # the same leg INSIDE this process, after everything above: what the device arena makes of a process that held other meshes before
            out["strong_scaling_n1"]["warm_process"] = strong_n1(args, torch)
        except Exception as e:   # noqa: BLE001 -- the line above is complete
            out["strong_scaling_n1"]["warm_process"] = dict(error="%s: %s" % (type(e).__name__, e))
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_with_solve(args)
    print(json.dumps(out), flush=True)


def config1_p1(args, n=35):
    import meshfem_amd as M
    from meshfem_amd import grid
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    T = np.ascontiguousarray(T, dtype=np.int32)
    c = M.Context(0)
    t0 = time.time(); c.mesh_build(T, V, 1); t_build = time.time() - t0
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    t0 = time.time(); c.symbolic(False); t_sym = time.time() - t0
    c.set_option("reembed", 1)
    t0 = time.time(); c.assemble(); c.dev_sync(); t_first = time.time() - t0
    for _ in range(args.warmup):
        c.assemble()
    c.dev_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        c.assemble()
    c.dev_sync()
    dt = (time.perf_counter() - t0) / args.steps
    nE = c.n_elem
    k_ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, max(3, args.steps))
    nr, nc, nnzb = c.matrix_info()
    upper, stored = c.matrix_storage()
    bpe = ALG_BYTES_UPPER[(3, 1)] if upper else ALG_BYTES[(3, 1)]
    alg = bpe * nE
    comp = compulsory_assembly_bytes(c, nE, stored)
    res = dict(workload="configs[1]: %d^3 grid -> %d P1 tets, %d DOF" % (n, nE, 3 * c.n_dof), elements=nE, dof=3 * c.n_dof, nnz_blocks=nnzb, stored_blocks=stored,
               matrix_storage="upper" if upper else "full", value=nE / dt, unit="elements/s", ms_per_step=dt * 1e3,
               roofline=dict(kernel="k_assemble_gather", kernel_ms=k_ms, bytes_per_element=bpe, frac=alg / k_ms / 1e6 / HBM_PEAK_GBS,
                             compulsory_bytes=comp, frac_compulsory=comp / k_ms / 1e6 / HBM_PEAK_GBS,
                             note="%.0f MB per launch: the working set is of the order of the 256 MiB memory-side cache; the fractions are against the "
                                  "8 TB/s HBM peak all the same" % (comp / 1e6)),
               setup=dict(femmesh_build_s=t_build, symbolic_s=t_sym, first_assemble_call_s=t_first, first_assembly_ms=(t_build + t_sym + t_first) * 1e3),
               kernel_trace="profiles/r06_config1_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py --leg config1`)")
    if not args.no_solve:
        op_ms = c.time_spmv_kernel(20)
        mf_info = c.matrix_free_info()
        res["operator"] = dict(kernels="k_mf_cluster + k_mf_rows (matrix-free, the PCG default of elasticity since round 5)", kernels_ms=op_ms, lists=mf_info)
        for name, pre in (("pcg_block_jacobi", M.PRECOND_BLOCK_JACOBI), ("pcg_multigrid", M.PRECOND_MULTIGRID)):
            c.set_preconditioner(pre)
            t0 = time.time()
            u = c.sim_solve(rtol=args.rtol, maxit=args.maxit)
            i = dict(c.last_info)
            res[name] = dict(iterations=i["iterations"], converged=bool(i["converged"]), true_rel_residual=i["true_rel_residual"], solve_ms=i["solve_ms"],
                             ms_per_iteration=i["solve_ms"] / max(1, i["iterations"]), wall_s=time.time() - t0, max_abs_u=float(np.abs(u).max()),
                             dof_per_s=3 * c.n_dof * i["iterations"] / (i["solve_ms"] * 1e-3))
        res["pcg_multigrid"]["hierarchy_setup_ms"] = c.multigrid_info()["setup_ms"]
        # the assembled SpMV of rounds 1-4 beside it (needs both triangles: the option re-assembles)
        c.set_option("matrix_free", 0)
        sp_ms = c.time_spmv_kernel(20)
        sp_bytes = nnzb * 76 + nr * 3 * 16 + nr * 4
        res["spmv"] = dict(kernel="k_spmv", kernel_ms=sp_ms, alg_bytes_per_launch=sp_bytes, achieved=sp_bytes / sp_ms / 1e6, frac=sp_bytes / sp_ms / 1e6 / HBM_PEAK_GBS,
                           unit="GB/s", note="both triangles x 76 B + vectors; not what the solves above ran on")
        res["operator"]["speedup_vs_assembled_spmv"] = sp_ms / op_ms
    c.close()
    return res


def config3_homogenization(args, n=44):
    """BASELINE configs[3]: periodic homogenization of a 44^3 grid -> 2,044,416 P2 tets with a per-element orthotropic field (SURVEY 8d
    ranges, numpy default_rng(0)), six cell problems (PeriodicHomogenization.hh:34-54 solveCellProblems, :72-100 Ch): FEMMesh build +
    periodic DoF map + assembly + multigrid hierarchy + 6 PCG solves + Ch, each phase timed, and Ch checked in-line against the
    size-independent properties (major symmetry, positive definiteness, Reuss <= Ch <= Voigt on the diagonal)."""
    import meshfem_amd as M
    from meshfem_amd import grid, homogenization as H
    from meshfem_amd.linear_elasticity import Simulator
    t_all = time.time()
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    T = np.ascontiguousarray(T, dtype=np.int32)
    P = grid.synthetic_orthotropic_field(len(T), 3, 0)
    t_gen = time.time() - t_all
    t0 = time.time(); sim = Simulator(T, V, 2, 0); t_build = time.time() - t0
    sim.rtol = args.rtol
    sim.ctx.set_preconditioner(M.PRECOND_MULTIGRID)
    t0 = time.time(); sim.setOrthotropicField(P); t_mat = time.time() - t0
    t0 = time.time(); w, infos = H.solve_cell_problems(sim); t_cell = time.time() - t0
    t0 = time.time(); Ch = H.homogenized_elasticity_tensor(sim, w); t_ch = time.time() - t0
    c = sim.ctx
    nE, dof = int(c.n_elem), 3 * int(c.n_dof)
    its = [int(i["iterations"]) for i in infos]
    solve_ms = [float(i["solve_ms"]) for i in infos]
    # right-hand sides solved in one batch report the batch's device time each: count every batch once
    batch_sizes = [max(1, int(i.get("reserved", 1))) for i in infos]
    device_ms, k = 0.0, 0
    while k < len(infos):
        device_ms += solve_ms[k]
        k += batch_sizes[k]
    g = c.multigrid_info()
    tm = c.timing()
    # bounds of the field on a sample of the elements (equal volumes): Voigt = mean stiffness, Reuss = inverse of the mean compliance
    D = np.stack([c.material_get(e) for e in range(0, nE, 997)])
    voigt, reuss = D.mean(axis=0), np.linalg.inv(np.linalg.inv(D).mean(axis=0))
    sym_err = float(np.abs(Ch - Ch.T).max() / np.abs(Ch).max())
    min_eig = float(np.linalg.eigvalsh(0.5 * (Ch + Ch.T)).min())
    in_bounds = bool(all(reuss[i, i] * 0.97 <= Ch[i, i] <= voigt[i, i] * 1.03 for i in range(6)))
    res = dict(workload="configs[3]: %d^3 periodic cell -> %d P2 tets, per-element orthotropic field (seed 0), 6 cell problems, multigrid PCG to %g" % (n, nE, args.rtol),
               elements=nE, dof=dof, nodes=int(c.n_node),
               wall_s=dict(mesh_gen=t_gen, femmesh_build=t_build, material_field=t_mat,
                           cell_problems=t_cell, Ch=t_ch, total_without_mesh_gen=t_build + t_mat + t_cell + t_ch),
               cell_problems=dict(iterations=its, solve_ms=solve_ms, converged=[bool(i["converged"]) for i in infos],
                                  true_rel_residual=[float(i["true_rel_residual"]) for i in infos],
                                  batch_sizes=batch_sizes, device_ms_all_solves=device_ms,
                                  dof_per_s=dof * sum(its) / (device_ms * 1e-3), hierarchy_setup_ms=g["setup_ms"],
                                  symbolic_ms=tm["symbolic_ms"], assemble_ms=tm["assemble_ms"]),
               elements_per_s_end_to_end=nE / (t_build + t_mat + t_cell + t_ch),
               Ch_diag=[float(Ch[i, i]) for i in range(6)], voigt_diag=[float(voigt[i, i]) for i in range(6)], reuss_diag=[float(reuss[i, i]) for i in range(6)],
               checks=dict(major_symmetry_rel_err=sym_err, min_eigenvalue=min_eig, reuss_le_Ch_le_voigt_on_diagonal=in_bounds,
                           all_converged=bool(all(i["converged"] for i in infos)),
                           passed=bool(sym_err <= 1e-9 and min_eig > 0 and in_bounds and all(i["converged"] for i in infos))),
               trace="profiles/r06_config3_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py --leg config3`)")
    c.close()
    return res


def run_leg_subprocess(args, leg, timeout_s=900, extra=()):
    """One leg of the line in a FRESH process (a one-shot caller of that size is one): returns the leg's JSON object."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", leg, "--steps", str(args.steps), "--warmup", str(args.warmup), "--rtol", str(args.rtol),
           "--maxit", str(args.maxit), "--deg", str(args.deg)] + (["--no-solve"] if (args.no_solve or "--no-solve" in extra) else [])
    t0 = time.time()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return dict(error="leg %s: no result after %d s" % (leg, timeout_s))
    for line in reversed(p.stdout.strip().splitlines()):
        if line.startswith("{"):
            try:
                res = json.loads(line)
                res["process"] = "fresh process (python bench.py --leg %s), %.1f s of wall time with interpreter start-up and mesh generation" % (leg, time.time() - t0)
                return res
            except ValueError:
                pass
    return dict(error="leg %s: exit code %d, no JSON; stderr tail: %s" % (leg, p.returncode, p.stderr[-400:]))


def config2_trace_leg(args, n=60):
    """configs[2] alone, for a per-workload kernel trace: the timed assembly passes, one multigrid solve to rtol, 300
    block-Jacobi PCG iterations -- no deterministic launches, no other mesh in the process."""
    import meshfem_amd as M
    from meshfem_amd import grid
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    T = np.ascontiguousarray(T, dtype=np.int32)
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.symbolic(False)
    c.set_option("reembed", 1)
    for _ in range(args.warmup):
        c.assemble()
    c.dev_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        c.assemble()
    c.dev_sync()
    dt = (time.perf_counter() - t0) / args.steps
    k_ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, max(3, args.steps))
    res = dict(workload="configs[2]: %d^3 grid -> %d P2 tets" % (n, c.n_elem), elements=int(c.n_elem), ms_per_step=dt * 1e3, value=c.n_elem / dt, kernel_ms=k_ms)
    if not args.no_solve:
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u = c.sim_solve(rtol=args.rtol, maxit=500)
        res["pcg_multigrid"] = dict(iterations=c.last_info["iterations"], solve_ms=c.last_info["solve_ms"], hierarchy_setup_ms=c.multigrid_info()["setup_ms"],
                                    max_abs_u=float(np.abs(u).max()))
        c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
        try:
            c.sim_solve(rtol=1e-30, maxit=300)
        except M.MeshFEMHipError:
            pass
        res["pcg_block_jacobi_300"] = dict(ms_per_iteration=c.last_info["solve_ms"] / 300)
        res["operator_ms"] = c.time_spmv_kernel(50)
    c.close()
    return res


def run_leg(args):
    if args.leg == "config2" and not os.environ.get("MFH_BENCH_NO_RESERVE"):
        # as the line itself does, and in the same place: FIRST in the process, before torch touches the device (the order of events in the
        # process is one of the inputs of the placement lottery, profiles/r06_config2_profile_spread.txt: the trace should draw as the line draws)
        import meshfem_amd as M
        reserve_for(M, 60, 2)
    import torch
    torch.cuda.set_device(0)
    if args.leg == "config2":
        res = config2_trace_leg(args)
    elif args.leg == "strong_n1":
        res = strong_n1(args, torch)
    elif args.leg == "config3":
        res = config3_homogenization(args)
    elif args.leg == "config1":
        res = config1_p1(args)
    else:
        raise SystemExit("unknown leg " + args.leg)
    print(json.dumps(res), flush=True)


def strong_n1(args, torch, n=119):
    """BASELINE configs[4]'s cube on one GPU: the same timed step as the line itself (embedding + blocks + assembly), the one-shot cost
    of the first assembly, and the time to solution of the multigrid PCG -- what `bench.py --gpus N` (N > 1) reports for the cube dealt
    out over N ranks."""
    import gc
    import meshfem_amd as M
    from meshfem_amd import grid
    gc.collect(); torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info(0)
    if free < 200e9:
        return dict(skipped="needs 200 GB of free device memory, %.0f GB free" % (free / 1e9))
    deg = args.deg
    # "reserve once" (LinearElasticity.hh:1441-1443): the caller knows its mesh size before it has the mesh -- 3.6 kB of device memory per
    # quadratic tet cover assembly + multigrid solve -- and asks for it FIRST, asynchronously: the driver hands the memory out (and, on a
    # box nobody has used since boot, clears it: 2-3 s, profiles/r05_large_allocation_trace_119.log) while the mesh is being generated
    t_res = time.time()
    reserve = reserve_for(M, n, deg) if not os.environ.get("MFH_BENCH_NO_RESERVE") else 0
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    T = np.ascontiguousarray(T, dtype=np.int32)
    c = M.Context(0)
    t0 = time.time(); c.mesh_build(T, V, deg); t_build = time.time() - t0
    del V, T
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    t0 = time.time(); c.symbolic(False); t_sym = time.time() - t0
    c.set_option("reembed", 1)
    nE = c.n_elem
    t0 = time.time(); c.assemble(); c.dev_sync(); t_first = time.time() - t0
    steps = max(3, min(args.steps, 5))
    c.assemble(); c.dev_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        c.assemble()
    c.dev_sync()
    dt = (time.perf_counter() - t0) / steps
    k_ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, 3)
    res = dict(workload="configs[4]: %d^3 grid -> %d P%d tets in ONE context" % (n, nE, deg), elements=nE, dof=3 * c.n_dof,
               value=nE / dt, unit="elements/s", ms_per_step=dt * 1e3, steps=steps, kernel_ms=k_ms,
               kernel_trace="profiles/r06_strong_n1_kernel_stats.csv + r06_strong_n1_kernel_trace_summary.txt (per launch size: the aggregate levels apart)",
               setup=dict(femmesh_build_s=t_build, symbolic_s=t_sym, first_assemble_call_s=t_first,
                          first_assembly_ms=(t_build + t_sym + t_first) * 1e3, first_assembly_elements_per_s=nE / (t_build + t_sym + t_first),
                          device_memory_reserved_GB=reserve / 1e9,
                          reservation="mfh_device_reserve(%.1f GB in two segments, asynchronous) issued before the mesh generation (outside the timed phases, like process "
                                      "start-up); MFH_BENCH_NO_RESERVE=1 runs without" % (reserve / 1e9) if reserve else "none"))
    if not args.no_solve:
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        t0 = time.time()
        u = c.sim_solve(rtol=args.rtol, maxit=min(args.maxit, 2000))
        i3, g3 = dict(c.last_info), c.multigrid_info()
        res["pcg_multigrid"] = dict(iterations=i3["iterations"], converged=bool(i3["converged"]), true_rel_residual=i3["true_rel_residual"],
                                    solve_s=i3["solve_ms"] * 1e-3, ms_per_iteration=i3["solve_ms"] / max(1, i3["iterations"]),
                                    hierarchy_setup_ms=g3["setup_ms"], wall_s_with_setup=time.time() - t0, max_abs_u=float(np.abs(u).max()))
        # the one-shot sum of configs[4]'s cube in one context: the phases above ran once each, one after the other, on this
        # context -- mesh build, symbolic phase, first assembly, then Simulator::solve (hierarchy + PCG + download)
        w = t_build + t_sym + t_first + res["pcg_multigrid"]["wall_s_with_setup"]
        res["one_shot"] = dict(wall_s=w, elements_per_s=nE / w,
                               breakdown_s=dict(femmesh_build=t_build, symbolic=t_sym, first_assemble_call=t_first, hierarchy=g3["setup_ms"] * 1e-3,
                                                pcg=i3["solve_ms"] * 1e-3,
                                                rest_of_solve_call=res["pcg_multigrid"]["wall_s_with_setup"] - (g3["setup_ms"] + i3["solve_ms"]) * 1e-3),
                               note="sum of the consecutive phases of this context, each run once (between them: the %d timed assembly passes of the step)" % (steps + 1))
    free2, _ = torch.cuda.mem_get_info(0)
    res["memory"] = dict(device_used_GB=(total - free2) / 1e9, arena={k: (v / 1e9 if k.endswith("bytes") else v) for k, v in M.device_arena_stats(0).items()})
    c.close()
    return res


def run_multi(args):
    import torch
    import torch.distributed as dist
    from meshfem_amd import distributed as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", rank))
    ndev = torch.cuda.device_count()
    shared = ndev < world                       # fewer GPUs than ranks (a 1-GPU box): ranks share devices, gloo transport
    if shared and not (args.ranks_per_gpu_ok or os.environ.get("MFH_BENCH_SHARE_GPUS")):
        raise SystemExit("bench.py --gpus %d needs %d GPUs, %d visible (pass --ranks-per-gpu-ok to share them)" % (world, world, ndev))
    device = local % max(1, ndev)
    if shared:          # the ranks of a device bound their arenas to their share of it (mfh_pool.cpp)
        os.environ.setdefault("MFH_DEVICE_SHARERS", str((world + max(1, ndev) - 1) // max(1, ndev)))
    torch.cuda.set_device(device)
    if shared:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    res = D.bench_multi(args, rank, world, device, shared_gpus=shared)
    dist.barrier()
    dist.destroy_process_group()
    # RCCL writes its version banner through C stdio, which is flushed late when stdout is a pipe:
    # flush it first so that the JSON line is the last line of the output
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(res), flush=True)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver's command line
    would (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MFH_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    a = parse()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("MFH_BENCH_FORCE_DISTRIBUTED"):
        run_multi(a)
    elif a.leg:
        run_leg(a)
    elif a.gpus > 1:
        self_launch(a)
    else:
        run_single(a)
