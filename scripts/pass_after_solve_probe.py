"""Why does an assembly pass take twice as long after solves (bench variants.deterministic: 7.6 ms against 3.7 ms with a 3.7 ms kernel)?
Times 20 passes at configs[2] after each stage of the bench's sequence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T.astype("int32"), V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.set_option("reembed", 1)


def passes(tag):
    c.assemble(); c.assemble(); c.dev_sync()
    t0 = time.perf_counter()
    for _ in range(20):
        c.assemble()
    c.dev_sync()
    print("%-52s pass %.3f ms   timers %s" % (tag, (time.perf_counter() - t0) / 20 * 1e3, {k: round(v, 3) for k, v in c.timing().items()}), flush=True)


passes("fresh context")
if len(sys.argv) > 2:
    import torch
    a = torch.ones(1 << 27, dtype=torch.float64, device="cuda"); b = a * 2; del a, b
    passes("after torch allocated and released 2 GiB (kept in its allocator)")
    c.time_assembly_kernel(M.ASSEMBLE_GATHER, 5)
    passes("after time_assembly_kernel(gather)")
    c.time_assembly_kernel(M.ASSEMBLE_ATOMIC, 3); c.assemble()
    passes("after the atomic-scatter variant (scatter map built)")
    c.set_option("pcg_variant", 1)
    try:
        c.sim_solve(rtol=1e-30, maxit=50)
    except M.MeshFEMHipError:
        pass
    c.set_option("pcg_variant", -1)
    passes("after a Chronopoulos-Gear solve")
try:
    c.sim_solve(rtol=1e-30, maxit=100)
except M.MeshFEMHipError:
    pass
passes("after a block-Jacobi solve")
c.set_option("matrix_free", 0); c.time_spmv_kernel(3); c.set_option("matrix_free", 1); c.time_spmv_kernel(3); c.set_option("matrix_free", -1)
passes("after the operator timings (storage switched twice)")
c.set_preconditioner(M.PRECOND_TWO_LEVEL); c.sim_solve(rtol=1e-8)
passes("after a two-level solve")
c.set_preconditioner(M.PRECOND_MULTIGRID); c.sim_solve(rtol=1e-8)
passes("after a multigrid solve")
c.set_option("deterministic", 1); c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
passes("deterministic on")
c.set_option("deterministic", 0)
passes("deterministic off again")
