"""Per-kernel cost of option "deterministic" (run under rocprofv3 --kernel-trace --stats): block-Jacobi PCG iterations at configs[2]
in the default mode and in the deterministic mode.    python scripts/det_probe.py [grid] [det 0|1] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
det = int(sys.argv[2]) if len(sys.argv) > 2 else 1
its = int(sys.argv[3]) if len(sys.argv) > 3 else 300
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.set_option("deterministic", det)
c.mesh_build(T.astype("int32"), V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
c.set_option("reembed", 1)
import time
for _ in range(5):
    c.assemble()
c.dev_sync()
t0 = time.perf_counter()
for _ in range(20):
    c.assemble()
c.dev_sync()
print("det %d: assembly pass %.3f ms" % (det, (time.perf_counter() - t0) / 20 * 1e3))
if det:
    c.set_option("deterministic", 0)
    c.assemble(); c.assemble(); c.dev_sync()
    t0 = time.perf_counter()
    for _ in range(20):
        c.assemble()
    c.dev_sync()
    print("det %d -> 0 on the same context: assembly pass %.3f ms" % (det, (time.perf_counter() - t0) / 20 * 1e3))
    c.set_option("deterministic", 1)
    c.assemble(); c.assemble(); c.dev_sync()
    t0 = time.perf_counter()
    for _ in range(20):
        c.assemble()
    c.dev_sync()
    print("det 0 -> 1 on the same context: assembly pass %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
try:
    c.sim_solve(rtol=1e-30, maxit=its)
except M.MeshFEMHipError:
    pass
i = c.last_info
print("det %d: %d iterations, %.3f ms per iteration, assembly kernel %.3f ms" % (det, i["iterations"], i["solve_ms"] / max(1, i["iterations"]), c.time_assembly_kernel(M.ASSEMBLE_GATHER, 5)))
