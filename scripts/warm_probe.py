"""Is the spread of the assembly kernel's time between processes a matter of how long the GPU has been busy (clock / power state)? Back-to-back samples
of time_assembly_kernel in one process, with and without idle gaps.     python scripts/warm_probe.py [grid]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(np.ascontiguousarray(T, dtype=np.int32), V, 2)
c.material_isotropic(200.0, 0.35)
c.symbolic(False)
c.assemble(); c.dev_sync()
print("back to back, 20 launches per sample:", " ".join("%.3f" % c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20) for _ in range(12)), flush=True)
print("back to back, 100 launches per sample:", " ".join("%.3f" % c.time_assembly_kernel(M.ASSEMBLE_GATHER, 100) for _ in range(6)), flush=True)
for gap in (0.2, 1.0, 3.0):
    out = []
    for _ in range(4):
        time.sleep(gap)
        out.append(c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20))
    print("after %.1f s of idle, 20 launches:" % gap, " ".join("%.3f" % x for x in out), flush=True)
c.set_option("reembed", 1)
for _ in range(60):
    c.assemble()
c.dev_sync()
print("after 60 full passes (reembed 1):", " ".join("%.3f" % c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20) for _ in range(4)), flush=True)
c.set_option("reembed", 0)
for _ in range(60):
    c.assemble()
c.dev_sync()
print("after 60 full passes (reembed 0):", " ".join("%.3f" % c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20) for _ in range(4)), flush=True)
