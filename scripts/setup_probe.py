"""Cost of the first assembly (what a one-shot caller pays before the steady-state kernels): mesh build laps (MFH_MESH_TIMING),
symbolic phase, uploads, first numeric assembly.     python scripts/setup_probe.py [grid] [degree]"""
import os
import sys
import time

os.environ.setdefault("MFH_MESH_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = T.astype("int32")      # the ABI's index type (the generator returns int64: converting 20 M indices in numpy would be 20-30 ms of the first lap)
for rep in range(int(os.environ.get("REPS", "2"))):
    c = M.Context(0)
    t0 = time.perf_counter()
    c.mesh_build(T, V, deg)
    t1 = time.perf_counter()
    c.material_isotropic(200.0, 0.35)
    c.symbolic(False)
    t2 = time.perf_counter()
    c.assemble(); c.dev_sync()
    t3 = time.perf_counter()
    c.assemble(); c.dev_sync()
    t4 = time.perf_counter()
    print("rep %d: %d elements: mesh_build %.1f ms, symbolic %.1f ms, first assemble %.1f ms, second assemble %.2f ms; %s"
          % (rep, len(T), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), c.timing()), flush=True)
    c.close()
