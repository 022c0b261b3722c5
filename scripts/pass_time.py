import sys
sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid
V, T = grid.grid_tet_mesh(60, 60, 60, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.symbolic(False)
c.set_option("reembed", 1)
import time
for _ in range(5): c.assemble()
c.dev_sync(); t0 = time.perf_counter()
for _ in range(40): c.assemble()
c.dev_sync(); print("pass %.3f ms" % ((time.perf_counter() - t0) / 40 * 1e3), c.timing())
print("kernel %.3f" % c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20))
