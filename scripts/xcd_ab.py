"""XCD-grouped work mapping (runs of G chunks / blocks per XCD): assembly kernel and matrix-free operator times.
    python scripts/xcd_ab.py [grid]"""
import sys

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.symbolic(False)
c.assemble()
for G in (0, 2, 4, 8, 16, 32, 64, 0):
    c.set_option("xcd_swizzle", G)
    c.assemble()
    a = c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20)
    o = c.time_spmv_kernel(50)
    print("G %3d: assembly %.3f ms, matrix-free operator %.4f ms" % (G, a, o), flush=True)
