"""Timing of the gather assembly kernel (HIP events, 20 launches) for build / option variants.
    python scripts/asm_ab.py [grid] [deg] [chunk_slots]"""
import sys

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
if len(sys.argv) > 3:
    c.set_option("chunk_slots", int(sys.argv[3]))
c.mesh_build(T, V, deg)
c.material_isotropic(200.0, 0.35)
c.symbolic(False)
for rep in range(3):
    c.assemble()
    print("assembly kernel: %.3f ms" % c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20), flush=True)
