"""Assembly kernel time against the chunk size (a symbolic-phase option: one context per value).
    python scripts/asm_chunks.py [grid] [deg] [chunk_slots ...]"""
import sys

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sizes = [int(v) for v in sys.argv[3:]] or [128, 192, 256, 320, 384, 512]
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
for cs in sizes:
    c = M.Context(0)
    c.set_option("chunk_slots", cs)
    c.mesh_build(T, V, deg)
    c.material_isotropic(200.0, 0.35)
    c.symbolic(False)
    c.assemble()
    t = [c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20) for _ in range(3)]
    sz = c.symbolic_sizes()
    print("chunk_slots %4d: %s ms   chunks %d, contributions per chunk %.0f" % (cs, " ".join("%.3f" % x for x in t), sz["n_chunk"], sz["n_contrib"] / sz["n_chunk"]),
          flush=True)
    c.close()
