"""Same-process A/B of the gather assembly kernel (HIP events, 20 launches per sample, alternating) over the values of
one option that does not change the symbolic phase.
    python scripts/asm_ab.py [grid] [deg] [option] [values ...]      default: xcd_swizzle 0 8"""
import sys

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
opt = sys.argv[3] if len(sys.argv) > 3 else "xcd_swizzle"
vals = [int(v) for v in sys.argv[4:]] or [0, 8]
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, deg)
c.material_isotropic(200.0, 0.35)
c.symbolic(False)
c.assemble()
res = {v: [] for v in vals}
for rep in range(4):
    for v in vals:
        c.set_option(opt, v)
        c.assemble()
        res[v].append(c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20))
for v in vals:
    print("%s = %d: %s ms (min %.3f)" % (opt, v, " ".join("%.3f" % t for t in res[v]), min(res[v])), flush=True)
