"""Tile geometry of K's value array against the placement penalty of the assembly kernel: the kernel of configs[2] timed with tiles of
32 / 64 / 128 slots and with 128 / 2 176 bytes of padding behind every 4 608-byte tile, on plain and on physically contiguous memory
(MFH_ARENA_ALLOC). Timing-only builds (scripts/r06/build_tile_variants.sh); every sample a process of its own.
    python scripts/r06/tile_probe.py            (child: python scripts/r06/tile_probe.py child)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import meshfem_amd as M
    from meshfem_amd import grid
    n = 60
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200.0, 0.35)
    c.symbolic(False)
    c.assemble()
    t = [c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20) for _ in range(3)]
    print("kernel_ms %s" % " ".join("%.3f" % v for v in t), flush=True)
    sys.exit(0)

variants = [("tile 64 (default)", None), ("tile 32", "t32"), ("tile 128", "t128"), ("tile 64 + 128 B pad", "pad16"), ("tile 64 + 2176 B pad", "pad272")]
for rep in range(2):
    for kind in ("plain", "contiguous"):
        for name, tag in variants:
            env = dict(os.environ, MFH_ARENA_ALLOC=kind)
            if tag:
                env["MESHFEM_HIP_LIB"] = os.path.join(ROOT, "meshfem_amd", "variants", "libmeshfem_hip_%s.so" % tag)
            try:
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=300)
                line = [l for l in out.stdout.splitlines() if l.startswith("kernel_ms")]
                print("%-11s %-24s %s" % (kind, name, line[0] if line else "FAILED " + out.stderr[-300:].replace("\n", " | ")), flush=True)
            except subprocess.TimeoutExpired:
                print("%-11s %-24s TIMEOUT" % (kind, name), flush=True)
