"""Quick on-GPU timing probe of the assembly variants and the SpMV (not part of the test-suite)."""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["g0", "g1", "atomic"]
t = time.time(); V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1]); tgen = time.time() - t
res = dict(n=n, deg=deg, elems=len(T), gen_s=tgen)
for var in variants:
    c = M.Context(0)
    if var.startswith("g"):
        c.set_option("contrib_order", int(var[1]))
        if len(var) > 2:
            c.set_option("chunk_slots", int(var[3:]))
    t = time.time(); c.mesh_build(T, V, deg); tb = time.time() - t
    c.material_isotropic(200, 0.35)
    mode = M.ASSEMBLE_ATOMIC if var == "atomic" else M.ASSEMBLE_GATHER
    t = time.time(); c.symbolic(mode == M.ASSEMBLE_ATOMIC); ts = time.time() - t
    ms = c.time_assembly_kernel(mode, 5)
    nr, nc, nnzb = c.matrix_info()
    bytes_alg = (7736 if deg == 2 else 1328) * len(T)
    r = dict(build_s=tb, symbolic_s=ts, asm_ms=ms, elem_per_s=len(T) / ms * 1e3, alg_GBs=bytes_alg / ms / 1e6,
             nnzb=nnzb, nodes=c.n_node, **c.symbolic_sizes(), geometry_ms=c.timing()["geometry_ms"])
    if var == variants[0]:
        sp_ms = c.time_spmv_kernel(10)
        r.update(spmv_ms=sp_ms, spmv_GBs=(nnzb * 76 + nr * 3 * 16) / sp_ms / 1e6)
    res[var] = r
    print(var, json.dumps(r), flush=True)
    c.close()
