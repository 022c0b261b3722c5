"""y = K x from the stored upper triangle (k_spmv_sym: the transposed half through global FP64 atomics) against k_spmv on both
triangles, same mesh, same process. usage: spmv_sym_probe.py [grid=100] [degree=1] [order=reference|morton]
order morton (round 5): the rows numbered along the Morton curve of the node positions through a bijective mfh_dof_map -- the order
in which the transposed half of a row chunk would have the best chance to stay near the chunk."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
order = sys.argv[3] if len(sys.argv) > 3 else "reference"
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = T.astype(np.int32)
perm = None
if order == "morton":
    def part1by2(v):
        v = v.astype(np.uint64) & np.uint64(0x1fffff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    c0 = M.Context(0); c0.mesh_build(T, V, deg); P = c0.node_positions(); c0.close()
    q = np.minimum((P * n * 4).astype(np.int64), 4 * n)
    key = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << np.uint64(1)) | (part1by2(q[:, 2]) << np.uint64(2))
    perm = np.empty(len(P), dtype=np.int32)
    perm[np.argsort(key, kind="stable")] = np.arange(len(P), dtype=np.int32)
print("row order:", order)
res = {}
x = None
for storage in (0, 1, 0, 1):
    c = M.Context(0)
    c.set_option("matrix_storage", storage)
    c.set_option("matrix_free", 0)
    c.mesh_build(T, V, deg)
    if perm is not None:
        c.dof_map(perm, len(perm))
    c.material_isotropic(200.0, 0.35)
    c.assemble()
    if x is None:
        x = np.random.default_rng(0).standard_normal(3 * c.n_dof)
    y = c.apply_K(x)
    ms = [c.time_spmv_kernel(20) for _ in range(3)]
    nr, nc, nnzb = c.matrix_info()
    stored = c.matrix_storage()[1]
    res.setdefault(storage, []).append(min(ms))
    if storage == 0:
        y0 = y
    else:
        print("max |y_sym - y_full| / max |y| = %.2e" % (np.abs(y - y0).max() / np.abs(y0).max()))
    print("grid %d deg %d storage %s: %d rows, %d stored blocks (%.2f GB), SpMV %.3f / %.3f / %.3f ms -> %.2f TB/s on the stored bytes" %
          (n, deg, "upper (k_spmv_sym)" if storage else "both (k_spmv)", nr, stored, stored * 76 / 1e9, *ms, stored * 76 / min(ms) / 1e9))
    c.close()
print("ratio sym / full: %.2f" % (min(res[1]) / min(res[0])))
