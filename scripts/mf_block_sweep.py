"""Sweep of the cluster block size (elements per workgroup) of the matrix-free operator at the bench size."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import meshfem_amd as M
from meshfem_amd import grid
V, T = grid.grid_tet_mesh(60, 60, 60, [0, 0, 0], [1, 1, 1])
for be in [int(a) for a in sys.argv[1:]] or (512, 480, 384, 320, 256, 192, 128):
    c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(200., 0.35)
    c.set_option("mf_block_elems", be); c.set_option("matrix_free", 1); c.assemble()
    ms = c.time_spmv_kernel(10)
    print(be, "ms %.4f" % ms, c.matrix_free_info(), flush=True)
    c.close()
