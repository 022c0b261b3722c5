"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one counter per run): a few launches of the
assembly kernel, the assembled SpMV and the matrix-free operator on a mesh whose K exceeds the 256 MiB Infinity
Cache, plus a calibration launch with a known byte count in the SAME 8-byte-per-lane access pattern
(k_axpby over n doubles: reads 8n bytes, writes 8n bytes)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
slab = sys.argv[2] if len(sys.argv) > 2 else ""        # "slab:WORLD:RANK": that rank's share of bench.py --gpus WORLD (grid n per rank)
c = M.Context(0)
if slab:
    from meshfem_amd import distributed as D
    _, world, rank = slab.split(":")
    world, rank = int(world), int(rank)
    ng = int(round(n * world ** (1.0 / 3.0)))
    layers = max(1, int(round(n ** 3 / float(ng * ng))))
    lm = D.slab_local_mesh(ng, rank, world, 2, layers, device=0)
    c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned)
    c.material_isotropic(200, 0.35)
    c.symbolic(False)
    c.set_option("reembed", 1)
    ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, 3)
    upper, stored = c.matrix_storage()      # automatic storage: upper triangle of the owned rows for quadratic elasticity
    sp = c.time_spmv_kernel(3)      # k_axpby calibration launch + the operator on the local columns
    nr, nc, nnzb = c.matrix_info()
    nE = len(lm.elem_nodes)
    # SURVEY 8(d) per quadratic tet: 7 736 B with both triangles, 4 316 B with the upper triangle (the row of the storage in use)
    print(json.dumps(dict(n=n, slab=slab, storage="upper" if upper else "full", stored_blocks=stored, global_grid=[ng, ng, layers * world], elems=nE, nnzb=nnzb, rows=nr, asm_ms=ms, spmv_ms=sp,
                          calib_axpby_doubles=3 * nc, asm_alg_bytes=(4316 if upper else 7736) * nE,
                          asm_expected_hbm_bytes=stored * 72 + nE * (55 if upper else 100) * 6 + nE * 128)))
    sys.exit(0)
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
if os.environ.get("PMC_UPPER_STORAGE") == "1":      # the upper-triangle storage variant: assembly and matrix-free operator only
    c.set_option("matrix_storage", 1)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200, 0.35)
    c.symbolic(False)
    ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, 3)
    mf = c.time_spmv_kernel(3)      # k_axpby calibration launch, then the matrix-free operator
    nr, nc, nnzb = c.matrix_info()
    upper, stored = c.matrix_storage()
    # SURVEY 8(d), upper-only row: 4 316 B per quadratic tet; expected HBM bytes: the stored blocks once + gather codes + element records
    print(json.dumps(dict(n=n, storage="upper", elems=len(T), nnzb=nnzb, stored_blocks=stored, rows=nr, asm_ms=ms, mf_ms=mf, calib_axpby_doubles=3 * nc,
                          asm_alg_bytes=4316 * len(T), asm_expected_hbm_bytes=stored * 72 + len(T) * 55 * 6 + len(T) * 128)))
    sys.exit(0)
c.set_option("matrix_storage", 0)       # both triangles throughout (the assembled SpMV below needs them); PMC_UPPER_STORAGE=1 for the other
c.mesh_build(T, V, 2)
c.material_isotropic(200, 0.35)
c.symbolic(False)
ms = c.time_assembly_kernel(M.ASSEMBLE_GATHER, 3)
c.set_option("matrix_free", 0)
sp = c.time_spmv_kernel(3)          # launches k_axpby(n=3*nNode, b=0) once, then 1+3 k_spmv
c.set_option("matrix_free", 1)
mf = c.time_spmv_kernel(3)          # k_axpby again, then 1+3 x (k_mf_cluster, k_mf_rows over the interface partials)
nr, nc, nnzb = c.matrix_info()
nE, npe = len(T), 10
print(json.dumps(dict(n=n, elems=nE, nnzb=nnzb, rows=nr, asm_ms=ms, spmv_ms=sp, mf_ms=mf,
                      calib_axpby_doubles=3 * nc, asm_alg_bytes=7736 * nE, spmv_alg_bytes=nnzb * 76 + nr * 3 * 16 + nr * 4,
                      asm_expected_hbm_bytes=nnzb * 72 + nE * 100 * 6 + nE * 128,
                      mf_lists=c.matrix_free_info())))
