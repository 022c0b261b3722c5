/* meshfem_hip_extras.h -- entry points of libmeshfem_hip.so that are NOT part of the drop-in boundary (include/meshfem_hip.h holds
 * that: SURVEY.md section 8b): measurement hooks of bench.py / scripts/, test hooks, and the device-pointer building blocks for
 * callers that write their own (distributed) solver loop around the library's kernels. Plain C like the main header. */
#ifndef MESHFEM_HIP_EXTRAS_H
#define MESHFEM_HIP_EXTRAS_H
#include "meshfem_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- measurement */
/* average device time (ms, HIP events on the context stream) of `reps` back-to-back launches of
 * the numeric assembly kernel alone (geometry kernel excluded) -- used by bench.py's roofline   */
mfh_status mfh_time_assembly_kernel(mfh_ctx* ctx, int32_t mode, int32_t reps, double* avg_ms);
/* the same for one application of the operator the PCG uses (see "matrix_free") on internal scratch vectors */
mfh_status mfh_time_spmv_kernel(mfh_ctx* ctx, int32_t reps, double* avg_ms);
/* matrix-free operator in use? (see option "matrix_free"); for the cluster variant (mode 4): number of element blocks,
 * of (block, row) accumulators, of interface partial sums kept in HBM, and the largest block (LDS accumulators) */
mfh_status mfh_matrix_free_info(mfh_ctx* ctx, int32_t* active, int32_t* mode, int64_t* nBlocks, int64_t* nBlockRows,
                                int64_t* nInterface, int32_t* maxBlockRows);

/* the aggregate levels of the multigrid hierarchy (MFH_PRECOND_MULTIGRID), finest first: out7[l] = {aggregates of the level, rows this rank
 * smooths, entries of its vectors (rows + halo), 1 if the level is partitioned over the ranks (row-partitioned contexts: levels with more
 * than option mg_replicate_max aggregates) else 0 (replicated / unpartitioned context), exchange peers, halo aggregates received and owned
 * aggregates sent per exchange}; *nLevels = levels of the hierarchy (cap = rows of out7) */
mfh_status mfh_multigrid_level_info(const mfh_ctx* ctx, int32_t cap, int64_t* out7, int32_t* nLevels);

/* ---------------------------------------------------------------- test hooks */
/* test hook: in-place inverse of a dense SPD matrix (row-major n x n) with the threaded blocked
 * Cholesky that inverts the two-level preconditioner's coarse operator; MFH_ERR_INVALID if not SPD */
mfh_status mfh_debug_spd_inverse(int64_t n, double* A);
/* the same with the device implementation (blocked 64x64 Cholesky inverse in HBM) */
mfh_status mfh_debug_spd_inverse_device(mfh_ctx* ctx, int64_t n, double* A);
/* test hook: the DEVICE copies of the node table (nElem x nodesPerElem) and of the node positions (nNode x dim) that the kernels
 * read -- mfh_mesh_build writes them on the device from the uploaded vertices; they must equal what mfh_mesh_elem_nodes /
 * mfh_mesh_node_positions return from the host tables (FEMMesh.inl:17-59), bit for bit */
mfh_status mfh_debug_device_node_tables(mfh_ctx* ctx, int32_t* elemNodes, double* nodePos);
/* test hooks: one allocation / release through the library's device arena (meshfem_hip.h "Device memory"), as the library's own buffers
 * make them; tests/test_gpu_arena.py checks splitting, merging, the bounds and the trims with them */
mfh_status mfh_debug_arena_alloc(mfh_ctx* ctx, int64_t bytes, void** out);
mfh_status mfh_debug_arena_free(mfh_ctx* ctx, void* p);
/* test hook (host only, no context): the row chunks of the assembly kernel -- greedy, whole rows, at most chunkSlots slots each, a chunk ends at
 * every row listed in breaks -- scanned by `threads` host threads over ranges of `grain` rows and stitched (threads = 1: the plain sequential
 * scan the result must equal). Writes the chunks' first rows + nRows to chunkRow (capacity cap); *nOut = entries (chunks + 1) */
mfh_status mfh_debug_row_chunks(int64_t nRows, const int32_t* rowPtr, int32_t chunkSlots, int64_t nBreaks, const int64_t* breaks, int64_t grain,
                                int32_t threads, int32_t* chunkRow, int64_t cap, int64_t* nOut);


/* ---------------------------------------------------------------- device-pointer building blocks
 * (multi-GPU driver: local kernels here, RCCL halo exchange / all-reduce in between)           */
/* y[0:dim*nOwnedDoF] = K x ; x has dim*nColDoF entries (owned then halo). fixed-variable mask NOT applied */
mfh_status mfh_dev_spmv(mfh_ctx* ctx, const double* x_dev, double* y_dev);
/* z = M^-1 r on the owned rows (block-Jacobi of the assembled K, fixed variables decoupled)     */
mfh_status mfh_dev_precond(mfh_ctx* ctx, const double* r_dev, double* z_dev);
/* Two-level preconditioner with CALLER-supplied aggregates (row-partitioned contexts; the aggregates are global,
 * the caller reduces over ranks): begin() takes, for every local node (owned then halo), its aggregate id in
 * [0,nAgg) and relPos = (position - aggregate centre)/H (3 doubles per node, z = 0 in 2D), and writes this
 * rank's Galerkin contribution Z^T K_ownedRows Z (m x m row-major, m = nAgg * (dim==3 ? 6 : 3)) into Ac_dev.
 * The fixed-variable mask must cover halo nodes as well. After summing Ac_dev over ranks, finish() inverts it on
 * the device. restrict: rc[m] = Z_owned^T r (to be summed over ranks); apply: z = D^-1 r + Z_owned (A_c^-1 rc). */
mfh_status mfh_tl_partitioned_begin(mfh_ctx* ctx, int32_t nAgg, const int32_t* aggOfNode, const double* relPos, double* Ac_dev);
mfh_status mfh_tl_partitioned_finish(mfh_ctx* ctx, const double* Ac_dev);
mfh_status mfh_dev_tl_restrict(mfh_ctx* ctx, const double* r_dev, double* rc_dev);
mfh_status mfh_dev_tl_apply(mfh_ctx* ctx, const double* r_dev, const double* rc_dev, double* z_dev);
/* fused vector updates of the distributed PCG over the owned rows; scalars are read from DEVICE memory (they are
 * results of all-reduces): x += (num/den) p, r -= (num/den) Ap ;  p = z + (num/den) p ;  out2 = {r.z, r.r} */
mfh_status mfh_dev_pcg_update_xr(mfh_ctx* ctx, const double* num_dev, const double* den_dev, const double* p_dev, const double* Ap_dev,
                                 double* x_dev, double* r_dev);
mfh_status mfh_dev_pcg_direction(mfh_ctx* ctx, const double* num_dev, const double* den_dev, const double* z_dev, double* p_dev);
mfh_status mfh_dev_dots(mfh_ctx* ctx, const double* r_dev, const double* z_dev, double* out2_dev);
/* r[fixed] = 0 */
mfh_status mfh_dev_mask_fixed(mfh_ctx* ctx, double* r_dev);
/* copy the fixed-variable values into u (u[fixed] = value) */
mfh_status mfh_dev_set_fixed_values(mfh_ctx* ctx, double* u_dev);
mfh_status mfh_dev_sync(mfh_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MESHFEM_HIP_EXTRAS_H */
