// C ABI of libmeshfem_hip.so (include/meshfem_hip.h). Host orchestration only: every numeric
// loop of the hot path runs in mfh_kernels.hip / mfh_kernels_solver.hip.
#include "mfh_ctx.hh"
#include <dlfcn.h>

using namespace mfh;

// ---- optional roctx ranges (see mfh_internal.hh)
namespace {
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        const char *en = getenv("MFH_ROCTX");
        if (!en || en[0] == '0') return;
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "/opt/rocm/lib/libroctx64.so"}) {
            if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
RoctxApi &roctx_api() { static RoctxApi a; return a; }
}   // namespace
mfh::RoctxRange::RoctxRange(const char *name) {
    RoctxApi &a = roctx_api();
    if (a.push) { a.push(name); active = true; }
}
mfh::RoctxRange::~RoctxRange() {
    if (active) roctx_api().pop();
}


namespace mfhi {
using namespace mfh;

void pack_gather_codes(mfh_ctx *c);

void invalidate_matrix(mfh_ctx *c) {
    c->dVals32.release();
    c->assembled = false;
    c->dinvValid = false;
    c->tl.valid = false;
    c->mg.valid = false;
}
void invalidate_symbolic(mfh_ctx *c) {
    ++c->listsGen;                  // overlap lists of a partitioned solve (Dist::opList) are stale from here on
    c->symValid = false;
    c->mfValid = false;
    c->mfcValid = false;
    c->mfClusterUnfit = false;
    c->dofUploaded = false;
    invalidate_matrix(c);
}
// v = n copies of value, written by all host threads (these per-node tables are hundreds of MB at the benchmark sizes: a
// single-threaded assign was a third of the mesh upload)
template <class V, class T>
static void parallel_assign(V &v, size_t n, T value) {
    v.resize(n);
    auto *p = v.data();
    parallel_ranges((int64_t)n, [&](int64_t b, int64_t e, int) { for (int64_t k = b; k < e; ++k) p[k] = value; });
}

// The Dirichlet tables (mask + value per BOUNDARY node and component, in the order of mesh.bdryNodes: only boundary nodes take Dirichlet
// conditions, BoundaryNode::setDirichlet, LinearElasticity.hh:390-403) exist from the first Dirichlet condition on. Indexed by volume node they
// were 1.6 GB at 57.6 M nodes: 60-90 ms of every mfh_mesh_build at 119^3, then 0.1 s of the first condition. Empty = no condition set.
void ensure_dirichlet_tables(mfh_ctx *c) {
    const size_t n = c->mesh.bdryNodes.size() * (size_t)c->mesh.dim;
    if (c->dirMask.size() == n && c->dirVal.size() == n) return;
    parallel_assign(c->dirMask, n, (uint8_t)0);
    parallel_assign(c->dirVal, n, 0.0);
}
void reset_bcs(mfh_ctx *c) {
    const HostMesh &m = c->mesh;
    c->neumannTraction.assign((size_t)m.nBE() * m.dim, 0.0);
    const size_t n = m.bdryNodes.size() * (size_t)m.dim;
    if (n > 0 && c->dirMask.size() == n && c->dirVal.size() == n) {     // same boundary size as before: cleared in place
        parallel_assign(c->dirMask, n, (uint8_t)0);
        parallel_assign(c->dirVal, n, 0.0);
    } else {
        RawVec<uint8_t>().swap(c->dirMask);
        RawVec<double>().swap(c->dirVal);
    }
    c->deltaForces.clear();
}
void clear_fixed(mfh_ctx *c) {
    c->fixedVars.clear();
    c->fixedVals.clear();
    if (c->hFixedMask.size() == (size_t)c->bs() * c->nDoF) parallel_assign(c->hFixedMask, c->hFixedMask.size(), (uint8_t)0);
    else RawVec<uint8_t>().swap(c->hFixedMask);                // (sized again by the first use: mfh_fix_variables / the first solve)
    c->fixedUploaded = false;
    c->anyFixedNonzero = false;
    c->dinvValid = false;
    c->tl.valid = false;
    c->mg.valid = false;
}

void upload_mesh(mfh_ctx *c, bool deviceTables) {
    double t0 = now_ms();
    c->autoStretch = -1.0;           // MFH_PRECOND_AUTO looks at the new vertices
    if (!c->hostOnly && !deviceTables) {      // (deviceTables: the device topology has written both from the vertices)
        require_device(c);
        MFH_HIP(hipSetDevice(c->device));
        c->dElemNodes.upload(c->mesh.elemNodes, c->stream);
        c->dVertPos.upload(c->mesh.nodePos, c->stream);   // corner nodes index into the node table
    }
    c->timing.upload_ms = now_ms() - t0;
    c->haveMesh = true;
    c->external = false;
    c->geoValid = false;
    c->hGeoValid = false;
    c->dofForNode.clear();
    c->nDoF = c->mesh.nNode;
    build_shape_tables(c->mesh.dim, c->mesh.deg, c->tables);
    if (!c->hostOnly) { c->dPairTable.upload(c->tables.pairTable, c->stream); c->dMassTable.upload(c->tables.massRef, c->stream); }
    invalidate_symbolic(c);
    dist_detach(c);                  // exchange lists describe the previous mesh: mfh_dist_setup must run again
    reset_bcs(c);
    clear_fixed(c);
}

// default material: E = 1, nu = 0.3 (Materials.hh:408)
void set_isotropic(mfh_ctx *c, double E, double nu) {
    double lam = (nu * E) / ((1.0 + nu) * (1.0 - 2.0 * nu));
    if (c->haveMesh && c->mesh.dim == 2) lam = (nu * E) / (1.0 - nu * nu);
    c->matMode = 0;
    c->matKind = MAT_ISO;
    c->matParams = {lam, E / (2.0 + 2.0 * nu), E, nu};
    c->geoValid = false;
    c->hGeoValid = false;
    invalidate_matrix(c);
}

void ensure_pass_events(mfh_ctx *c) {
    if (c->passEv[0]) return;
    for (auto &e : c->passEv) MFH_HIP(hipEventCreate(&e));
    MFH_HIP(hipHostMalloc((void **)&c->negHost, 2 * sizeof(int), hipHostMallocDefault));
}

// deferCheck: launch only; the caller runs finish_geometry after it has queued more work behind the embedding kernel
void ensure_geometry(mfh_ctx *c, bool deferCheck) {
    require(c->haveMesh, MFH_ERR_STATE, "no mesh set");
    if (c->geoValid) return;
    require(!c->hostOnly, MFH_ERR_HIP, "host-only context (device -1): no HIP device, and there is no CPU fallback");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    const HostMesh &m = c->mesh;
    if (c->matParams.empty() && !c->dMatBorrowed) set_isotropic(c, 1.0, 0.3);
    if (c->matMode == 0 && !c->dMatBorrowed) {
        // plane stress lambda depends on dim: recompute now that the mesh is known
        double E = c->matParams[2], nu = c->matParams[3];
        double lam = (nu * E) / ((1.0 + nu) * (1.0 - 2.0 * nu));
        if (m.dim == 2) lam = (nu * E) / (1.0 - nu * nu);
        c->matParams[0] = lam;
    }
    c->geoStride = c->matKind == MAT_ISO ? GEO_ISO_STRIDE : (c->matKind == MAT_ORTHO ? GEO_ORTHO_STRIDE : GEO_GEN_STRIDE);
    if (!c->dMatBorrowed) c->dMatParams.upload(c->matParams, c->stream);
    const double *dMat = c->dMatBorrowed ? c->dMatBorrowed : c->dMatParams.p;
    c->dGeo.alloc((size_t)m.nElem * c->geoStride);
    c->dNeg.alloc(2);   // [0] inverted elements, [1] elements with an indefinite orthotropic tensor
    c->dNeg.zero(c->stream);
    ensure_pass_events(c);
    MFH_HIP(hipEventRecord(c->passEv[0], c->stream));
    k::launch_geometry(m.dim, m.deg, c->matKind, m.nElem, c->dElemNodes.p, m.npe, c->dVertPos.p, dMat, c->matMode,
                       c->dGeo.p, c->geoStride, c->dNeg.p, c->stream);
    MFH_HIP(hipEventRecord(c->passEv[1], c->stream));
    MFH_HIP(hipMemcpyAsync(c->negHost, c->dNeg.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    c->geoPending = true;
    if (!deferCheck) finish_geometry(c);
}

// second half of the embedding: wait for it, read its time and its counters of inverted / indefinite elements
void finish_geometry(mfh_ctx *c) {
    if (!c->geoPending) return;
    c->geoPending = false;
    MFH_HIP(hipStreamSynchronize(c->stream));
    float ms = 0;
    MFH_HIP(hipEventElapsedTime(&ms, c->passEv[0], c->passEv[1]));
    c->timing.geometry_ms = ms;
    const int neg[2] = {c->negHost[0], c->negHost[1]};
    if (neg[0] > 0)   // LinearElasticity.hh:465-472
        throw Error(MFH_ERR_INVALID, "Mesh has negatively oriented elements.\nCorrect with: mesh_convert --reorientNegativeElements. (" +
                                         std::to_string(neg[0]) + " elements)");
    if (neg[1] > 0)   // the reference inverts the compliance blindly (ElasticityTensor.hh:136-164); a non-SPD K would only show up in the solver
        throw Error(MFH_ERR_INVALID, std::string(c->matMode == 1 ? "Isotropic" : "Orthotropic") + " parameters of " + std::to_string(neg[1]) +
                                         " elements give an indefinite elasticity tensor (" +
                                         (c->matMode == 1 ? "need E > 0 and -1 < nu < 1/2, nu < 1 in plane stress" : "compliance matrix not positive definite") + ").");
    c->geoValid = true;
    c->hGeoValid = false;
}

const std::vector<double> &host_geo(mfh_ctx *c) {
    ensure_geometry(c);
    if (!c->hGeoValid) {
        c->hGeo.resize(c->dGeo.n);
        c->dGeo.download(c->hGeo.data(), c->dGeo.n, c->stream);
        c->hGeoValid = true;
    }
    return c->hGeo;
}

const int32_t *device_dof_map(mfh_ctx *c) {
    if (c->dofForNode.empty()) return nullptr;
    if (!c->dofUploaded) { c->dDofForNode.upload(c->dofForNode, c->stream); c->dofUploaded = true; }
    return c->dDofForNode.p;
}

void ensure_host_colidx(mfh_ctx *c) {
    if (!c->sym.colIdx.empty() || c->sym.nnzb == 0) return;
    c->sym.colIdx.resize((size_t)c->sym.nnzb);
    c->dColIdx.download(c->sym.colIdx.data(), c->sym.colIdx.size(), c->stream);
}

// an option / operator change may have changed the storage the matrix should have: drop the K pattern (the operator lists stay)
void refresh_storage_rule(mfh_ctx *c) {
    if (c->symValid && resolve_upper_storage(c) != c->upperOnly) { c->symValid = false; invalidate_matrix(c); }
}

void ensure_symbolic(mfh_ctx *c, bool wantScatter) {
    require(c->haveMesh, MFH_ERR_STATE, "no mesh set");
    refresh_storage_rule(c);
    if (c->symValid && (!wantScatter || c->symHasScatter)) return;
    c->upperOnly = resolve_upper_storage(c);
    RoctxRange range("Compress Matrix");   // sumRepeated's sort / merge, hoisted into the once-per-mesh symbolic phase
    double t0 = now_ms();
    if (!c->hostOnly && c->symbolicDevice && c->contribOrder == 1) {
        // ---- device path: two radix sorts (mfh_symbolic_gpu.hip)
        require_device(c);
        MFH_HIP(hipSetDevice(c->device));
        // the gather codes leave the phase chunk-relative and packed (what k_assemble_gather reads) unless the host copy of the lists is
        // wanted (absolute codes, the host implementation's format) or the option is off
        const bool directPacked = c->asmPackedCodes && !c->keepHostSymbolic;
        bool packed = false;
        build_symbolic_device(c->mesh, c->dElemNodes.p, device_dof_map(c), c->nDoF, c->nOwnedDoF(), c->chunkSlots, wantScatter, c->stream,
                              c->sym, c->dRowPtr, c->dColIdx, c->dContribCode, c->dContribSlot, c->dScatter, c->upperOnly,
                              directPacked ? &c->dChunkElemBase : nullptr, &packed);
        const Symbolic &S = c->sym;
        c->dChunkRow.upload(S.chunkRow, c->stream);
        c->dSpmvChunkRow.upload(S.spmvChunkRow, c->stream);
        c->dContribPtr.upload(S.contribPtr, c->stream);
        c->symHasScatter = wantScatter;
        if (c->keepHostSymbolic) {
            ensure_host_colidx(c);
            c->sym.contribCode.resize(c->dContribCode.n);
            c->sym.contribSlot.resize(c->dContribSlot.n);
            c->dContribCode.download(c->sym.contribCode.data(), c->dContribCode.n, c->stream);
            c->dContribSlot.download(c->sym.contribSlot.data(), c->dContribSlot.n, c->stream);
            if (wantScatter) { c->sym.scatterSlot.resize(c->dScatter.n); c->dScatter.download(c->sym.scatterSlot.data(), c->dScatter.n, c->stream); }
        }
        if (directPacked) c->codesPacked = packed;
        else pack_gather_codes(c);
        c->timing.symbolic_ms = now_ms() - t0;
        const size_t tilesD = (size_t)((S.nnzb + 63) / 64);
        const double tV = now_ms();
        // the value array of a caller's context lives in a segment class of its own (mfh_pool.cpp); the linear level of a hierarchy is served from the
        // general class -- its 7.7 GB at 119^3 would otherwise miss the values' reservation (sized for the quadratic K) and wait 0.2 s for the driver
        { mfh::PoolTag values(c->hierarchyLevel ? 0 : 1); c->dVals.alloc(tiled_count(S.nnzb, c->dim() * c->dim())); }
        if (getenv("MFH_SYM_TIMING")) fprintf(stderr, "[symbolic] value array %.1f MB: allocation %.2f ms\n", tilesD * 64.0 * c->dim() * c->dim() * 8 / 1e6, now_ms() - tV);
        c->symValid = true;
        ++c->listsGen;
        ++c->valsGen;
        invalidate_matrix(c);
        return;
    }
    build_symbolic(c->mesh, c->dofForNode, c->nDoF, c->nOwnedDoF(), c->chunkSlots, c->contribOrder, wantScatter, c->sym, c->upperOnly);
    c->timing.symbolic_ms = now_ms() - t0;
    const Symbolic &S = c->sym;
    c->symHasScatter = wantScatter;
    if (c->hostOnly) { c->symValid = true; return; }
    MFH_HIP(hipSetDevice(c->device));
    c->dRowPtr.upload(S.rowPtr, c->stream);
    c->dColIdx.upload(S.colIdx, c->stream);
    c->dChunkRow.upload(S.chunkRow, c->stream);
    c->dSpmvChunkRow.upload(S.spmvChunkRow, c->stream);
    c->dContribPtr.upload(S.contribPtr, c->stream);
    c->dContribCode.upload(S.contribCode, c->stream);
    c->dContribSlot.upload(S.contribSlot, c->stream);
    if (wantScatter) c->dScatter.upload(S.scatterSlot, c->stream);
    pack_gather_codes(c);
    if (!c->keepHostSymbolic) {   // host copies of the big gather lists are no longer needed
        std::vector<uint32_t>().swap(c->sym.contribCode);
        std::vector<uint16_t>().swap(c->sym.contribSlot);
        std::vector<int32_t>().swap(c->sym.scatterSlot);
    }
    { mfh::PoolTag values(c->hierarchyLevel ? 0 : 1); c->dVals.alloc(tiled_count(S.nnzb, c->dim() * c->dim())); }
    c->symValid = true;
    ++c->listsGen;
    ++c->valsGen;
    invalidate_matrix(c);
}

// the device copy of the gather codes goes chunk-relative and packed (the host copy, mfh_symbolic_get, stays e npe^2 + ij)
void pack_gather_codes(mfh_ctx *c) {
    c->codesPacked = false;
    if (!c->asmPackedCodes || c->hostOnly || c->sym.nChunk() == 0) return;
    c->dChunkElemBase.alloc((size_t)c->sym.nChunk());
    DBuf<int> flag;
    flag.alloc(1);
    c->codesPacked = k::launch_pack_codes(c->sym.nChunk(), c->dContribPtr.p, c->dContribCode.p, c->mesh.npe, c->dChunkElemBase.p, flag.p, c->stream);
}

k::AsmArgs asm_args(mfh_ctx *c) {
    const HostMesh &m = c->mesh;
    k::AsmArgs a{};
    a.dim = m.dim; a.deg = m.deg; a.npe = m.npe; a.mat = c->asmMat();
    a.geoStride = c->geoStride; a.geo = c->dGeo.p; a.pairTable = c->dPairTable.p; a.massTable = c->dMassTable.p;
    for (int k2 = 0; k2 < 6; ++k2) a.pairConst[k2] = c->tables.pairConst[k2];
    a.nChunk = c->sym.nChunk(); a.chunkRow = c->dChunkRow.p; a.rowPtr = c->dRowPtr.p;
    a.contribPtr = c->dContribPtr.p; a.contribCode = c->dContribCode.p; a.contribSlot = c->dContribSlot.p;
    a.chunkSlots = c->sym.chunkSlots;
    a.nElem = m.nElem; a.scatterSlot = c->dScatter.p;
    a.vals = c->dVals.p; a.nnzb = c->sym.nnzb;
    a.xcd = c->xcdSwizzle;
    a.upperOnly = c->upperOnly ? 1 : 0;
    a.det = c->deterministic ? 1 : 0;
    a.chunkElemBase = c->codesPacked ? c->dChunkElemBase.p : nullptr;
    a.chunkOrder = nullptr;
    if (c->asmChunkOrder && !c->hostOnly && a.nChunk > 1 && c->symValid && c->dContribPtr.p && c->dContribCode.p && c->contribOrder == 1) {
        if (c->chunkOrderGen != c->listsGen) {
            // launch order of the row chunks: by the element their median contribution comes from (the gather lists are element-major
            // inside a chunk), so that the chunks holding an element's vertex rows and edge rows run at about the same time
            DBuf<uint32_t> keys;
            keys.alloc((size_t)a.nChunk);
            k::launch_chunk_keys(a, keys.p, c->stream);
            std::vector<uint32_t> hk((size_t)a.nChunk);
            keys.download(hk.data(), hk.size(), c->stream);
            std::vector<int32_t> order((size_t)a.nChunk);
            for (int64_t b = 0; b < a.nChunk; ++b) order[(size_t)b] = (int32_t)b;
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return hk[(size_t)x] < hk[(size_t)y]; });
            c->dChunkOrder.upload(order, c->stream);
            c->chunkOrderGen = c->listsGen;
        }
        a.chunkOrder = c->dChunkOrder.p;
    }
    return a;
}

void run_assembly(mfh_ctx *c, int mode) {
    c->dVals32.release();             // (a copy of the values about to be rewritten)
    k::AsmArgs a = asm_args(c);
    if (mode == MFH_ASSEMBLE_ATOMIC) {
        c->dVals.zero(c->stream);
        k::launch_assemble_atomic(a, c->stream);
    } else {
        k::launch_assemble_gather(a, c->stream);
    }
}

// Option "placement_trials": see mfh_ctx::placementTrials. Called after a gather-mode pass has written K into c->dVals. Every trial takes a
// values buffer from the arena while the earlier candidates are still held (so it is other memory), writes K into it with two timed passes
// and keeps it when it is at least 1 % faster than the best so far; the losers go back to the arena at the end. Trials stop when the device
// has less than twice the buffer free. Whatever happens, c->dVals holds a completely assembled K afterwards.
void placement_trials(mfh_ctx *c) {
    c->placementGen = c->valsGen;
    c->placementMs.clear();
    c->placementNote.clear();
    const size_t n = c->dVals.n, bytes = n * sizeof(double);
    if (c->placementTrials <= 0 || c->hostOnly || c->external || bytes < ((size_t)256 << 20)) return;
    auto timed = [&]() {
        k::AsmArgs a = asm_args(c);
        double best = 1e300;
        for (int r = 0; r < 2; ++r) {
            EventTimer t(c->stream);
            k::launch_assemble_gather(a, c->stream);
            best = std::min(best, t.stop());
        }
        return best;
    };
    double best = timed();
    c->placementMs.push_back(best);
    std::vector<std::unique_ptr<DBuf<double>>> held;
    for (int trial = 0; trial < c->placementTrials; ++trial) {
        // room for a candidate: what the driver still has, or a free chunk the arena already holds (e.g. a reservation) -- ADVICE r5: the
        // driver's figure alone skipped the trials silently while the memory was sitting in the arena
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); break; }
        if (fr < 2 * bytes + ((size_t)1 << 30) && mfh::device_arena_largest_free(c->device) < bytes) { c->placementNote = "stopped after " + std::to_string(trial) + " trials: no room for another candidate"; break; }
        std::unique_ptr<DBuf<double>> alt(new DBuf<double>());
        { mfh::PoolTag values(1); alt->alloc(n); }
        alt->zero(c->stream);              // (the padding of the last tile is never read; zero all the same)
        c->dVals.swap(*alt);               // dVals = candidate, alt = best so far
        const double t = timed();
        c->placementMs.push_back(t);
        if (t < 0.99 * best) best = t;
        else c->dVals.swap(*alt);          // back to the best so far (it still holds the K of its own passes)
        held.push_back(std::move(alt));
    }
    // (held buffers are released here, inside the API scope of the caller: they become free chunks of the arena)
}

void ensure_assembled(mfh_ctx *c) {
    if (c->assembled) return;
    require(!c->external, MFH_ERR_STATE, "no matrix set");
    ensure_geometry(c);
    ensure_symbolic(c, false);
    RoctxRange range("Assemble System");
    EventTimer t(c->stream);
    run_assembly(c, MFH_ASSEMBLE_GATHER);
    c->timing.assemble_ms = t.stop();
    if (c->placementTrials > 0 && c->placementGen != c->valsGen) placement_trials(c);
    c->assembled = true;
    c->dinvValid = false;
}

k::SpmvArgs spmv_args(mfh_ctx *c, bool masked) {
    k::SpmvArgs a{};
    a.dim = c->bs(); a.nChunk = (int64_t)c->sym.spmvChunkRow.size() - 1; a.chunkRow = c->dSpmvChunkRow.p; a.rowPtr = c->dRowPtr.p;
    a.colIdx = c->dColIdx.p; a.vals = c->dVals.p; a.chunkSlots = c->sym.spmvChunkSlots;
    a.fixedMask = masked ? c->dFixedMask.p : nullptr;
    a.xcd = c->xcdSwizzle == 1;
    return a;
}

void ensure_mf(mfh_ctx *c) {
    if (c->mfValid) return;
    require(c->haveMesh && !c->hostOnly, MFH_ERR_STATE, "the matrix-free operator needs a mesh on a device");
    ensure_symbolic(c, false);   // row ownership (nRows) and the DoF map on the device
    build_mf_lists_device(c->mesh, c->dElemNodes.p, device_dof_map(c), c->sym.nRows, c->stream, c->mf, c->dMfPairCode, c->dMfPairRow,
                          c->dMfPairPos, c->mfChunkRows, c->mfChunkPairs);
    c->dMfChunkRow.upload(c->mf.chunkRow, c->stream);
    c->dMfPairPtr.upload(c->mf.pairPtr, c->stream);
    c->dMfSig.alloc((size_t)c->mesh.npe * c->mesh.dim * (size_t)c->mesh.nElem);   // nodal forces, element-major
    c->mfValid = true;
}

void ensure_mf_cluster(mfh_ctx *c) {
    if (c->mfcValid) return;
    ensure_symbolic(c, false);
    // Block size (elements per workgroup). Round 5: the lanes take the elements of a block in rounds, and 512 elements per block (two
    // rounds) is the default where the block's rows still fit the LDS budget of three workgroups per CU -- the staging of x, the barriers
    // and the write-out of a block are shared by twice the elements, and a larger clump has fewer interface rows (scripts/mf_block_sweep.py:
    // 0.70 / 0.68 / 0.60 / 0.50 / 0.47 ms at 96 / 128 / 144 / 192 / 256 elements). Deterministic accumulation keeps four more accumulator
    // arrays and batched right-hand sides NRS-fold ones: 256 there. (Earlier: 256 / 240 / 192 give the same time with the generator's
    // element order: the interface is set by the shape of a run of consecutive elements.)
    const bool wide = c->deterministic || c->batchRhs;
    // linear elements have a fifth of the rows per element: 1 024 per block once the mesh is large enough to fill the device with such blocks
    // (scripts/p1_block_sweep.py: 0.192 / 0.157 / 0.147 / 0.148 ms at 256 / 512 / 1 024 / 2 048 for 6.3 M tets; 0.0375 / 0.0339 / 0.0344 for 1 M)
    const int autoBlock = (c->mesh.deg == 1 && c->mesh.nElem >= (int64_t)3000000) ? 1024 : 512;
    int be = c->mfBlockElems > 0 ? std::min(c->mfBlockElems, (int)MF_BLOCK_ELEMS_MAX) : (wide ? (int)MF_BLOCK : autoBlock);
    const size_t ldsBudget = 52 * 1024;        // three workgroups per CU (160 KB of LDS) -- what the registers of the quadratic kernel allow anyway
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int32_t *conn = c->dElemNodes.p;
        std::vector<int32_t> blockStart;
        if (c->mfReorder) {
            build_element_order_device(c->mesh, c->dElemNodes.p, c->dVertPos.p, c->stream, c->dMfElemPerm, c->dMfElemNodes, be, blockStart);
            conn = c->dMfElemNodes.p;
        } else { c->dMfElemPerm.release(); c->dMfElemNodes.release(); }
        build_mf_cluster_lists_device(c->mesh, conn, device_dof_map(c), c->sym.nRows, c->stream, c->mfc, c->mfcDev, be,
                                      blockStart.empty() ? nullptr : &blockStart);
        // blocks of more than MF_BLOCK elements whose rows do not fit the budget: once more with one round per block
        if (be > (int)MF_BLOCK && (size_t)2 * c->mfc.maxLocal * c->mesh.dim * sizeof(double) > ldsBudget && c->mfBlockElems <= 0) { be = MF_BLOCK; continue; }
        break;
    }
    // The cluster variant pays when most rows are finished inside a block: one LDS accumulator per distinct row of a
    // block plus the staged x of those rows and an interface buffer well below the per-pair force buffer it replaces.
    // An element order without locality (e.g. shuffled: ~2560 distinct rows per block) uses the two-pass variant instead.
    if ((size_t)2 * c->mfc.maxLocal * c->mesh.dim * sizeof(double) > ldsBudget || c->mfc.nIface * 3 > c->mesh.nElem * c->mesh.npe) {
        c->mfClusterUnfit = true;
        return;
    }
    c->mfcValid = true;
    ++c->listsGen;
}

k::SpmvMfArgs spmv_mf_cluster_args(mfh_ctx *c, bool masked) {
    const HostMesh &m = c->mesh;
    k::SpmvMfArgs a{};
    a.dim = m.dim; a.deg = m.deg; a.npe = m.npe; a.mat = c->asmMat();
    a.elemNodes = c->dElemNodes.p; a.dofForNode = device_dof_map(c);
    a.geo = c->dGeo.p; a.geoStride = c->geoStride; a.pairTable = c->dPairTable.p; a.massTable = c->dMassTable.p;
    a.nElem = m.nElem;
    if (c->dMfElemPerm.p) { a.elemNodes = c->dMfElemNodes.p; a.clElemPerm = c->dMfElemPerm.p; }
    a.clElemPtr = c->mfcDev.elemPtr.p;
    a.clLaneStride = 1;
    if (c->mfLaneStride > 1) {   // usable only if coprime to the block size (a bijection of the lanes)
        int x = c->mfLaneStride, y = (c->mfcDev.elemPtr.p || c->mfc.blockElems > (int)MF_BLOCK) ? (int)MF_BLOCK : c->mfc.blockElems;
        while (y) { const int t = x % y; x = y; y = t; }
        if (x == 1) a.clLaneStride = c->mfLaneStride;
    }
    a.xcd = c->mfXcdGroup > 1 ? c->mfXcdGroup : 0;    // runs of G consecutive element blocks per XCD (xcd_group_item)
    a.fixedMask = masked ? c->dFixedMask.p : nullptr;
    const auto &D = c->mfcDev;
    a.clBlocks = c->mfc.nBlocks; a.clMaxLocal = c->mfc.maxLocal; a.clBlockElems = c->mfc.blockElems; a.clBlockPtr = D.blockPtr.p; a.clEntryRow = D.entryRow.p;
    a.clEntryDest = D.entryDest.p; a.clLocalIdx = D.localIdx.p; a.clIfaceBuf = D.ifaceBuf.p; a.rowWrite = nullptr; a.rowMap = D.rowMap.p;
    // second pass: k_mf_rows streaming the interface partials (pairPos != null selects the sequential read)
    a.nChunk = c->mfc.nIface > 0 ? c->mfc.nChunk : 0;
    a.chunkRow = D.chunkRow.p; a.pairPtr = D.pairPtr.p; a.pairRow = D.ifaceRow.p; a.pairCode = nullptr;
    a.pairPos = reinterpret_cast<const uint32_t *>(D.ifaceRow.p);   // only tested against null
    a.sig = D.ifaceBuf.p; a.maxRows = c->mfc.maxRows;
    // constant material (k_geometry modes 0, 2, 5): the operator recomputes the gradients from the corner positions
    a.vertPos = (c->mfGeoFromVerts && (c->matMode == 0 || c->matMode == 2 || c->matMode == 5)) ? c->dVertPos.p : nullptr;
    return a;
}

k::SpmvMfArgs spmv_mf_args(mfh_ctx *c, bool masked) {
    const HostMesh &m = c->mesh;
    k::SpmvMfArgs a{};
    a.dim = m.dim; a.deg = m.deg; a.npe = m.npe; a.mat = c->asmMat();
    a.nChunk = (int64_t)c->mf.chunkRow.size() - 1; a.chunkRow = c->dMfChunkRow.p; a.pairPtr = c->dMfPairPtr.p;
    a.pairCode = c->dMfPairCode.p; a.pairRow = c->dMfPairRow.p; a.pairPos = c->mfModeEff() == 3 ? c->dMfPairPos.p : nullptr; a.elemNodes = c->dElemNodes.p; a.dofForNode = device_dof_map(c);
    a.geo = c->dGeo.p; a.geoStride = c->geoStride; a.pairTable = c->dPairTable.p; a.massTable = c->dMassTable.p;
    for (int k2 = 0; k2 < 6; ++k2) a.pairConst[k2] = c->tables.pairConst[k2];
    a.maxRows = c->mf.maxRows;
    a.xcd = c->xcdSwizzle == 1;
    a.nElem = m.nElem;
    a.sig = c->dMfSig.p;
    a.fixedMask = masked ? c->dFixedMask.p : nullptr;
    return a;
}

// lists of the matrix-free operator in use: the cluster variant needs only its own; the pair lists (modes 1-3, scalar
// operators, fallback of the cluster variant) are built on demand
bool prepare_matrix_free(mfh_ctx *c) {
    if (!c->use_mf()) return false;
    ensure_geometry(c);
    if (c->mfModeEff() == 4 && c->op == MFH_OP_ELASTICITY) ensure_mf_cluster(c);
    if (!(c->mfModeEff() == 4 && c->op == MFH_OP_ELASTICITY)) ensure_mf(c);
    return true;
}

// y = K x through the assembled matrix or the matrix-free operator (option "matrix_free")
void apply_operator(mfh_ctx *c, bool masked, const double *x, double *y, double *dotOut) {
    if (prepare_matrix_free(c)) {
        if (c->mfModeEff() == 4 && c->op == MFH_OP_ELASTICITY) k::launch_spmv_mf_cluster(spmv_mf_cluster_args(c, masked), x, y, dotOut, nullptr, 0, nullptr, false, c->stream);
        else if (c->mfModeEff() >= 2 && c->op == MFH_OP_ELASTICITY) k::launch_spmv_mf2(spmv_mf_args(c, masked), x, y, dotOut, nullptr, 0, nullptr, false, c->stream);
        else k::launch_spmv_mf(spmv_mf_args(c, masked), x, y, dotOut, nullptr, 0, nullptr, false, c->stream);
    } else if (c->upperOnly) {
        // the stored triangle serves both halves of the product (k_spmv_sym: transposed parts added with global atomics; measured slower than
        // k_spmv on both triangles, DESIGN 4.4: the PCG keeps asking for the full storage, this serves mfh_apply_K / mfh_dev_spmv)
        // k_spmv_sym adds in arrival order: not run-to-run reproducible, which option "deterministic" promises for every operator
        require(!c->deterministic, MFH_ERR_UNSUPPORTED, "deterministic 1: the product with an upper-triangle matrix (matrix_storage 1) adds with global atomics in arrival order; use matrix_storage 0 (both triangles) or the matrix-free operator");
        k::launch_spmv_sym(spmv_args(c, masked), c->sym.nRows, x, y, dotOut, c->stream);
    } else {
        k::launch_spmv(spmv_args(c, masked), x, y, dotOut, c->stream);
    }
}

// y = K x for a smoother inside the multigrid preconditioner: the assembled matrix read from its FP32 copy when the hierarchy made one
// (option mg_coarse_fp32; products and sums in FP64), the context's operator otherwise
void apply_operator_smoother(mfh_ctx *c, bool masked, const double *x, double *y) {
    if (c->dVals32.p && c->dVals32.n == c->dVals.n && c->assembled && !c->use_mf() && !c->upperOnly) {
        k::SpmvArgs a = spmv_args(c, masked);
        a.vals32 = c->dVals32.p;
        k::launch_spmv(a, x, y, nullptr, c->stream);
    } else
        apply_operator(c, masked, x, y, nullptr);
}

void ensure_fixed_uploaded(mfh_ctx *c) {
    if (c->fixedUploaded) return;
    const size_t n = (size_t)c->bs() * c->nDoF;
    if (c->hFixedMask.size() != n) parallel_assign(c->hFixedMask, n, (uint8_t)0);
    c->dFixedMask.upload(c->hFixedMask, c->stream);
    c->dFixedIdx.upload(c->fixedVars, c->stream);
    c->dFixedVal.upload(c->fixedVals, c->stream);
    c->fixedUploaded = true;
    c->dinvValid = false;
}

// sqrt(lambda_max / lambda_min) of the edge covariance of the mesh: 1 for an isotropic mesh, s for one stretched s : 1 : 1 (k_edge_covariance)
static double mesh_stretch(mfh_ctx *c) {
    const HostMesh &m = c->mesh;
    const int d = m.dim;
    DBuf<double> acc;
    acc.alloc(6);
    acc.zero(c->stream);
    k::launch_edge_covariance(m.nElem, d, m.npe, c->dElemNodes.p, c->dVertPos.p, acc.p, c->stream);
    double h[6];
    acc.download(h, 6, c->stream);
    double A[3][3] = {{h[0], h[5], h[4]}, {h[5], h[1], h[3]}, {h[4], h[3], h[2]}};
    if (d == 2) { A[2][2] = 0.5 * (A[0][0] + A[1][1]); A[0][2] = A[2][0] = A[1][2] = A[2][1] = 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep)            // cyclic Jacobi on a 3 x 3 symmetric matrix
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (std::fabs(A[p][q]) < 1e-300) continue;
                const double th = 0.5 * std::atan2(2.0 * A[p][q], A[q][q] - A[p][p]), cs = std::cos(th), sn = std::sin(th);
                double B[3][3];
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[i][j] = A[i][j];
                for (int k2 = 0; k2 < 3; ++k2) { B[p][k2] = cs * A[p][k2] - sn * A[q][k2]; B[q][k2] = sn * A[p][k2] + cs * A[q][k2]; }
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = B[i][j];
                for (int k2 = 0; k2 < 3; ++k2) { B[k2][p] = cs * A[k2][p] - sn * A[k2][q]; B[k2][q] = sn * A[k2][p] + cs * A[k2][q]; }
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = B[i][j];
            }
    double lo = 1e300, hi = 0;
    for (int i = 0; i < 3; ++i) { lo = std::min(lo, A[i][i]); hi = std::max(hi, A[i][i]); }
    return lo > 0 ? std::sqrt(hi / lo) : 1e300;
}

// MFH_PRECOND_AUTO: the preconditioner for the mesh in hand. The reference's direct solve is indifferent to the shape of the elements
// (SparseMatrices.hh:1984-2296); here the V-cycle's point smoothers and isotropic aggregates lose their grip on stretched meshes faster than the
// two-level preconditioner does, and past the measured crossover the latter is chosen. Row-partitioned contexts keep the V-cycle (the choice
// must be the same on every rank, and their meshes are slabs of the whole).
void resolve_auto_precond(mfh_ctx *c) {
    if (!c->precondAuto || c->autoStretch >= 0 || !c->haveMesh || c->hostOnly) return;
    int kind = MFH_PRECOND_MULTIGRID;
    c->autoStretch = 1.0;
    if (c->op == MFH_OP_ELASTICITY && !c->external && c->mesh.nOwned == c->mesh.nNode && c->dElemNodes.p && c->dVertPos.p) {
        c->autoStretch = mesh_stretch(c);
        if (c->autoStretch > c->autoStretchMax) kind = MFH_PRECOND_TWO_LEVEL;
    }
    if (kind != c->precond) { c->precond = kind; c->dinvValid = false; }
}

void ensure_precond(mfh_ctx *c) {
    resolve_auto_precond(c);
    if (c->deterministic) {
        if (c->use_mf() && c->mfModeEff() != 4)
            throw Error(MFH_ERR_UNSUPPORTED, "option deterministic: the matrix-free operator must be the cluster variant (matrix_free_mode 4) or the assembled SpMV (matrix_free 0)");
    }
    ensure_assembled(c);
    ensure_fixed_uploaded(c);
    if (c->dinvValid) return;
    const int d = c->bs();
    c->dDinv.alloc((size_t)c->sym.nRows * (d * (d + 1) / 2));   // symmetric-packed inverse diagonal blocks
    k::launch_extract_diag_inv(d, c->sym.nRows, c->dRowPtr.p, c->dColIdx.p, c->dVals.p, c->dFixedMask.p,
                               (c->precond == MFH_PRECOND_TWO_LEVEL || c->precond == MFH_PRECOND_MULTIGRID) ? MFH_PRECOND_BLOCK_JACOBI : c->precond, c->dDinv.p, c->stream);
    c->dinvValid = true;
    c->dinv32Valid = false;
}

// FP32 copy of the inverse diagonal blocks for the fused smoother kernels of the multigrid V-cycle (k_pcg_update's ZS flavour, k_mg_cheb_rz): the
// blocks are a third of those kernels' traffic, and a smoother does not need their last 29 bits -- both smoothing steps of a cycle read the SAME
// copy, so the preconditioner stays symmetric. The block-Jacobi preconditioner itself and every other kernel keep the FP64 blocks.
const float *smoother_dinv32(mfh_ctx *c) {
    if (!c->mgDinvFp32 || !c->dinvValid) return nullptr;
    if (!c->dinv32Valid || c->dDinv32.n != c->dDinv.n) {
        c->dDinv32.alloc(c->dDinv.n);
        k::launch_to_f32((int64_t)c->dDinv.n, c->dDinv.p, c->dDinv32.p, c->stream);
        c->dinv32Valid = true;
    }
    return c->dDinv32.p;
}

k::TLArgs tl_args(mfh_ctx *c) {
    k::TLArgs t{};
    t.dim = c->dim(); t.nModes = c->tl.nModes; t.nAgg = c->tl.nAgg; t.nDoF = c->sym.nRows;   // loops run over the owned rows
    t.aggOfDof = c->tl.aggOfDof.p; t.relPos = c->tl.relPos.p; t.fixedMask = c->fixedVars.empty() ? nullptr : c->dFixedMask.p;
    return t;
}

// Coarse operator -> its inverse, all in HBM: symmetrise + regularise (modes without support are
// decoupled), blocked Cholesky inverse. Ac: raw m x m row-major device matrix (left untouched).
bool dense_inverse_device(mfh_ctx *c, const double *Ac, int64_t mm, DBuf<double> &Ainv, int64_t &ldInv) {
    hipStream_t s = c->stream;
    std::vector<double> diag((size_t)mm);
    // pitch of (m + 1) doubles walks the diagonal
    MFH_HIP(hipMemcpy2DAsync(diag.data(), sizeof(double), Ac, (size_t)(mm + 1) * sizeof(double), sizeof(double), (size_t)mm,
                             hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    double maxd = 0;
    for (double v : diag) maxd = std::max(maxd, v);
    if (!(maxd > 0)) maxd = 1.0;
    std::vector<uint8_t> dead((size_t)mm);
    for (int64_t i = 0; i < mm; ++i) dead[i] = !(diag[i] > 1e-12 * maxd);   // mode without support: decoupled
    DBuf<uint8_t> dDead;
    dDead.upload(dead, s);
    const int64_t mp = ((mm + 63) / 64) * 64;
    DBuf<double> Ap, X, Dtile;
    DBuf<int> flag;
    Ap.alloc((size_t)mp * mp); X.alloc((size_t)mp * mp); Dtile.alloc((size_t)(mp / 64) * 64 * 64); flag.alloc(1);
    Ainv.alloc((size_t)mp * mp);
    k::launch_tl_prep(mm, mp, Ac, dDead.p, maxd, Ap.p, s);
    if (!k::dense_spd_inverse_device(Ap.p, X.p, Ainv.p, Dtile.p, mp, flag.p, s)) return false;
    ldInv = mp;
    return true;
}

bool tl_invert_device(mfh_ctx *c, const double *Ac, bool noteOnFailure) {
    auto &T = c->tl;
    if (!dense_inverse_device(c, Ac, T.m, T.Ainv, T.ldInv)) {
        if (noteOnFailure) c->precondNote = "two-level preconditioner: coarse operator not positive definite; using block-Jacobi";
        return false;
    }
    return true;
}

// Coarse space setup: aggregates, probing of Z^T K Z with 3^dim colours x nModes masked SpMVs, dense
// inverse on the host. Returns false (and leaves block-Jacobi in charge) when the coarse space does
// not apply: periodic DoF maps, partitioned rows, elements wider than an aggregate.
bool ensure_twolevel(mfh_ctx *c) {
    if (c->tl.valid) return true;
    c->precondNote.clear();
    if (c->op != MFH_OP_ELASTICITY || c->external) {
        c->precondNote = "two-level preconditioner is built on rigid-body modes (elasticity only): using Jacobi";
        return false;
    }
    const HostMesh &m = c->mesh;
    const int d = m.dim;
    if (c->sym.nRows != c->sym.nCols) {
        c->precondNote = "two-level preconditioner unavailable for partitioned rows: using block-Jacobi";
        return false;
    }
    if (!c->dofForNode.empty() && c->tlProbe) {
        c->precondNote = "two-level preconditioner: SpMV probing needs lattice-local coupling (no periodic DoF maps); using block-Jacobi";
        return false;
    }
    double t0 = now_ms();
    const int64_t nDoF = c->nDoF;
    // default: ~1000 aggregates (coarse dimension ~6000): setup (Galerkin pass + device dense inverse) ~0.3 s
    int target = c->aggNodes > 0 ? c->aggNodes : (int)std::max<int64_t>(512, nDoF / 1000);
    const bool tlTiming = getenv("MFH_TL_TIMING") != nullptr;
    double tp = now_ms();
    auto lap = [&](const char *what) {
        if (!tlTiming) return;
        MFH_HIP(hipStreamSynchronize(c->stream));
        const double t = now_ms();
        fprintf(stderr, "[two-level setup] %-28s %8.2f ms\n", what, t - tp);
        tp = t;
    };
    hipStream_t s = c->stream;
    auto &T = c->tl;
    Aggregates A;
    std::vector<double> relPos;   // host copy: validation / debug paths only
    if (!c->tlProbe && c->tlDeviceAggregates) {
        // ---- aggregates on the device: positions, bins, DoFs by aggregate, centroids and relPos never leave HBM
        DBuf<double> dDofPos;
        const double *dPos = c->dVertPos.p;
        if (!c->dofForNode.empty()) {
            dof_positions_device(m.nNode, d, device_dof_map(c), c->dVertPos.p, nDoF, s, dDofPos);
            dPos = dDofPos.p;
        }
        build_aggregates_device(d, nDoF, dPos, target, s, A, T.aggOfDof, T.relPos, T.aggPtr, T.dofsByAgg);
        lap("aggregates (device)");
        T.nModes = d == 3 ? 6 : 3;
        T.nAgg = A.nAgg; T.nColor = A.nColor; T.H = A.H;
        T.m = (int64_t)T.nAgg * T.nModes;
        T.colorOfAgg.upload(A.colorOfAgg, s); T.nbrOfColor.upload(A.nbrOfColor, s);
    } else {
    // position of a DoF = position of its first node (identity map: the node itself). With a periodic map the
    // modes of aggregates at the seam are no longer exact rigid motions, but any full-rank Z is a valid
    // Galerkin coarse space.
    std::vector<double> dofPosStore;
    const std::vector<double> *dofPosPtr = &dofPosStore;
    if (c->dofForNode.empty()) dofPosStore.assign(m.nodePos.begin(), m.nodePos.end());     // (host validation variant of the setup)
    else {
        dofPosStore.assign((size_t)nDoF * d, 0.0);
        std::vector<uint8_t> seen((size_t)nDoF, 0);
        for (int64_t n = 0; n < m.nNode; ++n) {
            const int32_t q = c->dofForNode[n];
            if (seen[q]) continue;
            seen[q] = 1;
            for (int a = 0; a < d; ++a) dofPosStore[(size_t)q * d + a] = m.nodePos[(size_t)n * d + a];
        }
        dofPosPtr = &dofPosStore;
    }
    const std::vector<double> &dofPos = *dofPosPtr;
    for (int attempt = 0; attempt < 6; ++attempt) {
        build_aggregates(d, nDoF, dofPos, target, A);
        // K couples only DoFs of one element: every element must fit into adjacent bins
        std::vector<uint8_t> bad((size_t)host_threads() + 1, 0);
        if (c->tlProbe) parallel_ranges(m.nElem, [&](int64_t eb, int64_t ee, int tid) {
            for (int64_t e = eb; e < ee; ++e) {
                const int32_t *c0 = &A.binCoord[(size_t)A.aggOfDof[m.elemNodes[(size_t)e * m.npe]] * 3];
                for (int k2 = 1; k2 < m.npe; ++k2) {
                    const int32_t *c1 = &A.binCoord[(size_t)A.aggOfDof[m.elemNodes[(size_t)e * m.npe + k2]] * 3];
                    if (std::abs(c0[0] - c1[0]) > 1 || std::abs(c0[1] - c1[1]) > 1 || std::abs(c0[2] - c1[2]) > 1) { bad[tid] = 1; return; }
                }
            }
        });
        bool ok = true;
        for (uint8_t b : bad) ok &= !b;
        if (!c->tlProbe) ok = true;   // the Galerkin kernel handles any aggregate pair; only probing needs adjacency
        if (ok) break;
        target *= 4;
        if (attempt == 5) { c->precondNote = "two-level preconditioner: elements span non-adjacent aggregates; using block-Jacobi"; return false; }
    }
    lap("aggregates (host)");
    T.nModes = d == 3 ? 6 : 3;
    T.nAgg = A.nAgg; T.nColor = A.nColor; T.H = A.H;
    T.m = (int64_t)T.nAgg * T.nModes;
    relPos.assign((size_t)nDoF * 3, 0.0);
    for (int64_t n = 0; n < nDoF; ++n)
        for (int a = 0; a < d; ++a) relPos[(size_t)n * 3 + a] = (dofPos[(size_t)n * d + a] - A.centroid[(size_t)A.aggOfDof[n] * 3 + a]) / A.H;
    T.aggOfDof.upload(A.aggOfDof, s); T.relPos.upload(relPos, s); T.aggPtr.upload(A.aggPtr, s); T.dofsByAgg.upload(A.dofsByAgg, s);
    T.colorOfAgg.upload(A.colorOfAgg, s); T.nbrOfColor.upload(A.nbrOfColor, s);
    lap("relPos + uploads");
    }
    if (c->tlProbe || !c->tlRapAgg) require_full_storage(c, "this construction of the coarse operator (options tl_probe / tl_rap_agg 0)");
    T.rc.alloc((size_t)T.m); T.yc.alloc((size_t)T.m);
    DBuf<double> Ac;
    Ac.alloc((size_t)T.m * T.m);
    Ac.zero(s);
    const int64_t n = (int64_t)d * nDoF;
    const k::TLArgs ta = tl_args(c);
    if (c->tlProbe) {   // reference construction: probe Z^T K Z with 3^dim colours x nModes masked SpMVs
        c->wx.alloc(n); c->wAp.alloc(n);
        const k::SpmvArgs sa = spmv_args(c, !c->fixedVars.empty());
        for (int color = 0; color < T.nColor; ++color)
            for (int mode = 0; mode < T.nModes; ++mode) {
                k::launch_tl_fill(ta, T.colorOfAgg.p, color, mode, c->wx.p, s);
                k::launch_spmv(sa, c->wx.p, c->wAp.p, nullptr, s);
                k::launch_tl_restrict(ta, T.aggPtr.p, T.dofsByAgg.p, c->wAp.p, T.rc.p, s);
                k::launch_tl_scatter(T.nAgg, T.nModes, T.nColor, T.nbrOfColor.p, color, mode, T.rc.p, Ac.p, s);
            }
    } else {            // one Galerkin pass over the assembled K
        T.binCoord.upload(A.binCoord, s);
        if (c->tlRapAgg) k::launch_tl_rap_agg(ta, T.aggPtr.p, T.dofsByAgg.p, T.binCoord.p, c->dRowPtr.p, c->dColIdx.p, c->dVals.p, Ac.p, s, c->upperOnly, c->sym.nRows);
        else k::launch_tl_rap(ta, c->sym.nRows, c->dRowPtr.p, c->dColIdx.p, c->dVals.p, Ac.p, s);
    }
    lap("Galerkin product");
    const int64_t mm = T.m;
    if (!c->tlHostInverse) {
        if (!tl_invert_device(c, Ac.p, true)) return false;
        Ac.release();
        lap("dense inverse");
    } else {
    std::vector<double> hA((size_t)T.m * T.m);
    Ac.download(hA.data(), hA.size(), s);
    Ac.release();
    double maxd = 0;
    for (int64_t i = 0; i < mm; ++i) maxd = std::max(maxd, hA[(size_t)i * mm + i]);
    parallel_ranges(mm, [&](int64_t b, int64_t e, int) {   // symmetrise
        for (int64_t i = b; i < e; ++i)
            for (int64_t j = 0; j < i; ++j) {
                const double v = 0.5 * (hA[(size_t)i * mm + j] + hA[(size_t)j * mm + i]);
                hA[(size_t)i * mm + j] = v;
            }
    }, 64);
    for (int64_t i = 0; i < mm; ++i)
        for (int64_t j = 0; j < i; ++j) hA[(size_t)j * mm + i] = hA[(size_t)i * mm + j];
    // modes without support (aggregate fully fixed, degenerate rotation): decouple them
    for (int64_t i = 0; i < mm; ++i)
        if (!(hA[(size_t)i * mm + i] > 1e-12 * maxd)) {
            for (int64_t j = 0; j < mm; ++j) { hA[(size_t)i * mm + j] = 0; hA[(size_t)j * mm + i] = 0; }
            hA[(size_t)i * mm + i] = maxd > 0 ? maxd : 1.0;
        }
    for (int64_t i = 0; i < mm; ++i) hA[(size_t)i * mm + i] *= 1.0 + 1e-10;
    if (!spd_inverse_inplace(mm, hA.data())) {
        c->precondNote = "two-level preconditioner: coarse operator not positive definite; using block-Jacobi";
        return false;
    }
    T.Ainv.upload(hA, s);
    T.ldInv = mm;
    }
    T.setup_ms = now_ms() - t0;
    T.valid = true;
    return true;
}

// the levels below the Jacobi one that the preconditioner in use needs; the multigrid hierarchy falls back to the two-level
// construction where it does not apply (linear elements, partitioned rows: precondNote says so)
void ensure_coarse_levels(mfh_ctx *c, int /*nrhs*/) {
    if (c->precond == MFH_PRECOND_MULTIGRID) {
        if (ensure_multigrid(c)) return;
        const std::string note = c->precondNote;
        if (ensure_twolevel(c) && c->precondNote.empty()) c->precondNote = note;
        return;
    }
    if (c->precond == MFH_PRECOND_TWO_LEVEL) ensure_twolevel(c);
}

// z = M^-1 r with the two-level preconditioner (restrict -> dense coarse solve -> prolong + block-Jacobi)
void tl_precond(mfh_ctx *c, const double *r, double *z, double *scal, int it) {
    const k::TLArgs ta = tl_args(c);
    auto &T = c->tl;
    k::launch_tl_restrict(ta, T.aggPtr.p, T.dofsByAgg.p, r, T.rc.p, c->stream);
    k::launch_tl_gemv(T.m, T.ldInv, T.Ainv.p, T.rc.p, T.yc.p, c->stream);
    k::launch_tl_apply(ta, c->dDinv.p, r, T.yc.p, z, scal, it, c->stop.p, c->stream);
}

double device_dot(mfh_ctx *c, int64_t n, const double *a, const double *b) {
    c->stop.alloc(4);
    MFH_HIP(hipMemsetAsync(c->stop.p + 1, 0, sizeof(double), c->stream));
    k::launch_dot(n, a, b, c->stop.p + 1, c->stream);
    double v = 0;
    MFH_HIP(hipMemcpyAsync(&v, c->stop.p + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MFH_HIP(hipStreamSynchronize(c->stream));
    return v;
}

// Classic PCG (two reduction points per iteration) on the free variables of K, one right-hand side: kept as option
// "pcg_variant" 0 and for the operator variants without a batched kernel; the default solver is mfh_solver.cpp.
void solve_one_classic(mfh_ctx *c, const double *f, double *u, double rtol, int maxit, mfh_solve_info *info) {
    RoctxRange range("Elasticity Solve");
    const int d = c->bs();
    const int64_t n = (int64_t)d * c->nDoF;
    require(c->sym.nRows == c->sym.nCols, MFH_ERR_STATE,
            "mfh_solve needs all rows owned; use the mfh_dev_* building blocks for partitioned meshes");
    hipStream_t s = c->stream;
    c->wx.alloc(n); c->wr.alloc(n); c->wz.alloc(n); c->wp.alloc(n); c->wAp.alloc(n); c->wb.alloc(n); c->wf.alloc(n);
    c->stop.alloc(4);
    EventTimer tsetup(s);
    prepare_matrix_free(c);   // gather lists of the operator: once per mesh / DoF map, part of the setup time
    if (f) MFH_HIP(hipMemcpyAsync(c->wf.p, f, n * sizeof(double), hipMemcpyHostToDevice, s));      // (null: the caller has formed it in c->wf on the device)
    // b = f - K ubar on the free variables (SparseMatrices.hh:2457-2470,2526-2535)
    MFH_HIP(hipMemcpyAsync(c->wb.p, c->wf.p, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (c->anyFixedNonzero && !c->solveHomogeneous) {
        c->wu0.alloc(n);
        c->wu0.zero(s);
        k::launch_scatter_values((int64_t)c->fixedVars.size(), c->dFixedIdx.p, c->dFixedVal.p, c->wu0.p, n, s);
        apply_operator(c, false, c->wu0.p, c->wAp.p, nullptr);
        k::launch_axpby(n, -1.0, c->wAp.p, 1.0, c->wb.p, s);
    }
    if (!c->fixedVars.empty()) k::launch_mask(n, c->dFixedMask.p, c->wb.p, s);
    const double bb = device_dot(c, n, c->wb.p, c->wb.p);
    mfh_solve_info li{};
    // the graph path runs whole blocks of check_every iterations, so the history may run past maxit
    const size_t scalN = ((size_t)maxit + (size_t)c->checkEvery + 2) * 4;
    c->scal.alloc(scalN);
    c->scal.zero(s);
    const double stopv = rtol * rtol * bb;
    MFH_HIP(hipMemsetAsync(c->stop.p, 0, 4 * sizeof(double), s));   // [0] threshold, [1] dot scratch, [3] iteration base
    MFH_HIP(hipMemcpyAsync(c->stop.p, &stopv, sizeof(double), hipMemcpyHostToDevice, s));
    li.setup_ms = tsetup.stop();
    int itDone = 0;
    double rrFinal = 0;
    if (bb == 0.0) {
        c->wx.zero(s);
        li.converged = 1;
    } else {
        EventTimer tsolve(s);
        const bool useMG = c->precond == MFH_PRECOND_MULTIGRID && c->mg.valid && c->mg.singular == c->tlSuppress;
        const bool useTL = !useMG && (c->precond == MFH_PRECOND_TWO_LEVEL || c->precond == MFH_PRECOND_MULTIGRID) && c->tl.valid && !c->tlSuppress;
        const uint8_t *maskPtr = c->fixedVars.empty() ? nullptr : c->dFixedMask.p;
        k::launch_pcg_init(d, c->sym.nRows, c->dDinv.p, c->wb.p, c->wx.p, c->wr.p, c->wz.p, c->wp.p, c->scal.p, s);
        if (useTL || useMG) {   // replace z, p and r.z of the block-Jacobi initialisation
            MFH_HIP(hipMemsetAsync(c->scal.p, 0, sizeof(double), s));
            if (useMG) {
                mg_precond(c, c->wr.p, c->wz.p, nullptr, -1, nullptr);
                k::launch_mg_rz(n, c->wr.p, c->wz.p, maskPtr, c->scal.p, -1, nullptr, nullptr, s);
            } else
                tl_precond(c, c->wr.p, c->wz.p, c->scal.p, -1);
            MFH_HIP(hipMemcpyAsync(c->wp.p, c->wz.p, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        const k::SpmvArgs sa = spmv_args(c, !c->fixedVars.empty());
        const bool useMF = c->use_mf();
        if (!useMF) require_full_storage(c, "the assembled SpMV of the PCG");
        const bool useCluster = useMF && c->mfModeEff() == 4 && c->op == MFH_OP_ELASTICITY;
        const k::SpmvMfArgs mfa = useCluster ? spmv_mf_cluster_args(c, !c->fixedVars.empty())
                                             : (useMF ? spmv_mf_args(c, !c->fixedVars.empty()) : k::SpmvMfArgs{});
        const double mgZs = useMG && c->mgFuse ? mg_fuse_scale(c) : 0.0;
        const float *dinv32 = mgZs > 0 ? smoother_dinv32(c) : nullptr;
        std::vector<double> hs;
        int it = 0;
        bool done = false;
        int lastChecked = 0;
        auto enqueue = [&](int itLocal) {   // one PCG iteration; `itLocal` is relative to the iteration base stop[3]
            if (useCluster) k::launch_spmv_mf_cluster(mfa, c->wp.p, c->wAp.p, nullptr, c->scal.p, itLocal, c->stop.p, true, s);
            else if (useMF && c->mfModeEff() >= 2 && c->op == MFH_OP_ELASTICITY) k::launch_spmv_mf2(mfa, c->wp.p, c->wAp.p, nullptr, c->scal.p, itLocal, c->stop.p, true, s);
            else if (useMF) k::launch_spmv_mf(mfa, c->wp.p, c->wAp.p, nullptr, c->scal.p, itLocal, c->stop.p, true, s);
            else k::launch_pcg_spmv(sa, c->wp.p, c->wAp.p, c->scal.p, itLocal, c->stop.p, s);
            if (useMG && mgZs > 0) {
                // the V-cycle's first and last vector kernels folded into the loop's own (MgFuse): r -= alpha Ap and z = Dinv r / theta in one
                // pass, the last smoothing step and r.z in another -- three vector passes of 24.6 less per iteration
                k::launch_pcg_update_presmooth(d, c->sym.nRows, c->dDinv.p, dinv32, c->wAp.p, c->wr.p, c->wz.p, mgZs, c->scal.p, itLocal, c->stop.p, s);
                const MgFuse fz{true, c->scal.p, maskPtr};
                mg_precond(c, c->wr.p, c->wz.p, c->scal.p, itLocal, c->stop.p, &fz);
            } else if (useMG) {
                k::launch_pcg_update_noz(d, c->sym.nRows, c->wAp.p, c->wr.p, c->scal.p, itLocal, c->stop.p, s);
                mg_precond(c, c->wr.p, c->wz.p, c->scal.p, itLocal, c->stop.p);
                k::launch_mg_rz(n, c->wr.p, c->wz.p, maskPtr, c->scal.p, itLocal, c->scal.p, c->stop.p, s);
            } else if (useTL) {
                k::launch_pcg_update_noz(d, c->sym.nRows, c->wAp.p, c->wr.p, c->scal.p, itLocal, c->stop.p, s);
                tl_precond(c, c->wr.p, c->wz.p, c->scal.p, itLocal);
            } else
                k::launch_pcg_update(d, c->sym.nRows, c->dDinv.p, c->wAp.p, c->wr.p, c->wz.p, c->scal.p, itLocal, c->stop.p, s);
            k::launch_pcg_direction(n, c->wz.p, c->wp.p, c->wx.p, c->scal.p, itLocal, c->stop.p, s);
        };
        // Launch-bound regime (small meshes: a few tens of microseconds per kernel): capture one block of
        // check_every iterations in a hipGraph and replay it; the kernels find their iteration through the
        // device-side base stop[3], which the last node of the graph advances.
        // a multigrid iteration is tens of kernels and milliseconds long and only tens of them are needed: short blocks (the operator
        // applications inside the V-cycle are not gated, so iterations past convergence would cost real time)
        const int checkEvery = useMG ? std::min(c->checkEvery, 2) : c->checkEvery;
        hipGraphExec_t exec = nullptr;
        if (c->useGraph && checkEvery > 1) {
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
                bool ok = true;
                try {
                    for (int j = 0; j < checkEvery; ++j) enqueue(j);
                    k::launch_advance_base(c->stop.p, checkEvery, s);
                } catch (...) { ok = false; }
                if (hipStreamEndCapture(s, &graph) != hipSuccess || !ok || !graph) { graph = nullptr; (void)hipGetLastError(); }
            } else (void)hipGetLastError();
            if (graph) {
                if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { exec = nullptr; (void)hipGetLastError(); }
                (void)hipGraphDestroy(graph);
            }
        }
        li.used_graph = exec ? 1 : 0;
        double bestRR = 1e300;
        int itBest = 0;
        // (a plateau is not a stagnation: block-Jacobi PCG on a one-layer plate in bending, 59 k DOF, sits above its best residual for more than
        // 5 000 iterations and then converges at 5 913 -- CG owes its answer within about n iterations, so the window grows with n)
        const int stagnationWindow = (int)std::max<int64_t>(std::max(5000, 40 * c->checkEvery), std::min<int64_t>(n, 50000));
        while (!done && it < maxit) {
            if (exec) {
                MFH_HIP(hipGraphLaunch(exec, s));
                it += checkEvery;
            } else {
                const int itEnd = std::min(maxit, it + checkEvery);
                for (; it < itEnd; ++it) enqueue(it);
            }
            // scan the residual history of the iterations just enqueued
            hs.resize((size_t)(it - lastChecked + 1) * 4);
            MFH_HIP(hipMemcpyAsync(hs.data(), c->scal.p + (size_t)lastChecked * 4, hs.size() * sizeof(double), hipMemcpyDeviceToHost, s));
            MFH_HIP(hipStreamSynchronize(s));
            for (int k2 = lastChecked; k2 <= std::min(it, maxit); ++k2) {
                const double rr = hs[(size_t)(k2 - lastChecked) * 4 + 2];
                if (rr <= stopv) { done = true; itDone = k2; rrFinal = rr; break; }
                if (!(rr == rr)) throw Error(MFH_ERR_NOT_CONVERGED, "PCG breakdown (NaN residual): K is not SPD on the free variables");
                // CHOLMOD reports "not positive definite" at once; the iterative counterpart: negative curvature, or a
                // residual whose best value has not improved by 10 % for thousands of iterations (a singular system with
                // an inconsistent right-hand side: missing boundary conditions, unbalanced loads on a free body)
                const double pAp = hs[(size_t)(k2 - lastChecked) * 4 + 1];
                if (k2 < it && pAp < 0.0)
                    throw Error(MFH_ERR_NOT_CONVERGED, "PCG breakdown (p.Kp = " + std::to_string(pAp) + " < 0 at iteration " + std::to_string(k2) +
                                                           ", residual^2 " + std::to_string(rr) + "): K is not positive definite on the free variables");
                if (rr < 0.9 * bestRR) { bestRR = rr; itBest = k2; }
                else if (k2 - itBest > stagnationWindow)
                    throw Error(MFH_ERR_NOT_CONVERGED, "PCG stagnated (no progress of the residual for " + std::to_string(stagnationWindow) +
                                                           " iterations): the system is singular with an inconsistent right-hand side "
                                                           "(missing boundary conditions?) or too ill-conditioned for this preconditioner");
            }
            if (!done) { itDone = std::min(it, maxit); rrFinal = hs[(size_t)(itDone - lastChecked) * 4 + 2]; }
            lastChecked = it;
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        li.solve_ms = tsolve.stop();
        li.converged = done ? 1 : 0;
    }
    li.iterations = itDone;
    li.rel_residual = bb > 0 ? std::sqrt(rrFinal / bb) : 0.0;
    // u = x + ubar  (SparseMatrices.hh:2592-2605)
    if (!c->fixedVars.empty() && !c->solveHomogeneous)
        k::launch_scatter_values((int64_t)c->fixedVars.size(), c->dFixedIdx.p, c->dFixedVal.p, c->wx.p, n, s);
    // true residual on the free variables: || mask(f - K u) || / ||b||
    if (bb > 0) {
        apply_operator(c, false, c->wx.p, c->wAp.p, nullptr);
        k::launch_axpby(n, 1.0, c->wf.p, -1.0, c->wAp.p, s);
        if (!c->fixedVars.empty()) k::launch_mask(n, c->dFixedMask.p, c->wAp.p, s);
        li.true_rel_residual = std::sqrt(device_dot(c, n, c->wAp.p, c->wAp.p) / bb);
    }
    MFH_HIP(hipMemcpyAsync(u, c->wx.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    if (info) *info = li;
    check_residual_gap(li, rtol);
}

// constantStrainLoad (LinearElasticity.hh:551-562) on the device through the lists of the cluster operator: the element routine with u = 0 and
// the constant strain added, summed in LDS like an application of the operator (0.25 ms at 2 M quadratic tets; the stand-alone kernel adds
// every (element, node) contribution with a global FP64 atomic: 2.3 ms). outDev: dim * nDoF doubles on the device.
bool constant_strain_load_device(mfh_ctx *c, const double *cstrainFlat, double *outDev) {
    if (c->op != MFH_OP_ELASTICITY || !c->use_mf()) return false;
    ensure_geometry(c);
    prepare_matrix_free(c);
    if (c->mfModeEff() != 4) return false;
    const int d = c->bs();
    if (c->mfcDev.ifaceBuf.n < (size_t)std::max<int64_t>(c->mfc.nIface, 1) * d) c->mfcDev.ifaceBuf.alloc((size_t)std::max<int64_t>(c->mfc.nIface, 1) * d);
    MFH_HIP(hipMemsetAsync(outDev, 0, (size_t)d * c->nDoF * sizeof(double), c->stream));
    k::launch_mf_cluster_constant_strain(spmv_mf_cluster_args(c, false), cstrainFlat, outDev, c->stream);
    return true;
}

// NR right-hand sides under the multigrid preconditioner of an unpartitioned quadratic context (solve_many, option "mg_batch"): NR classic PCG
// loops advancing in lockstep -- loop k with vectors of its own (k vecStride apart), its own history scal + k scalStride and control block
// stop + 4 k, hence its own convergence: a loop that has met its threshold is frozen by its gate like in solve_one_classic -- and ONE pass
// through the linear / aggregate / dense levels of every V-cycle for all of them (mg_precond_batch). The quadratic level's kernels are the
// single-vector ones at their single-vector cost; what the batch shares is everything below, which is bound by matrix bytes and launch latency.
// The reference's counterpart: one factorisation, one back-substitution per right-hand side (SparseMatrices.hh:2106-2124).
void solve_multigrid_batch(mfh_ctx *c, int NR, const double *f, double *u, int64_t hostStride, double rtol, int maxit, mfh_solve_info *infos, const BatchIO *io) {
    RoctxRange range("Elasticity Solve");
    const int d = c->bs();
    const int64_t n = (int64_t)d * c->nDoF;
    require(c->sym.nRows == c->sym.nCols, MFH_ERR_STATE, "batched multigrid solve needs all rows owned");
    require(NR >= 2 && NR <= 6, MFH_ERR_INVALID, "batch size");
    hipStream_t s = c->stream;
    const int64_t vs = (n + 31) / 32 * 32;                   // doubles between the vectors of consecutive loops (256-byte aligned: the vector kernels use 16-byte accesses)
    const size_t tot = (size_t)vs * NR;
    c->wx.reserve(tot); c->wr.reserve(tot); c->wz.reserve(tot); c->wp.reserve(tot); c->wAp.reserve(tot); c->wb.reserve(tot); c->wf.reserve(tot);
    EventTimer tsetup(s);
    prepare_matrix_free(c);
    const bool masked = !c->fixedVars.empty();
    const uint8_t *maskPtr = masked ? c->dFixedMask.p : nullptr;
    if (io && io->cstrains) {               // constantStrainLoad vectors, formed where they are needed
        const int fl = c->dim() * (c->dim() + 1) / 2;
        for (int k2 = 0; k2 < NR; ++k2)
            if (!constant_strain_load_device(c, io->cstrains + (size_t)k2 * fl, c->wf.p + (size_t)k2 * vs)) throw Error(MFH_ERR_STATE, "constant-strain loads need the cluster operator");
    } else
        for (int k2 = 0; k2 < NR; ++k2)
            MFH_HIP(hipMemcpyAsync(c->wf.p + (size_t)k2 * vs, f + (size_t)k2 * hostStride, n * sizeof(double), hipMemcpyHostToDevice, s));
    MFH_HIP(hipMemcpyAsync(c->wb.p, c->wf.p, tot * sizeof(double), hipMemcpyDeviceToDevice, s));
    // b = f - K ubar on the free variables (SparseMatrices.hh:2457-2470,2526-2535): the same lift for every right-hand side
    if (c->anyFixedNonzero && !c->solveHomogeneous) {
        c->wu0.alloc(n);
        c->wu0.zero(s);
        k::launch_scatter_values((int64_t)c->fixedVars.size(), c->dFixedIdx.p, c->dFixedVal.p, c->wu0.p, n, s);
        apply_operator(c, false, c->wu0.p, c->wAp.p, nullptr);
        for (int k2 = 0; k2 < NR; ++k2) k::launch_axpby(n, -1.0, c->wAp.p, 1.0, c->wb.p + (size_t)k2 * vs, s);
    }
    if (masked) for (int k2 = 0; k2 < NR; ++k2) k::launch_mask(n, c->dFixedMask.p, c->wb.p + (size_t)k2 * vs, s);
    double bb[6] = {0, 0, 0, 0, 0, 0};
    for (int k2 = 0; k2 < NR; ++k2) bb[k2] = device_dot(c, n, c->wb.p + (size_t)k2 * vs, c->wb.p + (size_t)k2 * vs);      // (uses c->stop: before the control blocks are set up)
    const int checkEvery = std::min(c->checkEvery, 2);       // a V-cycle is milliseconds long and tens of them are needed: short blocks
    const size_t scalStride = ((size_t)maxit + (size_t)checkEvery + 2) * 4;
    c->scal.alloc(scalStride * NR);
    c->scal.zero(s);
    c->stop.alloc(4 * (size_t)NR);
    double hstop[24] = {0};
    bool anyWork = false;
    for (int k2 = 0; k2 < NR; ++k2) { hstop[4 * k2] = rtol * rtol * bb[k2]; anyWork |= bb[k2] > 0; }
    MFH_HIP(hipMemcpyAsync(c->stop.p, hstop, 4 * NR * sizeof(double), hipMemcpyHostToDevice, s));
    const double setupMs = tsetup.stop();
    std::vector<int> itConv((size_t)NR, -1);
    std::vector<double> rrFinal((size_t)NR, 0.0);
    double solveMs = 0;
    bool usedGraph = false;
    int itRun = 0;
    auto vec = [&](DBuf<double> &b, int k2) { return b.p + (size_t)k2 * vs; };
    auto sck = [&](int k2) { return c->scal.p + (size_t)k2 * scalStride; };
    auto stk = [&](int k2) { return c->stop.p + 4 * (size_t)k2; };
    if (!anyWork) {
        MFH_HIP(hipMemsetAsync(c->wx.p, 0, tot * sizeof(double), s));
        for (int k2 = 0; k2 < NR; ++k2) itConv[k2] = 0;
    } else {
        EventTimer tsolve(s);
        // x = 0, r = b, z = M^-1 r (one V-cycle for all), p = z; {r.z, -, r.r}_0 per loop. A zero right-hand side starts converged (0 <= 0).
        for (int k2 = 0; k2 < NR; ++k2) {
            k::launch_pcg_init(d, c->sym.nRows, c->dDinv.p, vec(c->wb, k2), vec(c->wx, k2), vec(c->wr, k2), vec(c->wz, k2), vec(c->wp, k2), sck(k2), s);
            MFH_HIP(hipMemsetAsync(sck(k2), 0, sizeof(double), s));
        }
        mg_precond_batch(c, NR, c->wr.p, c->wz.p, vs, nullptr, 0, -1, nullptr);
        for (int k2 = 0; k2 < NR; ++k2) {
            k::launch_mg_rz(n, vec(c->wr, k2), vec(c->wz, k2), maskPtr, sck(k2), -1, nullptr, nullptr, s);
            MFH_HIP(hipMemcpyAsync(vec(c->wp, k2), vec(c->wz, k2), n * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        const k::SpmvMfArgs mfa = spmv_mf_cluster_args(c, masked);
        const double mgZs = c->mgFuse ? mg_fuse_scale(c) : 0.0;
        const float *dinv32 = mgZs > 0 ? smoother_dinv32(c) : nullptr;
        auto enqueue = [&](int itLocal) {   // one iteration of every loop; `itLocal` is relative to the iteration bases stop[4 k + 3]
            for (int k2 = 0; k2 < NR; ++k2) {
                k::launch_spmv_mf_cluster(mfa, vec(c->wp, k2), vec(c->wAp, k2), nullptr, sck(k2), itLocal, stk(k2), true, s);
                if (mgZs > 0) k::launch_pcg_update_presmooth(d, c->sym.nRows, c->dDinv.p, dinv32, vec(c->wAp, k2), vec(c->wr, k2), vec(c->wz, k2), mgZs, sck(k2), itLocal, stk(k2), s);
                else k::launch_pcg_update_noz(d, c->sym.nRows, vec(c->wAp, k2), vec(c->wr, k2), sck(k2), itLocal, stk(k2), s);
            }
            const MgFuse fz{mgZs > 0, mgZs > 0 ? c->scal.p : nullptr, maskPtr};      // (see solve_one_classic)
            mg_precond_batch(c, NR, c->wr.p, c->wz.p, vs, c->scal.p, (int64_t)scalStride, itLocal, c->stop.p, &fz);
            for (int k2 = 0; k2 < NR; ++k2) {
                if (!(mgZs > 0)) k::launch_mg_rz(n, vec(c->wr, k2), vec(c->wz, k2), maskPtr, sck(k2), itLocal, sck(k2), stk(k2), s);
                k::launch_pcg_direction(n, vec(c->wz, k2), vec(c->wp, k2), vec(c->wx, k2), sck(k2), itLocal, stk(k2), s);
            }
        };
        hipGraphExec_t exec = nullptr;
        if (c->useGraph && checkEvery > 1) {
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
                bool ok = true;
                try {
                    for (int j = 0; j < checkEvery; ++j) enqueue(j);
                    for (int k2 = 0; k2 < NR; ++k2) k::launch_advance_base(stk(k2), checkEvery, s);
                } catch (...) { ok = false; }
                if (hipStreamEndCapture(s, &graph) != hipSuccess || !ok || !graph) { graph = nullptr; (void)hipGetLastError(); }
            } else (void)hipGetLastError();
            if (graph) {
                if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { exec = nullptr; (void)hipGetLastError(); }
                (void)hipGraphDestroy(graph);
            }
        }
        usedGraph = exec != nullptr;
        std::vector<double> hs;
        int it = 0, lastChecked = 0, nConv = 0;
        for (int k2 = 0; k2 < NR; ++k2) if (!(bb[k2] > 0)) { itConv[k2] = 0; ++nConv; }
        while (nConv < NR && it < maxit) {
            if (exec) { MFH_HIP(hipGraphLaunch(exec, s)); it += checkEvery; }
            else {
                const int itEnd = std::min(maxit, it + checkEvery);
                for (; it < itEnd; ++it) enqueue(it);
            }
            const size_t per = (size_t)(it - lastChecked + 1) * 4;
            hs.resize(per * NR);
            for (int k2 = 0; k2 < NR; ++k2)
                MFH_HIP(hipMemcpyAsync(hs.data() + per * k2, sck(k2) + (size_t)lastChecked * 4, per * sizeof(double), hipMemcpyDeviceToHost, s));
            MFH_HIP(hipStreamSynchronize(s));
            for (int k2 = 0; k2 < NR; ++k2) {
                if (itConv[k2] >= 0) continue;
                for (int q = lastChecked; q <= std::min(it, maxit); ++q) {
                    const double *sc = &hs[per * k2 + (size_t)(q - lastChecked) * 4];
                    const double rr = sc[2];
                    if (rr <= hstop[4 * k2]) { itConv[k2] = q; rrFinal[k2] = rr; ++nConv; break; }
                    if (!(rr == rr)) throw Error(MFH_ERR_NOT_CONVERGED, "PCG breakdown (NaN residual): K is not SPD on the free variables");
                    if (q < it && sc[1] < 0.0)
                        throw Error(MFH_ERR_NOT_CONVERGED, "PCG breakdown (p.Kp = " + std::to_string(sc[1]) + " < 0 at iteration " + std::to_string(q) +
                                                               ", residual^2 " + std::to_string(rr) + "): K is not positive definite on the free variables");
                    rrFinal[k2] = rr;
                }
            }
            lastChecked = it;
        }
        itRun = it;
        if (exec) (void)hipGraphExecDestroy(exec);
        solveMs = tsolve.stop();
    }
    // u = x + ubar (SparseMatrices.hh:2592-2605); true residual on the free variables: || mask(f - K u) || / ||b||
    double tr[6] = {0, 0, 0, 0, 0, 0};
    for (int k2 = 0; k2 < NR; ++k2) {
        if (masked && !c->solveHomogeneous) k::launch_scatter_values((int64_t)c->fixedVars.size(), c->dFixedIdx.p, c->dFixedVal.p, vec(c->wx, k2), n, s);
        if (bb[k2] > 0) {
            apply_operator(c, false, vec(c->wx, k2), c->wAp.p, nullptr);
            k::launch_axpby(n, 1.0, vec(c->wf, k2), -1.0, c->wAp.p, s);
            if (masked) k::launch_mask(n, c->dFixedMask.p, c->wAp.p, s);
            tr[k2] = device_dot(c, n, c->wAp.p, c->wAp.p);
        }
        if (io && io->uNodes) {             // dofToNodeField (LinearElasticity.hh:664-677) on the device, then one download per nodal field
            const int64_t nn = c->mesh.nNode * (int64_t)d;
            const double *src = vec(c->wx, k2);
            if (!c->dofForNode.empty()) {
                c->wNodeField.reserve((size_t)nn * 2);
                double *dst = c->wNodeField.p + (size_t)(k2 & 1) * nn;     // two halves: the gather of field k + 1 does not wait for the download of field k
                k::launch_pack_rows(c->mesh.nNode, d, device_dof_map(c), src, dst, s);
                src = dst;
            }
            MFH_HIP(hipMemcpyAsync(io->uNodes + (size_t)k2 * io->nodeStride, src, (size_t)nn * sizeof(double), hipMemcpyDeviceToHost, s));
        } else
            MFH_HIP(hipMemcpyAsync(u + (size_t)k2 * hostStride, vec(c->wx, k2), n * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    MFH_HIP(hipStreamSynchronize(s));
    mfh_solve_info gap[6];
    for (int k2 = 0; k2 < NR; ++k2) {
        mfh_solve_info li{};
        li.converged = itConv[k2] >= 0 ? 1 : 0;
        li.iterations = itConv[k2] >= 0 ? itConv[k2] : std::min(itRun, maxit);
        li.rel_residual = bb[k2] > 0 ? std::sqrt(rrFinal[k2] / bb[k2]) : 0.0;
        li.true_rel_residual = bb[k2] > 0 ? std::sqrt(tr[k2] / bb[k2]) : 0.0;
        li.solve_ms = solveMs;               // the batch's device time (shared by its right-hand sides)
        li.setup_ms = setupMs;
        li.used_graph = usedGraph ? 1 : 0;
        li.reserved = NR;
        if (infos) infos[k2] = li;
        gap[k2] = li;
    }
    for (int k2 = 0; k2 < NR; ++k2) check_residual_gap(gap[k2], rtol);
}

void box_corners(mfh_ctx *c, const double *mn, const double *mx, int relative, double *omn, double *omx) {
    const HostMesh &m = c->mesh;
    const int d = m.dim;
    if (!relative) {
        for (int a = 0; a < d; ++a) { omn[a] = mn[a]; omx[a] = mx[a]; }
        return;
    }
    // bounding box of the nodes on the host threads (minima / maxima: exact in any order; one thread over the 57.6 M nodes of a 119^3 grid took 0.13 s)
    double bmn[3] = {1e300, 1e300, 1e300}, bmx[3] = {-1e300, -1e300, -1e300};
    {
        const int nt = host_threads();
        std::vector<double> part((size_t)(nt + 1) * 6);
        for (int t = 0; t <= nt; ++t)
            for (int a = 0; a < 3; ++a) { part[(size_t)t * 6 + a] = 1e300; part[(size_t)t * 6 + 3 + a] = -1e300; }
        parallel_ranges(m.nNode, [&](int64_t lo, int64_t hi, int tid) {
            double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
            for (int64_t n = lo; n < hi; ++n)
                for (int a = 0; a < d; ++a) {
                    mn[a] = std::min(mn[a], m.nodePos[(size_t)n * d + a]);
                    mx[a] = std::max(mx[a], m.nodePos[(size_t)n * d + a]);
                }
            for (int a = 0; a < 3; ++a) {
                part[(size_t)tid * 6 + a] = std::min(part[(size_t)tid * 6 + a], mn[a]);
                part[(size_t)tid * 6 + 3 + a] = std::max(part[(size_t)tid * 6 + 3 + a], mx[a]);
            }
        });
        for (int t = 0; t <= nt; ++t)
            for (int a = 0; a < 3; ++a) { bmn[a] = std::min(bmn[a], part[(size_t)t * 6 + a]); bmx[a] = std::max(bmx[a], part[(size_t)t * 6 + 3 + a]); }
    }
    for (int a = 0; a < d; ++a) {   // BoundaryConditions.cc:310-316
        omn[a] = bmn[a] + mn[a] * (bmx[a] - bmn[a]);
        omx[a] = bmn[a] + mx[a] * (bmx[a] - bmn[a]);
    }
}


void dirichlet_vars(mfh_ctx *c, std::vector<int64_t> &vars, std::vector<double> &vals) {
    // m_getDirichletVarsAndValues (LinearElasticity.hh:1469-1518)
    const HostMesh &m = c->mesh;
    const int d = m.dim;
    if (c->dirMask.empty()) return;               // no Dirichlet condition was ever set (ensure_dirichlet_tables)
    // two boundary nodes can share a DoF only under a DoF map (periodic identification): without one every constrained node is the first of
    // its DoF and the DoF -> constraint table (230 MB filled with -1 at 119^3: 65 ms of every solve) is not needed
    const bool shared = !c->dofForNode.empty();
    std::vector<int32_t> constraintIndex(shared ? (size_t)c->nDoF : (size_t)0, -1);
    std::vector<int32_t> cDoF;
    std::vector<int64_t> cNode;                    // position of the constrained node in mesh.bdryNodes (the index of the tables)
    for (size_t bi = 0; bi < m.bdryNodes.size(); ++bi) {
        const int32_t bn = m.bdryNodes[bi];
        bool has = false;
        for (int a = 0; a < d; ++a) has |= c->dirMask[bi * d + a] != 0;
        if (!has) continue;
        const int32_t dof = dof_of(c, bn);
        if (!shared) {
            cDoF.push_back(dof);
            cNode.push_back((int64_t)bi);
        } else if (constraintIndex[dof] < 0) {
            constraintIndex[dof] = (int32_t)cDoF.size();
            cDoF.push_back(dof);
            cNode.push_back((int64_t)bi);
        } else {
            const int64_t o = cNode[constraintIndex[dof]];
            double diff = 0;
            bool cdiffer = false;
            for (int a = 0; a < d; ++a) {
                const double dd = c->dirVal[bi * d + a] - c->dirVal[(size_t)o * d + a];
                diff += dd * dd;
                cdiffer |= c->dirMask[bi * d + a] != c->dirMask[(size_t)o * d + a];
            }
            if (std::sqrt(diff) > 1e-10 || cdiffer) throw Error(MFH_ERR_INVALID, "Mismatched Dirichlet constraint on periodic DoF");
        }
    }
    if (c->op != MFH_OP_ELASTICITY) {
        // scalar PDE: the Dirichlet value is the first component of the "displacement" (Poisson.hh:57-60,80-84)
        for (size_t k = 0; k < cDoF.size(); ++k) {
            vars.push_back((int64_t)cDoF[k]);
            vals.push_back(c->dirVal[(size_t)cNode[k] * d]);
        }
        return;
    }
    for (size_t k = 0; k < cDoF.size(); ++k)
        for (int a = 0; a < d; ++a)
            if (c->dirMask[(size_t)cNode[k] * d + a]) {
                vars.push_back((int64_t)d * cDoF[k] + a);
                vals.push_back(c->dirVal[(size_t)cNode[k] * d + a]);
            }
}

int64_t pin_node(const mfh_ctx *c) {   // LinearElasticity.hh:1595-1609
    const HostMesh &m = c->mesh;
    for (int64_t i = 0; i < m.nNode; ++i)
        if (!m.isBdryNode[i]) return i;
    return 0;
}

void add_fixed(mfh_ctx *c, int64_t n, const int64_t *vars, const double *vals) {
    const int64_t nv = (int64_t)c->bs() * c->nDoF;
    if (c->hFixedMask.size() != (size_t)nv) parallel_assign(c->hFixedMask, (size_t)nv, (uint8_t)0);
    for (int64_t k = 0; k < n; ++k) {
        require(vars[k] >= 0 && vars[k] < nv, MFH_ERR_INVALID, "fixed variable index out of range");
        require(!c->hFixedMask[vars[k]], MFH_ERR_INVALID, "Variable already fixed.");   // SparseMatrices.hh:2433
        c->hFixedMask[vars[k]] = 1;
        c->fixedVars.push_back(vars[k]);
        const double v = vals ? vals[k] : 0.0;
        c->fixedVals.push_back(v);
        if (v != 0.0) c->anyFixedNonzero = true;
    }
    c->fixedUploaded = false;
    c->dinvValid = false;
    c->tl.valid = false;
    c->mg.valid = false;
}

// element material tensor D (flatLen x flatLen) from the geometry record
void elem_D(const mfh_ctx *c, const double *g, double *D) {
    const int d = c->mesh.dim, fl = flat_len(d);
    std::fill(D, D + fl * fl, 0.0);
    if (c->matKind == MAT_ISO) {
        const double lam = g[13], mu = g[14];
        for (int i = 0; i < d; ++i) {
            for (int j = 0; j < d; ++j) D[i * fl + j] = lam;
            D[i * fl + i] = lam + 2 * mu;
        }
        for (int k = d; k < fl; ++k) D[k * fl + k] = mu;
    } else if (c->matKind == MAT_ORTHO) {
        int idx = 0;
        for (int i = 0; i < d; ++i)
            for (int j = i; j < d; ++j, ++idx) D[i * fl + j] = D[j * fl + i] = g[13 + idx];
        for (int k = d; k < fl; ++k) D[k * fl + k] = g[13 + d * (d + 1) / 2 + k - d];
    } else {
        int idx = 0;
        for (int r = 0; r < fl; ++r)
            for (int cc = r; cc < fl; ++cc, ++idx) D[r * fl + cc] = D[cc * fl + r] = g[13 + idx];
    }
}

} // namespace mfhi
using namespace mfhi;

// =================================================================================================
extern "C" {

const char *mfh_version(void) { return "meshfem_hip 0.1 (gfx950)"; }

mfh_status mfh_create(int32_t device, mfh_ctx **out) {
    if (!out) return MFH_ERR_INVALID;
    *out = nullptr;
    mfh_ctx *c = new (std::nothrow) mfh_ctx();
    if (!c) return MFH_ERR_HIP;
    try {
        if (device == -1) {   // host-only context for CPU tests of the mesh / symbolic logic
            c->hostOnly = true;
            c->keepHostSymbolic = true;
            *out = c;
            return MFH_OK;
        }
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0)
            throw Error(MFH_ERR_HIP, "no HIP device available: libmeshfem_hip has no CPU fallback");
        require(device >= 0 && device < count, MFH_ERR_INVALID, "bad device ordinal");
        c->device = device;
        MFH_HIP(hipSetDevice(device));
        MFH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        mfh::device_arena_context_opened(device);
        c->arenaCounted = true;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->nCU = cus;
    } catch (const mfh::Error &e) {
        static thread_local std::string lastCreateError;
        lastCreateError = e.what();
        fprintf(stderr, "mfh_create: %s\n", e.what());
        mfh_status code = e.code;
        if (c->arenaCounted) mfh::device_arena_context_closed(c->device);
        delete c;
        return code;
    }
    *out = c;
    return MFH_OK;
}

void mfh_destroy(mfh_ctx *c) {
    if (!c) return;
    if (!c->hostOnly) (void)hipSetDevice(c->device);
    if (!c->hostOnly) (void)hipStreamSynchronize(c->stream);   // (a borrowed null stream too: the buffers below go back to the block cache)
    if (!c->hostOnly && c->dist.commStream) (void)hipStreamSynchronize(c->dist.commStream);
    hipStream_t s = c->stream;
    const bool own = c->ownStream;
    const bool counted = c->arenaCounted;
    const int dev = c->device;
    {
        // the context's streams are idle (synchronised above; a hierarchy's child context synchronises its own in its mfh_destroy): every device
        // buffer released from here on -- hierarchy levels, partition state, the context's own -- goes back to the cache without waiting for
        // anybody else's work (a device-wide wait would invalidate the stream captures of other host threads)
        mfh::PoolScope idle(nullptr, nullptr, c->hostOnly ? 0 : 2);
        for (auto &e : c->passEv) if (e) (void)hipEventDestroy(e);
        dist_detach(c);
        destroy_multigrid(c);
        for (auto &e : c->dist.ev) if (e) (void)hipEventDestroy(e);
        for (auto &P : c->dist.prof) for (auto &e : P.ev) if (e) (void)hipEventDestroy(e);
        if (c->dist.commStream && !c->dist.commStreamBorrowed) (void)hipStreamDestroy(c->dist.commStream);
        if (c->negHost) (void)hipHostFree(c->negHost);
        delete c;   // device buffers are freed while the stream is still alive
    }
    if (s && own) (void)hipStreamDestroy(s);
    // the arena keeps no more than the process has ever had live, and next to nothing once the last context of the device is gone (mfh_pool.cpp)
    if (counted) mfh::device_arena_context_closed(dev);
}

const char *mfh_last_error(const mfh_ctx *c) { return c ? c->err.c_str() : "null context"; }
void *mfh_stream(mfh_ctx *c) { return c ? (void *)c->stream : nullptr; }

/* the per-process device arena (mfh_pool.cpp) */
mfh_status mfh_device_cache_trim(void) {
    try { mfh::device_cache_trim(); } catch (...) { return MFH_ERR_HIP; }
    return MFH_OK;
}
mfh_status mfh_device_cache_stats(int32_t device, int64_t *cachedBytes, int64_t *blocks, int64_t *hits, int64_t *misses, int64_t *flushes) {
    mfh::device_cache_stats(device, cachedBytes, blocks, hits, misses, flushes);
    return MFH_OK;
}
// Bytes a context on a mesh of nElem simplices of that kind will hold at its peak (assembly + multigrid solve), and the share of it that is the
// value array of K. Measured on the generator's meshes (24 tets per hex / 8 triangles per quad patch; upper-triangle storage): 3.9 kB per
// quadratic tet of which 1.45 kB are K values (the rest must hold the peak of the symbolic phase by itself -- sort keys and temporaries next to
// the mesh tables: 2.3 kB per tet at 40 M tets -- since nothing shares the values' segment), 1.0 kB per linear tet (0.39), 1.3 kB per quadratic
// triangle (0.42), 0.4 kB per linear one (0.12). An estimate: the arena asks the driver for more when it is short and returns what stays free.
mfh_status mfh_context_bytes_estimate(int32_t dim, int32_t deg, int64_t nElem, int64_t *totalBytes, int64_t *kValueBytes) {
    if (!(dim == 2 || dim == 3) || !(deg == 1 || deg == 2) || nElem < 0) return MFH_ERR_INVALID;
    const double total = dim == 3 ? (deg == 2 ? 3900.0 : 1000.0) : (deg == 2 ? 1300.0 : 400.0);
    const double vals = dim == 3 ? (deg == 2 ? 1450.0 : 390.0) : (deg == 2 ? 420.0 : 120.0);
    if (totalBytes) *totalBytes = (int64_t)(total * (double)nElem);
    if (kValueBytes) *kValueBytes = (int64_t)(vals * (double)nElem * 1.07);      // (+7 %: a value array that misses its segment by a few MB would land in the other one)
    return MFH_OK;
}

static mfh_status reserve_split(int32_t device, int64_t bytes, int64_t valueBytes, int32_t async) {
    if (bytes < 0 || valueBytes < 0 || valueBytes > bytes) return MFH_ERR_INVALID;
    try {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) { (void)hipGetLastError(); return MFH_ERR_INVALID; }
        // the value array of K gets a segment of its own (the smaller request first: both may be under way at once): in one physical run with the
        // context's other buffers the assembly kernel sits at the slow end of its placement spread (docs/design/04_2_k_assemble_gather.md (xi))
        // the segment for everything else FIRST: the mesh tables and the symbolic phase need it at once, the value array is allocated at the END of the
        // symbolic phase -- and what a process takes beyond its first ~66 GB the driver clears while allocating (1 s for 83 GB at 119^3): with the
        // values' segment first, the symbolic phase stood waiting for the slow one while the fast one lay idle (round 6, first version: 0.86 instead of 0.38 s)
        bool ok = true;
        if (bytes - valueBytes > 0) ok = mfh::device_arena_reserve(device, (size_t)(bytes - valueBytes), async != 0, 1) && ok;
        if (valueBytes > 0) ok = mfh::device_arena_reserve(device, (size_t)valueBytes, async != 0, 2) && ok;
        if (!ok) return MFH_ERR_HIP;            // (synchronous reservations only: an asynchronous one that fails leaves the arena to ask again when it needs the memory)
    } catch (...) { return MFH_ERR_HIP; }
    return MFH_OK;
}

mfh_status mfh_device_reserve(int32_t device, int64_t bytes, int32_t async) {
    // without knowledge of the mesh: the value array's share of a quadratic 3D context (40 %, rounded up)
    return reserve_split(device, bytes, bytes >= ((int64_t)1 << 30) ? (int64_t)(0.43 * (double)bytes) : 0, async);
}

mfh_status mfh_device_reserve_for(int32_t device, int32_t dim, int32_t deg, int64_t nElem, int32_t async) {
    int64_t total = 0, vals = 0;
    const mfh_status st = mfh_context_bytes_estimate(dim, deg, nElem, &total, &vals);
    if (st != MFH_OK) return st;
    return reserve_split(device, total, vals, async);
}
mfh_status mfh_device_arena_stats(int32_t device, int64_t *out8) {
    if (!out8) return MFH_ERR_INVALID;
    mfh::device_arena_stats(device, out8);
    return MFH_OK;
}

mfh_status mfh_set_stream(mfh_ctx *c, void *stream) {
    MFH_TRY(c)
    require(c && !c->hostOnly, MFH_ERR_STATE, "no device context");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    MFH_HIP(hipStreamSynchronize(c->stream));
    if (c->dist.commStream) MFH_HIP(hipStreamSynchronize(c->dist.commStream));
    {
        // the linear level of a multigrid hierarchy is a context of its own that launches on a copy of the parent's stream: drop the hierarchy
        // (it is rebuilt on the new stream by the next solve that wants it) rather than leave it on a stream that may be gone --
        // and drop it BEFORE the old stream is destroyed, with the streams known idle: the buffers it releases must not wait on a handle
        // that no longer exists (use-after-destroy of the stream in the arena's wait)
        mfh::PoolScope idle(nullptr, nullptr, 2);
        destroy_multigrid(c);
    }
    hipStream_t old = c->stream;
    const bool ownOld = c->ownStream;
    c->stream = (hipStream_t)stream;   // nullptr = the legacy default stream
    c->ownStream = false;
    if (ownOld && old) (void)hipStreamDestroy(old);
    MFH_CATCH(c)
}

// ---------------------------------------------------------------- mesh
mfh_status mfh_mesh_build(mfh_ctx *c, int32_t dim, int32_t deg, int64_t nElem, int64_t nVert, const int32_t *elemVerts,
                          const double *vertPos) {
    MFH_TRY(c)
    require(c && elemVerts && vertPos, MFH_ERR_INVALID, "null argument");
    if (!c->hostOnly) MFH_HIP(hipSetDevice(c->device));
    c->haveMesh = false;              // a build that throws half way (bad indices, non-manifold input) leaves a context without a mesh, not with half of one
    bool deviceTables = false;
    build_fem_mesh(c->mesh, dim, deg, nElem, nVert, elemVerts, vertPos, !c->hostOnly && c->topologyDevice, c->stream, &c->dElemNodes, &c->dVertPos, &deviceTables);
    const bool timing = getenv("MFH_MESH_TIMING") != nullptr;
    double t0 = now_ms();
    upload_mesh(c, deviceTables);
    if (timing) { if (!c->hostOnly) (void)hipStreamSynchronize(c->stream); fprintf(stderr, "[mesh build] %-30s %8.2f ms\n", "upload + tables", now_ms() - t0); t0 = now_ms(); }
    if (!c->hostOnly) ensure_geometry(c);   // throws on negative volumes like the Simulator ctor
    if (timing) fprintf(stderr, "[mesh build] %-30s %8.2f ms\n", "embedding kernel + check", now_ms() - t0);
    MFH_CATCH(c)
}

mfh_status mfh_mesh_set(mfh_ctx *c, int32_t dim, int32_t deg, int64_t nElem, int64_t nNode, int64_t nOwned,
                        const int32_t *elemNodes, const double *nodePos) {
    MFH_TRY(c)
    require(c && elemNodes && nodePos, MFH_ERR_INVALID, "null argument");
    require((dim == 2 || dim == 3) && (deg == 1 || deg == 2), MFH_ERR_INVALID, "dim must be 2/3 and deg 1/2");
    require(nElem > 0 && nNode > 0 && nOwned > 0 && nOwned <= nNode, MFH_ERR_INVALID, "bad mesh sizes");
    HostMesh &m = c->mesh;
    c->haveMesh = false;
    m = HostMesh();
    m.dim = dim; m.deg = deg; m.npe = nodes_per_elem(dim, deg); m.npbe = nodes_per_bdry_elem(dim, deg);
    m.nElem = nElem; m.nNode = nNode; m.nVert = nNode; m.nOwned = nOwned;
    m.elemNodes.assign(elemNodes, elemNodes + nElem * m.npe);
    for (int64_t k = 0; k < nElem * m.npe; ++k)
        require(elemNodes[k] >= 0 && elemNodes[k] < nNode, MFH_ERR_INVALID, "Bad node index encountered.");
    m.nodePos.assign(nodePos, nodePos + nNode * dim);
    m.vertPos = m.nodePos;
    m.isBdryNode.assign((size_t)nNode, 0);
    upload_mesh(c, false);
    if (!c->hostOnly) ensure_geometry(c);
    MFH_CATCH(c)
}

mfh_status mfh_mesh_update_vertices(mfh_ctx *c, const double *vertPos) {
    MFH_TRY(c)
    require(c && c->haveMesh && vertPos, MFH_ERR_STATE, "no mesh set");
    HostMesh &m = c->mesh;
    // row-partitioned contexts come from mfh_mesh_set: the caller hands over the positions of ALL local nodes, the halo nodes with the
    // coordinates their owners hold (a shape-optimisation step moves the whole mesh on every rank)
    require(m.nOwned == m.nNode || !m.hasTopology, MFH_ERR_UNSUPPORTED, "vertex updates on a partitioned mesh need the caller's node table (mfh_mesh_set)");
    if (m.hasTopology) {
        m.vertPos.assign(vertPos, vertPos + (size_t)m.nVert * m.dim);
        compute_node_positions(m);
        compute_boundary_geometry(m, m.vertPos.data());
    } else {
        // mfh_mesh_set: the caller owns the node table; vertPos then holds every node (nVert == nNode)
        m.vertPos.assign(vertPos, vertPos + (size_t)m.nNode * m.dim);
        m.nodePos = m.vertPos;
    }
    if (!c->hostOnly) {
        require_device(c);
        MFH_HIP(hipSetDevice(c->device));
        c->dVertPos.upload(m.nodePos, c->stream);
    }
    // topology, DoF map, sparsity pattern, gather lists and the matrix-free lists only depend on connectivity: kept
    c->autoStretch = -1.0;
    c->geoValid = false;
    c->hGeoValid = false;
    invalidate_matrix(c);
    MFH_CATCH(c)
}

mfh_status mfh_mesh_sizes(const mfh_ctx *c, int64_t *nElem, int64_t *nNode, int64_t *nVert, int64_t *nBdryElem, int64_t *nBdryNode,
                          int32_t *npe, int32_t *npbe) {
    if (!c || !c->haveMesh) return MFH_ERR_STATE;
    const HostMesh &m = c->mesh;
    if (nElem) *nElem = m.nElem;
    if (nNode) *nNode = m.nNode;
    if (nVert) *nVert = m.nVert;
    if (nBdryElem) *nBdryElem = m.nBE();
    if (nBdryNode) *nBdryNode = (int64_t)m.bdryNodes.size();
    if (npe) *npe = m.npe;
    if (npbe) *npbe = m.npbe;
    return MFH_OK;
}

mfh_status mfh_mesh_get_elem_nodes(const mfh_ctx *c, int32_t *out) {
    if (!c || !c->haveMesh || !out) return MFH_ERR_STATE;
    std::copy(c->mesh.elemNodes.begin(), c->mesh.elemNodes.end(), out);
    return MFH_OK;
}
mfh_status mfh_mesh_get_node_positions(const mfh_ctx *c, double *out) {
    if (!c || !c->haveMesh || !out) return MFH_ERR_STATE;
    std::copy(c->mesh.nodePos.begin(), c->mesh.nodePos.end(), out);
    return MFH_OK;
}
mfh_status mfh_mesh_get_boundary_elem_nodes(const mfh_ctx *c, int32_t *out) {
    if (!c || !c->haveMesh || !out) return MFH_ERR_STATE;
    std::copy(c->mesh.bdryElemNodes.begin(), c->mesh.bdryElemNodes.end(), out);
    return MFH_OK;
}
mfh_status mfh_mesh_get_boundary_elem_parents(const mfh_ctx *c, int32_t *out) {
    if (!c || !c->haveMesh || !out) return MFH_ERR_STATE;
    std::copy(c->mesh.bdryParent.begin(), c->mesh.bdryParent.end(), out);
    return MFH_OK;
}
mfh_status mfh_mesh_get_boundary_elem_internal(const mfh_ctx *c, uint8_t *out) {
    if (!c || !c->haveMesh || !out) return MFH_ERR_STATE;
    std::copy(c->mesh.bdryInternal.begin(), c->mesh.bdryInternal.end(), out);
    return MFH_OK;
}
mfh_status mfh_mesh_get_boundary_nodes(const mfh_ctx *c, int32_t *out) {
    if (!c || !c->haveMesh || !out) return MFH_ERR_STATE;
    std::copy(c->mesh.bdryNodes.begin(), c->mesh.bdryNodes.end(), out);
    return MFH_OK;
}
mfh_status mfh_mesh_get_boundary_elem_geometry(const mfh_ctx *c, double *volume, double *normal) {
    if (!c || !c->haveMesh) return MFH_ERR_STATE;
    if (volume) std::copy(c->mesh.bdryVol.begin(), c->mesh.bdryVol.end(), volume);
    if (normal) std::copy(c->mesh.bdryNormal.begin(), c->mesh.bdryNormal.end(), normal);
    return MFH_OK;
}
mfh_status mfh_mesh_get_elem_volumes(mfh_ctx *c, double *out) {
    MFH_TRY(c)
    require(c && out, MFH_ERR_INVALID, "null argument");
    const std::vector<double> &g = host_geo(c);
    for (int64_t e = 0; e < c->mesh.nElem; ++e) out[e] = g[(size_t)e * c->geoStride + 12];
    MFH_CATCH(c)
}

// ---------------------------------------------------------------- materials
mfh_status mfh_material_isotropic(mfh_ctx *c, double E, double nu) {
    MFH_TRY(c)
    require(c, MFH_ERR_INVALID, "null context");
    set_isotropic(c, E, nu);
    MFH_CATCH(c)
}

mfh_status mfh_material_const(mfh_ctx *c, const double *D) {
    MFH_TRY(c)
    require(c && D && c->haveMesh, MFH_ERR_STATE, "set the mesh before the material");
    const int d = c->dim(), fl = flat_len(d);
    c->matParams.clear();
    // a tensor with the orthotropic pattern in the coordinate axes (no normal-shear coupling, diagonal shear block; every
    // isotropic / orthotropic base material of the CLIs) takes the compact orthotropic record and kernels
    bool ortho = true;
    for (int r = 0; r < fl && ortho; ++r)
        for (int cc = r + 1; cc < fl; ++cc)
            if (cc >= d && D[r * fl + cc] != 0.0) { ortho = false; break; }
    if (ortho) {
        for (int r = 0; r < d; ++r)
            for (int cc = r; cc < d; ++cc) c->matParams.push_back(D[r * fl + cc]);
        for (int k2 = d; k2 < fl; ++k2) c->matParams.push_back(D[k2 * fl + k2]);
        c->matMode = 5; c->matKind = MAT_ORTHO;
    } else {
        for (int r = 0; r < fl; ++r)
            for (int cc = r; cc < fl; ++cc) c->matParams.push_back(D[r * fl + cc]);   // upper triangle (_MajorSymmetry)
        c->matMode = 2; c->matKind = MAT_GENERAL;
    }
    c->geoValid = false; c->hGeoValid = false;
    invalidate_matrix(c);
    MFH_CATCH(c)
}

mfh_status mfh_material_iso_field(mfh_ctx *c, const double *E, const double *nu) {
    MFH_TRY(c)
    require(c && E && nu && c->haveMesh, MFH_ERR_STATE, "set the mesh before the material");
    const int64_t n = c->mesh.nElem;
    c->matParams.assign((size_t)2 * n, 0.0);
    std::copy(E, E + n, c->matParams.begin());
    std::copy(nu, nu + n, c->matParams.begin() + n);
    c->matMode = 1; c->matKind = MAT_ISO;
    c->geoValid = false; c->hGeoValid = false;
    invalidate_matrix(c);
    MFH_CATCH(c)
}

mfh_status mfh_material_ortho_field(mfh_ctx *c, const double *params) {
    MFH_TRY(c)
    require(c && params && c->haveMesh, MFH_ERR_STATE, "set the mesh before the material");
    const int np = c->dim() == 3 ? 9 : 4;
    c->matParams.assign(params, params + (size_t)np * c->mesh.nElem);
    c->matMode = 3; c->matKind = MAT_ORTHO;   // compact record: 6 + 3 (3D) / 3 + 1 (2D) stiffnesses instead of the 21 / 6 of a general D
    c->geoValid = false; c->hGeoValid = false;
    invalidate_matrix(c);
    MFH_CATCH(c)
}

mfh_status mfh_material_tensor_field(mfh_ctx *c, const double *D) {
    MFH_TRY(c)
    require(c && D && c->haveMesh, MFH_ERR_STATE, "set the mesh before the material");
    const int fl = flat_len(c->dim());
    c->matParams.assign(D, D + (size_t)fl * fl * c->mesh.nElem);
    c->matMode = 4; c->matKind = MAT_GENERAL;
    c->geoValid = false; c->hGeoValid = false;
    invalidate_matrix(c);
    MFH_CATCH(c)
}

mfh_status mfh_material_get(mfh_ctx *c, int64_t elem, double *D) {
    MFH_TRY(c)
    require(c && D && c->haveMesh && elem >= 0 && elem < c->mesh.nElem, MFH_ERR_INVALID, "bad element");
    const std::vector<double> &g = host_geo(c);
    elem_D(c, &g[(size_t)elem * c->geoStride], D);
    MFH_CATCH(c)
}

// ---------------------------------------------------------------- DoF map
mfh_status mfh_dof_map(mfh_ctx *c, const int32_t *dofForNode, int64_t nDoF) {
    MFH_TRY(c)
    require(c && c->haveMesh, MFH_ERR_STATE, "no mesh set");
    c->nOwnedDoFSet = -1;
    if (!dofForNode) {
        c->dofForNode.clear();
        c->nDoF = c->mesh.nNode;
        // removePeriodicConditions (LinearElasticity.hh:874-879) also clears the isInternal flags
        std::fill(c->mesh.bdryInternal.begin(), c->mesh.bdryInternal.end(), (uint8_t)0);
    } else {
        require(c->mesh.nOwned == c->mesh.nNode, MFH_ERR_UNSUPPORTED, "a DoF map on a partitioned mesh says which DoFs are this rank's rows: mfh_dof_map_partitioned");
        require(nDoF > 0 && nDoF <= c->mesh.nNode, MFH_ERR_INVALID, "bad nDoF");
        for (int64_t n = 0; n < c->mesh.nNode; ++n)
            require(dofForNode[n] >= 0 && dofForNode[n] < nDoF, MFH_ERR_INVALID, "DoF index out of range");
        c->dofForNode.assign(dofForNode, dofForNode + c->mesh.nNode);
        c->nDoF = nDoF;
    }
    invalidate_symbolic(c);
    clear_fixed(c);
    MFH_CATCH(c)
}

// Row-partitioned context with a DoF map (periodic cell problems across GPUs): the rank's local nodes map to local DoFs numbered owned-first
// (rows of K = the first nOwnedDoF DoFs), halo DoFs after them grouped by owner; mfh_dist_setup then takes its lists in DoF numbers.
mfh_status mfh_dof_map_partitioned(mfh_ctx *c, const int32_t *dofForNode, int64_t nDoF, int64_t nOwnedDoF) {
    MFH_TRY(c)
    require(c && c->haveMesh && dofForNode, MFH_ERR_STATE, "no mesh set");
    require(!c->mesh.hasTopology, MFH_ERR_UNSUPPORTED, "partitioned contexts are built by mfh_mesh_set");
    require(nDoF > 0 && nDoF <= c->mesh.nNode && nOwnedDoF > 0 && nOwnedDoF <= nDoF, MFH_ERR_INVALID, "bad DoF counts");
    for (int64_t n = 0; n < c->mesh.nNode; ++n)
        require(dofForNode[n] >= 0 && dofForNode[n] < nDoF, MFH_ERR_INVALID, "DoF index out of range");
    c->dofForNode.assign(dofForNode, dofForNode + c->mesh.nNode);
    c->nDoF = nDoF;
    c->nOwnedDoFSet = nOwnedDoF;
    invalidate_symbolic(c);
    clear_fixed(c);
    dist_detach(c);                  // exchange lists are in DoF numbers: mfh_dist_setup must run (again)
    MFH_CATCH(c)
}

mfh_status mfh_apply_periodic_conditions(mfh_ctx *c, double eps, int64_t *nDoF) {
    MFH_TRY(c)
    require(c && c->haveMesh, MFH_ERR_STATE, "no mesh set");
    std::vector<int32_t> dof;
    int64_t nd = 0;
    periodic_dof_map(c->mesh, eps, dof, nd, c->mesh.bdryInternal, c->periodicIgnoreMismatch, c->periodicIgnoreDims);
    if (nDoF) *nDoF = nd;
    // the same identification as before (e.g. re-applied after a vertex update that left the cell faces alone):
    // pattern, lists and fixed variables stay valid
    if (nd == c->nDoF && dof == c->dofForNode) return MFH_OK;
    c->dofForNode.swap(dof);
    c->nDoF = nd;
    invalidate_symbolic(c);
    clear_fixed(c);
    MFH_CATCH(c)
}

mfh_status mfh_get_dof_map(const mfh_ctx *c, int32_t *dofForNode, int64_t *nDoF) {
    if (!c || !c->haveMesh) return MFH_ERR_STATE;
    if (nDoF) *nDoF = c->nDoF;
    if (dofForNode)
        for (int64_t n = 0; n < c->mesh.nNode; ++n) dofForNode[n] = dof_of(c, n);
    return MFH_OK;
}

// ---------------------------------------------------------------- assembly
mfh_status mfh_symbolic(mfh_ctx *c, int32_t withScatterMap) {
    MFH_TRY(c)
    require(c && c->haveMesh, MFH_ERR_STATE, "no mesh set");
    ensure_symbolic(c, withScatterMap != 0);
    MFH_CATCH(c)
}

mfh_status mfh_symbolic_sizes(const mfh_ctx *c, int64_t *nChunk, int64_t *nContrib, int32_t *chunkSlots, int32_t *maxRowLen) {
    if (!c || !c->symValid) return MFH_ERR_STATE;
    if (nChunk) *nChunk = c->sym.nChunk();
    if (nContrib) *nContrib = c->sym.contribPtr.empty() ? 0 : c->sym.contribPtr.back();
    if (chunkSlots) *chunkSlots = c->sym.chunkSlots;
    if (maxRowLen) *maxRowLen = c->sym.maxRowLen;
    return MFH_OK;
}

mfh_status mfh_symbolic_get(const mfh_ctx *c, int32_t *rowPtr, int32_t *colIdx, int32_t *chunkRow, int64_t *contribPtr,
                            uint32_t *contribCode, uint16_t *contribSlot, int32_t *scatterSlot) {
    if (!c || !c->symValid) return MFH_ERR_STATE;
    const Symbolic &S = c->sym;
    if (rowPtr) std::copy(S.rowPtr.begin(), S.rowPtr.end(), rowPtr);
    if (colIdx) std::copy(S.colIdx.begin(), S.colIdx.end(), colIdx);
    if (chunkRow) std::copy(S.chunkRow.begin(), S.chunkRow.end(), chunkRow);
    if (contribPtr) std::copy(S.contribPtr.begin(), S.contribPtr.end(), contribPtr);
    if (contribCode || contribSlot || scatterSlot) {
        if (S.contribCode.empty()) return MFH_ERR_STATE;   // needs option keep_host_symbolic
        if (contribCode) std::copy(S.contribCode.begin(), S.contribCode.end(), contribCode);
        if (contribSlot) std::copy(S.contribSlot.begin(), S.contribSlot.end(), contribSlot);
        if (scatterSlot) {
            if (S.scatterSlot.empty()) return MFH_ERR_STATE;
            std::copy(S.scatterSlot.begin(), S.scatterSlot.end(), scatterSlot);
        }
    }
    return MFH_OK;
}

mfh_status mfh_assemble(mfh_ctx *c, int32_t mode) {
    MFH_TRY(c)
    require(c && c->haveMesh, MFH_ERR_STATE, "no mesh set");
    require(mode == MFH_ASSEMBLE_GATHER || mode == MFH_ASSEMBLE_ATOMIC, MFH_ERR_INVALID, "bad assembly mode");
    require(!(c->deterministic && mode == MFH_ASSEMBLE_ATOMIC), MFH_ERR_UNSUPPORTED, "option deterministic: the global-atomic assembly variant is not reproducible");
    if (c->alwaysReembed) { c->geoValid = false; c->hGeoValid = false; }
    ensure_symbolic(c, mode == MFH_ASSEMBLE_ATOMIC);
    RoctxRange range("Assemble System");
    // embedding kernel and assembly kernel back to back, one synchronisation for both (their times and the embedding's
    // counters of inverted elements are read afterwards: an inverted mesh is reported after K has been written, and left invalid)
    ensure_geometry(c, true);
    ensure_pass_events(c);
    MFH_HIP(hipEventRecord(c->passEv[2], c->stream));
    run_assembly(c, mode);
    MFH_HIP(hipEventRecord(c->passEv[3], c->stream));
    if (c->geoPending) finish_geometry(c);
    else MFH_HIP(hipStreamSynchronize(c->stream));
    float ms = 0;
    MFH_HIP(hipEventElapsedTime(&ms, c->passEv[2], c->passEv[3]));
    c->timing.assemble_ms = ms;
    if (mode == MFH_ASSEMBLE_GATHER && c->placementTrials > 0 && c->placementGen != c->valsGen) placement_trials(c);
    c->assembled = true;
    c->dinvValid = false;
    MFH_CATCH(c)
}

mfh_status mfh_matrix_info(const mfh_ctx *c, int64_t *nBlockRows, int64_t *nBlockCols, int64_t *nnzBlocks) {
    if (!c || !c->symValid) return MFH_ERR_STATE;
    if (nBlockRows) *nBlockRows = c->sym.nRows;
    if (nBlockCols) *nBlockCols = c->sym.nCols;
    if (nnzBlocks) *nnzBlocks = c->sym.nnzb + (c->upperOnly ? c->sym.nMirror : 0);   // blocks of K (mfh_export_bsr), whatever the storage
    return MFH_OK;
}

mfh_status mfh_matrix_storage(const mfh_ctx *c, int32_t *upperOnly, int64_t *storedBlocks) {
    if (!c || !c->symValid) return MFH_ERR_STATE;
    if (upperOnly) *upperOnly = c->upperOnly ? 1 : 0;
    if (storedBlocks) *storedBlocks = c->sym.nnzb;
    return MFH_OK;
}

// SPSDSystem(K) for a caller-supplied matrix (SparseMatrices.hh:2332-2348): upper-triangle triplets (repeated
// entries are summed, like sumRepeated) become the full symmetric CSR the SpMV / PCG kernels run on (1x1 blocks).
mfh_status mfh_matrix_set_upper_triplets(mfh_ctx *c, int64_t n, int64_t nnz, const uint64_t *ti, const uint64_t *tj, const double *tv) {
    MFH_TRY(c)
    require(c && n > 0 && nnz >= 0 && (nnz == 0 || (ti && tj && tv)), MFH_ERR_INVALID, "bad matrix arguments");
    require(n < (int64_t)1 << 31, MFH_ERR_UNSUPPORTED, "matrix too large for 32-bit indices");
    // rowPtr / slots are 32-bit: the mirrored matrix has at most 2 nnz entries
    require(nnz < ((int64_t)1 << 30), MFH_ERR_UNSUPPORTED, "matrix has too many non-zeros for 32-bit slots (2 nnz must stay below 2^31)");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    struct Ent { int32_t r, c; double v; };
    std::vector<Ent> ent;
    ent.reserve((size_t)nnz * 2);
    for (int64_t k = 0; k < nnz; ++k) {
        require(ti[k] <= tj[k] && tj[k] < (uint64_t)n, MFH_ERR_INVALID, "triplet outside the upper triangle");
        ent.push_back({(int32_t)ti[k], (int32_t)tj[k], tv[k]});
        if (ti[k] != tj[k]) ent.push_back({(int32_t)tj[k], (int32_t)ti[k], tv[k]});
    }
    std::stable_sort(ent.begin(), ent.end(), [](const Ent &a, const Ent &b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
    Symbolic &S = c->sym;
    S = Symbolic();
    S.nRows = S.nCols = n;
    S.rowPtr.assign((size_t)n + 1, 0);
    std::vector<double> vals;
    for (size_t k = 0; k < ent.size();) {
        size_t e = k;
        double v = 0;
        while (e < ent.size() && ent[e].r == ent[k].r && ent[e].c == ent[k].c) v += ent[e++].v;
        S.colIdx.push_back(ent[k].c);
        vals.push_back(v);
        S.rowPtr[(size_t)ent[k].r + 1]++;
        k = e;
    }
    for (int64_t r = 0; r < n; ++r) S.rowPtr[(size_t)r + 1] += S.rowPtr[r];
    S.nnzb = (int64_t)vals.size();
    S.chunkSlots = c->chunkSlots;
    S.chunkRow.assign(1, 0);
    S.contribPtr.assign(1, 0);
    // a chunk holds whole rows: it grows to the longest row, up to what 64 KB of LDS partials allow (8000 entries;
    // the CHOLMOD path has no such limit -- rows denser than that need a different kernel and are rejected)
    int32_t maxRow = 0;
    for (int64_t r = 0; r < n; ++r) maxRow = std::max(maxRow, S.rowPtr[r + 1] - S.rowPtr[r]);
    S.maxRowLen = maxRow;
    S.spmvChunkSlots = std::max(std::max(512, c->chunkSlots), std::min(maxRow, 8000));
    S.spmvChunkRow.assign(1, 0);
    for (int64_t r = 0; r < n;) {
        require(S.rowPtr[r + 1] - S.rowPtr[r] <= S.spmvChunkSlots, MFH_ERR_UNSUPPORTED,
                "matrix row with more than 8000 entries: longer than the LDS partial-sum buffer of the SpMV kernel");
        const int32_t s0 = S.rowPtr[r];
        int64_t r2 = r + 1;
        while (r2 < n && S.rowPtr[r2 + 1] - s0 <= S.spmvChunkSlots) ++r2;
        S.spmvChunkRow.push_back((int32_t)r2);
        r = r2;
    }
    c->haveMesh = false;
    c->external = true;
    c->op = MFH_OP_ELASTICITY;
    c->dofForNode.clear();
    c->nDoF = n;
    c->dRowPtr.upload(S.rowPtr, c->stream);
    c->dColIdx.upload(S.colIdx, c->stream);
    c->dSpmvChunkRow.upload(S.spmvChunkRow, c->stream);
    vals.resize(((vals.size() + 63) / 64) * 64, 0.0);      // tiled layout with 1 component per entry = plain array
    c->dVals.upload(vals, c->stream);
    c->symValid = true;
    ++c->listsGen;
    c->assembled = true;
    c->dinvValid = false;
    c->tl.valid = false;
    clear_fixed(c);
    MFH_CATCH(c)
}

mfh_status mfh_export_bsr(mfh_ctx *c, int32_t *rowPtr, int32_t *colIdx, double *vals) {
    MFH_TRY(c)
    require(c && c->assembled, MFH_ERR_STATE, "matrix not assembled");
    ensure_host_colidx(c);
    const Symbolic &S = c->sym;
    if (!c->upperOnly) {
        if (rowPtr) std::copy(S.rowPtr.begin(), S.rowPtr.end(), rowPtr);
        if (colIdx) std::copy(S.colIdx.begin(), S.colIdx.end(), colIdx);
        if (vals && S.nnzb) {
            const int nb = c->bs() * c->bs();
            DBuf<double> aos;
            aos.alloc((size_t)S.nnzb * nb);
            k::launch_untile_vals(c->bs(), S.nnzb, c->dVals.p, aos.p, c->stream);
            aos.download(vals, (size_t)S.nnzb * nb, c->stream);
        }
    } else {
        // upper-triangle storage: the export is K all the same. Row r = the transposes of the stored blocks (q, r), q < r, in
        // ascending q (= ascending column), then the stored blocks (r, c >= r).
        const int d = c->bs(), nb = d * d;
        const int64_t nR = S.nRows;
        std::vector<int64_t> full((size_t)nR + 1, 0);
        for (int64_t r = 0; r < nR; ++r) {
            full[(size_t)r + 1] += S.rowPtr[r + 1] - S.rowPtr[r];
            for (int32_t q = S.rowPtr[r]; q < S.rowPtr[r + 1]; ++q) {
                const int64_t col = S.colIdx[q];
                if (col > r && col < nR) ++full[(size_t)col + 1];
            }
        }
        for (int64_t r = 0; r < nR; ++r) full[(size_t)r + 1] += full[(size_t)r];
        require(full[(size_t)nR] == S.nnzb + S.nMirror, MFH_ERR_STATE, "inconsistent mirror count");
        require(full[(size_t)nR] <= 2147483647LL, MFH_ERR_UNSUPPORTED, "more than 2^31 blocks: export the stored triangle (mfh_export_upper_triplets)");
        if (rowPtr) for (int64_t r = 0; r <= nR; ++r) rowPtr[r] = (int32_t)full[(size_t)r];
        std::vector<double> sv;
        if (vals && S.nnzb) {
            sv.resize((size_t)S.nnzb * nb);
            DBuf<double> aos;
            aos.alloc(sv.size());
            k::launch_untile_vals(d, S.nnzb, c->dVals.p, aos.p, c->stream);
            aos.download(sv.data(), sv.size(), c->stream);
        }
        if (colIdx || vals) {
            std::vector<int64_t> cur(full.begin(), full.end() - 1);
            for (int64_t r = 0; r < nR; ++r)        // transposed entries first: rows are visited in ascending order
                for (int32_t q = S.rowPtr[r]; q < S.rowPtr[r + 1]; ++q) {
                    const int64_t col = S.colIdx[q];
                    if (!(col > r && col < nR)) continue;
                    const int64_t at = cur[(size_t)col]++;
                    if (colIdx) colIdx[at] = (int32_t)r;
                    if (vals)
                        for (int a2 = 0; a2 < d; ++a2)
                            for (int b2 = 0; b2 < d; ++b2) vals[(size_t)at * nb + a2 * d + b2] = sv[(size_t)q * nb + b2 * d + a2];
                }
            for (int64_t r = 0; r < nR; ++r)
                for (int32_t q = S.rowPtr[r]; q < S.rowPtr[r + 1]; ++q) {
                    const int64_t at = cur[(size_t)r]++;
                    if (colIdx) colIdx[at] = S.colIdx[q];
                    if (vals) std::copy(sv.begin() + (size_t)q * nb, sv.begin() + (size_t)(q + 1) * nb, vals + (size_t)at * nb);
                }
        }
    }
    MFH_CATCH(c)
}

mfh_status mfh_export_upper_triplets(mfh_ctx *c, uint64_t *oi, uint64_t *oj, double *ov, uint64_t *nnz) {
    MFH_TRY(c)
    require(c && c->assembled && nnz, MFH_ERR_STATE, "matrix not assembled");
    require(c->sym.nRows == c->sym.nCols, MFH_ERR_STATE, "triplet export needs a square matrix");
    ensure_host_colidx(c);
    const Symbolic &S = c->sym;
    const int d = c->bs(), nb = d * d;
    if (!(oi && oj && ov)) {
        // capacity query: the structural entries of the upper triangle (an upper bound of the count: exact zeros are pruned below)
        uint64_t cap = 0;
        for (int64_t r = 0; r < S.nRows; ++r)
            for (int32_t q = S.rowPtr[r]; q < S.rowPtr[r + 1]; ++q) {
                const int64_t col = S.colIdx[q];
                if (col >= r) cap += col == r ? (uint64_t)d * (d + 1) / 2 : (uint64_t)nb;
            }
        *nnz = cap;
        return MFH_OK;
    }
    std::vector<double> vals((size_t)S.nnzb * nb);
    {
        DBuf<double> aos;
        aos.alloc(vals.size());
        k::launch_untile_vals(d, S.nnzb, c->dVals.p, aos.p, c->stream);
        aos.download(vals.data(), vals.size(), c->stream);
    }
    // (col, row) order without a comparison sort -- sumRepeated's own counting sort by column (SparseMatrices.hh:280-374): count the
    // entries of every scalar column, then visit the scalar ROWS in ascending order, so that each column receives its rows ascending
    const int64_t n = (int64_t)S.nRows * d;
    std::vector<uint64_t> colPtr((size_t)n + 1, 0);
    for (int64_t r = 0; r < S.nRows; ++r)
        for (int32_t q = S.rowPtr[r]; q < S.rowPtr[r + 1]; ++q) {
            const int64_t col = S.colIdx[q];
            if (col < r) continue;
            for (int a2 = 0; a2 < d; ++a2)
                for (int b2 = 0; b2 < d; ++b2)
                    if (r * d + a2 <= col * d + b2 && vals[(size_t)q * nb + a2 * d + b2] != 0.0) ++colPtr[(size_t)(col * d + b2) + 1];   // pruneTol = 0 (:370-373)
        }
    for (int64_t j2 = 0; j2 < n; ++j2) colPtr[(size_t)j2 + 1] += colPtr[(size_t)j2];
    require(*nnz >= colPtr[(size_t)n], MFH_ERR_INVALID, "triplet buffers too small");
    *nnz = colPtr[(size_t)n];
    for (int64_t r = 0; r < S.nRows; ++r)
        for (int a2 = 0; a2 < d; ++a2)
            for (int32_t q = S.rowPtr[r]; q < S.rowPtr[r + 1]; ++q) {
                const int64_t col = S.colIdx[q];
                if (col < r) continue;
                for (int b2 = 0; b2 < d; ++b2) {
                    const uint64_t gi = (uint64_t)r * d + a2, gj = (uint64_t)col * d + b2;
                    const double v = vals[(size_t)q * nb + a2 * d + b2];
                    if (gi <= gj && v != 0.0) {
                        const uint64_t at = colPtr[gj]++;
                        oi[at] = gi; oj[at] = gj; ov[at] = v;
                    }
                }
            }
    MFH_CATCH(c)
}

mfh_status mfh_element_stiffness(mfh_ctx *c, int64_t first, int64_t count, double *Ke) {
    MFH_TRY(c)
    require(c && c->haveMesh && Ke, MFH_ERR_STATE, "no mesh set");
    require(first >= 0 && count > 0 && first + count <= c->mesh.nElem, MFH_ERR_INVALID, "bad element range");
    ensure_geometry(c);
    const size_t ks = (size_t)c->mesh.npe * c->bs();
    DBuf<double> out;
    out.alloc((size_t)count * ks * ks);
    k::AsmArgs a = asm_args(c);
    k::launch_element_stiffness(a, first, count, out.p, c->stream);
    out.download(Ke, out.n, c->stream);
    MFH_CATCH(c)
}

// ---------------------------------------------------------------- constrained solve
mfh_status mfh_clear_fixed(mfh_ctx *c) {
    MFH_TRY(c)
    require(c && (c->haveMesh || c->external), MFH_ERR_STATE, "no mesh set");
    clear_fixed(c);
    MFH_CATCH(c)
}

mfh_status mfh_fix_variables(mfh_ctx *c, int64_t n, const int64_t *vars, const double *vals) {
    MFH_TRY(c)
    require(c && (c->haveMesh || c->external), MFH_ERR_STATE, "no mesh set");
    if (n == 0) return MFH_OK;
    require(vars && n > 0, MFH_ERR_INVALID, "null argument");
    add_fixed(c, n, vars, vals);
    MFH_CATCH(c)
}

mfh_status mfh_set_preconditioner(mfh_ctx *c, int32_t kind) {
    if (!c || kind < 0 || kind > MFH_PRECOND_AUTO) return MFH_ERR_INVALID;
    c->precondAuto = kind == MFH_PRECOND_AUTO;
    c->autoStretch = -1.0;
    c->precond = c->precondAuto ? MFH_PRECOND_MULTIGRID : kind;      // (AUTO: the choice is made for the mesh in hand when the next solve prepares itself)
    c->dinvValid = false;
    return MFH_OK;
}

mfh_status mfh_precond_choice(mfh_ctx *c, int32_t *kind, int32_t *isAuto, double *meshStretch) {
    MFH_TRY(c)
    require(c, MFH_ERR_INVALID, "null context");
    if (c->precondAuto && c->haveMesh && !c->hostOnly) { require_device(c); MFH_HIP(hipSetDevice(c->device)); resolve_auto_precond(c); }
    if (kind) *kind = c->precond;
    if (isAuto) *isAuto = c->precondAuto ? 1 : 0;
    if (meshStretch) *meshStretch = c->autoStretch;
    MFH_CATCH(c)
}

mfh_status mfh_precond_info(const mfh_ctx *c, int32_t *nAgg, int64_t *coarseDim, double *setup_ms, const char **note) {
    if (!c) return MFH_ERR_INVALID;
    // with the multigrid hierarchy in use: the aggregates below its linear level (the finest aggregate level), the dimension of the
    // level that is inverted densely, the setup time of the whole hierarchy
    if (c->precond == MFH_PRECOND_MULTIGRID && c->mg.valid && (c->mg.coarse || c->mg.linearOnly)) {
        const bool hier = !c->mg.agg.empty();
        const mfh_ctx *lin = c->mg.linearOnly ? c : c->mg.coarse;           // the context of the linear level
        if (nAgg) *nAgg = hier ? (int32_t)c->mg.agg[0]->nAgg : (lin->tl.valid ? lin->tl.nAgg : 0);
        if (coarseDim) *coarseDim = hier ? c->mg.denseM : (lin->tl.valid ? lin->tl.m : 0);
        if (setup_ms) *setup_ms = c->mg.setup_ms;
        if (note) *note = c->precondNote.c_str();
        return MFH_OK;
    }
    if (nAgg) *nAgg = c->tl.valid ? c->tl.nAgg : 0;
    if (coarseDim) *coarseDim = c->tl.valid ? c->tl.m : 0;
    if (setup_ms) *setup_ms = c->tl.valid ? c->tl.setup_ms : 0.0;
    if (note) *note = c->precondNote.c_str();
    return MFH_OK;
}

mfh_status mfh_multigrid_info(const mfh_ctx *c, int64_t *fineDoF, int64_t *coarseDoF, double *lambdaMaxFine, double *lambdaMaxCoarse, double *setup_ms) {
    if (!c) return MFH_ERR_INVALID;
    const bool v = c->mg.valid;
    if (fineDoF) *fineDoF = v ? c->mg.nFine : 0;
    if (coarseDoF) *coarseDoF = v ? c->mg.nCoarse : 0;
    if (lambdaMaxFine) *lambdaMaxFine = v ? c->mg.lmax0 : 0.0;
    if (lambdaMaxCoarse) *lambdaMaxCoarse = v ? c->mg.lmax1 : 0.0;
    if (setup_ms) *setup_ms = v ? c->mg.setup_ms : 0.0;
    return MFH_OK;
}

// the aggregate levels of the hierarchy: per level {aggregates (global), rows this rank smooths, entries of its vectors, 1 = partitioned,
// exchange peers, halo aggregates received per exchange, owned aggregates sent per exchange}
mfh_status mfh_multigrid_level_info(const mfh_ctx *c, int32_t cap, int64_t *out7, int32_t *nLevels) {
    if (!c || !nLevels) return MFH_ERR_INVALID;
    const auto &agg = c->mg.agg;
    *nLevels = c->mg.valid ? (int32_t)agg.size() : 0;
    for (int32_t l = 0; l < *nLevels && l < cap && out7; ++l) {
        const auto &L = *agg[(size_t)l];
        int64_t *o = out7 + (size_t)l * 7;
        o[0] = L.nAgg; o[1] = L.rows(); o[2] = L.size(); o[3] = L.part ? 1 : 0; o[4] = (int64_t)L.xPeers.size();
        o[5] = L.part ? L.xRecvPtr.back() : 0; o[6] = L.part ? L.xSendPtr.back() : 0;
    }
    return MFH_OK;
}

mfh_status mfh_solve(mfh_ctx *c, int32_t nrhs, const double *f, double *u, double rtol, int32_t maxit, mfh_solve_info *info) {
    MFH_TRY(c)
    require(c && (c->haveMesh || c->external) && f && u && nrhs > 0 && maxit > 0 && rtol > 0, MFH_ERR_INVALID, "bad solve arguments");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_precond(c);
    ensure_coarse_levels(c, nrhs);
    require(c->sym.nRows == c->sym.nCols, MFH_ERR_STATE, "mfh_solve needs all rows owned; use mfh_dist_solve for partitioned meshes");
    const int64_t n = (int64_t)c->bs() * c->nDoF;
    std::vector<mfh_solve_info> infos((size_t)nrhs);
    try { solve_many(c, nrhs, f, u, n, rtol, maxit, infos.data()); }
    catch (...) { if (info) info[0] = infos[(size_t)nrhs - 1]; throw; }   // what is known reaches the caller even when a solve throws
    bool allConverged = true;
    for (int k2 = 0; k2 < nrhs; ++k2) allConverged &= infos[k2].converged != 0;
    if (info) info[0] = infos[(size_t)nrhs - 1];
    if (!allConverged) throw Error(MFH_ERR_NOT_CONVERGED, "PCG did not reach the requested tolerance within maxit iterations");
    MFH_CATCH(c)
}

mfh_status mfh_solve_batch(mfh_ctx *c, int32_t nrhs, const double *f, double *u, double rtol, int32_t maxit, mfh_solve_info *info) {
    MFH_TRY(c)
    require(c && (c->haveMesh || c->external) && f && u && nrhs > 0 && maxit > 0 && rtol > 0, MFH_ERR_INVALID, "bad solve arguments");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_precond(c);
    ensure_coarse_levels(c, nrhs);
    require(c->sym.nRows == c->sym.nCols, MFH_ERR_STATE, "mfh_solve_batch needs all rows owned; use mfh_dist_solve for partitioned meshes");
    const int64_t n = (int64_t)c->bs() * c->nDoF;
    std::vector<mfh_solve_info> infos((size_t)nrhs);
    try { solve_many(c, nrhs, f, u, n, rtol, maxit, infos.data()); }
    catch (...) { if (info) std::copy(infos.begin(), infos.end(), info); throw; }
    bool allConverged = true;
    for (int k2 = 0; k2 < nrhs; ++k2) { allConverged &= infos[k2].converged != 0; if (info) info[k2] = infos[k2]; }
    if (!allConverged) throw Error(MFH_ERR_NOT_CONVERGED, "PCG did not reach the requested tolerance within maxit iterations");
    MFH_CATCH(c)
}

mfh_status mfh_apply_K(mfh_ctx *c, const double *u, double *Ku) {
    MFH_TRY(c)
    require(c && (c->haveMesh || c->external) && u && Ku, MFH_ERR_INVALID, "null argument");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_assembled(c);
    const int d = c->bs();
    const int64_t nin = (int64_t)d * c->sym.nCols, nout = (int64_t)d * c->sym.nRows;
    c->wx.alloc(nin);
    c->wAp.alloc(std::max(nin, nout));
    MFH_HIP(hipMemcpyAsync(c->wx.p, u, nin * sizeof(double), hipMemcpyHostToDevice, c->stream));
    apply_operator(c, false, c->wx.p, c->wAp.p, nullptr);
    MFH_HIP(hipMemcpyAsync(Ku, c->wAp.p, nout * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MFH_HIP(hipStreamSynchronize(c->stream));
    MFH_CATCH(c)
}

// ---------------------------------------------------------------- device-pointer building blocks
mfh_status mfh_dev_spmv(mfh_ctx *c, const double *x_dev, double *y_dev) {
    MFH_TRY(c)
    require(c && x_dev && y_dev, MFH_ERR_INVALID, "null argument");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_assembled(c);
    apply_operator(c, false, x_dev, y_dev, nullptr);
    MFH_CATCH(c)
}
mfh_status mfh_dev_precond(mfh_ctx *c, const double *r_dev, double *z_dev) {
    MFH_TRY(c)
    require(c && r_dev && z_dev, MFH_ERR_INVALID, "null argument");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_precond(c);
    k::launch_precond(c->bs(), c->sym.nRows, c->dDinv.p, r_dev, z_dev, c->stream);
    MFH_CATCH(c)
}
// ---- two-level preconditioner on a row-partitioned context: the caller owns the (global) aggregates
// and the reduction over ranks; the library owns Z (implicit), the Galerkin pass over its rows, the
// dense inverse and the restrict / prolong kernels. Works on an unpartitioned context too.
mfh_status mfh_tl_partitioned_begin(mfh_ctx *c, int32_t nAgg, const int32_t *aggOfNode, const double *relPos, double *Ac_dev) {
    MFH_TRY(c)
    require(c && c->haveMesh && aggOfNode && relPos && Ac_dev && nAgg > 0, MFH_ERR_INVALID, "bad two-level arguments");
    // (with mfh_dof_map_partitioned the two arrays are per local DoF: the rows are DoFs there)
    require(c->dofForNode.empty() || c->nOwnedDoFSet >= 0, MFH_ERR_UNSUPPORTED, "caller-supplied aggregates need the identity DoF map or mfh_dof_map_partitioned");
    require(c->op == MFH_OP_ELASTICITY, MFH_ERR_UNSUPPORTED, "the rigid-body-mode coarse space is defined for the elasticity operator");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_precond(c);
    auto &T = c->tl;
    T.valid = false;
    const int d = c->dim();
    const int64_t nLocal = c->nDoF, nRows = c->sym.nRows;
    double t0 = now_ms();
    T.nModes = d == 3 ? 6 : 3;
    T.nAgg = nAgg; T.nColor = 0; T.H = 1.0;
    T.m = (int64_t)nAgg * T.nModes;
    std::vector<int32_t> agg(aggOfNode, aggOfNode + nLocal), aggPtr((size_t)nAgg + 1, 0), byAgg((size_t)nRows);
    for (int64_t n = 0; n < nLocal; ++n) require(agg[n] >= 0 && agg[n] < nAgg, MFH_ERR_INVALID, "aggregate id out of range");
    for (int64_t n = 0; n < nRows; ++n) aggPtr[(size_t)agg[n] + 1]++;
    for (int a = 0; a < nAgg; ++a) aggPtr[(size_t)a + 1] += aggPtr[a];
    {
        std::vector<int32_t> cur(aggPtr.begin(), aggPtr.end() - 1);
        for (int64_t n = 0; n < nRows; ++n) byAgg[(size_t)cur[agg[n]]++] = (int32_t)n;
    }
    std::vector<double> rp(relPos, relPos + (size_t)nLocal * 3);
    hipStream_t s = c->stream;
    T.aggOfDof.upload(agg, s); T.relPos.upload(rp, s); T.aggPtr.upload(aggPtr, s); T.dofsByAgg.upload(byAgg, s);
    T.rc.alloc((size_t)T.m); T.yc.alloc((size_t)T.m);
    if (!c->tlRapAgg) require_full_storage(c, "this construction of the coarse operator (option tl_rap_agg 0)");
    MFH_HIP(hipMemsetAsync(Ac_dev, 0, (size_t)T.m * T.m * sizeof(double), s));
    // caller-supplied aggregates carry no lattice: neighbour blocks fall back to global atomics, the diagonal ones do not
    if (c->tlRapAgg) k::launch_tl_rap_agg(tl_args(c), T.aggPtr.p, T.dofsByAgg.p, nullptr, c->dRowPtr.p, c->dColIdx.p, c->dVals.p, Ac_dev, s, c->upperOnly, c->sym.nRows);
    else k::launch_tl_rap(tl_args(c), nRows, c->dRowPtr.p, c->dColIdx.p, c->dVals.p, Ac_dev, s);
    MFH_HIP(hipStreamSynchronize(s));
    T.setup_ms = now_ms() - t0;
    MFH_CATCH(c)
}

mfh_status mfh_tl_partitioned_finish(mfh_ctx *c, const double *Ac_dev) {
    MFH_TRY(c)
    require(c && Ac_dev && c->tl.m > 0 && c->tl.aggOfDof.p, MFH_ERR_STATE, "mfh_tl_partitioned_begin has not run");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    double t0 = now_ms();
    if (!tl_invert_device(c, Ac_dev, false)) throw Error(MFH_ERR_NOT_CONVERGED, "two-level preconditioner: coarse operator not positive definite");
    c->tl.setup_ms += now_ms() - t0;
    c->tl.valid = true;
    MFH_CATCH(c)
}

mfh_status mfh_dev_tl_restrict(mfh_ctx *c, const double *r_dev, double *rc_dev) {
    MFH_TRY(c)
    require(c && r_dev && rc_dev, MFH_ERR_INVALID, "null argument");
    require(c->tl.valid, MFH_ERR_STATE, "two-level preconditioner is not set up");
    MFH_HIP(hipSetDevice(c->device));
    k::launch_tl_restrict(tl_args(c), c->tl.aggPtr.p, c->tl.dofsByAgg.p, r_dev, rc_dev, c->stream);
    MFH_CATCH(c)
}

mfh_status mfh_dev_tl_apply(mfh_ctx *c, const double *r_dev, const double *rc_dev, double *z_dev) {
    MFH_TRY(c)
    require(c && r_dev && rc_dev && z_dev, MFH_ERR_INVALID, "null argument");
    require(c->tl.valid, MFH_ERR_STATE, "two-level preconditioner is not set up");
    MFH_HIP(hipSetDevice(c->device));
    auto &T = c->tl;
    k::launch_tl_gemv(T.m, T.ldInv, T.Ainv.p, rc_dev, T.yc.p, c->stream);
    k::launch_tl_apply(tl_args(c), c->dDinv.p, r_dev, T.yc.p, z_dev, nullptr, -1, nullptr, c->stream);
    MFH_CATCH(c)
}

// fused vector updates of the distributed PCG; every scalar is read from device memory
mfh_status mfh_dev_pcg_update_xr(mfh_ctx *c, const double *num_dev, const double *den_dev, const double *p_dev, const double *Ap_dev,
                                 double *x_dev, double *r_dev) {
    MFH_TRY(c)
    require(c && num_dev && den_dev && p_dev && Ap_dev && x_dev && r_dev, MFH_ERR_INVALID, "null argument");
    MFH_HIP(hipSetDevice(c->device));
    k::launch_dev_update_xr((int64_t)c->bs() * c->sym.nRows, num_dev, den_dev, p_dev, Ap_dev, x_dev, r_dev, c->stream);
    MFH_CATCH(c)
}
mfh_status mfh_dev_pcg_direction(mfh_ctx *c, const double *num_dev, const double *den_dev, const double *z_dev, double *p_dev) {
    MFH_TRY(c)
    require(c && num_dev && den_dev && z_dev && p_dev, MFH_ERR_INVALID, "null argument");
    MFH_HIP(hipSetDevice(c->device));
    k::launch_dev_direction((int64_t)c->bs() * c->sym.nRows, num_dev, den_dev, z_dev, p_dev, c->stream);
    MFH_CATCH(c)
}
mfh_status mfh_dev_dots(mfh_ctx *c, const double *r_dev, const double *z_dev, double *out2_dev) {
    MFH_TRY(c)
    require(c && r_dev && z_dev && out2_dev, MFH_ERR_INVALID, "null argument");
    MFH_HIP(hipSetDevice(c->device));
    k::launch_dev_dots((int64_t)c->bs() * c->sym.nRows, r_dev, z_dev, out2_dev, c->stream);
    MFH_CATCH(c)
}

mfh_status mfh_dev_mask_fixed(mfh_ctx *c, double *r_dev) {
    MFH_TRY(c)
    require(c && r_dev, MFH_ERR_INVALID, "null argument");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_fixed_uploaded(c);
    if (!c->fixedVars.empty()) k::launch_mask((int64_t)c->bs() * c->sym.nRows, c->dFixedMask.p, r_dev, c->stream);
    MFH_CATCH(c)
}
mfh_status mfh_dev_set_fixed_values(mfh_ctx *c, double *u_dev) {
    MFH_TRY(c)
    require(c && u_dev, MFH_ERR_INVALID, "null argument");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_fixed_uploaded(c);
    k::launch_scatter_values((int64_t)c->fixedVars.size(), c->dFixedIdx.p, c->dFixedVal.p, u_dev, (int64_t)c->bs() * c->sym.nCols, c->stream);
    MFH_CATCH(c)
}
mfh_status mfh_dev_sync(mfh_ctx *c) {
    MFH_TRY(c)
    require(c, MFH_ERR_INVALID, "null context");
    MFH_HIP(hipStreamSynchronize(c->stream));
    MFH_CATCH(c)
}

// ---------------------------------------------------------------- measurement
mfh_status mfh_get_timing(const mfh_ctx *c, mfh_timing *out) {
    if (!c || !out) return MFH_ERR_INVALID;
    *out = c->timing;
    return MFH_OK;
}

mfh_status mfh_time_assembly_kernel(mfh_ctx *c, int32_t mode, int32_t reps, double *avg_ms) {
    MFH_TRY(c)
    require(c && c->haveMesh && avg_ms && reps > 0, MFH_ERR_INVALID, "bad arguments");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_geometry(c);
    ensure_symbolic(c, mode == MFH_ASSEMBLE_ATOMIC);
    k::AsmArgs a = asm_args(c);
    double total = 0;
    for (int r = 0; r < reps; ++r) {
        if (mode == MFH_ASSEMBLE_ATOMIC) c->dVals.zero(c->stream);
        EventTimer t(c->stream);
        if (mode == MFH_ASSEMBLE_ATOMIC) k::launch_assemble_atomic(a, c->stream);
        else k::launch_assemble_gather(a, c->stream);
        total += t.stop();
    }
    *avg_ms = total / reps;
    c->assembled = true;
    c->dinvValid = false;
    MFH_CATCH(c)
}

mfh_status mfh_time_spmv_kernel(mfh_ctx *c, int32_t reps, double *avg_ms) {
    MFH_TRY(c)
    require(c && c->haveMesh && avg_ms && reps > 0, MFH_ERR_INVALID, "bad arguments");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    ensure_assembled(c);
    const int d = c->bs();
    const int64_t nin = (int64_t)d * c->sym.nCols, nout = (int64_t)d * c->sym.nRows;
    c->wx.alloc(nin);
    c->wAp.alloc(std::max(nin, nout));
    k::launch_axpby(nin, 0.0, c->wx.p, 0.0, c->wx.p, c->stream);
    std::vector<double> ones((size_t)nin, 1.0);
    MFH_HIP(hipMemcpyAsync(c->wx.p, ones.data(), nin * sizeof(double), hipMemcpyHostToDevice, c->stream));
    apply_operator(c, false, c->wx.p, c->wAp.p, nullptr);   // warm-up
    EventTimer t(c->stream);
    for (int r = 0; r < reps; ++r) apply_operator(c, false, c->wx.p, c->wAp.p, nullptr);
    *avg_ms = t.stop() / reps;
    MFH_CATCH(c)
}

mfh_status mfh_debug_row_chunks(int64_t nRows, const int32_t *rowPtr, int32_t chunkSlots, int64_t nBreaks, const int64_t *breaks, int64_t grain, int32_t threads,
                                int32_t *chunkRow, int64_t cap, int64_t *nOut) {
    mfh_ctx *none = nullptr;
    MFH_TRY(none)
    require(nRows > 0 && rowPtr && chunkRow && nOut && chunkSlots > 0, MFH_ERR_INVALID, "mfh_debug_row_chunks: arguments");
    const std::vector<int64_t> br(breaks, breaks + (breaks ? nBreaks : 0));
    const std::vector<int32_t> cr = mfh::make_chunks(rowPtr, nRows, chunkSlots, br, grain, threads);
    *nOut = (int64_t)cr.size();
    require((int64_t)cr.size() <= cap, MFH_ERR_INVALID, "mfh_debug_row_chunks: output capacity");
    std::copy(cr.begin(), cr.end(), chunkRow);
    MFH_CATCH(none)
}

mfh_status mfh_debug_device_node_tables(mfh_ctx *c, int32_t *elemNodes, double *nodePos) {
    MFH_TRY(c)
    require(c && c->haveMesh && !c->hostOnly && elemNodes && nodePos, MFH_ERR_STATE, "no mesh on a device");
    MFH_HIP(hipSetDevice(c->device));
    const HostMesh &m = c->mesh;
    require(c->dElemNodes.n == (size_t)m.nElem * m.npe && c->dVertPos.n == (size_t)m.nNode * m.dim, MFH_ERR_STATE, "device node tables have another size than the host's");
    c->dElemNodes.download(elemNodes, c->dElemNodes.n, c->stream);
    c->dVertPos.download(nodePos, c->dVertPos.n, c->stream);
    MFH_CATCH(c)
}

// test hook: one allocation / release through the device arena, in the scope of the context (its streams are what a release waits for)
mfh_status mfh_debug_arena_alloc(mfh_ctx *c, int64_t bytes, void **out) {
    MFH_TRY(c)
    require(c && !c->hostOnly && out && bytes > 0, MFH_ERR_INVALID, "mfh_debug_arena_alloc: arguments");
    MFH_HIP(hipSetDevice(c->device));
    *out = mfh::device_alloc((size_t)bytes);
    MFH_CATCH(c)
}
mfh_status mfh_debug_arena_free(mfh_ctx *c, void *p) {
    MFH_TRY(c)
    require(c && !c->hostOnly, MFH_ERR_INVALID, "mfh_debug_arena_free: arguments");
    MFH_HIP(hipSetDevice(c->device));
    mfh::device_free(p);
    MFH_CATCH(c)
}

// kernel time of every candidate of the last placement trials (option "placement_trials"), the first being the buffer of the symbolic phase
mfh_status mfh_placement_info(const mfh_ctx *c, int32_t cap, double *ms, int32_t *n) {
    if (!c || !n) return MFH_ERR_INVALID;
    *n = (int32_t)c->placementMs.size();
    for (int32_t k = 0; ms && k < cap && k < *n; ++k) ms[k] = c->placementMs[(size_t)k];
    return MFH_OK;
}

mfh_status mfh_debug_spd_inverse_device(mfh_ctx *c, int64_t n, double *A) {
    MFH_TRY(c)
    require(c && A && n > 0, MFH_ERR_INVALID, "bad arguments");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int64_t mp = ((n + 63) / 64) * 64;
    DBuf<double> Ac, Ap, X, Ainv, Dt;
    DBuf<uint8_t> dead;
    DBuf<int> flag;
    Ac.upload(A, (size_t)n * n, s);
    std::vector<uint8_t> hd((size_t)n, 0);
    dead.upload(hd, s);
    double maxd = 0;
    for (int64_t i = 0; i < n; ++i) maxd = std::max(maxd, A[(size_t)i * n + i]);
    Ap.alloc((size_t)mp * mp); X.alloc((size_t)mp * mp); Ainv.alloc((size_t)mp * mp); Dt.alloc((size_t)(mp / 64) * 4096); flag.alloc(1);
    k::launch_tl_prep(n, mp, Ac.p, dead.p, maxd, Ap.p, s);
    if (!k::dense_spd_inverse_device(Ap.p, X.p, Ainv.p, Dt.p, mp, flag.p, s)) throw Error(MFH_ERR_INVALID, "not SPD");
    MFH_HIP(hipMemcpy2DAsync(A, (size_t)n * sizeof(double), Ainv.p, (size_t)mp * sizeof(double), (size_t)n * sizeof(double), (size_t)n,
                             hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    MFH_CATCH(c)
}

mfh_status mfh_debug_spd_inverse(int64_t n, double *A) {
    try {
        return spd_inverse_inplace(n, A) ? MFH_OK : MFH_ERR_INVALID;
    } catch (...) { return MFH_ERR_INVALID; }
}

mfh_status mfh_set_option(mfh_ctx *c, const char *key, double value) {
    MFH_TRY(c)
    require(c && key, MFH_ERR_INVALID, "null argument");
    const std::string k2(key);
    if (k2 == "asm_chunk_order") { c->asmChunkOrder = (int)value; }
    else if (k2 == "mg_steps_fine") c->mgSteps0 = std::max(1, (int)value);
    else if (k2 == "mg_steps_coarse") c->mgSteps1 = std::max(1, (int)value);
    else if (k2 == "mg_ratio_fine") { require(value > 0 && value < 1, MFH_ERR_INVALID, "mg_ratio_fine must lie in (0, 1)"); c->mgRatio0 = value; }
    else if (k2 == "mg_ratio_coarse") { require(value > 0 && value < 1, MFH_ERR_INVALID, "mg_ratio_coarse must lie in (0, 1)"); c->mgRatio1 = value; }
    else if (k2 == "mg_coarse_cycles") c->mgCoarseCycles = std::max(1, (int)value);
    else if (k2 == "placement_trials") { c->placementTrials = std::max(0, std::min(8, (int)value)); c->placementGen = -1; }
    else if (k2 == "mg_eig_margin") { require(value >= 1.0, MFH_ERR_INVALID, "mg_eig_margin must be >= 1"); c->mgEigMargin = value; c->mg.valid = false; }
    else if (k2 == "mg_agg_target") { c->mgAggTarget = std::max(0, (int)value); c->mg.valid = false; }
    else if (k2 == "mg_coarse_fp32") { c->mgCoarseFp32 = value != 0; c->mg.valid = false; }
    else if (k2 == "mg_over_correction") { require(value > 0 && value < 4, MFH_ERR_INVALID, "mg_over_correction must lie in (0, 4)"); c->mgOverCorrection = value; }
    else if (k2 == "mg_dense_max") { c->mgDenseMax = std::max(8, (int)value); c->mg.valid = false; }
    else if (k2 == "mg_steps_agg") c->mgStepsAgg = std::max(1, (int)value);
    else if (k2 == "mg_ratio_agg") { require(value > 0 && value < 1, MFH_ERR_INVALID, "mg_ratio_agg must lie in (0, 1)"); c->mgRatioAgg = value; }
    else if (k2 == "mg_anisotropic_bins") { c->mgAnisotropicBins = value != 0; c->mg.valid = false; }
    else if (k2 == "mg_agg_nodes") { c->mgAggNodes = std::max(0, (int)value); c->mg.valid = false; }
    else if (k2 == "mg_replicate_max") { c->mgReplicateMax = std::max(0, (int)value); c->mg.valid = false; }
    else if (k2 == "asm_packed_codes") { c->asmPackedCodes = value != 0; invalidate_symbolic(c); }
    else if (k2 == "chunk_slots") { c->chunkSlots = (int)value; invalidate_symbolic(c); }
    else if (k2 == "contrib_order") { c->contribOrder = (int)value; invalidate_symbolic(c); }
    else if (k2 == "check_every") { c->checkEvery = std::max(1, (int)value); }
    else if (k2 == "keep_host_symbolic") { c->keepHostSymbolic = value != 0; }
    else if (k2 == "reembed") { c->alwaysReembed = value != 0; }
    else if (k2 == "periodic_ignore_mismatch") { c->periodicIgnoreMismatch = value != 0; }
    else if (k2 == "solve_homogeneous") { c->solveHomogeneous = value != 0; }
    else if (k2 == "periodic_ignore_dims") { c->periodicIgnoreDims = (int)value & 7; }
    else if (k2 == "agg_nodes") { c->aggNodes = (int)value; c->tl.valid = false; }
    else if (k2 == "topology_device") { c->topologyDevice = value != 0; }
    else if (k2 == "matrix_storage") {
        require(value == 0 || value == 1 || value == -1, MFH_ERR_INVALID, "matrix_storage: 0 = both triangles, 1 = blocks (r, c >= r) only, -1 = automatic");
        c->matrixStorage = (int)value;
    }
    else if (k2 == "symbolic_device") { c->symbolicDevice = value != 0; invalidate_symbolic(c); }
    else if (k2 == "xcd_swizzle") c->xcdSwizzle = std::max(0, (int)value);   // 1: contiguous eighths; G > 1: runs of G items per XCD
    else if (k2 == "pcg_graph") c->useGraph = value != 0;
    else if (k2 == "refine") c->refine = value != 0;
    else if (k2 == "mf_geometry_from_vertices") c->mfGeoFromVerts = value != 0;
    else if (k2 == "mf_xcd_group") c->mfXcdGroup = std::max(0, (int)value);
    else if (k2 == "vec_grid_cap") k::g_vecGridCap = std::max(256, (int)value);
    else if (k2 == "mf_lane_stride") c->mfLaneStride = std::max(1, (int)value);
    else if (k2 == "mf_reorder") { c->mfReorder = value != 0; c->mfcValid = false; c->mfClusterUnfit = false; }
    else if (k2 == "dist_pcg_variant") c->distPcgVariant = value != 0 ? 1 : 0;
    else if (k2 == "dist_profile") c->dist.profile = value != 0;
    else if (k2 == "deterministic") {
        const bool on = value != 0;
        if (on && !c->detPartials.p) {
            require_device(c);
            MFH_HIP(hipSetDevice(c->device));
            c->detPartials.alloc((size_t)8 + (size_t)4096 * 4);      // header + 4 096 workgroups x 4 partials (launches with global sums use at most that many
                                                                     // workgroups in this mode: the one-workgroup second stage reads 16 partials per lane and sum)
            c->detPartials.zero(c->stream);
            c->detCounter.alloc(64);
            c->detCounter.zero(c->stream);
            MFH_HIP(hipStreamSynchronize(c->stream));
        }
        if (on != c->deterministic) { c->deterministic = on; invalidate_matrix(c); destroy_multigrid(c); c->tl.valid = false; c->mfcValid = false; c->mfClusterUnfit = false; }   // (the block size of the cluster operator depends on it)
    }
    else if (k2 == "pcg_variant") c->pcgVariant = value < 0 ? -1 : (value != 0 ? 1 : 0);
    else if (k2 == "mg_batch") c->mgBatch = value != 0;
    else if (k2 == "mg_fuse") c->mgFuse = value != 0;
    else if (k2 == "mg_dinv_fp32") c->mgDinvFp32 = value != 0;
    else if (k2 == "auto_stretch_max") { c->autoStretchMax = value > 1.0 ? value : 1.0; c->autoStretch = -1.0; }
    else if (k2 == "batch_rhs") { if (c->batchRhs != (value != 0)) { c->mfcValid = false; c->mfClusterUnfit = false; } c->batchRhs = value != 0; }
    else if (k2 == "matrix_free_mode") { c->mfMode = (int)value; c->mfClusterUnfit = false; }
    else if (k2 == "mf_block_elems") { c->mfBlockElems = (int)value; c->mfcValid = false; c->mfClusterUnfit = false; }
    else if (k2 == "mf_chunk_rows") { c->mfChunkRows = std::max(16, std::min(4096, (int)value)); c->mfValid = false; }
    else if (k2 == "mf_chunk_pairs") { c->mfChunkPairs = std::max(256, (int)value); c->mfValid = false; }
    else if (k2 == "matrix_free") c->matrixFree = value < 0 ? -1 : (value != 0 ? 1 : 0);   // K x without reading the assembled K (k_spmv_mf)
    else if (k2 == "tl_rap_agg") { c->tlRapAgg = value != 0; c->tl.valid = false; }
    else if (k2 == "tl_device_aggregates") { c->tlDeviceAggregates = value != 0; c->tl.valid = false; }
    else if (k2 == "tl_probe") { c->tlProbe = value != 0; c->tl.valid = false; }
    else if (k2 == "tl_host_inverse") { c->tlHostInverse = value != 0; c->tl.valid = false; }
    else throw Error(MFH_ERR_INVALID, "unknown option " + k2);
    refresh_storage_rule(c);   // matrix_storage, matrix_free, tl_probe, tl_rap_agg decide the storage of K
    MFH_CATCH(c)
}

} // extern "C"
