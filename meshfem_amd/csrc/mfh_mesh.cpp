// Host-side FEM mesh construction: node numbering and boundary extraction in the reference's
// enumeration order (FEMMesh.inl:11-82, TetMesh.inl:15-120, TriMesh.inl:15-130), written with
// sorts/hash tables instead of std::map so that 5M-tet meshes build in seconds.
#include "mfh_internal.hh"
#include <cmath>
#include <numeric>
#include <unordered_map>
#include <sys/mman.h>

namespace mfh {

int host_threads() {
    static int n = [] {
        const char *e = getenv("MESHFEM_NUM_THREADS"); // python/parallelism.py:4-6
        if (e && atoi(e) > 0) return atoi(e);
        unsigned h = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(h, 64u));
    }();
    return n;
}

void host_advise_huge_pages(void *p, size_t bytes) {
    static const bool on = [] { const char *e = getenv("MFH_HOST_HUGE_PAGES"); return !(e && atoi(e) == 0); }();
    if (!on || bytes < ((size_t)8 << 20)) return;
    const uintptr_t a = ((uintptr_t)p + (((uintptr_t)2 << 20) - 1)) & ~(((uintptr_t)2 << 20) - 1);
    const uintptr_t e = ((uintptr_t)p + bytes) & ~(((uintptr_t)2 << 20) - 1);
    if (e > a) (void)madvise((void *)a, (size_t)(e - a), MADV_HUGEPAGE);    // advisory: failure changes nothing
}

void parallel_ranges(int64_t n, const std::function<void(int64_t, int64_t, int)> &f, int64_t minGrain) {
    int nt = host_threads();
    if (n <= minGrain || nt == 1) {
        f(0, n, 0);
        return;
    }
    nt = (int)std::min<int64_t>(nt, (n + minGrain - 1) / minGrain);
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(nt);
    for (int t = 0; t < nt; ++t) {
        int64_t b = n * t / nt, e = n * (t + 1) / nt;
        th.emplace_back([&, b, e, t] {
            try { f(b, e, t); } catch (...) { err[t] = std::current_exception(); }
        });
    }
    for (auto &t : th) t.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
}

// tet half-face corner table (TetMesh.hh:221-226)
static const int kFaceCorner[4][3] = {{1, 3, 2}, {0, 2, 3}, {0, 3, 1}, {0, 1, 2}};

namespace {
struct EdgeHash {
    // open addressing with growth, key = (min << 32) | max
    std::vector<uint64_t> keys;
    std::vector<int32_t> vals;
    uint64_t mask = 0;
    size_t count = 0;
    explicit EdgeHash(size_t expected) {
        size_t cap = 64;
        while (cap < expected * 2) cap <<= 1;
        keys.assign(cap, ~0ull);
        vals.assign(cap, -1);
        mask = cap - 1;
    }
    static uint64_t mix(uint64_t k) {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
        return k;
    }
    void grow() {
        std::vector<uint64_t> ok;
        std::vector<int32_t> ov;
        ok.swap(keys); ov.swap(vals);
        const size_t cap = ok.size() * 2;
        keys.assign(cap, ~0ull);
        vals.assign(cap, -1);
        mask = cap - 1;
        for (size_t i = 0; i < ok.size(); ++i) {
            if (ok[i] == ~0ull) continue;
            uint64_t h = mix(ok[i]) & mask;
            while (keys[h] != ~0ull) h = (h + 1) & mask;
            keys[h] = ok[i]; vals[h] = ov[i];
        }
    }
    // returns value; inserts `next` if absent
    int32_t get_or_insert(uint64_t key, int32_t next, bool &inserted) {
        if ((count + 1) * 2 > keys.size()) grow();
        uint64_t h = mix(key) & mask;
        while (true) {
            if (keys[h] == key) { inserted = false; return vals[h]; }
            if (keys[h] == ~0ull) { keys[h] = key; vals[h] = next; inserted = true; ++count; return next; }
            h = (h + 1) & mask;
        }
    }
    int32_t at(uint64_t key) const {
        uint64_t h = mix(key) & mask;
        while (true) {
            if (keys[h] == key) return vals[h];
            if (keys[h] == ~0ull) throw Error(MFH_ERR_INVALID, "edge lookup failed");
            h = (h + 1) & mask;
        }
    }
};
inline uint64_t edge_key(int32_t a, int32_t b) {
    return a < b ? ((uint64_t)(uint32_t)a << 32) | (uint32_t)b : ((uint64_t)(uint32_t)b << 32) | (uint32_t)a;
}
} // namespace

// Host topology (used by host-only contexts and as fallback): first-encounter edge numbering with a
// hash table, unmatched half-faces / half-edges by sorting. Same outputs as build_topology_device.
static void build_topology_host(int dim, int deg, int64_t nElem, int64_t nVert, const int32_t *ev, RawVec<int32_t> &instEdge,
                                int32_t &nEdgeNodes, std::vector<uint32_t> &bdryInst) {
    const int nv = dim + 1, nedge = dim == 3 ? 6 : 3;
    nEdgeNodes = 0;
    instEdge.clear();
    if (deg == 2) {
        instEdge.resize((size_t)nElem * nedge);
        EdgeHash eh((size_t)(nElem * (dim == 3 ? 1.4 : 1.7)) + 64);
        for (int64_t e = 0; e < nElem; ++e) {
            const int32_t *v = ev + e * nv;
            for (int ei = 0; ei < nedge; ++ei) {
                bool ins;
                int32_t k = eh.get_or_insert(edge_key(v[kEdgeStart[ei]], v[kEdgeEnd[ei]]), nEdgeNodes, ins);
                if (ins) ++nEdgeNodes;
                instEdge[(size_t)e * nedge + ei] = k;
            }
        }
    }
    bdryInst.clear();
    if (dim == 3) {
        struct HF { int32_t a, b, c; uint32_t hf; };
        std::vector<HF> hfs((size_t)nElem * 4);
        parallel_ranges(nElem, [&](int64_t b, int64_t e, int) {
            for (int64_t t = b; t < e; ++t)
                for (int f = 0; f < 4; ++f) {
                    int32_t x = ev[t * 4 + kFaceCorner[f][0]], y = ev[t * 4 + kFaceCorner[f][1]], z = ev[t * 4 + kFaceCorner[f][2]];
                    int32_t lo = std::min(x, std::min(y, z)), hi = std::max(x, std::max(y, z));
                    int32_t mid = x ^ y ^ z ^ lo ^ hi;
                    hfs[(size_t)t * 4 + f] = HF{lo, mid, hi, (uint32_t)(t * 4 + f)};
                }
        });
        std::sort(hfs.begin(), hfs.end(), [](const HF &p, const HF &q) {
            if (p.a != q.a) return p.a < q.a;
            if (p.b != q.b) return p.b < q.b;
            if (p.c != q.c) return p.c < q.c;
            return p.hf < q.hf;
        });
        for (size_t k = 0; k < hfs.size();) {
            size_t k2 = k + 1;
            while (k2 < hfs.size() && hfs[k2].a == hfs[k].a && hfs[k2].b == hfs[k].b && hfs[k2].c == hfs[k].c) ++k2;
            if (k2 - k > 2) throw Error(MFH_ERR_INVALID, "Non-manifold input detected.");
            if (k2 - k == 1) bdryInst.push_back(hfs[k].hf);
            k = k2;
        }
    } else {
        struct HE { int32_t a, b, tail, tip; uint32_t he; };
        std::vector<HE> hes((size_t)nElem * 3);
        for (int64_t t = 0; t < nElem; ++t)
            for (int c = 0; c < 3; ++c) {
                int32_t tail = ev[t * 3 + (c + 1) % 3], tip = ev[t * 3 + (c + 2) % 3]; // TriMesh.hh:285-298
                hes[(size_t)t * 3 + c] = HE{std::min(tail, tip), std::max(tail, tip), tail, tip, (uint32_t)(t * 3 + c)};
            }
        std::sort(hes.begin(), hes.end(), [](const HE &p, const HE &q) {
            if (p.a != q.a) return p.a < q.a;
            if (p.b != q.b) return p.b < q.b;
            return p.he < q.he;
        });
        for (size_t k = 0; k < hes.size();) {
            size_t k2 = k + 1;
            while (k2 < hes.size() && hes[k2].a == hes[k].a && hes[k2].b == hes[k].b) ++k2;
            if (k2 - k > 2) throw Error(MFH_ERR_INVALID, "Non-manifold edge detected");
            if (k2 - k == 2 && hes[k].tail != hes[k + 1].tip) throw Error(MFH_ERR_INVALID, "Inconsistent triangle orientations.");
            if (k2 - k == 1) bdryInst.push_back(hes[k].he);
            k = k2;
        }
    }
}

// boundary element embedding (EmbeddedElement.hh:128-149 tri in 3D, :87-104 edge in 2D) from the vertex positions vp
void compute_boundary_geometry(HostMesh &m, const double *vp) {
    const int dim = m.dim;
    const int64_t nBE = m.nBE();
    m.bdryVol.resize(nBE);
    m.bdryNormal.resize((size_t)nBE * dim);
    for (int64_t b = 0; b < nBE; ++b) {
        const int32_t *bn = &m.bdryElemNodes[(size_t)b * m.npbe];
        if (dim == 3) {
            const double *p0 = vp + 3 * (size_t)bn[0], *p1 = vp + 3 * (size_t)bn[1], *p2 = vp + 3 * (size_t)bn[2];
            double e1[3], e2[3];
            for (int a = 0; a < 3; ++a) { e1[a] = p0[a] - p2[a]; e2[a] = p1[a] - p0[a]; }
            double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            double dA = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            m.bdryVol[b] = dA / 2.0;
            for (int a = 0; a < 3; ++a) m.bdryNormal[(size_t)b * 3 + a] = n[a] / dA;
        } else {
            const double *p0 = vp + 2 * (size_t)bn[0], *p1 = vp + 2 * (size_t)bn[1];
            double e[2] = {p1[0] - p0[0], p1[1] - p0[1]};
            double L = std::sqrt(e[0] * e[0] + e[1] * e[1]);
            m.bdryVol[b] = L;
            m.bdryNormal[(size_t)b * 2 + 0] = -e[1] / L;
            m.bdryNormal[(size_t)b * 2 + 1] = e[0] / L;
        }
    }
}

void build_fem_mesh(HostMesh &m, int dim, int deg, int64_t nElem, int64_t nVert, const int32_t *ev, const double *vp, bool useDevice,
                    hipStream_t stream, DBuf<int32_t> *dElemNodesOut, DBuf<double> *dNodePosOut, bool *deviceTables) {
    if (dim != 2 && dim != 3) throw Error(MFH_ERR_INVALID, "dim must be 2 or 3");
    if (deg != 1 && deg != 2) throw Error(MFH_ERR_INVALID, "deg must be 1 or 2");
    if (nElem <= 0 || nVert <= 0) throw Error(MFH_ERR_INVALID, "empty mesh");
    const int nv = dim + 1;
    const int nedge = dim == 3 ? 6 : 3;
    const double tEnter = now_ms();
    m = HostMesh();
    m.dim = dim; m.deg = deg; m.npe = nodes_per_elem(dim, deg); m.npbe = nodes_per_bdry_elem(dim, deg);
    m.nElem = nElem; m.nVert = nVert;
    m.vertPos.resize((size_t)nVert * dim);
    parallel_ranges(nVert * dim, [&](int64_t b, int64_t e2, int) { std::copy(vp + b, vp + e2, m.vertPos.data() + b); });   // (first touch on all threads)
    {
        std::vector<uint8_t> bad((size_t)host_threads() + 1, 0);
        parallel_ranges(nElem * nv, [&](int64_t b, int64_t e2, int tid) {
            for (int64_t k = b; k < e2; ++k)
                if (ev[k] < 0 || ev[k] >= nVert) bad[(size_t)tid] = 1;
        });
        for (uint8_t b : bad)
            if (b) throw Error(MFH_ERR_INVALID, "Bad vertex index encountered.");
    }
    const bool timing = getenv("MFH_MESH_TIMING") != nullptr;
    double tp = tEnter;
    auto lap = [&](const char *what) {
        if (!timing) return;
        const double t = now_ms();
        fprintf(stderr, "[mesh build] %-30s %8.2f ms\n", what, t - tp);
        tp = t;
    };

    lap("vertex copy + index check");
    // ---- topology: edge nodes in first-encounter order over (element, local edge) (FEMMesh.inl:22-36) and
    //      the unmatched half-faces / half-edges in sorted-key order (TetMesh.inl:36-79, TriMesh.inl:60-100)
    RawVec<int32_t> instEdge;
    std::vector<uint32_t> bdryInst;
    int32_t nEdgeNodes = 0;
    if (deviceTables) *deviceTables = false;
    // the two large host tables are sized, and take their first-touch page faults on all host threads, while the device sorts edges and faces
    // (filled below: the fills then run at memory speed)
    auto sizeTables = [&](int stage, int32_t nEdge) {
        if (stage == 0) resize_prefaulted(m.elemNodes, (size_t)nElem * m.npe);
        else resize_prefaulted(m.nodePos, (size_t)(nVert + nEdge) * dim);
    };
    if (useDevice && build_topology_device(dim, deg, nElem, nVert, ev, stream, instEdge, nEdgeNodes, bdryInst, vp, dElemNodesOut, dNodePosOut, sizeTables)) {
        if (deviceTables) *deviceTables = dElemNodesOut && dNodePosOut;
    } else
        build_topology_host(dim, deg, nElem, nVert, ev, instEdge, nEdgeNodes, bdryInst);

    lap("topology (edges, boundary)");
    m.nNode = nVert + nEdgeNodes;
    m.nOwned = m.nNode;
    // the two large host tables, on the host threads. (Filling them on a second thread beside the boundary section below was measured SLOWER at
    // 57.6 M nodes -- 0.85-1.1 s against 0.68-0.85 s for the whole build on one box: 128 threads taking their first-touch page faults contend
    // with the allocations of the boundary section for the process's memory map.)
    m.elemNodes.resize((size_t)nElem * m.npe);
    parallel_ranges(nElem, [&](int64_t b, int64_t e2, int) {
        for (int64_t e = b; e < e2; ++e) {
            int32_t *out = &m.elemNodes[(size_t)e * m.npe];
            for (int c = 0; c < nv; ++c) out[c] = ev[e * nv + c];
            if (deg == 2)
                for (int ei = 0; ei < nedge; ++ei) out[nv + ei] = (int32_t)nVert + instEdge[(size_t)e * nedge + ei];
        }
    });
    lap("element node table");
    compute_node_positions(m);
    lap("node positions");

    // ---- boundary elements from the unmatched instances
    const int64_t nBE = (int64_t)bdryInst.size();
    const int nbv = dim; // vertices per boundary element
    std::vector<std::array<int32_t, 3>> bfaceVolCorners((size_t)nBE);   // volume half-face corner order
    m.bdryParent.resize((size_t)nBE);
    for (int64_t b = 0; b < nBE; ++b) {
        const uint32_t inst = bdryInst[b];
        m.bdryParent[b] = (int32_t)(inst / (dim + 1));
        if (dim == 3) {
            const int64_t t = inst / 4; const int f = inst % 4;
            bfaceVolCorners[b] = {ev[t * 4 + kFaceCorner[f][0]], ev[t * 4 + kFaceCorner[f][1]], ev[t * 4 + kFaceCorner[f][2]]};
        } else {
            const int64_t t = inst / 3; const int c = inst % 3;
            bfaceVolCorners[b] = {ev[t * 3 + (c + 1) % 3], ev[t * 3 + (c + 2) % 3], -1};   // (tail, tip)
        }
    }
    // boundary vertex numbering: first encounter in volume-corner order (TetMesh.inl:82-89;
    // TriMesh.inl:103-104 visits tipVV = vol tail first, then tailVV = vol tip)
    std::vector<int32_t> Vb((size_t)nVert, -1);
    std::vector<int32_t> bV;
    for (auto &fc : bfaceVolCorners)
        for (int c = 0; c < nbv; ++c) {
            int32_t v = fc[c];
            if (Vb[v] < 0) { Vb[v] = (int32_t)bV.size(); bV.push_back(v); }
        }
    // boundary element vertices: 3D corner c = volume corner 2-c (TetMesh.hh:463-469);
    // 2D vertex0 = boundary tail = volume tip, vertex1 = volume tail (TriMeshHandles.hh:269)
    m.bdryElemNodes.assign((size_t)nBE * m.npbe, -1);
    for (int64_t b = 0; b < nBE; ++b) {
        int32_t *out = &m.bdryElemNodes[(size_t)b * m.npbe];
        if (dim == 3) { out[0] = bfaceVolCorners[b][2]; out[1] = bfaceVolCorners[b][1]; out[2] = bfaceVolCorners[b][0]; }
        else { out[0] = bfaceVolCorners[b][1]; out[1] = bfaceVolCorners[b][0]; }
    }
    m.bdryNodes.assign(bV.begin(), bV.end());
    if (deg == 2) {
        // local edge index of a pair of local corners (Simplex.hh:43-44)
        int pairToEdge[4][4];
        for (auto &row : pairToEdge) for (int &x : row) x = -1;
        for (int ei = 0; ei < nedge; ++ei) { pairToEdge[kEdgeStart[ei]][kEdgeEnd[ei]] = ei; pairToEdge[kEdgeEnd[ei]][kEdgeStart[ei]] = ei; }
        const int nbedge = dim == 3 ? 3 : 1;
        m.isBdryNode.assign((size_t)m.nNode, 0);      // doubles as the "edge node already listed" marker (one byte per node instead of a 4-byte table over all edges)
        for (int64_t b = 0; b < nBE; ++b) {
            const uint32_t inst = bdryInst[b];
            int32_t *out = &m.bdryElemNodes[(size_t)b * m.npbe];
            // local element corners of the boundary element's vertices, in boundary order
            int lc[3];
            int64_t t;
            if (dim == 3) { t = inst / 4; const int f = inst % 4; lc[0] = kFaceCorner[f][2]; lc[1] = kFaceCorner[f][1]; lc[2] = kFaceCorner[f][0]; }
            else { t = inst / 3; const int c = inst % 3; lc[0] = (c + 2) % 3; lc[1] = (c + 1) % 3; lc[2] = -1; }
            for (int ei = 0; ei < nbedge; ++ei) {   // boundary-local edges (0,1),(1,2),(2,0)   FEMMesh.inl:43-58
                const int le = pairToEdge[lc[kEdgeStart[ei]]][lc[kEdgeEnd[ei]]];
                const int32_t volEdge = instEdge[(size_t)t * nedge + le];
                if (!m.isBdryNode[(size_t)nVert + volEdge]) {
                    m.isBdryNode[(size_t)nVert + volEdge] = 1;
                    m.bdryNodes.push_back((int32_t)nVert + volEdge);
                }
                out[nbv + ei] = (int32_t)nVert + volEdge;
            }
        }
    }
    if (deg != 2) m.isBdryNode.assign((size_t)m.nNode, 0);
    for (int32_t n : m.bdryNodes) m.isBdryNode[n] = 1;
    compute_boundary_geometry(m, vp);
    m.bdryInternal.assign((size_t)nBE, 0);
    m.hasTopology = true;
    lap("boundary elements + geometry");
}

// Node positions: vertex nodes = vertices; P2 edge node = midpoint of its end vertices
// (FEMMesh.hh:221-237). Derived from the element node table so it also works for mesh_set.
void compute_node_positions(HostMesh &m) {
    const int dim = m.dim, nv = dim + 1;
    m.nodePos.resize((size_t)m.nNode * dim);
    parallel_ranges(m.nVert * dim, [&](int64_t b, int64_t e, int) { std::copy(m.vertPos.data() + b, m.vertPos.data() + e, m.nodePos.data() + b); });
    if (m.deg == 2) {
        const int nedge = dim == 3 ? 6 : 3;
        // on the host threads; the elements sharing an edge all store the same midpoint (relaxed atomic stores of equal bit patterns)
        parallel_ranges(m.nElem, [&](int64_t eb, int64_t ee, int) {
            for (int64_t e = eb; e < ee; ++e) {
                const int32_t *en = &m.elemNodes[(size_t)e * m.npe];
                for (int ei = 0; ei < nedge; ++ei) {
                    const int32_t node = en[nv + ei];
                    const double *pa = &m.vertPos[(size_t)en[kEdgeStart[ei]] * dim], *pb = &m.vertPos[(size_t)en[kEdgeEnd[ei]] * dim];
                    for (int a = 0; a < dim; ++a) {
                        double v = 0.5 * (pa[a] + pb[a]);
                        __atomic_store(&m.nodePos[(size_t)node * dim + a], &v, __ATOMIC_RELAXED);
                    }
                }
            }
        });
    }
}

// PeriodicCondition (BoundaryConditions.hh:452-561, PeriodicBoundaryMatcher.hh:111-260):
// nodes on opposite faces of the bounding-box cell are identified; DoF ids are assigned in
// volume-node order, every identified node receiving the id at the first one's turn (:533-554).
void periodic_dof_map(const HostMesh &m, double eps, std::vector<int32_t> &dofForNode, int64_t &nDoF,
                      std::vector<uint8_t> &bdryInternal, bool ignoreMismatch, int ignoreDimsMask) {
    if (!m.hasTopology) throw Error(MFH_ERR_STATE, "periodic conditions need mesh topology (mfh_mesh_build)");
    const int dim = m.dim;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    {   // bounding box on the host threads (minima / maxima: exact in any order)
        const int nt = host_threads();
        std::vector<double> part((size_t)(nt + 1) * 6);
        for (int t = 0; t <= nt; ++t) for (int a = 0; a < 3; ++a) { part[(size_t)t * 6 + a] = 1e300; part[(size_t)t * 6 + 3 + a] = -1e300; }
        parallel_ranges(m.nNode, [&](int64_t lo, int64_t hi, int tid) {
            double a0[3] = {1e300, 1e300, 1e300}, a1[3] = {-1e300, -1e300, -1e300};
            for (int64_t n = lo; n < hi; ++n)
                for (int a = 0; a < dim; ++a) { a0[a] = std::min(a0[a], m.nodePos[(size_t)n * dim + a]); a1[a] = std::max(a1[a], m.nodePos[(size_t)n * dim + a]); }
            for (int a = 0; a < 3; ++a) { part[(size_t)tid * 6 + a] = std::min(part[(size_t)tid * 6 + a], a0[a]); part[(size_t)tid * 6 + 3 + a] = std::max(part[(size_t)tid * 6 + 3 + a], a1[a]); }
        });
        for (int t = 0; t <= nt; ++t) for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], part[(size_t)t * 6 + a]); mx[a] = std::max(mx[a], part[(size_t)t * 6 + 3 + a]); }
    }
    // face membership restricted to the periodic dimensions (PeriodicCondition's ignoreDims, BoundaryConditions.hh:470-500:
    // a node keeps only its memberships of non-ignored faces)
    auto periodicDim = [&](int a) { return !(ignoreDimsMask & (1 << a)); };
    auto onMin = [&](int64_t n, int a) { return periodicDim(a) && std::fabs(m.nodePos[(size_t)n * dim + a] - mn[a]) <= eps; };
    auto onMax = [&](int64_t n, int a) { return periodicDim(a) && std::fabs(m.nodePos[(size_t)n * dim + a] - mx[a]) <= eps; };
    // Identified node sets over the boundary nodes that lie on some periodic cell face. Matching is by DISTANCE, like the
    // reference's CollisionGrid::getClosestPoint(query, eps) (CollisionGrid.hh:58-88): the closest candidate within eps of
    // the translated position, found through a hash grid of cell size max(eps, 1e-7) whose cells overlapping
    // [q - eps, q + eps] are probed.
    std::vector<int32_t> faceNodes;                   // boundary nodes on a periodic face, in boundary-node order
    for (int32_t n : m.bdryNodes) {
        bool onFace = false;
        for (int a = 0; a < dim; ++a) onFace |= onMin(n, a) || onMax(n, a);
        if (onFace) faceNodes.push_back(n);
    }
    const double cs = std::max(eps, 1.0e-7);
    struct CellKey { int64_t q[3]; bool operator==(const CellKey &o) const { return q[0] == o.q[0] && q[1] == o.q[1] && q[2] == o.q[2]; } };
    auto hashOf = [](const CellKey &k) {
        uint64_t h = 1469598103934665603ull;
        for (int c = 0; c < 3; ++c) { h ^= (uint64_t)k.q[c]; h *= 1099511628211ull; h ^= h >> 29; }
        return (size_t)h; };
    // Hash grid in flat arrays (round 6: a std::unordered_map of std::vectors cost one heap allocation per cell -- with cells of 1e-7 that is one per point, 75 k of
    // them at configs[3]): open addressing over the cells, the points of one cell chained through `next`.
    struct Grid {
        std::vector<CellKey> key;
        std::vector<int32_t> first;        // slot -> first entry of the cell (-1: empty slot)
        std::vector<int32_t> node, next;   // entries
        size_t mask = 0;
        void reserve(size_t nPoints) {
            size_t cap = 16;
            while (cap < 2 * nPoints + 2) cap <<= 1;
            key.assign(cap, CellKey{{0, 0, 0}}); first.assign(cap, -1); mask = cap - 1;
            node.clear(); next.clear(); node.reserve(nPoints); next.reserve(nPoints);
        }
    };
    auto cellOf = [&](const double *p) { CellKey k{{0, 0, 0}}; for (int a = 0; a < dim; ++a) k.q[a] = (int64_t)std::floor(p[a] / cs); return k; };
    auto slotOf = [&](const Grid &g, const CellKey &k) {          // the slot of cell k, or the empty slot where it would go
        size_t sl = hashOf(k) & g.mask;
        while (g.first[sl] >= 0 && !(g.key[sl] == k)) sl = (sl + 1) & g.mask;
        return sl;
    };
    auto addPoint = [&](Grid &g, int32_t n) {
        const CellKey k = cellOf(&m.nodePos[(size_t)n * dim]);
        const size_t sl = slotOf(g, k);
        g.node.push_back(n);
        g.next.push_back(g.first[sl]);
        g.key[sl] = k;
        g.first[sl] = (int32_t)g.node.size() - 1;
    };
    auto closest = [&](const Grid &g, const double *q) -> int32_t {
        int64_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int a = 0; a < dim; ++a) { lo[a] = (int64_t)std::floor((q[a] - eps) / cs); hi[a] = (int64_t)std::floor((q[a] + eps) / cs); }
        int32_t best = -1;
        double bestDist = eps;
        for (int64_t i = lo[0]; i <= hi[0]; ++i)
            for (int64_t j = lo[1]; j <= hi[1]; ++j)
                for (int64_t k = lo[2]; k <= hi[2]; ++k) {
                    const size_t sl = slotOf(g, CellKey{{i, j, k}});
                    // (entries of a cell are chained newest first; ties in distance go to the LATEST candidate probed, like the loop over a cell's vector did:
                    // walk the chain into a small buffer and test oldest first)
                    int32_t chain[16];
                    int nc = 0;
                    for (int32_t en = g.first[sl]; en >= 0; en = g.next[(size_t)en]) { if (nc < 16) chain[nc++] = g.node[(size_t)en]; }
                    for (int t = nc - 1; t >= 0; --t) {
                        const int32_t c = chain[t];
                        double d2 = 0;
                        for (int a = 0; a < dim; ++a) { const double d = q[a] - m.nodePos[(size_t)c * dim + a]; d2 += d * d; }
                        const double dist = std::sqrt(d2);
                        if (dist <= bestDist) { bestDist = dist; best = c; }
                    }
                }
        return best;
    };
    auto isMinimal = [&](int32_t n) { for (int a = 0; a < dim; ++a) if (onMax(n, a)) return false; return true; };
    std::vector<int32_t> groupOf((size_t)m.nNode, -1);
    std::vector<std::vector<int32_t>> groups;
    if (!ignoreMismatch) {
        // PeriodicBoundaryMatcher::match (PeriodicBoundaryMatcher.hh:149-260): every MINIMAL node (on min faces only) on d
        // periodic faces looks up its 2^d - 1 translates among the non-minimal nodes; a missing translate, a node claimed
        // twice or a non-minimal node left over is an error
        Grid grid;
        grid.reserve(faceNodes.size());
        for (int32_t n : faceNodes) if (!isMinimal(n)) addPoint(grid, n);
        char buf[320];
        for (int32_t n : faceNodes) {
            if (!isMinimal(n)) continue;
            int dims[3], d = 0;
            for (int a = 0; a < dim; ++a) if (onMin(n, a)) dims[d++] = a;
            groupOf[n] = (int32_t)groups.size();
            groups.emplace_back(1, n);
            const double *p = &m.nodePos[(size_t)n * dim];
            for (int t = 1; t < (1 << d); ++t) {
                double q[3] = {p[0], p[1], dim == 3 ? p[2] : 0.0};
                for (int b = 0; b < d; ++b) if (t & (1 << b)) q[dims[b]] = mx[dims[b]];
                const int32_t r = closest(grid, q);
                if (r < 0) {
                    snprintf(buf, sizeof buf, "Couldn't find %dth periodic-identified node for minimal boundary node %d at (%g, %g, %g); looking for (%g, %g, %g)",
                             t, n, p[0], p[1], dim == 3 ? p[2] : 0.0, q[0], q[1], q[2]);
                    throw Error(MFH_ERR_INVALID, buf);
                }
                if (groupOf[r] >= 0) throw Error(MFH_ERR_INVALID, "Non bijective node set assignment.");
                groupOf[r] = groupOf[n];
                groups[(size_t)groupOf[n]].push_back(r);
            }
        }
        for (int32_t n : faceNodes)
            if (groupOf[n] < 0) {
                const double *p = &m.nodePos[(size_t)n * dim];
                snprintf(buf, sizeof buf, "Unmatched non-minimal boundary node %d at (%g, %g, %g)", n, p[0], p[1], dim == 3 ? p[2] : 0.0);
                throw Error(MFH_ERR_INVALID, buf);
            }
    } else {
        // PeriodicBoundaryMatcher::matchPermittingMismatch (:262-360): per dimension, every node of the max face looks up its
        // translate on the min face (a miss is a mismatch, not an error); node sets = connected components of the pair graph
        std::unordered_map<int32_t, std::array<int32_t, 3>> pair;
        for (int a = 0; a < dim; ++a) {
            Grid grid;
            grid.reserve(faceNodes.size());
            for (int32_t n : faceNodes) if (onMin(n, a)) addPoint(grid, n);
            for (int32_t n : faceNodes) {
                if (!onMax(n, a)) continue;
                const double *p = &m.nodePos[(size_t)n * dim];
                double q[3] = {p[0], p[1], dim == 3 ? p[2] : 0.0};
                q[a] = mn[a];
                const int32_t r = closest(grid, q);
                if (r < 0) continue;
                if (!pair.count(n)) pair[n] = {-1, -1, -1};
                if (!pair.count(r)) pair[r] = {-1, -1, -1};
                if (pair[n][a] != -1 || pair[r][a] != -1) throw Error(MFH_ERR_INVALID, "Non-bijective boundary matching");
                pair[n][a] = r;
                pair[r][a] = n;
            }
        }
        for (int32_t n : faceNodes) {
            if (groupOf[n] >= 0) continue;
            const int32_t gi = (int32_t)groups.size();
            groupOf[n] = gi;
            groups.emplace_back(1, n);
            for (size_t head = 0; head < groups[(size_t)gi].size(); ++head) {
                const int32_t u = groups[(size_t)gi][head];
                auto it = pair.find(u);
                if (it == pair.end()) continue;
                for (int a = 0; a < dim; ++a) {
                    const int32_t v = it->second[a];
                    if (v < 0 || groupOf[v] >= 0) continue;
                    groupOf[v] = gi;
                    groups[(size_t)gi].push_back(v);
                }
            }
        }
    }
    dofForNode.assign((size_t)m.nNode, -1);
    int32_t nd = 0;
    for (int64_t n = 0; n < m.nNode; ++n) {
        if (dofForNode[n] >= 0) continue;
        if (groupOf[n] >= 0) for (int32_t o : groups[groupOf[n]]) dofForNode[o] = nd;
        else dofForNode[n] = nd;
        ++nd;
    }
    nDoF = nd;
    // boundary elements whose nodes all lie on one (periodic) cell face are internal (PeriodicBoundaryMatcher.hh:127-145)
    const int64_t nBE = m.nBE();
    bdryInternal.assign((size_t)nBE, 0);
    for (int64_t b = 0; b < nBE; ++b) {
        const int32_t *bn = &m.bdryElemNodes[(size_t)b * m.npbe];
        bool internal = false;
        for (int a = 0; a < dim && !internal; ++a) {
            bool allMin = true, allMax = true;
            for (int k = 0; k < m.npbe; ++k) { allMin &= onMin(bn[k], a); allMax &= onMax(bn[k], a); }
            internal = allMin || allMax;
        }
        bdryInternal[b] = internal;
    }
}

} // namespace mfh
