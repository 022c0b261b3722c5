// `mesh` (== /root/reference/src/python_bindings/mesh.cc:42-134,258-330): Mesh(path | V, F, degree, embeddingDimension) and
// PeriodicCondition. Node numbering, boundary extraction and periodic matching are the library's host-side restatements of
// FEMMesh.inl / TetMesh.inl / PeriodicBoundaryMatcher.hh behind the C ABI (a device -1 context: no GPU is needed for this module).
#include "common.hh"

#include <memory>

namespace {

struct HostCtx {
    mfh_ctx *c = nullptr;
    HostCtx() { if (mfh_create(-1, &c) != MFH_OK) throw std::runtime_error("mfh_create(-1) failed"); }
    ~HostCtx() { if (c) mfh_destroy(c); }
    HostCtx(const HostCtx &) = delete;
    HostCtx &operator=(const HostCtx &) = delete;
    void ck(mfh_status st) const { if (st != MFH_OK) throw std::runtime_error(mfh_last_error(c)); }
};

struct Mesh {
    size_t N = 3, K = 3, degree = 1;          // embedding dimension, simplex dimension
    std::vector<Real> V;                      // nV x N
    std::vector<int32_t> F;                   // nE x (K + 1)
    std::shared_ptr<HostCtx> h;
    int64_t nElem = 0, nNode = 0, nVert = 0, nBE = 0, nBN = 0;
    int32_t npe = 0, npbe = 0;

    Mesh(const ArrD &Vin, const ArrI &Fin, size_t deg, size_t embeddingDimension) {
        if (Vin.ndim() != 2 || Fin.ndim() != 2) throw std::runtime_error("V and F must be two-dimensional arrays");
        K = (size_t)Fin.shape(1) - 1;
        const size_t vd = (size_t)Vin.shape(1);
        N = (K == 3 || embeddingDimension == 2) ? K : vd;
        auto v = Vin.unchecked<2>();
        if (K == 2 && N == 3 && embeddingDimension != 3) {
            bool flat = true;
            for (py::ssize_t i = 0; i < Vin.shape(0) && flat; ++i) flat = vd < 3 || v(i, 2) == 0.0;
            if (flat) N = 2;
        }
        if ((N != 2 && N != 3) || N != K) throw std::runtime_error("only tet meshes in 3D and triangle meshes in 2D are on the GPU path");
        if (deg != 1 && deg != 2) throw std::runtime_error("degree must be 1 or 2");
        degree = deg;
        nVert = Vin.shape(0);
        V.resize((size_t)nVert * N);
        for (int64_t i = 0; i < nVert; ++i) for (size_t a = 0; a < N; ++a) V[(size_t)i * N + a] = v(i, a);
        auto f = Fin.unchecked<2>();
        F.resize((size_t)Fin.shape(0) * (K + 1));
        for (py::ssize_t e = 0; e < Fin.shape(0); ++e) for (size_t k = 0; k <= K; ++k) F[(size_t)e * (K + 1) + k] = (int32_t)f(e, k);
        build();
    }
    void build() {
        h = std::make_shared<HostCtx>();
        h->ck(mfh_mesh_build(h->c, (int32_t)N, (int32_t)degree, (int64_t)(F.size() / (K + 1)), nVert, F.data(), V.data()));
        h->ck(mfh_mesh_sizes(h->c, &nElem, &nNode, &nVert, &nBE, &nBN, &npe, &npbe));
    }
    ArrD vertices() const { ArrD o = make2d((size_t)nVert, N); std::copy(V.begin(), V.end(), o.mutable_data()); return o; }
    ArrI elements() const {
        ArrI o(std::vector<py::ssize_t>{(py::ssize_t)nElem, (py::ssize_t)(K + 1)});
        for (size_t k = 0; k < F.size(); ++k) o.mutable_data()[k] = F[k];
        return o;
    }
    ArrD nodes() const { ArrD o = make2d((size_t)nNode, N); h->ck(mfh_mesh_get_node_positions(h->c, o.mutable_data())); return o; }
    ArrI elementNodes() const {
        std::vector<int32_t> en((size_t)nElem * npe);
        h->ck(mfh_mesh_get_elem_nodes(h->c, en.data()));
        ArrI o(std::vector<py::ssize_t>{(py::ssize_t)nElem, (py::ssize_t)npe});
        for (size_t k = 0; k < en.size(); ++k) o.mutable_data()[k] = en[k];
        return o;
    }
    ArrI boundaryNodes() const {
        std::vector<int32_t> bn((size_t)nBN);
        h->ck(mfh_mesh_get_boundary_nodes(h->c, bn.data()));
        ArrI o((py::ssize_t)nBN);
        for (size_t k = 0; k < bn.size(); ++k) o.mutable_at(k) = bn[k];
        return o;
    }
    ArrI boundaryElements() const {
        std::vector<int32_t> be((size_t)nBE * npbe);
        h->ck(mfh_mesh_get_boundary_elem_nodes(h->c, be.data()));
        ArrI o(std::vector<py::ssize_t>{(py::ssize_t)nBE, (py::ssize_t)K});
        for (int64_t b = 0; b < nBE; ++b) for (size_t k = 0; k < K; ++k) o.mutable_at(b, k) = be[(size_t)b * npbe + k];
        return o;
    }
    ArrD elementVolumes() const {
        ArrD o((py::ssize_t)nElem);
        for (int64_t e = 0; e < nElem; ++e) {
            const int32_t *f = &F[(size_t)e * (K + 1)];
            auto P = [&](int k, size_t a) { return V[(size_t)f[k] * N + a]; };
            if (K == 3) {
                double d[3][3];
                for (int r = 0; r < 3; ++r) for (size_t a = 0; a < 3; ++a) d[r][a] = P(r + 1, a) - P(0, a);
                o.mutable_at(e) = (d[0][0] * (d[1][1] * d[2][2] - d[1][2] * d[2][1]) - d[0][1] * (d[1][0] * d[2][2] - d[1][2] * d[2][0]) +
                                   d[0][2] * (d[1][0] * d[2][1] - d[1][1] * d[2][0])) / 6.0;
            } else
                o.mutable_at(e) = 0.5 * ((P(1, 0) - P(0, 0)) * (P(2, 1) - P(0, 1)) - (P(1, 1) - P(0, 1)) * (P(2, 0) - P(0, 0)));
        }
        return o;
    }
    std::pair<ArrD, ArrD> bbox() const {
        ArrD lo((py::ssize_t)N), hi((py::ssize_t)N);
        for (size_t a = 0; a < N; ++a) { lo.mutable_at(a) = 1e300; hi.mutable_at(a) = -1e300; }
        for (int64_t i = 0; i < nVert; ++i) for (size_t a = 0; a < N; ++a) {
            lo.mutable_at(a) = std::min(lo.at(a), V[(size_t)i * N + a]); hi.mutable_at(a) = std::max(hi.at(a), V[(size_t)i * N + a]);
        }
        return {lo, hi};
    }
};

struct PeriodicCondition {
    std::vector<int32_t> dofs;
    int64_t nDoF = 0;
    PeriodicCondition(const Mesh &m, Real eps, bool ignoreMismatch, const std::vector<int> &ignoreDims) {
        HostCtx h;
        h.ck(mfh_mesh_build(h.c, (int32_t)m.N, (int32_t)m.degree, m.nElem, m.nVert, m.F.data(), m.V.data()));
        int mask = 0;
        for (int d : ignoreDims) mask |= 1 << d;
        h.ck(mfh_set_option(h.c, "periodic_ignore_mismatch", ignoreMismatch ? 1.0 : 0.0));
        h.ck(mfh_set_option(h.c, "periodic_ignore_dims", (double)mask));
        h.ck(mfh_apply_periodic_conditions(h.c, eps, &nDoF));
        dofs.resize((size_t)m.nNode);
        int64_t n2 = 0;
        h.ck(mfh_get_dof_map(h.c, dofs.data(), &n2));
    }
};

}   // namespace

PYBIND11_MODULE(mesh, m) {
    m.doc() = "MeshFEM finite element mesh data structure bindings (MI355X path: host topology through libmeshfem_hip)";
    py::class_<Mesh, std::shared_ptr<Mesh>>(m, "FEMMesh")
        .def("vertices", &Mesh::vertices).def("nodes", &Mesh::nodes).def("elements", &Mesh::elements).def("elementNodes", &Mesh::elementNodes)
        .def("boundaryNodes", &Mesh::boundaryNodes).def("boundaryElements", &Mesh::boundaryElements)
        .def("numVertices", [](const Mesh &a) { return a.nVert; }).def("numElements", [](const Mesh &a) { return a.nElem; })
        .def("numNodes", [](const Mesh &a) { return a.nNode; })
        .def("elementVolumes", &Mesh::elementVolumes)
        .def("setVertices", [](Mesh &a, const ArrD &Vn) {
            if (Vn.ndim() != 2 || Vn.shape(0) != a.nVert || (size_t)Vn.shape(1) < a.N) throw std::runtime_error("bad vertex array");
            auto v = Vn.unchecked<2>();
            for (int64_t i = 0; i < a.nVert; ++i) for (size_t k = 0; k < a.N; ++k) a.V[(size_t)i * a.N + k] = v(i, k);
            a.h->ck(mfh_mesh_update_vertices(a.h->c, a.V.data()));          // same connectivity: no topology rebuild
        })
        .def("copy", [](const Mesh &a) { auto b = std::make_shared<Mesh>(a); b->build(); return b; })
        .def_property_readonly("bbox", [](const Mesh &a) { auto b = a.bbox(); return py::make_tuple(b.first, b.second); })
        .def_property_readonly("bbox_volume", [](const Mesh &a) { auto b = a.bbox(); double v = 1; for (size_t k = 0; k < a.N; ++k) v *= b.second.at(k) - b.first.at(k); return v; })
        .def_property_readonly("volume", [](const Mesh &a) { auto v = a.elementVolumes(); double s = 0; for (py::ssize_t k = 0; k < v.size(); ++k) s += v.at(k); return s; })
        .def_property_readonly("degree", [](const Mesh &a) { return a.degree; })
        .def_property_readonly("simplexDimension", [](const Mesh &a) { return a.K; })
        .def_property_readonly("embeddingDimension", [](const Mesh &a) { return a.N; });
    // Mesh(path, degree = 1, embeddingDimension = 3) / Mesh(V, F, degree = 1, embeddingDimension = 3)   (mesh.cc:293-330)
    m.def("Mesh", [](const std::string &path, size_t degree, size_t embeddingDimension) {
        py::tuple vf = py::module::import("meshfem_amd.mesh_io").attr("load_mesh")(path);      // MSH / OFF / OBJ / MEDIT readers
        return std::make_shared<Mesh>(vf[0].cast<ArrD>(), vf[1].cast<ArrI>(), degree, embeddingDimension);
    }, py::arg("path"), py::arg("degree") = 1, py::arg("embeddingDimension") = 3);
    m.def("Mesh", [](const ArrD &V, const ArrI &F, size_t degree, size_t embeddingDimension) { return std::make_shared<Mesh>(V, F, degree, embeddingDimension); },
          py::arg("V"), py::arg("F"), py::arg("degree") = 1, py::arg("embeddingDimension") = 3);

    py::class_<PeriodicCondition>(m, "PeriodicCondition")
        .def(py::init([](const Mesh &mesh, Real eps, bool ignoreMismatch, const std::vector<int> &ignoreDims) { return PeriodicCondition(mesh, eps, ignoreMismatch, ignoreDims); }),
             py::arg("mesh"), py::arg("eps") = 1e-7, py::arg("ignore_mismatch") = false, py::arg("ignore_dims") = std::vector<int>())
        .def("periodicDoFsForNodes", [](const PeriodicCondition &p) { ArrI o((py::ssize_t)p.dofs.size()); for (size_t k = 0; k < p.dofs.size(); ++k) o.mutable_at(k) = p.dofs[k]; return o; })
        .def("numPeriodicDoFs", [](const PeriodicCondition &p) { return p.nDoF; });
}
