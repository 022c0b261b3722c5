// A C++ Simulate_cli on the facade (the driver of src/bin/Simulate_cli.cc:91-245 with the includes and the namespace
// swapped): mesh.msh + .material + .bc -> u, load, strain, stress, Ku in an output .msh.
//
//     simulate_cli mesh.msh -m material.json -b conditions.bc -o out.msh [-d 1|2] [--device 0] [--ascii]
//                  [--preconditioner auto|block_jacobi|two_level|multigrid] [--rtol 1e-8] [--dumpMatrix K.bin]
//                  [--printMaterial]         (the parsed material in the reference's getJson form)
//                  [--dumpConditions file]   (after applying the conditions: fixed variables + values and the load, as
//                                             text, then exit -- works on a host-only context, --device -1)
//
// exit code 3 + "runtime_error: ..." on any std::runtime_error (no device, bad files, conflicting conditions, ...)
#include <MeshFEMHip/LinearElasticity.hh>
#include <MeshFEMHip/MeshIO.hh>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

using namespace MeshFEMHip;

struct Args {
    std::string mesh, material, boundaryConditions, outputMSH, dumpMatrix, dumpConditions, preconditioner = "auto";
    int degree = 2, device = 0;
    bool ascii = false, printMaterial = false;
    double rtol = 1e-8;
};

template <size_t N, size_t Deg>
int execute(const Args &args, const std::vector<MeshIO::IOVertex> &inVertices, const std::vector<MeshIO::IOElement> &inElements) {
    std::vector<std::array<Real, N>> V(inVertices.size());
    std::vector<std::array<int32_t, N + 1>> T(inElements.size());
    for (size_t i = 0; i < V.size(); ++i) {
        for (size_t c = 0; c < N; ++c) V[i][c] = inVertices[i][c];
        if (N == 2 && inVertices[i][2] != 0.0) throw std::runtime_error("2D simulation needs a planar (z = 0) triangle mesh");
    }
    for (size_t e = 0; e < T.size(); ++e)
        for (size_t c = 0; c < N + 1; ++c) T[e][c] = (int32_t)inElements[e][c];

    LinearElasticity::Simulator<N, Deg> sim(T, V, args.device);
    sim.rtol = args.rtol;
    if (args.device >= 0)
        // auto (default): the multigrid V-cycle unless the mesh as a whole is stretched past the measured crossover with the two-level preconditioner
        sim.setPreconditioner(args.preconditioner == "block_jacobi" ? MFH_PRECOND_BLOCK_JACOBI
                          : args.preconditioner == "multigrid" ? MFH_PRECOND_MULTIGRID
                          : args.preconditioner == "two_level" ? MFH_PRECOND_TWO_LEVEL : MFH_PRECOND_AUTO);
    if (!args.material.empty()) {
        Materials::Constant<N> mat(args.material);
        if (args.printMaterial) printf("material %s\n", mat.getJsonString().c_str());
        sim.setMaterial(mat);
    }
    if (!args.dumpMatrix.empty() && args.boundaryConditions.empty()) {        // Simulate_cli.cc:178-184
        TripletMatrix K;
        sim.m_assembleStiffnessMatrix(K);
        K.dumpBinary(args.dumpMatrix);
        return 0;
    }

    bool noRigidMotion = false;
    std::vector<PeriodicPairDirichletCondition<N>> pps;
    ComponentMask pinTranslation;
    auto bconds = readBoundaryConditions<N>(args.boundaryConditions, sim.boundingBox(), noRigidMotion, pps, pinTranslation);
    if (noRigidMotion) sim.applyNoRigidMotionConstraint();
    sim.applyTranslationPins(pinTranslation);
    sim.applyBoundaryConditions(bconds);
    sim.applyPeriodicPairDirichletConditions(pps);
    if (!args.dumpConditions.empty()) {
        int64_t nFixed = 0;
        check(sim.ctx(), mfh_bc_dirichlet_vars(sim.ctx(), nullptr, nullptr, &nFixed));
        std::vector<int64_t> vars((size_t)nFixed);
        std::vector<double> vals((size_t)nFixed);
        if (nFixed) check(sim.ctx(), mfh_bc_dirichlet_vars(sim.ctx(), vars.data(), vals.data(), &nFixed));
        auto load = sim.neumannLoad();
        FILE *out = fopen(args.dumpConditions.c_str(), "w");
        if (!out) throw std::runtime_error("Couldn't open output file " + args.dumpConditions);
        fprintf(out, "%d %lld %zu\n", noRigidMotion ? 1 : 0, (long long)nFixed, load.size());
        for (int64_t i = 0; i < nFixed; ++i) fprintf(out, "%lld %.17g\n", (long long)vars[(size_t)i], vals[(size_t)i]);
        for (const auto &l : load) { for (size_t c = 0; c < N; ++c) fprintf(out, "%.17g ", l[c]); fprintf(out, "\n"); }
        fclose(out);
        return 0;
    }

    auto u = sim.solve();
    printf("PCG: %d iterations, relative residual %.3e\n", sim.info.iterations, sim.info.true_rel_residual);
    auto f = sim.neumannLoad();
    auto e = sim.averageStrainField(u), s = sim.averageStressField(u);
    auto Ku = sim.applyStiffnessMatrix(u);

    // piecewise-linear subsample of the nodal fields (MSHFieldWriter.hh:74-83): vertex nodes come first
    using Domain = MSHFieldWriter::Domain;
    const size_t nv = V.size();
    auto head = [nv](const typename LinearElasticity::Simulator<N, Deg>::VField &x) { return typename LinearElasticity::Simulator<N, Deg>::VField(x.begin(), x.begin() + (long)nv); };
    MSHFieldWriter writer(args.outputMSH, V, T, !args.ascii);
    writer.addField("u", head(u), Domain::PER_NODE);
    writer.addField("load", head(f), Domain::PER_NODE);
    writer.addSymmetricMatrixField("strain", e, Domain::PER_ELEMENT);
    writer.addSymmetricMatrixField("stress", s, Domain::PER_ELEMENT);
    sim.reportRegionSurfaceForces(u);
    writer.addField("Ku", head(Ku), Domain::PER_NODE);
    writer.close();
    return 0;
}

int main(int argc, char **argv) {
    Args args;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&]() -> std::string {
            if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); }
            return argv[++i];
        };
        if (a == "-m" || a == "--material") args.material = value();
        else if (a == "-b" || a == "--boundaryConditions") args.boundaryConditions = value();
        else if (a == "-o" || a == "--outputMSH") args.outputMSH = value();
        else if (a == "-d" || a == "--degree") args.degree = atoi(value().c_str());
        else if (a == "--device") args.device = atoi(value().c_str());
        else if (a == "--rtol") args.rtol = atof(value().c_str());
        else if (a == "--preconditioner") args.preconditioner = value();
        else if (a == "--dumpMatrix") args.dumpMatrix = value();
        else if (a == "--dumpConditions") args.dumpConditions = value();
        else if (a == "--ascii") args.ascii = true;
        else if (a == "--printMaterial") args.printMaterial = true;
        else if (a[0] == '-') { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
        else args.mesh = a;
    }
    if (args.mesh.empty() || (args.outputMSH.empty() && args.dumpMatrix.empty() && args.dumpConditions.empty())) {
        fprintf(stderr, "usage: simulate_cli mesh.msh -m material -b conditions.bc -o out.msh [-d degree]\n");
        return 1;
    }
    if (!args.outputMSH.empty() && args.boundaryConditions.empty()) {
        fprintf(stderr, "Error: must specify boundary conditions to run a simulation\n");
        return 1;
    }
    try {
        std::vector<MeshIO::IOVertex> vertices;
        std::vector<MeshIO::IOElement> elements;
        auto type = MeshIO::load(args.mesh, vertices, elements);
        size_t dim = type == MeshIO::MeshType::TET ? 3 : type == MeshIO::MeshType::TRI ? 2 : 0;
        if (dim == 0) throw std::runtime_error("Mesh must be pure triangle or tet.");
        if (dim == 3 && args.degree == 2) return execute<3, 2>(args, vertices, elements);
        if (dim == 3 && args.degree == 1) return execute<3, 1>(args, vertices, elements);
        if (dim == 2 && args.degree == 2) return execute<2, 2>(args, vertices, elements);
        if (dim == 2 && args.degree == 1) return execute<2, 1>(args, vertices, elements);
        throw std::runtime_error("Unsupported degree");
    } catch (const std::runtime_error &e) {
        printf("runtime_error: %s\n", e.what());
        return 3;
    }
}
