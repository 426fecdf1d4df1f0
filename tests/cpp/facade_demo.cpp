// tests/cpp/facade_demo.cpp — the reference's call sites re-expressed against the C++ façade.
//
//   facade_demo                     host logic only (no GPU): decimation.h functions, IsometryXd, the job-line grammar
//                                   of src/main.cpp:9-90, result-file naming of src/evaluate.cpp:36-83
//   facade_demo gpu                 src/test_marginalize_se3.cpp:20-48 (3-pose SE3 chain, marginalise the middle pose,
//                                   Tree) with FIXED numbers, then optimize / marginalize / kullbackLeibler / chi2 /
//                                   covariance / clonePortion / vertex(id)->edges() / write(stream) on a small loop
//   facade_demo evaluate F.g2o DEST "<job line>"
//                                   src/evaluate.cpp:32-221 + src/main.cpp: evaluate() of one job through the abstract
//                                   GraphWrapper interface; prints the .kld series
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include "../../include/spg_evaluate.hpp"

static int host_checks() {
    DecimateOptions dopts{2, 1};
    DecimateFunction decimate = globalDecimate;
    std::vector<int> which = decimate(10, 10, dopts);
    if (which != std::vector<int>({5, 7, 9})) { std::printf("globalDecimate mismatch\n"); return 1; }
    if (!decimate(9, 10, dopts).empty() || onlineDecimate(7, 10, dopts) != std::vector<int>({7})) return 1;
    spg::IsometryXd a(std::vector<double>{0.1, 0.2, 0.3, 0.0, 0.0, 0.0998334166, 0.9950041653});
    spg::IsometryXd id = a * a.inverse();
    for (int i = 0; i < 3; i++) if (std::fabs(id.vector()[i]) > 1e-12) return 1;
    // a rotation of 0.2 rad about z: yaw = 0.2, roll = pitch = 0; ISAM order is (x y z yaw pitch roll)
    spg::VectorXd ei = a.vector(spg::IsometryXd::EulerAnglesISAM), eg = a.vector(spg::IsometryXd::EulerAnglesG2O);
    if (std::fabs(ei[3] - 0.2) > 1e-8 || std::fabs(eg[5] - 0.2) > 1e-8 || std::fabs(ei[5]) > 1e-12) return 1;
    if (a.vector(spg::IsometryXd::CondensedQuaternion).size() != 6) return 1;
    // job-line grammar (scripts/inputgenerator.sh:18-27)
    spg::EvaluateInfo j = spg::parseLine("glc datasets/intel.g2o online tree global 3 20 chi2 50");
    if (j.algorithm != spg::EvaluateInfo::GLC || j.decimate != onlineDecimate || j.sparsityOptions.topology != spg::SparsityOptions::Tree ||
        j.sparsityOptions.linPoint != spg::SparsityOptions::Global || j.decimateOptions.sparsity != 3 || j.kldPeriod != 20 || !j.useChi2 ||
        j.decimateOptions.clusterSize != 50) { std::printf("parseLine mismatch\n"); return 1; }
    spg::EvaluateInfo k = spg::parseLine("se3 /x/sphere.g2o global cldense local 2");
    if (k.algorithm != spg::EvaluateInfo::NFR || k.decimate != globalDecimate || k.kldPeriod != std::numeric_limits<int>::max() ||
        k.useChi2 || k.decimateOptions.clusterSize != 100 || k.sparsityOptions.topology != spg::SparsityOptions::CliqueyDense ||
        k.sparsityOptions.linPoint != spg::SparsityOptions::Local) { std::printf("parseLine defaults mismatch\n"); return 1; }
    k.destdir = "out";
    if (spg::resultStem(k, false, false) != "out/global/2/sphere/se3_cldense_l") { std::printf("resultStem mismatch: %s\n", spg::resultStem(k, false, false).c_str()); return 1; }
    std::printf("host ok\n");
    return 0;
}

static int run_evaluate(const char *g2o, const char *dest, const char *jobline) {
    spg::EvaluateInfo job = spg::parseLine(jobline);
    job.g2oname = g2o;
    job.destdir = dest;
    // src/evaluate.cpp:404-405: the backend object of the job (here the HIP backend), optimised at load
    spg::GraphWrapperHIP full(g2o, /*optimizeAtLoad=*/true, job.algorithm == spg::EvaluateInfo::GLC);
    spg::GraphWrapper *gw = &full;      // evaluate() sees the abstract interface only
    spg::EvaluateResult r = spg::evaluate(gw, job);
    for (auto &p : r.series) std::printf("kld %d %.17g\n", p.first, p.second);
    std::printf("stem %s\n", r.stem.c_str());
    return 0;
}

int main(int argc, char **argv) {
    if (int rc = host_checks()) return rc;
    if (argc >= 5 && std::strcmp(argv[1], "evaluate") == 0) return run_evaluate(argv[2], argv[3], argv[4]);
    if (argc < 2 || std::strcmp(argv[1], "gpu") != 0) {
        try {
            spg::GraphWrapperHIP w(6);
        } catch (const std::exception &e) {
            std::printf("no device: %s\n", e.what());  // expected on a GPU-less machine: there is no CPU fallback
        }
        return 0;
    }
    spg::GraphWrapperHIP w(6, /*useGLC=*/false);
    spg::IsometryXd m1(std::vector<double>{0.13, 0.08, 0.11, 0.04, 0.05, 0.06, 0.996}), m2(std::vector<double>{0.09, 0.12, 0.10, 0.05, 0.03, 0.05, 0.997});
    spg::IsometryXd x0(false), x1 = m1, x2 = m1 * m2;
    spg::MatrixXd I = spg::MatrixXd::Identity(6);
    w.addVertex(0, x0); w.addVertex(1, x1); w.addVertex(2, x2);
    spg::IsometryXd z(std::vector<double>{0.1, 0.1, 0.1, 0.05, 0.05, 0.05, 0.996});
    w.addEdge(0, 1, z, I);
    w.addEdge(1, 2, z, I);
    spg::SparsityOptions opts;
    opts.linPoint = spg::SparsityOptions::Global;
    opts.topology = spg::SparsityOptions::Tree;
    w.marginalize({1}, opts);
    auto es = w.edgeRecords();
    w.printStats(std::cout);
    std::printf("\nedges=%zu kld=%.3e\n", es.size(), w.lastKullbackLeiblerSum());
    // k = 2: a single new edge 0-2 that reproduces the target exactly
    if (es.size() != 1 || es[0].vertices != std::vector<int>({0, 2}) || std::fabs(w.lastKullbackLeiblerSum()) > 1e-9) return 2;

    // the rest of the interface on a 6-pose loop, through GraphWrapper pointers as the reference's callers hold them
    spg::GraphWrapperHIP full(6);
    spg::IsometryXd step(std::vector<double>{0.5, 0.0, 0.0, 0, 0, 0, 1}), closure(std::vector<double>{2.5, 0.05, 0.0, 0, 0, 0, 1});
    for (int i = 0; i < 6; i++) full.addVertex(i, spg::IsometryXd(std::vector<double>{0.5 * i + 0.01 * (i % 3), 0.02 * i, 0.0, 0, 0, 0, 1}));
    for (int i = 0; i + 1 < 6; i++) full.addEdge(i, i + 1, step, I);
    full.addEdge(0, 5, closure, I);
    spg::GraphWrapper *src = &full;
    std::unique_ptr<spg::GraphWrapper> base(src->clonePortion(5)), sparse(src->clonePortion(5)), head(src->clonePortion(3));
    if (head->vertices().size() != 4 || base->vertices().size() != 6) { std::printf("clonePortion mismatch\n"); return 3; }
    // vertex(id) views: id, estimate, incident edges with endpoints / measurement / information
    spg::GraphWrapper::Vertex *v5 = src->vertex(5);
    if (!v5 || v5->id() != 5 || !v5->is3d() || v5->edges().size() != 2 || src->vertex(17) != nullptr) { std::printf("vertex() mismatch\n"); return 3; }
    for (const spg::GraphWrapper::Edge *e : v5->edges()) {
        auto ends = e->vertices();
        if (ends.size() != 2 || (ends[0]->id() != 5 && ends[1]->id() != 5) || e->information().rows() != 6 || e->measurement().is2d()) return 3;
    }
    const double chi_before = base->chi2();
    base->optimize();
    sparse->optimize();
    const double chi_after = base->chi2();
    sparse->marginalize({2, 4}, opts);
    const double kld = base->kullbackLeibler(sparse.get());
    const double dchi = base->chi2(sparse.get()) - base->chi2();
    if (!(chi_after <= chi_before + 1e-12) || sparse->vertices().size() != 4 || !(kld > -1e-9) || !(dchi > -1e-6)) {
        std::printf("pipeline mismatch: chi2 %g -> %g, V = %zu, kld %g, dchi2 %g\n", chi_before, chi_after, sparse->vertices().size(), kld, dchi);
        return 3;
    }
    // covariance = information^-1 (src/graph_wrapper_g2o.cpp:368-373); estimate() stacks every vertex but the first
    spg::MatrixXd H = base->information(), S = base->covariance();
    double worst = 0;
    for (int i = 0; i < H.rows(); i++)
        for (int j = 0; j < H.cols(); j++) {
            double s = 0;
            for (int k2 = 0; k2 < H.cols(); k2++) s += H(i, k2) * S(k2, j);
            worst = std::max(worst, std::fabs(s - (i == j ? 1.0 : 0.0)));
        }
    if (H.rows() != 30 || worst > 1e-9 || base->estimate().size() != 30) { std::printf("covariance mismatch: n %d, |H S - I| %g\n", H.rows(), worst); return 3; }
    std::ostringstream text, dbg;
    sparse->write(text);
    dbg << *sparse;
    if (text.str().find("VERTEX_SE3:QUAT 0 ") == std::string::npos || text.str().find("EDGE_SE3:QUAT") == std::string::npos ||
        dbg.str().find("+ vertices: 0 1 3 5") == std::string::npos) { std::printf("write/debugPrint mismatch\n%s\n", dbg.str().c_str()); return 3; }
    std::printf("chi2 %.4g -> %.4g; marginalize + optimize; global KLD %.4g, delta chi2 %.4g; |H S - I| %.2e\n", chi_before, chi_after, kld, dchi, worst);
    sparse->printStats(std::cout);
    std::printf("\ngpu ok\n");
    return 0;
}
