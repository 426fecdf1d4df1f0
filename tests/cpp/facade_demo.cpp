// tests/cpp/facade_demo.cpp — the reference's smallest end-to-end example
// (src/test_marginalize_se3.cpp:20-48: 3-pose SE3 chain, marginalise the middle pose, Tree topology)
// written against the C++ façade, with FIXED numbers instead of Eigen::Random(), plus the
// decimation.h calls of src/evaluate.cpp:125.  `facade_demo` alone exercises only host logic (no
// GPU needed); `facade_demo gpu` runs the marginalisation on device 0 and prints the result.
#include <cstdio>
#include <cstring>
#include <iostream>
#include "../../include/spg_graph_wrapper.hpp"

int main(int argc, char **argv) {
    DecimateOptions dopts{2, 1};
    DecimateFunction decimate = globalDecimate;
    std::vector<int> which = decimate(10, 10, dopts);
    if (which != std::vector<int>({5, 7, 9})) { std::printf("globalDecimate mismatch\n"); return 1; }
    if (!decimate(9, 10, dopts).empty() || onlineDecimate(7, 10, dopts) != std::vector<int>({7})) return 1;
    spg::IsometryXd a(std::vector<double>{0.1, 0.2, 0.3, 0.0, 0.0, 0.0998334166, 0.9950041653});
    spg::IsometryXd id = a * a.inverse();
    for (int i = 0; i < 3; i++) if (std::fabs(id.vector()[i]) > 1e-12) return 1;
    std::printf("host ok\n");
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
    std::vector<double> I(36, 0.0);
    for (int i = 0; i < 6; i++) I[i * 7] = 1.0;
    w.addVertex(0, x0); w.addVertex(1, x1); w.addVertex(2, x2);
    spg::IsometryXd z(std::vector<double>{0.1, 0.1, 0.1, 0.05, 0.05, 0.05, 0.996});
    w.addEdge(0, 1, z, I);
    w.addEdge(1, 2, z, I);
    spg::SparsityOptions opts;
    opts.linPoint = spg::SparsityOptions::Global;
    opts.topology = spg::SparsityOptions::Tree;
    w.marginalize({1}, opts);
    auto es = w.edges();
    w.printStats(std::cout);
    std::printf("\nedges=%zu kld=%.3e\n", es.size(), w.lastKullbackLeiblerSum());
    // k = 2: a single new edge 0-2 that reproduces the target exactly
    if (es.size() != 1 || es[0].vertices != std::vector<int>({0, 2}) || std::fabs(w.lastKullbackLeiblerSum()) > 1e-9) return 2;
    // the reference's evaluation flow on a longer chain with a loop closure: optimize() the baseline,
    // marginalize() a copy (= marginalizeNoOptimize + optimize), global KLD between the two
    // (GraphWrapperG2O::optimize / ::marginalize / ::kullbackLeibler, src/graph_wrapper_g2o.cpp:250-269,455-463,531-548)
    spg::GraphWrapperHIP base(6), sparse(6);
    spg::IsometryXd step(std::vector<double>{0.5, 0.0, 0.0, 0, 0, 0, 1}), closure(std::vector<double>{2.5, 0.05, 0.0, 0, 0, 0, 1});
    for (spg::GraphWrapperHIP *gw : {&base, &sparse}) {
        for (int i = 0; i < 6; i++) gw->addVertex(i, spg::IsometryXd(std::vector<double>{0.5 * i + 0.01 * (i % 3), 0.02 * i, 0.0, 0, 0, 0, 1}));
        for (int i = 0; i + 1 < 6; i++) gw->addEdge(i, i + 1, step, I);
        gw->addEdge(0, 5, closure, I);
    }
    spg_optimize_stats ob = base.optimize();
    sparse.optimize();
    sparse.marginalize({2, 4}, opts);
    spg_kld_terms terms;
    double kld = base.kullbackLeibler(&sparse, &terms);
    if (!(ob.chi2_final <= ob.chi2_initial) || sparse.vertices().size() != 4 || !(kld > -1e-9) || terms.n != 6 * 3) {
        std::printf("pipeline mismatch: chi2 %g -> %g, V = %zu, kld %g, n %lld\n", ob.chi2_initial, ob.chi2_final, sparse.vertices().size(), kld, (long long)terms.n);
        return 3;
    }
    std::printf("optimize chi2 %.4g -> %.4g in %d iterations; marginalize + optimize; global KLD %.4g over %lld variables\n",
                ob.chi2_initial, ob.chi2_final, ob.iterations, kld, (long long)terms.n);
    std::printf("gpu ok\n");
    return 0;
}
