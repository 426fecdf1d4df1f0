// include/spg_evaluate.hpp — the reference's evaluation driver on top of the GraphWrapper interface, header-only:
//
//   struct EvaluateInfo                       src/evaluate.h:16-29
//   parseLine / loadEvaluateInfo              src/main.cpp:9-121   (the job-file grammar documented in
//                                             scripts/inputgenerator.sh:18-27:  <alg> <file.g2o> <online|cluster|global>
//                                             <tree|subgr|clsubgr|dense|cldense> <local|global> <sparsity>
//                                             [kldPeriod] [chi2|kld] [clusterSize])
//   evaluate(gw, info)                        src/evaluate.cpp:32-221: replay the graph vertex by vertex into an
//                                             incremental and a baseline graph, decimate, substitute edges to removed
//                                             vertices, optimise, marginalise, record the KLD (or delta chi2) series
//   result files                              <dest>/<profile>/<sparsity>/<dataset>/<alg>_<topo>_<l|g>.kld  ("<i> <kld>" per
//                                             line) and .txt (algorithm, nodes / edges / fill-in of both graphs, last value),
//                                             src/evaluate.cpp:62-97,199-204
//
// The job farm around it (pthreads, memory heuristic, MySQL scheduler: src/evaluate.cpp:223-433, scripts/*) is out of
// scope; a caller loops over loadEvaluateInfo()'s jobs itself.
#pragma once
#include <sys/stat.h>

#include <fstream>
#include <limits>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "spg_graph_wrapper.hpp"

namespace spg {

struct EvaluateInfo {
    enum Algorithm { NFR, GLC, None };
    std::string g2oname;
    std::string destdir;
    Algorithm algorithm = NFR;
    bool useChi2 = false;
    int kldPeriod = 10;
    DecimateFunction decimate = globalDecimate;
    DecimateOptions decimateOptions{2, 100};
    SparsityOptions sparsityOptions;
};

namespace detail {
inline std::string lower(std::string s) { for (char &c : s) c = (char)std::tolower((unsigned char)c); return s; }
}

// One job line (src/main.cpp:9-90). Unknown words select the same defaults as the reference's else-branches.
inline EvaluateInfo parseLine(const std::string &line) {
    EvaluateInfo job;
    std::istringstream in(line);
    std::string w;
    in >> w; w = detail::lower(w);
    job.algorithm = (w == "glc") ? EvaluateInfo::GLC : (w == "none") ? EvaluateInfo::None : EvaluateInfo::NFR;
    in >> job.g2oname;
    in >> w; w = detail::lower(w);
    job.decimate = (w == "online") ? onlineDecimate : (w == "cluster") ? clusterDecimate : globalDecimate;
    in >> w; w = detail::lower(w);
    job.sparsityOptions.topology = (w == "tree") ? SparsityOptions::Tree : (w == "subgr") ? SparsityOptions::Subgraph
                                 : (w == "clsubgr") ? SparsityOptions::CliqueySubgraph : (w == "dense") ? SparsityOptions::Dense
                                 : SparsityOptions::CliqueyDense;
    in >> w; w = detail::lower(w);
    job.sparsityOptions.linPoint = (w == "local") ? SparsityOptions::Local : SparsityOptions::Global;
    in >> job.decimateOptions.sparsity;
    job.kldPeriod = 10;
    if (in.good()) in >> job.kldPeriod;
    if (job.decimate == globalDecimate) job.kldPeriod = std::numeric_limits<int>::max();
    job.useChi2 = false;
    if (in.good()) { in >> w; job.useChi2 = (detail::lower(w) == "chi2"); }
    job.decimateOptions.clusterSize = 100;
    if (in.good()) in >> job.decimateOptions.clusterSize;
    return job;
}

// The job file: one job per line, blank lines and '#' comments skipped (src/main.cpp:104-121)
inline std::vector<EvaluateInfo> loadEvaluateInfo(const char *destdir, const char *infoname) {
    std::vector<EvaluateInfo> jobs;
    std::ifstream f(infoname);
    std::string line;
    while (std::getline(f, line)) {
        size_t a = line.find_first_not_of(" \t\r\n");
        if (a == std::string::npos || line[a] == '#') continue;
        size_t b = line.find_last_not_of(" \t\r\n");
        jobs.push_back(parseLine(line.substr(a, b - a + 1)));
        jobs.back().destdir = destdir;
    }
    return jobs;
}

// <dest>/<profile>/<sparsity>/<dataset>/<alg>_<topo>_<l|g> without extension (src/evaluate.cpp:36-83); creates the
// directories when `make_dirs`.
inline std::string resultStem(const EvaluateInfo &info, bool is2d, bool make_dirs, std::string *longtype = nullptr, std::string *algname = nullptr) {
    static const char *shortn[] = {"tree", "subgr", "clsubgr", "dense", "cldense"};
    static const char *longn[] = {"Tree", "Subgraph", "Cliquey Subgraph", "Dense", "Cliquey Dense"};
    std::string alg = info.algorithm == EvaluateInfo::NFR ? (is2d ? "se2" : "se3") : info.algorithm == EvaluateInfo::GLC ? "glc" : "none";
    std::string profile = info.decimate == onlineDecimate ? "online" : info.decimate == clusterDecimate ? "cluster"
                        : info.decimate == globalDecimate ? "global" : "unknown";
    size_t slash = info.g2oname.rfind('/'), dot = info.g2oname.rfind('.');
    size_t b = slash == std::string::npos ? 0 : slash + 1, e = dot == std::string::npos ? info.g2oname.size() : dot;
    std::string dataset = info.g2oname.substr(b, e > b ? e - b : 0);
    std::string dirs[4] = {info.destdir, info.destdir + "/" + profile, "", ""};
    dirs[2] = dirs[1] + "/" + std::to_string(info.decimateOptions.sparsity);
    dirs[3] = dirs[2] + "/" + dataset;
    if (make_dirs) for (const std::string &d : dirs) mkdir(d.c_str(), 0755);
    if (longtype) *longtype = longn[(int)info.sparsityOptions.topology];
    if (algname) *algname = alg;
    return dirs[3] + "/" + alg + "_" + shortn[(int)info.sparsityOptions.topology] + "_" + (info.sparsityOptions.linPoint == SparsityOptions::Local ? "l" : "g");
}

struct EvaluateResult {
    std::vector<std::pair<int, double>> series;   // the lines of the .kld file
    double last = 0;
    std::string stem;                              // result files are stem + ".kld" / ".txt" ("" when nothing was written)
};

// src/evaluate.cpp:32-221. gw holds the FULL graph (ids 0..last). write_files = false keeps everything in memory.
// RESTRICTION (a deviation from the reference, which takes any GraphWrapper): the source graph must be a GraphWrapperHIP.
// Everything else in the loop goes through the abstract interface; computeSubstituteEdge (src/compute_substitute_edge.cpp:
// 13-96) does not — it is one C-ABI call (spg_graph_substitute_edge: breadth-first search to the nearest surviving vertex,
// composition of the measurements along the path, sum of the covariances) on the library's own graph object instead of a
// walk over vertex(id)->edges() with Eigen inverses, which this Eigen-free header does not carry. A caller with another
// backend gets a std::runtime_error here, before anything is computed.
inline EvaluateResult evaluate(GraphWrapper *gw, const EvaluateInfo &info, bool write_files = true) {
    EvaluateResult res;
    GraphWrapperHIP *source = dynamic_cast<GraphWrapperHIP *>(gw);
    if (!source) throw std::runtime_error("evaluate: computeSubstituteEdge needs a GraphWrapperHIP source graph (see the note above evaluate() in spg_evaluate.hpp)");
    std::unique_ptr<GraphWrapper> incremental(gw->clonePortion(3)), baseline(gw->clonePortion(3));
    const bool is2d = gw->vertex(1)->is2d();
    const bool sparsify = info.algorithm != EvaluateInfo::None;
    std::string longtype, alg;
    std::ofstream kldf, txtf;
    if (write_files) {
        res.stem = resultStem(info, is2d, true, &longtype, &alg);
        kldf.open((res.stem + ".kld").c_str());
        txtf.open((res.stem + ".txt").c_str());
    } else {
        (void)resultStem(info, is2d, false, &longtype, &alg);
    }
    auto record = [&](int i, double v) {
        res.series.push_back({i, v});
        res.last = v;
        if (write_files) kldf << i << " " << v << std::endl;
    };
    std::set<int> removed;
    const int lastid = gw->vertices().back()->id();
    for (int i = 4; i <= lastid; i++) {
        GraphWrapper::Vertex *latest = gw->vertex(i);
        if (!latest) throw std::runtime_error("evaluate: vertex ids must be contiguous");
        incremental->addVertex(i, latest->estimate());
        baseline->addVertex(i, latest->estimate());
        for (const GraphWrapper::Edge *e : latest->edges()) {
            std::vector<const GraphWrapper::Vertex *> ends = e->vertices();
            if (ends.size() != 2) continue;
            int from = ends[0]->id(), to = ends[1]->id();
            if (from > i || to > i) continue;
            const int other = (from == i) ? to : from;
            MatrixXd einfo;
            IsometryXd emeas(is2d);
            if (removed.count(other)) {
                computeSubstituteEdge(source, std::vector<int>(removed.begin(), removed.end()), i, from, to, emeas, einfo);
            } else {
                einfo = e->information();
                emeas = e->measurement();
            }
            incremental->addEdge(from, to, emeas, einfo);
            baseline->addEdge(from, to, emeas, einfo);
        }
        const std::vector<int> which = info.decimate(i, lastid, info.decimateOptions);
        const bool report = (i % info.kldPeriod == 0) || i == lastid;
        if (sparsify && (!which.empty() || report)) {
            incremental->optimize();
            baseline->optimize();
        }
        if (!which.empty() && sparsify) incremental->marginalize(which, info.sparsityOptions);
        removed.insert(which.begin(), which.end());
        if (!report) continue;
        if (!sparsify) {
            baseline->optimize();
            record(i, info.useChi2 ? baseline->chi2() : 0.0);
        } else if (info.useChi2) {
            record(i, baseline->chi2(incremental.get()) - baseline->chi2());
        } else {
            record(i, baseline->kullbackLeibler(incremental.get()));
        }
    }
    if (write_files) {
        for (char &c : alg) c = (char)std::toupper((unsigned char)c);
        txtf << alg << " " << longtype << std::endl << "    baseline:     ";
        baseline->printStats(txtf);
        txtf << std::endl << "    marginalized: ";
        incremental->printStats(txtf);
        txtf << std::endl << "    last " << (info.useChi2 ? "chi2: " : "kld: ") << res.last << std::endl;
    }
    return res;
}

}  // namespace spg
