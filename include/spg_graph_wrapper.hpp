// include/spg_graph_wrapper.hpp — C++ host façade over the C ABI (include/spg.h), header-only.
//
// Mirrors the reference's operator interface for the accelerated path so that its call sites
// (src/evaluate.cpp:103-179, src/test_marginalize_se3.cpp:20-48,
// src/test_marginalize_within_window.cpp:117-182) keep their shape:
//
//   reference                                          here
//   ------------------------------------------------   -----------------------------------------
//   class GraphWrapper        (src/graph_wrapper.h:17)  spg::GraphWrapper (same virtuals that
//                                                        exist without g2o's optimiser)
//   class GraphWrapperG2O     (src/graph_wrapper_g2o.h:29)  spg::GraphWrapperHIP
//   struct SparsityOptions    (src/sparsity_options.h:11)   spg::SparsityOptions (same fields/defaults)
//   struct DecimateOptions, globalDecimate, onlineDecimate, clusterDecimate, DecimateFunction
//                             (src/decimation.h:13-22)      identical names and signatures
//   class IsometryXd          (src/isometryxd.h:14)         spg::IsometryXd (SE2: x y theta;
//                                                            SE3: t + unit quaternion), compose /
//                                                            inverse / vector()
//
// Eigen is not a dependency: information matrices travel as row-major std::vector<double> (d*d).
// optimize() (g2o Levenberg-Marquardt, src/graph_wrapper_g2o.cpp:250-269) runs dense on the device for
// graphs of up to 32k scalar variables; marginalize() = marginalizeNoOptimize() + optimize() as in the reference.
// Errors: the reference asserts/aborts; this façade throws std::runtime_error with the library's
// message. Thread model: one GraphWrapperHIP per host thread (as the reference's one VertexRemover
// per call, src/vertex_remover.h:92-98).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "spg.h"

// ---- src/decimation.h:13-22, verbatim API -------------------------------------------------
struct DecimateOptions {
    int sparsity;
    int clusterSize;
};
inline std::vector<int> spg_decimate_call_(int (*fn)(int, int, int, int, int32_t *, int), int last, int endvert,
                                           const DecimateOptions &o) {
    std::vector<int32_t> buf((size_t)(endvert > 0 ? endvert : 0) + 8);
    int n = fn(last, endvert, o.sparsity, o.clusterSize, buf.data(), (int)buf.size());
    return std::vector<int>(buf.begin(), buf.begin() + n);
}
inline std::vector<int> clusterDecimate(int last, int endvert, const DecimateOptions &o) { return spg_decimate_call_(spg_decimate_cluster, last, endvert, o); }
inline std::vector<int> onlineDecimate(int last, int endvert, const DecimateOptions &o) { return spg_decimate_call_(spg_decimate_online, last, endvert, o); }
inline std::vector<int> globalDecimate(int last, int endvert, const DecimateOptions &o) { return spg_decimate_call_(spg_decimate_global, last, endvert, o); }
typedef std::vector<int> (*DecimateFunction)(int last, int endvert, const DecimateOptions &opts);

namespace spg {

// ---- src/sparsity_options.h:11-30 -----------------------------------------------------------
struct SparsityOptions {
    enum SparsityTopology { Tree, Subgraph, CliqueySubgraph, Dense, CliqueyDense };
    enum LinearizationPoint { Local, Global };
    SparsityTopology topology;
    double chordRatio;
    LinearizationPoint linPoint;
    bool includeIntraClique;
    SparsityOptions() : topology(Tree), chordRatio(1), linPoint(Local), includeIntraClique(true) {}
};

// ---- src/isometryxd.h:14-47 (the parts the path needs) ----------------------------------------
class IsometryXd {
public:
    explicit IsometryXd(bool is2d = true) : _2d(is2d) {
        if (is2d) _v = {0, 0, 0};
        else _v = {0, 0, 0, 0, 0, 0, 1};
    }
    // SE2: (x, y, theta); SE3: (tx, ty, tz, qx, qy, qz, qw) — IsometryXd::Quaternion mode
    explicit IsometryXd(const std::vector<double> &v) : _2d(v.size() == 3), _v(v) {
        if (v.size() != 3 && v.size() != 7) throw std::runtime_error("IsometryXd: need 3 (SE2) or 7 (SE3, t+quat) numbers");
        if (!_2d) {
            double n = std::sqrt(_v[3] * _v[3] + _v[4] * _v[4] + _v[5] * _v[5] + _v[6] * _v[6]);
            for (int i = 3; i < 7; i++) _v[i] /= n;
        }
    }
    bool is2d() const { return _2d; }
    bool is3d() const { return !_2d; }
    const std::vector<double> &vector() const { return _v; }

    IsometryXd inverse() const {
        if (_2d) {
            double c = std::cos(_v[2]), s = std::sin(_v[2]);
            return IsometryXd(std::vector<double>{-(c * _v[0] + s * _v[1]), -(-s * _v[0] + c * _v[1]), wrap(-_v[2])});
        }
        double q[4] = {-_v[3], -_v[4], -_v[5], _v[6]}, t[3];
        rot(q, &_v[0], t);
        return IsometryXd(std::vector<double>{-t[0], -t[1], -t[2], q[0], q[1], q[2], q[3]});
    }
    IsometryXd operator*(const IsometryXd &o) const {
        if (_2d != o._2d) throw std::runtime_error("Incompatible product");
        if (_2d) {
            double c = std::cos(_v[2]), s = std::sin(_v[2]);
            return IsometryXd(std::vector<double>{_v[0] + c * o._v[0] - s * o._v[1], _v[1] + s * o._v[0] + c * o._v[1], wrap(_v[2] + o._v[2])});
        }
        double t[3], q[4];
        rot(&_v[3], &o._v[0], t);
        const double *a = &_v[3], *b = &o._v[3];
        q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
        q[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
        q[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
        q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
        return IsometryXd(std::vector<double>{_v[0] + t[0], _v[1] + t[1], _v[2] + t[2], q[0], q[1], q[2], q[3]});
    }
    IsometryXd &operator*=(const IsometryXd &o) { return (*this) = (*this) * o; }

private:
    static double wrap(double th) {
        const double PI = 3.14159265358979323846;
        if (th >= -PI && th < PI) return th;
        double m = std::fmod(th, 2 * PI);
        if (m >= PI) m -= 2 * PI;
        if (m < -PI) m += 2 * PI;
        return m;
    }
    static void rot(const double *q, const double *v, double *out) {  // out = R(q) v
        double x = q[0], y = q[1], z = q[2], w = q[3];
        double tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]), tz = 2 * (x * v[1] - y * v[0]);
        out[0] = v[0] + w * tx + (y * tz - z * ty);
        out[1] = v[1] + w * ty + (z * tx - x * tz);
        out[2] = v[2] + w * tz + (x * ty - y * tx);
    }
    bool _2d;
    std::vector<double> _v;
};

// ---- src/graph_wrapper.h:17-78 ---------------------------------------------------------------
class GraphWrapper {
public:
    struct Vertex { int id; IsometryXd estimate; };
    struct Edge {
        int kind;                    // SPG_EDGE_BINARY | SPG_EDGE_GLC
        std::vector<int> vertices;   // ids
        std::vector<double> data;    // record as in spg_batch.edge_data
    };
    virtual ~GraphWrapper() {}
    virtual void addVertex(int id, const IsometryXd &init) = 0;
    virtual void addEdge(int from, int to, const IsometryXd &meas, const std::vector<double> &info /*d*d row-major*/) = 0;
    virtual void marginalize(const std::vector<int> &which, const SparsityOptions &options) = 0;
    virtual void write(const char *fname) = 0;
    virtual std::vector<Vertex> vertices() = 0;
    virtual std::vector<Edge> edges() = 0;
    virtual void printStats(std::ostream &s) const = 0;
    virtual void setEstimate(int vertexid, const IsometryXd &est) = 0;
};

// ---- GraphWrapperG2O(verbose, useGLC) (src/graph_wrapper_g2o.cpp:102) on the MI355X ----------
class GraphWrapperHIP : public GraphWrapper {
public:
    explicit GraphWrapperHIP(int pose_dim, bool useGLC = false, int device = 0) : _glc(useGLC) {
        check(spg_ctx_create(&_ctx, device), "spg_ctx_create (no gfx950 device? there is no CPU fallback)");
        check(spg_graph_create(_ctx, pose_dim, &_g), "spg_graph_create");
    }
    // GraphWrapperG2O(fname, optimize=false, useGLC) (src/graph_wrapper_g2o.cpp:107-154)
    GraphWrapperHIP(const char *fname, bool useGLC = false, int device = 0) : _glc(useGLC) {
        check(spg_ctx_create(&_ctx, device), "spg_ctx_create (no gfx950 device? there is no CPU fallback)");
        check(spg_graph_load_g2o(_ctx, fname, &_g), "spg_graph_load_g2o");
    }
    GraphWrapperHIP(const GraphWrapperHIP &) = delete;
    GraphWrapperHIP &operator=(const GraphWrapperHIP &) = delete;
    ~GraphWrapperHIP() override {
        if (_g) spg_graph_destroy(_g);
        if (_ctx) spg_ctx_destroy(_ctx);
    }

    void addVertex(int id, const IsometryXd &init) override { check(spg_graph_add_vertex(_g, id, init.vector().data()), "addVertex"); }
    void addEdge(int from, int to, const IsometryXd &meas, const std::vector<double> &info) override {
        int d = spg_graph_pose_dim(_g);
        if ((int)info.size() != d * d) throw std::runtime_error("addEdge: information must be d*d row-major");
        std::vector<double> up;
        for (int i = 0; i < d; i++) for (int j = i; j < d; j++) up.push_back(info[(size_t)i * d + j]);
        check(spg_graph_add_edge(_g, from, to, meas.vector().data(), up.data()), "addEdge");
    }
    // GraphWrapperG2O::marginalizeNoOptimize (src/graph_wrapper_g2o.cpp:398-453)
    void marginalizeNoOptimize(const std::vector<int> &which, const SparsityOptions &o) {
        spg_options so;
        so.pose_dim = spg_graph_pose_dim(_g);
        so.algorithm = _glc ? SPG_ALG_GLC : SPG_ALG_NFR;
        so.topology = (int)o.topology;
        so.lin_point = (int)o.linPoint;
        so.include_intra_clique = o.includeIntraClique ? 1 : 0;
        so.flags = 0;
        so.chord_ratio = o.chordRatio;
        std::vector<int32_t> w(which.begin(), which.end());
        check(spg_graph_marginalize(_g, w.data(), (int)w.size(), &so, &_stats), "marginalize");
    }
    // GraphWrapperG2O::marginalize (src/graph_wrapper_g2o.cpp:455-463): marginalizeNoOptimize + optimize().
    // The dense optimiser takes graphs of up to 32k scalar variables: beyond that call marginalizeNoOptimize.
    void marginalize(const std::vector<int> &which, const SparsityOptions &o) override {
        marginalizeNoOptimize(which, o);
        optimize();
    }
    void write(const char *fname) override { check(spg_graph_write_g2o(_g, fname), "write"); }
    void setEstimate(int vertexid, const IsometryXd &est) override { check(spg_graph_set_estimate(_g, vertexid, est.vector().data()), "setEstimate"); }

    std::vector<Vertex> vertices() override {
        int n = spg_graph_num_vertices(_g), ps = spg_graph_pose_dim(_g) == 3 ? 3 : 7;
        std::vector<int32_t> ids(n);
        std::vector<double> p((size_t)n * ps);
        check(spg_graph_get_vertices(_g, ids.data(), p.data()), "vertices");
        std::vector<Vertex> out;
        for (int i = 0; i < n; i++) out.push_back({ids[i], IsometryXd(std::vector<double>(p.begin() + (size_t)i * ps, p.begin() + (size_t)(i + 1) * ps))});
        return out;
    }
    std::vector<Edge> edges() override {
        int ne = spg_graph_num_edges(_g);
        std::vector<int32_t> kind(ne), voff(ne + 1), vids((size_t)spg_graph_edge_vert_size(_g) + 1);
        std::vector<int64_t> doff(ne + 1);
        std::vector<double> data((size_t)spg_graph_edge_data_size(_g) + 1);
        check(spg_graph_get_edges(_g, kind.data(), voff.data(), vids.data(), doff.data(), data.data()), "edges");
        std::vector<Edge> out(ne);
        for (int e = 0; e < ne; e++) {
            out[e].kind = kind[e];
            out[e].vertices.assign(vids.begin() + voff[e], vids.begin() + voff[e + 1]);
            out[e].data.assign(data.begin() + doff[e], data.begin() + doff[e + 1]);
        }
        return out;
    }
    // GraphWrapperG2O::optimize (src/graph_wrapper_g2o.cpp:250-269): first vertex fixed, g2o LM x 50, dense on
    // the device (graphs of up to 32k scalar variables; larger ones need the sparse solver of SURVEY.md 8f.1)
    spg_optimize_stats optimize(int iterations = 50) {
        spg_optimize_stats st;
        check(spg_graph_optimize(_g, iterations, -1, &st), "optimize");
        return st;
    }
    // GraphWrapperG2O::information (src/graph_wrapper_g2o.cpp:351-358): n x n row-major, first vertex fixed
    std::vector<double> information() {
        int64_t n = spg_graph_information(_g, -1, nullptr, 0);
        check((int)std::min<int64_t>(n, 0), "information");
        std::vector<double> H((size_t)n * n);
        check((int)std::min<int64_t>(spg_graph_information(_g, -1, H.data(), (int64_t)H.size()), 0), "information");
        return H;
    }
    // GraphWrapperG2O::kullbackLeibler(other), called on the baseline (src/graph_wrapper_g2o.cpp:531-548)
    double kullbackLeibler(GraphWrapperHIP *other, spg_kld_terms *terms = nullptr) {
        spg_kld_terms t;
        check(spg_graph_kullback_leibler(_g, other->_g, -1, &t), "kullbackLeibler");
        if (terms) *terms = t;
        return t.kld;
    }
    // "nodes = ..; edges = .." of src/graph_wrapper_g2o.cpp:606-612 (fill-in needs the LM Hessian: omitted)
    void printStats(std::ostream &s) const override;
    // sum over blankets of LogdetFunction::value (src/logdet_function.cpp:119-133) of the last marginalize()
    double lastKullbackLeiblerSum() const { return _stats.kld_sum; }
    const spg_marg_stats &lastStats() const { return _stats; }
    spg_graph *handle() { return _g; }

private:
    void check(int rc, const char *what) const {
        if (rc < 0) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + (_ctx ? spg_last_error(_ctx) : ""));
    }
    spg_ctx *_ctx = nullptr;
    spg_graph *_g = nullptr;
    bool _glc;
    spg_marg_stats _stats{};
};

}  // namespace spg

#include <ostream>
inline void spg::GraphWrapperHIP::printStats(std::ostream &s) const {
    s << "nodes = " << spg_graph_num_vertices(_g) - 1 << "; edges = " << spg_graph_num_edges(_g);
}
