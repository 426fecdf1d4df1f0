// include/spg_graph_wrapper.hpp — C++ host façade over the C ABI (include/spg.h), header-only.
//
// Mirrors the reference's operator interface for the accelerated path so that its call sites
// (src/evaluate.cpp:32-221, src/compute_substitute_edge.cpp:13-96, src/test_marginalize_se3.cpp:20-48,
// src/test_marginalize_within_window.cpp:117-182, src/kld_compare.cpp) keep their shape:
//
//   reference                                          here
//   ------------------------------------------------   -----------------------------------------
//   class GraphWrapper        (src/graph_wrapper.h:17-78)   spg::GraphWrapper — the SAME virtual set:
//                                                        addVertex addEdge optimize clonePortion estimate
//                                                        information covariance marginalize write(stream)
//                                                        write(fname) kullbackLeibler chi2(other) chi2()
//                                                        vertices vertex debugPrint printStats setEstimate,
//                                                        nested Vertex {id estimate edges is2d is3d} and
//                                                        Edge {vertices measurement information}
//   class GraphWrapperG2O     (src/graph_wrapper_g2o.h:29)  spg::GraphWrapperHIP (+ marginalizeNoOptimize,
//                                                        optimizeFromId, the file constructor)
//   struct SparsityOptions    (src/sparsity_options.h:11)   spg::SparsityOptions (same fields/defaults)
//   struct DecimateOptions, globalDecimate, onlineDecimate, clusterDecimate, DecimateFunction
//                             (src/decimation.h:13-22)      identical names and signatures
//   class IsometryXd          (src/isometryxd.h:14)         spg::IsometryXd: compose / inverse / vector(mode)
//                                                        with the four SE3 modes of src/isometryxd.cpp:105-128
//   Eigen::MatrixXd / VectorXd                               spg::MatrixXd (row-major, rows() cols() (i,j) data())
//                                                        / spg::VectorXd — Eigen is not a dependency
//
// Errors: the reference asserts/aborts; this façade throws std::runtime_error with the library's
// message. Thread model: one GraphWrapperHIP per host thread (as the reference's one VertexRemover
// per call, src/vertex_remover.h:92-98). Vertex / Edge views returned by vertices() / vertex(id) are owned
// by the wrapper and stay valid until its next mutating call (the reference's views die with the g2o
// object they wrap, src/vertex_remover.cpp:518).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <memory>
#include <ostream>
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

// ---- stand-ins for Eigen::VectorXd / Eigen::MatrixXd in the signatures ---------------------------
typedef std::vector<double> VectorXd;
class MatrixXd {
public:
    MatrixXd() : _r(0), _c(0) {}
    MatrixXd(int r, int c, double fill = 0.0) : _r(r), _c(c), _a((size_t)r * c, fill) {}
    static MatrixXd Identity(int n) { MatrixXd m(n, n); for (int i = 0; i < n; i++) m(i, i) = 1.0; return m; }
    int rows() const { return _r; }
    int cols() const { return _c; }
    double &operator()(int i, int j) { return _a[(size_t)i * _c + j]; }
    double operator()(int i, int j) const { return _a[(size_t)i * _c + j]; }
    double *data() { return _a.data(); }
    const double *data() const { return _a.data(); }
    const std::vector<double> &storage() const { return _a; }   // row-major
private:
    int _r, _c;
    std::vector<double> _a;
};

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

// ---- src/isometryxd.h:14-47 ----------------------------------------------------------------------
class IsometryXd {
public:
    // SE3 vector forms of src/isometryxd.h:17-22 / src/isometryxd.cpp:105-128. The two Euler forms are g2o's
    // internal::toVectorET (translation + roll, pitch, yaw) and the reference's toVectorISAM (the same with
    // components 3 and 5 swapped, src/isometryxd.cpp:131-136); g2o is an un-vendored dependency, its toEuler is
    // restated from the published definition.
    enum SE3Mode { EulerAnglesISAM, EulerAnglesG2O, CondensedQuaternion, Quaternion };

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
    // storage form: SE2 (x y theta), SE3 (t, unit quaternion) = vector(Quaternion)
    const std::vector<double> &vector() const { return _v; }
    VectorXd vector(SE3Mode mode) const {
        if (_2d || mode == Quaternion) return _v;
        double qx = _v[3], qy = _v[4], qz = _v[5], qw = _v[6];
        if (mode == CondensedQuaternion) {   // g2o toVectorMQT: t + (qx qy qz) of the quaternion with w >= 0
            double s = qw < 0 ? -1.0 : 1.0;
            return VectorXd{_v[0], _v[1], _v[2], s * qx, s * qy, s * qz};
        }
        double roll = std::atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy));
        double sp = 2 * (qw * qy - qz * qx);
        double pitch = std::asin(sp > 1 ? 1 : (sp < -1 ? -1 : sp));
        double yaw = std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
        if (mode == EulerAnglesG2O) return VectorXd{_v[0], _v[1], _v[2], roll, pitch, yaw};
        return VectorXd{_v[0], _v[1], _v[2], yaw, pitch, roll};   // EulerAnglesISAM
    }

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

// ---- src/graph_wrapper.h:17-78: the full virtual set ------------------------------------------------
class GraphWrapper {
public: /* Types */
    class Edge;
    class Vertex {
    public:
        virtual ~Vertex() {}
        virtual int id() const = 0;
        virtual IsometryXd estimate() const = 0;
        virtual std::vector<const Edge *> edges() const = 0;
        virtual bool is2d() const = 0;
        virtual bool is3d() const { return !is2d(); }
    };
    class Edge {
    public:
        virtual ~Edge() {}
        virtual std::vector<const Vertex *> vertices() const = 0;
        virtual IsometryXd measurement() const = 0;
        virtual MatrixXd information() const = 0;
    };

public: /* Methods */
    virtual ~GraphWrapper() {}
    virtual void addVertex(int id, const IsometryXd &init) = 0;
    virtual void addEdge(int from, int to, const IsometryXd &meas, const MatrixXd &info) = 0;
    virtual void optimize() = 0;
    virtual GraphWrapper *clonePortion(int maxid) = 0;
    virtual VectorXd estimate() = 0;
    virtual MatrixXd information() = 0;
    virtual MatrixXd covariance() = 0;
    virtual void marginalize(const std::vector<int> &which, const SparsityOptions &options) = 0;
    virtual void write(std::ostream &s) = 0;
    virtual void write(const char *fname) { std::ofstream f(fname); write(f); f.close(); }
    virtual double kullbackLeibler(GraphWrapper *other) = 0;
    virtual double chi2(GraphWrapper *other) = 0;
    virtual double chi2() const = 0;
    virtual const std::vector<Vertex *> &vertices() const = 0;
    virtual Vertex *vertex(int id) = 0;
    virtual void debugPrint(std::ostream &s) const = 0;
    virtual void printStats(std::ostream &s) const = 0;
    virtual void setEstimate(int vertexid, const IsometryXd &est) = 0;
};
inline std::ostream &operator<<(std::ostream &s, const GraphWrapper &gw) { gw.debugPrint(s); return s; }

// ---- GraphWrapperG2O (src/graph_wrapper_g2o.h:29-156) on the MI355X ---------------------------------
class GraphWrapperHIP : public GraphWrapper {
    struct Ctx {
        spg_ctx *h = nullptr;
        ~Ctx() { if (h) spg_ctx_destroy(h); }
    };
    class EdgeView;
    class VertexView : public GraphWrapper::Vertex {
    public:
        int id() const override { return _id; }
        IsometryXd estimate() const override { return _est; }
        std::vector<const GraphWrapper::Edge *> edges() const override { return _edges; }
        bool is2d() const override { return _est.is2d(); }
        int _id = 0;
        IsometryXd _est;
        std::vector<const GraphWrapper::Edge *> _edges;
    };
    class EdgeView : public GraphWrapper::Edge {
    public:
        std::vector<const GraphWrapper::Vertex *> vertices() const override { return _verts; }
        // pose-pose edge: its measurement; n-ary GLC edge: the reparametrised measurement has no IsometryXd form
        IsometryXd measurement() const override {
            if (_kind != SPG_EDGE_BINARY) throw std::runtime_error("measurement(): n-ary edge (GLC / MultiEdgeCorrelated: use record())");
            return IsometryXd(std::vector<double>(_rec.begin(), _rec.begin() + (_d == 3 ? 3 : 7)));
        }
        // pose-pose edge: d x d; GLC edge: I_r (src/topology_provider_glc.cpp:91)
        MatrixXd information() const override {
            if (_kind == SPG_EDGE_MULTI) {   // MultiEdgeCorrelated: the joint information W^T W of its nm measurements (include/spg.h)
                const int nm = (int)_rec[0], r = _d * nm;
                const double *W = _rec.data() + 1 + 2 * nm + nm * (_d == 3 ? 3 : 7);
                MatrixXd m(r, r);
                for (int i = 0; i < r; i++) for (int j = 0; j < r; j++) { double v = 0; for (int t = 0; t < r; t++) v += W[t * r + i] * W[t * r + j]; m(i, j) = v; }
                return m;
            }
            if (_kind != SPG_EDGE_BINARY) { int n = _d * (int)_verts.size(); return MatrixXd::Identity(((int)_rec.size() - n) / n); }
            MatrixXd m(_d, _d);
            int ps = _d == 3 ? 3 : 7, p = 0;
            for (int i = 0; i < _d; i++) for (int j = i; j < _d; j++) { m(i, j) = m(j, i) = _rec[ps + p]; p++; }
            return m;
        }
        int kind() const { return _kind; }
        const std::vector<double> &record() const { return _rec; }   // as in spg_batch.edge_data
        int _kind = 0, _d = 3;
        std::vector<double> _rec;
        std::vector<const GraphWrapper::Vertex *> _verts;
    };

public:
    // GraphWrapperG2O(verbose, useGLC) (src/graph_wrapper_g2o.cpp:102)
    explicit GraphWrapperHIP(int pose_dim, bool useGLC = false, int device = 0) : _ctx(std::make_shared<Ctx>()), _glc(useGLC) {
        check(spg_ctx_create(&_ctx->h, device), "spg_ctx_create (no gfx950 device? there is no CPU fallback)");
        check(spg_graph_create(_ctx->h, pose_dim, &_g), "spg_graph_create");
    }
    // GraphWrapperG2O(fname, optimize, useGLC) (src/graph_wrapper_g2o.cpp:107-154)
    GraphWrapperHIP(const char *fname, bool optimizeAtLoad = false, bool useGLC = false, int device = 0) : _ctx(std::make_shared<Ctx>()), _glc(useGLC) {
        check(spg_ctx_create(&_ctx->h, device), "spg_ctx_create (no gfx950 device? there is no CPU fallback)");
        check(spg_graph_load_g2o(_ctx->h, fname, &_g), "spg_graph_load_g2o");
        if (optimizeAtLoad) optimize();
    }
    // dense / block-sparse factorisation behind optimize() and kullbackLeibler() of this wrapper's context (SPG_SOLVER_AUTO by
    // default: by size) — CHOLMOD's / SimplicialLLT's role in src/graph_wrapper_g2o.cpp:250-269,531-548
    // arena capacity up front + the host-side buffers of a marginalisation sized and touched (include/spg.h: spg_graph_reserve)
    void reserve(long long arena_doubles) { check(spg_graph_reserve(_g, arena_doubles), "spg_graph_reserve"); }
    void setLinearSolver(int solver) { check(spg_ctx_set_linear_solver(_ctx->h, solver), "spg_ctx_set_linear_solver"); }
    GraphWrapperHIP(const GraphWrapperHIP &) = delete;
    GraphWrapperHIP &operator=(const GraphWrapperHIP &) = delete;
    ~GraphWrapperHIP() override {
        drop_views();
        if (_g) spg_graph_destroy(_g);
    }

    void addVertex(int id, const IsometryXd &init) override { drop_views(); check(spg_graph_add_vertex(_g, id, init.vector().data()), "addVertex"); }
    void addEdge(int from, int to, const IsometryXd &meas, const MatrixXd &info) override {
        int d = spg_graph_pose_dim(_g);
        if (info.rows() != d || info.cols() != d) throw std::runtime_error("addEdge: information must be d x d");
        std::vector<double> up;
        for (int i = 0; i < d; i++) for (int j = i; j < d; j++) up.push_back(info(i, j));
        drop_views();
        check(spg_graph_add_edge(_g, from, to, meas.vector().data(), up.data()), "addEdge");
    }
    // convenience: information as d*d row-major numbers
    void addEdge(int from, int to, const IsometryXd &meas, const std::vector<double> &info) {
        int d = spg_graph_pose_dim(_g);
        if ((int)info.size() != d * d) throw std::runtime_error("addEdge: information must be d*d row-major");
        MatrixXd m(d, d);
        std::copy(info.begin(), info.end(), m.data());
        addEdge(from, to, meas, m);
    }
    // GraphWrapperG2O::optimize (src/graph_wrapper_g2o.cpp:250-269): the smallest-id vertex fixed, g2o LM x 50
    void optimize() override { optimizeFromId(-1); }
    // GraphWrapperG2O::optimizeFromId (src/graph_wrapper_g2o.cpp:271-289): the given vertex fixed
    spg_optimize_stats optimizeFromId(int first_vertex_id, int iterations = 50) {
        spg_optimize_stats st;
        drop_views();
        check(spg_graph_optimize(_g, iterations, first_vertex_id, &st), "optimize");
        _last_opt = st;
        return st;
    }
    const spg_optimize_stats &lastOptimize() const { return _last_opt; }
    // GraphWrapperG2O::clonePortion (src/graph_wrapper_g2o.cpp:334-356): vertices / edges up to maxid, optimised
    GraphWrapperHIP *clonePortion(int maxid) override {
        spg_graph *c = nullptr;
        check(spg_graph_clone_portion(_g, maxid, &c), "clonePortion");
        GraphWrapperHIP *gw = new GraphWrapperHIP(_ctx, c, _glc);
        if (spg_graph_num_vertices(c) > 1 && spg_graph_num_edges(c) > 0) gw->optimize();
        return gw;
    }
    // GraphWrapperG2O::estimate (src/graph_wrapper_g2o.cpp:358-361) = stack() (src/utils.cpp:200-221): every vertex
    // but the first, id order, SE2 (x y theta) / SE3 toVectorISAM
    VectorXd estimate() override {
        VectorXd out;
        const std::vector<Vertex *> &vs = vertices();
        for (size_t i = 1; i < vs.size(); i++) {
            VectorXd v = vs[i]->estimate().vector(IsometryXd::EulerAnglesISAM);
            out.insert(out.end(), v.begin(), v.end());
        }
        return out;
    }
    // GraphWrapperG2O::information (src/graph_wrapper_g2o.cpp:363-366): n x n, first vertex fixed
    MatrixXd information() override {
        int64_t n = spg_graph_information(_g, -1, nullptr, 0);
        check((int)std::min<int64_t>(n, 0), "information");
        MatrixXd H((int)n, (int)n);
        check((int)std::min<int64_t>(spg_graph_information(_g, -1, H.data(), n * n), 0), "information");
        return H;
    }
    // GraphWrapperG2O::covariance (src/graph_wrapper_g2o.cpp:368-373)
    MatrixXd covariance() override {
        int64_t n = spg_graph_covariance(_g, -1, nullptr, 0);
        check((int)std::min<int64_t>(n, 0), "covariance");
        MatrixXd S((int)n, (int)n);
        check((int)std::min<int64_t>(spg_graph_covariance(_g, -1, S.data(), n * n), 0), "covariance");
        return S;
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
        drop_views();
        check(spg_graph_marginalize(_g, w.data(), (int)w.size(), &so, &_stats), "marginalize");
    }
    // GraphWrapperG2O::marginalize (src/graph_wrapper_g2o.cpp:455-463): marginalizeNoOptimize + optimize()
    void marginalize(const std::vector<int> &which, const SparsityOptions &o) override {
        marginalizeNoOptimize(which, o);
        optimize();
    }
    // GraphWrapper::write (src/graph_wrapper.h:62-64, src/graph_wrapper_g2o.cpp:467-470)
    void write(std::ostream &s) override {
        char *text = nullptr;
        size_t len = 0;
        check(spg_graph_write_g2o_mem(_g, &text, &len), "write");
        s.write(text, (std::streamsize)len);
        spg_free(text);
    }
    void write(const char *fname) override { check(spg_graph_write_g2o(_g, fname), "write"); }
    // GraphWrapperG2O::kullbackLeibler(other), called on the baseline (src/graph_wrapper_g2o.cpp:531-548)
    double kullbackLeibler(GraphWrapper *other) override { return kullbackLeibler(other, nullptr); }
    double kullbackLeibler(GraphWrapper *other, spg_kld_terms *terms) {
        GraphWrapperHIP *o = dynamic_cast<GraphWrapperHIP *>(other);
        if (!o) throw std::runtime_error("kullbackLeibler: the other graph is not a GraphWrapperHIP");
        spg_kld_terms t;
        check(spg_graph_kullback_leibler(_g, o->_g, -1, &t), "kullbackLeibler");
        if (terms) *terms = t;
        return t.kld;
    }
    // GraphWrapperG2O::chi2(other) (src/graph_wrapper_g2o.cpp:503-529): push; other's vertices take other's estimates
    // and are held fixed; optimise the rest; read chi2; pop
    double chi2(GraphWrapper *other) override {
        std::vector<int32_t> ids;
        std::vector<double> saved;
        snapshot(ids, saved);
        std::vector<int32_t> fx;
        if (!ids.empty()) fx.push_back(ids[0]);   // optimize() inside keeps the first vertex fixed (:250-252)
        for (const Vertex *v : other->vertices()) {
            check(spg_graph_set_estimate(_g, v->id(), v->estimate().vector().data()), "chi2(other): the other graph holds a vertex this one lacks");
            if (ids.empty() || v->id() != ids[0]) fx.push_back(v->id());
        }
        spg_optimize_stats st;
        int rc = spg_graph_optimize_fixed(_g, 50, fx.data(), (int)fx.size(), &st);
        restore(ids, saved);
        check(rc, "chi2(other)");
        return st.chi2_final;
    }
    // _so->chi2() (src/graph_wrapper_g2o.cpp:501)
    double chi2() const override {
        double v = 0;
        check(spg_graph_chi2(_g, &v), "chi2");
        return v;
    }
    const std::vector<Vertex *> &vertices() const override { build_views(); return _vviews; }
    // GraphWrapperG2O::vertex (src/graph_wrapper_g2o.cpp:207-212): lower_bound over the id-sorted views
    Vertex *vertex(int id) override {
        build_views();
        auto it = std::lower_bound(_vviews.begin(), _vviews.end(), id, [](const Vertex *v, int i) { return v->id() < i; });
        return (it != _vviews.end() && (*it)->id() == id) ? *it : nullptr;
    }
    // GraphWrapperG2O::debugPrint (src/graph_wrapper_g2o.cpp:577-594)
    void debugPrint(std::ostream &s) const override {
        build_views();
        s << "+ vertices: ";
        for (const Vertex *v : _vviews) s << v->id() << " ";
        s << "\n+ edges: ";
        for (const EdgeView *e : _eviews) {
            s << "(";
            for (size_t i = 0; i < e->_verts.size(); i++) s << e->_verts[i]->id() << (i + 1 < e->_verts.size() ? "," : "");
            s << ") ";
        }
        s << std::endl;
    }
    // GraphWrapperG2O::printStats (src/graph_wrapper_g2o.cpp:606-612): nodes, edges and the fill-in of the
    // information matrix (block pattern: a vertex pair is filled when some edge joins it)
    void printStats(std::ostream &s) const override {
        build_views();
        const int d = spg_graph_pose_dim(_g);
        const double nfree = (double)_vviews.size() - 1;
        std::vector<std::pair<int, int>> blocks;
        const int first = _vviews.empty() ? -1 : _vviews[0]->id();
        for (const Vertex *v : _vviews) if (v->id() != first) blocks.push_back({v->id(), v->id()});
        for (const EdgeView *e : _eviews)
            for (const Vertex *a : e->_verts) for (const Vertex *b : e->_verts)
                if (a->id() != first && b->id() != first && a->id() != b->id()) blocks.push_back({a->id(), b->id()});
        std::sort(blocks.begin(), blocks.end());
        blocks.erase(std::unique(blocks.begin(), blocks.end()), blocks.end());
        double fillin = nfree > 0 ? (double)blocks.size() * d * d / ((nfree * d) * (nfree * d)) : 0.0;
        s << "nodes = " << spg_graph_num_vertices(_g) - 1 << "; edges = " << spg_graph_num_edges(_g) << "; fillin = " << fillin * 100 << "%";
    }
    void setEstimate(int vertexid, const IsometryXd &est) override {
        drop_views();
        check(spg_graph_set_estimate(_g, vertexid, est.vector().data()), "setEstimate");
    }

    // ---- beyond the reference interface -----------------------------------------------------------------
    struct EdgeRecord {
        int kind;                    // SPG_EDGE_BINARY | SPG_EDGE_GLC | SPG_EDGE_MULTI
        std::vector<int> vertices;   // ids
        std::vector<double> data;    // record as in spg_batch.edge_data
    };
    std::vector<EdgeRecord> edgeRecords() {
        int ne = spg_graph_num_edges(_g);
        std::vector<int32_t> kind(ne + 1), voff(ne + 1), vids((size_t)spg_graph_edge_vert_size(_g) + 1);
        std::vector<int64_t> doff(ne + 1);
        std::vector<double> data((size_t)spg_graph_edge_data_size(_g) + 1);
        check(spg_graph_get_edges(_g, kind.data(), voff.data(), vids.data(), doff.data(), data.data()), "edges");
        std::vector<EdgeRecord> out(ne);
        for (int e = 0; e < ne; e++) {
            out[e].kind = kind[e];
            out[e].vertices.assign(vids.begin() + voff[e], vids.begin() + voff[e + 1]);
            out[e].data.assign(data.begin() + doff[e], data.begin() + doff[e + 1]);
        }
        return out;
    }
    // sum over blankets of LogdetFunction::value (src/logdet_function.cpp:119-133) of the last marginalize()
    double lastKullbackLeiblerSum() const { return _stats.kld_sum; }
    const spg_marg_stats &lastStats() const { return _stats; }
    spg_graph *handle() { return _g; }
    spg_ctx *context() { return _ctx->h; }

private:
    GraphWrapperHIP(std::shared_ptr<Ctx> ctx, spg_graph *g, bool glc) : _ctx(std::move(ctx)), _g(g), _glc(glc) {}
    void check(int rc, const char *what) const {
        if (rc < 0) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + ((_ctx && _ctx->h) ? spg_last_error(_ctx->h) : ""));
    }
    void snapshot(std::vector<int32_t> &ids, std::vector<double> &poses) const {
        int n = spg_graph_num_vertices(_g), ps = spg_graph_pose_dim(_g) == 3 ? 3 : 7;
        ids.resize(n);
        poses.resize((size_t)n * ps);
        check(spg_graph_get_vertices(_g, ids.data(), poses.data()), "vertices");
    }
    void restore(const std::vector<int32_t> &ids, const std::vector<double> &poses) {
        int ps = spg_graph_pose_dim(_g) == 3 ? 3 : 7;
        drop_views();
        for (size_t i = 0; i < ids.size(); i++) check(spg_graph_set_estimate(_g, ids[i], poses.data() + i * ps), "setEstimate");
    }
    void drop_views() const {
        for (Vertex *v : _vviews) delete v;
        for (EdgeView *e : _eviews) delete e;
        _vviews.clear();
        _eviews.clear();
        _views_ok = false;
    }
    void build_views() const {
        if (_views_ok) return;
        drop_views();
        const int d = spg_graph_pose_dim(_g), ps = d == 3 ? 3 : 7;
        std::vector<int32_t> ids;
        std::vector<double> poses;
        snapshot(ids, poses);
        for (size_t i = 0; i < ids.size(); i++) {
            VertexView *v = new VertexView;
            v->_id = ids[i];
            v->_est = IsometryXd(std::vector<double>(poses.begin() + i * ps, poses.begin() + (i + 1) * ps));
            _vviews.push_back(v);
        }
        auto find = [&](int id) -> VertexView * {
            auto it = std::lower_bound(_vviews.begin(), _vviews.end(), id, [](const Vertex *v, int i) { return v->id() < i; });
            return static_cast<VertexView *>(*it);
        };
        std::vector<EdgeRecord> recs = const_cast<GraphWrapperHIP *>(this)->edgeRecords();
        for (EdgeRecord &r : recs) {
            EdgeView *e = new EdgeView;
            e->_kind = r.kind; e->_d = d; e->_rec.swap(r.data);
            for (int id : r.vertices) e->_verts.push_back(find(id));
            _eviews.push_back(e);
            for (size_t i = 0; i < e->_verts.size(); i++) {
                bool dup = false;
                for (size_t j = 0; j < i; j++) dup |= (e->_verts[j] == e->_verts[i]);
                if (!dup) const_cast<VertexView *>(static_cast<const VertexView *>(e->_verts[i]))->_edges.push_back(e);
            }
        }
        _views_ok = true;
    }
    std::shared_ptr<Ctx> _ctx;
    spg_graph *_g = nullptr;
    bool _glc;
    spg_marg_stats _stats{};
    spg_optimize_stats _last_opt{};
    mutable std::vector<Vertex *> _vviews;     // id-sorted
    mutable std::vector<EdgeView *> _eviews;   // insertion order of the live edges
    mutable bool _views_ok = false;
};

// ---- computeSubstituteEdge (src/compute_substitute_edge.h:13-16, .cpp:13-96) ---------------------------------
// gw is the FULL source graph; on return the marginalised endpoint of (from, to) is replaced by the nearest
// surviving vertex and edgemeas / edgeinfo hold the composed measurement and sym((sum Omega^-1)^-1).
inline void computeSubstituteEdge(GraphWrapperHIP *gw, const std::vector<int> &marginalized, int maxid, int &from, int &to,
                                  IsometryXd &edgemeas, MatrixXd &edgeinfo) {
    const int d = spg_graph_pose_dim(gw->handle()), ps = d == 3 ? 3 : 7;
    std::vector<int32_t> m(marginalized.begin(), marginalized.end());
    std::vector<double> meas(ps), up((size_t)d * (d + 1) / 2);
    int rc = spg_graph_substitute_edge(gw->handle(), m.data(), (int)m.size(), maxid, &from, &to, meas.data(), up.data());
    if (rc < 0) throw std::runtime_error(std::string("computeSubstituteEdge failed: ") + spg_last_error(gw->context()));
    edgemeas = IsometryXd(meas);
    edgeinfo = MatrixXd(d, d);
    int p = 0;
    for (int i = 0; i < d; i++) for (int j = i; j < d; j++) { edgeinfo(i, j) = edgeinfo(j, i) = up[p]; p++; }
}

}  // namespace spg
