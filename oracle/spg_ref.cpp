// oracle/spg_ref.cpp — C entry points of the CPU oracle (libspg_ref.so).
//
// TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg as the checker / reported baseline; never linked or loaded by the product path.
// PARITY UNPINNED: no reference-owned golden vector exists for this path and the reference cannot be
// built in this container (Eigen, g2o, iSAM, CHOLMOD absent) — see DESIGN.md "Oracle".
//
// Exports
//   spg_marginalize_batch / spg_run_round   same signatures as include/spg.h (ctx ignored)
//   spgref_graph_*                          literal, strictly sequential restatement of
//                                           GraphWrapperG2O::marginalizeNoOptimize -> VertexRemover::remove
//                                           (src/graph_wrapper_g2o.cpp:398-453, src/vertex_remover.cpp:83-251,500-546)
//   spgref_* unit functions                 so tests can pin the building blocks against numpy
#include <chrono>
#include <cstdio>
#include <map>
#include <set>
#include <limits>
#include <thread>
#include "ref_blanket.hpp"

using namespace spgref;

// ------------------------------------------------------------------------------ batch entry
extern "C" int spg_marginalize_batch(spg_ctx *, const spg_options *o, const spg_batch *b, spg_result *r) {
    if (!o || !b || !r) return SPG_EINVAL;
    int d = o->pose_dim, ps = pose_stride(d);
    int32_t ne = 0, nev = 0;
    int64_t ned = 0;
    r->new_edge_off[0] = 0;
    r->new_edge_vert_off[0] = 0;
    r->new_edge_data_off[0] = 0;
    for (int bi = 0; bi < b->B; bi++) {
        BlanketIn in;
        in.d = d;
        in.nv = b->vert_off[bi + 1] - b->vert_off[bi];
        in.m = b->n_remove[bi];
        in.pose = b->pose + (size_t)b->vert_off[bi] * ps;
        for (int e = b->edge_off[bi]; e < b->edge_off[bi + 1]; e++) {
            EdgeIn ei;
            ei.kind = b->edge_kind[e];
            for (int v = b->edge_vert_off[e]; v < b->edge_vert_off[e + 1]; v++) ei.v.push_back(b->edge_vert[v]);
            ei.data = b->edge_data + b->edge_data_off[e];
            ei.len = b->edge_data_off[e + 1] - b->edge_data_off[e];
            in.edges.push_back(ei);
        }
        BlanketOut out = run_blanket(*o, in);
        r->status[bi] = out.status;
        if (r->info) r->info[bi] = out.info;
        r->kld[bi] = out.kld;
        if (r->min_gap) r->min_gap[bi] = out.min_gap;
        if (r->target_info && out.target.r > 0) {
            double *dst = r->target_info + r->target_info_off[bi];
            std::memcpy(dst, out.target.a.data(), out.target.a.size() * sizeof(double));
        }
        for (const NewEdge &e : out.edges) {
            if (ne + 1 > r->new_edge_cap || nev + (int)e.v.size() > r->new_edge_vert_cap ||
                ned + (int64_t)e.data.size() > r->new_edge_data_cap)
                return SPG_ECAPACITY;
            r->new_edge_kind[ne] = e.kind;
            for (int v : e.v) r->new_edge_vert[nev++] = b->vert_id[b->vert_off[bi] + v];
            std::memcpy(r->new_edge_data + ned, e.data.data(), e.data.size() * sizeof(double));
            ned += (int64_t)e.data.size();
            ne++;
            r->new_edge_vert_off[ne] = nev;
            r->new_edge_data_off[ne] = ned;
        }
        r->new_edge_off[bi + 1] = ne;
    }
    return 0;
}

// ------------------------------------------------------------------------------ round entry
static int run_round_range(double *arena, const spg_round_desc *rd, int lo, int hi, int step) {
    const spg_options &o = *rd->opts;
    int d = o.pose_dim, ps = pose_stride(d);
    for (int bi = lo; bi < hi; bi += step) {
        const spg_blanket_desc &bd = rd->blankets[bi];
        std::vector<double> poses((size_t)bd.n_vert * ps);
        for (int v = 0; v < bd.n_vert; v++)
            std::memcpy(&poses[(size_t)v * ps], arena + rd->vert_pose_off[bd.vert_begin + v], ps * sizeof(double));
        BlanketIn in;
        in.d = d; in.nv = bd.n_vert; in.m = bd.n_remove; in.pose = poses.data();
        for (int e = bd.edge_begin; e < bd.edge_begin + bd.n_edge; e++) {
            const spg_edge_ref &er = rd->edges[e];
            EdgeIn ei;
            ei.kind = er.kind;
            for (int v = 0; v < er.nv; v++) ei.v.push_back(rd->edge_vert[er.vbegin + v]);
            ei.data = arena + er.off;
            ei.len = er.len;
            in.edges.push_back(ei);
        }
        BlanketOut out = run_blanket(o, in);
        double *rec = arena + bd.out_off;
        int64_t reclen = SPG_OUT_LEN(bd.n_new_max, bd.n_new_vert_max);
        for (int64_t i = 0; i < reclen; i++) rec[i] = 0;
        rec[0] = out.status; rec[1] = out.info; rec[2] = out.kld; rec[3] = out.min_gap;
        int nn = 0, nvv = 0;
        int64_t off = 0;
        for (const NewEdge &e : out.edges) {
            if (nn >= bd.n_new_max || nvv + (int)e.v.size() > bd.n_new_vert_max || off + (int64_t)e.data.size() > bd.new_len)
                return SPG_ECAPACITY;
            rec[SPG_OUT_HDR + 4 * nn + 0] = e.kind;
            rec[SPG_OUT_HDR + 4 * nn + 1] = (double)off;
            rec[SPG_OUT_HDR + 4 * nn + 2] = (double)e.data.size();
            rec[SPG_OUT_HDR + 4 * nn + 3] = (double)e.v.size();
            for (int v : e.v) rec[SPG_OUT_HDR + 4 * bd.n_new_max + nvv++] = v;
            std::memcpy(arena + bd.new_off + off, e.data.data(), e.data.size() * sizeof(double));
            off += (int64_t)e.data.size();
            nn++;
        }
        rec[4] = nn;
        rec[5] = SPG_READY_WORD(rd->tag);
        if (bd.tinfo_off >= 0 && out.target.r > 0)
            std::memcpy(arena + bd.tinfo_off, out.target.a.data(), out.target.a.size() * sizeof(double));
    }
    return 0;
}


extern "C" int spg_run_round(double *arena, const spg_round_desc *rd) {
    return run_round_range(arena, rd, rd->first, rd->first + rd->count, 1);
}

// The same round with its (independent) blankets spread over host threads: bench.py's second CPU
// baseline. Blankets write disjoint records, so no synchronisation is needed.
static int g_round_threads = 1;
extern "C" void spgref_set_round_threads(int n) { g_round_threads = n < 1 ? 1 : n; }
extern "C" int spg_run_round_mt(double *arena, const spg_round_desc *rd) {
    int nt = std::min(g_round_threads, std::max(1, rd->count / 8));
    if (nt <= 1) return spg_run_round(arena, rd);
    std::vector<std::thread> th;
    std::vector<int> rc(nt, 0);
    for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] { rc[t] = run_round_range(arena, rd, rd->first + t, rd->first + rd->count, nt); });
    for (auto &x : th) x.join();
    for (int t = 0; t < nt; t++) if (rc[t]) return rc[t];
    return 0;
}

// ------------------------------------------------------------------------------ sequential graph
namespace {
struct REdge {
    int kind;
    std::vector<int> ids;
    std::vector<double> data;
    bool alive;
};
struct RBlanketLog { int root, status, info; double kld, min_gap; int k; };
struct RGraph {
    int d;
    std::map<int, std::vector<double>> pose;  // id -> estimate
    std::map<int, std::set<int>> adj;         // id -> live edge indices
    std::vector<REdge> edges;
    std::vector<RBlanketLog> log;
    double seconds = 0;
};

// markovBlanketVertices (src/vertex_remover.cpp:197-215)
std::set<int> blanket_vertices(const RGraph &g, int root) {
    std::set<int> vs;
    vs.insert(root);
    for (int e : g.adj.at(root)) for (int id : g.edges[e].ids) vs.insert(id);
    return vs;
}

// extendedMarkovBlanketVertices, active (#else) branch (src/vertex_remover.cpp:142-195): one ascending
// pass over the growing id-ordered set.
std::set<int> extended_blanket_vertices(const RGraph &g, int root, const std::set<int> &pickBin, std::set<int> &picked) {
    picked.clear();
    std::set<int> ret = blanket_vertices(g, root);
    picked.insert(root);
    for (auto it = ret.begin(); it != ret.end(); ++it) {
        int v = *it;
        if (pickBin.count(v) > 0 && picked.count(v) == 0) {
            picked.insert(v);
            std::set<int> other = blanket_vertices(g, v);
            ret.insert(other.begin(), other.end());
        }
    }
    return ret;
}

// markovBlanketEdges (src/vertex_remover.cpp:225-251)
std::set<int> blanket_edges(const RGraph &g, const std::set<int> &vs, const std::set<int> &hubs, bool intra) {
    std::set<int> es;
    for (int v : vs)
        for (int e : g.adj.at(v)) {
            bool is_markov = true, found_hub = false;
            for (int id : g.edges[e].ids) {
                if (vs.count(id) == 0) { is_markov = false; break; }
                if (hubs.count(id) > 0) found_hub = true;
            }
            if (is_markov && (intra || found_hub)) es.insert(e);
        }
    return es;
}
}  // namespace

extern "C" {

void *spgref_graph_create(int pose_dim) {
    RGraph *g = new RGraph;
    g->d = pose_dim;
    return g;
}
void spgref_graph_destroy(void *h) { delete (RGraph *)h; }

int spgref_graph_add_vertex(void *h, int id, const double *pose) {
    RGraph *g = (RGraph *)h;
    int ps = pose_stride(g->d);
    g->pose[id] = std::vector<double>(pose, pose + ps);
    g->adj[id];
    return 0;
}

int spgref_graph_add_edge(void *h, int kind, int nv, const int32_t *ids, const double *data, int64_t len) {
    RGraph *g = (RGraph *)h;
    REdge e;
    e.kind = kind;
    e.ids.assign(ids, ids + nv);
    e.data.assign(data, data + len);
    e.alive = true;
    for (int i = 0; i < nv; i++) if (!g->pose.count(ids[i])) return SPG_EINVAL;
    int idx = (int)g->edges.size();
    g->edges.push_back(e);
    for (int i = 0; i < nv; i++) g->adj[ids[i]].insert(idx);
    return 0;
}

// VertexRemover::remove (src/vertex_remover.cpp:83-140), strictly sequential.
int spgref_graph_marginalize(void *h, const int32_t *which, int n, const spg_options *o) {
    RGraph *g = (RGraph *)h;
    auto t0 = std::chrono::steady_clock::now();
    int d = g->d, ps = pose_stride(d);
    std::set<int> toRemoveSet(which, which + n), deleted;
    g->log.clear();
    int rc = 0;
    for (int i = 0; i < n; i++) {
        int root = which[i];
        if (deleted.count(root)) continue;
        if (!g->pose.count(root)) return SPG_EINVAL;
        std::set<int> vmarkov, toRemoveNow;
        if (o->topology == SPG_TOPO_DENSE || o->topology == SPG_TOPO_CLIQUEY_DENSE) {
            vmarkov = extended_blanket_vertices(*g, root, toRemoveSet, toRemoveNow);
        } else {
            vmarkov = blanket_vertices(*g, root);
            toRemoveNow.insert(root);
        }
        std::set<int> emarkov = blanket_edges(*g, vmarkov, toRemoveNow, o->include_intra_clique != 0);
        // buildSubgraph (src/vertex_remover.cpp:349-361): removed first, then kept, both ascending id
        std::vector<int> order(toRemoveNow.begin(), toRemoveNow.end());
        for (int v : vmarkov) if (!toRemoveNow.count(v)) order.push_back(v);
        std::map<int, int> local;
        for (size_t k = 0; k < order.size(); k++) local[order[k]] = (int)k;
        std::vector<double> poses(order.size() * ps);
        for (size_t k = 0; k < order.size(); k++) std::memcpy(&poses[k * ps], g->pose[order[k]].data(), ps * sizeof(double));
        BlanketIn in;
        in.d = d; in.nv = (int)order.size(); in.m = (int)toRemoveNow.size(); in.pose = poses.data();
        for (int e : emarkov) {
            EdgeIn ei;
            ei.kind = g->edges[e].kind;
            for (int id : g->edges[e].ids) ei.v.push_back(local[id]);
            ei.data = g->edges[e].data.data();
            ei.len = (int64_t)g->edges[e].data.size();
            in.edges.push_back(ei);
        }
        BlanketOut out = run_blanket(*o, in);
        g->log.push_back({root, out.status, out.info, out.kld, out.min_gap, in.nv - in.m});
        bool fatal = !(out.status == SPG_OK || out.status == SPG_ST_KLD_NOT_PD);
        if (fatal) { rc = SPG_EBLANKET; break; }
        // updateInputGraph (src/vertex_remover.cpp:500-546)
        for (int e : emarkov) {
            g->edges[e].alive = false;
            for (int id : g->edges[e].ids) g->adj[id].erase(e);
        }
        for (int v : toRemoveNow) { g->pose.erase(v); g->adj.erase(v); }
        for (const NewEdge &ne : out.edges) {
            REdge e;
            e.kind = ne.kind;
            for (int v : ne.v) e.ids.push_back(order[v]);
            e.data = ne.data;
            e.alive = true;
            int idx = (int)g->edges.size();
            g->edges.push_back(e);
            for (int id : e.ids) g->adj[id].insert(idx);
        }
        deleted.insert(toRemoveNow.begin(), toRemoveNow.end());
    }
    g->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

double spgref_graph_last_seconds(void *h) { return ((RGraph *)h)->seconds; }
int spgref_graph_num_vertices(void *h) { return (int)((RGraph *)h)->pose.size(); }
int spgref_graph_num_edges(void *h) {
    int c = 0;
    for (auto &e : ((RGraph *)h)->edges) c += e.alive;
    return c;
}
int64_t spgref_graph_edge_data_size(void *h) {
    int64_t c = 0;
    for (auto &e : ((RGraph *)h)->edges) if (e.alive) c += (int64_t)e.data.size();
    return c;
}
int64_t spgref_graph_edge_vert_size(void *h) {
    int64_t c = 0;
    for (auto &e : ((RGraph *)h)->edges) if (e.alive) c += (int64_t)e.ids.size();
    return c;
}
int spgref_graph_get_vertices(void *h, int32_t *ids, double *poses) {
    RGraph *g = (RGraph *)h;
    int ps = pose_stride(g->d), k = 0;
    for (auto &kv : g->pose) {
        ids[k] = kv.first;
        std::memcpy(poses + (size_t)k * ps, kv.second.data(), ps * sizeof(double));
        k++;
    }
    return k;
}
int spgref_graph_get_edges(void *h, int32_t *kind, int32_t *vert_off, int32_t *vert_ids, int64_t *data_off, double *data) {
    RGraph *g = (RGraph *)h;
    int ne = 0, nv = 0;
    int64_t nd = 0;
    vert_off[0] = 0; data_off[0] = 0;
    for (auto &e : g->edges) {
        if (!e.alive) continue;
        kind[ne] = e.kind;
        for (int id : e.ids) vert_ids[nv++] = id;
        std::memcpy(data + nd, e.data.data(), e.data.size() * sizeof(double));
        nd += (int64_t)e.data.size();
        ne++;
        vert_off[ne] = nv; data_off[ne] = nd;
    }
    return ne;
}
int spgref_graph_last_blanket_count(void *h) { return (int)((RGraph *)h)->log.size(); }
int spgref_graph_last_blankets(void *h, int32_t *root, int32_t *status, int32_t *info, double *kld, double *min_gap, int32_t *k) {
    RGraph *g = (RGraph *)h;
    for (size_t i = 0; i < g->log.size(); i++) {
        root[i] = g->log[i].root; status[i] = g->log[i].status; info[i] = g->log[i].info;
        kld[i] = g->log[i].kld; min_gap[i] = g->log[i].min_gap; k[i] = g->log[i].k;
    }
    return (int)g->log.size();
}

// ------------------------------------------------------------------------------ global KLD (a18)
// other->information() == sparseInformation(): Hpp of the optimiser over the non-fixed vertices in id
// order (src/graph_wrapper_g2o.cpp:351-396). No LM in scope: the Gauss-Newton Hessian at the stored
// estimates (g2o's LM restores the undamped diagonal after each solve). ids_out: the variables' vertices.
static Mat graph_information(RGraph *g, int fixed_id, std::vector<int> &ids_out) {
    int d = g->d, ps = pose_stride(d);
    std::vector<int> ids;
    std::map<int, int> loc;
    for (auto &kv : g->pose) { loc[kv.first] = (int)ids.size(); ids.push_back(kv.first); }
    std::vector<double> poses(ids.size() * (size_t)ps);
    for (size_t i = 0; i < ids.size(); i++) std::memcpy(&poses[i * ps], g->pose[ids[i]].data(), ps * sizeof(double));
    BlanketIn in;
    in.d = d; in.nv = (int)ids.size(); in.m = 0; in.pose = poses.data();
    for (auto &e : g->edges) {
        if (!e.alive) continue;
        EdgeIn ei;
        ei.kind = e.kind;
        for (int id : e.ids) ei.v.push_back(loc[id]);
        ei.data = e.data.data();
        ei.len = (int64_t)e.data.size();
        in.edges.push_back(ei);
    }
    Mat H = assemble_hessian(in);
    std::vector<int> keep;
    ids_out.clear();
    for (size_t i = 0; i < ids.size(); i++) {
        if (ids[i] == fixed_id) continue;
        ids_out.push_back(ids[i]);
        for (int a = 0; a < d; a++) keep.push_back((int)i * d + a);
    }
    return select(H, keep, keep);
}

int64_t spgref_graph_information(void *h, int32_t fixed_id, double *out, int64_t cap) {
    RGraph *g = (RGraph *)h;
    std::vector<int> ids;
    Mat H = graph_information(g, fixed_id, ids);
    if (out && (int64_t)H.a.size() <= cap) std::memcpy(out, H.a.data(), H.a.size() * sizeof(double));
    return H.r;
}

// GraphWrapperG2O::kullbackLeibler(other) called on the baseline (src/graph_wrapper_g2o.cpp:531-548):
// marginal of the baseline information onto other's vertices (computeIndices :472-499, first vertex
// skipped), estimateDifference (:550-575), kullbackLeiblerDivergence(..., InformationInformation)
// (src/utils.cpp:70-97). terms: kld, innerprod, mahalanobis, logdetx, logdety, n.
int spgref_graph_kullback_leibler(void *hb, void *ho, int32_t fixed_id, double *terms) {
    RGraph *gb = (RGraph *)hb, *go = (RGraph *)ho;
    if (gb->d != go->d) return SPG_EINVAL;
    int d = gb->d;
    std::vector<int> ids_b, ids_o;
    Mat Hb = graph_information(gb, fixed_id, ids_b);
    Mat infox = graph_information(go, fixed_id, ids_o);
    std::vector<int> keep, marg;
    size_t j = 0;
    for (size_t i = 0; i < ids_b.size(); i++) {
        bool kept = j < ids_o.size() && ids_o[j] == ids_b[i];
        if (kept) j++;
        for (int a = 0; a < d; a++) (kept ? keep : marg).push_back((int)i * d + a);
    }
    if (j != ids_o.size()) return SPG_EINVAL;  // other holds a vertex the baseline lacks
    Mat maty = select(Hb, keep, keep);
    if (!marg.empty()) {
        Mat Hmm = select(Hb, marg, marg), Hmk = select(Hb, marg, keep);
        if (!chol_lower(Hmm)) return SPG_EBLANKET;
        Mat Y = Hmk;
        chol_solve(Hmm, Y);
        Mat C = matmul(transpose(Hmk), Y);
        for (size_t i = 0; i < maty.a.size(); i++) maty.a[i] -= C.a[i];
    }
    int n = infox.r;
    // estimateDifference: baseline estimate vs other's, per kept vertex
    std::vector<double> diff((size_t)n, 0.0);
    for (size_t i = 0; i < ids_o.size(); i++) {
        const double *xb = gb->pose[ids_o[i]].data(), *xo = go->pose[ids_o[i]].data();
        if (d == 3) {
            diff[i * 3] = xb[0] - xo[0]; diff[i * 3 + 1] = xb[1] - xo[1]; diff[i * 3 + 2] = normalize_theta(xb[2] - xo[2]);
        } else {
            iso_to_mqt(iso_mul(iso_inv(iso_from_tq(xb)), iso_from_tq(xo)), &diff[i * 6]);
        }
    }
    bool okx = true, oky = true;
    double logdetx = spd_logdet(infox, okx);
    Mat Ly = maty;
    oky = chol_lower(Ly);
    if (!okx || !oky) return SPG_EBLANKET;
    double logdety = 0;
    for (int i = 0; i < n; i++) logdety -= 2.0 * std::log(Ly(i, i));   // mode InformationInformation: -sum log D
    Mat Z = infox;
    chol_solve(Ly, Z);
    double innerprod = 0;
    for (int i = 0; i < n; i++) innerprod += Z(i, i);
    double mahal = 0;
    for (int i = 0; i < n; i++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += infox(i, k) * diff[k];
        mahal += diff[i] * s;
    }
    terms[0] = 0.5 * (innerprod + mahal - logdetx - logdety - n);
    terms[1] = innerprod; terms[2] = mahal; terms[3] = logdetx; terms[4] = logdety; terms[5] = n;
    return 0;
}

// ------------------------------------------------------------------------------ optimize()
// GraphWrapperG2O::optimize (src/graph_wrapper_g2o.cpp:250-269): vertex 0 fixed, then
// SparseOptimizer::optimize(50) with g2o's OptimizationAlgorithmLevenberg. g2o is an un-vendored,
// un-pinned dependency of the reference; its LM is restated here from the published algorithm
// (g2o/core/optimization_algorithm_levenberg.cpp): lambda_0 = 1e-5 * max|diag H|; per iteration up to 10
// trials of (H + lambda I) x = b, gain ratio rho = (chi2 - chi2') / (x.(lambda x + b) + 1e-3),
// good step: lambda *= clamp(1 - (2 rho - 1)^3, 1/3, 2/3), ni = 2; bad step: lambda *= ni, ni *= 2;
// Terminate when 10 trials failed, rho == 0 or lambda is not finite. Dense linear algebra.
namespace {
struct LinSys { Mat H; std::vector<double> b; double chi2 = 0; std::vector<int> ids; };

// chi2, H = sum J^T Omega J and b = -sum J^T Omega e over the live edges, fixed vertex dropped
LinSys build_system(RGraph *g, const std::set<int> &fixed, bool want_H) {
    int d = g->d, ps = pose_stride(d);
    LinSys S;
    std::map<int, int> loc;
    for (auto &kv : g->pose) if (!fixed.count(kv.first)) { loc[kv.first] = (int)S.ids.size(); S.ids.push_back(kv.first); }
    int n = d * (int)S.ids.size();
    if (want_H) S.H = Mat(n, n);
    S.b.assign(n, 0.0);
    for (auto &e : g->edges) {
        if (!e.alive) continue;
        int q = (int)e.ids.size();
        std::vector<Mat> A(q);          // weighted Jacobian blocks: sqrt-information form, rows = error dim
        std::vector<double> r;          // weighted error
        if (e.kind == SPG_EDGE_BINARY) {
            Mat Ji, Jj;
            double err[6];
            binary_edge_jac(d, g->pose[e.ids[0]].data(), g->pose[e.ids[1]].data(), e.data.data(), Ji, Jj, err);
            Mat O = info_from_upper(e.data.data() + ps, d);
            // contributions with the information matrix itself (no square root needed)
            Mat OJi = matmul(O, Ji), OJj = matmul(O, Jj);
            std::vector<double> Oe(d, 0.0);
            for (int i = 0; i < d; i++) for (int k = 0; k < d; k++) Oe[i] += O(i, k) * err[k];
            for (int i = 0; i < d; i++) S.chi2 += err[i] * Oe[i];
            const Mat *J[2] = {&Ji, &Jj};
            const Mat *OJ[2] = {&OJi, &OJj};
            for (int a = 0; a < 2; a++) {
                auto ia = loc.find(e.ids[a]);
                if (ia == loc.end()) continue;
                for (int rr = 0; rr < d; rr++) {
                    double s = 0;
                    for (int k = 0; k < d; k++) s += (*J[a])(k, rr) * Oe[k];
                    S.b[ia->second * d + rr] -= s;
                }
                if (!want_H) continue;
                for (int c = 0; c < 2; c++) {
                    auto ic = loc.find(e.ids[c]);
                    if (ic == loc.end()) continue;
                    if (a != c && e.ids[0] == e.ids[1]) continue;
                    Mat blk = matmul(transpose(*J[a]), *OJ[c]);
                    for (int rr = 0; rr < d; rr++) for (int cc = 0; cc < d; cc++) S.H(ia->second * d + rr, ic->second * d + cc) += blk(rr, cc);
                }
            }
        } else {
            std::vector<const double *> poses(q);
            for (int i = 0; i < q; i++) poses[i] = g->pose[e.ids[i]].data();
            std::vector<double> wr;
            Mat Aw = nary_weighted_jacobian(e.kind, d, q, poses.data(), e.data.data(), (int64_t)e.data.size(), &wr);   // GLC or MULTI
            const int rdim = Aw.r;
            for (int i = 0; i < rdim; i++) S.chi2 += wr[i] * wr[i];
            for (int a = 0; a < q; a++) {
                auto ia = loc.find(e.ids[a]);
                if (ia == loc.end()) continue;
                for (int rr = 0; rr < d; rr++) {
                    double s = 0;
                    for (int p = 0; p < rdim; p++) s += Aw(p, a * d + rr) * wr[p];
                    S.b[ia->second * d + rr] -= s;
                }
                if (!want_H) continue;
                for (int c = 0; c < q; c++) {
                    auto ic = loc.find(e.ids[c]);
                    if (ic == loc.end()) continue;
                    for (int rr = 0; rr < d; rr++) for (int cc = 0; cc < d; cc++) {
                        double s = 0;
                        for (int p = 0; p < rdim; p++) s += Aw(p, a * d + rr) * Aw(p, c * d + cc);
                        S.H(ia->second * d + rr, ic->second * d + cc) += s;
                    }
                }
            }
        }
    }
    return S;
}

// VertexSE2 / VertexSE3 oplus: additive (x, y, theta wrapped) / right-multiplicative fromVectorMQT
void apply_update(RGraph *g, const std::vector<int> &ids, const std::vector<double> &x) {
    int d = g->d;
    for (size_t i = 0; i < ids.size(); i++) {
        std::vector<double> &p = g->pose[ids[i]];
        if (d == 3) {
            p[0] += x[i * 3]; p[1] += x[i * 3 + 1]; p[2] = normalize_theta(p[2] + x[i * 3 + 2]);
        } else {
            Iso3 X = se3_oplus(iso_from_tq(p.data()), &x[i * 6]);
            iso_to_tq(X, p.data());
            double nq = std::sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
            for (int a = 3; a < 7; a++) p[a] /= nq;
        }
    }
}
}  // namespace

// stats: iterations done, LM trials, chi2 before, chi2 after, final lambda
static int optimize_fixed(RGraph *g, int iterations, const std::set<int> &fixed_id, double *stats) {
    double lambda = 0, ni = 2;
    int it = 0, trials = 0;
    double chi_first = NAN, chi_last = NAN;
    for (; it < iterations; it++) {
        LinSys S = build_system(g, fixed_id, true);
        int n = (int)S.b.size();
        double currentChi = S.chi2;
        if (it == 0) {
            chi_first = currentChi;
            double md = 0;
            for (int i = 0; i < n; i++) md = std::max(md, std::fabs(S.H(i, i)));
            lambda = 1e-5 * md;
            ni = 2;
        }
        chi_last = currentChi;
        double rho = 0;
        int qmax = 0;
        bool lambda_ok = true;
        do {
            std::map<int, std::vector<double>> backup = g->pose;       // push()
            Mat A = S.H;
            for (int i = 0; i < n; i++) A(i, i) += lambda;
            bool ok2 = chol_lower(A);
            Mat X(n, 1);
            for (int i = 0; i < n; i++) X(i, 0) = S.b[i];
            std::vector<double> x(n, 0.0);
            if (ok2) { chol_solve(A, X); for (int i = 0; i < n; i++) x[i] = X(i, 0); }
            apply_update(g, S.ids, x);
            double tempChi = build_system(g, fixed_id, false).chi2;
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            double scale = 1e-3;
            for (int i = 0; i < n; i++) scale += x[i] * (lambda * x[i] + S.b[i]);
            rho = (currentChi - tempChi) / scale;
            trials++;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1.0 - std::pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2.0 / 3.0);
                lambda *= std::max(1.0 / 3.0, alpha);
                ni = 2;
                currentChi = tempChi;
                chi_last = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                g->pose = backup;                                       // pop()
                if (!std::isfinite(lambda)) { lambda_ok = false; break; }
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0 || !lambda_ok) { it++; break; }      // Terminate
    }
    if (stats) { stats[0] = it; stats[1] = trials; stats[2] = chi_first; stats[3] = chi_last; stats[4] = lambda; }
    return 0;
}

int spgref_graph_optimize(void *h, int iterations, int32_t fixed_id, double *stats) {
    return optimize_fixed((RGraph *)h, iterations, std::set<int>{fixed_id}, stats);
}
// optimize() with several vertices held fixed: the inner step of GraphWrapperG2O::chi2(other)
// (src/graph_wrapper_g2o.cpp:503-529)
int spgref_graph_optimize_fixed(void *h, int iterations, const int32_t *fixed_ids, int n_fixed, double *stats) {
    return optimize_fixed((RGraph *)h, iterations, std::set<int>(fixed_ids, fixed_ids + n_fixed), stats);
}

double spgref_graph_chi2(void *h, int32_t fixed_id) { return build_system((RGraph *)h, std::set<int>{fixed_id}, false).chi2; }

int spgref_graph_set_estimate(void *h, int id, const double *pose) {
    RGraph *g = (RGraph *)h;
    auto it = g->pose.find(id);
    if (it == g->pose.end()) return SPG_EINVAL;
    it->second.assign(pose, pose + pose_stride(g->d));
    return 0;
}

// buildSubgraph, Local linearisation point without a closed form (src/vertex_remover.cpp:382-391): the
// blanket becomes a small graph (local vertex index = id), the first removed vertex is fixed, 10 LM iterations.
static bool blanket_local_lm(const BlanketIn &in, std::vector<double> &pose) {
    // clusters (m > 1, Dense / CliqueyDense): as in the reference, only the FIRST removed vertex (*toRemove.begin(), local
    // index 0) is held fixed; the other removed vertices are optimised with the kept ones (src/vertex_remover.cpp:382-391)
    int ps = pose_stride(in.d);
    RGraph g;
    g.d = in.d;
    for (int v = 0; v < in.nv; v++) { g.pose[v] = std::vector<double>(in.pose + (size_t)v * ps, in.pose + (size_t)(v + 1) * ps); g.adj[v]; }
    for (const EdgeIn &e : in.edges) {
        if (e.kind == SPG_EDGE_GLC) return false;   // (NFR blankets hold pose-pose and correlated MULTI edges; GLC has no Local point: src/topology_provider_glc.cpp:110-111)
        REdge re;
        re.kind = e.kind; re.ids.assign(e.v.begin(), e.v.end()); re.data.assign(e.data, e.data + e.len); re.alive = true;
        g.edges.push_back(re);
    }
    double st[5];
    optimize_fixed(&g, 10, std::set<int>{0}, st);
    pose.resize((size_t)in.nv * ps);
    for (int v = 0; v < in.nv; v++) std::memcpy(&pose[(size_t)v * ps], g.pose[v].data(), ps * sizeof(double));
    return true;
}
namespace { struct HookInit { HookInit() { local_lm_hook() = blanket_local_lm; } } hook_init_; }

// Batch entry spread over host threads (cpu_baseline "B": same rounds, all cores). Blankets are
// independent; outputs are produced per thread and stitched in order.
int spgref_marginalize_batch_mt(const spg_options *o, const spg_batch *b, spg_result *r, int nthreads) {
    if (nthreads <= 1) return spg_marginalize_batch(nullptr, o, b, r);
    int d = o->pose_dim, ps = pose_stride(d);
    std::vector<BlanketOut> outs(b->B);
    auto work = [&](int t) {
        for (int bi = t; bi < b->B; bi += nthreads) {
            BlanketIn in;
            in.d = d;
            in.nv = b->vert_off[bi + 1] - b->vert_off[bi];
            in.m = b->n_remove[bi];
            in.pose = b->pose + (size_t)b->vert_off[bi] * ps;
            for (int e = b->edge_off[bi]; e < b->edge_off[bi + 1]; e++) {
                EdgeIn ei;
                ei.kind = b->edge_kind[e];
                for (int v = b->edge_vert_off[e]; v < b->edge_vert_off[e + 1]; v++) ei.v.push_back(b->edge_vert[v]);
                ei.data = b->edge_data + b->edge_data_off[e];
                ei.len = b->edge_data_off[e + 1] - b->edge_data_off[e];
                in.edges.push_back(ei);
            }
            outs[bi] = run_blanket(*o, in);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
    for (auto &t : th) t.join();
    int32_t ne = 0, nev = 0;
    int64_t ned = 0;
    r->new_edge_off[0] = 0; r->new_edge_vert_off[0] = 0; r->new_edge_data_off[0] = 0;
    for (int bi = 0; bi < b->B; bi++) {
        BlanketOut &out = outs[bi];
        r->status[bi] = out.status;
        if (r->info) r->info[bi] = out.info;
        r->kld[bi] = out.kld;
        if (r->min_gap) r->min_gap[bi] = out.min_gap;
        if (r->target_info && out.target.r > 0)
            std::memcpy(r->target_info + r->target_info_off[bi], out.target.a.data(), out.target.a.size() * sizeof(double));
        for (const NewEdge &e : out.edges) {
            if (ne + 1 > r->new_edge_cap || nev + (int)e.v.size() > r->new_edge_vert_cap ||
                ned + (int64_t)e.data.size() > r->new_edge_data_cap)
                return SPG_ECAPACITY;
            r->new_edge_kind[ne] = e.kind;
            for (int v : e.v) r->new_edge_vert[nev++] = b->vert_id[b->vert_off[bi] + v];
            std::memcpy(r->new_edge_data + ned, e.data.data(), e.data.size() * sizeof(double));
            ned += (int64_t)e.data.size();
            ne++;
            r->new_edge_vert_off[ne] = nev;
            r->new_edge_data_off[ne] = ned;
        }
        r->new_edge_off[bi + 1] = ne;
    }
    return 0;
}

// ------------------------------------------------------------------------------ unit functions
void spgref_se2_edge(const double *xi, const double *xj, const double *z, double *err, double *Ji, double *Jj) {
    se2_edge(xi, xj, z, err, Ji, Jj);
}
void spgref_se3_edge(const double *xi, const double *xj, const double *z, double *err, double *Ji, double *Jj) {
    se3_edge(iso_from_tq(xi), iso_from_tq(xj), iso_from_tq(z), err, Ji, Jj);
}
// error after X <- X * fromVectorMQT(delta) on either endpoint (finite-difference checks)
void spgref_se3_error_perturbed(const double *xi, const double *xj, const double *z, const double *di, const double *dj, double *err) {
    Iso3 Xi = se3_oplus(iso_from_tq(xi), di), Xj = se3_oplus(iso_from_tq(xj), dj);
    se3_edge(Xi, Xj, iso_from_tq(z), err, nullptr, nullptr);
}
void spgref_se3_between(const double *xi, const double *xj, double *z) {
    iso_to_tq(iso_mul(iso_inv(iso_from_tq(xi)), iso_from_tq(xj)), z);
}
void spgref_se2_between(const double *xi, const double *xj, double *z) { se2_between(xi, xj, z); }
int spgref_chol(int n, const double *A, double *L) {
    Mat M(n, n);
    std::memcpy(M.a.data(), A, sizeof(double) * n * n);
    bool ok = chol_lower(M);
    std::memcpy(L, M.a.data(), sizeof(double) * n * n);
    return ok ? 0 : 1;
}
int spgref_eigh(int n, const double *A, double *w, double *V) {
    Mat M(n, n), Vm;
    std::memcpy(M.a.data(), A, sizeof(double) * n * n);
    std::vector<double> ww;
    bool ok = jacobi_eigh(M, ww, Vm);
    std::memcpy(w, ww.data(), sizeof(double) * n);
    std::memcpy(V, Vm.a.data(), sizeof(double) * n * n);
    return ok ? 0 : 1;
}
int spgref_lu_inverse(int n, const double *A, double *X) {
    Mat M(n, n);
    std::memcpy(M.a.data(), A, sizeof(double) * n * n);
    bool ok;
    Mat I = lu_inverse(M, ok);
    std::memcpy(X, I.a.data(), sizeof(double) * n * n);
    return ok ? 0 : 1;
}
// fillCliques on a given tree (pairs: k - 1 edges as 2 ints each): clique_of[i] = the clique that holds tree edge i,
// count[i] = the number of cliques that hold it. Returns the number of cliques. (Tests: every tree edge lies in exactly one.)
int spgref_fill_cliques(int k, int m, const int32_t *pairs, int32_t *clique_of, int32_t *count) {
    std::vector<std::pair<int, int>> bin;
    for (int i = 0; i < k - 1; i++) bin.push_back({pairs[2 * i], pairs[2 * i + 1]});
    Pattern pattern;
    fill_cliques(bin, k, m, pattern);
    for (int i = 0; i < k - 1; i++) { clique_of[i] = -1; count[i] = 0; }
    for (size_t j = 0; j < pattern.size(); j++)
        for (auto &pr : pattern[j])
            for (int i = 0; i < k - 1; i++)
                if (bin[i] == pr) { clique_of[i] = (int)j; count[i]++; }
    return (int)pattern.size();
}
double spgref_spd_logdet(int n, const double *A) {
    Mat M(n, n);
    std::memcpy(M.a.data(), A, sizeof(double) * n * n);
    bool ok;
    double v = spd_logdet(M, ok);
    return ok ? v : NAN;
}
// GLCReparamBinary: err (d*q) and J (dq x dq) for q poses (q x ps) with measurement meas (d*q, may be NULL)
void spgref_glc_reparam(int d, int q, const double *poses, const double *meas, double *err, double *J) {
    int ps = pose_stride(d);
    std::vector<const double *> pp(q);
    for (int i = 0; i < q; i++) pp[i] = poses + (size_t)i * ps;
    std::vector<double> e;
    Mat Jm;
    glc_reparam(d, q, pp.data(), meas, &e, &Jm);
    std::memcpy(err, e.data(), sizeof(double) * d * q);
    std::memcpy(J, Jm.a.data(), sizeof(double) * d * q * d * q);
}

}  // extern "C"
