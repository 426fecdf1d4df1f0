// csrc/spg_sparse_plan.hpp — symbolic phase of the block-sparse multifrontal Cholesky.
//
// Reference: GraphWrapperG2O::optimize (src/graph_wrapper_g2o.cpp:250-269) hands its linear systems to CHOLMOD through
// g2o, kullbackLeibler (:531-548) marginalises the baseline with Eigen::SimplicialLLT. Both are un-vendored third
// parties with their own orderings (AMD / COLAMD); nothing of their symbolic phase is observable in the reference's
// results, so this file follows no reference code: it is the host half of the device solver in spg_sparse.inc.
//
// Pure host C++ (no HIP): unknowns are d x d blocks (one per free vertex). The plan holds
//   * a fill-reducing elimination order from nested dissection with breadth-first level-set separators; when some
//     blocks are to be marginalised (global KLD) they are ordered first — nested dissection of the subgraph they induce —
//     and the kept blocks are dissected on the quotient graph in which every connected component of marginalised blocks
//     is an element that makes its kept neighbours a clique (that IS the graph of the Schur complement);
//   * supernodes = the groups nested dissection emits (a separator or a leaf), each with its sorted boundary row set
//     (supernodal symbolic factorisation), its parent in the assembly tree, its level (leaves = 0) and, for every
//     child, the position of the child's boundary rows inside the parent's front;
//   * the dense layout of the fronts: pivot block and boundary block each padded to 64 scalars, one (NP+NB)^2
//     row-major square per front in one pool.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <vector>

namespace spg {
namespace sparse {

struct BlockGraph {              // symmetric adjacency of the unknown blocks, no self loops, duplicates allowed
    int n = 0;
    std::vector<int32_t> ptr, adj;
};

struct Plan {
    int D = 0, n = 0;                         // block size, number of blocks
    std::vector<int32_t> perm, iperm;         // perm[position] = block, iperm[block] = position
    int nsn = 0, n_marg_sn = 0;               // supernodes; the first n_marg_sn eliminate marginalised blocks only
    std::vector<int32_t> first;               // [nsn+1] first position of a supernode
    std::vector<int32_t> sn_of;               // [n] supernode of a position
    std::vector<int32_t> rowptr, rows;        // boundary positions (ascending, all beyond the supernode's columns)
    std::vector<int32_t> parent, level;       // assembly tree
    std::vector<int32_t> rel;                 // [rows.size()] scalar offset of a boundary block in the PARENT's front
    std::vector<int32_t> childptr, child;     // children of a supernode, ascending
    std::vector<int32_t> NP, NB;              // padded pivot / boundary scalars (multiples of 64)
    std::vector<int64_t> foff, loff, woff;    // front offset in the pool, offset of its diagonal-tile inverses, of its local vector
    int64_t pool = 0, linv_pool = 0, work = 0;
    int nlevels = 0;
    std::vector<int32_t> level_ptr, level_sn; // supernodes grouped by level
    double flops = 0;                         // partial-factorisation flops of all fronts
    int ld(int s) const { return NP[s] + NB[s]; }
    int ncols(int s) const { return first[s + 1] - first[s]; }
    int nrows(int s) const { return rowptr[s + 1] - rowptr[s]; }
};

namespace detail {

// Nested dissection of `verts` (a subset of the blocks) on the graph g, optionally with elements: element e makes
// its members mutually adjacent. Emits groups (in elimination order) through `emit`.
struct Dissector {
    const BlockGraph &g;
    const std::vector<int32_t> *veptr = nullptr, *velem = nullptr, *emptr = nullptr, *emem = nullptr;   // quotient part
    int leaf;
    std::vector<int32_t> part, lvl, estamp, vstamp, emax;
    int32_t token = 0, stamp = 0;
    std::function<void(const std::vector<int32_t> &)> emit;

    Dissector(const BlockGraph &g_, int leaf_) : g(g_), leaf(leaf_), part((size_t)g_.n, 0), lvl((size_t)g_.n, -1), vstamp((size_t)g_.n, 0) {}
    void set_elements(const std::vector<int32_t> &vp, const std::vector<int32_t> &ve, const std::vector<int32_t> &ep, const std::vector<int32_t> &em) {
        veptr = &vp; velem = &ve; emptr = &ep; emem = &em;
        estamp.assign(ep.size() - 1, 0);
        emax.assign(ep.size() - 1, -1);
    }
    template <class F>
    void neighbours(int32_t v, int32_t tok, F f) {    // every neighbour of v inside part `tok`; an element is expanded once per stamp
        for (int32_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) { int32_t u = g.adj[p]; if (part[u] == tok) f(u); }
        if (!veptr) return;
        for (int32_t p = (*veptr)[v]; p < (*veptr)[v + 1]; p++) {
            int32_t e = (*velem)[p];
            if (estamp[e] == stamp) continue;
            estamp[e] = stamp;
            touched.push_back(e);
            for (int32_t q = (*emptr)[e]; q < (*emptr)[e + 1]; q++) { int32_t u = (*emem)[q]; if (u != v && part[u] == tok) f(u); }
        }
    }
    std::vector<int32_t> touched;
    // breadth-first search from r inside part tok; order receives the vertices level by level, lvl[] their levels
    int bfs(int32_t r, int32_t tok, std::vector<int32_t> &order) {
        stamp++;
        touched.clear();
        order.clear();
        order.push_back(r);
        vstamp[r] = stamp;
        lvl[r] = 0;
        int maxl = 0;
        for (size_t h = 0; h < order.size(); h++) {
            int32_t v = order[h];
            int l = lvl[v];
            neighbours(v, tok, [&](int32_t u) {
                if (vstamp[u] == stamp) return;
                vstamp[u] = stamp;
                lvl[u] = l + 1;
                if (l + 1 > maxl) maxl = l + 1;
                order.push_back(u);
            });
        }
        return maxl + 1;
    }
    void run(std::vector<int32_t> verts) {
        if (verts.empty()) return;
        const int32_t tok = ++token;
        for (int32_t v : verts) part[v] = tok;
        // connected components, each dissected on its own
        std::vector<std::vector<int32_t>> comps;
        {
            std::vector<int32_t> order;
            for (int32_t v : verts) lvl[v] = -2;     // -2 = not reached yet (bfs() assigns levels >= 0)
            for (int32_t v : verts) {
                if (lvl[v] != -2) continue;
                bfs(v, tok, order);
                comps.push_back(order);
            }
        }
        for (auto &comp : comps) {
            if ((int)comp.size() <= leaf) { emit(comp); continue; }
            // part token private to this component, so that searches stay inside it
            const int32_t ctok = ++token;
            for (int32_t v : comp) part[v] = ctok;
            std::vector<int32_t> order;
            int nlev = bfs(comp[0], ctok, order);
            for (int sweep = 0; sweep < 2; sweep++) {          // pseudo-peripheral start: restart from the far end
                std::vector<int32_t> o2;
                const int n2 = bfs(order.back(), ctok, o2);
                const bool grew = n2 > nlev;
                order.swap(o2);
                nlev = n2;
                if (!grew) break;
            }
            if (nlev < 3) { emit(comp); continue; }
            // level sizes and the median level
            std::vector<int32_t> cnt((size_t)nlev, 0);
            for (int32_t v : order) cnt[lvl[v]]++;
            int sepl = 1;
            {
                int64_t acc = 0, half = (int64_t)order.size() / 2;
                for (int l = 0; l < nlev; l++) { acc += cnt[l]; if (acc >= half) { sepl = l; break; } }
                sepl = std::max(1, std::min(nlev - 2, sepl));
                // a thinner level next to the median one is the better separator
                int best = sepl;
                for (int l = std::max(1, sepl - 1); l <= std::min(nlev - 2, sepl + 1); l++) if (cnt[l] < cnt[best]) best = l;
                sepl = best;
            }
            // element -> highest level among its members (for the narrowing test below)
            if (veptr) for (int32_t e : touched) {
                int m = -1;
                for (int32_t q = (*emptr)[e]; q < (*emptr)[e + 1]; q++) { int32_t u = (*emem)[q]; if (part[u] == ctok && lvl[u] > m) m = lvl[u]; }
                emax[e] = m;
            }
            std::vector<int32_t> A, B, S;
            for (int32_t v : order) {
                int l = lvl[v];
                if (l < sepl) A.push_back(v);
                else if (l > sepl) B.push_back(v);
                else {
                    bool up = false;   // only separator vertices that touch the next level have to separate
                    for (int32_t p = g.ptr[v]; p < g.ptr[v + 1] && !up; p++) { int32_t u = g.adj[p]; if (part[u] == ctok && lvl[u] == sepl + 1) up = true; }
                    if (veptr) for (int32_t p = (*veptr)[v]; p < (*veptr)[v + 1] && !up; p++) if (emax[(*velem)[p]] == sepl + 1) up = true;
                    (up ? S : A).push_back(v);
                }
            }
            if (S.empty() || A.empty() || B.empty()) { emit(comp); continue; }
            run(std::move(A));
            run(std::move(B));
            emit(S);
        }
    }
};

}  // namespace detail

// is_marg: nullptr, or one flag per block (1 = marginalised: ordered before every kept block). leaf: blocks per leaf.
inline void build_plan(const BlockGraph &g, int D, const uint8_t *is_marg, int leaf, Plan &P) {
    const int n = g.n;
    P = Plan{};
    P.D = D; P.n = n;
    P.perm.reserve((size_t)n);
    std::vector<int32_t> group_first;
    auto emit = [&](const std::vector<int32_t> &grp) {
        group_first.push_back((int32_t)P.perm.size());
        P.perm.insert(P.perm.end(), grp.begin(), grp.end());
    };
    if (!is_marg) {
        detail::Dissector ds(g, leaf);
        ds.emit = emit;
        std::vector<int32_t> all((size_t)n);
        for (int i = 0; i < n; i++) all[i] = i;
        ds.run(std::move(all));
        P.n_marg_sn = 0;
    } else {
        // stage 1: the marginalised blocks, on the subgraph they induce
        BlockGraph gm;
        gm.n = n;
        gm.ptr.assign((size_t)n + 1, 0);
        for (int v = 0; v < n; v++) {
            if (is_marg[v]) for (int32_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) if (is_marg[g.adj[p]]) gm.adj.push_back(g.adj[p]);
            gm.ptr[v + 1] = (int32_t)gm.adj.size();
        }
        std::vector<int32_t> marg, kept;
        for (int v = 0; v < n; v++) (is_marg[v] ? marg : kept).push_back(v);
        {
            detail::Dissector ds(gm, leaf);
            ds.emit = emit;
            ds.run(marg);
        }
        P.n_marg_sn = (int)group_first.size();
        // stage 2: elements = connected components of the marginalised subgraph with their kept neighbours
        std::vector<int32_t> comp((size_t)n, -1);
        int ncomp = 0;
        {
            std::vector<int32_t> stack;
            for (int32_t v : marg) {
                if (comp[v] >= 0) continue;
                comp[v] = ncomp;
                stack.push_back(v);
                while (!stack.empty()) {
                    int32_t x = stack.back();
                    stack.pop_back();
                    for (int32_t p = gm.ptr[x]; p < gm.ptr[x + 1]; p++) { int32_t u = gm.adj[p]; if (comp[u] < 0) { comp[u] = ncomp; stack.push_back(u); } }
                }
                ncomp++;
            }
        }
        std::vector<std::pair<int32_t, int32_t>> ev;   // (element, kept member)
        for (int32_t v : marg)
            for (int32_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) if (!is_marg[g.adj[p]]) ev.push_back({comp[v], g.adj[p]});
        std::sort(ev.begin(), ev.end());
        ev.erase(std::unique(ev.begin(), ev.end()), ev.end());
        std::vector<int32_t> emptr((size_t)ncomp + 1, 0), emem, veptr((size_t)n + 1, 0), velem;
        for (auto &pr : ev) { emptr[pr.first + 1]++; veptr[pr.second + 1]++; }
        for (int e = 0; e < ncomp; e++) emptr[e + 1] += emptr[e];
        for (int v = 0; v < n; v++) veptr[v + 1] += veptr[v];
        emem.resize(ev.size());
        velem.resize(ev.size());
        {
            std::vector<int32_t> ef(emptr.begin(), emptr.end() - 1), vf(veptr.begin(), veptr.end() - 1);
            for (auto &pr : ev) { emem[ef[pr.first]++] = pr.second; velem[vf[pr.second]++] = pr.first; }
        }
        BlockGraph gk;   // kept-kept adjacency
        gk.n = n;
        gk.ptr.assign((size_t)n + 1, 0);
        for (int v = 0; v < n; v++) {
            if (!is_marg[v]) for (int32_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) if (!is_marg[g.adj[p]]) gk.adj.push_back(g.adj[p]);
            gk.ptr[v + 1] = (int32_t)gk.adj.size();
        }
        detail::Dissector ds(gk, leaf);
        ds.set_elements(veptr, velem, emptr, emem);
        ds.emit = emit;
        ds.run(kept);
    }
    // ---- supernodes and their boundary rows
    const int nsn = (int)group_first.size();
    P.nsn = nsn;
    P.first = group_first;
    P.first.push_back(n);
    P.iperm.assign((size_t)n, 0);
    for (int k = 0; k < n; k++) P.iperm[P.perm[k]] = k;
    P.sn_of.assign((size_t)n, 0);
    for (int s = 0; s < nsn; s++) for (int k = P.first[s]; k < P.first[s + 1]; k++) P.sn_of[k] = s;
    P.parent.assign((size_t)nsn, -1);
    P.level.assign((size_t)nsn, 0);
    P.rowptr.assign((size_t)nsn + 1, 0);
    std::vector<int32_t> mark((size_t)n, -1), head((size_t)nsn, -1), next((size_t)nsn, -1), tailc((size_t)nsn, -1);
    std::vector<int32_t> cur;
    for (int s = 0; s < nsn; s++) {
        cur.clear();
        const int32_t end = P.first[s + 1];
        for (int k = P.first[s]; k < end; k++) {
            int32_t v = P.perm[k];
            for (int32_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) {
                int32_t q = P.iperm[g.adj[p]];
                if (q >= end && mark[q] != s) { mark[q] = s; cur.push_back(q); }
            }
        }
        for (int c = head[s]; c >= 0; c = next[c])
            for (int32_t t = P.rowptr[c]; t < P.rowptr[c + 1]; t++) {
                int32_t q = P.rows[t];
                if (q >= end && mark[q] != s) { mark[q] = s; cur.push_back(q); }
            }
        std::sort(cur.begin(), cur.end());
        P.rows.insert(P.rows.end(), cur.begin(), cur.end());
        P.rowptr[s + 1] = (int32_t)P.rows.size();
        if (!cur.empty()) {
            int par = P.sn_of[cur[0]];
            P.parent[s] = par;
            if (head[par] < 0) head[par] = s; else next[tailc[par]] = s;
            tailc[par] = s;
            P.level[par] = std::max(P.level[par], P.level[s] + 1);
        }
    }
    P.childptr.assign((size_t)nsn + 1, 0);
    for (int s = 0; s < nsn; s++) {
        for (int c = head[s]; c >= 0; c = next[c]) P.child.push_back(c);
        P.childptr[s + 1] = (int32_t)P.child.size();
    }
    // ---- layout
    P.NP.resize((size_t)nsn); P.NB.resize((size_t)nsn);
    P.foff.resize((size_t)nsn); P.loff.resize((size_t)nsn); P.woff.resize((size_t)nsn);
    for (int s = 0; s < nsn; s++) {
        const int64_t np = ((int64_t)D * P.ncols(s) + 63) / 64 * 64, nb = ((int64_t)D * P.nrows(s) + 63) / 64 * 64;
        P.NP[s] = (int32_t)np; P.NB[s] = (int32_t)nb;
        P.foff[s] = P.pool; P.pool += (np + nb) * (np + nb);
        P.loff[s] = P.linv_pool; P.linv_pool += np / 64 * 4096;
        P.woff[s] = P.work; P.work += np + nb;
        const double a = (double)D * P.ncols(s), b = (double)D * P.nrows(s);
        P.flops += a * a * a / 3 + a * a * b + a * b * b;
    }
    // ---- a child's boundary rows inside its parent's front
    P.rel.assign(P.rows.size(), 0);
    for (int c = 0; c < nsn; c++) {
        const int par = P.parent[c];
        if (par < 0) continue;
        int32_t j = P.rowptr[par];
        for (int32_t t = P.rowptr[c]; t < P.rowptr[c + 1]; t++) {
            const int32_t q = P.rows[t];
            if (q < P.first[par + 1]) { P.rel[t] = D * (q - P.first[par]); continue; }
            while (j < P.rowptr[par + 1] && P.rows[j] < q) j++;
            // multifrontal containment: every boundary row of a child is a column or a boundary row of its parent
            P.rel[t] = (j < P.rowptr[par + 1] && P.rows[j] == q) ? P.NP[par] + D * (j - P.rowptr[par]) : -1;
        }
    }
    // ---- levels
    P.nlevels = 0;
    for (int s = 0; s < nsn; s++) P.nlevels = std::max(P.nlevels, P.level[s] + 1);
    P.level_ptr.assign((size_t)P.nlevels + 1, 0);
    for (int s = 0; s < nsn; s++) P.level_ptr[P.level[s] + 1]++;
    for (int l = 0; l < P.nlevels; l++) P.level_ptr[l + 1] += P.level_ptr[l];
    P.level_sn.resize((size_t)nsn);
    {
        std::vector<int32_t> fill(P.level_ptr.begin(), P.level_ptr.end() - 1);
        for (int s = 0; s < nsn; s++) P.level_sn[fill[P.level[s]]++] = s;
    }
}

}  // namespace sparse
}  // namespace spg
