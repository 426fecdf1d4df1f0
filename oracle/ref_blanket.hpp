// oracle/ref_blanket.hpp — one Markov blanket through the reference's per-vertex pipeline (CPU oracle).
//
// TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (see ref_la.hpp header / DESIGN.md).
//
// Follows, in the reference's own operation order:
//   computeTargetInformation      src/vertex_remover.cpp:394-450  (H = sum J^T Omega J, dense Schur)
//   PseudoChowLiu                 src/pseudo_chow_liu.cpp:33-87,130-196,253-289
//   TopologyProviderBinary        src/topology_provider_binary.hpp:23-70   (NFR skeleton)
//   buildJacobianMapping          src/vertex_remover.cpp:466-498
//   optimizeInformation           src/optimizer.cpp:16-22 -> closed form only
//   LogdetFunction ctor/closedFormSolution/value/informationProduct
//                                 src/logdet_function.cpp:14-86,119-133,236-346
//   TopologyProviderGLC           src/topology_provider_glc.cpp:42-185
//   GLCEdge / GLCReparamBinary    src/glc_edge.cpp:28-49, src/glc_reparam_binary.hpp:35-127
#pragma once
#include <cstdint>
#include <limits>
#include <set>
#include <utility>
#include <vector>
#include "../include/spg.h"
#include "ref_geom.hpp"
#include "ref_la.hpp"

namespace spgref {

struct EdgeIn {
    int kind;
    std::vector<int> v;  // local blanket indices
    const double *data;
    int64_t len;
};

struct BlanketIn {
    int d;               // 3 | 6
    int nv, m;           // vertices (removed first), removed count
    const double *pose;  // nv x ps
    std::vector<EdgeIn> edges;
};

struct NewEdge {
    int kind;
    std::vector<int> v;  // local blanket indices (>= m)
    std::vector<double> data;
};

struct BlanketOut {
    int status = SPG_OK;
    int info = 0;
    Mat target;
    std::vector<NewEdge> edges;
    double kld = std::numeric_limits<double>::quiet_NaN();
    double min_gap = std::numeric_limits<double>::infinity();
};

inline int pose_stride(int d) { return d == 3 ? 3 : 7; }
inline int info_len(int d) { return d * (d + 1) / 2; }

inline Mat info_from_upper(const double *u, int d) {
    Mat O(d, d);
    int p = 0;
    for (int i = 0; i < d; i++)
        for (int j = i; j < d; j++) { O(i, j) = u[p]; O(j, i) = u[p]; p++; }
    return O;
}

// Error and Jacobians of a pose-pose edge at the given estimates (a15).
inline void binary_edge_jac(int d, const double *xi, const double *xj, const double *z, Mat &Ji, Mat &Jj,
                            double *err = nullptr) {
    Ji = Mat(d, d); Jj = Mat(d, d);
    if (d == 3) se2_edge(xi, xj, z, err, Ji.a.data(), Jj.a.data());
    else se3_edge(iso_from_tq(xi), iso_from_tq(xj), iso_from_tq(z), err, Ji.a.data(), Jj.a.data());
}

// GLCReparamBinary::reparametrize / jacobian (src/glc_reparam_binary.hpp:35-127): first pose absolute
// (mock edge from a zero vertex), the others relative to the first; `meas` plays the role of the
// mock edges' measurements (errorToMeasurement, src/glc_reparam_se2.h:38-40, src/glc_reparam_se3.h:25-27).
// poses: q x ps. err (d*q) and J (d*q x d*q) returned.
inline void glc_reparam(int d, int q, const double *const *poses, const double *meas, std::vector<double> *err,
                        Mat *J) {
    int n = d * q;
    if (err) err->assign(n, 0.0);
    if (J) *J = Mat(n, n);
    Mat Ji, Jj;
    double e[6];
    if (d == 3) {
        double zero[3] = {0, 0, 0};
        for (int i = 0; i < q; i++) {
            const double *a = (i == 0) ? zero : poses[0];
            const double *z = meas ? meas + 3 * i : zero;
            Ji = Mat(3, 3); Jj = Mat(3, 3);
            se2_edge(a, poses[i], z, e, Ji.a.data(), Jj.a.data());
            if (err) for (int c = 0; c < 3; c++) (*err)[3 * i + c] = e[c];
            if (J) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
                if (i > 0) (*J)(3 * i + r, c) = Ji(r, c);
                (*J)(3 * i + r, 3 * i + c) = Jj(r, c);
            }
        }
    } else {
        Iso3 I0;
        double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(I0.R, I, sizeof I);
        I0.t[0] = I0.t[1] = I0.t[2] = 0;
        Iso3 X0 = iso_from_tq(poses[0]);
        double zv[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < q; i++) {
            Iso3 Xa = (i == 0) ? I0 : X0;
            Iso3 Xb = iso_from_tq(poses[i]);
            Iso3 Z = iso_from_mqt(meas ? meas + 6 * i : zv);
            Ji = Mat(6, 6); Jj = Mat(6, 6);
            se3_edge(Xa, Xb, Z, e, Ji.a.data(), Jj.a.data());
            if (err) for (int c = 0; c < 6; c++) (*err)[6 * i + c] = e[c];
            if (J) for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
                if (i > 0) (*J)(6 * i + r, c) = Ji(r, c);
                (*J)(6 * i + r, 6 * i + c) = Jj(r, c);
            }
        }
    }
}

// per-vertex Jacobian blocks A_v = W * Jreparam[:, block v] of a GLC edge (src/glc_edge.cpp:40-49)
inline Mat glc_weighted_jacobian(int d, int q, const double *const *poses, const double *data, int64_t len) {
    int n = d * q;
    int r = (int)((len - n) / n);
    Mat Jr;
    glc_reparam(d, q, poses, data, nullptr, &Jr);
    Mat W(r, n);
    for (int i = 0; i < r; i++) for (int j = 0; j < n; j++) W(i, j) = data[n + (int64_t)i * n + j];
    return matmul(W, Jr);
}

// MultiEdgeCorrelated (src/multi_edge_correlated.hpp:65-140): stacked pose-pose errors of nm measurements between pairs
// of the edge's q vertices, Jacobian (d nm x d q) with two d x d blocks per measurement row block, W from the record
// (include/spg.h, SPG_EDGE_MULTI). Returns A = W J; *werr = W e.
inline Mat multi_weighted_jacobian(int d, int q, const double *const *poses, const double *data, std::vector<double> *werr) {
    const int ps = pose_stride(d), nm = (int)data[0], r = d * nm;
    const double *meas = data + 1 + 2 * nm, *Wd = meas + (size_t)nm * ps;
    Mat J(r, d * q);
    std::vector<double> err((size_t)r, 0.0);
    for (int i = 0; i < nm; i++) {
        const int a = (int)data[1 + 2 * i], b = (int)data[2 + 2 * i];
        Mat Ja, Jb;
        binary_edge_jac(d, poses[a], poses[b], meas + (size_t)i * ps, Ja, Jb, err.data() + (size_t)i * d);
        for (int x = 0; x < d; x++) for (int y = 0; y < d; y++) { J(i * d + x, a * d + y) += Ja(x, y); J(i * d + x, b * d + y) += Jb(x, y); }
    }
    Mat W(r, r);
    for (int i = 0; i < r; i++) for (int j = 0; j < r; j++) W(i, j) = Wd[(size_t)i * r + j];
    if (werr) { werr->assign((size_t)r, 0.0); for (int i = 0; i < r; i++) for (int j = 0; j < r; j++) (*werr)[i] += W(i, j) * err[j]; }
    return matmul(W, J);
}

// weighted Jacobian of any n-ary edge kind
inline Mat nary_weighted_jacobian(int kind, int d, int q, const double *const *poses, const double *data, int64_t len, std::vector<double> *werr) {
    if (kind == SPG_EDGE_MULTI) return multi_weighted_jacobian(d, q, poses, data, werr);
    int n = d * q, r = (int)((len - n) / n);
    std::vector<double> rerr;
    Mat Jr;
    glc_reparam(d, q, poses, data, werr ? &rerr : nullptr, &Jr);
    Mat W(r, n);
    for (int i = 0; i < r; i++) for (int j = 0; j < n; j++) W(i, j) = data[n + (int64_t)i * n + j];
    if (werr) { werr->assign((size_t)r, 0.0); for (int i = 0; i < r; i++) for (int j = 0; j < n; j++) (*werr)[i] += W(i, j) * rerr[j]; }
    return matmul(W, Jr);
}

// a6 + a14: H = sum_e J_e^T Omega_e J_e over the blanket edges, order [removed..., kept asc].
inline Mat assemble_hessian(const BlanketIn &in) {
    int d = in.d, ps = pose_stride(d), N = in.nv * d;
    Mat H(N, N);
    for (const EdgeIn &e : in.edges) {
        if (e.kind == SPG_EDGE_BINARY) {
            int vi = e.v[0], vj = e.v[1];
            Mat Ji, Jj;
            binary_edge_jac(d, in.pose + (size_t)vi * ps, in.pose + (size_t)vj * ps, e.data, Ji, Jj);
            Mat O = info_from_upper(e.data + ps, d);
            Mat OJi = matmul(O, Ji), OJj = matmul(O, Jj);
            Mat Hii = matmul(transpose(Ji), OJi), Hij = matmul(transpose(Ji), OJj), Hjj = matmul(transpose(Jj), OJj);
            for (int r = 0; r < d; r++) for (int c = 0; c < d; c++) {
                H(vi * d + r, vi * d + c) += Hii(r, c);
                H(vj * d + r, vj * d + c) += Hjj(r, c);
                if (vi != vj) {
                    H(vi * d + r, vj * d + c) += Hij(r, c);
                    H(vj * d + c, vi * d + r) += Hij(r, c);
                }
            }
        } else {
            int q = (int)e.v.size();
            std::vector<const double *> poses(q);
            for (int i = 0; i < q; i++) poses[i] = in.pose + (size_t)e.v[i] * ps;
            Mat A = nary_weighted_jacobian(e.kind, d, q, poses.data(), e.data, e.len, nullptr);  // r x dq, information = I
            Mat AtA = matmul(transpose(A), A);
            for (int a = 0; a < q; a++) for (int b = 0; b < q; b++)
                for (int r = 0; r < d; r++) for (int c = 0; c < d; c++)
                    H(e.v[a] * d + r, e.v[b] * d + c) += AtA(a * d + r, b * d + c);
        }
    }
    return H;
}

// a7: Lambda_t = H_kk - H_mk^T LLT(H_mm)^-1 H_mk, strict upper mirrored to lower.
inline bool schur_target(const Mat &H, int nm, Mat &target) {
    int N = H.r, n = N - nm;
    Mat Hmm = block(H, 0, 0, nm, nm), Hmk = block(H, 0, nm, nm, n);
    target = block(H, nm, nm, n, n);
    if (nm > 0) {
        Mat L = Hmm;
        if (!chol_lower(L)) return false;
        Mat Y = Hmk;
        chol_solve(L, Y);
        Mat P = matmul(transpose(Hmk), Y);
        for (size_t i = 0; i < target.a.size(); i++) target.a[i] -= P.a[i];
    }
    mirror_upper(target);
    return true;
}

struct ChowLiu {
    // a8: accepted pairs first (pop order), then rejected; kept-local vertex indices, i < j
    std::vector<std::pair<int, int>> bin;
    int n_accept = 0;
    double min_gap = std::numeric_limits<double>::infinity();
    bool ok = true;
};

// PseudoChowLiu::fillEdges + weight + doKruskal (src/pseudo_chow_liu.cpp:169-196,253-289).
// Ties in the max-heap are broken by (weight desc, i asc, j asc) — defined by this build.
inline ChowLiu chow_liu(const Mat &target, int d, int k) {
    ChowLiu cl;
    int n = d * k;
    Mat T = target;
    for (int i = 0; i < n; i++) T(i, i) += 1.0;  // tikhonov_eps = 1
    bool ok;
    Mat C = spd_inverse(T, ok);
    if (!ok) { cl.ok = false; return cl; }
    struct WE { double w; int i, j; };
    std::vector<WE> es;
    std::vector<double> ld(k);
    for (int i = 0; i < k; i++) { bool o; ld[i] = spd_logdet(block(C, i * d, i * d, d, d), o); if (!o) cl.ok = false; }
    for (int i = 0; i < k - 1; i++)
        for (int j = i + 1; j < k; j++) {
            std::vector<int> idx;
            for (int a = 0; a < d; a++) idx.push_back(i * d + a);
            for (int a = 0; a < d; a++) idx.push_back(j * d + a);
            bool o;
            double lxy = spd_logdet(select(C, idx, idx), o);
            if (!o) cl.ok = false;
            es.push_back({ld[i] + ld[j] - lxy, i, j});
        }
    if (!cl.ok) return cl;
    std::stable_sort(es.begin(), es.end(), [](const WE &a, const WE &b) {
        if (a.w != b.w) return a.w > b.w;
        if (a.i != b.i) return a.i < b.i;
        return a.j < b.j;
    });
    std::vector<int> comp(k);
    for (int i = 0; i < k; i++) comp[i] = i;
    std::vector<std::pair<int, int>> acc, rej;
    for (size_t s = 0; s < es.size(); s++) {
        int ci = comp[es[s].i], cj = comp[es[s].j];
        if (ci != cj) {
            acc.push_back({es[s].i, es[s].j});
            for (int v = 0; v < k; v++) if (comp[v] == cj) comp[v] = ci;
        } else {
            rej.push_back({es[s].i, es[s].j});
        }
    }
    // min relative gap between consecutive pops, up to the pop that follows the last accepted edge
    {
        std::vector<int> c2(k);
        for (int i = 0; i < k; i++) c2[i] = i;
        int nacc = 0;
        size_t last = 0;
        for (size_t s = 0; s < es.size() && nacc < k - 1; s++) {
            int ci = c2[es[s].i], cj = c2[es[s].j];
            if (ci != cj) { nacc++; for (int v = 0; v < k; v++) if (c2[v] == cj) c2[v] = ci; }
            last = s;
        }
        size_t upto = std::min(last + 1, es.size() - 1);
        for (size_t s = 0; s < upto; s++) {
            double a = es[s].w, b = es[s + 1].w;
            double den = std::max(std::max(std::fabs(a), std::fabs(b)), 1e-300);
            cl.min_gap = std::min(cl.min_gap, (a - b) / den);
        }
    }
    cl.n_accept = (int)acc.size();
    cl.bin = acc;
    cl.bin.insert(cl.bin.end(), rej.begin(), rej.end());
    return cl;
}

// PseudoChowLiu::marginal (src/pseudo_chow_liu.cpp:130-138): Schur complement onto `keep`
// (scalar indices, ascending), symmetric from the upper triangle.
inline bool cl_marginal(const Mat &info, const std::vector<int> &keep, Mat &out) {
    int n = info.r;
    std::vector<int> marg;
    for (int i = 0, j = 0; i < n; i++) {
        if (j < (int)keep.size() && keep[j] == i) j++;
        else marg.push_back(i);
    }
    out = select(info, keep, keep);
    if (!marg.empty()) {
        Mat mixed = select(info, keep, marg);
        Mat L = select(info, marg, marg);
        if (!chol_lower(L)) return false;
        Mat Y = transpose(mixed);
        chol_solve(L, Y);
        Mat P = matmul(mixed, Y);
        for (size_t i = 0; i < out.a.size(); i++) out.a[i] -= P.a[i];
    }
    mirror_upper(out);
    return true;
}

// posdef_pinv (src/topology_provider_glc.cpp:42-56)
inline bool posdef_pinv(const Mat &a, Mat &out) {
    std::vector<double> w;
    Mat V;
    if (!jacobi_eigh(a, w, V)) return false;
    double mx = 0;
    for (double x : w) mx = std::max(mx, std::fabs(x));
    double tol = std::numeric_limits<double>::epsilon() * std::max(a.r, a.c) * mx;
    int n = a.r;
    out = Mat(n, n);
    for (int k = 0; k < n; k++) {
        if (!(w[k] > tol)) continue;
        double inv = 1.0 / w[k];
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) out(i, j) += V(i, k) * inv * V(j, k);
    }
    return true;
}

// TopologyProviderGLC::getEdge + glc_chol (src/topology_provider_glc.cpp:59-98).
// verts: kept-local indices; returns false when W has no rows (NULL edge in the reference).
inline bool glc_get_edge(const BlanketIn &in, const Mat &targetInfo, const std::vector<int> &verts_local,
                         NewEdge &edge, int &status) {
    static const double glc_eps = 1e-8;
    int d = in.d, ps = pose_stride(d), q = (int)verts_local.size(), n = d * q;
    std::vector<const double *> poses(q);
    for (int i = 0; i < q; i++) poses[i] = in.pose + (size_t)verts_local[i] * ps;
    std::vector<double> meas;
    glc_reparam(d, q, poses.data(), nullptr, &meas, nullptr);  // reparametrize(vc)
    Mat J;
    glc_reparam(d, q, poses.data(), meas.data(), nullptr, &J);  // jacobian(vc, meas)
    bool ok;
    Mat invJ = lu_inverse(J, ok);
    if (!ok) { status = SPG_ST_NONFINITE; return false; }
    Mat m2 = matmul(transpose(invJ), matmul(targetInfo, invJ));
    // SelfAdjointEigenSolver reads the lower triangle; symmetrise so both conventions agree
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) { double s = 0.5 * (m2(i, j) + m2(j, i)); m2(i, j) = m2(j, i) = s; }
    std::vector<double> w;
    Mat V;
    if (!jacobi_eigh(m2, w, V)) { status = SPG_ST_EIG_FAIL; return false; }
    int i0 = 0;
    while (i0 < n && w[i0] < glc_eps) i0++;
    int r = n - i0;
    if (r == 0) return false;
    edge.kind = SPG_EDGE_GLC;
    edge.v = verts_local;
    edge.data.assign((size_t)n + (size_t)r * n, 0.0);
    for (int c = 0; c < n; c++) edge.data[c] = meas[c];  // computeMeasurement(): reparametrize(vertices)
    for (int e = 0; e < r; e++) {
        double s = std::sqrt(w[i0 + e]);
        for (int c = 0; c < n; c++) edge.data[n + (size_t)e * n + c] = V(c, i0 + e) * s;
    }
    return true;
}

struct Spectrum {  // LogdetFunction members _S, _U, _logdet (src/logdet_function.cpp:14-64)
    std::vector<double> S;
    Mat U;
    double logdet = 0;
};

struct JacEntry { Mat J; int off; };
typedef std::vector<JacEntry> MeasJac;  // MeasurementJacobian
typedef std::vector<MeasJac> JacMapping;

inline Mat sparse_jacobian(const JacMapping &mapping, int ncols) {  // src/logdet_function.cpp:325-346
    int rows = 0;
    for (auto &mj : mapping) rows += mj.front().J.r;
    Mat J(rows, ncols);
    int k = 0;
    for (auto &mj : mapping) {
        for (auto &je : mj)
            for (int ii = 0; ii < je.J.r; ii++) for (int jj = 0; jj < je.J.c; jj++)
                if (std::fabs(je.J(ii, jj)) >= std::numeric_limits<double>::epsilon()) J(k + ii, je.off + jj) += je.J(ii, jj);
        k += mj.front().J.r;
    }
    return J;
}

inline bool logdet_spectrum(const Mat &target, const JacMapping &mapping, Spectrum &sp, int &info) {
    int n = target.r;
    std::vector<double> w;
    Mat V;
    if (!jacobi_eigh(target, w, V)) return false;
    static const double cutoff = 1e-5;
    int smalleigs = 0;
    for (double x : w) if (x < cutoff) smalleigs++;
    int dim = mapping.front().front().J.c;
    int r = n - dim;
    sp.S.assign(r, 0.0);
    sp.U = Mat(n, r);
    if (smalleigs <= dim) {
        for (int j = 0; j < r; j++) {
            sp.S[j] = 1.0 / w[dim + j];
            for (int i = 0; i < n; i++) sp.U(i, j) = V(i, dim + j);
        }
    } else {
        info |= SPG_INFO_RANK_DEFICIENT;
        // chooseDimensions (src/logdet_function.cpp:66-81)
        Mat cand(n, smalleigs);
        for (int c = 0; c < smalleigs; c++) for (int i = 0; i < n; i++) cand(i, c) = V(i, c);
        Mat JU = matmul(sparse_jacobian(mapping, n), cand);
        std::vector<std::pair<double, int>> norms;
        for (int c = 0; c < JU.c; c++) {
            double s = 0;
            for (int i = 0; i < JU.r; i++) s += JU(i, c) * JU(i, c);
            norms.push_back({std::sqrt(s), c});
        }
        std::sort(norms.begin(), norms.end());
        std::set<int> drop;
        for (int i = 0; i < dim; i++) drop.insert(norms[i].second);
        for (int i = 0, j = 0; i < n; i++) {
            if (drop.count(i)) continue;
            sp.S[j] = std::min(std::fabs(1.0 / w[i]), 1e6 / w[n - 1]);
            for (int a = 0; a < n; a++) sp.U(a, j) = V(a, i);
            j++;
        }
    }
    sp.logdet = 0;
    for (double s : sp.S) sp.logdet += std::log(s);
    return true;
}

// closedFormSolution (src/logdet_function.cpp:236-279)
inline bool closed_form(const JacMapping &mapping, const Spectrum &sp, std::vector<Mat> &X) {
    int n = sp.U.r, r = sp.U.c;
    Mat Sigma(n, n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
        double s = 0;
        for (int k = 0; k < r; k++) s += sp.U(i, k) * sp.S[k] * sp.U(j, k);
        Sigma(i, j) = s;
    }
    mirror_lower(Sigma);
    X.clear();
    if (mapping.size() == 1) {
        Mat J = sparse_jacobian(mapping, n);
        Mat blockm = matmul(J, matmul(Sigma, transpose(J)));
        bool ok;
        Mat inv = spd_inverse(blockm, ok);
        if (!ok) return false;
        X.push_back(inv);
        return true;
    }
    for (auto &mj : mapping) {
        int dd = mj.front().J.r;
        Mat blk(dd, dd);
        for (size_t s1 = 0; s1 < mj.size(); s1++) {
            int m1 = mj[s1].J.c;
            Mat tb = matmul(mj[s1].J, matmul(block(Sigma, mj[s1].off, mj[s1].off, m1, m1), transpose(mj[s1].J)));
            for (int i = 0; i < dd; i++) for (int j = 0; j < dd; j++) blk(i, j) += 0.5 * (tb(i, j) + tb(j, i));
            for (size_t s2 = s1 + 1; s2 < mj.size(); s2++) {
                int m2 = mj[s2].J.c;
                Mat t2 = matmul(mj[s1].J, matmul(block(Sigma, mj[s1].off, mj[s2].off, m1, m2), transpose(mj[s2].J)));
                for (int i = 0; i < dd; i++) for (int j = 0; j < dd; j++) blk(i, j) += t2(i, j) + t2(j, i);
            }
        }
        bool ok;
        Mat inv = spd_inverse(blk, ok);
        if (!ok) return false;
        X.push_back(inv);
    }
    return true;
}

// informationProduct (src/logdet_function.cpp:287-323): J^T X J, symmetric from the upper triangle
inline Mat information_product(const JacMapping &mapping, const std::vector<Mat> &X, int n) {
    Mat JXJ(n, n);
    if (X.size() == 1) {
        Mat J = sparse_jacobian(mapping, n);
        JXJ = matmul(matmul(transpose(J), X[0]), J);
    } else {
        size_t e = 0;
        for (auto &mj : mapping) {
            for (size_t s1 = 0; s1 < mj.size(); s1++)
                for (size_t s2 = s1; s2 < mj.size(); s2++) {
                    const JacEntry &a = (mj[s1].off <= mj[s2].off) ? mj[s1] : mj[s2];
                    const JacEntry &b = (mj[s1].off <= mj[s2].off) ? mj[s2] : mj[s1];
                    Mat t = matmul(transpose(a.J), matmul(X[e], b.J));
                    for (int i = 0; i < t.r; i++) for (int j = 0; j < t.c; j++) JXJ(a.off + i, b.off + j) += t(i, j);
                }
            e++;
        }
    }
    mirror_upper(JXJ);
    return JXJ;
}

// LogdetFunction::value (src/logdet_function.cpp:119-133) at the given product information A = J^T X J
inline double kld_value(const Mat &A, const Spectrum &sp) {
    Mat M = matmul(transpose(sp.U), matmul(A, sp.U));
    mirror_upper(M);
    bool ok;
    double ld = spd_logdet(M, ok);
    if (!ok) return std::numeric_limits<double>::infinity();
    double tr = 0;
    int r = (int)sp.S.size();
    for (int i = 0; i < r; i++) tr += M(i, i) * sp.S[i];
    return 0.5 * (tr - ld - sp.logdet - r);
}

// ---- interior-point NFR: optimizeInformation without a closed form (src/optimizer.cpp:38-79) ----
// LogdetFunctionWithConstraints (src/logdet_function.cpp:87-214,348-427) over x = the measurement blocks X_e
// (d x d each, all d^2 entries are variables; decondense reads the LOWER triangle of the column-major block), the
// "dirty interior point" loop over rho and PQNOptimizer::optimize with useHessian = true and
// LineSearchSimpleBacktracking (src/pqn/pqn_optimizer.cpp:29-126, src/pqn/line_search.cpp:12-37), literally — the
// Hessian without the factor 1/2 of the gradient, optcond taken from the gradient BEFORE the step, and the line
// search accepting any non-increase included. Eigen's LLT of the Hessian is restated as a plain Cholesky; where
// Eigen would go on with an indefinite matrix this returns "line search failed" (infinity).
struct IpFunction {
    const JacMapping &mapping;
    const Spectrum &sp;
    int n, d, E, nx, q;
    double rho = 0;
    // state left behind by value() / gradient(), as the reference's members _chol / _xinv / _invXblocks
    bool chol_ok = false;
    Mat Mchol, xinv;
    std::vector<Mat> invX;
    Mat JU;   // sparseJacobian() * U, constant
    IpFunction(const JacMapping &m, const Spectrum &s, int n_) : mapping(m), sp(s), n(n_) {
        d = mapping.front().front().J.r; E = (int)mapping.size(); nx = d * d * E; q = d * E;
        JU = matmul(sparse_jacobian(mapping, n), sp.U);
    }
    std::vector<Mat> decondense(const std::vector<double> &x) const {   // selfadjointView<Lower> of a column-major block
        std::vector<Mat> X;
        for (int e = 0; e < E; e++) {
            Mat B(d, d);
            for (int i = 0; i < d; i++) for (int j = 0; j <= i; j++) { double v = x[(size_t)e * d * d + (size_t)j * d + i]; B(i, j) = v; B(j, i) = v; }
            X.push_back(B);
        }
        return X;
    }
    double base_value(const std::vector<double> &x) {   // LogdetFunction::value
        Mat M = matmul(transpose(sp.U), matmul(information_product(mapping, decondense(x), n), sp.U));
        mirror_upper(M);
        int r = (int)sp.S.size();
        double tr = 0;
        for (int i = 0; i < r; i++) tr += M(i, i) * sp.S[i];
        Mchol = M;
        chol_ok = chol_lower(Mchol);
        if (!chol_ok) return std::numeric_limits<double>::infinity();
        double ld = 0;
        for (int i = 0; i < r; i++) ld += std::log(Mchol(i, i));
        return 0.5 * (tr - 2.0 * ld - sp.logdet - r);
    }
    double value(const std::vector<double> &x) {        // LogdetFunctionWithConstraints::value
        double f = base_value(x);
        for (const Mat &B : decondense(x)) {
            bool ok;
            double ld = spd_logdet(B, ok);
            if (!ok) return std::numeric_limits<double>::infinity();
            f -= rho * ld;
        }
        return f;
    }
    void gradient(const std::vector<double> &x, std::vector<double> &g) {
        g.assign((size_t)nx, 0.0);
        if (!chol_ok) return;
        int r = (int)sp.S.size();
        xinv = Mat::identity(r);
        chol_solve(Mchol, xinv);
        Mat mid(r, r);
        for (int i = 0; i < r; i++) for (int j = 0; j < r; j++) mid(i, j) = -xinv(i, j) + (i == j ? sp.S[i] : 0.0);
        Mat Y = matmul(sp.U, matmul(mid, transpose(sp.U)));
        for (int e = 0; e < E; e++) {
            const MeasJac &mj = mapping[e];
            Mat blk(d, d);
            for (size_t s1 = 0; s1 < mj.size(); s1++) {
                int m1 = mj[s1].J.c;
                Mat tb = matmul(mj[s1].J, matmul(block(Y, mj[s1].off, mj[s1].off, m1, m1), transpose(mj[s1].J)));
                for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) blk(i, j) += 0.5 * (tb(i, j) + tb(j, i));
                for (size_t s2 = s1 + 1; s2 < mj.size(); s2++) {
                    int m2 = mj[s2].J.c;
                    Mat t2 = matmul(mj[s1].J, matmul(block(Y, mj[s1].off, mj[s2].off, m1, m2), transpose(mj[s2].J)));
                    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) blk(i, j) += t2(i, j) + t2(j, i);
                }
            }
            for (int j = 0; j < d; j++) for (int i = 0; i < d; i++) g[(size_t)e * d * d + (size_t)j * d + i] = 0.5 * blk(i, j);
        }
        // constraint part (src/logdet_function.cpp:372-397)
        invX.clear();
        std::vector<Mat> X = decondense(x);
        for (int e = 0; e < E; e++) {
            bool ok;
            Mat inv = spd_inverse(X[e], ok);
            if (!ok) { g.assign((size_t)nx, 0.0); return; }
            invX.push_back(inv);
            for (int j = 0; j < d; j++) for (int i = 0; i < d; i++) g[(size_t)e * d * d + (size_t)j * d + i] -= rho * inv(i, j);
        }
    }
    void hessian(Mat &H) {   // uses xinv / invX of the last gradient()
        Mat P = matmul(JU, matmul(xinv, transpose(JU)));
        for (int i = 0; i < q; i++) for (int j = i + 1; j < q; j++) { double v = 0.5 * (P(i, j) + P(j, i)); P(i, j) = v; P(j, i) = v; }
        H = Mat(nx, nx);
        for (int e = 0; e < E; e++)
            for (int jj = 0; jj < d; jj++) for (int ii = 0; ii < d; ii++) {
                const int s = e * d * d + jj * d + ii, i = e * d + ii, j = e * d + jj;
                for (int e2 = 0; e2 < E; e2++)
                    for (int vv = 0; vv < d; vv++) for (int uu = 0; uu < d; uu++)
                        H(s, e2 * d * d + vv * d + uu) = P(e2 * d + uu, i) * P(j, e2 * d + vv);
            }
        if (!chol_ok) return;
        for (int e = 0; e < (int)invX.size(); e++)
            for (int j = 0; j < d; j++) for (int i = 0; i < d; i++) {
                const int s = e * d * d + j * d + i;
                for (int v = 0; v < d; v++) for (int u = 0; u < d; u++) H(s, e * d * d + v * d + u) += rho * invX[e](u, i) * invX[e](j, v);
            }
    }
};

// PQNOptimizer::optimize (useHessian, maxIters = 0) + LineSearchSimpleBacktracking. Returns the function value,
// infinity when the line search (or the Hessian factorisation) fails; iters counts Newton steps (diagnostic).
inline double pqn_newton(IpFunction &fun, std::vector<double> &x, double tol, long &iters, bool &hessian_failed) {
    const int nx = fun.nx;
    std::vector<double> g, gn, xn((size_t)nx), dvec((size_t)nx);
    double f = fun.value(x);
    fun.gradient(x, g);
    for (;;) {
        Mat H;
        fun.hessian(H);
        if (!chol_lower(H)) { hessian_failed = true; return std::numeric_limits<double>::infinity(); }
        Mat rhs(nx, 1);
        for (int i = 0; i < nx; i++) rhs(i, 0) = -g[i];
        chol_solve(H, rhs);
        double gdotd = 0, dabs = 0;
        for (int i = 0; i < nx; i++) { dvec[i] = rhs(i, 0); gdotd += g[i] * dvec[i]; dabs += std::fabs(dvec[i]); }
        if (std::fabs(gdotd) < tol) return f;
        const double f_old = f;
        double step = 1, f_new = f;
        for (;;) {
            for (int i = 0; i < nx; i++) xn[i] = x[i] + step * dvec[i];
            f_new = fun.value(xn);      // (the reference also re-evaluates value(x) here; it has no effect on the result)
            if (step < 1e-12) return std::numeric_limits<double>::infinity();
            if (!std::isfinite(f_new) || f_new > f) { step /= 2; continue; }
            fun.gradient(xn, gn);
            break;
        }
        double optcond = 0;
        for (int i = 0; i < nx; i++) optcond += std::fabs(g[i]);   // the gradient before the step, as in the reference
        x = xn; f = f_new; g = gn;
        iters++;
        if (optcond < tol) return f;
        if (step * dabs < tol) return f;
        if (std::fabs(f - f_old) < tol) return f;
    }
}

// the interior-point branch of optimizeInformation (src/optimizer.cpp:38-79)
inline bool interior_point(const JacMapping &mapping, const Spectrum &sp, int n, std::vector<Mat> &X, double &final_value, long &iters, bool &hessian_failed) {
    IpFunction fun(mapping, sp, n);
    std::vector<double> x((size_t)fun.nx, 0.0);
    for (int e = 0; e < fun.E; e++) for (int i = 0; i < fun.d; i++) x[(size_t)e * fun.d * fun.d + (size_t)i * fun.d + i] = 1.0;   // educatedGuess
    const double startRho = 1, endRho = 5e-8, stepRho = std::sqrt(10.0);
    double tol = 1e-4;
    iters = 0;
    for (double rho = startRho; rho >= endRho; rho /= stepRho) {
        fun.rho = rho;
        if (rho / stepRho < endRho) tol = 1e-12;
        pqn_newton(fun, x, tol, iters, hessian_failed);     // the reference ignores the return value too: the next rho goes on from x
    }
    final_value = fun.base_value(x);
    if (!std::isfinite(final_value)) return false;
    X = fun.decondense(x);
    return true;
}

typedef std::vector<std::vector<std::pair<int, int>>> Pattern;   // SparsityPattern: one skeleton tree per new edge

// PseudoChowLiu::fillCliques (src/pseudo_chow_liu.cpp:198-251): the tree edges are merged into cliques while the number
// of pairwise edges the cliques stand for stays within m; every clique becomes one correlated edge over its tree edges.
inline void fill_cliques(const std::vector<std::pair<int, int>> &bin, int k, int m, Pattern &pattern) {
    std::vector<std::set<int>> cliques;
    for (int i = 0; i < k - 1; i++) cliques.push_back({bin[i].first, bin[i].second});
    bool joined = true;
    for (int nedges = k - 1, maxfill = 1; nedges < m && joined; maxfill++) {
        joined = false;
        int minfill = std::numeric_limits<int>::max();
        for (size_t i = 0; i < cliques.size(); i++)
            for (size_t j = i + 1; j < cliques.size(); j++) {
                bool meet = false;
                for (int v : cliques[j]) if (cliques[i].count(v)) { meet = true; break; }
                if (!meet) continue;
                int thisfill = ((int)cliques[i].size() - 1) * ((int)cliques[j].size() - 1);
                minfill = std::min(thisfill, minfill);
                if (thisfill <= maxfill && nedges + thisfill <= m) {
                    cliques[i].insert(cliques[j].begin(), cliques[j].end());
                    nedges += thisfill;
                    cliques.erase(cliques.begin() + (long)j);
                    joined = true;
                    j--;
                }
            }
        if (!joined && minfill > maxfill) { joined = true; maxfill = minfill - 1; }
    }
    pattern.assign(cliques.size(), {});
    for (int i = 0; i < k - 1; i++)
        for (size_t j = 0; j < cliques.size(); j++)
            if (cliques[j].count(bin[i].first) && cliques[j].count(bin[i].second)) pattern[j].push_back(bin[i]);
}

// PseudoChowLiu::computeSparsityPattern (src/pseudo_chow_liu.cpp:33-87). Kept-local pairs; an entry with more than one
// pair is a MultiEdgeCorrelated over those measurements.
inline bool sparsity_pattern(const spg_options &o, const Mat &target, int d, int k, Pattern &pattern, BlanketOut &out) {
    pattern.clear();
    int m = int((1 + o.chord_ratio) * (k - 1));
    bool full = (m >= k * (k - 1) / 2);
    if (k == 2) {
        pattern.push_back({{0, 1}});
    } else if (o.topology == SPG_TOPO_DENSE || (o.topology == SPG_TOPO_SUBGRAPH && full)) {
        for (int i = 0; i < k - 1; i++) for (int j = i + 1; j < k; j++) pattern.push_back({{i, j}});
    } else {
        ChowLiu cl = chow_liu(target, d, k);
        if (!cl.ok) { out.status = SPG_ST_TIKHONOV_NOT_PD; return false; }
        out.min_gap = cl.min_gap;
        if (o.topology == SPG_TOPO_TREE || o.topology == SPG_TOPO_SUBGRAPH) {
            int ne = (o.topology == SPG_TOPO_TREE) ? k - 1 : m;
            for (int i = 0; i < ne; i++) pattern.push_back({cl.bin[i]});
        } else if (o.topology == SPG_TOPO_CLIQUEY_DENSE || (o.topology == SPG_TOPO_CLIQUEY_SUBGRAPH && full)) {
            pattern.push_back(std::vector<std::pair<int, int>>(cl.bin.begin(), cl.bin.begin() + (k - 1)));
        } else if (o.topology == SPG_TOPO_CLIQUEY_SUBGRAPH) {
            fill_cliques(cl.bin, k, m, pattern);
        } else {
            out.status = SPG_ST_UNSUPPORTED;
            return false;
        }
    }
    return true;
}

inline void run_nfr(const spg_options &o, const BlanketIn &in, BlanketOut &out) {
    int d = in.d, ps = pose_stride(d), k = in.nv - in.m, n = d * k;
    if (k < 2) return;  // src/topology_provider_binary.hpp:28
    Pattern pattern;
    if (!sparsity_pattern(o, out.target, d, k, pattern, out)) return;
    // hasClosedFormSolution (src/logdet_function.cpp:83-86): sum of edge dims == rank (n - d)
    int jacsize = 0;
    for (auto &tree : pattern) jacsize += d * (int)tree.size();
    const bool has_closed_form = jacsize == n - d;
    JacMapping mapping;
    std::vector<NewEdge> edges;
    for (auto &tree : pattern) {
        const int nm = (int)tree.size();
        NewEdge ne;
        std::vector<int> lv;      // the edge's vertices (kept-local) in order of first appearance (addMeasurement, :28-63)
        std::vector<std::pair<int, int>> idx;
        for (auto &pr : tree) {
            int ia = -1, ib = -1;
            for (size_t t = 0; t < lv.size(); t++) { if (lv[t] == pr.first) ia = (int)t; }
            if (ia < 0) { ia = (int)lv.size(); lv.push_back(pr.first); }
            for (size_t t = 0; t < lv.size(); t++) { if (lv[t] == pr.second) ib = (int)t; }
            if (ib < 0) { ib = (int)lv.size(); lv.push_back(pr.second); }
            idx.push_back({ia, ib});
        }
        for (int v : lv) ne.v.push_back(in.m + v);
        const int q = (int)lv.size(), r = d * nm;
        std::vector<Mat> Jv((size_t)q, Mat(r, d));
        std::vector<double> meas((size_t)nm * ps, 0.0);
        for (int i = 0; i < nm; i++) {
            int a = in.m + tree[i].first, b = in.m + tree[i].second;
            const double *xa = in.pose + (size_t)a * ps, *xb = in.pose + (size_t)b * ps;
            double *z = meas.data() + (size_t)i * ps;
            Mat Ja, Jb;
            if (d == 3) {
                se2_between(xa, xb, z);  // setMeasurementFromState
                binary_edge_jac(3, xa, xb, z, Ja, Jb);
            } else {
                Iso3 Xa = iso_from_tq(xa), Xb = iso_from_tq(xb);
                Iso3 Z = iso_mul(iso_inv(Xa), Xb);
                iso_to_tq(Z, z);
                Ja = Mat(6, 6); Jb = Mat(6, 6);
                se3_edge(Xa, Xb, Z, nullptr, Ja.a.data(), Jb.a.data());
            }
            for (int x = 0; x < d; x++) for (int y = 0; y < d; y++) { Jv[idx[i].first](i * d + x, y) += Ja(x, y); Jv[idx[i].second](i * d + x, y) += Jb(x, y); }
        }
        MeasJac mj;
        for (int t = 0; t < q; t++) mj.push_back({Jv[t], lv[t] * d});
        mapping.push_back(mj);
        if (nm == 1) {
            ne.kind = SPG_EDGE_BINARY;
            ne.data.assign(ps + info_len(d), 0.0);
            for (int a = 0; a < ps; a++) ne.data[a] = meas[a];
        } else {
            ne.kind = SPG_EDGE_MULTI;
            ne.data.assign((size_t)SPG_MULTI_LEN(d, nm), 0.0);
            ne.data[0] = nm;
            for (int i = 0; i < nm; i++) { ne.data[1 + 2 * i] = idx[i].first; ne.data[2 + 2 * i] = idx[i].second; }
            for (size_t a = 0; a < meas.size(); a++) ne.data[1 + 2 * nm + a] = meas[a];
        }
        edges.push_back(ne);
    }
    Spectrum sp;
    if (!logdet_spectrum(out.target, mapping, sp, out.info)) { out.status = SPG_ST_EIG_FAIL; return; }
    std::vector<Mat> X;
    if (has_closed_form) {
        if (!closed_form(mapping, sp, X)) { out.status = SPG_ST_CLOSED_FORM_NOT_PD; return; }
    } else {
        for (auto &tree : pattern) if (tree.size() != 1) { out.status = SPG_ST_UNSUPPORTED; return; }
        double fin = 0;
        long iters = 0;
        bool hfail = false;
        if (!interior_point(mapping, sp, n, X, fin, iters, hfail)) { out.status = SPG_ST_KLD_NOT_PD; return; }   // the reference exit(0)s here
        if (hfail) out.info |= SPG_INFO_IP_HESSIAN_NOT_PD;
        out.info |= (int)std::min<long>(iters, 32767) << 8;   // Newton steps taken (diagnostic, bits 8..)
    }
    for (size_t e = 0; e < edges.size(); e++) {
        if (edges[e].kind == SPG_EDGE_BINARY) {
            int p = ps;
            for (int i = 0; i < d; i++) for (int j = i; j < d; j++) edges[e].data[p++] = X[e](i, j);
        } else {
            // information X = W^T W with W = chol(X)^T (any factor does: consumers use W^T W)
            const int nm = (int)edges[e].data[0], r = d * nm;
            Mat L = X[e];
            for (int i = 0; i < r; i++) for (int j = i + 1; j < r; j++) { double v = 0.5 * (L(i, j) + L(j, i)); L(i, j) = v; L(j, i) = v; }
            if (!chol_lower(L)) { out.status = SPG_ST_CLOSED_FORM_NOT_PD; return; }
            double *W = edges[e].data.data() + 1 + 2 * nm + (size_t)nm * ps;
            for (int i = 0; i < r; i++) for (int j = 0; j < r; j++) W[(size_t)i * r + j] = (j >= i) ? L(j, i) : 0.0;
        }
    }
    out.edges = edges;
    out.kld = kld_value(information_product(mapping, X, n), sp);
    if (!std::isfinite(out.kld)) out.status = SPG_ST_KLD_NOT_PD;
}

inline void run_glc(const spg_options &o, const BlanketIn &in, BlanketOut &out) {
    int d = in.d, k = in.nv - in.m, n = d * k;
    if (!(o.topology == SPG_TOPO_DENSE || o.topology == SPG_TOPO_TREE) || o.lin_point != SPG_LIN_GLOBAL) {
        out.status = SPG_ST_UNSUPPORTED;  // asserts at src/topology_provider_glc.cpp:107-111
        return;
    }
    if (k == 0) return;
    std::vector<int> all(k);
    for (int i = 0; i < k; i++) all[i] = in.m + i;
    if (k == 1 || o.topology == SPG_TOPO_DENSE) {
        NewEdge e;
        int st = SPG_OK;
        if (glc_get_edge(in, out.target, all, e, st)) {
            out.edges.push_back(e);
            if (k == 1) out.info |= SPG_INFO_GLC_ROOT_EDGE;
        }
        if (st != SPG_OK) out.status = st;
    } else {
        std::vector<std::pair<int, int>> pairs;
        {
            Pattern pat;
            if (!sparsity_pattern(o, out.target, d, k, pat, out)) return;
            for (auto &tree : pat) pairs.push_back(tree.front());
        }
        // root unary edge from the marginal of the first vertex of the first tree edge
        {
            int root = pairs.front().first;
            std::vector<int> keep;
            for (int a = 0; a < d; a++) keep.push_back(root * d + a);
            Mat rootInfo;
            if (!cl_marginal(out.target, keep, rootInfo)) { out.status = SPG_ST_MARGINAL_NOT_PD; return; }
            NewEdge e;
            int st = SPG_OK;
            if (glc_get_edge(in, rootInfo, {in.m + root}, e, st)) {
                out.edges.push_back(e);
                out.info |= SPG_INFO_GLC_ROOT_EDGE;
            }
            if (st != SPG_OK) { out.status = st; return; }
        }
        for (auto &pr : pairs) {
            std::vector<int> keep;
            for (int a = 0; a < d; a++) keep.push_back(pr.first * d + a);
            for (int a = 0; a < d; a++) keep.push_back(pr.second * d + a);
            Mat joint;
            if (!cl_marginal(out.target, keep, joint)) { out.status = SPG_ST_MARGINAL_NOT_PD; return; }
            Mat b1 = block(joint, d, 0, d, d), b3 = block(joint, 0, d, d, d), b2;
            if (!posdef_pinv(block(joint, 0, 0, d, d), b2)) { out.status = SPG_ST_EIG_FAIL; return; }
            Mat m4 = matmul(b1, matmul(b2, b3));
            Mat tgt(2 * d, 2 * d);
            for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) {
                tgt(i, j) = joint(i, j);
                tgt(i, d + j) = joint(i, d + j);
                tgt(d + i, j) = joint(d + i, j);
                tgt(d + i, d + j) = m4(i, j);
            }
            mirror_upper(tgt);  // target.selfadjointView<Upper>()
            NewEdge e;
            int st = SPG_OK;
            if (glc_get_edge(in, tgt, {in.m + pr.first, in.m + pr.second}, e, st)) out.edges.push_back(e);
            if (st != SPG_OK) { out.status = st; return; }
        }
    }
    if (o.flags & 1) {   // oracle-private diagnostic (bit 0 is reserved in include/spg.h; the product ignores it)
        // diagnostic defined by this build: value() of src/logdet_function.cpp:119-133 evaluated at
        // the product information of the GLC edges, spectrum as in the LogdetFunction constructor
        int ps = pose_stride(d);
        Mat A(n, n);
        for (auto &e : out.edges) {
            int q = (int)e.v.size();
            std::vector<const double *> poses(q);
            for (int i = 0; i < q; i++) poses[i] = in.pose + (size_t)e.v[i] * ps;
            Mat Aw = nary_weighted_jacobian(e.kind, d, q, poses.data(), e.data.data(), (int64_t)e.data.size(), nullptr);
            Mat AtA = matmul(transpose(Aw), Aw);
            for (int a = 0; a < q; a++) for (int b = 0; b < q; b++)
                for (int r = 0; r < d; r++) for (int c = 0; c < d; c++)
                    A((e.v[a] - in.m) * d + r, (e.v[b] - in.m) * d + c) += AtA(a * d + r, b * d + c);
        }
        JacMapping dummy;
        dummy.push_back({{Mat(d, d), 0}});
        Spectrum sp;
        int inf = 0;
        if (k >= 2 && logdet_spectrum(out.target, dummy, sp, inf) && !(inf & SPG_INFO_RANK_DEFICIENT))
            out.kld = kld_value(A, sp);
    }
}

// buildSubgraph, Local linearisation point (src/vertex_remover.cpp:304-381): a closed-form estimate
// exists iff every vertex other than the first removed one appears in exactly one blanket edge and
// every edge can propagate an estimate (pose-pose edges can, GLC edges cannot:
// src/glc_edge.cpp:58-63). The removed vertex is placed at the origin and every neighbour is set from
// its edge's measurement (g2o EdgeSE2/EdgeSE3::initialEstimate: to = from * z, or from = to * z^-1).
inline bool local_linearization_point(const BlanketIn &in, std::vector<double> &pose) {
    int ps = pose_stride(in.d);
    std::vector<int> cnt(in.nv, 0);
    for (const EdgeIn &e : in.edges) {
        if (e.kind != SPG_EDGE_BINARY) return false;
        for (int v : e.v) if (v != 0) cnt[v]++;
    }
    for (int v = 1; v < in.nv; v++) if (cnt[v] > 1) return false;
    pose.assign(in.pose, in.pose + (size_t)in.nv * ps);
    if (in.d == 3) { pose[0] = pose[1] = pose[2] = 0; }
    else { double I7[7] = {0, 0, 0, 0, 0, 0, 1}; std::memcpy(pose.data(), I7, sizeof I7); }
    for (const EdgeIn &e : in.edges) {
        int vi = e.v[0], vj = e.v[1];
        if (vi == 0 && vj == 0) continue;
        if (in.d == 3) {
            if (vi == 0) se2_compose(&pose[0], e.data, &pose[(size_t)vj * ps]);
            else { double zi[3]; se2_inverse(e.data, zi); se2_compose(&pose[(size_t)vj * ps], zi, &pose[(size_t)vi * ps]); }
        } else {
            Iso3 Z = iso_from_tq(e.data);
            if (vi == 0) iso_to_tq(iso_mul(iso_from_tq(&pose[0]), Z), &pose[(size_t)vj * ps]);
            else iso_to_tq(iso_mul(iso_from_tq(&pose[(size_t)vj * ps]), iso_inv(Z)), &pose[(size_t)vi * ps]);
        }
    }
    return true;
}

inline BlanketOut run_blanket_at(const spg_options &o, const BlanketIn &in);

// set by spg_ref.cpp: optimise the blanket's estimates in place of a closed form (returns false if it cannot)
typedef bool (*LocalLmFn)(const BlanketIn &in, std::vector<double> &pose);
inline LocalLmFn &local_lm_hook() { static LocalLmFn f = nullptr; return f; }

// One iteration of VertexRemover::remove (src/vertex_remover.cpp:108-132) on a gathered blanket.
inline BlanketOut run_blanket(const spg_options &o, const BlanketIn &in) {
    BlanketOut out;
    if (in.edges.empty() || in.m < 1) { out.status = SPG_ST_EMPTY_BLANKET; return out; }
    BlanketIn local = in;
    std::vector<double> local_pose;
    if (o.lin_point != SPG_LIN_GLOBAL) {
        if (!local_linearization_point(in, local_pose)) {
            // no closed-form estimate: 10 LM iterations on the subgraph with the first removed vertex fixed at
            // its current estimate (src/vertex_remover.cpp:382-391). The LM lives with the graph code.
            if (!local_lm_hook() || !local_lm_hook()(in, local_pose)) { out.status = SPG_ST_NEEDS_LOCAL_OPTIMIZATION; return out; }
        }
        local.pose = local_pose.data();
    }
    return run_blanket_at(o, local);
}

inline BlanketOut run_blanket_at(const spg_options &o, const BlanketIn &in) {
    BlanketOut out;
    if (o.algorithm == SPG_ALG_NFR)  // binary providers take pose-pose and correlated multi edges, not GLC edges (src/topology_provider_binary.hpp:16-21)
        for (const EdgeIn &e : in.edges) if (e.kind == SPG_EDGE_GLC) { out.status = SPG_ST_UNSUPPORTED; return out; }
    Mat H = assemble_hessian(in);
    if (!schur_target(H, in.m * in.d, out.target)) { out.status = SPG_ST_HMM_NOT_PD; return out; }
    for (double v : out.target.a) if (!std::isfinite(v)) { out.status = SPG_ST_NONFINITE; return out; }
    if (o.algorithm == SPG_ALG_GLC) run_glc(o, in, out);
    else run_nfr(o, in, out);
    return out;
}

}  // namespace spgref
