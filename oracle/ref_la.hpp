// oracle/ref_la.hpp — dense fp64 linear algebra for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing under oracle/ is linked, loaded or
// called by the product path. PARITY UNPINNED: the reference owns no golden vectors for this path
// and cannot be built here (Eigen/g2o/iSAM absent) — see DESIGN.md.
//
// The reference does all of this through Eigen (LLT, LDLT, SelfAdjointEigenSolver, PartialPivLU);
// every call site is cited where the routine is used (ref_blanket.hpp). Eigen is not in this
// container, so the routines are restated from their textbook definitions; the quantities the
// reference consumes (A^-1, log det, invariant subspaces) do not depend on the factorisation
// variant beyond rounding.  All matrices are row-major, leading dimension = number of columns.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

namespace spgref {

struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
    static Mat identity(int n) {
        Mat m(n, n);
        for (int i = 0; i < n; i++) m(i, i) = 1.0;
        return m;
    }
};

inline Mat matmul(const Mat &A, const Mat &B) {
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; i++)
        for (int k = 0; k < A.c; k++) {
            double aik = A(i, k);
            if (aik == 0.0) continue;
            for (int j = 0; j < B.c; j++) C(i, j) += aik * B(k, j);
        }
    return C;
}

inline Mat transpose(const Mat &A) {
    Mat T(A.c, A.r);
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) T(j, i) = A(i, j);
    return T;
}

inline Mat block(const Mat &A, int i0, int j0, int nr, int nc) {
    Mat B(nr, nc);
    for (int i = 0; i < nr; i++)
        for (int j = 0; j < nc; j++) B(i, j) = A(i0 + i, j0 + j);
    return B;
}

// selectVariables (reference src/utils.cpp:27-46): gather rows/cols by index list.
inline Mat select(const Mat &A, const std::vector<int> &rows, const std::vector<int> &cols) {
    Mat B((int)rows.size(), (int)cols.size());
    for (size_t i = 0; i < rows.size(); i++)
        for (size_t j = 0; j < cols.size(); j++) B((int)i, (int)j) = A(rows[i], cols[j]);
    return B;
}

// Lower Cholesky A = L L^T (reads the lower triangle, like Eigen::LLT<…, Lower>). Returns false if
// a pivot is not strictly positive / not finite.
inline bool chol_lower(Mat &A) {
    int n = A.r;
    for (int j = 0; j < n; j++) {
        double d = A(j, j);
        for (int k = 0; k < j; k++) d -= A(j, k) * A(j, k);
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        double l = std::sqrt(d);
        A(j, j) = l;
        for (int i = j + 1; i < n; i++) {
            double s = A(i, j);
            for (int k = 0; k < j; k++) s -= A(i, k) * A(j, k);
            A(i, j) = s / l;
        }
        for (int i = 0; i < j; i++) A(i, j) = 0.0;
    }
    return true;
}

// Solve L L^T X = B in place (B is n x m).
inline void chol_solve(const Mat &L, Mat &B) {
    // row-sweeping form (contiguous inner loops); per entry the same operations in the same order as
    // the textbook column-by-column substitution
    int n = L.r, m = B.c;
    for (int i = 0; i < n; i++) {
        double *bi = &B.a[(size_t)i * m];
        for (int k = 0; k < i; k++) {
            double l = L(i, k);
            if (l == 0.0) continue;
            const double *bk = &B.a[(size_t)k * m];
            for (int c = 0; c < m; c++) bi[c] -= l * bk[c];
        }
        double d = L(i, i);
        for (int c = 0; c < m; c++) bi[c] /= d;
    }
    for (int i = n - 1; i >= 0; i--) {
        double *bi = &B.a[(size_t)i * m];
        for (int k = i + 1; k < n; k++) {
            double l = L(k, i);
            if (l == 0.0) continue;
            const double *bk = &B.a[(size_t)k * m];
            for (int c = 0; c < m; c++) bi[c] -= l * bk[c];
        }
        double d = L(i, i);
        for (int c = 0; c < m; c++) bi[c] /= d;
    }
}

// A^-1 of an SPD matrix via Cholesky (Eigen idiom: A.llt().solve(Identity)). ok=false if not PD.
inline Mat spd_inverse(const Mat &A, bool &ok) {
    Mat L = A;
    ok = chol_lower(L);
    Mat X = Mat::identity(A.r);
    if (ok) chol_solve(L, X);
    return X;
}

// log det of an SPD matrix. The reference takes sum(log(vectorD)) of a pivoted LDLT
// (src/pseudo_chow_liu.cpp:178-182, src/logdet_function.cpp:123-127); the value is the same
// 2*sum(log L_ii) up to rounding. ok=false mirrors "isPositive() && all D > 0" failing.
inline double spd_logdet(const Mat &A, bool &ok) {
    Mat L = A;
    ok = chol_lower(L);
    if (!ok) return NAN;
    double s = 0;
    for (int i = 0; i < A.r; i++) s += std::log(L(i, i));
    return 2.0 * s;
}

// Symmetric eigendecomposition by cyclic Jacobi (reads the full matrix, assumes symmetry).
// w ascending, V columns = eigenvectors (Eigen::SelfAdjointEigenSolver contract). The reference
// consumers only use invariant-subspace quantities (U S U^T, U^T A U traces/dets), so the
// algorithm choice is immaterial beyond rounding. Returns false if not converged.
inline bool jacobi_eigh(const Mat &Ain, std::vector<double> &w, Mat &V, int max_sweeps = 60) {
    int n = Ain.r;
    Mat A = Ain;
    V = Mat::identity(n);
    double fro2 = 0;
    for (double v : A.a) fro2 += v * v;
    bool converged = (n <= 1) || fro2 == 0.0;
    for (int sweep = 0; sweep < max_sweeps && !converged; sweep++) {
        double off2 = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                if (i != j) off2 += A(i, j) * A(i, j);
        if (off2 <= 1e-31 * fro2) { converged = true; break; }
        bool rotated = false;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A(p, q);
                if (apq == 0.0) continue;
                double app = A(p, p), aqq = A(q, q);
                if (std::fabs(apq) <= 1e-300) { A(p, q) = A(q, p) = 0; continue; }
                double tau = (aqq - app) / (2.0 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
                rotated = true;
                for (int i = 0; i < n; i++) {
                    double aip = A(i, p), aiq = A(i, q);
                    A(i, p) = c * aip - s * aiq;
                    A(i, q) = s * aip + c * aiq;
                }
                for (int j = 0; j < n; j++) {
                    double apj = A(p, j), aqj = A(q, j);
                    A(p, j) = c * apj - s * aqj;
                    A(q, j) = s * apj + c * aqj;
                }
                A(p, q) = A(q, p) = 0.0;
                for (int i = 0; i < n; i++) {
                    double vip = V(i, p), viq = V(i, q);
                    V(i, p) = c * vip - s * viq;
                    V(i, q) = s * vip + c * viq;
                }
            }
        if (!rotated) converged = true;
    }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return A(a, a) < A(b, b); });
    w.resize(n);
    Mat Vs(n, n);
    for (int k = 0; k < n; k++) {
        w[k] = A(order[k], order[k]);
        for (int i = 0; i < n; i++) Vs(i, k) = V(i, order[k]);
    }
    V = Vs;
    return converged;
}

// General inverse by LU with partial pivoting (Eigen::PartialPivLU(J).solve(Identity),
// src/topology_provider_glc.cpp:63-64). ok=false on an exactly zero pivot.
inline Mat lu_inverse(const Mat &Ain, bool &ok) {
    int n = Ain.r;
    Mat A = Ain, X = Mat::identity(n);
    ok = true;
    for (int k = 0; k < n; k++) {
        int piv = k;
        double best = std::fabs(A(k, k));
        for (int i = k + 1; i < n; i++)
            if (std::fabs(A(i, k)) > best) { best = std::fabs(A(i, k)); piv = i; }
        if (best == 0.0 || !std::isfinite(best)) { ok = false; return X; }
        if (piv != k)
            for (int j = 0; j < n; j++) {
                std::swap(A(k, j), A(piv, j));
                std::swap(X(k, j), X(piv, j));
            }
        for (int i = k + 1; i < n; i++) {
            double f = A(i, k) / A(k, k);
            if (f == 0.0) continue;
            A(i, k) = 0;
            for (int j = k + 1; j < n; j++) A(i, j) -= f * A(k, j);
            for (int j = 0; j < n; j++) X(i, j) -= f * X(k, j);
        }
    }
    for (int c = 0; c < n; c++)
        for (int i = n - 1; i >= 0; i--) {
            double s = X(i, c);
            for (int k = i + 1; k < n; k++) s -= A(i, k) * X(k, c);
            X(i, c) = s / A(i, i);
        }
    return X;
}

// copy strict upper -> strict lower (reference idiom
// `M.triangularView<StrictlyLower>() = M.triangularView<StrictlyUpper>().transpose()`).
inline void mirror_upper(Mat &A) {
    for (int i = 0; i < A.r; i++)
        for (int j = i + 1; j < A.c; j++) A(j, i) = A(i, j);
}
inline void mirror_lower(Mat &A) {
    for (int i = 0; i < A.r; i++)
        for (int j = i + 1; j < A.c; j++) A(i, j) = A(j, i);
}

}  // namespace spgref
