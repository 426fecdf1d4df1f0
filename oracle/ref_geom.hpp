// oracle/ref_geom.hpp — SE2 / SE3 pose algebra, pose-pose edge error and Jacobians (CPU oracle).
//
// TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (see ref_la.hpp header / DESIGN.md).
//
// SE2: error and Jacobians are the reference's own EdgeSE2ISAM (src/se2_compatibility.h:26-51),
//      pose algebra is g2o::SE2 (third party, absent; semantics restated from the call sites
//      src/se2_compatibility.h:30, src/topology_provider_binary.hpp:46).
// SE3: with -DG2S_QUATERNIONS (CMakeLists.txt:18) EdgeSE3ISAM == g2o::EdgeSE3
//      (src/se3_compatibility.h:25-114): error = toVectorMQT(Z^-1 Xi^-1 Xj), vertex update
//      X <- X * fromVectorMQT(delta).  g2o is an un-vendored, un-pinned dependency (pre-2017-09 API,
//      cmake/FindG2O.cmake); its published definitions are restated here and the Jacobians are the
//      exact derivative of that error w.r.t. that update at delta = 0 (checked by central differences
//      in tests/test_oracle_geometry.py).
//
// Storage: SE2 pose = (x, y, theta); SE3 pose = (tx, ty, tz, qx, qy, qz, qw) as in the .g2o text.
#pragma once
#include <cmath>
#include "ref_la.hpp"

namespace spgref {

// ------------------------------------------------------------------------------------------ SE2
inline double normalize_theta(double th) {  // g2o::normalize_theta
    if (th >= -M_PI && th < M_PI) return th;
    double m = std::fmod(th, 2.0 * M_PI);
    if (m >= M_PI) m -= 2.0 * M_PI;
    if (m < -M_PI) m += 2.0 * M_PI;
    return m;
}

// x_i^-1 * x_j as (dx, dy, dtheta) — g2o::SE2 inverse()/operator* restated.
inline void se2_between(const double *xi, const double *xj, double *out) {
    double c = std::cos(xi[2]), s = std::sin(xi[2]);
    double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
    out[0] = c * dx + s * dy;
    out[1] = -s * dx + c * dy;
    out[2] = normalize_theta(xj[2] - xi[2]);
}

inline void se2_compose(const double *a, const double *b, double *out) {
    double c = std::cos(a[2]), s = std::sin(a[2]);
    out[0] = a[0] + c * b[0] - s * b[1];
    out[1] = a[1] + s * b[0] + c * b[1];
    out[2] = normalize_theta(a[2] + b[2]);
}

inline void se2_inverse(const double *a, double *out) {
    double c = std::cos(a[2]), s = std::sin(a[2]);
    out[0] = -(c * a[0] + s * a[1]);
    out[1] = -(-s * a[0] + c * a[1]);
    out[2] = normalize_theta(-a[2]);
}

// EdgeSE2ISAM::computeError / linearizeOplus, src/se2_compatibility.h:26-51.
// Ji, Jj are 3x3 row-major.
inline void se2_edge(const double *xi, const double *xj, const double *z, double *err, double *Ji,
                     double *Jj) {
    double d[3];
    se2_between(xi, xj, d);
    if (err) {
        err[0] = d[0] - z[0];
        err[1] = d[1] - z[1];
        err[2] = normalize_theta(d[2] - z[2]);
    }
    if (Ji && Jj) {
        double si = std::sin(xi[2]), ci = std::cos(xi[2]);
        double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
        Ji[0] = -ci; Ji[1] = -si; Ji[2] = -si * dx + ci * dy;
        Ji[3] = si;  Ji[4] = -ci; Ji[5] = -ci * dx - si * dy;
        Ji[6] = 0;   Ji[7] = 0;   Ji[8] = -1;
        Jj[0] = ci;  Jj[1] = si;  Jj[2] = 0;
        Jj[3] = -si; Jj[4] = ci;  Jj[5] = 0;
        Jj[6] = 0;   Jj[7] = 0;   Jj[8] = 1;
    }
}

// ------------------------------------------------------------------------------------------ SE3
struct Iso3 {
    double R[9];  // row-major
    double t[3];
};

inline void quat_to_R(const double *q /*x y z w*/, double *R) {  // Eigen::Quaternion::toRotationMatrix
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Eigen::Quaternion(Matrix3) followed by g2o::internal::normalize (unit norm, w >= 0).
inline void R_to_quat(const double *R, double *q /*x y z w*/) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
    double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; a++) q[a] /= nrm;
    if (q[3] < 0)
        for (int a = 0; a < 4; a++) q[a] = -q[a];
}

inline Iso3 iso_from_tq(const double *p) {
    Iso3 X;
    quat_to_R(p + 3, X.R);
    X.t[0] = p[0]; X.t[1] = p[1]; X.t[2] = p[2];
    return X;
}

inline void iso_to_tq(const Iso3 &X, double *p) {
    p[0] = X.t[0]; p[1] = X.t[1]; p[2] = X.t[2];
    R_to_quat(X.R, p + 3);
}

inline Iso3 iso_mul(const Iso3 &A, const Iso3 &B) {
    Iso3 C;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            C.R[i * 3 + j] = A.R[i * 3] * B.R[j] + A.R[i * 3 + 1] * B.R[3 + j] + A.R[i * 3 + 2] * B.R[6 + j];
        C.t[i] = A.R[i * 3] * B.t[0] + A.R[i * 3 + 1] * B.t[1] + A.R[i * 3 + 2] * B.t[2] + A.t[i];
    }
    return C;
}

inline Iso3 iso_inv(const Iso3 &A) {
    Iso3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C.R[i * 3 + j] = A.R[j * 3 + i];
    for (int i = 0; i < 3; i++) C.t[i] = -(C.R[i * 3] * A.t[0] + C.R[i * 3 + 1] * A.t[1] + C.R[i * 3 + 2] * A.t[2]);
    return C;
}

// g2o::internal::toVectorMQT: (t, compact quaternion with w >= 0).
inline void iso_to_mqt(const Iso3 &X, double *v) {
    double q[4];
    R_to_quat(X.R, q);
    v[0] = X.t[0]; v[1] = X.t[1]; v[2] = X.t[2];
    v[3] = q[0]; v[4] = q[1]; v[5] = q[2];
}

// g2o::internal::fromVectorMQT / fromCompactQuaternion.
inline Iso3 iso_from_mqt(const double *v) {
    Iso3 X;
    double w = 1.0 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (w < 0) {
        double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(X.R, I, sizeof I);
    } else {
        double q[4] = {v[3], v[4], v[5], std::sqrt(w)};
        quat_to_R(q, X.R);
    }
    X.t[0] = v[0]; X.t[1] = v[1]; X.t[2] = v[2];
    return X;
}

// d(compact quaternion of R)/d(R entries), entries treated as independent, 3 x 9 with the nine
// columns in COLUMN-MAJOR order of R (r00 r10 r20 r01 r11 r21 r02 r12 r22) — the contract of g2o's
// generated compute_dq_dR (dquat2mat) as used by computeEdgeSE3Gradient. Branch selection and the
// final sign flip (qw <= 0) follow g2o's _q2m.  Restated by differentiating the branch formulas.
inline void dq_dR(const double *R /*row-major*/, double *dq /*3x9 row-major*/) {
    const double r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6],
                 r21 = R[7], r22 = R[8];
    for (int i = 0; i < 27; i++) dq[i] = 0;
    auto D = [&](int comp, int row, int col) -> double & { return dq[comp * 9 + col * 3 + row]; };
    double tr = r00 + r11 + r22, S, qw;
    if (tr > 0) {
        S = std::sqrt(tr + 1.0) * 2;  // 4 qw
        qw = 0.25 * S;
        // qx = (r21 - r12)/S, qy = (r02 - r20)/S, qz = (r10 - r01)/S ; dS/dr_ii = 2/S
        double nx = r21 - r12, ny = r02 - r20, nz = r10 - r01;
        double g = -2.0 / (S * S * S);
        for (int i = 0; i < 3; i++) { D(0, i, i) = nx * g; D(1, i, i) = ny * g; D(2, i, i) = nz * g; }
        D(0, 2, 1) = 1 / S; D(0, 1, 2) = -1 / S;
        D(1, 0, 2) = 1 / S; D(1, 2, 0) = -1 / S;
        D(2, 1, 0) = 1 / S; D(2, 0, 1) = -1 / S;
    } else {
        int i;
        if ((r00 > r11) && (r00 > r22)) i = 0;
        else if (r11 > r22) i = 1;
        else i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        auto Rij = [&](int a, int b) { return R[a * 3 + b]; };
        S = std::sqrt(1.0 + Rij(i, i) - Rij(j, j) - Rij(k, k)) * 2;  // 4 q_i
        qw = (Rij(k, j) - Rij(j, k)) / S;
        // q_i = S/4 ; q_j = (R_ij + R_ji)/S ; q_k = (R_ik + R_ki)/S
        // dS/dR_ii = 2/S, dS/dR_jj = dS/dR_kk = -2/S
        double sgn[3];
        sgn[i] = 1; sgn[j] = -1; sgn[k] = -1;
        double nj = Rij(i, j) + Rij(j, i), nk = Rij(i, k) + Rij(k, i);
        for (int a = 0; a < 3; a++) {
            double dS = sgn[a] * 2.0 / S;
            D(i, a, a) = 0.25 * dS;
            D(j, a, a) = -nj / (S * S) * dS;
            D(k, a, a) = -nk / (S * S) * dS;
        }
        D(j, i, j) = 1 / S; D(j, j, i) = 1 / S;
        D(k, i, k) = 1 / S; D(k, k, i) = 1 / S;
    }
    if (qw <= 0)
        for (int a = 0; a < 27; a++) dq[a] = -dq[a];
}

// g2o::EdgeSE3::computeError + internal::computeEdgeSE3Gradient (no sensor offsets), restated.
//   E = Z^-1 Xi^-1 Xj ; err = toVectorMQT(E)
//   dE/d(delta_i), dE/d(delta_j) for X <- X * (dt, I + 2[dq]x) ; rotation rows through dq_dR(Re).
// Ji, Jj 6x6 row-major.
inline void se3_edge(const Iso3 &Xi, const Iso3 &Xj, const Iso3 &Z, double *err, double *Ji, double *Jj) {
    Iso3 A = iso_inv(Z);
    Iso3 B = iso_mul(iso_inv(Xi), Xj);
    Iso3 E = iso_mul(A, B);
    if (err) iso_to_mqt(E, err);
    if (!(Ji && Jj)) return;
    for (int a = 0; a < 36; a++) Ji[a] = Jj[a] = 0;
    const double *Ra = A.R, *Rb = B.R, *Re = E.R, *tb = B.t;
    double dq[27];
    dq_dR(Re, dq);
    // dte/dti = -Ra ; dte/dtj = Re
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { Ji[r * 6 + c] = -Ra[r * 3 + c]; Jj[r * 6 + c] = Re[r * 3 + c]; }
    // dte/dqi = Ra * (2 [tb]x)
    double S[9] = {0, -2 * tb[2], 2 * tb[1], 2 * tb[2], 0, -2 * tb[0], -2 * tb[1], 2 * tb[0], 0};
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            Ji[r * 6 + 3 + c] = Ra[r * 3] * S[c] + Ra[r * 3 + 1] * S[3 + c] + Ra[r * 3 + 2] * S[6 + c];
    // rotation rows: column c of the block = dq_dR * vec_colmajor(dRe/d(delta_c))
    for (int c = 0; c < 3; c++) {
        // generator G_c = 2 [e_c]x
        double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int a = (c + 1) % 3, b = (c + 2) % 3;
        G[b * 3 + a] = 2; G[a * 3 + b] = -2;
        // dRe/d(delta_i,c) = Ra * (-G_c) * Rb ; dRe/d(delta_j,c) = Re * G_c
        double Mi[9], Mj[9], T[9];
        for (int r = 0; r < 3; r++)
            for (int s = 0; s < 3; s++)
                T[r * 3 + s] = -(G[r * 3] * Rb[s] + G[r * 3 + 1] * Rb[3 + s] + G[r * 3 + 2] * Rb[6 + s]);
        for (int r = 0; r < 3; r++)
            for (int s = 0; s < 3; s++) {
                Mi[r * 3 + s] = Ra[r * 3] * T[s] + Ra[r * 3 + 1] * T[3 + s] + Ra[r * 3 + 2] * T[6 + s];
                Mj[r * 3 + s] = Re[r * 3] * G[s] + Re[r * 3 + 1] * G[3 + s] + Re[r * 3 + 2] * G[6 + s];
            }
        for (int comp = 0; comp < 3; comp++) {
            double si = 0, sj = 0;
            for (int col = 0; col < 3; col++)
                for (int row = 0; row < 3; row++) {
                    double d = dq[comp * 9 + col * 3 + row];
                    si += d * Mi[row * 3 + col];
                    sj += d * Mj[row * 3 + col];
                }
            Ji[(3 + comp) * 6 + 3 + c] = si;
            Jj[(3 + comp) * 6 + 3 + c] = sj;
        }
    }
}

// VertexSE3::oplus : X <- X * fromVectorMQT(delta) (used by the finite-difference tests only).
inline Iso3 se3_oplus(const Iso3 &X, const double *delta) { return iso_mul(X, iso_from_mqt(delta)); }

}  // namespace spgref
