// csrc/spg_dev_geom.hpp — SE2 / SE3 pose-pose edge error Jacobians on the device (gfx950).
//
// Device twin of the arithmetic the reference reaches through
//   EdgeSE2ISAM::linearizeOplus          src/se2_compatibility.h:35-51
//   g2o::EdgeSE3 (== EdgeSE3ISAM under G2S_QUATERNIONS, src/se3_compatibility.h:25-114):
//       error toVectorMQT(Z^-1 Xi^-1 Xj), update X <- X * fromVectorMQT(delta)
//   setMeasurementFromState              src/topology_provider_binary.hpp:46
// One lane evaluates one edge; every loop has compile-time bounds so all temporaries stay in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

namespace spgdev {

// SE3 pose in kernel form: R row-major (9) + t (3)
constexpr int kIso = 12;

__device__ __forceinline__ void quat_to_R(const double *q, double *R) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

template <int I>
__device__ __forceinline__ void R_to_quat_case(const double *R, double *q) {
    constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
    double t = sqrt(R[I * 4] - R[J * 4] - R[K * 4] + 1.0);
    q[I] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[K * 3 + J] - R[J * 3 + K]) * t;
    q[J] = (R[J * 3 + I] + R[I * 3 + J]) * t;
    q[K] = (R[K * 3 + I] + R[I * 3 + K]) * t;
}

// Eigen::Quaternion(Matrix3) then unit norm with w >= 0 (g2o::internal::normalize)
__device__ __forceinline__ void R_to_quat(const double *R, double *q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > (i == 0 ? R[0] : R[4])) i = 2;
        if (i == 0) R_to_quat_case<0>(R, q);
        else if (i == 1) R_to_quat_case<1>(R, q);
        else R_to_quat_case<2>(R, q);
    }
    double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double sc = (q[3] < 0 ? -1.0 : 1.0) / nrm;
#pragma unroll
    for (int a = 0; a < 4; a++) q[a] *= sc;
}

__device__ __forceinline__ void iso_from_tq(const double *p, double *X) {
    quat_to_R(p + 3, X);
    X[9] = p[0]; X[10] = p[1]; X[11] = p[2];
}

// C = A^-1 * B
__device__ __forceinline__ void iso_inv_mul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
        C[9 + i] = A[i] * (B[9] - A[9]) + A[3 + i] * (B[10] - A[10]) + A[6 + i] * (B[11] - A[11]);
    }
}

// g2o::internal::fromVectorMQT
__device__ __forceinline__ void iso_from_mqt(const double *v, double *X) {
    double w = 1.0 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (w < 0) {
        X[0] = 1; X[1] = 0; X[2] = 0; X[3] = 0; X[4] = 1; X[5] = 0; X[6] = 0; X[7] = 0; X[8] = 1;
    } else {
        double q[4] = {v[3], v[4], v[5], sqrt(w)};
        quat_to_R(q, X);
    }
    X[9] = v[0]; X[10] = v[1]; X[11] = v[2];
}

// d(compact quaternion)/d(R), 3 x 9, columns in column-major order of R (contract of g2o's compute_dq_dR)
template <int I>
__device__ __forceinline__ void dq_dR_case(const double *R, double *dq, double &qw) {
    constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
    double S = sqrt(1.0 + R[I * 4] - R[J * 4] - R[K * 4]) * 2;
    qw = (R[K * 3 + J] - R[J * 3 + K]) / S;
    double nj = R[I * 3 + J] + R[J * 3 + I], nk = R[I * 3 + K] + R[K * 3 + I];
    double iS = 1.0 / S, iS2 = iS * iS;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double dS = ((a == I) ? 2.0 : -2.0) * iS;
        dq[I * 9 + a * 3 + a] = 0.25 * dS;
        dq[J * 9 + a * 3 + a] = -nj * iS2 * dS;
        dq[K * 9 + a * 3 + a] = -nk * iS2 * dS;
    }
    // D(comp,row,col) = dq[comp*9 + col*3 + row]
    dq[J * 9 + J * 3 + I] = iS; dq[J * 9 + I * 3 + J] = iS;
    dq[K * 9 + K * 3 + I] = iS; dq[K * 9 + I * 3 + K] = iS;
}

__device__ __forceinline__ void dq_dR(const double *R, double *dq) {
#pragma unroll
    for (int i = 0; i < 27; i++) dq[i] = 0;
    double tr = R[0] + R[4] + R[8], qw;
    if (tr > 0) {
        double S = sqrt(tr + 1.0) * 2;
        qw = 0.25 * S;
        double nx = R[7] - R[5], ny = R[2] - R[6], nz = R[3] - R[1];
        double iS = 1.0 / S;
        double g = -2.0 * iS * iS * iS;
#pragma unroll
        for (int i = 0; i < 3; i++) { dq[0 * 9 + i * 4] = nx * g; dq[1 * 9 + i * 4] = ny * g; dq[2 * 9 + i * 4] = nz * g; }
        dq[0 * 9 + 1 * 3 + 2] = iS; dq[0 * 9 + 2 * 3 + 1] = -iS;
        dq[1 * 9 + 2 * 3 + 0] = iS; dq[1 * 9 + 0 * 3 + 2] = -iS;
        dq[2 * 9 + 0 * 3 + 1] = iS; dq[2 * 9 + 1 * 3 + 0] = -iS;
    } else if ((R[0] > R[4]) && (R[0] > R[8])) {
        dq_dR_case<0>(R, dq, qw);
    } else if (R[4] > R[8]) {
        dq_dR_case<1>(R, dq, qw);
    } else {
        dq_dR_case<2>(R, dq, qw);
    }
    if (qw <= 0) {
#pragma unroll
        for (int a = 0; a < 27; a++) dq[a] = -dq[a];
    }
}

// Jacobians of err = toVectorMQT(Z^-1 Xi^-1 Xj) w.r.t. the right-multiplicative updates of Xi, Xj.
// A = Z^-1 is passed in (for a new edge built from the state A = (Xi^-1 Xj)^-1 without a round trip).
// Ji/Jj: 6x6 row-major, written with stride 1 into the given buffers (LDS or global).
__device__ __forceinline__ void se3_edge_jac(const double *Xi, const double *Xj, const double *Z,
                                             double *Ji, double *Jj, double *err) {
    double B[kIso], E[kIso], Ra[9];
    iso_inv_mul(Xi, Xj, B);   // B = Xi^-1 Xj
    iso_inv_mul(Z, B, E);     // E = Z^-1 B
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Ra[i * 3 + j] = Z[j * 3 + i];
    if (err) {
        double q[4];
        R_to_quat(E, q);
        err[0] = E[9]; err[1] = E[10]; err[2] = E[11]; err[3] = q[0]; err[4] = q[1]; err[5] = q[2];
    }
    double dq[27];
    dq_dR(E, dq);
    const double *tb = B + 9;
    double S[9] = {0, -2 * tb[2], 2 * tb[1], 2 * tb[2], 0, -2 * tb[0], -2 * tb[1], 2 * tb[0], 0};
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            Ji[r * 6 + c] = -Ra[r * 3 + c];
            Jj[r * 6 + c] = E[r * 3 + c];
            Ji[r * 6 + 3 + c] = Ra[r * 3] * S[c] + Ra[r * 3 + 1] * S[3 + c] + Ra[r * 3 + 2] * S[6 + c];
            Jj[r * 6 + 3 + c] = 0;
            Ji[(3 + r) * 6 + c] = 0;
            Jj[(3 + r) * 6 + c] = 0;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        constexpr int A1[3] = {1, 2, 0}, B1[3] = {2, 0, 1};
        const int a = A1[c], b = B1[c];
        // G = 2[e_c]x : G[b][a] = 2, G[a][b] = -2
        // T = -G * Rb : row b = -2 Rb[a,:], row a = +2 Rb[b,:], row c = 0
        double T[9];
#pragma unroll
        for (int s = 0; s < 3; s++) { T[c * 3 + s] = 0; T[b * 3 + s] = -2 * B[a * 3 + s]; T[a * 3 + s] = 2 * B[b * 3 + s]; }
        double Mi[9], Mj[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++) {
                Mi[r * 3 + s] = Ra[r * 3] * T[s] + Ra[r * 3 + 1] * T[3 + s] + Ra[r * 3 + 2] * T[6 + s];
                // Re * G : column a of result = 2 * Re[:, b], column b = -2 * Re[:, a], column c = 0
                Mj[r * 3 + s] = (s == a) ? 2 * E[r * 3 + b] : ((s == b) ? -2 * E[r * 3 + a] : 0.0);
            }
#pragma unroll
        for (int comp = 0; comp < 3; comp++) {
            double si = 0, sj = 0;
#pragma unroll
            for (int col = 0; col < 3; col++)
#pragma unroll
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

// ------------------------------------------------------------------------------------------ SE2
__device__ __forceinline__ double normalize_theta(double th) {
    const double PI = 3.14159265358979323846;
    if (th >= -PI && th < PI) return th;
    double m = fmod(th, 2.0 * PI);
    if (m >= PI) m -= 2.0 * PI;
    if (m < -PI) m += 2.0 * PI;
    return m;
}

__device__ __forceinline__ void se2_between(const double *xi, const double *xj, double *out) {
    double c = cos(xi[2]), s = sin(xi[2]);
    double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
    out[0] = c * dx + s * dy;
    out[1] = -s * dx + c * dy;
    out[2] = normalize_theta(xj[2] - xi[2]);
}

__device__ __forceinline__ void se2_edge_jac(const double *xi, const double *xj, const double *z, double *Ji,
                                             double *Jj, double *err) {
    double si = sin(xi[2]), ci = cos(xi[2]);
    double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
    if (err) {
        err[0] = ci * dx + si * dy - z[0];
        err[1] = -si * dx + ci * dy - z[1];
        err[2] = normalize_theta(normalize_theta(xj[2] - xi[2]) - z[2]);
    }
    Ji[0] = -ci; Ji[1] = -si; Ji[2] = -si * dx + ci * dy;
    Ji[3] = si;  Ji[4] = -ci; Ji[5] = -ci * dx - si * dy;
    Ji[6] = 0;   Ji[7] = 0;   Ji[8] = -1;
    Jj[0] = ci;  Jj[1] = si;  Jj[2] = 0;
    Jj[3] = -si; Jj[4] = ci;  Jj[5] = 0;
    Jj[6] = 0;   Jj[7] = 0;   Jj[8] = 1;
}

}  // namespace spgdev
