// csrc/spg_dev_wave.hpp — register-resident SPD kernels for ONE wavefront (n <= N <= 64).
//
// Lane i owns row i of the matrix in N fp64 registers; cross-lane operands travel through
// v_readlane (SGPR broadcast), so a factorisation runs without a single LDS round trip or barrier.
// This is what makes a 24x24 Cholesky cost ~7k cycles instead of ~44k on a lone wavefront (the
// LDS-cooperative versions in spg_dev_la.hpp pay one exposed ~100-cycle LDS latency per inner step;
// measured with tools/stamp_bench.py). Rows beyond the true size n hold identity rows, so three
// instantiations (N = 12, 24) serve every blanket up to k = 4 (SE3) / k = 8 (SE2).
//
// Same arithmetic as the Eigen calls they stand in for (LLT, LLT::solve(I), LDLT log-det):
// src/pseudo_chow_liu.cpp:189-190, src/logdet_function.cpp:123-127,246-247.
#pragma once
#include <hip/hip_runtime.h>
#include "spg_dev_la.hpp"

namespace spgdev {

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return r.d;
}

// a[c] = M[lane][c] for lane < n, c < n; identity elsewhere
template <int N>
__device__ __forceinline__ void wave_load_rows(double (&a)[N], const double *M, int ld, int n, int lane, double diag_shift = 0.0) {
#pragma unroll
    for (int c = 0; c < N; c++) {
        double v = (lane == c) ? 1.0 : 0.0;
        if (lane < n && c < n) v = M[lane * ld + c] + ((lane == c) ? diag_shift : 0.0);
        a[c] = v;
    }
}

template <int N>
__device__ __forceinline__ void wave_store_rows(const double (&a)[N], double *M, int ld, int n, int lane) {
#pragma unroll
    for (int c = 0; c < N; c++)
        if (lane < n && c < n) M[lane * ld + c] = a[c];
}

// In-place lower Cholesky, row per lane. On exit a[j] (j < lane) = L[lane][j] and the DIAGONAL holds
// 1/L_jj. Returns false if a pivot was not positive. logdet receives 2*sum(log L_jj).
template <int N>
__device__ __forceinline__ bool wave_chol(double (&a)[N], int lane, double &logdet) {
    bool ok = true;
    LogProd lp;
#pragma unroll
    for (int j = 0; j < N; j++) {
        double d = readlane_f64(a[j], j);
        if (!(d > 0.0) || !isfinite(d)) { ok = false; d = 1.0; }
        double rs = fast_rsqrt(d);
        lp.mul(rs);
        a[j] = (lane == j) ? rs : a[j] * rs;
#pragma unroll
        for (int c = j + 1; c < N; c++) {
            double lc = readlane_f64(a[j], c);
            a[c] -= a[j] * lc;
        }
    }
    logdet = -2.0 * lp.value();
    return ok;
}

// x[i] = (L^-1)[i][lane] from the factor produced by wave_chol (column per lane)
template <int N>
__device__ __forceinline__ void wave_lower_inverse(const double (&a)[N], double (&x)[N], int lane) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        double s = (lane == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) s -= readlane_f64(a[k], i) * x[k];
        x[i] = s * readlane_f64(a[i], i);
    }
}

// g[j] = (X^T X)[lane][j] for lower-triangular X held column per lane; returns sum_i x[i]^2 of this lane
template <int N>
__device__ __forceinline__ double wave_gram(const double (&x)[N], double (&g)[N]) {
    double sq = 0;
#pragma unroll
    for (int j = 0; j < N; j++) g[j] = 0.0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        sq += x[i] * x[i];
#pragma unroll
        for (int j = 0; j <= i; j++) g[j] += x[i] * readlane_f64(x[i], j);
    }
    return sq;
}


// In-place inverse of an SPD matrix by Gauss-Jordan elimination without pivoting (pivots of an SPD
// matrix are the positive LDL^T pivots), row per lane: per column one broadcast of the scaled pivot
// row (N readlanes) feeds one FMA per lane and entry — N^2 broadcast+FMA pairs in total and no value
// is needed twice, where Cholesky + triangular inverse + Gram needs ~1.5 N^2 and keeps the factor's
// broadcasts live. logdet = sum(log pivots). Returns false on a non-positive pivot.
template <int N>
__device__ __forceinline__ bool wave_gj_inverse(double (&a)[N], int lane, double &logdet) {
    bool ok = true;
    LogProd lp;
#pragma unroll
    for (int j = 0; j < N; j++) {
        double p = readlane_f64(a[j], j);
        if (!(p > 0.0) || !isfinite(p)) { ok = false; p = 1.0; }
        lp.mul(p);
        double ip = fast_rcp(p);
        const bool me = (lane == j);
        double scale = me ? ip : 1.0;
        double f = me ? 0.0 : a[j];
#pragma unroll
        for (int t = 0; t < N; t++) a[t] *= scale;      // lane j: row j / pivot
        a[j] = me ? ip : 0.0;
#pragma unroll
        for (int t = 0; t < N; t++) a[t] -= f * readlane_f64(a[t], j);
    }
    logdet = lp.value();
    return ok;
}

// (src + shift*I)^-1 -> dst (rows, may alias src), log det and trace of the inverse. One wavefront.
template <int N>
__device__ __forceinline__ bool wave_spd_inverse_n(const double *src, int ld, int n, int lane, double shift, double *dst,
                                                   double &logdet, double &trace_inv) {
    double a[N];
    wave_load_rows<N>(a, src, ld, n, lane, shift);
    bool ok = wave_gj_inverse<N>(a, lane, logdet);
    double dg = 0;
#pragma unroll
    for (int t = 0; t < N; t++) dg = (lane == t) ? a[t] : dg;
    if (lane >= n) dg = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dg += __shfl_xor(dg, o, 64);
    trace_inv = dg;
    wave_store_rows<N>(a, dst, ld, n, lane);
    return ok;
}

template <int N>
__device__ __forceinline__ bool wave_spd_logdet_n(const double *src, int ld, int n, int lane, double &logdet) {
    double a[N];
    wave_load_rows<N>(a, src, ld, n, lane, 0.0);
    return wave_chol<N>(a, lane, logdet);
}

__device__ __forceinline__ bool wave_spd_inverse(const double *src, int ld, int n, int lane, double shift, double *dst,
                                                 double &logdet, double &trace_inv) {
    if (n <= 12) return wave_spd_inverse_n<12>(src, ld, n, lane, shift, dst, logdet, trace_inv);
    return wave_spd_inverse_n<24>(src, ld, n, lane, shift, dst, logdet, trace_inv);
}

__device__ __forceinline__ bool wave_spd_logdet(const double *src, int ld, int n, int lane, double &logdet) {
    if (n <= 12) return wave_spd_logdet_n<12>(src, ld, n, lane, logdet);
    return wave_spd_logdet_n<24>(src, ld, n, lane, logdet);
}

// N = 36 would serve k <= 6 (SE3) too, but its 72 live fp64 registers push the whole kernel to 256
// VGPRs + scratch spills (rocprofv3: 8.3 MB of spill writes per 200-blanket launch); larger tiles use
// the LDS-cooperative routines instead.
constexpr int kWaveMax = 24;

}  // namespace spgdev
