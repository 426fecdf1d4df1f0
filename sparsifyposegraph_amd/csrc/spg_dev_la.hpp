// csrc/spg_dev_la.hpp — workgroup-cooperative dense fp64 linear algebra on LDS / L2-resident tiles.
//
// One workgroup (NT = 64: a single wavefront, or 256 for large blankets) owns one Markov blanket.
// All routines are written as strided loops over the NT lanes with workgroup barriers between
// dependent phases, so the same code serves a 12x12 SE2 block in LDS and a 132x132 SE3 tile in the
// global workspace. Matrices are row-major with an odd leading dimension (bank-conflict-free column
// walks on the 64-bank LDS). These replace the Eigen calls of the reference:
//   LLT / LLT::solve(I)            src/vertex_remover.cpp:444-447, src/pseudo_chow_liu.cpp:189-190,
//                                  src/logdet_function.cpp:246-247,273-274
//   LDLT::vectorD().log().sum()    src/pseudo_chow_liu.cpp:178-182, src/logdet_function.cpp:123-127
//   SelfAdjointEigenSolver         src/logdet_function.cpp:19, src/topology_provider_glc.cpp:45,66
//   PartialPivLU                   src/topology_provider_glc.cpp:63-64
#pragma once
#include <hip/hip_runtime.h>

namespace spgdev {

template <int NT>
struct Team {
    static constexpr int size = NT;
    int tid;
    double *red;  // NT doubles of LDS for reductions
    int *flag;    // one int of LDS (sticky failure / broadcast)
    // NT == 64: the team is one wavefront; its LDS operations execute in program order, so a "barrier"
    // only has to stop the compiler from moving memory operations across it (no s_barrier is issued)
    __device__ __forceinline__ void sync() const {
        if (NT == 64) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    }

    // deterministic sum of one value per lane (fixed butterfly order), result broadcast to all lanes
    __device__ __forceinline__ double sum(double v) const {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (NT == 64) return v;
        if ((tid & 63) == 0) red[tid >> 6] = v;
        sync();
        double s = 0;
#pragma unroll
        for (int i = 0; i < NT / 64; i++) s += red[i];
        sync();
        return s;
    }
};

// 1/sqrt(x) and 1/x from the hardware seed (v_rsq_f64 / v_rcp_f64) + two Newton steps: fp64 division
// and sqrt expand to ~40 dependent instructions each on gfx950, and a single wavefront cannot hide
// that chain; the seeds are good to ~2^-26, two quadratic steps reach rounding level.
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}

// sum of logs as the log of a running product with the exponent carried separately (one log per
// factorisation instead of one per pivot)
struct LogProd {
    double m = 1.0;
    int e = 0;
    __device__ __forceinline__ void mul(double x) {
        int ex;
        double mx = frexp(x, &ex);
        m *= mx; e += ex;
        int e2;
        m = frexp(m, &e2);
        e += e2;
    }
    __device__ __forceinline__ double value() const { return log(m) + (double)e * 0.6931471805599453; }
};

__device__ __forceinline__ int ceil_log2(int n) {
    int s = 0;
    while ((1 << s) < n) s++;
    return s;
}

// In-place lower Cholesky (right-looking). Reads/writes the lower triangle only. On failure sets
// *T.flag = 1 (checked by the caller after the call). Two barriers per column: every lane derives
// 1/sqrt(pivot) itself, lane 0 alone stores the pivot. If rdiag != nullptr it receives 1/L_jj.
template <int NT>
__device__ void chol_lower(const Team<NT> T, double *A, int n, int ld, double *rdiag = nullptr) {
    for (int j = 0; j < n; j++) {
        // every lane reads the pivot (ordered after the previous column's trailing update by barrier #2)
        double d = A[j * ld + j];
        bool bad = !(d > 0.0) || !isfinite(d);
        if (bad) d = 1.0;
        double rs = fast_rsqrt(d);
        for (int i = j + 1 + T.tid; i < n; i += NT) A[i * ld + j] *= rs;
        T.sync();  // #1: column j scaled, pivot consumed by everyone
        if (T.tid == 0) {
            if (bad) *T.flag = 1;
            A[j * ld + j] = d * rs;
            if (rdiag) rdiag[j] = rs;
        }
        // trailing update: rows i > j, cols j < c <= i
        int m = n - j - 1;
        int sh = ceil_log2(m > 0 ? m : 1);
        int tot = m << sh;
        for (int it = T.tid; it < tot; it += NT) {
            int r = it >> sh, c = it & ((1 << sh) - 1);
            if (c <= r && c < m) {
                int i = j + 1 + r, cc = j + 1 + c;
                A[i * ld + cc] -= A[i * ld + j] * A[cc * ld + j];
            }
        }
        T.sync();  // #2
    }
}

// 2 * sum(log L_ii) of a Cholesky factor
template <int NT>
__device__ double chol_logdet(const Team<NT> T, const double *L, int n, int ld) {
    // one lane walks the diagonal (n <= a few hundred): a single log of the running product
    double v = 0;
    if (T.tid == 0) {
        LogProd lp;
        for (int i = 0; i < n; i++) lp.mul(L[i * ld + i]);
        v = 2.0 * lp.value();
    }
    return T.sum(v);
}

// Linv = L^-1 (lower), one lane per column; Linv may not alias L. Strict upper of Linv is zeroed.
template <int NT>
__device__ void tri_inverse_lower(const Team<NT> T, const double *L, double *Li, int n, int ld, const double *rdiag = nullptr) {
    for (int c = T.tid; c < n; c += NT) {
        for (int i = 0; i < c; i++) Li[i * ld + c] = 0.0;
        Li[c * ld + c] = rdiag ? rdiag[c] : fast_rcp(L[c * ld + c]);
        for (int i = c + 1; i < n; i++) {
            double s = 0;
            for (int k = c; k < i; k++) s += L[i * ld + k] * Li[k * ld + c];
            Li[i * ld + c] = -s * (rdiag ? rdiag[i] : fast_rcp(L[i * ld + i]));
        }
    }
    T.sync();
}

// Out = Li^T Li (full symmetric), i.e. (L L^T)^-1 from Li = L^-1. Out may not alias Li.
template <int NT>
__device__ void gram_lower_inverse(const Team<NT> T, const double *Li, double *Out, int n, int ld) {
    int sh = ceil_log2(n);
    int tot = n << sh;
    for (int it = T.tid; it < tot; it += NT) {
        int i = it >> sh, j = it & ((1 << sh) - 1);
        if (j <= i) {
            double s = 0;
            for (int k = i; k < n; k++) s += Li[k * ld + i] * Li[k * ld + j];
            Out[i * ld + j] = s;
            Out[j * ld + i] = s;
        }
    }
    T.sync();
}

// Y <- L^-1 Y for a (n x m) right-hand side stored row-major with leading dimension ldy; lane per column.
template <int NT>
__device__ void tri_solve_lower(const Team<NT> T, const double *L, int n, int ld, double *Y, int m, int ldy) {
    for (int c = T.tid; c < m; c += NT) {
        for (int i = 0; i < n; i++) {
            double s = Y[i * ldy + c];
            for (int k = 0; k < i; k++) s -= L[i * ld + k] * Y[k * ldy + c];
            Y[i * ldy + c] = s * fast_rcp(L[i * ld + i]);
        }
    }
    T.sync();
}

// strict upper -> strict lower
template <int NT>
__device__ void mirror_upper(const Team<NT> T, double *A, int n, int ld) {
    int sh = ceil_log2(n);
    int tot = n << sh;
    for (int it = T.tid; it < tot; it += NT) {
        int i = it >> sh, j = it & ((1 << sh) - 1);
        if (j > i && j < n) A[j * ld + i] = A[i * ld + j];
    }
    T.sync();
}

// Round-robin pairing for parallel Jacobi: n2 even players, step s in [0, n2-1), pair index pi in [0, n2/2)
__device__ __forceinline__ void rr_pair(int s, int pi, int n2, int &p, int &q) {
    int m = n2 - 1;
    if (pi == 0) { p = m; q = s; }
    else {
        p = s + pi; if (p >= m) p -= m;
        q = s - pi; if (q < 0) q += m;
    }
    if (p > q) { int t = p; p = q; q = t; }
}

// Symmetric eigendecomposition of a LARGE matrix in the L2 workspace (clusters of several hundred vertices, where the
// Jacobi sweeps below take tens of seconds): Householder tridiagonalisation, Q formed explicitly, implicit-shift QL on
// the tridiagonal matrix with the rotations applied to the columns of Q (the tred2 / tql2 pair, arranged for a workgroup).
// A (n x n, full symmetric, destroyed: on exit its diagonal holds the eigenvalues, unordered), V receives the
// eigenvectors in columns — the interface of jacobi_eigh. gl: 3 n doubles of global scratch (d, e, beta). lds: 6 n + 8
// doubles. Returns false (uniformly) if an eigenvalue does not converge in 60 iterations. n <= 8 * NT.
// One workgroup streaming matrices out of L2 is bound by the memory round trips it waits for, so every pass keeps many
// independent accesses in flight per lane:
//   * column sums (A v = A^T v, v^T V): wavefront w takes the rows i = w (mod 4), a lane up to 16 columns 64 apart, two
//     rows per step — 32 loads in flight, no reduction across lanes; the four partial vectors meet in LDS;
//   * rank-1 / rank-2 updates: the same tiling, loads of a step issued before its stores;
//   * QL: wavefront 0 runs the scalar recurrence of a sweep (d, e in LDS) and leaves the rotations in LDS; the other
//     wavefronts apply the previous sweep's rotations to Q^T meanwhile — rows side by side, eight rotations and four entries
//     per lane per step (a rotation of columns i, i + 1 of Q treats the rows independently). The first form — rotations
//     applied to Q as they were produced, one load and one store per rotation, a lane per row — waited a full memory round
//     trip per rotation and touched 64 cache lines per access: 0.9 s of 1.4 s at n = 792.
template <int NT>
__device__ bool tridiag_eigh(const Team<NT> T, double *A, double *V, int n, int ld, double *gl, double *lds) {
    constexpr int NW = NT / 64, CMAX = 16;
    const int tid = T.tid, lane = tid & 63, wv = tid >> 6;
    double *dg = gl, *eg = gl + n, *bg = gl + 2 * n;
    if (n == 1) { if (tid == 0) V[0] = 1.0; T.sync(); return true; }
    double *vb = lds, *wb = lds + n, *part = lds + 2 * n;      // Householder vector, p / w / u, NW partial column sums
#ifdef SPG_CF_PROF
    long long tq0 = wall_clock64(), tq1, tq2, tq3;
#endif
    // out[j] = scale * sum_i M[i][j] vec[i] over the m x m block at M (leading dimension ld)
    auto colsum = [&](const double *M, int m, const double *vec, double *out, double scale) {
        for (int c0 = 0; c0 < m; c0 += 64 * CMAX) {
            double acc[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; c++) acc[c] = 0.0;
            for (int i = wv; i < m; i += 2 * NW) {
                const int i2 = i + NW;
                const double *r1 = M + (long long)i * ld + c0 + lane, *r2 = M + (long long)i2 * ld + c0 + lane;
                const double v1 = vec[i], v2 = (i2 < m) ? vec[i2] : 0.0;
                double x1[CMAX], x2[CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; c++) {
                    const bool in = c0 + lane + 64 * c < m;
                    x1[c] = in ? r1[64 * c] : 0.0;
                    x2[c] = (in && i2 < m) ? r2[64 * c] : 0.0;
                }
#pragma unroll
                for (int c = 0; c < CMAX; c++) acc[c] += x1[c] * v1 + x2[c] * v2;
            }
#pragma unroll
            for (int c = 0; c < CMAX; c++) { const int j = c0 + lane + 64 * c; if (j < m) part[wv * n + j] = acc[c]; }
        }
        T.sync();
        for (int j = tid; j < m; j += NT) {
            double sacc = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) sacc += part[w2 * n + j];
            out[j] = scale * sacc;
        }
        T.sync();
    };
    // M[i][j] -= a[i] b[j] (+ b[i] a[j] when sym: products rounded separately, the update stays symmetric bit for bit)
    auto rank_update = [&](double *M, int m, const double *av, const double *bv, bool sym) {
        for (int c0 = 0; c0 < m; c0 += 64 * CMAX)
            for (int i = wv; i < m; i += 2 * NW) {
                const int i2 = i + NW;
                double *r1 = M + (long long)i * ld + c0 + lane, *r2 = M + (long long)i2 * ld + c0 + lane;
                const double a1 = av[i], b1 = bv[i], a2 = (i2 < m) ? av[i2] : 0.0, b2 = (i2 < m) ? bv[i2] : 0.0;
                double x1[CMAX], x2[CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; c++) {
                    const bool in = c0 + lane + 64 * c < m;
                    x1[c] = in ? r1[64 * c] : 0.0;
                    x2[c] = (in && i2 < m) ? r2[64 * c] : 0.0;
                }
#pragma unroll
                for (int c = 0; c < CMAX; c++) {
                    const int j = c0 + lane + 64 * c;
                    if (j < m) {
                        const double aj = av[j], bj = bv[j];
                        r1[64 * c] = x1[c] - (sym ? __dadd_rn(__dmul_rn(a1, bj), __dmul_rn(b1, aj)) : a1 * bj);
                        if (i2 < m) r2[64 * c] = x2[c] - (sym ? __dadd_rn(__dmul_rn(a2, bj), __dmul_rn(b2, aj)) : a2 * bj);
                    }
                }
            }
        T.sync();
    };
    // ---------------------------------------------------------------- tridiagonalisation
    for (int k = 0; k + 2 < n; k++) {
        const int m = n - k - 1;
        const double *x = A + (long long)k * ld + k + 1;      // row k right of the diagonal = column k below it
        double sg = 0;
        for (int j = 1 + tid; j < m; j += NT) sg += x[j] * x[j];
        const double sigma = T.sum(sg), x0 = x[0];
        double beta = 0.0, mu = x0, v0 = 1.0;
        if (sigma > 0.0) {
            mu = sqrt(x0 * x0 + sigma);
            v0 = (x0 <= 0.0) ? x0 - mu : -sigma / (x0 + mu);
            beta = 2.0 * v0 * v0 / (sigma + v0 * v0);
        }
        for (int j = tid; j < m; j += NT) vb[j] = (j == 0) ? 1.0 : x[j] / v0;
        T.sync();
        if (tid == 0) { dg[k] = A[(long long)k * ld + k]; eg[k] = (sigma > 0.0) ? mu : x0; bg[k] = beta; }
        for (int j = 1 + tid; j < m; j += NT) A[(long long)k * ld + k + 1 + j] = vb[j];      // kept for Q
        if (beta != 0.0) {
            double *A22 = A + (long long)(k + 1) * ld + k + 1;
            colsum(A22, m, vb, wb, beta);                      // p = beta A22 v
            double pv = 0;
            for (int j = tid; j < m; j += NT) pv += wb[j] * vb[j];
            const double alpha = 0.5 * beta * T.sum(pv);
            for (int j = tid; j < m; j += NT) wb[j] -= alpha * vb[j];
            T.sync();
            rank_update(A22, m, vb, wb, true);                 // A22 -= v w^T + w v^T
        } else T.sync();
    }
    if (tid == 0) {
        dg[n - 2] = A[(long long)(n - 2) * ld + n - 2];
        dg[n - 1] = A[(long long)(n - 1) * ld + n - 1];
        eg[n - 2] = A[(long long)(n - 2) * ld + n - 1];
        eg[n - 1] = 0.0;
    }
#ifdef SPG_CF_PROF
    T.sync(); tq1 = wall_clock64();
#endif
    // ---------------------------------------------------------------- V = Q = H_0 H_1 ... H_{n-3}
    for (long long it = tid; it < (long long)n * n; it += NT) { const int i = (int)(it / n), j = (int)(it - (long long)i * n); V[(long long)i * ld + j] = (i == j) ? 1.0 : 0.0; }
    T.sync();
    for (int k = n - 3; k >= 0; k--) {
        const int m = n - k - 1;
        const double beta = bg[k];
        if (beta == 0.0) continue;
        for (int j = tid; j < m; j += NT) vb[j] = (j == 0) ? 1.0 : A[(long long)k * ld + k + 1 + j];
        T.sync();
        double *V22 = V + (long long)(k + 1) * ld + k + 1;
        colsum(V22, m, vb, wb, beta);                          // u = beta v^T V22
        rank_update(V22, m, vb, wb, false);                    // V22 -= v u^T
    }
#ifdef SPG_CF_PROF
    T.sync(); tq2 = wall_clock64();
#endif
    // Q^T into A (the Householder vectors are spent): a rotation of two COLUMNS of V is then a rotation of two ROWS of
    // A, which the lanes read and write side by side — with V itself a lane would own a row and every access of the
    // wavefront would touch 64 cache lines (measured: 1 us per rotation, three quarters of the whole decomposition)
    for (long long it = tid; it < (long long)n * n; it += NT) { const int rr = (int)(it / n), c = (int)(it - (long long)rr * n); A[(long long)c * ld + rr] = V[(long long)rr * ld + c]; }
    T.sync();
    // ---------------------------------------------------------------- implicit QL (tql2), two stages in a pipeline
    // Stage A (wavefront 0): the scalar recurrence of the next sweep — d, e and the sweep's rotations in LDS. Stage B (the other
    // wavefronts): the previous sweep's rotations applied to Q^T. A sweep's rotations depend on d, e only, never on Q, so the two
    // overlap; the rotation buffers alternate. (One after the other they took 198 + 145 ms at n = 792.)
    double *d = lds, *e = lds + n;
    double *csb2[2] = {lds + 2 * n, lds + 4 * n}, *snb2[2] = {lds + 3 * n, lds + 5 * n};
    int *ctl = reinterpret_cast<int *>(lds + 6 * n);       // per buffer: [0] m, [1] lowest rotation index, [2] state: 0 sweep, 1 all done, 2 failed
    for (int i = tid; i < n; i += NT) { d[i] = dg[i]; e[i] = eg[i]; }
    T.sync();
    constexpr int RC = 4, U8 = 8;                    // entries of a row of Q^T per lane and pass, rotations per step
    constexpr int NTB = NT - 64;                     // lanes of stage B
    const int tb = tid - 64;                         // stage B's lane index
    const int nrows = tb >= 0 ? (n - tb + NTB - 1) / NTB : 0;     // entries tb, tb + NTB, ...
    bool ok = true;
#ifdef SPG_CF_PROF
    long long nsweep = 0, nrot = 0;
#endif
    int l = 0, iter = 0, cur = 0;     // (stage A's state, kept by every lane of wavefront 0)
    bool pending = false;             // a sweep in buffer cur ^ 1 waits for stage B
    int pm = 0, plo = 0;
    for (;;) {
        if (wv == 0) {
            // ---- stage A: advance to the next sweep (or to the end) and leave it in buffer `cur`
            int state = 1, m = 0, lo = 0;
            double *csb = csb2[cur], *snb = snb2[cur];
            while (l < n) {
                m = l;
                for (; m < n - 1; m++) {
                    const double dd = fabs(d[m]) + fabs(d[m + 1]);
                    if (fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
                }
                if (m == l) { l++; iter = 0; continue; }           // eigenvalue l has converged
                if (iter++ == 60) { state = 2; break; }
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = sqrt(g * g + 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? r : -r));
                double sn = 1.0, cs = 1.0, pp = 0.0;
                bool underflow = false;
                int i = m - 1;
                // One wavefront alone: nothing hides the latency of a dependent operation, so the chain is kept short — the
                // next step's d, e are read a step ahead, 1 / r comes from the reciprocal square root (hardware seed + two Newton
                // steps; no division, no sqrt sequence), and all lanes store (same value, same address) instead of branching.
                double e_i = e[i], d_i = d[i], d_i1 = d[i + 1];
                for (; i >= l; i--) {
                    const double e_n = (i > l) ? e[i - 1] : 0.0, d_n = (i > l) ? d[i - 1] : 0.0;
                    const double f = sn * e_i, b = cs * e_i;
                    const double h2 = f * f + g * g;
                    if (h2 == 0.0) { e[i + 1] = 0.0; d[i + 1] = d_i1 - pp; e[m] = 0.0; underflow = true; break; }
                    const double rinv = fast_rsqrt(h2);
                    r = h2 * rinv;
                    e[i + 1] = r;
                    sn = f * rinv; cs = g * rinv;
                    g = d_i1 - pp;
                    r = (d_i - g) * sn + 2.0 * cs * b;
                    pp = sn * r;
                    d[i + 1] = g + pp;
                    csb[i] = cs; snb[i] = sn;
                    g = cs * r - b;
                    // (d[i + 1] as the next step sees it is d[i] of this one, untouched so far in this sweep)
                    d_i1 = d_i; d_i = d_n; e_i = e_n;
                }
                lo = i + 1;
                if (!underflow) { d[l] -= pp; e[l] = g; e[m] = 0.0; }
                state = 0;
                break;
            }
            if (lane == 0) { ctl[4 * cur] = m; ctl[4 * cur + 1] = lo; ctl[4 * cur + 2] = state; }
        } else if (pending && pm - 1 >= plo) {
            // ---- stage B: rotations i = m - 1 .. lo of the previous sweep on rows (i, i + 1) of Q^T, for the entries of this
            // lane: four entries at a time, eight rotations per step, the next step's loads issued before this step's stores
            // (memory operations complete in order: loads behind stores would wait for the stores too)
            const int m = pm, lo = plo;
            const double *csb = csb2[cur ^ 1], *snb = snb2[cur ^ 1];
            for (int t0 = 0; t0 < nrows; t0 += RC) {
                double zc[RC];
                int col[RC];
                bool live[RC];
#pragma unroll
                for (int t = 0; t < RC; t++) { live[t] = t0 + t < nrows; col[t] = tb + (t0 + t) * NTB; zc[t] = live[t] ? A[(long long)m * ld + col[t]] : 0.0; }
                double ba[RC][U8], bb[RC][U8];
                auto load = [&](int i0, double (&buf)[RC][U8]) {
#pragma unroll
                    for (int t = 0; t < RC; t++)
#pragma unroll
                        for (int u = 0; u < U8; u++) buf[t][u] = (live[t] && i0 - u >= lo) ? A[(long long)(i0 - u) * ld + col[t]] : 0.0;
                };
                auto apply = [&](int i0, const double (&buf)[RC][U8]) {
#pragma unroll
                    for (int u = 0; u < U8; u++) {
                        const int i = i0 - u;
                        if (i >= lo) {
                            const double cs = csb[i], sn = snb[i];
#pragma unroll
                            for (int t = 0; t < RC; t++)
                                if (live[t]) {
                                    A[(long long)(i + 1) * ld + col[t]] = sn * buf[t][u] + cs * zc[t];
                                    zc[t] = cs * buf[t][u] - sn * zc[t];
                                }
                        }
                    }
                };
                int i0 = m - 1;
                load(i0, ba);
                while (i0 >= lo) {
                    if (i0 - U8 >= lo) load(i0 - U8, bb);
                    apply(i0, ba);
                    i0 -= U8;
                    if (i0 < lo) break;
                    if (i0 - U8 >= lo) load(i0 - U8, ba);
                    apply(i0, bb);
                    i0 -= U8;
                }
#pragma unroll
                for (int t = 0; t < RC; t++) if (live[t]) A[(long long)lo * ld + col[t]] = zc[t];
            }
        }
        T.sync();
        // the sweep stage A has just produced becomes stage B's next job; the one stage B has just applied is done
        const int state = ctl[4 * cur + 2];
        pending = state == 0;
        pm = ctl[4 * cur]; plo = ctl[4 * cur + 1];
#ifdef SPG_CF_PROF
        if (pending) { nsweep++; nrot += pm - plo; }
#endif
        if (state == 2) ok = false;
        if (state != 0) break;              // (nothing pending any more: the last sweep was applied in this very round)
        cur ^= 1;                           // (stage A writes the other buffer and the other ctl slot next: no second barrier needed)
    }
    T.sync();
#ifdef SPG_CF_PROF
    tq3 = wall_clock64();
    if (tid == 0) printf("tridiag_eigh n=%d (us): tridiagonalise %lld, form Q %lld, QL %lld (%lld sweeps, %lld rotations)\n", n, (tq1 - tq0) / 100, (tq2 - tq1) / 100, (tq3 - tq2) / 100, nsweep, nrot);
#endif
    for (long long it = tid; it < (long long)n * n; it += NT) { const int rr = (int)(it / n), c = (int)(it - (long long)rr * n); V[(long long)rr * ld + c] = A[(long long)c * ld + rr]; }
    T.sync();
    for (int i = tid; i < n; i += NT) A[(long long)i * ld + i] = d[i];
    T.sync();
    return ok;
}

// Symmetric eigendecomposition by parallel-order two-sided Jacobi. A (n x n, full, destroyed: on
// exit its diagonal holds the eigenvalues), V receives the eigenvectors in columns. cs: 2*(n/2+1)
// doubles of scratch. Returns false (uniformly) if not converged in max_sweeps.
template <int NT>
__device__ bool jacobi_eigh(const Team<NT> T, double *A, double *V, int n, int ld, double *cs, int max_sweeps = 60) {
    int sh = ceil_log2(n);
    int tot = n << sh;
    double f = 0;
    for (int it = T.tid; it < tot; it += NT) {
        int i = it >> sh, j = it & ((1 << sh) - 1);
        if (j < n) {
            V[i * ld + j] = (i == j) ? 1.0 : 0.0;
            double a = A[i * ld + j];
            f += a * a;
        }
    }
    double fro2 = T.sum(f);
    if (n <= 1 || fro2 == 0.0) return true;
    int np = (n + 1) >> 1, n2 = np * 2;
    int shp = ceil_log2(n);
    bool converged = false;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        double o = 0;
        for (int it = T.tid; it < tot; it += NT) {
            int i = it >> sh, j = it & ((1 << sh) - 1);
            if (j < n && i != j) { double a = A[i * ld + j]; o += a * a; }
        }
        double off2 = T.sum(o);
        if (off2 <= 1e-31 * fro2) { converged = true; break; }
        for (int s = 0; s < n2 - 1; s++) {
            for (int pi = T.tid; pi < np; pi += NT) {
                int p, q;
                rr_pair(s, pi, n2, p, q);
                double c = 1.0, sn = 0.0;
                if (q < n) {
                    double apq = A[p * ld + q];
                    if (fabs(apq) > 1e-300) {
                        double tau = (A[q * ld + q] - A[p * ld + p]) / (2.0 * apq);
                        double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        sn = t * c;
                    }
                }
                cs[2 * pi] = c; cs[2 * pi + 1] = sn;
            }
            T.sync();
            // column pass on A and V
            int totc = np << shp;
            for (int it = T.tid; it < totc; it += NT) {
                int pi = it >> shp, i = it & ((1 << shp) - 1);
                if (i >= n) continue;
                int p, q;
                rr_pair(s, pi, n2, p, q);
                if (q >= n) continue;
                double c = cs[2 * pi], sn = cs[2 * pi + 1];
                if (sn == 0.0) continue;
                double aip = A[i * ld + p], aiq = A[i * ld + q];
                A[i * ld + p] = c * aip - sn * aiq;
                A[i * ld + q] = sn * aip + c * aiq;
                double vip = V[i * ld + p], viq = V[i * ld + q];
                V[i * ld + p] = c * vip - sn * viq;
                V[i * ld + q] = sn * vip + c * viq;
            }
            T.sync();
            // row pass on A
            for (int it = T.tid; it < totc; it += NT) {
                int pi = it >> shp, j = it & ((1 << shp) - 1);
                if (j >= n) continue;
                int p, q;
                rr_pair(s, pi, n2, p, q);
                if (q >= n) continue;
                double c = cs[2 * pi], sn = cs[2 * pi + 1];
                if (sn == 0.0) continue;
                double apj = A[p * ld + j], aqj = A[q * ld + j];
                double np_ = c * apj - sn * aqj, nq_ = sn * apj + c * aqj;
                if (j == q) np_ = 0.0;
                if (j == p) nq_ = 0.0;
                A[p * ld + j] = np_;
                A[q * ld + j] = nq_;
            }
            T.sync();
        }
    }
    return converged;
}

// rank-by-counting sort of n keys (ascending, ties by index): perm[rank] = index
template <int NT>
__device__ void sort_ascending(const Team<NT> T, const double *key, int stride, int n, int *perm) {
    for (int i = T.tid; i < n; i += NT) {
        double ki = key[i * stride];
        int r = 0;
        for (int j = 0; j < n; j++) {
            double kj = key[j * stride];
            r += (kj < ki) || (kj == ki && j < i);
        }
        perm[r] = i;
    }
    T.sync();
}

// Small fixed-size helpers evaluated by ONE lane in registers (D = 3 or 6) --------------------
template <int D>
__device__ __forceinline__ bool chol_reg(double *a /*D*D row-major, lower used, in place*/) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < D; j++) {
        double d = a[j * D + j];
#pragma unroll
        for (int k = 0; k < D; k++) if (k < j) d -= a[j * D + k] * a[j * D + k];
        if (!(d > 0.0) || !isfinite(d)) { ok = false; d = 1.0; }
        double inv = fast_rsqrt(d);
        a[j * D + j] = d * inv;
#pragma unroll
        for (int i = 0; i < D; i++) if (i > j) {
            double s = a[i * D + j];
#pragma unroll
            for (int k = 0; k < D; k++) if (k < j) s -= a[i * D + k] * a[j * D + k];
            a[i * D + j] = s * inv;
        }
    }
    return ok;
}

// X = (L L^T)^-1 from the lower factor in a[] (registers); writes full symmetric X
template <int D>
__device__ __forceinline__ void chol_inverse_reg(const double *L, double *X) {
    double Li[D * D];
#pragma unroll
    for (int c = 0; c < D; c++) {
#pragma unroll
        for (int i = 0; i < D; i++) {
            if (i < c) Li[i * D + c] = 0.0;
            else if (i == c) Li[i * D + c] = fast_rcp(L[c * D + c]);
            else {
                double s = 0;
#pragma unroll
                for (int k = 0; k < D; k++) if (k >= c && k < i) s += L[i * D + k] * Li[k * D + c];
                Li[i * D + c] = -s * fast_rcp(L[i * D + i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = 0; j < D; j++) if (j <= i) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < D; k++) if (k >= i) s += Li[k * D + i] * Li[k * D + j];
            X[i * D + j] = s;
            X[j * D + i] = s;
        }
}

// partial-pivot Gauss-Jordan inverse of a D x D block in registers (Eigen PartialPivLU::inverse on the full J reduces to
// these blocks for a block-arrow matrix)
template <int D>
__device__ __forceinline__ bool small_inverse(const double *A, double *X) {
    double a[D * D], x[D * D];
#pragma unroll
    for (int i = 0; i < D * D; i++) { a[i] = A[i]; x[i] = ((i / D) == (i % D)) ? 1.0 : 0.0; }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < D; c++) {
        int p = c;
        double best = fabs(a[c * D + c]);
#pragma unroll
        for (int r = 0; r < D; r++) if (r > c && fabs(a[r * D + c]) > best) { best = fabs(a[r * D + c]); p = r; }
        if (!(best > 0.0)) ok = false;
#pragma unroll
        for (int r = 0; r < D; r++) if (r == p && p != c) {
#pragma unroll
            for (int j = 0; j < D; j++) { double t = a[c * D + j]; a[c * D + j] = a[r * D + j]; a[r * D + j] = t; t = x[c * D + j]; x[c * D + j] = x[r * D + j]; x[r * D + j] = t; }
        }
        double ip = 1.0 / a[c * D + c];
#pragma unroll
        for (int j = 0; j < D; j++) { a[c * D + j] *= ip; x[c * D + j] *= ip; }
#pragma unroll
        for (int r = 0; r < D; r++) if (r != c) {
            double f = a[r * D + c];
#pragma unroll
            for (int j = 0; j < D; j++) { a[r * D + j] -= f * a[c * D + j]; x[r * D + j] -= f * x[c * D + j]; }
        }
    }
#pragma unroll
    for (int i = 0; i < D * D; i++) X[i] = x[i];
    return ok;
}

}  // namespace spgdev

// ============================================================================================
// Batched variants: `cnt` independent s x s matrices (matrix b at base + b*stride, leading dim ld)
// advanced in lockstep by the whole team. Used by the GLC tail, where one blanket produces k-1
// equally sized 2d x 2d problems (src/topology_provider_glc.cpp:143-182).
namespace spgdev {

// In-place inverse by Gauss-Jordan with partial pivoting (the reference goes through
// Eigen::PartialPivLU(J).solve(I), src/topology_provider_glc.cpp:63-64). scratch: cnt*(2*s+2) doubles.
// Sets *T.flag on a zero / non-finite pivot.
template <int NT>
__device__ void gj_inverse_batch(const Team<NT> T, double *A, int cnt, int s, int ld, int stride, double *scratch) {
    double *fcol = scratch;                       // cnt * s
    int *ipiv = reinterpret_cast<int *>(scratch + (size_t)cnt * s);  // cnt * s ints (fits in cnt*s doubles)
    int sh = ceil_log2(s);
    for (int j = 0; j < s; j++) {
        for (int b = T.tid; b < cnt; b += NT) {
            double *M = A + (size_t)b * stride;
            int p = j;
            double best = fabs(M[j * ld + j]);
            for (int i = j + 1; i < s; i++) { double v = fabs(M[i * ld + j]); if (v > best) { best = v; p = i; } }
            if (!(best > 0.0) || !isfinite(best)) *T.flag = 1;
            ipiv[b * s + j] = p;
        }
        T.sync();
        for (int it = T.tid; it < (cnt << sh); it += NT) {
            int b = it >> sh, c = it & ((1 << sh) - 1);
            if (c < s) {
                double *M = A + (size_t)b * stride;
                int p = ipiv[b * s + j];
                if (p != j) { double t = M[j * ld + c]; M[j * ld + c] = M[p * ld + c]; M[p * ld + c] = t; }
            }
        }
        T.sync();
        for (int it = T.tid; it < (cnt << sh); it += NT) {
            int b = it >> sh, i = it & ((1 << sh) - 1);
            if (i < s) {
                double *M = A + (size_t)b * stride;
                fcol[b * s + i] = (i == j) ? (1.0 / M[j * ld + j]) : M[i * ld + j];
            }
        }
        T.sync();
        // scale pivot row (with the unit substitution), clear column j elsewhere
        for (int it = T.tid; it < (cnt << sh); it += NT) {
            int b = it >> sh, c = it & ((1 << sh) - 1);
            if (c < s) {
                double *M = A + (size_t)b * stride;
                double pv = fcol[b * s + j];
                M[j * ld + c] = ((c == j) ? 1.0 : M[j * ld + c]) * pv;
            }
        }
        T.sync();
        for (int it = T.tid; it < ((cnt * s) << sh); it += NT) {
            int bi = it >> sh, c = it & ((1 << sh) - 1);
            int b = bi / s, i = bi - b * s;
            if (c < s && i != j) {
                double *M = A + (size_t)b * stride;
                double f = fcol[b * s + i];
                double base = (c == j) ? 0.0 : M[i * ld + c];
                M[i * ld + c] = base - f * M[j * ld + c];
            }
        }
        T.sync();
    }
    for (int j = s - 1; j >= 0; j--) {
        for (int it = T.tid; it < (cnt << sh); it += NT) {
            int b = it >> sh, i = it & ((1 << sh) - 1);
            if (i < s) {
                double *M = A + (size_t)b * stride;
                int p = ipiv[b * s + j];
                if (p != j) { double t = M[i * ld + j]; M[i * ld + j] = M[i * ld + p]; M[i * ld + p] = t; }
            }
        }
        T.sync();
    }
}

// Batched parallel-order Jacobi (see jacobi_eigh). cs: cnt*(s+4) doubles; done: cnt ints.
// On exit the diagonals hold the eigenvalues, V the eigenvectors. Sets *T.flag if any matrix fails
// to converge.
template <int NT>
__device__ void jacobi_batch(const Team<NT> T, double *A, double *V, int cnt, int s, int ld, int stride, double *cs,
                             int *done, int max_sweeps = 60) {
    int sh = ceil_log2(s);
    int np = (s + 1) >> 1, n2 = np * 2;
    for (int it = T.tid; it < ((cnt * s) << sh); it += NT) {
        int bi = it >> sh, j = it & ((1 << sh) - 1);
        int b = bi / s, i = bi - b * s;
        if (j < s) V[(size_t)b * stride + i * ld + j] = (i == j) ? 1.0 : 0.0;
    }
    for (int b = T.tid; b < cnt; b += NT) done[b] = (s <= 1) ? 1 : 0;
    T.sync();
    for (int sweep = 0; sweep <= max_sweeps; sweep++) {
        // convergence test per matrix (one lane each)
        for (int b = T.tid; b < cnt; b += NT) {
            if (done[b]) continue;
            const double *M = A + (size_t)b * stride;
            double off2 = 0, fro2 = 0;
            for (int i = 0; i < s; i++)
                for (int j = 0; j < s; j++) { double a = M[i * ld + j]; fro2 += a * a; if (i != j) off2 += a * a; }
            if (off2 <= 1e-31 * fro2 || fro2 == 0.0) done[b] = 1;
            else if (sweep == max_sweeps) { done[b] = 1; *T.flag = 1; }
        }
        T.sync();
        int alldone = 1;
        for (int b = 0; b < cnt; b++) alldone &= done[b];
        if (alldone) break;
        for (int st = 0; st < n2 - 1; st++) {
            for (int it = T.tid; it < cnt * np; it += NT) {
                int b = it / np, pi = it - b * np;
                double c = 1.0, sn = 0.0;
                if (!done[b]) {
                    const double *M = A + (size_t)b * stride;
                    int p, q;
                    rr_pair(st, pi, n2, p, q);
                    if (q < s) {
                        double apq = M[p * ld + q];
                        if (fabs(apq) > 1e-300) {
                            double tau = (M[q * ld + q] - M[p * ld + p]) / (2.0 * apq);
                            double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                            c = 1.0 / sqrt(1.0 + t * t);
                            sn = t * c;
                        }
                    }
                }
                cs[2 * it] = c; cs[2 * it + 1] = sn;
            }
            T.sync();
            for (int it = T.tid; it < ((cnt * np) << sh); it += NT) {
                int bp = it >> sh, i = it & ((1 << sh) - 1);
                if (i >= s) continue;
                int b = bp / np, pi = bp - b * np;
                double c = cs[2 * bp], sn = cs[2 * bp + 1];
                if (sn == 0.0) continue;
                int p, q;
                rr_pair(st, pi, n2, p, q);
                double *M = A + (size_t)b * stride, *W = V + (size_t)b * stride;
                double aip = M[i * ld + p], aiq = M[i * ld + q];
                M[i * ld + p] = c * aip - sn * aiq;
                M[i * ld + q] = sn * aip + c * aiq;
                double vip = W[i * ld + p], viq = W[i * ld + q];
                W[i * ld + p] = c * vip - sn * viq;
                W[i * ld + q] = sn * vip + c * viq;
            }
            T.sync();
            for (int it = T.tid; it < ((cnt * np) << sh); it += NT) {
                int bp = it >> sh, j = it & ((1 << sh) - 1);
                if (j >= s) continue;
                int b = bp / np, pi = bp - b * np;
                double c = cs[2 * bp], sn = cs[2 * bp + 1];
                if (sn == 0.0) continue;
                int p, q;
                rr_pair(st, pi, n2, p, q);
                double *M = A + (size_t)b * stride;
                double apj = M[p * ld + j], aqj = M[q * ld + j];
                double np_ = c * apj - sn * aqj, nq_ = sn * apj + c * aqj;
                if (j == q) np_ = 0.0;
                if (j == p) nq_ = 0.0;
                M[p * ld + j] = np_;
                M[q * ld + j] = nq_;
            }
            T.sync();
        }
    }
}

}  // namespace spgdev
