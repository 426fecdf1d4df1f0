// csrc/spg_nfr_ip.hip — interior-point NFR on the device: the blankets whose sparsity pattern has
// no closed form — SparsityOptions::Dense and ::Subgraph with more than k-1 uncorrelated pose-pose edges.
//
// Reference: optimizeInformation's "dirty interior point" (src/optimizer.cpp:38-79) over
// LogdetFunctionWithConstraints (src/logdet_function.cpp:87-214,348-427) with PQNOptimizer::optimize in its
// useHessian form and LineSearchSimpleBacktracking (src/pqn/pqn_optimizer.cpp:29-126, src/pqn/line_search.cpp:12-37);
// pattern selection PseudoChowLiu::computeSparsityPattern (src/pseudo_chow_liu.cpp:33-87, doKruskal :253-289). The
// control flow is the reference's, quirks included (the test checker restates the same loop independently):
// 15 barrier weights rho = 1 ... 5.6e-8, per rho Newton steps H d = -g with the reference's Hessian (P (x) P without the
// 1/2 of the gradient), step halving until the value does not increase, three termination tests with tolerance 1e-4
// (1e-12 for the last rho).
//
// One workgroup (256 threads) per blanket, every matrix in a per-blanket slice of a global workspace (L2-resident): a
// blanket is ~50-100 dependent Newton steps whose largest object is the (d^2 E)^2 Hessian (k = 4, SE3, Dense: 216^2),
// i.e. a latency chain like the closed-form kernel's but ~1000 x longer; the blankets of a batch run side by side.
// The kernel does the whole blanket itself — gather, Hessian of the blanket's pose-pose edges, Schur complement,
// pattern, spectrum, interior point, records — and hands the result over exactly as blanket_kernel does (new edge
// records in the arena, out record + ready / final word in the mailbox or the arena).
// Scope: pose-pose edges in, Global linearisation point, the `smalleigs <= dim` branch of the spectrum
// (src/logdet_function.cpp:36-39); anything else reports a status and writes no edges.
#include <type_traits>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>

#include "spg_dev_geom.hpp"
#include "spg_dev_la.hpp"
#include "spg_internal.h"

using namespace spgdev;

namespace {

constexpr int NT = 256;
constexpr int RIF = 4;    // rows in flight per wavefront in the trailing update of the out-of-L2 Cholesky
#ifdef SPG_IP_PROF
#define IPT(k) do { long long t_ = (long long)__builtin_amdgcn_s_memtime(); ipt[k] += t_ - ipt_last; ipt_last = t_; } while (0)
#else
#define IPT(k) do { } while (0)
#endif
// phases of the closed-form path of a cluster (development builds with -DSPG_CF_PROF print them per blanket)
#ifdef SPG_CF_PROF
#define CFP(k) do { __syncthreads(); long long t_ = wall_clock64(); cfp[k] += t_ - cfp_last; cfp_last = t_; } while (0)
#else
#define CFP(k) do { } while (0)
#endif
constexpr int kIpLdsVector = 2048;     // Newton systems up to this size can keep their running vector in static LDS (paths A, A', B, B')
constexpr int kPanelDoubles = 8192;   // 64 KB: 16 columns of a panel up to 480 rows tall, fewer columns for taller ones

struct IpLayout {   // offsets (doubles) of one blanket's buffers: a cold part in the global workspace, a hot part (everything the
                    // Newton iterations touch except the Hessian) relative to its own base — LDS when it fits, else behind the cold part
    int n, nm, N, r, E, q, nx;
    int64_t H, Lam, A1, V, Hx, pose, w, Sc, zbuf, grp, otab, ibuf, Ng, cold_total;
    int64_t S, U, J, JU, Ai, T1, M, Mc, Mi, Li, Y, P, T2, x, xn, g, gn, dv, Xi, hot_total;
    int64_t total;
};

__host__ __device__ inline IpLayout ip_layout(int D, int k, int m, int E, bool closed) {
    IpLayout L;
    L.n = D * k; L.nm = D * m; L.N = L.n + L.nm; L.r = L.n - D; L.E = E; L.q = D * E; L.nx = D * D * E;
    int64_t o = 0;
    auto take = [&](int64_t len) { int64_t at = o; o += (len + 7) & ~(int64_t)7; return at; };
    const int64_t n = L.n, N = L.N, r = L.r > 0 ? L.r : 1, q = L.q, nx = L.nx, P2 = (int64_t)k * (k - 1) / 2;
    L.H = take(N * N); L.Lam = take(n * n); L.A1 = take(n * n); L.V = take(n * n); L.Hx = take(closed ? 8 : (nx + 8) * (nx + 8) + 64 * (nx + 8));   // rows of nx + 8 doubles, nx + 8 of them: whole 8 x 8 tiles everywhere (no Newton iterations when the pattern has a closed form)
    L.pose = take(12 * (int64_t)(k + m)); L.w = take(4 * P2 + 8);
    L.Sc = take(3 * N * N + 5 * n * n);            // correlated input edges (J, W J), closed form of correlated new edges (J_e, G, C, W, X)
    L.zbuf = take(7 * (int64_t)(E + 1)); L.grp = take(2 * (int64_t)k + 8); L.otab = take(8 * (int64_t)k + 8);
    L.Ng = take(n * D);                             // orthonormal gauge basis (closed-form path)
    L.ibuf = take(4 * (int64_t)k + 16);             // ints: Kruskal components (k), regrouped measurement list (2 k), vertex list of a record (k)
    L.cold_total = o;
    o = 0;
    L.S = take(n); L.U = take(n * r); L.J = take(2 * (int64_t)D * D * E); L.JU = take(q * r); L.Ai = take(n * n); L.T1 = take(n * r);
    L.M = take(r * r); L.Mc = take(r * r); L.Mi = take(r * r); L.Li = take(r * r); L.Y = take(n * n);
    L.P = take(q * q); L.T2 = take(q * r);
    L.x = take(nx); L.xn = take(nx); L.g = take(nx); L.gn = take(nx); L.dv = take(nx); L.Xi = take(nx);
    L.hot_total = o;
    L.total = L.cold_total + L.hot_total;
    return L;
}

__host__ __device__ inline int ip_pattern_size(int topology, double chord_ratio, int k) {
    if (k < 2) return 0;
    if (k == 2) return 1;
    const int msub = (int)((1 + chord_ratio) * (k - 1));
    const bool full = msub >= k * (k - 1) / 2;
    if (topology == SPG_TOPO_TREE) return k - 1;
    if (topology == SPG_TOPO_DENSE || (topology == SPG_TOPO_SUBGRAPH && full)) return k * (k - 1) / 2;
    if (topology == SPG_TOPO_SUBGRAPH) return msub;
    return k - 1;     // CliqueySubgraph / CliqueyDense: the Chow-Liu tree's measurements, grouped into correlated edges
}

// lower Cholesky of a small matrix by ONE thread (IEEE divide / sqrt: this is the control path of the line search)
__device__ inline bool chol_serial(double *A, int n, int ld) {
    for (int j = 0; j < n; j++) {
        double d = A[j * ld + j];
        for (int k = 0; k < j; k++) d -= A[j * ld + k] * A[j * ld + k];
        if (!(d > 0.0) || !isfinite(d)) return false;
        const double l = sqrt(d);
        A[j * ld + j] = l;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * ld + j];
            for (int k = 0; k < j; k++) s -= A[i * ld + k] * A[j * ld + k];
            A[i * ld + j] = s / l;
        }
    }
    return true;
}

// a / b given rb = 1 / b correctly rounded (one IEEE division, off the dependent chain): the quotient estimate a rb is
// within an ulp, its remainder a - q b is exact in an fma, and q + rem rb rounds to the correctly rounded quotient
// (Markstein; the one exception, a divisor whose significand is all ones, and operands within 2^-970 of the
// underflow threshold do not occur here: b is a Cholesky pivot). Three dependent operations instead of the ~13 of the
// IEEE division sequence — the factorisation and the substitution are chains of those.
__device__ __forceinline__ double div_by(double a, double b, double rb) {
    const double q = a * rb;
    const double rem = __builtin_fma(-q, b, a);
    return __builtin_fma(rem, rb, q);
}

// chol_serial for a compile-time size, fully unrolled so that the matrix stays in registers (a runtime-indexed private
// array lives in scratch memory: every access a round trip to L2); rl[j] = 1 / L[j][j]. Same operations in the same order.
template <int N>
__device__ __forceinline__ bool chol_static(double (&A)[N * N], double (&rl)[N]) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        double d = A[j * N + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j * N + k] * A[j * N + k];
        if (!(d > 0.0) || !isfinite(d)) return false;
        const double l = sqrt(d);
        A[j * N + j] = l;
        rl[j] = 1.0 / l;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double s = A[i * N + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i * N + k] * A[j * N + k];
            A[i * N + j] = div_by(s, l, rl[j]);
        }
    }
    return true;
}

// the value lane `lane` (uniform over the wavefront) holds, through the scalar unit
__device__ __forceinline__ double read_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// spgdev::tri_inverse_lower for n <= NMAX with thread c's column of L^-1 in registers (the LDS version reads back what
// the same thread wrote a moment before: one LDS round trip per row and column). The sum for row i starts at column 0
// instead of c: the entries above the diagonal are zeros, and adding zeros changes nothing. Same operations otherwise.
template <int NMAX>
__device__ __forceinline__ void tri_inverse_lower_reg(int tid, const double *L, double *Li, int n, int ld) {
    if (tid < n) {
        const int c = tid;
        double col[NMAX];
#pragma unroll
        for (int i = 0; i < NMAX; i++) {
            if (i < n) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < i; k++) s += L[i * ld + k] * col[k];
                const double rd = spgdev::fast_rcp(L[i * ld + i]);
                col[i] = (i < c) ? 0.0 : (i == c) ? rd : -s * rd;
            } else col[i] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < NMAX; i++) if (i < n) Li[i * ld + c] = col[i];
    }
    __syncthreads();
}

// spgdev::chol_lower for n <= NMAX by ONE wavefront with lane i holding row i in registers: the pivot and the scaled
// column travel by lane reads, no LDS round trips and no barriers inside (the team version: two barriers and four LDS
// round trips per column). Same operations in the same order; *flag = 1 on a non-positive pivot.
template <int NMAX>
__device__ __forceinline__ void chol_lower_reg(int tid, double *A, int n, int ld, int *flag) {
    if (tid < 64) {
        const int lane = tid;
        const int ns = __builtin_amdgcn_readfirstlane(n);
        double row[NMAX];
#pragma unroll
        for (int c = 0; c < NMAX; c++) row[c] = (lane < ns && c <= lane) ? A[lane * ld + c] : 0.0;
        bool anybad = false;
#pragma unroll
        for (int j = 0; j < NMAX; j++) {
            if (j < ns) {
                double d = read_lane(row[j], j);
                const bool bad = !(d > 0.0) || !isfinite(d);
                if (bad) { d = 1.0; anybad = true; }
                const double rs = spgdev::fast_rsqrt(d);
                row[j] = (lane > j) ? row[j] * rs : (lane == j ? d * rs : row[j]);
#pragma unroll
                for (int cc = j + 1; cc < NMAX; cc++) {
                    if (cc < ns) {
                        const double v = read_lane(row[j], cc);
                        const double upd = row[cc] - row[j] * v;
                        row[cc] = (lane >= cc) ? upd : row[cc];
                    }
                }
            }
        }
        if (anybad && lane == 0) *flag = 1;
#pragma unroll
        for (int c = 0; c < NMAX; c++) if (lane < ns && c <= lane) A[lane * ld + c] = row[c];
    }
    __syncthreads();
}

// C (M x N, row-major, ldc) = / += / -= op(A) op(B) by one workgroup on the fp64 matrix cores, operands in the L2 workspace
// (or LDS). op(A) is M x K: A[i * lda + k], or A[k * lda + i] when ta; op(B) is K x N: B[k * ldb + j], or B[j * ldb + k]
// when tb. The scalar loops this replaces (one lane per output entry, two reads per FMA) ran a 600 x 600 x 600 product of a
// cluster in 100-190 ms, 2 GFMA/s. 64 x 64 output tiles; wavefront w owns rows 16 w .. 16 w + 15 of the tile as four
// v_mfma_f64_16x16x4_f64 accumulators; K in chunks of 32 staged through LDS as As[i][k], Bs[j][k] (row stride 34: the
// fragment reads of a half wave fall on distinct bank pairs), the next chunk already in registers while the matrix
// cores work on the current one. lower_only skips the tiles strictly above the block diagonal (M = N products whose
// upper part is mirrored or not needed). lds: 2 * 64 * 34 doubles. mode 0: C = AB, 1: C += AB, 2: C -= AB.
constexpr int kGemmLds = 2 * 64 * 34;
constexpr int kBlockedLds = kGemmLds + 2 * 64 * 65;      // the blocked routines below: staging area + two 64 x 64 blocks
template <int NT_>
__device__ void team_gemm(const Team<NT_> T, double *C, int ldc, const double *A, int lda, bool ta, const double *B, int ldb, bool tb,
                          int M, int N, int K, int mode, bool lower_only, double *lds) {
    static_assert(NT_ == 256, "four wavefronts per tile");
    using d4 = __attribute__((ext_vector_type(4))) double;
    constexpr int KC = 32, LDK = 34;
    double *As = lds, *Bs = lds + 64 * LDK;
    const int tid = T.tid, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    // staging maps. k-contiguous operand: thread -> (row tid / 4, 8 consecutive k at (tid & 3) * 8);
    // row-contiguous (transposed) operand: thread -> (k = tid / 8, 8 consecutive rows at (tid & 7) * 8)
    const int rA = ta ? (tid & 7) * 8 : tid >> 2, kA = ta ? tid >> 3 : (tid & 3) * 8;
    const int rB = tb ? tid >> 2 : (tid & 7) * 8, kB = tb ? (tid & 3) * 8 : tid >> 3;
    for (int i0 = 0; i0 < M; i0 += 64)
        for (int j0 = 0; j0 < N; j0 += 64) {
            if (lower_only && j0 > i0) continue;
            d4 acc[4];
#pragma unroll
            for (int y = 0; y < 4; y++) acc[y] = d4{0, 0, 0, 0};
            double ra[8], rb[8];
            auto fetch = [&](int k0) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (ta) { const int i = i0 + rA + u, kk = k0 + kA; ra[u] = (i < M && kk < K) ? A[(long long)kk * lda + i] : 0.0; }
                    else    { const int i = i0 + rA, kk = k0 + kA + u; ra[u] = (i < M && kk < K) ? A[(long long)i * lda + kk] : 0.0; }
                    if (tb) { const int j = j0 + rB, kk = k0 + kB + u; rb[u] = (j < N && kk < K) ? B[(long long)j * ldb + kk] : 0.0; }
                    else    { const int j = j0 + rB + u, kk = k0 + kB; rb[u] = (j < N && kk < K) ? B[(long long)kk * ldb + j] : 0.0; }
                }
            };
            fetch(0);
            for (int k0 = 0; k0 < K; k0 += KC) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (ta) As[(rA + u) * LDK + kA] = ra[u]; else As[rA * LDK + kA + u] = ra[u];
                    if (tb) Bs[rB * LDK + kB + u] = rb[u]; else Bs[(rB + u) * LDK + kB] = rb[u];
                }
                T.sync();
                if (k0 + KC < K) fetch(k0 + KC);
#pragma unroll
                for (int kk = 0; kk < KC; kk += 4) {
                    const double av = As[(16 * w + li) * LDK + kk + lk];
#pragma unroll
                    for (int y = 0; y < 4; y++) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[(16 * y + li) * LDK + kk + lk], acc[y], 0, 0, 0);
                }
                T.sync();
            }
            // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const int row = i0 + 16 * w + lk + 4 * r4, col = j0 + 16 * y + li;
                    if (row < M && col < N) {
                        double *dst = C + (long long)row * ldc + col;
                        const double v = acc[y][r4];
                        *dst = mode == 0 ? v : (mode == 1 ? *dst + v : *dst - v);
                    }
                }
        }
    T.sync();
}

// Li = L^-1 (lower; strict upper zeroed) for a matrix of a cluster, in block columns of 64: the diagonal block is inverted by
// one lane per column as tri_inverse_lower does, the block below it is  -Li[below, below] (L[below, jb] Li[jb, jb])  — two
// products on the matrix cores, right to left so that the inverse of the trailing block is already in place. Li may not
// alias L; tmp: (n - 64) x 64 doubles of workspace; lds: kBlockedLds doubles. (tri_inverse_lower: 39 ms at n = 600, three calls per cluster.)
template <int NT_>
__device__ void tri_inverse_lower_blocked(const Team<NT_> T, const double *L, int ldl, double *Li, int ldi, int n, double *tmp, double *lds) {
    const int tid = T.tid;
    for (long long it = tid; it < (long long)n * n; it += NT_) { const int i = (int)(it / n), j = (int)(it - (long long)i * n); if (j > i) Li[(long long)i * ldi + j] = 0.0; }
    const int nb = (n + 63) / 64;
    for (int jb = nb - 1; jb >= 0; jb--) {
        const int j0 = jb * 64, w = min(64, n - j0), below = n - j0 - w;
        // diagonal block: staged into LDS, one lane per column there (out of L2 each of the 2 000 dependent steps of a lane
        // is a memory round trip: 2 ms per block), written back
        double *Lb = lds + kGemmLds, *Ib = Lb + 64 * 65;
        for (int it = tid; it < w * w; it += NT_) { const int i = it / w, c = it - i * w; Lb[i * 65 + c] = (c <= i) ? L[(long long)(j0 + i) * ldl + j0 + c] : 0.0; }
        T.sync();
        tri_inverse_lower<NT_>(T, Lb, Ib, w, 65);
        for (int it = tid; it < w * w; it += NT_) { const int i = it / w, c = it - i * w; if (c <= i) Li[(long long)(j0 + i) * ldi + j0 + c] = Ib[i * 65 + c]; }
        T.sync();
        if (below > 0) {
            // tmp = L[below, jb] * Li[jb, jb]   (below x w, K = w)
            team_gemm<NT_>(T, tmp, 64, L + (long long)(j0 + w) * ldl + j0, ldl, false, Li + (long long)j0 * ldi + j0, ldi, false, below, w, w, 0, false, lds);
            // Li[below, jb] = -Li[below, below] * tmp   (K = below; the factor is lower triangular with zeros above)
            team_gemm<NT_>(T, Li + (long long)(j0 + w) * ldi + j0, ldi, Li + (long long)(j0 + w) * ldi + j0 + w, ldi, false, tmp, 64, false, below, w, below, 0, false, lds);
            for (long long it = tid; it < (long long)below * w; it += NT_) {
                const int i = (int)(it / w), c = (int)(it - (long long)i * w);
                double *e = Li + (long long)(j0 + w + i) * ldi + j0 + c;
                *e = -*e;
            }
            T.sync();
        }
    }
}

// In-place lower Cholesky of a cluster's matrix (n x n in the L2 workspace, leading dimension ld; lower triangle read and
// written) in block columns of 64 with the trailing update on the matrix cores: the diagonal block is factorised in LDS
// (chol_lower, the LDS routine) and inverted there, the panel below it is its product with that inverse, the trailing
// matrix takes P P^T through team_gemm (lower block triangle, K = 64). lds: 64 * 65 doubles (the same region team_gemm
// stages through — the two never run together — and its inverse behind that area; kBlockedLds doubles in all); tmp: n * 64
// doubles of workspace. Failure as chol_lower: *T.flag = 1.
template <int NT_>
__device__ void chol_lower_blocked(const Team<NT_> T, double *A, int n, int ld, double *tmp, double *lds) {
    static_assert(kGemmLds >= 64 * 65, "the diagonal block borrows the staging area of team_gemm");
    const int tid = T.tid;
    double *Db = lds, *Dinv = lds + kGemmLds, *P = tmp;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int w = min(64, n - j0), below = n - j0 - w;
        for (int it = tid; it < w * w; it += NT_) { const int i = it / w, c = it - i * w; Db[i * 65 + c] = (c <= i) ? A[(long long)(j0 + i) * ld + j0 + c] : 0.0; }
        T.sync();
        chol_lower<NT_>(T, Db, w, 65);
        for (int it = tid; it < w * w; it += NT_) { const int i = it / w, c = it - i * w; if (c <= i) A[(long long)(j0 + i) * ld + j0 + c] = Db[i * 65 + c]; }
        if (below == 0) { T.sync(); break; }
        tri_inverse_lower<NT_>(T, Db, Dinv, w, 65);       // (reads the factor from LDS with stride 65, writes with the same stride)
        // (tri_inverse_lower uses one leading dimension for both matrices: Dinv has row stride 65 too)
        team_gemm<NT_>(T, P, 64, A + (long long)(j0 + w) * ld + j0, ld, false, Dinv, 65, true, below, w, w, 0, false, lds);     // P = A21 L11^-T
        for (long long it = tid; it < (long long)below * w; it += NT_) {
            const int i = (int)(it / w), c = (int)(it - (long long)i * w);
            A[(long long)(j0 + w + i) * ld + j0 + c] = P[(long long)i * 64 + c];
        }
        T.sync();
        team_gemm<NT_>(T, A + (long long)(j0 + w) * (ld + 1), ld, P, 64, false, P, 64, true, below, below, w, 2, true, lds);     // A22 -= P P^T
    }
}

// In-place lower Cholesky of a matrix in the L2 workspace (n x n, leading dimension ld; lower triangle read and written),
// for the matrices of a cluster (n = 600 .. 1200): chol_lower of spg_dev_la.hpp is written for LDS — one column per step,
// every entry of the trailing update three reads and a write, the pivot-column reads strided — and took 50-100 ms per call
// out of L2, five calls per cluster. Here a panel of up to 16 columns (all rows below the diagonal block) is factorised in
// LDS and applied to the trailing matrix in one sweep: lane = column (coalesced row segments), the panel rows of the
// lane's column in registers, kRif rows in flight. Same structure as chol_rows of the interior point below, without the
// right-hand-side row. On a non-positive pivot *T.flag = 1 (and the pivot is replaced by 1), as chol_lower does.
template <int NT_>
__device__ void chol_lower_panel(const Team<NT_> T, double *A, int n, int ld, double *panel /* LDS, kPanelDoubles */) {
    const int tid = T.tid;
    const int PB = max(1, min(16, kPanelDoubles / max(n, 1) - 1));
    const int PBS = PB + 1;
    bool bad = false;
    for (int j0 = 0; j0 < n; j0 += PB) {
        const int pb = min(PB, n - j0), nrows = n - j0;
        for (int rr = tid >> 4; rr < nrows; rr += NT_ / 16) {
            const int pc = tid & 15;
            if (pc < pb) panel[rr * PBS + pc] = (pc <= rr) ? A[(long long)(j0 + rr) * ld + j0 + pc] : 0.0;
        }
        T.sync();
        for (int pc = 0; pc < pb; pc++) {
            double dpiv = panel[pc * PBS + pc];
            if (!(dpiv > 0.0) || !isfinite(dpiv)) { bad = true; dpiv = 1.0; }
            const double l = sqrt(dpiv);
            T.sync();                                     // everybody has read the pivot
            for (int rr = pc + 1 + tid; rr < nrows; rr += NT_) panel[rr * PBS + pc] /= l;
            if (tid == 0) panel[pc * PBS + pc] = l;
            T.sync();
            const int p2 = pc + 1 + (tid & 15);
            if (p2 < pb)
                for (int rr = pc + 1 + (tid >> 4); rr < nrows; rr += NT_ / 16)
                    if (rr >= p2) panel[rr * PBS + p2] -= panel[rr * PBS + pc] * panel[p2 * PBS + pc];
            T.sync();
        }
        for (int rr = tid >> 4; rr < nrows; rr += NT_ / 16) {
            const int pc = tid & 15;
            if (pc < pb && pc <= rr) A[(long long)(j0 + rr) * ld + j0 + pc] = panel[rr * PBS + pc];
        }
        // trailing update: rows i >= j0 + pb, columns j0 + pb <= c <= i
        const int t0 = j0 + pb, wv = tid >> 6, lane = tid & 63;
        constexpr int kRif = 4, kWaves = NT_ / 64;
        for (int cb = t0 + lane; cb < n; cb += 64) {
            double pcv[16];
            const double *pcp = panel + (cb - j0) * PBS;
#pragma unroll
            for (int pc = 0; pc < 16; pc++) pcv[pc] = (pc < pb) ? pcp[pc] : 0.0;
            for (int ib = (cb - lane) + kRif * wv; ib < n; ib += kWaves * kRif) {
                double v[kRif];
#pragma unroll
                for (int u = 0; u < kRif; u++) { const int i = ib + u; v[u] = (i < n && cb <= i) ? A[(long long)i * ld + cb] : 0.0; }
#pragma unroll
                for (int u = 0; u < kRif; u++) {
                    const int i = ib + u;
                    if (i < n && cb <= i) {
                        const double *piv = panel + (i - j0) * PBS;
                        double acc = v[u];
#pragma unroll
                        for (int pc = 0; pc < 16; pc++) if (pc < pb) acc -= piv[pc] * pcv[pc];
                        A[(long long)i * ld + cb] = acc;
                    }
                }
            }
        }
        T.sync();
    }
    if (bad && tid == 0) *T.flag = 1;
    T.sync();
}

// Two instantiations per pose dimension: CLOSED takes the blankets whose pattern has a closed form (every correlated pattern,
// trees with correlated input edges — the cluster path with its matrix-core routines), the other one the interior point.
// Each leaves the other's blankets alone, and the compiler drops the other's code: with both in one kernel the register
// allocation of the interior point's two-tiles-per-thread factorisation degraded by half (sphere.g2o under Subgraph:
// 3.3 -> 4.4 s) once the cluster routines had grown.
template <int D, bool CLOSED>
__global__ __launch_bounds__(NT) void nfr_ip_kernel(spg::IpArgs a) {
    constexpr int DD = D * D, PS = (D == 6) ? 7 : 3, PSZ = (D == 6) ? 12 : 3, REC = PS + D * (D + 1) / 2;
    extern __shared__ double lds_pool[];
    __shared__ double red[NT];
    __shared__ int flag_s;
    __shared__ double sc[8];      // scalars broadcast by thread 0
    __shared__ int si[8];
    __shared__ double colbuf_s[2048];   // running vector of the Newton solve (d^2 E <= 2048)
    double *const colbuf = colbuf_s;
    const int tid = threadIdx.x;
    const int b = a.list[blockIdx.x];
    const spg_blanket_desc bd = a.blk[b];
    const int nv = bd.n_vert, m = bd.n_remove, k = nv - m;
    const int E = ip_pattern_size(a.topology, a.chord_ratio, k);
    const bool cliquey = a.topology == SPG_TOPO_CLIQUEY_SUBGRAPH || a.topology == SPG_TOPO_CLIQUEY_DENSE;
    // closed form (src/logdet_function.cpp:83-86): as many measurement rows as the target has rank — every correlated
    // pattern, and the uncorrelated ones that are trees (they come here when the blanket holds correlated input edges)
    if ((cliquey || E <= k - 1) != CLOSED) return;        // the other instantiation's blanket
    constexpr bool closed = CLOSED;
    const IpLayout L = ip_layout(D, k, m, E > 0 ? E : 1, closed);
    const int n = L.n, nm = L.nm, N = L.N, r = L.r, q = L.q, nx = L.nx;
    double *ws = a.ws + (int64_t)blockIdx.x * a.ws_stride;
    // hot buffers in LDS when this blanket's fit into what the launch reserved (latency per dependent phase ~0.1 us
    // instead of ~1 us out of L2: a Newton step is a few hundred such phases)
    // Dynamic LDS, one of two uses per blanket. (A) the whole Hessian, lower triangle packed by rows with the Newton
    // right-hand side as an extra row — its Cholesky and the substitution then never leave LDS (a 144-variable problem:
    // 85 KB), and behind it the hot small matrices if they still fit (they do at 144 variables: 48 KB); (B) when that does not fit: a 64 KB panel for the blocked out-of-L2 factorisation, and behind it the hot
    // small matrices if they fit.
    const long long packed_len = (long long)(nx + 1) * (nx + 2) / 2;
    const bool hx_lds = !closed && packed_len <= (long long)a.lds_doubles;
    const bool hx_tiled = hx_lds && nx + 1 <= 176 && !a.ip_untiled;     // (A') below: the Hessian goes straight into register tiles
    const bool hx_tiled2 = !closed && !hx_tiled && nx + 1 > 176 && nx + 1 <= 248 && !a.ip_untiled;     // two tiles per thread, factor in the L2 workspace
    const bool hx_streamed = !closed && !hx_lds && !hx_tiled2 && ((nx + 8) >> 3) * 73 <= kPanelDoubles && !a.ip_untiled;   // (B') below: up to 895 variables
    // (C): blocked Cholesky through team_gemm, running vector in dynamic LDS — everything beyond the streamed sizes (the
    // column-panel form (B) it replaces needed 84 s for a 1 980-variable blanket that (C) does in 9; SPG_IP_UNTILED=1 keeps (B))
    const bool hx_big = !closed && !hx_lds && !hx_tiled2 && !hx_streamed && (!a.ip_untiled || nx > kIpLdsVector);
    double *panel = lds_pool;
    const long long packed_pad = (packed_len + 1) & ~1LL;
    double *hot = hx_lds ? ((packed_pad + L.hot_total <= (int64_t)a.lds_doubles) ? lds_pool + packed_pad : ws + L.cold_total)
                         : ((kPanelDoubles + L.hot_total <= (int64_t)a.lds_doubles) ? lds_pool + kPanelDoubles : ws + L.cold_total);
    double *arena = a.arena;
    double *orec = a.mail ? (a.mail + (bd.out_off - a.mail_base)) : (arena + bd.out_off);
    if (tid == 0) flag_s = 0;
    __syncthreads();
    Team<NT> T{tid, red, &flag_s};
    // Cholesky of a matrix of the closed-form path: out of L2 in panels once it is large (never while the interior point's
    // packed Hessian owns the dynamic LDS), the LDS-style routine otherwise
    // a cluster: its matrices live in the L2 workspace and the whole dynamic LDS is free for the blocked routines
    const bool big = CLOSED && n >= 96 && !hx_lds && hot == ws + L.cold_total && kBlockedLds <= a.lds_doubles;
    // Cholesky of a matrix of the closed-form path: blocked on the matrix cores once it is large (tmp: cap doubles of free
    // workspace, nn * 64 needed — else in LDS panels), the LDS routine for the small ones and whenever the interior
    // point's packed Hessian owns the dynamic LDS
    auto chol_big = [&](double *A, int nn, int ld, double *tmp, long long cap) {
        if (nn >= 96 && big) {
            if ((long long)nn * 64 <= cap) chol_lower_blocked<NT>(T, A, nn, ld, tmp, panel);
            else chol_lower_panel<NT>(T, A, nn, ld, panel);
        } else chol_lower<NT>(T, A, nn, ld);
    };
    // strict lower triangle copied over the strict upper one (products of symmetric results are formed on the lower block
    // triangle only)
    auto mirror_lower = [&](double *Mx, int nn) {
        for (long long it = tid; it < (long long)nn * nn; it += NT) {
            const int i = (int)(it / nn), j = (int)(it - (long long)i * nn);
            if (j > i) Mx[it] = Mx[(long long)j * nn + i];
        }
        __syncthreads();
    };
    // (L L^T)^-1 into Out (full symmetric) with L^-1 left in Linv_; Out doubles as scratch of the blocked inverse
    auto spd_inverse_from_chol = [&](const double *Lc, double *Linv_, double *Out, int nn) {
        if (big && nn >= 96) {
            tri_inverse_lower_blocked<NT>(T, Lc, nn, Linv_, nn, nn, Out, panel);
            team_gemm<NT>(T, Out, nn, Linv_, nn, true, Linv_, nn, false, nn, nn, nn, 0, true, panel);      // lower block triangle,
            mirror_lower(Out, nn);                                                                           // then the other half
        } else {
            tri_inverse_lower<NT>(T, Lc, Linv_, nn, nn);
            gram_lower_inverse<NT>(T, Linv_, Out, nn, nn);
        }
    };
#ifdef SPG_IP_PROF
    long long ipt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ipt_last = (long long)__builtin_amdgcn_s_memtime();
#endif
#ifdef SPG_CF_PROF
    long long cfp[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, cfp_last = wall_clock64();
#endif
    int status = SPG_OK, info = 0, n_new = 0;
    double kld = __builtin_nan(""), min_gap = __builtin_inf();
    // L.w: [0, P2) pair weights | [P2, 2 P2) pair (i, j) ints | [2 P2, 3 P2) pop order + rejected bin (ints) | 3 P2 + 2: the pattern (2 E ints)
    int *pairs = reinterpret_cast<int *>(ws + L.w + 3 * ((int64_t)k * (k - 1) / 2) + 2);

    bool tab_mode = false;          // correlated patterns: the new-edge table was prepared in ws + L.otab
    auto finish = [&]() {
        __syncthreads();
        if (tid == 0) {
            orec[0] = (double)status; orec[1] = (double)info; orec[2] = kld; orec[3] = min_gap; orec[4] = (double)n_new;
            if (tab_mode) {
                const double *ot = ws + L.otab;
                int nvt = 0;
                for (int e = 0; e < n_new; e++) {
                    for (int c = 0; c < 4; c++) orec[SPG_OUT_HDR + 4 * e + c] = ot[4 * e + c];
                    nvt += (int)ot[4 * e + 3];
                }
                for (int i = 0; i < nvt; i++) orec[SPG_OUT_HDR + 4 * bd.n_new_max + i] = ot[4 * k + i];
            } else
            for (int e = 0; e < n_new; e++) {
                orec[SPG_OUT_HDR + 4 * e + 0] = (double)SPG_EDGE_BINARY;
                orec[SPG_OUT_HDR + 4 * e + 1] = (double)(e * REC);
                orec[SPG_OUT_HDR + 4 * e + 2] = (double)REC;
                orec[SPG_OUT_HDR + 4 * e + 3] = 2.0;
                orec[SPG_OUT_HDR + 4 * bd.n_new_max + 2 * e + 0] = (double)(m + pairs[2 * e]);
                orec[SPG_OUT_HDR + 4 * bd.n_new_max + 2 * e + 1] = (double)(m + pairs[2 * e + 1]);
            }
        }
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&orec[5], SPG_FINAL_WORD(a.tag), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    if (k < 2) { finish(); return; }
    if (E < 0) { status = SPG_ST_UNSUPPORTED; finish(); return; }
    if (a.lin_point != SPG_LIN_GLOBAL) { status = SPG_ST_NEEDS_LOCAL_OPTIMIZATION; finish(); return; }
    for (int e = 0; e < bd.n_edge; e++)    // pose-pose and correlated multi edges (src/topology_provider_binary.hpp:16-21); GLC edges belong to the GLC provider
        if (a.er[bd.edge_begin + e].kind == SPG_EDGE_GLC) { status = SPG_ST_UNSUPPORTED; finish(); return; }

    // ---- gather poses; H = sum_e J^T Omega J over the blanket's edges (one edge at a time: sizes are tiny next to
    // the Newton iterations below), vertices ordered [removed | kept] as the descriptors list them
    double *pose = ws + L.pose, *H = ws + L.H;
    for (int v = tid; v < nv; v += NT) {
        const double *p = arena + a.vpo[bd.vert_begin + v];
        if (D == 6) iso_from_tq(p, pose + v * PSZ);
        else { pose[v * 3] = p[0]; pose[v * 3 + 1] = p[1]; pose[v * 3 + 2] = p[2]; }
    }
    for (int it = tid; it < N * N; it += NT) H[it] = 0.0;
    __syncthreads();
    {
        double *Je = hot + L.Ai;    // scratch: Ji, Jj, T = Omega [Ji Jj]   (Ai is free until the interior point)
        for (int e = 0; e < bd.n_edge; e++) {
            const spg_edge_ref er = a.er[bd.edge_begin + e];
            const double *rec = arena + er.off;
            if (er.kind == SPG_EDGE_MULTI) {
                // MultiEdgeCorrelated::linearizeOplus (src/multi_edge_correlated.hpp:101-140): J (r x d q) has the two
                // pose-pose blocks of measurement i in row block i; information W^T W  =>  H += (W J)^T (W J)
                const int q = er.nv, nmi = (int)rec[0], rr = D * nmi, dq = D * q;
                const double *meas = rec + 1 + 2 * nmi, *Wd = meas + nmi * PS;
                double *Jm = ws + L.Sc, *Am = Jm + (int64_t)rr * dq;
                for (int it = tid; it < rr * dq; it += NT) Jm[it] = 0.0;
                __syncthreads();
                for (int i = tid; i < nmi; i += NT) {
                    const int la = (int)rec[1 + 2 * i], lb = (int)rec[2 + 2 * i];
                    const int va = a.ev[er.vbegin + la], vb = a.ev[er.vbegin + lb];
                    double Ja[DD], Jbb[DD];
                    if (D == 6) {
                        double Z[kIso];
                        iso_from_tq(meas + i * PS, Z);
                        se3_edge_jac(pose + va * PSZ, pose + vb * PSZ, Z, Ja, Jbb, nullptr);
                    } else {
                        se2_edge_jac(pose + va * PSZ, pose + vb * PSZ, meas + i * PS, Ja, Jbb, nullptr);
                    }
                    for (int x = 0; x < D; x++) for (int y = 0; y < D; y++) { Jm[(i * D + x) * dq + la * D + y] += Ja[x * D + y]; Jm[(i * D + x) * dq + lb * D + y] += Jbb[x * D + y]; }
                }
                __syncthreads();
                if (rr >= 64 && big) {
                    // a correlated edge of a cluster (a 180-measurement edge of parking.g2o: two products of 1.4 G FMAs, seconds
                    // in scalar loops): both on the matrix cores, the second into scratch behind Am and scattered from there
                    double *Pm = Am + (int64_t)rr * dq;
                    // Am = W J: a column of J belongs to one pose of the edge and is non-zero in the rows of the measurements at
                    // that pose only (lists in LDS; the Newton solve's vector is idle here)
                    int *incp = reinterpret_cast<int *>(colbuf), *inc = incp + 264, *icnt = inc + 528;
                    for (int v = tid; v <= q; v += NT) { incp[v] = 0; icnt[v] = 0; }
                    __syncthreads();
                    if (tid == 0) {
                        for (int i = 0; i < nmi; i++) { incp[(int)rec[1 + 2 * i] + 1]++; incp[(int)rec[2 + 2 * i] + 1]++; }
                        for (int v = 0; v < q; v++) incp[v + 1] += incp[v];
                        for (int i = 0; i < nmi; i++)
                            for (int side = 0; side < 2; side++) { const int v = (int)rec[1 + 2 * i + side]; inc[incp[v] + icnt[v]++] = i; }
                    }
                    __syncthreads();
                    for (int it = tid; it < rr * dq; it += NT) {
                        const int p = it / dq, c = it - p * dq, v = c / D;
                        double sacc = 0;
                        for (int e2 = incp[v]; e2 < incp[v + 1]; e2++) {
                            const int j = inc[e2];
#pragma unroll
                            for (int x = 0; x < D; x++) sacc += Wd[p * rr + j * D + x] * Jm[(j * D + x) * dq + c];
                        }
                        Am[it] = sacc;
                    }
                    __syncthreads();
                    team_gemm<NT>(T, Pm, dq, Am, dq, true, Am, dq, false, dq, dq, rr, 0, true, panel);      // (W J)^T (W J), lower block triangle
                    for (long long it = tid; it < (long long)dq * dq; it += NT) { const int i = (int)(it / dq), j = (int)(it - (long long)i * dq); if (j > i) Pm[it] = Pm[(long long)j * dq + i]; }
                    __syncthreads();
                    for (int it = tid; it < dq * dq; it += NT) {
                        const int r1 = it / dq, c1 = it - r1 * dq;
                        const int gi = a.ev[er.vbegin + r1 / D] * D + r1 % D, gj = a.ev[er.vbegin + c1 / D] * D + c1 % D;
                        H[gi * N + gj] += Pm[it];
                    }
                    __syncthreads();
                    continue;
                }
                for (int it = tid; it < rr * dq; it += NT) {
                    const int p = it / dq, c = it - p * dq;
                    double sacc = 0;
                    for (int t = 0; t < rr; t++) sacc += Wd[p * rr + t] * Jm[t * dq + c];
                    Am[it] = sacc;
                }
                __syncthreads();
                for (int it = tid; it < dq * dq; it += NT) {
                    const int r1 = it / dq, c1 = it - r1 * dq;
                    double sacc = 0;
                    for (int p = 0; p < rr; p++) sacc += Am[p * dq + r1] * Am[p * dq + c1];
                    const int gi = a.ev[er.vbegin + r1 / D] * D + r1 % D, gj = a.ev[er.vbegin + c1 / D] * D + c1 % D;
                    H[gi * N + gj] += sacc;
                }
                __syncthreads();
                continue;
            }
            const int vi = a.ev[er.vbegin], vj = a.ev[er.vbegin + 1];
            if (vi == vj) continue;
            if (tid == 0) {
                if (D == 6) {
                    double Z[kIso];
                    iso_from_tq(rec, Z);
                    se3_edge_jac(pose + vi * PSZ, pose + vj * PSZ, Z, Je, Je + DD, nullptr);
                } else {
                    se2_edge_jac(pose + vi * PSZ, pose + vj * PSZ, rec, Je, Je + DD, nullptr);
                }
            }
            __syncthreads();
            // T[p][c'] = sum_t Omega[p][t] * Jcat[t][c'], Jcat = [Ji | Jj] (D x 2D)
            for (int it = tid; it < D * 2 * D; it += NT) {
                const int p = it / (2 * D), c = it - p * 2 * D;
                const double *Jc = (c < D) ? Je : Je + DD;
                const int cc = (c < D) ? c : c - D;
                double s = 0;
                for (int t = 0; t < D; t++) {
                    const int lo = p < t ? p : t, hi = p < t ? t : p;
                    s += rec[PS + lo * D - lo * (lo - 1) / 2 + (hi - lo)] * Jc[t * D + cc];
                }
                Je[2 * DD + it] = s;
            }
            __syncthreads();
            for (int it = tid; it < 4 * DD; it += NT) {
                const int rr = it / (2 * D), c = it - rr * 2 * D;      // entry (rr, c) of Jcat^T T, 2D x 2D
                const double *Jr = (rr < D) ? Je : Je + DD;
                const int r2 = (rr < D) ? rr : rr - D;
                double s = 0;
                for (int p = 0; p < D; p++) s += Jr[p * D + r2] * Je[2 * DD + p * 2 * D + c];
                const int gi = ((rr < D) ? vi : vj) * D + r2, gj = ((c < D) ? vi : vj) * D + ((c < D) ? c : c - D);
                H[gi * N + gj] += s;
            }
            __syncthreads();
        }
    }
    CFP(0);
    // ---- Schur complement onto the kept block (src/vertex_remover.cpp:409-449)
    double *Lam = ws + L.Lam;
    if (nm > 0) {
        chol_big(H, nm, N, ws + L.A1, (long long)n * n);        // H_mm = L L^T (lower, in place)
        if (flag_s) { status = SPG_ST_HMM_NOT_PD; finish(); return; }
        // W = L^-1 H_mk (nm x n), stored over the upper right block
        if (big && nm >= 96 && nm <= n) {
            // through L^-1 (blocked inverse into A1, scratch in V) and one product on the matrix cores into Lam, copied back
            double *Linv_m = ws + L.A1, *Wt = Lam;
            tri_inverse_lower_blocked<NT>(T, H, N, Linv_m, nm, nm, ws + L.V, panel);
            team_gemm<NT>(T, Wt, n, Linv_m, nm, false, H + nm, N, false, nm, n, nm, 0, false, panel);
            for (int it = tid; it < nm * n; it += NT) { const int i = it / n, c = it - i * n; H[i * N + nm + c] = Wt[it]; }
            __syncthreads();
        } else {
            for (int c = tid; c < n; c += NT)
                for (int i = 0; i < nm; i++) {
                    double s = H[i * N + nm + c];
                    for (int t = 0; t < i; t++) s -= H[i * N + t] * H[t * N + nm + c];
                    H[i * N + nm + c] = s / H[i * N + i];
                }
            __syncthreads();
        }
    }
    CFP(1);
    if (big && nm > 0) {
        for (int it = tid; it < n * n; it += NT) { const int i = it / n, j = it - i * n; Lam[it] = H[(nm + i) * N + nm + j]; }
        __syncthreads();
        team_gemm<NT>(T, Lam, n, H + nm, N, true, H + nm, N, false, n, n, nm, 2, false, panel);      // Lambda_t = H_kk - W^T W
    } else {
        for (int it = tid; it < n * n; it += NT) {
            const int i = it / n, j = it - i * n;
            double s = H[(nm + i) * N + nm + j];
            for (int t = 0; t < nm; t++) s -= H[t * N + nm + i] * H[t * N + nm + j];
            Lam[it] = s;
        }
        __syncthreads();
    }
    for (int it = tid; it < n * n; it += NT) { const int i = it / n, j = it - i * n; if (j > i) Lam[it] = 0.5 * (Lam[it] + Lam[j * n + i]); }
    __syncthreads();
    for (int it = tid; it < n * n; it += NT) { const int i = it / n, j = it - i * n; if (j < i) Lam[it] = Lam[j * n + i]; }
    __syncthreads();
    // the blanket's target information for the batch interface (spg_batch_result::target_info), as the blanket kernel leaves it
    if (bd.tinfo_off >= 0) {
        double *dst = arena + bd.tinfo_off;
        for (int it = tid; it < n * n; it += NT) dst[it] = Lam[it];
    }
    {
        double bad = 0;
        for (int it = tid; it < n * n; it += NT) if (!isfinite(Lam[it])) bad = 1;
        if (T.sum(bad) > 0) { status = SPG_ST_NONFINITE; finish(); return; }
    }

    CFP(2);
    // ---- sparsity pattern (src/pseudo_chow_liu.cpp:33-87)
    if (k == 2) {
        if (tid == 0) { pairs[0] = 0; pairs[1] = 1; }
    } else if (!cliquey && E == k * (k - 1) / 2) {
        if (tid == 0) { int e = 0; for (int i = 0; i < k - 1; i++) for (int j = i + 1; j < k; j++) { pairs[2 * e] = i; pairs[2 * e + 1] = j; e++; } }
    } else {
        // pseudo-Chow-Liu weights from Sigma~ = (Lambda_t + I)^-1 (:169-196), Kruskal in pop order, accepted edges first,
        // then the rejected ones (:253-289); the pattern is the first E of that bin
        double *A1 = ws + L.A1, *Vv = ws + L.V, *Sg = hot + L.Y;
        for (int it = tid; it < n * n; it += NT) { const int i = it / n, j = it - i * n; A1[it] = Lam[it] + (i == j ? 1.0 : 0.0); }
        __syncthreads();
        chol_big(A1, n, n, Vv, (long long)n * n);
        if (flag_s) { status = SPG_ST_TIKHONOV_NOT_PD; finish(); return; }
        spd_inverse_from_chol(A1, Vv, Sg, n);
        CFP(15);
        const int P2 = k * (k - 1) / 2;
        double *w = ws + L.w;               // w[p], then (after P2) scratch
        int *pij = reinterpret_cast<int *>(ws + L.w + P2);    // 2 P2 ints
        for (int p = tid; p < P2; p += NT) {
            int i = 0, rem = p;
            while (rem >= k - 1 - i) { rem -= k - 1 - i; i++; }
            const int j = i + 1 + rem;
            pij[2 * p] = i; pij[2 * p + 1] = j;
            // log det of the marginal covariance blocks: ld_i + ld_j - ld_ij
            double B[4 * DD];
            auto ld_of = [&](const int *vs, int cnt) {
                const int s = cnt * D;
                for (int x = 0; x < s; x++) for (int y = 0; y < s; y++) B[x * s + y] = Sg[(vs[x / D] * D + x % D) * n + vs[y / D] * D + y % D];
                if (!chol_serial(B, s, s)) return __builtin_nan("");
                double l = 0;
                for (int x = 0; x < s; x++) l += log(B[x * s + x]);
                return 2.0 * l;
            };
            const int vi[1] = {i}, vj[1] = {j}, vij[2] = {i, j};
            const double li = ld_of(vi, 1), lj = ld_of(vj, 1), lij = ld_of(vij, 2);
            w[p] = -((li + lj) - lij);      // stored negated: ascending sort == max-heap pop order
            if (!isfinite(w[p])) flag_s = 1;
        }
        __syncthreads();
        CFP(16);
        if (flag_s) { status = SPG_ST_TIKHONOV_NOT_PD; finish(); return; }
        // pop order: weight descending, ties by (i, j) ascending = index ascending (rank by counting, all lanes: a
        // serial sort of the k (k - 1) / 2 weights of a 100-vertex cluster would take seconds)
        if (big && P2 > 2048) {
            // the same ranks with the keys passing through LDS in chunks (every lane reads every key: out of L2 the 300 M
            // comparisons of a 190-vertex cluster took tens of milliseconds)
            int *perm_o = pij + 2 * P2;
            for (int base = 0; base < P2; base += NT) {
                const int i = base + tid;
                const double ki = (i < P2) ? w[i] : 0.0;
                int rk = 0;
                for (int c0 = 0; c0 < P2; c0 += kPanelDoubles) {
                    const int cnt = min(kPanelDoubles, P2 - c0);
                    __syncthreads();
                    for (int t = tid; t < cnt; t += NT) panel[t] = w[c0 + t];
                    __syncthreads();
                    if (i < P2) {
                        // branch-free, eight keys per step (the short-circuit form compiled to a divergent branch per key)
                        int j = 0;
                        for (; j + 8 <= cnt; j += 8) {
                            double kj[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) kj[u] = panel[j + u];
#pragma unroll
                            for (int u = 0; u < 8; u++) rk += (int)(kj[u] < ki) | ((int)(kj[u] == ki) & (int)(c0 + j + u < i));
                        }
                        for (; j < cnt; j++) { const double kj = panel[j]; rk += (int)(kj < ki) | ((int)(kj == ki) & (int)(c0 + j < i)); }
                    }
                }
                if (i < P2) perm_o[rk] = i;
            }
        } else sort_ascending<NT>(T, w, 1, P2, pij + 2 * P2);
        __syncthreads();
        CFP(17);
        // Kruskal over the pairs in pop order. The union-find labels live in LDS and the pairs reach the one lane that walks
        // them in batches staged by all lanes (a lane alone reading order -> pair -> labels out of L2 spends three memory
        // round trips per pair: 50 ms for the 17 000 pairs of a 190-vertex cluster); the scan stops when the tree is
        // complete — every later pair is a rejected one, in pop order.
        int *order = pij + 2 * P2;     // P2 ints (inside the 4 P2 + 8 doubles reserved at L.w)
        int *rej = order + P2;         // rejected pair indices, in pop order (P2 more ints)
        {
            int *kcomp = reinterpret_cast<int *>(colbuf);     // (the Newton solve's vector is idle here) k <= 256 labels
            int *kb = kcomp + 256;                            // a batch: (pair index, i, j) per lane
            for (int v = tid; v < k; v += NT) kcomp[v] = v;
            if (tid == 0) { si[4] = 0; si[5] = 0; si[6] = 0; }
            __syncthreads();
            int s0 = 0;
            while (s0 < P2) {
                if (s0 + tid < P2) { const int pp = order[s0 + tid]; kb[3 * tid] = pp; kb[3 * tid + 1] = pij[2 * pp]; kb[3 * tid + 2] = pij[2 * pp + 1]; }
                __syncthreads();
                if (tid == 0) {
                    int nacc = si[4], nrej = si[5], last = si[6];
                    const int cnt = min(NT, P2 - s0);
                    for (int t = 0; t < cnt; t++) {
                        const int pp = kb[3 * t], i = kb[3 * t + 1], j = kb[3 * t + 2];
                        if (kcomp[i] != kcomp[j]) {
                            pairs[2 * nacc] = i; pairs[2 * nacc + 1] = j;
                            nacc++;
                            const int ci = kcomp[i], cj = kcomp[j];
                            for (int v = 0; v < k; v++) if (kcomp[v] == cj) kcomp[v] = ci;
                            last = s0 + t;
                        } else rej[nrej++] = pp;
                    }
                    si[4] = nacc; si[5] = nrej; si[6] = last;
                }
                __syncthreads();
                s0 += NT;
                if (si[4] == k - 1) break;
            }
            // the rest of the bin: rejected pairs in pop order (only the first E - (k - 1) of the bin are ever read)
            const int nrej0 = si[5];
            for (int s2 = s0 + tid; s2 < P2 && nrej0 + (s2 - s0) < E; s2 += NT) rej[nrej0 + (s2 - s0)] = order[s2];
            __syncthreads();
            if (tid == 0) si[5] = nrej0 + max(0, P2 - s0);
            __syncthreads();
        }
        if (tid == 0) {
            const int nacc = si[4], nrej = si[5];
            for (int t = 0; nacc + t < E && t < nrej; t++) { pairs[2 * (nacc + t)] = pij[2 * rej[t]]; pairs[2 * (nacc + t) + 1] = pij[2 * rej[t] + 1]; }
            if (cliquey) {
                // groups of tree measurements = correlated edges (src/pseudo_chow_liu.cpp:62-85, fillCliques :198-251);
                // cliques as bit masks over the k <= 64 kept vertices. grp: [0] number of groups, [1 + g] start of group g
                // in the regrouped measurement list, which replaces pairs[] (bin order inside a group).
                int *grp = reinterpret_cast<int *>(ws + L.grp);
                int *tmp = reinterpret_cast<int *>(ws + L.ibuf) + k;
                const int msub = (int)((1 + a.chord_ratio) * (k - 1));
                if (a.topology == SPG_TOPO_CLIQUEY_DENSE || msub >= k * (k - 1) / 2) {
                    // one fully correlated edge over the whole tree, in bin order (any k)
                    grp[0] = 1; grp[1] = 0; grp[2] = k - 1;
                    si[1] = k - 1;
                } else if (k > 64) {
                    si[1] = -1;     // the clique masks below hold 64 vertices
                } else {
                unsigned long long mask[64];
                int ncl = k - 1;
                for (int i = 0; i < ncl; i++) mask[i] = (1ull << pairs[2 * i]) | (1ull << pairs[2 * i + 1]);
                {
                    bool joined = true;
                    for (int nedges = k - 1, maxfill = 1; nedges < msub && joined; maxfill++) {
                        joined = false;
                        int minfill = 0x7fffffff;
                        for (int i = 0; i < ncl; i++)
                            for (int j = i + 1; j < ncl; j++) {
                                if (!(mask[i] & mask[j])) continue;
                                const int thisfill = (__popcll(mask[i]) - 1) * (__popcll(mask[j]) - 1);
                                minfill = min(thisfill, minfill);
                                if (thisfill <= maxfill && nedges + thisfill <= msub) {
                                    mask[i] |= mask[j];
                                    nedges += thisfill;
                                    for (int t = j; t + 1 < ncl; t++) mask[t] = mask[t + 1];
                                    ncl--;
                                    joined = true;
                                    j--;
                                }
                            }
                        if (!joined && minfill > maxfill) { joined = true; maxfill = minfill - 1; }
                    }
                }
                int nout = 0;
                grp[0] = ncl;
                for (int gidx = 0; gidx < ncl; gidx++) {
                    grp[1 + gidx] = nout;
                    for (int i = 0; i < k - 1; i++)
                        if (((mask[gidx] >> pairs[2 * i]) & 1) && ((mask[gidx] >> pairs[2 * i + 1]) & 1)) {
                            if (nout < k - 1) { tmp[2 * nout] = pairs[2 * i]; tmp[2 * nout + 1] = pairs[2 * i + 1]; }
                            nout++;
                        }
                }
                grp[1 + ncl] = nout;
                si[1] = nout;
                if (nout == k - 1) for (int i = 0; i < 2 * nout; i++) pairs[i] = tmp[i];
                }
            }
        }
        __syncthreads();
        {
            // smallest relative gap between consecutive weights up to the last accepted pair
            const int upto = min(si[6] + 1, P2 - 1);
            double g = __builtin_inf();
            for (int s2 = tid; s2 < upto; s2 += NT) {
                const double x = -w[order[s2]], y = -w[order[s2 + 1]];
                const double den = fmax(fmax(fabs(x), fabs(y)), 1e-300);
                g = fmin(g, (x - y) / den);
            }
            red[tid] = g;
            __syncthreads();
            if (tid == 0) { for (int t = 1; t < NT; t++) g = fmin(g, red[t]); sc[0] = g; }
        }
        __syncthreads();
        min_gap = sc[0];
        // a tree edge inside two cliques would mean more measurements than rank and no closed form (the reference would then
        // run the interior point over correlated blocks, src/logdet_function.cpp:135-214). fillCliques cannot produce it: the
        // cliques start as the k - 1 edges of the Chow-Liu TREE and only ever merge where they intersect, so they stay
        // edge-disjoint connected subtrees, two of which share at most one vertex — every tree edge lies in exactly one
        // clique and the measurements always add up to the rank. Kept as a check of that invariant.
        if (cliquey && si[1] != k - 1) { status = SPG_ST_UNSUPPORTED; finish(); return; }
    }
    __syncthreads();
    if (closed && (!cliquey || k == 2) && tid == 0) {      // uncorrelated closed form: one group per measurement
        int *grp = reinterpret_cast<int *>(ws + L.grp);
        grp[0] = E;
        for (int e = 0; e <= E; e++) grp[1 + e] = e;
    }
    __syncthreads();

    CFP(3);
    // ---- new edge skeleton: measurement from the state, Jacobians (src/topology_provider_binary.hpp:38-47)
    double *Jb = hot + L.J;
    for (int e = tid; e < E; e += NT) {
        const int va = m + pairs[2 * e], vb = m + pairs[2 * e + 1];
        double *rec = closed ? (ws + L.zbuf + (int64_t)e * PS) : (arena + bd.new_off + (int64_t)e * REC);
        if (D == 6) {
            double Z[kIso], qd[4];
            iso_inv_mul(pose + va * PSZ, pose + vb * PSZ, Z);
            R_to_quat(Z, qd);
            rec[0] = Z[9]; rec[1] = Z[10]; rec[2] = Z[11]; rec[3] = qd[0]; rec[4] = qd[1]; rec[5] = qd[2]; rec[6] = qd[3];
            se3_edge_jac(pose + va * PSZ, pose + vb * PSZ, Z, Jb + e * 2 * DD, Jb + e * 2 * DD + DD, nullptr);
        } else {
            double z[3];
            se2_between(pose + va * PSZ, pose + vb * PSZ, z);
            rec[0] = z[0]; rec[1] = z[1]; rec[2] = z[2];
            se2_edge_jac(pose + va * PSZ, pose + vb * PSZ, z, Jb + e * 2 * DD, Jb + e * 2 * DD + DD, nullptr);
        }
    }
    __syncthreads();
    // sparseJacobian() drops entries below epsilon (src/logdet_function.cpp:335); it feeds J U of the Hessian
    double *A1 = ws + L.A1, *Vv = ws + L.V, *Sv = hot + L.S, *U = hot + L.U;
    // ---- closed-form path, gauge route (as blanket_kernel): Lambda_t of relative-pose edges has the rigid motions
    // of the blanket as its exact null space; with N^ an orthonormal basis of it and C = Lambda_t + N^ N^^T (SPD):
    //   U S U^T = C^-1 - N^ N^^T,  J_e N^ = 0 => J_e Sigma J_e^T = J_e C^-1 J_e^T,  log det S = -log det C,
    //   tr(S M) = tr(C^-1 A),  log det(U^T A U) = log det(A + N^ N^^T)
    // — Cholesky-class work instead of the Jacobi sweeps below (a 600 x 600 target of a Dense cluster: seconds of them).
    // Taken only if trace(C^-1) < 5e4, which proves lambda_{d+1}(Lambda_t) > 1e-5, the reference's `smalleigs <= dim` branch.
    bool gauge_ok = false;
    double logdetC = 0;
    double *Ng = ws + L.Ng;
    if (closed) {
        for (int v = tid; v < k; v += NT) {
            const double *X = pose + (m + v) * PSZ;
            double *Gv = Ng + v * DD;
            if (D == 6) {
                for (int rr = 0; rr < 3; rr++)
                    for (int c = 0; c < 3; c++) {
                        // rows of vertex v: [R^T | -R^T [t]x ; 0 | 1/2 R^T]  (update X <- X * fromVectorMQT(delta))
                        const int ca = (c + 1) % 3, cb = (c + 2) % 3;
                        const double rt = X[c * 3 + rr];
                        const double cx_a = X[9 + cb], cx_b = -X[9 + ca];
                        const double val = -(X[ca * 3 + rr] * cx_a + X[cb * 3 + rr] * cx_b);
                        Gv[rr * 6 + c] = rt;
                        Gv[rr * 6 + 3 + c] = val;
                        Gv[(3 + rr) * 6 + c] = 0.0;
                        Gv[(3 + rr) * 6 + 3 + c] = 0.5 * rt;
                    }
            } else {
                Gv[0] = 1; Gv[1] = 0; Gv[2] = -X[1];
                Gv[3] = 0; Gv[4] = 1; Gv[5] = X[0];
                Gv[6] = 0; Gv[7] = 0; Gv[8] = 1;
            }
        }
        __syncthreads();
        double *nn_s = hot + L.T2;       // D x D: N^T N, then L^-1 of its Cholesky factor
        if (tid < DD) {
            const int rr = tid / D, c = tid - rr * D;
            double sacc = 0;
            for (int i = 0; i < n; i++) sacc += Ng[i * D + rr] * Ng[i * D + c];
            nn_s[tid] = sacc;
        }
        __syncthreads();
        if (tid == 0) {
            double Ab[DD], Lin[DD];
            for (int i = 0; i < DD; i++) Ab[i] = nn_s[i];
            si[3] = chol_serial(Ab, D, D) ? 1 : 0;
            for (int c = 0; c < D; c++)
                for (int i = 0; i < D; i++) {
                    if (i < c) Lin[i * D + c] = 0.0;
                    else if (i == c) Lin[i * D + c] = 1.0 / Ab[c * D + c];
                    else {
                        double sacc = 0;
                        for (int qq = c; qq < i; qq++) sacc += Ab[i * D + qq] * Lin[qq * D + c];
                        Lin[i * D + c] = -sacc / Ab[i * D + i];
                    }
                }
            for (int i = 0; i < DD; i++) nn_s[DD + i] = Lin[i];
        }
        __syncthreads();
        bool gfail = si[3] == 0;
        for (int i = tid; i < n; i += NT) {
            double nv_[D];
            for (int c = 0; c < D; c++) {
                double sacc = 0;
                for (int qq = 0; qq <= c; qq++) sacc += Ng[i * D + qq] * nn_s[DD + c * D + qq];
                nv_[c] = sacc;
            }
            for (int c = 0; c < D; c++) Ng[i * D + c] = nv_[c];
        }
        __syncthreads();
        for (int it = tid; it < n * n; it += NT) {
            const int i = it / n, j = it - i * n;
            double sacc = Lam[it];
            for (int qq = 0; qq < D; qq++) sacc += Ng[i * D + qq] * Ng[j * D + qq];
            A1[it] = sacc;
        }
        if (tid == 0) flag_s = 0;
        __syncthreads();
        chol_big(A1, n, n, Vv, (long long)n * n);
        gfail |= flag_s != 0;
        __syncthreads();
        if (tid == 0) flag_s = 0;
        __syncthreads();
        if (!gfail) {
            double l = 0;
            for (int i = tid; i < n; i += NT) l += log(A1[i * n + i]);
            logdetC = 2.0 * T.sum(l);
            double *Cinv = hot + L.Y;
            spd_inverse_from_chol(A1, Vv, Cinv, n);
            double trc = 0;
            for (int i = tid; i < n; i += NT) trc += Cinv[i * n + i];
            trc = T.sum(trc);
            gauge_ok = isfinite(trc) && trc < 5e4 && isfinite(logdetC);
        }
    }
    CFP(4);
    // ---- spectrum of the target (src/logdet_function.cpp:14-64): the interior point needs U and S; the closed form only
    // when the gauge route did not apply
    if (!gauge_ok) {
    for (int it = tid; it < n * n; it += NT) A1[it] = Lam[it];
    __syncthreads();
    {
        // spectrum of a cluster's target (n of several hundred: Jacobi sweeps out of L2 take tens of seconds) by
        // tridiagonalisation + implicit QL; needs 8 n doubles of LDS, which the launch has free when the hot buffers do not
        // fit there anyway. Small targets keep the Jacobi route the LDS kernels and the oracle share.
        const bool eig_big = !a.eig_jacobi && big && n >= 128 && hot == ws + L.cold_total && 8LL * n <= (long long)a.lds_doubles && (long long)n * r >= 3LL * n && n <= 8 * NT;
        const bool eig_ok = eig_big ? tridiag_eigh<NT>(T, A1, Vv, n, n, hot + L.T1, lds_pool) : jacobi_eigh<NT>(T, A1, Vv, n, n, hot + L.T1);
        if (!eig_ok) { status = SPG_ST_EIG_FAIL; finish(); return; }
    }
    {
        int *perm = reinterpret_cast<int *>(hot + L.T1);     // n ints
        sort_ascending<NT>(T, A1, n + 1, n, perm);
        __syncthreads();
        double small = 0;
        for (int i = tid; i < n; i += NT) if (A1[i * (n + 1)] < 1e-5) small += 1;
        const int smalleigs = (int)T.sum(small);
        if (smalleigs > D) {
            // chooseDimensions (src/logdet_function.cpp:40-59,66-81): of the candidate directions (eigenvalues below the cutoff)
            // drop the d whose image under the new measurements' Jacobian is smallest; keep the others with clamped 1 / lambda
            info |= SPG_INFO_RANK_DEFICIENT;
            double *norms = hot + L.T1 + n;          // (behind the n ints of perm)
            int *keepidx = reinterpret_cast<int *>(hot + L.T1 + 2 * n);
            for (int c = tid; c < smalleigs; c += NT) {
                const int col = perm[c];
                double nrm2 = 0;
                for (int e = 0; e < E; e++) {
                    const int oa = pairs[2 * e] * D, ob = pairs[2 * e + 1] * D;
                    const double *Ja = Jb + e * 2 * DD, *Jbb = Ja + DD;
                    for (int p = 0; p < D; p++) {
                        double sacc = 0;
                        for (int t = 0; t < D; t++) {
                            const double ja = Ja[p * D + t], jb = Jbb[p * D + t];
                            if (fabs(ja) >= 2.220446049250313e-16) sacc += ja * Vv[(oa + t) * n + col];
                            if (fabs(jb) >= 2.220446049250313e-16) sacc += jb * Vv[(ob + t) * n + col];
                        }
                        nrm2 += sacc * sacc;
                    }
                }
                norms[c] = sqrt(nrm2);
            }
            __syncthreads();
            if (tid == 0) {
                // the d smallest (norm, index) pairs are dropped; the kept eigenpairs stay in ascending eigenvalue order
                for (int c = 0; c < smalleigs; c++) keepidx[n + c] = 0;            // dropped flags
                for (int t = 0; t < D; t++) {
                    int best = -1;
                    for (int c = 0; c < smalleigs; c++)
                        if (!keepidx[n + c] && (best < 0 || norms[c] < norms[best])) best = c;
                    keepidx[n + best] = 1;
                }
                int j = 0;
                for (int i = 0; i < n; i++) if (!(i < smalleigs && keepidx[n + i])) keepidx[j++] = i;
            }
            __syncthreads();
            const double wmax = A1[perm[n - 1] * (n + 1)];
            for (int j = tid; j < r; j += NT) { const double wv = A1[perm[keepidx[j]] * (n + 1)]; Sv[j] = fmin(fabs(1.0 / wv), 1e6 / wmax); }
            for (int it = tid; it < n * r; it += NT) { const int i = it / r, j = it - i * r; U[it] = Vv[i * n + perm[keepidx[j]]]; }
            __syncthreads();
        } else {
        for (int j = tid; j < r; j += NT) Sv[j] = 1.0 / A1[perm[D + j] * (n + 1)];
        for (int it = tid; it < n * r; it += NT) { const int i = it / r, j = it - i * r; U[it] = Vv[i * n + perm[D + j]]; }
        __syncthreads();
        }
    }
    }
    CFP(5);
    double logdetS = 0;
    if (!gauge_ok) {
        double l = 0;
        for (int j = tid; j < r; j += NT) l += log(Sv[j]);
        logdetS = T.sum(l);
    } else logdetS = -logdetC;
    double *Ai = hot + L.Ai, *T1 = hot + L.T1, *M = hot + L.M, *Mc = hot + L.Mc, *Mi = hot + L.Mi, *Li = hot + L.Li, *Y = hot + L.Y;
    // LogdetFunction::value (src/logdet_function.cpp:119-133) of the product information held in Ai (full symmetric):
    // M = U^T A U, 1/2 (tr(M S) - log det M - log det S - r); leaves chol(M) in Mc
    auto value_from_A = [&](bool &ok) -> double {
        for (int it = tid; it < n * r; it += NT) {
            const int i = it / r, c = it - i * r;
            double sacc = 0;
#pragma unroll 8
            for (int t = 0; t < n; t++) sacc += Ai[i * n + t] * U[t * r + c];
            T1[it] = sacc;
        }
        __syncthreads();
        for (int it = tid; it < r * r; it += NT) {
            const int i = it / r, j = it - i * r;
            if (j >= i) {
                double sacc = 0;
#pragma unroll 8
                for (int t = 0; t < n; t++) sacc += U[t * r + i] * T1[t * r + j];
                M[i * r + j] = sacc; M[j * r + i] = sacc;
            }
        }
        __syncthreads();
        IPT(8);
        double tr = 0;
        for (int i = tid; i < r; i += NT) tr += M[i * r + i] * Sv[i];
        tr = T.sum(tr);
        for (int it = tid; it < r * r; it += NT) Mc[it] = M[it];
        if (tid == 0) flag_s = 0;
        __syncthreads();
        IPT(10);
        if (r <= 32) chol_lower_reg<32>(tid, Mc, r, r, &flag_s);
        else chol_lower<NT>(T, Mc, r, r);
        ok = flag_s == 0;
        __syncthreads();
        IPT(9);
        if (!ok) { if (tid == 0) flag_s = 0; __syncthreads(); return __builtin_inf(); }
        double l = 0;
        for (int i = tid; i < r; i += NT) l += log(Mc[i * r + i]);
        l = T.sum(l);
        return 0.5 * (tr - 2.0 * l - logdetS - r);
    };
    // the same value for a cluster on the eigen route, products and factorisation on the matrix cores. A function of its own:
    // value_from_A sits inside every line-search step of the interior point, and the extra branches cost that loop 30 %
    // (register allocation of the whole kernel) when they lived there.
    auto value_from_A_cluster = [&](bool &ok) -> double {
        team_gemm<NT>(T, T1, r, Ai, n, false, U, r, false, n, r, n, 0, false, panel);         // T1 = A U
        team_gemm<NT>(T, M, r, U, r, true, T1, r, false, r, r, n, 0, true, panel);            // M = U^T (A U), lower block triangle
        for (long long it = tid; it < (long long)r * r; it += NT) { const int i = (int)(it / r), j = (int)(it - (long long)i * r); if (j > i) M[it] = M[(long long)j * r + i]; }
        __syncthreads();
        double tr = 0;
        for (int i = tid; i < r; i += NT) tr += M[i * r + i] * Sv[i];
        tr = T.sum(tr);
        for (int it = tid; it < r * r; it += NT) Mc[it] = M[it];
        if (tid == 0) flag_s = 0;
        __syncthreads();
        chol_big(Mc, r, r, T1, (long long)n * r);
        ok = flag_s == 0;
        __syncthreads();
        if (!ok) { if (tid == 0) flag_s = 0; __syncthreads(); return __builtin_inf(); }
        double l = 0;
        for (int i = tid; i < r; i += NT) l += log(Mc[i * r + i]);
        l = T.sum(l);
        return 0.5 * (tr - 2.0 * l - logdetS - r);
    };
    if (closed) {
        // ---- closed form per group, X_e = (J_e Sigma J_e^T)^-1 (src/logdet_function.cpp:236-279),
        // one SPG_EDGE_MULTI record per group with more than one measurement (src/topology_provider_binary.hpp:48-67)
        const int *grp = reinterpret_cast<const int *>(ws + L.grp);
        const int ng = grp[0];
        double *Sig = Y, *otab = ws + L.otab;      // gauge route: Y already holds C^-1, which stands in for Sigma
        if (!gauge_ok) {
            for (int it = tid; it < n * r; it += NT) { const int c = it % r; T1[it] = U[it] * Sv[c]; }
            __syncthreads();
            if (big) team_gemm<NT>(T, Sig, n, T1, r, false, U, r, true, n, n, r, 0, false, panel);      // Sigma = (U S) U^T
            else
            for (int it = tid; it < n * n; it += NT) {
                const int i = it / n, j = it - i * n;
                if (j <= i) { double sacc = 0; for (int t = 0; t < r; t++) sacc += T1[i * r + t] * U[j * r + t]; Sig[i * n + j] = sacc; Sig[j * n + i] = sacc; }
            }
        }
        for (int it = tid; it < n * n; it += NT) Ai[it] = 0.0;
        __syncthreads();
        int64_t rel = 0;
        int nvt = 0;
        for (int gi = 0; gi < ng; gi++) {
            const int i0 = grp[1 + gi], nmg = grp[2 + gi] - i0, re = D * nmg;
            double *Je = ws + L.Sc, *G = Je + (int64_t)re * n, *C = G + re * re, *Wm = C + re * re, *X = Wm + re * re, *Tm = ws + L.A1;
            for (int it = tid; it < re * n; it += NT) Je[it] = 0.0;
            __syncthreads();
            for (int i = tid; i < nmg; i += NT) {
                const int oa = pairs[2 * (i0 + i)] * D, ob = pairs[2 * (i0 + i) + 1] * D;
                const double *Ja = Jb + (i0 + i) * 2 * DD, *Jbb = Ja + DD;
                for (int x = 0; x < D; x++) for (int y = 0; y < D; y++) {
                    double ja = Ja[x * D + y], jb = Jbb[x * D + y];
                    if (ng == 1 && nmg > 0) {      // a single measurement block goes through sparseJacobian(): entries below epsilon are dropped (:243-247, :335)
                        if (fabs(ja) < 2.220446049250313e-16) ja = 0.0;
                        if (fabs(jb) < 2.220446049250313e-16) jb = 0.0;
                    }
                    Je[(i * D + x) * n + oa + y] += ja;
                    Je[(i * D + x) * n + ob + y] += jb;
                }
            }
            __syncthreads();
            CFP(6);
            // A row of J_e has two d x d blocks (the measurement's two poses): the four products with J_e below walk those
            // blocks instead of the whole row — 12 terms instead of n per entry (dense on the matrix cores they were 110 ms of a
            // 190-vertex cluster). inc: the measurements at each kept vertex (LDS; the Newton solve's vector is idle here).
            int *incp = reinterpret_cast<int *>(colbuf), *inc = incp + 264, *icnt = inc + 528;
            if (big) {
                for (int v = tid; v <= k; v += NT) { incp[v] = 0; icnt[v] = 0; }
                __syncthreads();
                if (tid == 0) {
                    for (int i = 0; i < nmg; i++) { incp[pairs[2 * (i0 + i)] + 1]++; incp[pairs[2 * (i0 + i) + 1] + 1]++; }
                    for (int v = 0; v < k; v++) incp[v + 1] += incp[v];
                    for (int i = 0; i < nmg; i++)
                        for (int side = 0; side < 2; side++) { const int v = pairs[2 * (i0 + i) + side]; inc[incp[v] + icnt[v]++] = i; }
                }
                __syncthreads();
                for (int it = tid; it < re * n; it += NT) {           // Tm = J Sigma
                    const int pr = it / n, c = it - pr * n, i = pr / D;
                    const int oa = pairs[2 * (i0 + i)] * D, ob = pairs[2 * (i0 + i) + 1] * D;
                    double sacc = 0;
#pragma unroll
                    for (int y = 0; y < D; y++) sacc += Je[pr * n + oa + y] * Sig[(oa + y) * n + c] + Je[pr * n + ob + y] * Sig[(ob + y) * n + c];
                    Tm[it] = sacc;
                }
                __syncthreads();
            } else {
                for (int it = tid; it < re * n; it += NT) {
                    const int pr = it / n, c = it - pr * n;
                    double sacc = 0;
                    for (int t = 0; t < n; t++) sacc += Je[pr * n + t] * Sig[t * n + c];
                    Tm[it] = sacc;
                }
                __syncthreads();
            }
            CFP(7);
            if (big) {
                for (int it = tid; it < re * re; it += NT) {          // G = (J Sigma) J^T
                    const int pr = it / re, c = it - pr * re, j = c / D;
                    const int oa = pairs[2 * (i0 + j)] * D, ob = pairs[2 * (i0 + j) + 1] * D;
                    double sacc = 0;
#pragma unroll
                    for (int y = 0; y < D; y++) sacc += Tm[pr * n + oa + y] * Je[c * n + oa + y] + Tm[pr * n + ob + y] * Je[c * n + ob + y];
                    G[it] = sacc;
                }
                __syncthreads();
            } else {
                for (int it = tid; it < re * re; it += NT) {
                    const int pr = it / re, c = it - pr * re;
                    double sacc = 0;
                    for (int t = 0; t < n; t++) sacc += Tm[pr * n + t] * Je[c * n + t];
                    G[it] = sacc;
                }
                __syncthreads();
            }
            CFP(8);
            for (int it = tid; it < re * re; it += NT) { const int pr = it / re, c = it - pr * re; C[it] = 0.5 * (G[pr * re + c] + G[c * re + pr]); }
            if (tid == 0) flag_s = 0;
            __syncthreads();
            chol_big(C, re, re, Wm, (long long)re * re);
            if (flag_s) { status = SPG_ST_CLOSED_FORM_NOT_PD; n_new = 0; finish(); return; }
            CFP(9);
            spd_inverse_from_chol(C, Wm, X, re);            // W = C^-1 (lower): X = W^T W
            CFP(10);
            CFP(11);
            // A += J_e^T X J_e
            if (big) {
                for (int it = tid; it < re * n; it += NT) {           // Tm = X J
                    const int pr = it / n, c = it - pr * n, v = c / D;
                    double sacc = 0;
                    for (int e2 = incp[v]; e2 < incp[v + 1]; e2++) {
                        const int j = inc[e2];
#pragma unroll
                        for (int x = 0; x < D; x++) sacc += X[pr * re + j * D + x] * Je[(j * D + x) * n + c];
                    }
                    Tm[it] = sacc;
                }
                __syncthreads();
                for (int it = tid; it < n * n; it += NT) {            // A += J^T (X J)
                    const int r1 = it / n, c = it - r1 * n, v = r1 / D;
                    double sacc = 0;
                    for (int e2 = incp[v]; e2 < incp[v + 1]; e2++) {
                        const int j = inc[e2];
#pragma unroll
                        for (int x = 0; x < D; x++) sacc += Je[(j * D + x) * n + r1] * Tm[(j * D + x) * n + c];
                    }
                    Ai[it] += sacc;
                }
            } else {
                for (int it = tid; it < re * n; it += NT) {
                    const int pr = it / n, c = it - pr * n;
                    double sacc = 0;
                    for (int t = 0; t < re; t++) sacc += X[pr * re + t] * Je[t * n + c];
                    Tm[it] = sacc;
                }
                __syncthreads();
                for (int it = tid; it < n * n; it += NT) {
                    const int i = it / n, j = it - i * n;
                    double sacc = 0;
                    for (int t = 0; t < re; t++) sacc += Je[t * n + i] * Tm[t * n + j];
                    Ai[it] += sacc;
                }
            }
            CFP(12);
            // the record
            double *rec = arena + bd.new_off + rel;
            const double *zb = ws + L.zbuf + (int64_t)i0 * PS;
            if (nmg == 1) {
                for (int it = tid; it < PS; it += NT) rec[it] = zb[it];
                for (int it = tid; it < D * (D + 1) / 2; it += NT) {
                    int o = it, i = 0;
                    while (o >= D - i) { o -= D - i; i++; }
                    rec[PS + it] = X[i * re + i + o];
                }
                if (tid == 0) {
                    otab[4 * gi] = (double)SPG_EDGE_BINARY; otab[4 * gi + 1] = (double)rel; otab[4 * gi + 2] = (double)REC; otab[4 * gi + 3] = 2.0;
                    otab[4 * k + nvt] = (double)(m + pairs[2 * i0]); otab[4 * k + nvt + 1] = (double)(m + pairs[2 * i0 + 1]);
                }
                rel += REC; nvt += 2;
            } else {
                if (tid == 0) {
                    int *lv = reinterpret_cast<int *>(ws + L.ibuf) + 3 * k, nq = 0;
                    rec[0] = (double)nmg;
                    for (int i = 0; i < nmg; i++)
                        for (int side = 0; side < 2; side++) {
                            const int v = pairs[2 * (i0 + i) + side];
                            int at = -1;
                            for (int t = 0; t < nq; t++) if (lv[t] == v) at = t;
                            if (at < 0) { at = nq; lv[nq++] = v; }
                            rec[1 + 2 * i + side] = (double)at;
                        }
                    otab[4 * gi] = (double)SPG_EDGE_MULTI; otab[4 * gi + 1] = (double)rel; otab[4 * gi + 2] = (double)SPG_MULTI_LEN(D, nmg); otab[4 * gi + 3] = (double)nq;
                    for (int t = 0; t < nq; t++) otab[4 * k + nvt + t] = (double)(m + lv[t]);
                    si[2] = nq;
                }
                for (int it = tid; it < nmg * PS; it += NT) rec[1 + 2 * nmg + it] = zb[it];
                double *Wr = rec + 1 + 2 * nmg + nmg * PS;
                for (int it = tid; it < re * re; it += NT) Wr[it] = Wm[it];
                __syncthreads();
                rel += SPG_MULTI_LEN(D, nmg); nvt += si[2];
            }
            __syncthreads();
        }
        CFP(13);
        bool okc = false;
        double fv;
        if (gauge_ok) {
            // 1/2 (tr(C^-1 A) - log det(A + N^ N^^T) + log det C - r)
            double tr = 0;
            for (int it = tid; it < n * n; it += NT) tr += Sig[it] * Ai[it];
            tr = T.sum(tr);
            for (int it = tid; it < n * n; it += NT) {
                const int i = it / n, j = it - i * n;
                double sacc = Ai[it];
                for (int qq = 0; qq < D; qq++) sacc += Ng[i * D + qq] * Ng[j * D + qq];
                A1[it] = sacc;
            }
            if (tid == 0) flag_s = 0;
            __syncthreads();
            chol_big(A1, n, n, ws + L.V, (long long)n * n);
            okc = flag_s == 0;
            __syncthreads();
            if (tid == 0) flag_s = 0;
            __syncthreads();
            double l = 0;
            for (int i = tid; i < n; i += NT) l += log(A1[i * n + i]);
            l = T.sum(l);
            fv = okc ? 0.5 * (tr - 2.0 * l + logdetC - r) : __builtin_inf();
        } else fv = big ? value_from_A_cluster(okc) : value_from_A(okc);
        CFP(14);
#ifdef SPG_CF_PROF
        if (tid == 0 && k >= 40) printf("cf prof k=%d m=%d n=%d gauge=%d (us): assemble %lld | Hmm chol+W %lld | schur %lld | pattern: inverse %lld weights %lld sort %lld rest %lld | skeleton+gauge %lld | eig %lld | group: zero+Je %lld, J Sig %lld, G %lld, chol %lld, tri inv %lld, gram %lld, A+= %lld | record %lld | value %lld\n",
                                        k, m, n, (int)gauge_ok, cfp[0] / 100, cfp[1] / 100, cfp[2] / 100, cfp[15] / 100, cfp[16] / 100, cfp[17] / 100, cfp[3] / 100, cfp[4] / 100, cfp[5] / 100, cfp[6] / 100, cfp[7] / 100, cfp[8] / 100, cfp[9] / 100, cfp[10] / 100, cfp[11] / 100, cfp[12] / 100, cfp[13] / 100, cfp[14] / 100);
#endif
        tab_mode = true;
        n_new = ng;
        if (!okc || !isfinite(fv)) { status = SPG_ST_KLD_NOT_PD; kld = __builtin_nan(""); finish(); return; }
        kld = fv;
        finish();
        return;
    }
    // JU = sparseJacobian * U  (q x r): row (e, p) = Ja[p,:] U[a-block,:] + Jb[p,:] U[b-block,:]
    double *JU = hot + L.JU;
    for (int it = tid; it < q * r; it += NT) {
        const int row = it / r, c = it - row * r, e = row / D, p = row - e * D;
        const int oa = pairs[2 * e] * D, ob = pairs[2 * e + 1] * D;
        const double *Ja = Jb + e * 2 * DD, *Jbb = Ja + DD;
        double s = 0;
        for (int t = 0; t < D; t++) {
            const double ja = Ja[p * D + t], jb = Jbb[p * D + t];
            if (fabs(ja) >= 2.220446049250313e-16) s += ja * U[(oa + t) * r + c];
            if (fabs(jb) >= 2.220446049250313e-16) s += jb * U[(ob + t) * r + c];
        }
        JU[it] = s;
    }
    __syncthreads();

    // ---- the function: value(xv) leaves chol(M) in Mc; gradient(xv, gv) leaves M^-1 in Mi and the X_e^-1 in Xi
    double *P = hot + L.P, *T2 = hot + L.T2, *Hx = ws + L.Hx, *Xi = hot + L.Xi;
    const int ldh = nx + 8;       // row stride of the Hessian / its factor in the workspace
    double *x = hot + L.x, *xn = hot + L.xn, *g = hot + L.g, *gn = hot + L.gn, *dv = hot + L.dv;
    double rho = 0;
    // symmetric view of block e of xv from its lower triangle (column-major): X(i, j), i >= j, at xv[e DD + j D + i]
    auto Xat = [&](const double *xv, int e, int i, int j) { return (i >= j) ? xv[e * DD + j * D + i] : xv[e * DD + i * D + j]; };
    auto base_value = [&](const double *xv, bool &ok) -> double {
        // A = J^T X J (upper triangle accumulated, then mirrored), one (edge, block pair) product at a time
        for (int it = tid; it < n * n; it += NT) Ai[it] = 0.0;
        __syncthreads();
        for (int e = 0; e < E; e++) {
            const int oa = pairs[2 * e] * D, ob = pairs[2 * e + 1] * D;
            const double *Ja = Jb + e * 2 * DD, *Jbb = Ja + DD;
            // T = X [Ja | Jb] (D x 2D) in T2, then [Ja | Jb]^T T into A
            for (int it = tid; it < D * 2 * D; it += NT) {
                const int p = it / (2 * D), c = it - p * 2 * D;
                const double *Jc = (c < D) ? Ja : Jbb;
                const int cc = (c < D) ? c : c - D;
                double s = 0;
                for (int t = 0; t < D; t++) s += Xat(xv, e, p, t) * Jc[t * D + cc];
                T2[it] = s;
            }
            __syncthreads();
            for (int it = tid; it < 4 * DD; it += NT) {
                const int rr = it / (2 * D), c = it - rr * 2 * D;
                const double *Jr = (rr < D) ? Ja : Jbb;
                const int r2 = (rr < D) ? rr : rr - D;
                double s = 0;
                for (int p = 0; p < D; p++) s += Jr[p * D + r2] * T2[p * 2 * D + c];
                const int gi = ((rr < D) ? oa : ob) + r2, gj = ((c < D) ? oa : ob) + ((c < D) ? c : c - D);
                // the reference accumulates block (lower offset, higher offset) only and mirrors the upper triangle:
                // an entry below the diagonal of the whole matrix is the transpose of one above
                if (gi <= gj) Ai[gi * n + gj] += s;
            }
            __syncthreads();
        }
        for (int it = tid; it < n * n; it += NT) { const int i = it / n, j = it - i * n; if (j < i) Ai[it] = Ai[j * n + i]; }
        __syncthreads();
        IPT(11);
        return value_from_A(ok);
    };
    bool chol_ok = false;
    auto value = [&](const double *xv) -> double {
        double f = base_value(xv, chol_ok);
        // barrier: - rho * sum_e log det X_e, +inf when a block is not positive definite
        double pen = 0;
        for (int e = tid; e < E; e += NT) {
            double B[DD], rB[D];
#pragma unroll
            for (int i = 0; i < D; i++)
#pragma unroll
                for (int j = 0; j < D; j++) B[i * D + j] = Xat(xv, e, i, j);
            if (!chol_static<D>(B, rB)) pen = __builtin_inf();
            else {
                double l = 0;
#pragma unroll
                for (int i = 0; i < D; i++) l += log(B[i * D + i]);
                pen += 2.0 * l;
            }
        }
        // deterministic: per-edge terms summed in edge order by thread 0
        red[tid] = pen;
        __syncthreads();
        if (tid == 0) { double s = 0; for (int e = 0; e < E && e < NT; e++) s += red[e]; sc[1] = s; }
        __syncthreads();
        const double ps = sc[1];
        __syncthreads();
        if (!isfinite(f)) return __builtin_inf();
        if (!isfinite(ps)) return __builtin_inf();
        return f - rho * ps;
    };
    auto gradient = [&](const double *xv, double *gv) {
        if (!chol_ok) { for (int it = tid; it < nx; it += NT) gv[it] = 0.0; __syncthreads(); return; }
        if (r <= 32) tri_inverse_lower_reg<32>(tid, Mc, Li, r, r);
        else tri_inverse_lower<NT>(T, Mc, Li, r, r);
        IPT(5);
        gram_lower_inverse<NT>(T, Li, Mi, r, r);           // xinv = M^-1
        IPT(12);
        // Y = U (diag(S) - xinv) U^T
        for (int it = tid; it < n * r; it += NT) {
            const int i = it / r, c = it - i * r;
            double s = 0;
#pragma unroll 8
            for (int t = 0; t < r; t++) s += U[i * r + t] * ((t == c ? Sv[t] : 0.0) - Mi[t * r + c]);
            T1[it] = s;
        }
        __syncthreads();
        for (int it = tid; it < n * n; it += NT) {
            const int i = it / n, j = it - i * n;
            double s = 0;
#pragma unroll 8
            for (int t = 0; t < r; t++) s += T1[i * r + t] * U[j * r + t];
            Y[it] = s;
        }
        __syncthreads();
        IPT(13);
        // per edge: block = sym(Ja Yaa Ja^T) + sym(Jb Ybb Jb^T) + (Ja Yab Jb^T + its transpose); g_e = block / 2
        for (int it = tid; it < nx; it += NT) {
            const int e = it / DD, jj = (it - e * DD) / D, ii = it - e * DD - jj * D;      // column-major (ii, jj)
            const int oa = pairs[2 * e] * D, ob = pairs[2 * e + 1] * D;
            const double *Ja = Jb + e * 2 * DD, *Jbb = Ja + DD;
            auto quad = [&](const double *J1, int o1, const double *J2, int o2, int i1, int i2) {
                double s = 0;
                for (int u = 0; u < D; u++) {
                    double t = 0;
                    for (int v = 0; v < D; v++) t += Y[(o1 + u) * n + o2 + v] * J2[i2 * D + v];
                    s += J1[i1 * D + u] * t;
                }
                return s;
            };
            const double aa = 0.5 * (quad(Ja, oa, Ja, oa, ii, jj) + quad(Ja, oa, Ja, oa, jj, ii));
            const double bb = 0.5 * (quad(Jbb, ob, Jbb, ob, ii, jj) + quad(Jbb, ob, Jbb, ob, jj, ii));
            const double ab = quad(Ja, oa, Jbb, ob, ii, jj) + quad(Ja, oa, Jbb, ob, jj, ii);
            gv[it] = 0.5 * (aa + bb + ab);
        }
        __syncthreads();
        IPT(14);
        // constraint part: X_e^-1, g_e -= rho X_e^-1
        if (tid == 0) si[0] = 0;
        __syncthreads();
        for (int e = tid; e < E; e += NT) {
            double B[DD], Inv[DD], rB[D];
#pragma unroll
            for (int i = 0; i < D; i++)
#pragma unroll
                for (int j = 0; j < D; j++) B[i * D + j] = Xat(xv, e, i, j);
            if (!chol_static<D>(B, rB)) { si[0] = 1; continue; }
            // solve L L^T Inv = I column by column (all in registers: every index is a compile-time constant)
#pragma unroll
            for (int c = 0; c < D; c++) {
                double y[D];
#pragma unroll
                for (int i = 0; i < D; i++) {
                    double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int t = 0; t < i; t++) s -= B[i * D + t] * y[t];
                    y[i] = div_by(s, B[i * D + i], rB[i]);
                }
#pragma unroll
                for (int i = D - 1; i >= 0; i--) {
                    double s = y[i];
#pragma unroll
                    for (int t = i + 1; t < D; t++) s -= B[t * D + i] * Inv[t * D + c];
                    Inv[i * D + c] = div_by(s, B[i * D + i], rB[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < DD; i++) Xi[e * DD + i] = Inv[i];      // row-major (u, v)
        }
        __syncthreads();
        if (si[0]) { for (int it = tid; it < nx; it += NT) gv[it] = 0.0; __syncthreads(); return; }
        for (int it = tid; it < nx; it += NT) {
            const int e = it / DD, jj = (it - e * DD) / D, ii = it - e * DD - jj * D;
            gv[it] -= rho * Xi[e * DD + ii * D + jj];
        }
        __syncthreads();
    };
    auto hessian = [&]() {
        // P = sym(JU xinv JU^T)
        for (int it = tid; it < q * r; it += NT) {
            const int i = it / r, c = it - i * r;
            double s = 0;
#pragma unroll 8
            for (int t = 0; t < r; t++) s += JU[i * r + t] * Mi[t * r + c];
            T2[it] = s;
        }
        __syncthreads();
        for (int it = tid; it < q * q; it += NT) {
            const int i = it / q, j = it - i * q;
            double s = 0;
#pragma unroll 8
            for (int t = 0; t < r; t++) s += T2[i * r + t] * JU[j * r + t];
            P[it] = s;
        }
        __syncthreads();
        for (int it = tid; it < q * q; it += NT) { const int i = it / q, j = it - i * q; if (j > i) { const double v = 0.5 * (P[it] + P[j * q + i]); P[it] = v; } }
        __syncthreads();
        for (int it = tid; it < q * q; it += NT) { const int i = it / q, j = it - i * q; if (j < i) P[it] = P[j * q + i]; }
        __syncthreads();
        // H[(e, ii, jj)][(e2, uu, vv)] = P(e2 D + uu, e D + ii) P(e D + jj, e2 D + vv)  (+ rho Xinv(uu, ii) Xinv(jj, vv) on e2 = e)
        if (hx_tiled || hx_tiled2) return;       // chol_solve_tiled builds its tiles from P and Xi
        if (hx_lds) {
            // lower triangle only, packed by rows in LDS: row s at s (s + 1) / 2
            for (int s0 = tid; s0 < nx; s0 += NT) {
                // (rows are dealt round-robin from both ends so that the triangular row lengths balance)
                const int sidx = (s0 & 1) ? (nx - 1 - (s0 >> 1)) : (s0 >> 1);
                const int e = sidx / DD, jj = (sidx - e * DD) / D, ii = sidx - e * DD - jj * D;
                double *row = lds_pool + (long long)sidx * (sidx + 1) / 2;
                int e2 = 0, vv = 0, uu = 0;
                for (int t = 0; t <= sidx; t++) {
                    double v = P[(e2 * D + uu) * q + e * D + ii] * P[(e * D + jj) * q + e2 * D + vv];
                    if (e2 == e && chol_ok) v += rho * Xi[e * DD + uu * D + ii] * Xi[e * DD + jj * D + vv];
                    row[t] = v;
                    if (++uu == D) { uu = 0; if (++vv == D) { vv = 0; e2++; } }
                }
            }
            __syncthreads();
            return;
        }
        // (the lower triangle only: nothing reads the rest; (s, t) walk the matrix without a division)
        for (int s = tid / nx, t = tid - (tid / nx) * nx; s < nx; ) {
            if (t <= s) {
                const int e = s / DD, jj = (s - e * DD) / D, ii = s - e * DD - jj * D;
                const int e2 = t / DD, vv = (t - e2 * DD) / D, uu = t - e2 * DD - vv * D;
                double v = P[(e2 * D + uu) * q + e * D + ii] * P[(e * D + jj) * q + e2 * D + vv];
                if (e2 == e && chol_ok) v += rho * Xi[e * DD + uu * D + ii] * Xi[e * DD + jj * D + vv];
                Hx[(long long)s * ldh + t] = v;
            }
            t += NT;
            while (t >= nx) { t -= nx; s++; }
        }
        __syncthreads();
    };
    // (A) Cholesky of the packed Hessian in LDS, right-looking, the scaled column copied to colbuf so that the trailing update
    // walks rows only; the right-hand side is row nn. Three barriers per column, no global memory. Then L^T x = y by one
    // wavefront (right-looking over the rows of L, no barriers). Same operations in the same order as the out-of-L2 version.
    auto chol_solve_packed = [&](int nn, const double *rhs_neg, double *out) -> bool {
        double *Hp = lds_pool;
        auto rowp = [&](int i) { return Hp + (long long)i * (i + 1) / 2; };
        for (int it = tid; it < nn; it += NT) rowp(nn)[it] = rhs_neg[it];
        __syncthreads();
        bool okc = true;
        for (int j = 0; j < nn; j++) {
            double dpiv = rowp(j)[j];
            if (!(dpiv > 0.0) || !isfinite(dpiv)) { okc = false; dpiv = 1.0; }
            const double l = sqrt(dpiv);
            __syncthreads();
            for (int i = j + 1 + tid; i <= nn; i += NT) { const double v = rowp(i)[j] / l; rowp(i)[j] = v; colbuf[i] = v; }
            if (tid == 0) rowp(j)[j] = l;
            __syncthreads();
            for (int i = j + 1 + (tid >> 6); i <= nn; i += NT / 64) {
                const double li = colbuf[i];
                double *row = rowp(i);
                const int cmax = min(i, nn - 1);
                for (int c = j + 1 + (tid & 63); c <= cmax; c += 64) row[c] -= li * colbuf[c];
            }
            __syncthreads();
        }
        if (!okc) return false;
        for (int it = tid; it < nn; it += NT) colbuf[it] = rowp(nn)[it];
        __syncthreads();
        if (tid < 64) {
            const int lane = tid;
            for (int i = nn - 1; i >= 0; i--) {
                const double *row = rowp(i);
                const double xi = colbuf[i] / row[i];
                for (int t = lane; t < i; t += 64) colbuf[t] -= row[t] * xi;
                if (lane == 0) colbuf[i] = xi;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int it = tid; it < nn; it += NT) out[it] = colbuf[it];
        __syncthreads();
        return true;
    };
    // L^T x = y for a factor in packed LDS rows (PACKED) or in the workspace square, nn <= 64 CH: see chol_solve_tiled
    auto solve_reg = [&](auto ch_c, auto packed_c, int nn, double *out) {
        constexpr int CH = decltype(ch_c)::value;
        constexpr bool PACKED = decltype(packed_c)::value;
        // L^T x = y (y = row nn) by one wavefront, right-looking over the rows of L from the last: the running vector lives in
        // registers (entries lane, lane + 64, ...), x_i is computed by every lane from lane reads, rows are fetched three
        // ahead (they come from L2 when two tiles per thread are in use). Same operations as the LDS version in (A).
        if (tid < 64) {
            const int lane = tid;
            const int nns = __builtin_amdgcn_readfirstlane(nn);       // the row loop runs on the scalar unit
            const int lds_ = __builtin_amdgcn_readfirstlane(ldh);
            const double *Lb = PACKED ? lds_pool : Hx;
            auto srow = [&](int i) { return PACKED ? Lb + (long long)i * (i + 1) / 2 : Lb + (long long)i * lds_; };
            double y[CH], dg[CH], rd[CH];       // the running vector, the diagonal of L and its reciprocals: entry u = lane + 64 c
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const int u = lane + 64 * c;
                y[c] = (u < nns) ? srow(nns)[u] : 0.0;
                dg[c] = (u < nns) ? srow(u)[u] : 1.0;
                rd[c] = 1.0 / dg[c];
            }
            // chunks 0..CM of row i (entries beyond the diagonal are read as the diagonal entry and never used: no branches)
            auto fetch = [&](int i, double *dst, auto cm_c) {
                constexpr int CM = decltype(cm_c)::value;
                const int ic = max(i, 0);
                const double *row = srow(ic);
#pragma unroll
                for (int c = 0; c <= CM; c++) dst[c] = row[min(lane + 64 * c, ic)];
            };
            // row i of slot SL = i >> 6 (compile time): x_i from its owner's y, then y_u -= L[i][u] x_i for u < i
            auto apply = [&](int i, const double *cur, auto sl_c) {
                constexpr int SL = decltype(sl_c)::value;
                const int owner = i & 63;
                const double xi = div_by(read_lane(y[SL], owner), read_lane(dg[SL], owner), read_lane(rd[SL], owner));   // every lane computes the same x_i
#pragma unroll
                for (int c = 0; c < SL; c++) y[c] -= cur[c] * xi;
                const double upd = y[SL] - cur[SL] * xi;
                y[SL] = (lane < owner) ? upd : (lane == owner ? xi : y[SL]);
            };
            auto run_slot = [&](auto sl_c, int &i) {
                constexpr int SL = decltype(sl_c)::value;
                if ((i >> 6) != SL) return;
                double r0[CH], r1[CH], r2[CH], r3[CH];
                // rows down to a multiple of four, one at a time; then four rows in flight down to the slot's first row
                while (((i + 1) & 3) != 0) { fetch(i, r0, sl_c); apply(i, r0, sl_c); i--; }
                if ((i >> 6) != SL) return;
                fetch(i, r0, sl_c); fetch(i - 1, r1, sl_c); fetch(i - 2, r2, sl_c); fetch(i - 3, r3, sl_c);
                for (; i >= 64 * SL; i -= 4) {
                    apply(i, r0, sl_c);     fetch(i - 4, r0, sl_c);
                    apply(i - 1, r1, sl_c); fetch(i - 5, r1, sl_c);
                    apply(i - 2, r2, sl_c); fetch(i - 6, r2, sl_c);
                    apply(i - 3, r3, sl_c); fetch(i - 7, r3, sl_c);
                }
            };
            int i = nns - 1;
            auto run_all = [&](auto self, auto sl_c, int &ii) -> void {
                constexpr int SL = decltype(sl_c)::value;
                run_slot(sl_c, ii);
                if constexpr (SL > 0) self(self, std::integral_constant<int, SL - 1>{}, ii);
            };
            run_all(run_all, std::integral_constant<int, CH - 1>{}, i);
#pragma unroll
            for (int c = 0; c < CH; c++) { const int u = lane + 64 * c; if (u < nns) out[u] = y[c]; }
        }
        __syncthreads();
    };
    // (A') the same factorisation with the running matrix in REGISTERS: thread t owns the 8 x 8 tile (I, C), C <= I, of the
    // lower triangle (rows incl. the right-hand side row nn), 253 tiles at most, so nn + 1 <= 176. Per block column J:
    //   1. the owners of column J's tiles put their running values into the LDS panel (row stride 9, block stride 73:
    //      rows and blocks spread over the banks);
    //   2. thread u takes row (J + 1) 8 + u: it factorises the diagonal block itself (every thread does, redundantly — the
    //      sqrt/divide chain is the sequential part anyway and this saves a barrier and a broadcast) and solves its row
    //      against it; the finished values go to the panel in place and to the packed matrix;
    //   3. every tile right of J takes its 8 x 8 x 8 products from two panel blocks.
    // Three barriers per EIGHT columns (the column version above: three per column, and a dependent LDS round trip per
    // entry and column — 1.4 M cycles for 144 variables; this one: see profiles/r02_interior_point.md). Every entry still
    // has its products subtracted one at a time in column order and is divided by the pivot: bit-identical to (A) and (B).
    // Beyond 175 variables (up to 247) each thread owns TWO tiles and the finished factor goes to the
    // L2 workspace instead of LDS (217 packed rows would be 189 KB); the substitution reads it back three rows ahead.
    constexpr int TR = 9, TB = 73;
    auto chol_solve_tiled = [&](auto ntl_c, int nn, const double *rhs_neg, double *out) -> bool {
        constexpr int NTL = decltype(ntl_c)::value;                       // tiles per thread
        constexpr int CH = (NTL == 1) ? 3 : 4;                            // 64-entry chunks of the solution vector
        double *pan = (NTL == 1) ? colbuf : panel;
        double *Lb = (NTL == 1) ? lds_pool : Hx;                          // NTL == 1: packed rows in LDS; else nn-strided rows in L2
        auto rowp = [&](int i) { return (NTL == 1) ? Lb + (long long)i * (i + 1) / 2 : Lb + (long long)i * ldh; };
        const int nr = nn + 1, nb = (nr + 7) >> 3;
        int I[NTL], C[NTL];
        double t[NTL][8][8];
#pragma unroll
        for (int s = 0; s < NTL; s++) {
            // slot 0: the tiles of the last 22 block columns (at most 253: one per thread — from block column c0 on only this
            // slot has work, a single tile update per step); slot 1: the tiles of the first c0 block columns, column by column
            const int c0 = (NTL == 1) ? 0 : max(nb - 22, 0), nl = nb - c0;
            I[s] = -1; C[s] = -1;
            if (s == 0) {
                if (tid < nl * (nl + 1) / 2) {
                    int i = (int)((sqrt(8.0 * tid + 1.0) - 1.0) * 0.5);
                    while ((i + 1) * (i + 2) / 2 <= tid) i++;
                    while (i * (i + 1) / 2 > tid) i--;
                    I[s] = c0 + i; C[s] = c0 + tid - i * (i + 1) / 2;
                }
            } else {
                int off = 0;
                for (int c = 0; c < c0; c++) {
                    if (tid >= off && tid < off + nb - c) { C[s] = c; I[s] = c + tid - off; }
                    off += nb - c;
                }
            }
            // the tile of the Hessian (the expression of hessian() above, entry by entry) or of the right-hand side row
            int ra[8], rb[8], rx1[8], rx2[8], re[8], cc[8], cd[8], cu[8], ce[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int sidx = min(max(I[s], 0) * 8 + r, nn - 1);
                const int e = sidx / DD, jj = (sidx - e * DD) / D, ii = sidx - e * DD - jj * D;
                re[r] = e; ra[r] = e * D + ii; rb[r] = (e * D + jj) * q; rx1[r] = e * DD + ii; rx2[r] = e * DD + jj * D;
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int tidx = min(max(C[s], 0) * 8 + c, nn - 1);
                const int e2 = tidx / DD, vv = (tidx - e2 * DD) / D, uu = tidx - e2 * DD - vv * D;
                ce[c] = e2; cc[c] = (e2 * D + uu) * q; cd[c] = e2 * D + vv; cu[c] = uu * D + (vv << 16);
            }
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int row = I[s] * 8 + r, col = C[s] * 8 + c;
                    double v = 0.0;
                    if (I[s] >= 0 && col < nn && col <= row) {
                        if (row < nn) {
                            v = P[cc[c] + ra[r]] * P[rb[r] + cd[c]];
                            const double vb = __builtin_fma(rho * Xi[rx1[r] + (cu[c] & 0xffff)], Xi[rx2[r] + (cu[c] >> 16)], v);   // v += rho Xi Xi, without a branch
                            v = (ce[c] == re[r] && chol_ok) ? vb : v;
                        } else if (row == nn) v = rhs_neg[col];
                    }
                    t[s][r][c] = v;
                }
        }
        bool okc = true;
        for (int J = 0; J < nb; J++) {
#pragma unroll
            for (int s = 0; s < NTL; s++)
                if (C[s] == J) {
                    double *pb = pan + I[s] * TB;
#pragma unroll
                    for (int r = 0; r < 8; r++)
#pragma unroll
                        for (int c = 0; c < 8; c++) pb[r * TR + c] = t[s][r][c];
                }
            __syncthreads();
            IPT(2);
            {
                double d[8][8], rl[8] = {1, 1, 1, 1, 1, 1, 1, 1};
                const double *pj = pan + J * TB;
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int c = 0; c <= r; c++) d[r][c] = pj[r * TR + c];
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    if (J * 8 + jj < nn) {
                        double dpiv = d[jj][jj];
                        if (!(dpiv > 0.0) || !isfinite(dpiv)) { okc = false; dpiv = 1.0; }
                        const double l = sqrt(dpiv);
                        d[jj][jj] = l;
                        rl[jj] = 1.0 / l;
#pragma unroll
                        for (int r = jj + 1; r < 8; r++) d[r][jj] = div_by(d[r][jj], l, rl[jj]);
#pragma unroll
                        for (int c = jj + 1; c < 8; c++)
#pragma unroll
                            for (int r = c; r < 8; r++) d[r][c] -= d[r][jj] * d[c][jj];
                    }
                }
                const int row = (J + 1) * 8 + tid;
                if (row < nr) {
                    double *pr = pan + (row >> 3) * TB + (row & 7) * TR;
                    double *hr = rowp(row) + J * 8;
                    double v[8];
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) v[jj] = pr[jj];
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) {
#pragma unroll
                        for (int u = 0; u < jj; u++) v[jj] -= v[u] * d[jj][u];
                        v[jj] = div_by(v[jj], d[jj][jj], rl[jj]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) { pr[jj] = v[jj]; hr[jj] = v[jj]; }
                } else if (tid == NT - 1) {
                    // the finished diagonal block (rows beyond nn and columns beyond nn - 1 do not exist in the factor)
#pragma unroll
                    for (int r = 0; r < 8; r++)
#pragma unroll
                        for (int c = 0; c <= r; c++) {
                            const int rw = J * 8 + r, cl = J * 8 + c;
                            if (rw < nr && cl < nn) rowp(rw)[cl] = d[r][c];
                        }
                }
            }
            __syncthreads();
            IPT(6);
#pragma unroll
            for (int s = 0; s < NTL; s++)
                if (C[s] > J) {
                    const double *pi = pan + I[s] * TB, *pc = pan + C[s] * TB;
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) {
                        double li[8], lc[8];
#pragma unroll
                        for (int r = 0; r < 8; r++) { li[r] = pi[r * TR + jj]; lc[r] = pc[r * TR + jj]; }
#pragma unroll
                        for (int r = 0; r < 8; r++)
#pragma unroll
                            for (int c = 0; c < 8; c++) t[s][r][c] -= li[r] * lc[c];
                    }
                }
            __syncthreads();        // the panel is rewritten by the next block column's step 1
        }
        IPT(2);
        if (!okc) return false;
        if constexpr (NTL == 1) solve_reg(std::integral_constant<int, 3>{}, std::true_type{}, nn, out);
        else solve_reg(std::integral_constant<int, 4>{}, std::false_type{}, nn, out);
        IPT(3);
        return true;
    };
    // (B) Cholesky of the Hessian (Eigen::LLT in the reference, src/pqn/pqn_optimizer.cpp:52-53), lower, in place, out of L2.
    // Blocked right-looking: a panel of PB columns is factorised in LDS, then the trailing matrix takes ONE pass of
    // read-modify-writes per panel (rows are walked, four rows in flight per wavefront so that the L2 round trips overlap:
    // a column-at-a-time version spent ~1 us of load latency per row and column, 5 ms per factorisation at 216^2).
    // Every entry still has its products subtracted one by one in column order, as the unblocked algorithm does.
    // Returns false (uniformly) on a non-positive pivot.
    // The right-hand side travels as row nn of the matrix (A has nn + 1 rows of nn columns): after the factorisation that
    // row holds L^-1 rhs — the forward substitution comes out of the trailing updates for free.
    auto chol_rows = [&](double *A, int nn) -> bool {
        const int PB = max(1, min(16, kPanelDoubles / (nn + 1) - 1));   // panel columns: nn + 1 rows of PB + 1 doubles fit the LDS panel
        const int PBS = PB + 1;                                           // row stride (odd for the usual PB = 16: column walks conflict-free)
        const int nr = nn + 1;                                            // rows incl. the right-hand side
        bool okc = true;
        for (int j0 = 0; j0 < nn; j0 += PB) {
            const int pb = min(PB, nn - j0), nrows = nr - j0;
            IPT(2);
            for (int rr = tid >> 4; rr < nrows; rr += NT / 16) {           // panel rows j0.., columns j0..j0+pb (lower part)
                const int pc = tid & 15;
                if (pc < pb) panel[rr * PBS + pc] = (pc <= rr) ? A[(long long)(j0 + rr) * ldh + j0 + pc] : 0.0;
            }
            __syncthreads();
            IPT(6);
            for (int pc = 0; pc < pb; pc++) {
                double dpiv = panel[pc * PBS + pc];
                if (!(dpiv > 0.0) || !isfinite(dpiv)) { okc = false; dpiv = 1.0; }
                const double l = sqrt(dpiv);
                __syncthreads();                                     // everybody has read the pivot
                for (int rr = pc + 1 + tid; rr < nrows; rr += NT) panel[rr * PBS + pc] /= l;
                if (tid == 0) panel[pc * PBS + pc] = l;
                __syncthreads();
                const int p2 = pc + 1 + (tid & 15);
                if (p2 < pb)
                    for (int rr = pc + 1 + (tid >> 4); rr < nrows; rr += NT / 16)
                        if (rr >= p2) panel[rr * PBS + p2] -= panel[rr * PBS + pc] * panel[p2 * PBS + pc];
                __syncthreads();
            }
            IPT(7);
            for (int rr = tid >> 4; rr < nrows; rr += NT / 16) {
                const int pc = tid & 15;
                if (pc < pb && pc <= rr) A[(long long)(j0 + rr) * ldh + j0 + pc] = panel[rr * PBS + pc];
            }
            // trailing update: rows i >= j0 + pb (the right-hand side row included), columns j0 + pb <= c <= min(i, nn - 1)
            const int t0 = j0 + pb, wv = tid >> 6, lane = tid & 63;
            for (int cb = t0 + lane; cb < nn; cb += 64) {
                double pcv[16];
                const double *pcp = panel + (cb - j0) * PBS;
#pragma unroll
                for (int pc = 0; pc < 16; pc++) pcv[pc] = (pc < pb) ? pcp[pc] : 0.0;
                // this lane's column cb; rows are uniform over the wavefront (coalesced row segments), starting at the first
                // column of this 64-column chunk (rows above it lie over the diagonal for every lane), RIF rows in flight
                for (int ib = (cb - lane) + RIF * wv; ib < nr; ib += 4 * RIF) {
                    double v[RIF];
#pragma unroll
                    for (int u = 0; u < RIF; u++) { const int i = ib + u; v[u] = (i < nr && cb <= i) ? A[(long long)i * ldh + cb] : 0.0; }
#pragma unroll
                    for (int u = 0; u < RIF; u++) {
                        const int i = ib + u;
                        if (i < nr && cb <= i) {
                            const double *piv = panel + (i - j0) * PBS;
                            double acc = v[u];
#pragma unroll
                            for (int pc = 0; pc < 16; pc++) if (pc < pb) acc -= piv[pc] * pcv[pc];
                            A[(long long)i * ldh + cb] = acc;
                        }
                    }
                }
            }
            __syncthreads();
        }
        return okc;
    };
    // out = L^-T y with y = row nn of the factorised matrix (= L^-1 rhs, see chol_rows), by ONE wavefront without
    // barriers: the running vector lives in LDS, a row of L is one coalesced read (the next row is fetched while the
    // current one is applied), right-looking: x_i = y_i / L_ii, then y_t -= L[i][t] x_i over the row.
    // (the running vector: the static LDS array, or — path (C), more than 2 048 variables possible — the dynamic LDS; chosen at
    //  compile time so that the accesses stay LDS instructions: through a pointer argument they became flat ones and the
    //  streamed sizes that use this substitution lost 30 %)
    auto solve_rows = [&](const double *Lc, int nn, double *out, auto big_c) {
        double *const colbuf = decltype(big_c)::value ? lds_pool : colbuf_s;
        for (int it = tid; it < nn; it += NT) colbuf[it] = Lc[(long long)nn * ldh + it];
        __syncthreads();
        if (tid < 64) {
            const int lane = tid;
            constexpr int CH = 4;                       // columns [0, 256) of a row travel in registers
            double cur[CH], nxt[CH];
            auto fetch = [&](int i, double *dst) {
                const double *row = Lc + (long long)i * ldh;
#pragma unroll
                for (int c = 0; c < CH; c++) { const int t = lane + 64 * c; dst[c] = (t <= i && t < nn) ? row[t] : 0.0; }
            };
            fetch(nn - 1, cur);
            for (int i = nn - 1; i >= 0; i--) {
                if (i > 0) fetch(i - 1, nxt);
                const double *row = Lc + (long long)i * ldh;
                const int owner = i & 63, slot = i >> 6;
                double dii = (slot < CH) ? 0.0 : row[i];
#pragma unroll
                for (int c = 0; c < CH; c++) if (slot == c) dii = __shfl(cur[c], owner, 64);
                const double xi = colbuf[i] / dii;
#pragma unroll
                for (int c = 0; c < CH; c++) { const int t = lane + 64 * c; if (t < i) colbuf[t] -= cur[c] * xi; }
                for (int t = 64 * CH + lane; t < i; t += 64) colbuf[t] -= row[t] * xi;
                if (lane == 0) colbuf[i] = xi;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < CH; c++) cur[c] = nxt[c];
            }
        }
        __syncthreads();
        for (int it = tid; it < nn; it += NT) out[it] = colbuf[it];
        __syncthreads();
    };
    // (B') the tiled factorisation of (A') for matrices that do not fit the registers: the running matrix stays in the L2
    // workspace (as in (B)) and a thread takes the 8 x 8 tiles right of the block column one after the other — load, 512
    // FMAs from two panel blocks, store. Same steps 1-3, same operations in the same order; the panel (block stride 73)
    // must fit its 64 KB of LDS: up to 895 variables. (B) swept the trailing matrix with one LDS read per FMA.
    auto chol_streamed = [&](double *A, int nn) -> bool {
        double *pan = panel;
        const int nr = nn + 1, nb = (nr + 7) >> 3;
        bool okc = true;
        for (int J = 0; J < nb; J++) {
            const int rb0 = J * 8;
            for (int it = tid; it < (nb * 8 - rb0) * 8; it += NT) {
                const int row = rb0 + (it >> 3), c = it & 7, col = rb0 + c;
                pan[(row >> 3) * TB + (row & 7) * TR + c] = (row < nr && col < nn && col <= row) ? A[(long long)row * ldh + col] : 0.0;
            }
            __syncthreads();
            IPT(2);
            {
                double d[8][8], rl[8] = {1, 1, 1, 1, 1, 1, 1, 1};
                const double *pj = pan + J * TB;
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int c = 0; c <= r; c++) d[r][c] = pj[r * TR + c];
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    if (J * 8 + jj < nn) {
                        double dpiv = d[jj][jj];
                        if (!(dpiv > 0.0) || !isfinite(dpiv)) { okc = false; dpiv = 1.0; }
                        const double l = sqrt(dpiv);
                        d[jj][jj] = l;
                        rl[jj] = 1.0 / l;
#pragma unroll
                        for (int r = jj + 1; r < 8; r++) d[r][jj] = div_by(d[r][jj], l, rl[jj]);
#pragma unroll
                        for (int c = jj + 1; c < 8; c++)
#pragma unroll
                            for (int r = c; r < 8; r++) d[r][c] -= d[r][jj] * d[c][jj];
                    }
                }
                for (int row = (J + 1) * 8 + tid; row < nr; row += NT) {
                    double *pr = pan + (row >> 3) * TB + (row & 7) * TR;
                    double *hr = A + (long long)row * ldh + J * 8;
                    double v[8];
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) v[jj] = pr[jj];
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) {
#pragma unroll
                        for (int u = 0; u < jj; u++) v[jj] -= v[u] * d[jj][u];
                        v[jj] = div_by(v[jj], d[jj][jj], rl[jj]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) { pr[jj] = v[jj]; hr[jj] = v[jj]; }
                }
                if (tid == NT - 1) {
#pragma unroll
                    for (int r = 0; r < 8; r++)
#pragma unroll
                        for (int c = 0; c <= r; c++) {
                            const int rw = J * 8 + r, cl = J * 8 + c;
                            if (rw < nr && cl < nn) A[(long long)rw * ldh + cl] = d[r][c];
                        }
                }
            }
            __syncthreads();
            IPT(6);
            const int na = nb - J - 1, nt = na * (na + 1) / 2;
            for (int tix = tid; tix < nt; tix += NT) {
                int i = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5);
                while ((i + 1) * (i + 2) / 2 <= tix) i++;
                while (i * (i + 1) / 2 > tix) i--;
                const int I = J + 1 + i, C = J + 1 + tix - i * (i + 1) / 2;
                double *at = A + (long long)(I * 8) * ldh + C * 8;
                double t[8][8];
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const int row = I * 8 + r, col = C * 8 + c;
                        t[r][c] = at[(long long)r * ldh + c];
                    }
                const double *pi = pan + I * TB, *pc = pan + C * TB;
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    double li[8], lc[8];
#pragma unroll
                    for (int r = 0; r < 8; r++) { li[r] = pi[r * TR + jj]; lc[r] = pc[r * TR + jj]; }
#pragma unroll
                    for (int r = 0; r < 8; r++)
#pragma unroll
                        for (int c = 0; c < 8; c++) t[r][c] -= li[r] * lc[c];
                }
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const int row = I * 8 + r, col = C * 8 + c;
                        at[(long long)r * ldh + c] = t[r][c];
                    }
            }
            __syncthreads();
            IPT(7);
        }
        return okc;
    };
    // PQNOptimizer::optimize with useHessian (src/pqn/pqn_optimizer.cpp:29-126)
    int newton_steps = 0;
    bool hessian_failed = false;     // a Newton system that was not positive definite: that barrier step was given up
    auto optimize = [&](double tol) {
        double f = value(x);
        gradient(x, g);
        for (;;) {
            IPT(0);
            hessian();
            IPT(1);
            if (hx_lds || hx_tiled2) {
                for (int it = tid; it < nx; it += NT) xn[it] = -g[it];
                __syncthreads();
                const bool hok = hx_tiled    ? chol_solve_tiled(std::integral_constant<int, 1>{}, nx, xn, dv)
                                 : hx_tiled2 ? chol_solve_tiled(std::integral_constant<int, 2>{}, nx, xn, dv)
                                             : chol_solve_packed(nx, xn, dv);      // d = -(L L^T)^-1 g
                IPT(2);
                if (!hok) { hessian_failed = true; return; }
            } else if (hx_big) {
                // (C) Newton systems beyond the LDS vector (k = 12 SE3 poses under Dense: 2 376 variables): the blocked
                // factorisation of the cluster path on the matrix cores. The right-hand side rides along as row nx as in (B) —
                // with a pivot of its own that cannot fail, so that the factor of the bordered matrix carries L^-1 rhs in that
                // row — and the substitution's running vector lives in the dynamic LDS the factorisation has left.
                for (int it = tid; it < nx; it += NT) Hx[(long long)nx * ldh + it] = -g[it];
                if (tid == 0) { Hx[(long long)nx * ldh + nx] = 1e300; flag_s = 0; }
                __syncthreads();
                chol_lower_blocked<NT>(T, Hx, nx + 1, ldh, Hx + (long long)(nx + 8) * ldh, lds_pool);
                const bool hok = flag_s == 0;
                __syncthreads();
                if (tid == 0) flag_s = 0;
                __syncthreads();
                IPT(2);
                if (!hok) { hessian_failed = true; return; }
                solve_rows(Hx, nx, dv, std::true_type{});
                IPT(3);
            } else {
                for (int it = tid; it < nx; it += NT) Hx[(long long)nx * ldh + it] = -g[it];     // the right-hand side rides along as row nx
                __syncthreads();
                const bool hok = hx_streamed ? chol_streamed(Hx, nx) : chol_rows(Hx, nx);
                IPT(2);
                if (!hok) { hessian_failed = true; return; }
                if (hx_streamed && nx <= 511) solve_reg(std::integral_constant<int, 8>{}, std::false_type{}, nx, dv);
                else solve_rows(Hx, nx, dv, std::false_type{});     // d = -(L L^T)^-1 g
                IPT(3);
            }
            double gd = 0, da = 0;
            for (int it = tid; it < nx; it += NT) { gd += g[it] * dv[it]; da += fabs(dv[it]); }
            gd = T.sum(gd); da = T.sum(da);
            if (fabs(gd) < tol) return;
            const double f_old = f;
            double step = 1, f_new = f;
            for (;;) {
                for (int it = tid; it < nx; it += NT) xn[it] = x[it] + step * dv[it];
                __syncthreads();
                IPT(0);
                f_new = value(xn);
                IPT(4);
                if (step < 1e-12) return;                       // line search failed: x stays
                if (!isfinite(f_new) || f_new > f) { step /= 2; continue; }
                gradient(xn, gn);
                IPT(5);
                break;
            }
            double oc = 0;
            for (int it = tid; it < nx; it += NT) oc += fabs(g[it]);   // the gradient BEFORE the step, as in the reference
            oc = T.sum(oc);
            for (int it = tid; it < nx; it += NT) { x[it] = xn[it]; g[it] = gn[it]; }
            __syncthreads();
            f = f_new;
            newton_steps++;
            if (oc < tol) return;
            if (step * da < tol) return;
            if (fabs(f - f_old) < tol) return;
        }
    };
    // ---- the interior point (src/optimizer.cpp:38-79)
#ifdef SPG_IP_PROF
    for (int u = 0; u < 16; u++) ipt[u] = 0;
    ipt_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int it = tid; it < nx; it += NT) { const int o = it % DD; x[it] = (o / D == o % D) ? 1.0 : 0.0; }   // educatedGuess
    __syncthreads();
    {
        const double endRho = 5e-8, stepRho = sqrt(10.0);
        double tol = 1e-4;
        for (rho = 1; rho >= endRho; rho /= stepRho) {
            if (rho / stepRho < endRho) tol = 1e-12;
            optimize(tol);
        }
    }
#ifdef SPG_IP_PROF
    if (tid == 0) { long long tot = 0; for (int u = 0; u < 16; u++) tot += ipt[u]; printf("ipb %d %d %d %d %c %lld %lld %lld | %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld\n", k, E, nx, newton_steps, hx_tiled ? 'T' : hx_tiled2 ? 'U' : hx_streamed ? 'S' : hx_lds ? 'A' : 'B', tot, ipt[2] + ipt[6] + ipt[7], ipt[3], ipt[0], ipt[1], ipt[2], ipt[3], ipt[4], ipt[5], ipt[6], ipt[7], ipt[8], ipt[9], ipt[10], ipt[11], ipt[12], ipt[13], ipt[14]); }
    if (tid == 0 && blockIdx.x == 0) printf("ip prof k=%d E=%d nx=%d steps=%d mode %c: other %lld hessian %lld chol(trailing+rest) %lld solve %lld value %lld gradient %lld | panel load %lld panel factor %lld (cycles)\n", k, E, nx, newton_steps, hx_lds ? 'A' : 'B', ipt[0], ipt[1], ipt[2], ipt[3], ipt[4], ipt[5], ipt[6], ipt[7]);
#endif
    bool okf = false;
    const double fin = base_value(x, okf);
    if (!okf || !isfinite(fin)) { status = SPG_ST_KLD_NOT_PD; finish(); return; }    // the reference exit(0)s here
    kld = fin;
    info |= min(newton_steps, 32767) << 8;
    if (hessian_failed) info |= SPG_INFO_IP_HESSIAN_NOT_PD;
    // information of the new edges: upper triangle of the symmetric view of x
    for (int it = tid; it < E * (D * (D + 1) / 2); it += NT) {
        const int e = it / (D * (D + 1) / 2);
        int o = it - e * (D * (D + 1) / 2), i = 0;
        while (o >= D - i) { o -= D - i; i++; }
        const int j = i + o;
        arena[bd.new_off + (int64_t)e * REC + PS + (it - e * (D * (D + 1) / 2))] = Xat(x, e, i, j);
    }
    n_new = E;
    finish();
}


// Test harness of the workgroup routines above (tests/test_device_la.py through spg_debug_la): one workgroup, matrices in
// device memory, the launch's dynamic LDS as the kernel itself has it. op 0: C = / += / -= op(A) op(B) (flags: 1 ta, 2 tb,
// 4 lower_only; mode); 1: A <- chol(A) blocked; 2: A <- chol(A) in LDS panels; 3: B <- A^-1 for lower triangular A;
// 4: eigen-decomposition of A (eigenvalues on its diagonal, vectors in B). ok[0] = 0 on failure.
__global__ __launch_bounds__(NT) void la_test_kernel(int op, int M, int N, int K, int flags, int mode, double *A, int lda, double *B, int ldb,
                                                      double *Cm, int ldc, double *tmp, int *ok) {
    extern __shared__ double lds_pool[];
    __shared__ double red[NT];
    __shared__ int flag_s;
    const int tid = threadIdx.x;
    if (tid == 0) flag_s = 0;
    __syncthreads();
    Team<NT> T{tid, red, &flag_s};
    bool good = true;
    if (op == 0) team_gemm<NT>(T, Cm, ldc, A, lda, (flags & 1) != 0, B, ldb, (flags & 2) != 0, M, N, K, mode, (flags & 4) != 0, lds_pool);
    else if (op == 1) chol_lower_blocked<NT>(T, A, M, lda, tmp, lds_pool);
    else if (op == 2) chol_lower_panel<NT>(T, A, M, lda, lds_pool);
    else if (op == 3) tri_inverse_lower_blocked<NT>(T, A, lda, B, ldb, M, tmp, lds_pool);
    else if (op == 4) good = tridiag_eigh<NT>(T, A, B, M, lda, tmp, lds_pool);
    __syncthreads();
    if (tid == 0) ok[0] = (good && flag_s == 0) ? 1 : 0;
}
}  // namespace

namespace spg {

int nfr_ip_pattern_size(int topology, double chord_ratio, int k) { return ip_pattern_size(topology, chord_ratio, k); }

int64_t nfr_ip_workspace(int D, int k, int m, int E, int closed, int64_t *hot) {
    const IpLayout L = ip_layout(D, k, m, E > 0 ? E : 1, closed != 0);
    if (hot) *hot = L.hot_total;
    return L.total;
}

int hip_nfr_ip_launch(void *stream, int D, IpArgs a, int count, int n_closed, int64_t hot_max) {
    if (count <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // dynamic LDS next to the 18 KB of static LDS: as much as the device gives one workgroup (packed Hessians up to
    // ~185 variables live there; otherwise the factorisation panel and, if they fit, the hot small matrices)
    (void)hot_max;
    const size_t lds = 140 * 1024;
    a.lds_doubles = (int)(lds / 8);
    static const bool no_packed = [] { const char *e = getenv("SPG_IP_NO_LDS_HESSIAN"); return e && e[0] == '1'; }();   // diagnostic: mode (B) for every blanket
    if (no_packed) a.lds_doubles = 9000;
    static const bool untiled = [] { const char *e = getenv("SPG_IP_UNTILED"); return e && e[0] == '1'; }();           // diagnostic: the column-at-a-time LDS factorisation
    a.ip_untiled = untiled ? 1 : 0;
    static const bool eig_jacobi = [] { const char *e = getenv("SPG_EIG_JACOBI"); return e && e[0] == '1'; }();        // diagnostic: Jacobi sweeps at every size
    a.eig_jacobi = eig_jacobi ? 1 : 0;
    // n_closed of the count blankets have a closed form: each kind has its kernel (both walk the whole list and skip the
    // other's blankets: the workspace slices stay indexed by the position in the list)
    for (int closed = 0; closed < 2; closed++) {
        if ((closed ? n_closed : count - n_closed) <= 0) continue;
        const void *fn = D == 6 ? (closed ? reinterpret_cast<const void *>(nfr_ip_kernel<6, true>) : reinterpret_cast<const void *>(nfr_ip_kernel<6, false>))
                                : (closed ? reinterpret_cast<const void *>(nfr_ip_kernel<3, true>) : reinterpret_cast<const void *>(nfr_ip_kernel<3, false>));
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return SPG_EHIP;
        }
        if (D == 6) { if (closed) hipLaunchKernelGGL((nfr_ip_kernel<6, true>), dim3(count), dim3(NT), lds, s, a); else hipLaunchKernelGGL((nfr_ip_kernel<6, false>), dim3(count), dim3(NT), lds, s, a); }
        else { if (closed) hipLaunchKernelGGL((nfr_ip_kernel<3, true>), dim3(count), dim3(NT), lds, s, a); else hipLaunchKernelGGL((nfr_ip_kernel<3, false>), dim3(count), dim3(NT), lds, s, a); }
        if (hipGetLastError() != hipSuccess) return SPG_EHIP;
    }
    return 0;
}


// host side of la_test_kernel: A (ra x lda), B (rb x ldb), C (rc x ldc) row-major host arrays, copied to the device, the kernel
// run on the default stream of device 0, copied back. Returns 0, or SPG_EHIP / SPG_ENOMEM; *ok as the kernel left it.
int hip_la_test(int op, int M, int N, int K, int flags, int mode, double *A, int ra, int lda, double *B, int rb, int ldb, double *Cm, int rc, int ldc, int *ok) {
    double *dA = nullptr, *dB = nullptr, *dC = nullptr, *dT = nullptr;
    int *dok = nullptr;
    const size_t sa = (size_t)std::max(ra, 1) * lda * 8, sb = (size_t)std::max(rb, 1) * ldb * 8, sc2 = (size_t)std::max(rc, 1) * ldc * 8;
    const size_t st = ((size_t)std::max(M, 64) + 64) * 65 * 8 + (size_t)3 * std::max(M, 1) * 8;
    int rcode = 0;
    const size_t lds = 140 * 1024;
    if (hipMalloc(&dA, sa) != hipSuccess || hipMalloc(&dB, sb) != hipSuccess || hipMalloc(&dC, sc2) != hipSuccess || hipMalloc(&dT, st) != hipSuccess ||
        hipMalloc(&dok, sizeof(int)) != hipSuccess) { rcode = SPG_ENOMEM; goto out; }
    if (hipMemcpy(dA, A, sa, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dB, B, sb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dC, Cm, sc2, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dT, 0, st) != hipSuccess) { rcode = SPG_EHIP; goto out; }
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(la_test_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { rcode = SPG_EHIP; goto out; }
    hipLaunchKernelGGL(la_test_kernel, dim3(1), dim3(NT), lds, 0, op, M, N, K, flags, mode, dA, lda, dB, ldb, dC, ldc, dT, dok);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { rcode = SPG_EHIP; goto out; }
    if (hipMemcpy(A, dA, sa, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(B, dB, sb, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(Cm, dC, sc2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(ok, dok, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rcode = SPG_EHIP;
out:
    if (dA) (void)hipFree(dA);
    if (dB) (void)hipFree(dB);
    if (dC) (void)hipFree(dC);
    if (dT) (void)hipFree(dT);
    if (dok) (void)hipFree(dok);
    return rcode;
}

}  // namespace spg
