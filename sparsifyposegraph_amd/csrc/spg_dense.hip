// csrc/spg_dense.hip — dense fp64 kernels for the global Kullback-Leibler divergence.
//
// Reference: GraphWrapperG2O::kullbackLeibler (src/graph_wrapper_g2o.cpp:531-548), computeIndices
// (:472-499), estimateDifference (:550-575), sparseInformation (:382-396) and
// kullbackLeiblerDivergence(..., InformationInformation) (src/utils.cpp:70-97).
//
// The reference marginalises the baseline with a sparse Cholesky on the host and then runs dense
// n_g x n_g LDLT / solve. Here everything is dense and stays in HBM (the shipped datasets need at most
// 15k x 15k fp64 = 1.8 GB of the 288 GB): the baseline information is assembled with the variables
// ordered [marginalised | kept], ONE blocked Cholesky of that matrix leaves chol(selectedInfo) in
// its trailing block (the Schur complement is what a right-looking factorisation has produced when
// it reaches the kept block), and
//     trace(maty^-1 infox) = || L_y^-1 L_x ||_F^2,   logdet = 2 sum log diag(L),   mahalanobis = || L_x^T diff ||^2.
// The O(n^3) work (trailing updates, panel solves, the triangular solve for L_y^-1 L_x) is one
// 64x64x64 tile kernel on the fp64 matrix cores (v_mfma_f64_16x16x4_f64); diagonal blocks are
// factorised and inverted by one workgroup in LDS, so panel solves are products with L_jj^-T.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <limits>
#include <cmath>
#include <chrono>
#include <mutex>
#include <vector>

#include "spg_dev_geom.hpp"
#include "spg_dev_la.hpp"
#include "spg_internal.h"

using namespace spgdev;

namespace {

constexpr int TB = 64;    // tile edge
using d4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ double readlane64(double v, int lane) {
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return r.d;
}

#define HIPCHK(x)                                                                                              \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess) {                                                                                \
            snprintf(err, errlen, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__);     \
            rc = SPG_EHIP;                                                                                     \
            goto done;                                                                                         \
        }                                                                                                      \
    } while (0)

// ------------------------------------------------------------------------------------------ tiles
// C(p,q) (op)= A(p) * B(q)^T with K = 32 * kchunks: one workgroup (4 wavefronts) per 128 x 128 output tile, one
// wavefront per 64 x 64 quadrant = 4 x 4 v_mfma_f64_16x16x4_f64 accumulators. Tiles are addressed in
// units of 64 rows/cols — base + p64*stride_p + q64*stride_q (elements), each operand with its own
// leading dimension — and a 128-tile at the edge of an odd tile count computes its valid 64-wide half.
// A (128 x K) and B (128 x K) pass through LDS in K-chunks of 32 (row stride 34 doubles: the
// fragment reads of a half wave — 16 rows x 2 k — fall on 32 distinct 8-byte bank pairs). 128 accumulator
// registers + 70 KB of LDS leave room for two workgroups per CU, which overlap each other's loads.
struct TileOp {
    double *C;
    const double *A, *B;
    long long c_p, c_q, a_p, b_q;   // element strides per 64-tile index (A depends on p only, B on q only)
    int ldc, lda, ldb;
    int P64, Q64;  // extent in 64-tiles (rows of A / rows of B)
    int tri;       // 1: blockIdx.x enumerates the 128-tile pairs q <= p of a triangle
    int subtract;  // 1: C -= A B^T; 0: C = A B^T (C may alias A: A is consumed before C is written)
    int kchunks;   // K = 32 * kchunks (2 for a 64-wide panel; 16 for the 512-wide outer block)
    // latency form only: the workgroup of tile (0, 0) goes on to factorise that tile (the next panel's diagonal block) and
    // to write its inverse here — the factorisation chain loses one launch per panel
    double *diag_linv;
    int *diag_bad;
};

constexpr int KC = 32, LDK = 34;

// p, q: the 128-tile this workgroup computes; As / Bs: 128 * LDK doubles of LDS each. Every thread of the workgroup
// takes part (barriers inside); the caller separates two tiles that reuse the LDS by a barrier of its own.
__device__ __forceinline__ void tile_abt_body(const TileOp &op, int p, int q, double *As, double *Bs) {
    const int rows = min(2, op.P64 - 2 * p) * TB, cols = min(2, op.Q64 - 2 * q) * TB;   // valid extent: 64 or 128
    const double *A = op.A + 2 * p * op.a_p;
    const double *B = op.B + 2 * q * op.b_q;
    double *C = op.C + 2 * p * op.c_p + 2 * q * op.c_q;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // staging map: thread -> (row = tid / 2, 16 consecutive k at (tid & 1) * 16) of a 128 x 32 chunk
    const int srow = tid >> 1, sk = (tid & 1) * 16;
    const bool a_ok = srow < rows, b_ok = srow < cols;
    const double *ga = A + (long long)srow * op.lda + sk, *gb = B + (long long)srow * op.ldb + sk;
    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double2 ra[4], rb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ra[i] = a_ok ? *reinterpret_cast<const double2 *>(ga + chunk * KC + 8 * h + 2 * i) : double2{0, 0};
                rb[i] = b_ok ? *reinterpret_cast<const double2 *>(gb + chunk * KC + 8 * h + 2 * i) : double2{0, 0};
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                *reinterpret_cast<double2 *>(&As[srow * LDK + sk + 8 * h + 2 * i]) = ra[i];
                *reinterpret_cast<double2 *>(&Bs[srow * LDK + sk + 8 * h + 2 * i]) = rb[i];
            }
        }
    };
    const int r0 = (w >> 1) * 64, c0 = (w & 1) * 64;   // this wave's quadrant
    const bool live = r0 < rows && c0 < cols;
    const int li = lane & 15, lk = lane >> 4;
    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++) acc[x][y] = d4{0, 0, 0, 0};
    auto multiply = [&]() {
        if (!live) return;
#pragma unroll 2
        for (int k0 = 0; k0 < KC; k0 += 4) {
            double a[4], b[4];
            // A fragment: A[row = li][k = lk]; B fragment: B[k = lk][col = li] = Bt[col][k]
#pragma unroll
            for (int x = 0; x < 4; x++) {
                a[x] = As[(r0 + 16 * x + li) * LDK + k0 + lk];
                b[x] = Bs[(c0 + 16 * x + li) * LDK + k0 + lk];
            }
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 4; y++) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
        }
    };
    for (int ch = 0; ch < op.kchunks; ch++) {
        if (ch) __syncthreads();
        load_chunk(ch);
        __syncthreads();
        multiply();
    }
    if (!live) return;
    // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int row = r0 + 16 * x + lk + 4 * r, col = c0 + 16 * y + li;
                double *dst = C + (long long)row * op.ldc + col;
                *dst = op.subtract ? (*dst - acc[x][y][r]) : acc[x][y][r];
            }
}

// the 128-tile (p, q) behind a linear index: row-major over the rectangle, or t = p(p+1)/2 + q over the triangle q <= p
__device__ __forceinline__ void tile_of(const TileOp &op, int t, int &p, int &q) {
    if (op.tri) {
        p = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((p + 1) * (p + 2) / 2 <= t) p++;
        while (p * (p + 1) / 2 > t) p--;
        q = t - p * (p + 1) / 2;
    } else {
        const int gq = (op.Q64 + 1) / 2;
        p = t / gq;
        q = t - p * gq;
    }
}

__global__ __launch_bounds__(256, 2) void tile_abt_kernel(TileOp op) {
    __shared__ __attribute__((aligned(16))) double As[128 * LDK], Bs[128 * LDK];
    int p = blockIdx.x, q = blockIdx.y;
    if (op.tri) tile_of(op, blockIdx.x, p, q);
    tile_abt_body(op, p, q, As, Bs);
}

// Factor the 64x64 diagonal block at Ajj (lower Cholesky, in place, strict upper zeroed) and write
// its inverse (dense 64x64, row-major, strict upper zero) to Linv; factor = 0: the block already holds
// a Cholesky factor, only invert it. *bad is set if a pivot is not positive. One wavefront per block — this is the serial
// link of every blocked factorisation here, so its latency counts, not its throughput. The block lives in LDS (Ls, the
// inverse grows in Li; 64 x 65 doubles each) and is processed in 16-wide block columns: the 16 x 16 diagonal block is
// factorised and inverted in registers (lane i owns row i, pivot rows broadcast with v_readlane: 2 x 120 dependent
// steps instead of the 2 x 2016 of a register-resident 64 x 64), the panel below it (product with the inverse), the
// trailing update and the block forward substitution for L^-1 are 16 x 16 x 16 products on v_mfma_f64_16x16x4_f64 with
// operands read from LDS. (Round 2: whole block in 128 registers per lane, 44 us; the LDS-cooperative version before
// that, 119 us.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// acc += A B^T (bt) or A B: A, B 16 x 16 blocks in LDS with row stride 65
__device__ __forceinline__ d4 mm16(d4 acc, const double *A, const double *B, bool bt, int li, int lk) {
#pragma unroll
    for (int s4 = 0; s4 < 16; s4 += 4) {
        const double a = A[li * 65 + s4 + lk];
        const double b = bt ? B[li * 65 + s4 + lk] : B[(s4 + lk) * 65 + li];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ void diag_potrf_body(double *Ls /* LDS, 64 * 65 */, double *Li /* LDS, 64 * 65 */, double *Ajj, int ld, double *Linv, int *bad, int factor) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    // coalesced load of the lower triangle into LDS (row stride 65): all 64 rows in flight at once — one memory latency,
    // not one per group of rows, on the critical path of the factorisation
    {
        double t[TB];
#pragma unroll
        for (int i = 0; i < TB; i++) t[i] = (lane <= i) ? Ajj[(long long)i * ld + lane] : 0.0;
#pragma unroll
        for (int i = 0; i < TB; i++) { Ls[i * 65 + lane] = t[i]; Li[i * 65 + lane] = 0.0; }
    }
    wave_sync();
    auto blk = [&](double *M, int i, int j) { return M + (16 * i) * 65 + 16 * j; };
    bool ok = true;
#pragma unroll 1
    for (int jb = 0; jb < 4; jb++) {
        double *Dg = blk(Ls, jb, jb);
        double a[16];   // row li of the diagonal block (each group of 16 lanes holds a copy; lanes 0-15 are the ones read)
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = Dg[li * 65 + c];
        if (factor) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                double d = readlane64(a[j], j);
                if (!(d > 0.0) || !isfinite(d)) { ok = false; d = 1.0; }
                const double rs = fast_rsqrt(d);
                a[j] = (li == j) ? d * rs : a[j] * rs;      // L_jj = sqrt(d); column j scaled
#pragma unroll
                for (int c = j + 1; c < 16; c++) a[c] -= a[j] * readlane64(a[j], c);     // (entries above the diagonal are never read)
            }
        }
        // x[i] = (L16^-1)[i][li]: one forward substitution per lane, L16[i][k] broadcast from lane i
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double sacc = (li == i) ? 1.0 : 0.0;
#pragma unroll
            for (int k2 = 0; k2 < i; k2++) sacc -= readlane64(a[k2], i) * x[k2];
            x[i] = (li <= i) ? sacc * fast_rcp(readlane64(a[i], i)) : 0.0;
        }
        if (lk == 0) {
            if (factor) {
#pragma unroll
                for (int c = 0; c < 16; c++) Dg[li * 65 + c] = (c <= li) ? a[c] : 0.0;
            }
            double *Dv = blk(Li, jb, jb);
#pragma unroll
            for (int i = 0; i < 16; i++) Dv[i * 65 + li] = x[i];
        }
        wave_sync();
        if (factor && jb < 3) {
            // panel: L[ib][jb] = A[ib][jb] L16^-T for the block rows below
            const double *Dv = blk(Li, jb, jb);
            d4 pc[3];
#pragma unroll
            for (int t = 0; t < 3; t++) { pc[t] = d4{0, 0, 0, 0}; if (jb + 1 + t < 4) pc[t] = mm16(pc[t], blk(Ls, jb + 1 + t, jb), Dv, true, li, lk); }
            wave_sync();
#pragma unroll
            for (int t = 0; t < 3; t++)
                if (jb + 1 + t < 4) {
                    double *P = blk(Ls, jb + 1 + t, jb);
#pragma unroll
                    for (int r = 0; r < 4; r++) P[(lk + 4 * r) * 65 + li] = pc[t][r];
                }
            wave_sync();
            // trailing blocks (ib >= kb > jb): A[ib][kb] -= L[ib][jb] L[kb][jb]^T
#pragma unroll 1
            for (int kb = jb + 1; kb < 4; kb++)
#pragma unroll 1
                for (int ib = kb; ib < 4; ib++) {
                    d4 acc = mm16(d4{0, 0, 0, 0}, blk(Ls, ib, jb), blk(Ls, kb, jb), true, li, lk);
                    double *Cb = blk(Ls, ib, kb);
#pragma unroll
                    for (int r = 0; r < 4; r++) Cb[(lk + 4 * r) * 65 + li] -= acc[r];
                }
            wave_sync();
        }
    }
    if (factor) {
        if (!ok && lane == 0) *bad = 1;
        for (int r = 0; r < TB; r++) Ajj[(long long)r * ld + lane] = Ls[r * 65 + lane];
    }
    // L^-1 below its diagonal blocks, block column by block column: Linv[ib][jb] = -L16_ib^-1 sum_{jb <= kb < ib} L[ib][kb] Linv[kb][jb]
    // (the product passes through the unused block (0, 3) of Ls to change from the result layout to the operand layout)
    wave_sync();
    double *Tmp = blk(Ls, 0, 3);
#pragma unroll 1
    for (int jb = 0; jb < 3; jb++)
#pragma unroll 1
        for (int ib = jb + 1; ib < 4; ib++) {
            d4 acc = d4{0, 0, 0, 0};
#pragma unroll 1
            for (int kb = jb; kb < ib; kb++) acc = mm16(acc, blk(Ls, ib, kb), blk(Li, kb, jb), false, li, lk);
#pragma unroll
            for (int r = 0; r < 4; r++) Tmp[(lk + 4 * r) * 65 + li] = acc[r];
            wave_sync();
            d4 res = mm16(d4{0, 0, 0, 0}, blk(Li, ib, ib), Tmp, false, li, lk);
            double *Ob = blk(Li, ib, jb);
#pragma unroll
            for (int r = 0; r < 4; r++) Ob[(lk + 4 * r) * 65 + li] = -res[r];
            wave_sync();
        }
    for (int r = 0; r < TB; r++) Linv[r * TB + lane] = Li[r * 65 + lane];
}

__global__ __launch_bounds__(64) void diag_potrf_kernel(double *Ajj, int ld, double *Linv, int *bad, int factor) {
    __shared__ double Ls[TB * 65], Li[TB * 65];
    diag_potrf_body(Ls, Li, Ajj + (long long)blockIdx.x * TB * ((long long)ld + 1), ld, Linv + (long long)blockIdx.x * TB * TB, bad, factor);
}

// The same step for the steps of a factorisation chain, which have a handful of tiles and are waited for: what counts
// is the time to the LAST tile, not the throughput. One workgroup per 64 x 64 tile (four times as many workgroups), its
// four wavefronts split K: each reads its quarter of the two operands straight from memory in the fragment layout (no
// LDS staging, no barrier in the loop — all loads of a wavefront are in flight together), the partial sums meet in LDS
// and wavefront w finishes rows 16 w .. 16 w + 15 of the tile. (128 x 128 x 512 on one CU is 27 us of matrix-core time
// alone; a 64 x 64 tile with K split four ways holds each wavefront for a sixteenth of that.)
__global__ __launch_bounds__(256) void tile_abt_small_kernel(TileOp op) {
    __shared__ double part[4][3][16][64];       // [owner block row x][source slot][4 block columns x 4 registers][lane]
    int p = blockIdx.x, q = blockIdx.y;
    if (op.tri) {
        const int t = blockIdx.x;   // t = p(p+1)/2 + q over 64-tiles
        p = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((p + 1) * (p + 2) / 2 <= t) p++;
        while (p * (p + 1) / 2 > t) p--;
        q = t - p * (p + 1) / 2;
    }
    const double *A = op.A + p * op.a_p;
    const double *B = op.B + q * op.b_q;
    double *C = op.C + p * op.c_p + q * op.c_q;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int kw = 8 * op.kchunks, kbeg = w * kw;      // K = 32 * kchunks, a quarter per wavefront
    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++) acc[x][y] = d4{0, 0, 0, 0};
    const double *ga = A + (long long)li * op.lda + kbeg + lk, *gb = B + (long long)li * op.ldb + kbeg + lk;
#pragma unroll 2
    for (int k0 = 0; k0 < kw; k0 += 4) {
        double a[4], b[4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            a[x] = ga[(long long)(16 * x) * op.lda + k0];
            b[x] = gb[(long long)(16 * x) * op.ldb + k0];
        }
#pragma unroll
        for (int x = 0; x < 4; x++)
#pragma unroll
            for (int y = 0; y < 4; y++) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
    }
    // block row x of the tile belongs to wavefront x: the other three park their partial sums for it
#pragma unroll
    for (int x = 0; x < 4; x++) {
        if (x == w) continue;
        const int slot = (w - x - 1) & 3;       // 0..2
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int r = 0; r < 4; r++) part[x][slot][4 * y + r][lane] = acc[x][y][r];
    }
    __syncthreads();
    // (the operands were read before the barrier: C may alias A as in the 128-tile kernel)
#pragma unroll
    for (int x = 0; x < 4; x++) {
        if (x != w) continue;
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double v = acc[x][y][r] + part[x][0][4 * y + r][lane] + part[x][1][4 * y + r][lane] + part[x][2][4 * y + r][lane];
                const int row = 16 * x + lk + 4 * r, col = 16 * y + li;      // C/D layout of the f64 MFMA
                double *dst = C + (long long)row * op.ldc + col;
                *dst = op.subtract ? (*dst - v) : v;
            }
    }
    if (op.diag_linv && p == 0 && q == 0) {
        // the next panel's diagonal block is this tile: factorise it here (the partial-sum area is free by now)
        __syncthreads();
        double *Ls = &part[0][0][0][0];
        if (w == 0) diag_potrf_body(Ls, Ls + TB * 65, C, op.ldc, op.diag_linv, op.diag_bad, 1);
    }
}

// pad rows [n, N) of an N x N matrix get a unit diagonal
__global__ void pad_identity_kernel(double *M, int ld, int n0, int n1) {
    int i = n0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) M[(long long)i * ld + i] = 1.0;
}

// Y = L^T (upper triangular copy; the strict lower triangle of Y becomes 0)
__global__ void transpose_lower_kernel(const double *L, int ldl, double *Y, int ldy, int n) {
    __shared__ double t[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // Y tile at rows by.., cols bx..
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 256 threads: 8 rows per pass
    for (int i = ty; i < 32; i += 8) {
        int r = bx + i, c = by + tx;                  // L[r][c] -> Y[c][r]
        t[i][tx] = (r < n && c < n && c <= r) ? L[(long long)r * ldl + c] : 0.0;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int r = by + i, c = bx + tx;
        if (r < n && c < n) Y[(long long)r * ldy + c] = t[tx][i];
    }
}

// out[0] += sum of squares of M (n x n); out[1] += 2 * sum log diag(La); out[2] += 2 * sum log diag(Lb)
__global__ __launch_bounds__(256) void sumsq_kernel(const double *M, int ld, int n, double *partial) {
    __shared__ double red[256];
    double s = 0;
    long long tot = (long long)n * n;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < tot; it += (long long)gridDim.x * 256) {
        long long r = it / n, c = it - r * n;
        double v = M[r * ld + c];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// rowsq[r] = (sum_{c >= r} Y[r][c] diff[c])^2: one wavefront per row of the upper-triangular Y = L_x^T
__global__ __launch_bounds__(256) void upper_matvec_sq_kernel(const double *Y, int ldy, int n, const double *diff, double *rowsq) {
    int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    double v = 0;
    for (int c = (r & ~63) + lane; c < n; c += 64)
        if (c >= r) v += Y[(long long)r * ldy + c] * diff[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) rowsq[r] = v * v;
}

// one workgroup: out[0] = sum partial[0..np), out[1] = 2 sum_{i<n} log La[i][i], out[2] = 2 sum log Lb[i][i],
// out[3] = || Y^T... the Mahalanobis term: sum_r (sum_c Y[r][c] diff[c])^2 with Y = L_x^T (upper)
__global__ __launch_bounds__(256) void finish_kernel(const double *partial, int np, const double *La, int lda, const double *Lb, int ldb,
                                                     int n, const double *Y, int ldy, const double *diff, double *out) {
    __shared__ double red[256];
    auto reduce = [&](double v) {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        double r = red[0];
        __syncthreads();
        return r;
    };
    double s = 0;
    for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
    double t0 = reduce(s);
    s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += log(La[(long long)i * lda + i]);
    double t1 = 2.0 * reduce(s);
    s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += log(Lb[(long long)i * ldb + i]);
    double t2 = 2.0 * reduce(s);
    s = 0;
    if (diff)   // here: the per-row squares of upper_matvec_sq_kernel
        for (int r = threadIdx.x; r < n; r += 256) s += diff[r];
    double t3 = reduce(s);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; out[2] = t2; out[3] = t3; }
}

// ------------------------------------------------------------------------------------------ assembly
struct GraphDev {
    const double *arena;
    const int32_t *pos;       // per vertex: scalar offset of its block in the matrix, -1 = not a variable
    const int64_t *vpo;       // per vertex: pose offset in the arena
    const int32_t *rowptr;    // CSR: incident live edges per vertex (ascending edge index)
    const int32_t *inc;
    const spg_edge_ref *er;
    const int32_t *ev;        // edge -> global vertex indices
    const int64_t *aw_off;    // per edge: offset of its weighted Jacobian in aw (GLC edges), -1 otherwise
    double *aw;
    int nv, ne;
};

template <int D>
__device__ __forceinline__ void load_pose(const double *arena, int64_t off, double *X) {
    if (D == 6) iso_from_tq(arena + off, X);
    else { X[0] = arena[off]; X[1] = arena[off + 1]; X[2] = arena[off + 2]; }
}

// rows of the weighted Jacobian A_e of an n-ary edge: GLC (W is r x dq after the dq measurement doubles) or MULTI (r = d nm)
__device__ __forceinline__ int nary_rows(const spg_edge_ref &er, const double *arena, int D) {
    if (er.kind == SPG_EDGE_MULTI) return D * (int)arena[er.off];
    const int dq = D * er.nv;
    return (er.len - dq) / dq;
}

// A_e = W_e * J_reparam (r x dq) of every GLC edge (src/glc_edge.cpp:40-49,
// src/glc_reparam_binary.hpp:78-127): one workgroup per edge, result in g.aw.
template <int D>
__global__ __launch_bounds__(64) void glc_weighted_jacobian_kernel(GraphDev g) {
    constexpr int DD = D * D, PSZ = (D == 6) ? kIso : 3;
    extern __shared__ double Jb[];   // q x (Ji0 | Jii), then q x D reparametrisation errors
    const int e = blockIdx.x, tid = threadIdx.x;
    const spg_edge_ref er = g.er[e];
    if (er.kind == SPG_EDGE_BINARY) return;
    const int q = er.nv, dq = D * q, rr = nary_rows(er, g.arena, D);
    const double *rec = g.arena + er.off;
    double *Aw = g.aw + g.aw_off[e];
    if (er.kind == SPG_EDGE_MULTI) {
        // MultiEdgeCorrelated (src/multi_edge_correlated.hpp:65-140): stacked pose-pose errors, two Jacobian blocks per
        // measurement, W from the record (include/spg.h): A = W J, weighted error W e
        const int nm = (int)rec[0];
        const double *meas = rec + 1 + 2 * nm, *Wd = meas + nm * ((D == 6) ? 7 : 3);
        for (int i = tid; i < nm; i += 64) {
            const int la = (int)rec[1 + 2 * i], lb = (int)rec[2 + 2 * i];
            double Xa[PSZ], Xb[PSZ];
            load_pose<D>(g.arena, g.vpo[g.ev[er.vbegin + la]], Xa);
            load_pose<D>(g.arena, g.vpo[g.ev[er.vbegin + lb]], Xb);
            double *er_i = Jb + q * 2 * DD + i * D;
            if (D == 6) {
                double Z[kIso];
                iso_from_tq(meas + 7 * i, Z);
                se3_edge_jac(Xa, Xb, Z, Jb + i * 2 * DD, Jb + i * 2 * DD + DD, er_i);
            } else {
                se2_edge_jac(Xa, Xb, meas + 3 * i, Jb + i * 2 * DD, Jb + i * 2 * DD + DD, er_i);
            }
        }
        __syncthreads();
        for (int row = tid; row < rr; row += 64) {
            const double *Wr = Wd + (int64_t)row * rr, *ee = Jb + q * 2 * DD;
            double s = 0;
            for (int j = 0; j < rr; j++) s += Wr[j] * ee[j];
            Aw[(int64_t)rr * dq + row] = s;
        }
        for (int it = tid; it < rr * dq; it += 64) {
            const int row = it / dq, col = it - row * dq, blk = col / D, c = col - blk * D;
            const double *Wr = Wd + (int64_t)row * rr;
            double s = 0;
            for (int i = 0; i < nm; i++) {
                const int la = (int)rec[1 + 2 * i], lb = (int)rec[2 + 2 * i];
                if (la == blk) for (int p = 0; p < D; p++) s += Wr[i * D + p] * Jb[i * 2 * DD + p * D + c];
                if (lb == blk) for (int p = 0; p < D; p++) s += Wr[i * D + p] * Jb[i * 2 * DD + DD + p * D + c];
            }
            Aw[it] = s;
        }
        return;
    }
    for (int i = tid; i < q; i += 64) {
        double X0[PSZ], Xi[PSZ];
        load_pose<D>(g.arena, g.vpo[g.ev[er.vbegin]], X0);
        load_pose<D>(g.arena, g.vpo[g.ev[er.vbegin + i]], Xi);
        double *er_i = Jb + q * 2 * DD + i * D;   // GLCEdge::computeError before the weighting (src/glc_edge.cpp:28-38)
        if (D == 6) {
            double Z[kIso], Xz[kIso] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
            iso_from_mqt(rec + 6 * i, Z);
            if (i == 0) se3_edge_jac(Xz, X0, Z, Jb, Jb + DD, er_i);
            else se3_edge_jac(X0, Xi, Z, Jb + i * 2 * DD, Jb + i * 2 * DD + DD, er_i);
        } else {
            double xz[3] = {0, 0, 0};
            if (i == 0) se2_edge_jac(xz, X0, rec, Jb, Jb + DD, er_i);
            else se2_edge_jac(X0, Xi, rec + 3 * i, Jb + i * 2 * DD, Jb + i * 2 * DD + DD, er_i);
        }
    }
    __syncthreads();
    for (int row = tid; row < rr; row += 64) {     // weighted error r = W * err
        const double *Wr = rec + dq + (int64_t)row * dq, *er = Jb + q * 2 * DD;
        double s = 0;
        for (int j = 0; j < dq; j++) s += Wr[j] * er[j];
        Aw[(int64_t)rr * dq + row] = s;
    }
    for (int it = tid; it < rr * dq; it += 64) {
        int row = it / dq, col = it - row * dq, blk = col / D, c = col - blk * D;
        const double *Wr = rec + dq + (int64_t)row * dq;
        double s = 0;
        if (blk == 0) {
            for (int p = 0; p < D; p++) s += Wr[p] * Jb[DD + p * D + c];
            for (int i = 1; i < q; i++)
                for (int p = 0; p < D; p++) s += Wr[i * D + p] * Jb[i * 2 * DD + p * D + c];
        } else {
            for (int p = 0; p < D; p++) s += Wr[blk * D + p] * Jb[blk * 2 * DD + DD + p * D + c];
        }
        Aw[it] = s;
    }
}

// Where the assembly puts block (row v, column u) of H. pv, pu: the scalar positions of the two vertices (GraphDev::pos).
struct DenseSink {
    double *M;
    int ld;
    __device__ __forceinline__ void add(int pv, int pu, int r, int c, double val) { M[(long long)(pv + r) * ld + pu + c] += val; }
    __device__ __forceinline__ void diag(int pv, int r, int c, double val) { M[(long long)(pv + r) * ld + pv + c] = val; }
    __device__ __forceinline__ void finish(int, int) {}
};

// H = sum_e J_e^T Omega_e J_e over the live edges, rows of vertex v by workgroup v (one wavefront):
// deterministic (edges in ascending index), no atomics. Only the lower block triangle (pos(u) <=
// pos(v)) is written; diagonal blocks are written in full.
template <int D, class Sink>
__global__ __launch_bounds__(64) void dense_assemble_kernel(GraphDev g, Sink sink, double *bvec) {
    constexpr int DD = D * D, PS = (D == 6) ? 7 : 3, PSZ = (D == 6) ? kIso : 3;
    __shared__ double Jv[DD], Ju[DD], Om[DD], Tv[DD], Tu[DD], Er[D];
    double bacc = 0;   // lanes < D: b_v[lane] = -sum_e (J_v^T Omega e)[lane]  (g2o's right-hand side)
    const int v = blockIdx.x, tid = threadIdx.x;
    const int pv = g.pos[v];
    if (pv < 0) return;
    const int r = tid / D, c = tid - r * D;      // lanes < DD own one entry of a d x d block
    const bool act = tid < DD;
    double diag = 0;
    for (int ii = g.rowptr[v]; ii < g.rowptr[v + 1]; ii++) {
        const int e = g.inc[ii];
        const spg_edge_ref er = g.er[e];
        if (er.kind == SPG_EDGE_BINARY) {
            const int vi = g.ev[er.vbegin], vj = g.ev[er.vbegin + 1];
            if (vi == vj) continue;
            const double *rec = g.arena + er.off;
            if (tid == 0) {
                double Xi[PSZ], Xj[PSZ];
                load_pose<D>(g.arena, g.vpo[vi], Xi);
                load_pose<D>(g.arena, g.vpo[vj], Xj);
                double *Ji = (v == vi) ? Jv : Ju, *Jj = (v == vi) ? Ju : Jv;
                if (D == 6) {
                    double Z[kIso];
                    iso_from_tq(rec, Z);
                    se3_edge_jac(Xi, Xj, Z, Ji, Jj, Er);
                } else {
                    se2_edge_jac(Xi, Xj, rec, Ji, Jj, Er);
                }
            }
            if (act) {
                int lo = r < c ? r : c, hi = r < c ? c : r;
                Om[tid] = rec[PS + lo * D - lo * (lo - 1) / 2 + (hi - lo)];
            }
            __syncthreads();
            if (act) {
                double sv = 0, su = 0;
#pragma unroll
                for (int p = 0; p < D; p++) { sv += Om[r * D + p] * Jv[p * D + c]; su += Om[r * D + p] * Ju[p * D + c]; }
                Tv[tid] = sv; Tu[tid] = su;
            }
            __syncthreads();
            const int u = (v == vi) ? vj : vi, pu = g.pos[u];
            if (act) {
                double sd = 0, so = 0;
#pragma unroll
                for (int p = 0; p < D; p++) { sd += Jv[p * D + r] * Tv[p * D + c]; so += Jv[p * D + r] * Tu[p * D + c]; }
                diag += sd;
                if (pu >= 0 && pu < pv) sink.add(pv, pu, r, c, so);
            }
            if (tid < D) {
#pragma unroll
                for (int p = 0; p < D; p++) bacc -= Tv[p * D + tid] * Er[p];
            }
            __syncthreads();
        } else {
            const int q = er.nv, dq = D * q, rr = nary_rows(er, g.arena, D);
            const double *Aw = g.aw + g.aw_off[e];
            int iv = 0;
            for (int i = 0; i < q; i++) if (g.ev[er.vbegin + i] == v) iv = i;
            if (act) {
                for (int iu = 0; iu < q; iu++) {
                    const int u = g.ev[er.vbegin + iu], pu = g.pos[u];
                    if (!(u == v || (pu >= 0 && pu < pv))) continue;
                    double s = 0;
                    for (int p = 0; p < rr; p++) s += Aw[(int64_t)p * dq + iv * D + r] * Aw[(int64_t)p * dq + iu * D + c];
                    if (u == v) diag += s;
                    else sink.add(pv, pu, r, c, s);
                }
            }
            if (tid < D) {
                const double *wr = Aw + (int64_t)rr * dq;
                for (int p = 0; p < rr; p++) bacc -= Aw[(int64_t)p * dq + iv * D + tid] * wr[p];
            }
        }
    }
    if (act) sink.diag(pv, r, c, diag);
    if (bvec && tid < D) bvec[pv + tid] = bacc;
    sink.finish(v, tid);
}

// chi2 per edge (binary: e^T Omega e; GLC: ||W e||^2 from glc_weighted_jacobian_kernel), one lane per edge
template <int D>
__global__ void edge_chi2_kernel(GraphDev g, double *chi) {
    constexpr int DD = D * D, PS = (D == 6) ? 7 : 3, PSZ = (D == 6) ? kIso : 3;
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.ne) return;
    const spg_edge_ref er = g.er[e];
    double s = 0;
    if (er.kind == SPG_EDGE_BINARY) {
        const double *rec = g.arena + er.off;
        double Xi[PSZ], Xj[PSZ], Ji[DD], Jj[DD], err[D];
        load_pose<D>(g.arena, g.vpo[g.ev[er.vbegin]], Xi);
        load_pose<D>(g.arena, g.vpo[g.ev[er.vbegin + 1]], Xj);
        if (D == 6) {
            double Z[kIso];
            iso_from_tq(rec, Z);
            se3_edge_jac(Xi, Xj, Z, Ji, Jj, err);
        } else {
            se2_edge_jac(Xi, Xj, rec, Ji, Jj, err);
        }
        int p = PS;
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int j = i; j < D; j++) { s += ((i == j) ? 1.0 : 2.0) * err[i] * rec[p] * err[j]; p++; }
    } else {
        const int dq = D * er.nv, rr = nary_rows(er, g.arena, D);
        const double *wr = g.aw + g.aw_off[e] + (int64_t)rr * dq;
        for (int p = 0; p < rr; p++) s += wr[p] * wr[p];
    }
    chi[e] = s;
}

// out[slot] = sum v[0..n) in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void sum_kernel(const double *v, int n, double *out, int slot) {
    __shared__ double red[256];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[slot] = red[0];
}

// out[0] = max_i |M[i][i]|, i < n
__global__ __launch_bounds__(256) void max_diag_kernel(const double *M, int ld, int n, double *out, int slot) {
    __shared__ double red[256];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s = fmax(s, fabs(M[(long long)i * ld + i]));
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[slot] = red[0];
}

__global__ void add_diag_kernel(double *M, int ld, int n, double lambda) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) M[(long long)i * ld + i] += lambda;
}

// out[slot] = sum_i x[i] * (lambda * x[i] + b[i])   (OptimizationAlgorithmLevenberg::computeScale)
__global__ __launch_bounds__(256) void lm_scale_kernel(const double *x, const double *b, int n, double lambda, double *out, int slot) {
    __shared__ double red[256];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i] * (lambda * x[i] + b[i]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[slot] = red[0];
}

// One block step of L y = rhs (lower, forward). Workgroup 0 stores y_j = L_jj^-1 rhs_j; workgroup g >= 1
// updates rhs_{j+g} -= L[j+g, j] y_j. Every workgroup recomputes y_j (a 64 x 64 product) itself.
__global__ __launch_bounds__(64) void trsv_forward_step(const double *L, int ld, const double *Linv_all, double *rhs, double *sol, int j) {
    __shared__ double xj[TB];
    const int lane = threadIdx.x, g = blockIdx.x;
    const double *Li = Linv_all + (long long)j * TB * TB;
    double s = 0;
    for (int c = 0; c <= lane; c++) s += Li[lane * TB + c] * rhs[j * TB + c];
    xj[lane] = s;
    __syncthreads();
    if (g == 0) { sol[j * TB + lane] = s; return; }
    const double *row = L + (long long)((j + g) * TB + lane) * ld + (long long)j * TB;
    double u = 0;
#pragma unroll 8
    for (int c = 0; c < TB; c++) u += row[c] * xj[c];
    rhs[(j + g) * TB + lane] -= u;
}

// One block step of L^T x = rhs (backward). Workgroup 0 stores x_j = L_jj^-T rhs_j; workgroup g >= 1
// updates rhs_{g-1} -= L[j, g-1]^T x_j for the blocks above.
__global__ __launch_bounds__(64) void trsv_backward_step(const double *L, int ld, const double *Linv_all, double *rhs, double *sol, int j) {
    __shared__ double xj[TB];
    const int lane = threadIdx.x, g = blockIdx.x;
    const double *Li = Linv_all + (long long)j * TB * TB;
    double s = 0;
    for (int r = lane; r < TB; r++) s += Li[r * TB + lane] * rhs[j * TB + r];
    xj[lane] = s;
    __syncthreads();
    if (g == 0) { sol[j * TB + lane] = s; return; }
    const int i = g - 1;
    const double *blk = L + (long long)j * TB * ld + (long long)i * TB;   // L[j-block rows][i-block cols]
    double u = 0;
#pragma unroll 8
    for (int r = 0; r < TB; r++) u += blk[(long long)r * ld + lane] * xj[r];
    rhs[i * TB + lane] -= u;
}

// VertexSE2 / VertexSE3 oplus with the solution vector; mode 0: apply, 1: save poses, 2: restore poses
template <int D>
__global__ void pose_update_kernel(double *arena, const int64_t *vpo, const int32_t *pos, int nv, const double *x, double *backup, int mode) {
    constexpr int PS = (D == 6) ? 7 : 3;
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv || pos[v] < 0) return;
    double *p = arena + vpo[v];
    if (mode == 1) { for (int a = 0; a < PS; a++) backup[(long long)v * PS + a] = p[a]; return; }
    if (mode == 2) { for (int a = 0; a < PS; a++) p[a] = backup[(long long)v * PS + a]; return; }
    const double *dx = x + pos[v];
    if (D == 3) {
        p[0] += dx[0]; p[1] += dx[1]; p[2] = normalize_theta(p[2] + dx[2]);
    } else {
        // X <- X * fromVectorMQT(dx), kept as translation + unit quaternion
        double X[kIso], Dl[kIso], R[9], q[4];
        iso_from_tq(p, X);
        iso_from_mqt(dx, Dl);
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int k = 0; k < 3; k++) R[i * 3 + k] = X[i * 3] * Dl[k] + X[i * 3 + 1] * Dl[3 + k] + X[i * 3 + 2] * Dl[6 + k];
            p[i] = X[9 + i] + X[i * 3] * Dl[9] + X[i * 3 + 1] * Dl[10] + X[i * 3 + 2] * Dl[11];
        }
        R_to_quat(R, q);
        p[3] = q[0]; p[4] = q[1]; p[5] = q[2]; p[6] = q[3];
    }
}

// estimateDifference (src/graph_wrapper_g2o.cpp:550-575): one lane per kept vertex
template <int D>
__global__ void pose_diff_kernel(const double *arena_b, const int64_t *vpo_b, const double *arena_o, const int64_t *vpo_o, int nk, double *diff) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nk) return;
    const double *xb = arena_b + vpo_b[i], *xo = arena_o + vpo_o[i];
    if (D == 3) {
        diff[3 * i] = xb[0] - xo[0];
        diff[3 * i + 1] = xb[1] - xo[1];
        diff[3 * i + 2] = normalize_theta(xb[2] - xo[2]);
    } else {
        double Xb[kIso], Xo[kIso], E[kIso], qd[4];
        iso_from_tq(xb, Xb);
        iso_from_tq(xo, Xo);
        iso_inv_mul(Xb, Xo, E);
        R_to_quat(E, qd);
        diff[6 * i] = E[9]; diff[6 * i + 1] = E[10]; diff[6 * i + 2] = E[11];
        diff[6 * i + 3] = qd[0]; diff[6 * i + 4] = qd[1]; diff[6 * i + 5] = qd[2];
    }
}


// ------------------------------------------------------------------------------------------ large blankets
// GLC Dense on a blanket too large for the LDS kernel (clusters of an SE3 lattice reach k + m = 150-200
// vertices, n + nm ~ 1000): the same steps as VertexRemover::remove + TopologyProviderGLC::getEdge on dense matrices
// in HBM, the O(n^3) parts on the fp64 matrix cores (tile_abt_kernel):
//   H over [removed | kept]            dense_assemble_kernel (binary + n-ary GLC edges)        src/vertex_remover.cpp:397-402
//   Lambda_t = H_kk - H_mk^T H_mm^-1 H_mk   blocked right-looking Cholesky stopped at the kept block: what it has
//                                      made of that block by then IS the Schur complement              :409-449
//   J = d reparam / d x (block arrow), T = J^-1 (block arrow too), M = sym(T^T Lambda_t T)            topology_provider_glc.cpp:73-84
//   W with W^T W = M                   the reference takes eig(M) and keeps lambda >= 1e-8 (:59-71). M is exactly
//                                      [0 0; 0 M_rel] (a blanket of relative measurements carries no absolute information
//                                      on its first vertex), so W = [0 | L^T] with M_rel = L L^T — the blocked fp64-MFMA
//                                      Cholesky again — has the same W^T W whenever every eigenvalue of M_rel passes the
//                                      cut; that is PROVEN per blanket by trace(M_rel^-1) = ||L^-1||_F^2 < 1e8
//                                      (=> lambda_min > 1e-8). Otherwise the blanket reports SPG_ST_EIG_FAIL: the
//                                      truncating eigen-decomposition of a rank-deficient n ~ 1000 target is not built.
struct BigArrow {
    double *Jinv;      // k x 2 x DD: per kept vertex i the blocks (T_i0 | T_ii) of T = J^-1 (vertex 0: (unused | T_00))
    double *meas;      // n: reparametrised measurement
};

// GLCReparamBinary (src/glc_reparam_binary.hpp:35-127) for the k kept vertices (arena poses at vpo[m + i]): the
// measurement and the blocks of T = J^-1. One lane per vertex.
template <int D>
__global__ void big_reparam_kernel(const double *arena, const int64_t *vpo, int m, int k, BigArrow out, int *bad) {
    constexpr int DD = D * D, PSZ = (D == 6) ? kIso : 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    double X0[PSZ], Xi[PSZ], Ji[DD], Jj[DD];
    load_pose<D>(arena, vpo[m], X0);
    load_pose<D>(arena, vpo[m + i], Xi);
    double *ms = out.meas + i * D;
    if (D == 6) {
        double Z[kIso], qv[4];
        if (i == 0) {
            double Xz[kIso] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < kIso; t++) Z[t] = X0[t];
            se3_edge_jac(Xz, X0, Z, Ji, Jj, nullptr);
        } else {
            iso_inv_mul(X0, Xi, Z);
            se3_edge_jac(X0, Xi, Z, Ji, Jj, nullptr);
        }
        R_to_quat(Z, qv);
        ms[0] = Z[9]; ms[1] = Z[10]; ms[2] = Z[11]; ms[3] = qv[0]; ms[4] = qv[1]; ms[5] = qv[2];
    } else {
        double z[3], xz[3] = {0, 0, 0};
        const double *xa = (i == 0) ? xz : X0;
        se2_between(xa, Xi, z);
        z[2] = normalize_theta(z[2]);
        se2_edge_jac(xa, Xi, z, Ji, Jj, nullptr);
        ms[0] = z[0]; ms[1] = z[1]; ms[2] = z[2];
    }
    // row i of J: (J_i0 = Ji | J_ii = Jj), row 0: J_00 = Jj. T_ii = J_ii^-1, T_i0 = -J_ii^-1 J_i0 J_00^-1 (second factor below)
    double Tii[DD];
    if (!small_inverse<D>(Jj, Tii)) *bad = 1;
    double *dst = out.Jinv + (size_t)i * 2 * DD;
    double Tmp[DD];
#pragma unroll
    for (int r = 0; r < D; r++)
#pragma unroll
        for (int c = 0; c < D; c++) {
            double sacc = 0;
#pragma unroll
            for (int p = 0; p < D; p++) sacc += Tii[r * D + p] * Ji[p * D + c];
            Tmp[r * D + c] = (i == 0) ? 0.0 : -sacc;     // -J_ii^-1 J_i0 (times J_00^-1 in big_arrow_finish_kernel)
        }
#pragma unroll
    for (int t = 0; t < DD; t++) { dst[t] = Tmp[t]; dst[DD + t] = Tii[t]; }
}
// T_i0 <- T_i0 * T_00 for i > 0
template <int D>
__global__ void big_arrow_finish_kernel(int k, BigArrow out) {
    constexpr int DD = D * D;
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const double *T00 = out.Jinv + DD;
    double *Ti0 = out.Jinv + (size_t)i * 2 * DD;
    double A[DD], R[DD];
#pragma unroll
    for (int t = 0; t < DD; t++) A[t] = Ti0[t];
#pragma unroll
    for (int r = 0; r < D; r++)
#pragma unroll
        for (int c = 0; c < D; c++) {
            double sacc = 0;
#pragma unroll
            for (int p = 0; p < D; p++) sacc += A[r * D + p] * T00[p * D + c];
            R[r * D + c] = sacc;
        }
#pragma unroll
    for (int t = 0; t < DD; t++) Ti0[t] = R[t];
}

// Y = S T for symmetric S (n x n, stored lower incl. diagonal in Lam with leading dimension ldl) and block-arrow T:
// column block b > 0: Y[:, b] = S[:, b] T_bb ; column block 0: Y[:, 0] = S[:, 0] T_00 + sum_{c > 0} S[:, c] T_c0.
// One wavefront per row; lanes over (block, column) pairs.
template <int D>
__global__ __launch_bounds__(64) void big_right_mul_kernel(const double *Lam, int ldl, int k, const double *Jinv, double *Y, int ldy) {
    constexpr int DD = D * D;
    const int i = blockIdx.x, n = D * k, lane = threadIdx.x;
    auto S = [&](int r, int c) { return (c <= r) ? Lam[(long long)r * ldl + c] : Lam[(long long)c * ldl + r]; };
    for (int j = D + lane; j < n; j += 64) {
        const int b = j / D, c = j - b * D;
        const double *Tbb = Jinv + (size_t)b * 2 * DD + DD;
        double sacc = 0;
#pragma unroll
        for (int p = 0; p < D; p++) sacc += S(i, b * D + p) * Tbb[p * D + c];
        Y[(long long)i * ldy + j] = sacc;
    }
    // the d columns of block 0 are sums over the whole row: the lanes share the row (entry t of it per lane and step),
    // each with d partial sums, and a butterfly adds them up (one lane per column used to walk the n terms alone)
    double part[D];
#pragma unroll
    for (int c = 0; c < D; c++) part[c] = 0.0;
    for (int t = lane; t < n; t += 64) {
        const int cb = t / D, p = t - cb * D;
        const double *Tc = (cb == 0) ? Jinv + DD : Jinv + (size_t)cb * 2 * DD;
        const double sv = S(i, t);
#pragma unroll
        for (int c = 0; c < D; c++) part[c] += sv * Tc[p * D + c];
    }
#pragma unroll
    for (int c = 0; c < D; c++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part[c] += __shfl_xor(part[c], o, 64);
        if (lane == c) Y[(long long)i * ldy + c] = part[c];
    }
}
// Mt = T^T Y (n x n): row block a > 0: T_aa^T Y[a, :]; row block 0: T_00^T Y[0, :] + sum_{c > 0} T_c0^T Y[c, :].
// blockIdx.x < n - d: row d + blockIdx.x, lanes over the columns. The remaining workgroups take the d rows of block 0 —
// sums over ALL rows of Y — for 64 columns each: lane = column, the four wavefronts split the row blocks of Y (one read of
// Y serves the d rows), partial sums meet in LDS. (One wavefront per such row walked the n terms of each of its columns
// alone: 243 us, four times the rest of the kernel.)
template <int D>
__global__ __launch_bounds__(256) void big_left_mul_kernel(const double *Y, int ldy, int k, const double *Jinv, double *Mt, int ldm) {
    constexpr int DD = D * D;
    __shared__ double red[3][D][64];
    const int n = D * k, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if ((int)blockIdx.x < n - D) {
        const int i = D + blockIdx.x, a = i / D, r = i - a * D;
        const double *Taa = Jinv + (size_t)a * 2 * DD + DD;
        for (int j = tid; j < n; j += 256) {
            double sacc = 0;
#pragma unroll
            for (int p = 0; p < D; p++) sacc += Taa[p * D + r] * Y[(long long)(a * D + p) * ldy + j];
            Mt[(long long)i * ldm + j] = sacc;
        }
        return;
    }
    const int j = ((int)blockIdx.x - (n - D)) * 64 + lane;
    double acc[D];
#pragma unroll
    for (int r = 0; r < D; r++) acc[r] = 0.0;
    if (j < n)
        for (int cb = w; cb < k; cb += 4) {
            const double *Tc = (cb == 0) ? Jinv + DD : Jinv + (size_t)cb * 2 * DD;
            double yv[D];
#pragma unroll
            for (int p = 0; p < D; p++) yv[p] = Y[(long long)(cb * D + p) * ldy + j];
#pragma unroll
            for (int p = 0; p < D; p++)
#pragma unroll
                for (int r = 0; r < D; r++) acc[r] += Tc[p * D + r] * yv[p];
        }
    if (w > 0) {
#pragma unroll
        for (int r = 0; r < D; r++) red[w - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (w == 0 && j < n) {
#pragma unroll
        for (int r = 0; r < D; r++) Mt[(long long)r * ldm + j] = acc[r] + red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    }
}
// Mrel (Nr x Nr, zeroed, padded with a unit diagonal) <- sym(Mt)[D.., D..] ; stats[0] = max |sym(Mt)| over the first D
// rows / columns (the absolute part, ~0), stats[1] = max |sym(Mt)| overall
template <int D>
__global__ __launch_bounds__(256) void big_extract_kernel(const double *Mt, int ldm, int n, double *Mrel, int Nr, double *stats) {
    __shared__ double red[2][256];
    double ga = 0, gm = 0;
    const long long tot = (long long)n * n;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < tot; it += (long long)gridDim.x * 256) {
        const int i = (int)(it / n), j = (int)(it - (long long)i * n);
        if (j > i) continue;
        const double v = 0.5 * (Mt[(long long)i * ldm + j] + Mt[(long long)j * ldm + i]);
        gm = fmax(gm, fabs(v));
        if (j < D) ga = fmax(ga, fabs(v));
        else Mrel[(long long)(i - D) * Nr + (j - D)] = v;
    }
    red[0][threadIdx.x] = ga; red[1][threadIdx.x] = gm;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] = fmax(red[0][threadIdx.x], red[0][threadIdx.x + o]); red[1][threadIdx.x] = fmax(red[1][threadIdx.x], red[1][threadIdx.x + o]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // (fp64 max through the integer ordering of non-negative doubles)
        atomicMax(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)__double_as_longlong(red[0][0]));
        atomicMax(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)__double_as_longlong(red[1][0]));
    }
}
// the edge record: meas (n) then W (r x n) = [0 | L^T], r = n - D
template <int D>
__global__ __launch_bounds__(256) void big_write_record_kernel(const double *L, int Nr, int n, const double *meas, double *rec) {
    const int r = n - D;
    const long long tot = (long long)n + (long long)r * n;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < tot; it += (long long)gridDim.x * 256) {
        if (it < n) { rec[it] = meas[it]; continue; }
        const long long w = it - n;
        const int e = (int)(w / n), c = (int)(w - (long long)e * n);
        double v = 0.0;
        if (c >= D) { const int j = c - D; if (j >= e) v = L[(long long)j * Nr + e]; }
        rec[it] = v;
    }
}
// Whether the Cholesky shortcut stands: flags[0] = H_mm not PD, flags[1] = a reparametrisation block is singular,
// flags[2] = M_rel not PD; stats[0] / stats[1] = largest entry of the absolute rows / of M; tr_inv = trace(M_rel^-1).
__device__ inline int big_shortcut_status(const int *flags, const double *stats, double tr_inv) {
    if (flags[0]) return SPG_ST_HMM_NOT_PD;
    if (flags[1] || !isfinite(stats[1])) return SPG_ST_NONFINITE;
    if (flags[2] || !(tr_inv < 1e8) || !(stats[0] <= 1e-9 * fmax(stats[1], 1e-300))) return SPG_ST_EIG_FAIL;   // needs the truncating eig route
    return SPG_OK;
}
// The truncating eigen route of glc_chol (src/topology_provider_glc.cpp:59-71) for a large blanket whose target fails the
// shortcut's guard (rank-deficient beyond the gauge, or lambda_min < 1e-8): eigen-decomposition of the whole n x n
// M = sym(Mt) by one workgroup (tridiagonalisation + implicit QL on matrices in HBM, spg_dev_la.hpp — the rare path: 0.3 s
// at n ~ 600; parallel-order Jacobi below 128 variables, where SPG_FORCE_BIG sends test blankets), rows sqrt(lambda) v^T for
// lambda >= 1e-8 in ascending order, as the LDS kernels emit them. A, V: n x n scratch with leading dimension lda;
// cs: 3 n + 2 doubles, perm: n ints; dynamic LDS: 6 n + 8 doubles. flags[4] = 1 done / 2 no convergence, flags[5] = rows kept.
__global__ __launch_bounds__(256) void big_glc_eig_kernel(int *flags, const double *stats, const double *partial, int np, const double *Mt, int ldm, int n,
                                                           int rmax, double *A, double *V, int lda, double *cs, int *perm, const double *meas, double *rec) {
    constexpr int NT = 256;
    extern __shared__ double eig_lds[];
    __shared__ double red[NT / 64];
    __shared__ int flag_s;
    const int tid = threadIdx.x;
    const Team<NT> T{tid, red, &flag_s};
    double sp = 0;
    for (int i = tid; i < np; i += NT) sp += partial[i];
    const double tr_inv = T.sum(sp);
    if (big_shortcut_status(flags, stats, tr_inv) != SPG_ST_EIG_FAIL) return;
    for (long long it = tid; it < (long long)n * n; it += NT) {
        const int i = (int)(it / n), j = (int)(it - (long long)i * n);
        A[(long long)i * lda + j] = 0.5 * (Mt[(long long)i * ldm + j] + Mt[(long long)j * ldm + i]);
    }
    T.sync();
    const bool eok = (n >= 128 && (6 * n + 8) * 8 <= 140 * 1024) ? tridiag_eigh<NT>(T, A, V, n, lda, cs, eig_lds) : jacobi_eigh<NT>(T, A, V, n, lda, cs);
    if (!eok) { if (tid == 0) flags[4] = 2; return; }
    double *ev = cs;
    for (int i = tid; i < n; i += NT) ev[i] = A[(long long)i * lda + i];
    T.sync();
    sort_ascending<NT>(T, ev, 1, n, perm);
    double below = 0;
    for (int i = tid; i < n; i += NT) below += (ev[i] < 1e-8) ? 1.0 : 0.0;      // glc_eps, src/topology_provider_glc.cpp:18,67
    int i0 = (int)T.sum(below);
    if (n - i0 > rmax) i0 = n - rmax;     // (the record has room for n - D rows: the gauge directions never pass the cut)
    const int r = n - i0;
    for (long long it = tid; it < (long long)n + (long long)r * n; it += NT) {
        if (it < n) { rec[it] = meas[it]; continue; }
        const long long w = it - n;
        const int e = (int)(w / n), c = (int)(w - (long long)e * n), col = perm[i0 + e];
        rec[it] = V[(long long)c * lda + col] * sqrt(ev[col]);
    }
    if (tid == 0) { flags[4] = 1; flags[5] = r; }
}
// out record of the blanket (layout in include/spg.h). flags / stats as above; partial[0..np) = sum of squares of L^-T
// (trace of M_rel^-1); flags[4], flags[5] = outcome of the eigen route when the shortcut's guard failed.
__global__ void big_out_record_kernel(double *orec, const int *flags, const double *stats, const double *partial, int np, int n, int D_, int m, int k,
                                      int n_new_max, int tag, long long rec_len) {
    __shared__ double red[256];
    double s = 0;
    for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x != 0) return;
    const double tr_inv = red[0] - 0.0;
    int status = big_shortcut_status(flags, stats, tr_inv);
    int n_new = (status == SPG_OK) ? 1 : 0;
    if (status == SPG_ST_EIG_FAIL && flags[4] == 1) {
        status = SPG_OK;
        n_new = flags[5] > 0 ? 1 : 0;
        rec_len = (long long)n + (long long)flags[5] * n;
    }
    orec[0] = (double)status; orec[1] = 0.0; orec[2] = __builtin_nan(""); orec[3] = __builtin_inf(); orec[4] = (double)n_new;
    if (n_new) {
        orec[SPG_OUT_HDR + 0] = (double)SPG_EDGE_GLC;
        orec[SPG_OUT_HDR + 1] = 0.0;
        orec[SPG_OUT_HDR + 2] = (double)rec_len;
        orec[SPG_OUT_HDR + 3] = (double)k;
        for (int i = 0; i < k; i++) orec[SPG_OUT_HDR + 4 * n_new_max + i] = (double)(m + i);
    }
    __threadfence_system();
    __hip_atomic_store(&orec[5], SPG_FINAL_WORD(tag), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------ host side
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

template <class T>
int upload(DevBuf &b, const T *src, size_t n, hipStream_t s) {
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (hipMalloc(&b.p, bytes) != hipSuccess) return SPG_ENOMEM;
    (void)s;   // synchronous: the sources are short-lived pageable host vectors
    if (n && hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return SPG_EHIP;
    return 0;
}

struct GraphBufs {
    DevBuf pos, vpo, rowptr, inc, er, ev, awoff, aw;
    GraphDev dev{};
    int max_q = 0;
    bool has_glc = false;
};

// per-edge offsets of the weighted Jacobians of n-ary edges (A_e and the weighted error) and what their staging needs
int plan_graph(const spg::DenseGraphIn &in, GraphBufs &gb, std::vector<int64_t> &awoff, int64_t &aw_total) {
    awoff.assign((size_t)in.ne, -1);
    aw_total = 0;
    for (int e = 0; e < in.ne; e++) {
        if (in.er[e].kind == SPG_EDGE_BINARY) continue;
        int q = in.er[e].nv, dq = in.D * q;
        if (in.er[e].kind == SPG_EDGE_MULTI) {
            // r = d nm rows; nm = q - 1 for the patterns that emit such edges, never more than len allows
            int nm = 1;
            while (SPG_MULTI_LEN(in.D, nm) < in.er[e].len) nm++;
            if (q < 2 || SPG_MULTI_LEN(in.D, nm) != in.er[e].len) return SPG_EINVAL;
            awoff[e] = aw_total;
            aw_total += (int64_t)in.D * nm * dq + in.D * nm;
            gb.max_q = std::max(gb.max_q, std::max(q, nm + 1));
            gb.has_glc = true;
            continue;
        }
        if (q <= 0 || in.er[e].len < dq || (in.er[e].len - dq) % dq) return SPG_EINVAL;
        awoff[e] = aw_total;
        aw_total += (int64_t)(in.er[e].len - dq) + (in.er[e].len - dq) / dq;   // A_e (r x dq) + weighted error (r)
        gb.max_q = std::max(gb.max_q, q);
        gb.has_glc = true;
    }
    if ((size_t)gb.max_q * (2 * in.D * in.D + in.D) * sizeof(double) > 160 * 1024) return SPG_ECAPACITY;   // LDS staging of one GLC edge's Jacobians
    return 0;
}

int stage_graph(const spg::DenseGraphIn &in, GraphBufs &gb, hipStream_t s) {
    std::vector<int64_t> awoff;
    int64_t aw_total = 0;
    int rc;
    if ((rc = plan_graph(in, gb, awoff, aw_total))) return rc;
    if ((rc = upload(gb.pos, in.pos, (size_t)in.nv, s))) return rc;
    if ((rc = upload(gb.vpo, in.vpo, (size_t)in.nv, s))) return rc;
    if ((rc = upload(gb.rowptr, in.rowptr, (size_t)in.nv + 1, s))) return rc;
    if ((rc = upload(gb.inc, in.inc, (size_t)in.rowptr[in.nv], s))) return rc;
    if ((rc = upload(gb.er, in.er, (size_t)in.ne, s))) return rc;
    if ((rc = upload(gb.ev, in.ev, (size_t)in.n_ev, s))) return rc;
    if ((rc = upload(gb.awoff, awoff.data(), awoff.size(), s))) return rc;
    if (hipMalloc(&gb.aw.p, std::max<int64_t>(aw_total, 1) * 8) != hipSuccess) return SPG_ENOMEM;
    gb.dev = GraphDev{(const double *)in.dev_arena, (const int32_t *)gb.pos.p, (const int64_t *)gb.vpo.p,
                      (const int32_t *)gb.rowptr.p, (const int32_t *)gb.inc.p, (const spg_edge_ref *)gb.er.p,
                      (const int32_t *)gb.ev.p, (const int64_t *)gb.awoff.p, (double *)gb.aw.p, in.nv, in.ne};
    return 0;
}

template <int D>
void launch_glc_jacobians(const GraphBufs &gb, hipStream_t s) {
    if (!gb.has_glc) return;
    size_t sh = (size_t)gb.max_q * (2 * D * D + D) * sizeof(double);
    // (a launch whose dynamic LDS exceeds what the attribute allows is rejected: the error surfaces at the caller's
    //  hipGetLastError / stream synchronisation as SPG_EHIP, never as a fault)
    if (sh > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(glc_weighted_jacobian_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) return;
    hipLaunchKernelGGL((glc_weighted_jacobian_kernel<D>), dim3(gb.dev.ne), dim3(64), sh, s, gb.dev);
}

template <int D, class Sink>
void launch_assemble_into(const GraphBufs &gb, Sink sink, hipStream_t s, double *bvec = nullptr) {
    launch_glc_jacobians<D>(gb, s);
    hipLaunchKernelGGL((dense_assemble_kernel<D, Sink>), dim3(gb.dev.nv), dim3(64), 0, s, gb.dev, sink, bvec);
}

template <int D>
void launch_assemble(const GraphBufs &gb, double *M, int ld, hipStream_t s, double *bvec = nullptr) {
    launch_assemble_into<D>(gb, DenseSink{M, ld}, s, bvec);
}

constexpr int kSmallTileSteps = 128;   // steps of at most this many 128-tiles take tile_abt_small_kernel

// returns true when the step also factorised its tile (0, 0) (op.diag_linv set and the latency form taken)
bool launch_tiles(const TileOp &op, hipStream_t s) {
    if (op.P64 <= 0 || op.Q64 <= 0) return false;
    const int gp = (op.P64 + 1) / 2, gq = (op.Q64 + 1) / 2;
    const int big_tiles = op.tri ? gp * (gp + 1) / 2 : gp * gq;
    if (big_tiles <= kSmallTileSteps) {      // a step that cannot fill the chip: latency form
        if (op.tri) hipLaunchKernelGGL(tile_abt_small_kernel, dim3(op.P64 * (op.P64 + 1) / 2), dim3(256), 0, s, op);
        else hipLaunchKernelGGL(tile_abt_small_kernel, dim3(op.P64, op.Q64), dim3(256), 0, s, op);
        return op.diag_linv != nullptr;
    }
    if (op.tri) hipLaunchKernelGGL(tile_abt_kernel, dim3(gp * (gp + 1) / 2), dim3(256), 0, s, op);
    else hipLaunchKernelGGL(tile_abt_kernel, dim3(gp, gq), dim3(256), 0, s, op);
    return false;
}

// Blocked right-looking lower Cholesky of the N x N matrix M (N a multiple of 64), in place.
// Two levels: a 512-wide outer block is factorised as eight 64-wide panels (diagonal block in one
// wavefront, panel solve = product with L_jj^-T, update of the remaining columns of the outer block),
// then the whole trailing matrix is updated ONCE with K = 512 — the trailing read-modify-write of C is
// what bounds a K = 64 update (10 flop/B), and it shrinks with the outer width.
constexpr int OUTER = 8;   // 64-tiles per outer block
// stop_tiles >= 0: factorise the first stop_tiles block columns only — the trailing block then holds the Schur complement
// of the leading one (lower triangle), which is how a blanket's target information is formed (big blankets below).
// linv_keep != nullptr: the inverses of the diagonal blocks are kept, block j at linv_keep + j * 64 * 64 (a triangular solve with
// the factor then needs no pass of its own over the diagonal)
void potrf_lower(double *M, int N, double *linv /*64*64*/, int *bad, hipStream_t s, int stop_tiles = -1, double *linv_keep = nullptr) {
    const int nt = N / TB;
    const int lim = stop_tiles < 0 ? nt : std::min(stop_tiles, nt);
    const long long ld = N;
    auto linv_of = [&](int j) { return linv_keep ? linv_keep + (long long)j * TB * TB : linv; };
    bool diag_done = false;      // the step that last updated the coming diagonal block has factorised it as well
    for (int J0 = 0; J0 < lim; J0 += OUTER) {
        const int J1 = std::min(lim, J0 + OUTER);
        for (int j = J0; j < J1; j++) {
            double *lj = linv_of(j);
            if (!diag_done) hipLaunchKernelGGL(diag_potrf_kernel, dim3(1), dim3(64), 0, s, M + (long long)j * TB * (ld + 1), N, lj, bad, 1);
            diag_done = false;
            const int rem = nt - j - 1;
            if (rem == 0) break;
            double *panel = M + ((long long)(j + 1) * TB * ld + (long long)j * TB);
            // panel <- panel * L_jj^-T
            TileOp trsm{panel, panel, lj, TB * ld, 0, TB * ld, 0, N, N, TB, rem, 1, 0, 0, 2};
            launch_tiles(trsm, s);
            // remaining columns of the outer block: M[j+1.., j+1..J1) -= panel * panel[0..J1-j-1]^T
            // (the few tiles above the diagonal that this rectangle covers are never read); its tile (0, 0) is the next
            // panel's diagonal block
            const int inner = J1 - j - 1;
            if (inner > 0) {
                double *sub = M + (long long)(j + 1) * TB * (ld + 1);
                TileOp upd{sub, panel, panel, TB * ld, TB, TB * ld, TB * ld, N, N, N, rem, inner, 0, 1, 2};
                upd.diag_linv = linv_of(j + 1); upd.diag_bad = bad;
                diag_done = launch_tiles(upd, s);
            }
        }
        const int rem = nt - J1;
        if (rem <= 0) break;
        // trailing(lower) -= P * P^T with P = M[J1.., J0*64 .. J1*64); its tile (0, 0) opens the next outer block — unless the
        // factorisation stops here and that block is the Schur complement the caller wants
        double *P = M + ((long long)J1 * TB * ld + (long long)J0 * TB);
        double *trail = M + (long long)J1 * TB * (ld + 1);
        TileOp syrk{trail, P, P, TB * ld, TB, TB * ld, TB * ld, N, N, N, rem, rem, 1, 1, 2 * (J1 - J0)};
        if (J1 < lim) { syrk.diag_linv = linv_of(J1); syrk.diag_bad = bad; }
        diag_done = launch_tiles(syrk, s);
    }
}

// Y <- Y * Ls^-T for upper-triangular Y (Ng x Ng, ldy) and lower-triangular Ls (ld), tile-aligned.
// Same two levels as the factorisation: inside an outer block of OUTER column tiles the columns are
// scaled (product with L_cc^-T) and the later columns of the block updated with K = 64; the columns to
// the right of the block take ONE update with K = 64 * OUTER.
// have_linv: linv_all already holds the inverses of the diagonal blocks of Ls (potrf_lower with linv_keep)
void rsolve_lower_transposed(double *Y, int ldy, const double *Ls, int ld, int Ng, double *linv_all, int *bad, hipStream_t s, bool have_linv = false) {
    const int nt = Ng / TB;
    if (!have_linv) hipLaunchKernelGGL(diag_potrf_kernel, dim3(nt), dim3(64), 0, s, const_cast<double *>(Ls), ld, linv_all, bad, 0);
    for (int C0 = 0; C0 < nt; C0 += OUTER) {
        const int C1 = std::min(nt, C0 + OUTER);
        for (int c = C0; c < C1; c++) {
            double *col = Y + (long long)c * TB;
            // rows 0..c of block column c: Y[:, c] <- Y[:, c] * L_cc^-T
            TileOp scale{col, col, linv_all + (long long)c * TB * TB, (long long)TB * ldy, 0, (long long)TB * ldy, 0, ldy, ldy, TB, c + 1, 1, 0, 0, 2};
            launch_tiles(scale, s);
            const int inner = C1 - c - 1;
            if (inner > 0) {
                // Y[0..c, c'] -= Y[0..c, c] * Ls[c', c]^T for the remaining columns c' of the outer block
                const double *lpanel = Ls + ((long long)(c + 1) * TB * ld + (long long)c * TB);
                TileOp upd{Y + (long long)(c + 1) * TB, col, lpanel, (long long)TB * ldy, TB, (long long)TB * ldy, (long long)TB * ld, ldy, ldy, ld, c + 1, inner, 0, 1, 2};
                launch_tiles(upd, s);
            }
        }
        const int rem = nt - C1;
        if (rem <= 0) break;
        // Y[0..C1-1, c' >= C1] -= Y[0..C1-1, C0..C1-1] * Ls[c', C0..C1-1]^T   (K = 64 * (C1 - C0))
        const double *lblock = Ls + ((long long)C1 * TB * ld + (long long)C0 * TB);
        TileOp upd{Y + (long long)C1 * TB, Y + (long long)C0 * TB, lblock, (long long)TB * ldy, TB, (long long)TB * ldy, (long long)TB * ld, ldy, ldy, ld, C1, rem, 0, 1, 2 * (C1 - C0)};
        launch_tiles(upd, s);
    }
}

inline int round_up(int n) { return (n + TB - 1) / TB * TB; }

// Scratch of the large-blanket pipeline: one device block and one pinned staging block per process, grown on demand and
// kept (a blanket used to pay 20 hipMalloc / hipFree pairs and 7 synchronous copies). Calls are serialised by `mu`; each
// ends with a stream synchronisation, so the blocks are idle when the next call lays them out.
struct BigPool {
    std::mutex mu;
    int device = -1;
    char *dev = nullptr, *pin = nullptr;
    size_t dcap = 0, pcap = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    void drop() {
        if (dev) (void)hipFree(dev);
        if (pin) (void)hipHostFree(pin);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        dev = pin = nullptr; dcap = pcap = 0; e0 = e1 = nullptr; device = -1;
    }
};
BigPool &big_pool() { static BigPool *p = new BigPool; return *p; }

struct Carver {      // offsets into a block, 256-byte aligned
    size_t used = 0;
    size_t take(size_t bytes) { size_t at = used; used += (bytes + 255) & ~(size_t)255; return at; }
};

// One large GLC Dense blanket, asynchronously on `stream` up to the final synchronisation.
// in: the blanket as a local graph (vertices = blanket-local indices, removed first; pos[l] = l*D for removed vertices,
// Nm + (l - m)*D for kept ones, Nm = nm rounded up to 64). The new edge record goes to arena[new_off ...), the out
// record (include/spg.h) to orec (device address: arena or pinned mailbox).
template <int D>
static int big_glc_dense_impl(hipStream_t s, const spg::DenseGraphIn &in, int m, int k, int Nm, int64_t new_off, double *orec, int n_new_max, int tag,
                              double *seconds, char *err, size_t errlen) {
    int rc = 0;
    constexpr int DD = D * D;
    const int n = D * k, nm = D * m, Ng = round_up(std::max(n, 1)), N = Nm + Ng, Nr = round_up(std::max(n - D, 1)), ntr = Nr / TB;
    double *arena = (double *)const_cast<void *>(in.dev_arena);
    const int np = 256;
    float ms = 0;
    BigPool &P = big_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    int dev_now = 0;
    GraphBufs gb;                      // (its DevBufs stay empty: everything lives in the pool)
    std::vector<int64_t> awoff;
    int64_t aw_total = 0;
    if ((rc = plan_graph(in, gb, awoff, aw_total))) { snprintf(err, errlen, "staging a large blanket failed (%d)", rc); return rc; }
    // ---- layout: matrices, small state, then the staged block (the graph arrays), mirrored in pinned memory
    Carver dc, pc;
    const size_t oH = dc.take((size_t)N * N * 8), oY = dc.take((size_t)Ng * Ng * 8), oMt = dc.take((size_t)Ng * Ng * 8);
    const size_t oMrel = dc.take((size_t)Nr * Nr * 8), oYi = dc.take((size_t)Nr * Nr * 8);      // adjacent: cleared together
    const size_t oLinv = dc.take(TB * TB * 8), oLinvAll = dc.take((size_t)ntr * TB * TB * 8);
    const size_t oState = dc.take(256);       // flags (8 ints) | stats (2 doubles) — cleared together
    const size_t oPartial = dc.take(np * 8), oJinv = dc.take((size_t)std::max(k, 1) * 2 * DD * 8), oMeas = dc.take((size_t)std::max(n, 1) * 8);
    const size_t oAw = dc.take((size_t)std::max<int64_t>(aw_total, 1) * 8);
    const size_t n_inc = (size_t)in.rowptr[in.nv];
    const size_t sPos = pc.take((size_t)in.nv * 4), sVpo = pc.take((size_t)in.nv * 8), sRow = pc.take(((size_t)in.nv + 1) * 4), sInc = pc.take(std::max<size_t>(n_inc, 1) * 4);
    const size_t sEr = pc.take(std::max<size_t>(in.ne, 1) * sizeof(spg_edge_ref)), sEv = pc.take(std::max<size_t>((size_t)in.n_ev, 1) * 4), sAwoff = pc.take(std::max<size_t>(in.ne, 1) * 8);
    const size_t oStaged = dc.take(pc.used);
    HIPCHK(hipGetDevice(&dev_now));
    if (P.device != dev_now || P.dcap < dc.used || P.pcap < pc.used) {
        if (P.device != dev_now) P.drop();
        if (P.dcap < dc.used) {
            if (P.dev) { (void)hipFree(P.dev); P.dev = nullptr; P.dcap = 0; }
            const size_t want = dc.used + dc.used / 4;
            if (hipMalloc((void **)&P.dev, want) != hipSuccess) {
                (void)hipGetLastError();
                snprintf(err, errlen, "hipMalloc of the dense matrices of a large blanket failed (N = %d, %.1f MB)", N, 1e-6 * (double)want);
                return SPG_ENOMEM;
            }
            P.dcap = want;
        }
        if (P.pcap < pc.used) {
            if (P.pin) { (void)hipHostFree(P.pin); P.pin = nullptr; P.pcap = 0; }
            const size_t want = pc.used + pc.used / 4;
            HIPCHK(hipHostMalloc((void **)&P.pin, want, hipHostMallocDefault));
            P.pcap = want;
        }
        if (!P.e0) { HIPCHK(hipEventCreate(&P.e0)); HIPCHK(hipEventCreate(&P.e1)); }
        P.device = dev_now;
    }
    {
        double *H = (double *)(P.dev + oH), *Y = (double *)(P.dev + oY), *Mt = (double *)(P.dev + oMt), *Mrel = (double *)(P.dev + oMrel), *Yi = (double *)(P.dev + oYi);
        double *linv = (double *)(P.dev + oLinv), *linv_all = (double *)(P.dev + oLinvAll), *partial = (double *)(P.dev + oPartial);
        double *jinv = (double *)(P.dev + oJinv), *meas = (double *)(P.dev + oMeas);
        int *flags = (int *)(P.dev + oState);
        double *stats = (double *)(P.dev + oState + 64);
        char *staged = P.dev + oStaged;
        // ---- the staged block
        memcpy(P.pin + sPos, in.pos, (size_t)in.nv * 4);
        memcpy(P.pin + sVpo, in.vpo, (size_t)in.nv * 8);
        memcpy(P.pin + sRow, in.rowptr, ((size_t)in.nv + 1) * 4);
        if (n_inc) memcpy(P.pin + sInc, in.inc, n_inc * 4);
        if (in.ne) memcpy(P.pin + sEr, in.er, (size_t)in.ne * sizeof(spg_edge_ref));
        if (in.n_ev) memcpy(P.pin + sEv, in.ev, (size_t)in.n_ev * 4);
        if (in.ne) memcpy(P.pin + sAwoff, awoff.data(), (size_t)in.ne * 8);
        HIPCHK(hipMemcpyAsync(staged, P.pin, pc.used, hipMemcpyHostToDevice, s));
        gb.dev = GraphDev{(const double *)in.dev_arena, (const int32_t *)(staged + sPos), (const int64_t *)(staged + sVpo),
                          (const int32_t *)(staged + sRow), (const int32_t *)(staged + sInc), (const spg_edge_ref *)(staged + sEr),
                          (const int32_t *)(staged + sEv), (const int64_t *)(staged + sAwoff), (double *)(P.dev + oAw), in.nv, in.ne};
        HIPCHK(hipEventRecord(P.e0, s));
        HIPCHK(hipMemsetAsync(H, 0, (size_t)N * N * 8, s));
        HIPCHK(hipMemsetAsync(Mrel, 0, (oYi - oMrel) + (size_t)Nr * Nr * 8, s));      // M_rel and Yi
        HIPCHK(hipMemsetAsync(flags, 0, 256, s));
        launch_assemble<D>(gb, H, N, s);
        if (Nm > nm) hipLaunchKernelGGL(pad_identity_kernel, dim3((Nm - nm + 255) / 256), dim3(256), 0, s, H, N, nm, Nm);
        if (Ng > n) hipLaunchKernelGGL(pad_identity_kernel, dim3((Ng - n + 255) / 256), dim3(256), 0, s, H, N, Nm + n, N);
        // Schur complement onto the kept block: the factorisation stops where that block begins
        potrf_lower(H, N, linv, flags, s, Nm / TB);
        const double *Lam = H + ((long long)Nm * N + Nm);
        BigArrow ar{jinv, meas};
        hipLaunchKernelGGL((big_reparam_kernel<D>), dim3((k + 63) / 64), dim3(64), 0, s, (const double *)arena, gb.dev.vpo, m, k, ar, flags + 1);
        if (k > 1) hipLaunchKernelGGL((big_arrow_finish_kernel<D>), dim3((k + 62) / 64), dim3(64), 0, s, k, ar);
        hipLaunchKernelGGL((big_right_mul_kernel<D>), dim3(n), dim3(64), 0, s, Lam, N, k, (const double *)jinv, Y, Ng);
        hipLaunchKernelGGL((big_left_mul_kernel<D>), dim3(n - D + (n + 63) / 64), dim3(256), 0, s, (const double *)Y, Ng, k, (const double *)jinv, Mt, Ng);
        hipLaunchKernelGGL((big_extract_kernel<D>), dim3(256), dim3(256), 0, s, (const double *)Mt, Ng, n, Mrel, Nr, stats);
        if (Nr > n - D) hipLaunchKernelGGL(pad_identity_kernel, dim3((Nr - (n - D) + 255) / 256), dim3(256), 0, s, Mrel, Nr, n - D, Nr);
        hipLaunchKernelGGL(pad_identity_kernel, dim3((Nr + 255) / 256), dim3(256), 0, s, Yi, Nr, 0, Nr);
        // lambda_min(M_rel) > 1e-8 is proven by trace(M_rel^-1) = ||L^-T||_F^2 < 1e8
        potrf_lower(Mrel, Nr, linv, flags + 2, s, -1, linv_all);
        rsolve_lower_transposed(Yi, Nr, Mrel, Nr, Nr, linv_all, flags + 3, s, true);
        hipLaunchKernelGGL(sumsq_kernel, dim3(np), dim3(256), 0, s, (const double *)Yi, Nr, n - D, partial);
        const long long rec_len = (long long)n + (long long)(n - D) * n;
        hipLaunchKernelGGL((big_write_record_kernel<D>), dim3(512), dim3(256), 0, s, (const double *)Mrel, Nr, n, (const double *)meas, arena + new_off);
        {
            // the eigen route, taken inside the kernel only when the shortcut's guard failed: A in Y (free since the left
            // multiplication), V and the small scratch in H (free since the Schur complement was read)
            double *Vs = H, *cs = Vs + (size_t)Ng * Ng;
            int *perm = reinterpret_cast<int *>(cs + 3 * n + 2);
            static_assert(TB >= 16, "the scratch behind V assumes N^2 - Ng^2 >= 4 n + 2");
            size_t eig_lds = n >= 128 ? ((size_t)6 * n + 8) * 8 : 0;
            if (eig_lds > 140 * 1024) eig_lds = 0;      // (beyond 2 986 variables: the Jacobi sweeps, which need no LDS)
            if (eig_lds > 64 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(big_glc_eig_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds));
            hipLaunchKernelGGL(big_glc_eig_kernel, dim3(1), dim3(256), eig_lds, s, flags, (const double *)stats, (const double *)partial, np,
                               (const double *)Mt, Ng, n, n - D, Y, Vs, Ng, cs, perm, (const double *)meas, arena + new_off);
        }
        hipLaunchKernelGGL(big_out_record_kernel, dim3(1), dim3(256), 0, s, orec, (const int *)flags, (const double *)stats, (const double *)partial, np,
                           n, D, m, k, n_new_max, tag, rec_len);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(P.e1, s));
        HIPCHK(hipStreamSynchronize(s));   // the pool is laid out anew by the next call
        HIPCHK(hipEventElapsedTime(&ms, P.e0, P.e1));
        if (seconds) *seconds = 1e-3 * ms;
    }
done:
    return rc;
}

}  // namespace

namespace spg {

// Dense information matrix of one graph (variables = vertices with pos >= 0, n = D * count), host
// output n x n row-major, full symmetric.
int hip_dense_information(void *stream, const DenseGraphIn &in, int n, double *out, char *err, size_t errlen) {
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
    const int N = round_up(std::max(n, 1));
    GraphBufs gb;
    DevBuf M;
    std::vector<double> h;
    if (hipMalloc(&M.p, (size_t)N * N * 8) != hipSuccess) { snprintf(err, errlen, "hipMalloc of a %d x %d matrix failed", N, N); return SPG_ENOMEM; }
    HIPCHK(hipMemsetAsync(M.p, 0, (size_t)N * N * 8, s));
    if ((rc = stage_graph(in, gb, s))) { snprintf(err, errlen, "staging the graph for dense assembly failed (%d)", rc); goto done; }
    if (in.D == 6) launch_assemble<6>(gb, (double *)M.p, N, s);
    else launch_assemble<3>(gb, (double *)M.p, N, s);
    HIPCHK(hipGetLastError());
    h.resize((size_t)N * N);
    HIPCHK(hipMemcpyAsync(h.data(), M.p, (size_t)N * N * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) out[(size_t)i * n + j] = (j <= i) ? h[(size_t)i * N + j] : h[(size_t)j * N + i];
done:
    return rc;
}

// GraphWrapperG2O::covariance() (src/graph_wrapper_g2o.cpp:368-373): info.llt().solve(I). Dense on the device:
// H = L L^T (blocked fp64-MFMA Cholesky), Y = L^-T by the blocked triangular solve on the identity, then
// Sigma = Y Y^T with one triangular sweep of the tile kernel (K = N). Host output n x n row-major, full symmetric.
int hip_dense_covariance(void *stream, const DenseGraphIn &in, int n, double *out, char *err, size_t errlen) {
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
    const int N = round_up(std::max(n, 1)), nt = N / TB;
    GraphBufs gb;
    DevBuf M, Y, C, linv, linv_all, bad;
    std::vector<double> h;
    int h_bad = 0;
    if (hipMalloc(&M.p, (size_t)N * N * 8) != hipSuccess || hipMalloc(&Y.p, (size_t)N * N * 8) != hipSuccess ||
        hipMalloc(&C.p, (size_t)N * N * 8) != hipSuccess) {
        snprintf(err, errlen, "hipMalloc of three %d x %d matrices failed", N, N);
        return SPG_ENOMEM;
    }
    HIPCHK(hipMalloc(&linv.p, TB * TB * 8));
    HIPCHK(hipMalloc(&linv_all.p, (size_t)nt * TB * TB * 8));
    HIPCHK(hipMalloc(&bad.p, sizeof(int)));
    HIPCHK(hipMemsetAsync(M.p, 0, (size_t)N * N * 8, s));
    HIPCHK(hipMemsetAsync(Y.p, 0, (size_t)N * N * 8, s));
    HIPCHK(hipMemsetAsync(C.p, 0, (size_t)N * N * 8, s));
    HIPCHK(hipMemsetAsync(bad.p, 0, sizeof(int), s));
    if ((rc = stage_graph(in, gb, s))) { snprintf(err, errlen, "staging the graph for dense assembly failed (%d)", rc); goto done; }
    if (in.D == 6) launch_assemble<6>(gb, (double *)M.p, N, s);
    else launch_assemble<3>(gb, (double *)M.p, N, s);
    if (N > n) hipLaunchKernelGGL(pad_identity_kernel, dim3((N - n + 255) / 256), dim3(256), 0, s, (double *)M.p, N, n, N);
    hipLaunchKernelGGL(pad_identity_kernel, dim3((N + 255) / 256), dim3(256), 0, s, (double *)Y.p, N, 0, N);
    potrf_lower((double *)M.p, N, (double *)linv.p, (int *)bad.p, s);
    rsolve_lower_transposed((double *)Y.p, N, (const double *)M.p, N, N, (double *)linv_all.p, (int *)bad.p, s);
    {
        TileOp yyt{(double *)C.p, (const double *)Y.p, (const double *)Y.p, (long long)TB * N, TB, (long long)TB * N, (long long)TB * N,
                   N, N, N, nt, nt, 1, 0, N / KC};
        launch_tiles(yyt, s);
    }
    HIPCHK(hipGetLastError());
    h.resize((size_t)N * N);
    HIPCHK(hipMemcpyAsync(h.data(), C.p, (size_t)N * N * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&h_bad, bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (h_bad) { snprintf(err, errlen, "covariance: the information matrix is not positive definite"); rc = SPG_ENOTPD; goto done; }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) out[(size_t)i * n + j] = (j <= i) ? h[(size_t)i * N + j] : h[(size_t)j * N + i];
done:
    return rc;
}

void hip_big_release_scratch() {
    BigPool &P = big_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    P.drop();
}

int hip_big_glc_dense(void *stream, const DenseGraphIn &in, int m, int k, int Nm, int64_t new_off, double *orec, int n_new_max, int tag,
                      double *seconds, double *flops, char *err, size_t errlen) {
    const double n = (double)in.D * k, nm = (double)in.D * m, nr = n - in.D;
    // the n^3-class work on the matrix cores: H_mm Cholesky + panel solves + Schur update, M_rel Cholesky, L^-T
    if (flops) *flops = nm * nm * nm / 3.0 + nm * nm * n + nm * n * n + nr * nr * nr / 3.0 + nr * nr * nr / 3.0;
    static const bool trace = [] { const char *e = getenv("SPG_BIG_TRACE"); return e && e[0] == '1'; }();      // diagnostic: host time of a call next to its device time
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = in.D == 6 ? big_glc_dense_impl<6>((hipStream_t)stream, in, m, k, Nm, new_off, orec, n_new_max, tag, seconds, err, errlen)
                             : big_glc_dense_impl<3>((hipStream_t)stream, in, m, k, Nm, new_off, orec, n_new_max, tag, seconds, err, errlen);
    if (trace) fprintf(stderr, "big blanket n=%d nm=%d: call %.3f ms, device %.3f ms\n", (int)n, (int)nm,
                       1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), seconds ? 1e3 * *seconds : 0.0);
    return rc;
}

// Global KLD. base.pos orders the baseline's variables [marginalised | pad | kept | pad] with the
// kept block starting at Nm (multiple of 64); other.pos orders other's variables [kept | pad].
// kept_vpo_*: pose offsets of the kept vertices in both arenas, in kept order.
int hip_dense_kld(void *stream, const DenseGraphIn &base, const DenseGraphIn &other, int n_marg, int n_keep,
                  const int64_t *kept_vpo_base, const int64_t *kept_vpo_other, double *terms, double *seconds,
                  char *err, size_t errlen) {
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
    const int D = base.D, nk = n_keep / D;
    const int Nm = round_up(n_marg), Ng = round_up(std::max(n_keep, 1)), N = Nm + Ng, ntg = Ng / TB;
    GraphBufs gb, go;
    DevBuf Mb, X, Y, linv, linv_all, bad, partial, outb, diff, rowsq, vb, vo;
    int h_bad[2] = {0, 0};
    double h_out[4] = {0, 0, 0, 0};
    const int np = 1024;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    if (hipMalloc(&Mb.p, (size_t)N * N * 8) != hipSuccess || hipMalloc(&X.p, (size_t)Ng * Ng * 8) != hipSuccess ||
        hipMalloc(&Y.p, (size_t)Ng * Ng * 8) != hipSuccess) {
        snprintf(err, errlen, "hipMalloc of the dense KLD matrices failed (N = %d, Ng = %d)", N, Ng);
        return SPG_ENOMEM;
    }
    HIPCHK(hipMalloc(&linv.p, TB * TB * 8));
    HIPCHK(hipMalloc(&linv_all.p, (size_t)ntg * TB * TB * 8));
    HIPCHK(hipMalloc(&bad.p, 2 * sizeof(int)));
    HIPCHK(hipMalloc(&partial.p, np * 8));
    HIPCHK(hipMalloc(&outb.p, 4 * 8));
    HIPCHK(hipMalloc(&diff.p, (size_t)Ng * 8));
    HIPCHK(hipMalloc(&rowsq.p, (size_t)Ng * 8));
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    if ((rc = upload(vb, kept_vpo_base, (size_t)nk, s)) || (rc = upload(vo, kept_vpo_other, (size_t)nk, s)) ||
        (rc = stage_graph(base, gb, s)) || (rc = stage_graph(other, go, s))) {
        snprintf(err, errlen, "staging the graphs for the dense KLD failed (%d)", rc);
        goto done;
    }
    HIPCHK(hipEventRecord(e0, s));
    HIPCHK(hipMemsetAsync(Mb.p, 0, (size_t)N * N * 8, s));
    HIPCHK(hipMemsetAsync(X.p, 0, (size_t)Ng * Ng * 8, s));
    HIPCHK(hipMemsetAsync(bad.p, 0, 2 * sizeof(int), s));
    HIPCHK(hipMemsetAsync(diff.p, 0, (size_t)Ng * 8, s));
    if (D == 6) { launch_assemble<6>(gb, (double *)Mb.p, N, s); launch_assemble<6>(go, (double *)X.p, Ng, s); }
    else { launch_assemble<3>(gb, (double *)Mb.p, N, s); launch_assemble<3>(go, (double *)X.p, Ng, s); }
    if (Nm > n_marg) hipLaunchKernelGGL(pad_identity_kernel, dim3((Nm - n_marg + 255) / 256), dim3(256), 0, s, (double *)Mb.p, N, n_marg, Nm);
    if (Ng > n_keep) {
        hipLaunchKernelGGL(pad_identity_kernel, dim3((Ng - n_keep + 255) / 256), dim3(256), 0, s, (double *)Mb.p, N, Nm + n_keep, N);
        hipLaunchKernelGGL(pad_identity_kernel, dim3((Ng - n_keep + 255) / 256), dim3(256), 0, s, (double *)X.p, Ng, n_keep, Ng);
    }
    if (D == 6) hipLaunchKernelGGL((pose_diff_kernel<6>), dim3((nk + 63) / 64), dim3(64), 0, s, (const double *)base.dev_arena, (const int64_t *)vb.p, (const double *)other.dev_arena, (const int64_t *)vo.p, nk, (double *)diff.p);
    else hipLaunchKernelGGL((pose_diff_kernel<3>), dim3((nk + 63) / 64), dim3(64), 0, s, (const double *)base.dev_arena, (const int64_t *)vb.p, (const double *)other.dev_arena, (const int64_t *)vo.p, nk, (double *)diff.p);
    potrf_lower((double *)Mb.p, N, (double *)linv.p, (int *)bad.p, s);
    potrf_lower((double *)X.p, Ng, (double *)linv.p, (int *)bad.p + 1, s);
    {
        const double *Ls = (const double *)Mb.p + ((long long)Nm * N + Nm);
        hipLaunchKernelGGL(transpose_lower_kernel, dim3(Ng / 32, Ng / 32), dim3(256), 0, s, (const double *)X.p, Ng, (double *)Y.p, Ng, Ng);
        // Mahalanobis and log-dets need Y = L_x^T before the solve overwrites it
        hipLaunchKernelGGL(upper_matvec_sq_kernel, dim3(Ng / 4), dim3(256), 0, s, (const double *)Y.p, Ng, Ng, (const double *)diff.p, (double *)rowsq.p);
        hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, s, (const double *)partial.p, 0, (const double *)X.p, Ng, Ls, N, Ng,
                           (const double *)Y.p, Ng, (const double *)rowsq.p, (double *)outb.p);
        HIPCHK(hipMemcpyAsync(h_out, outb.p, 4 * 8, hipMemcpyDeviceToHost, s));
        rsolve_lower_transposed((double *)Y.p, Ng, Ls, N, Ng, (double *)linv_all.p, (int *)bad.p, s);
        hipLaunchKernelGGL(sumsq_kernel, dim3(np), dim3(256), 0, s, (const double *)Y.p, Ng, Ng, (double *)partial.p);
        hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, s, (const double *)partial.p, np, (const double *)X.p, Ng, Ls, N, 0,
                           (const double *)Y.p, Ng, (const double *)nullptr, (double *)outb.p);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, s));
    {
        double h_sum[4];
        HIPCHK(hipStreamSynchronize(s));   // h_out (first finish) is in place
        double logdetx = h_out[1], logdet_s = h_out[2], mahal = h_out[3];
        HIPCHK(hipMemcpy(h_sum, outb.p, 4 * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_bad, bad.p, 2 * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (h_bad[0] || h_bad[1]) {
            snprintf(err, errlen, "global KLD: %s information matrix is not positive definite", h_bad[0] ? "the baseline" : "the sparsified");
            rc = SPG_ENOTPD;
            goto done;
        }
        const double innerprod = h_sum[0] - (double)(Ng - n_keep);   // the unit pad rows contribute 1 each
        const double logdety = -logdet_s;                            // reference sign (mode InformationInformation)
        terms[0] = 0.5 * (innerprod + mahal - logdetx - logdety - n_keep);
        terms[1] = innerprod; terms[2] = mahal; terms[3] = logdetx; terms[4] = logdety; terms[5] = n_keep;
        if (seconds) *seconds = 1e-3 * ms;
    }
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

// GraphWrapperG2O::optimize (src/graph_wrapper_g2o.cpp:250-269) = g2o Levenberg-Marquardt with one
// vertex fixed. in.pos gives every free vertex its scalar position in the solution vector. The poses are
// updated in place in the arena. LM control flow as in g2o's
// OptimizationAlgorithmLevenberg (an un-vendored dependency of the reference, restated from its
// published algorithm): lambda_0 = 1e-5 max|diag H|; per iteration up to 10 trials of
// (H + lambda I) x = b with gain ratio rho = (chi2 - chi2') / (x.(lambda x + b) + 1e-3); good step:
// lambda *= clamp(1 - (2 rho - 1)^3, 1/3, 2/3), ni = 2; bad step: lambda *= ni, ni *= 2, estimates
// restored; stop after 10 failed trials, rho == 0 or a non-finite lambda. The host sees four scalars
// per trial. The linear algebra behind it is either dense (DenseLM below) or the block-sparse multifrontal
// solver of spg_sparse.inc (SparseLM).
}  // namespace spg

namespace {

struct LMLinear {
    virtual ~LMLinear() {}
    // H and b = -sum J^T Omega e at the current estimates; scal[1] <- max |diag H|
    virtual int build(hipStream_t s, const GraphBufs &gb, double *b, double *scal, char *err, size_t errlen) = 0;
    // (H + lambda I) sol = b, enqueued on s; *bad (device) != 0 afterwards: not positive definite
    virtual int solve(hipStream_t s, const GraphBufs &gb, double lambda, const double *b, double *sol, int *bad, char *err, size_t errlen) = 0;
};

// n: number of unknowns; nvec >= n: length of the vectors (b, sol)
int lm_run(hipStream_t s, const spg::DenseGraphIn &in, GraphBufs &gb, int n, int nvec, int iterations, LMLinear &lin,
           double *stats, double *seconds, char *err, size_t errlen) {
    int rc = 0;
    const int D = in.D, PS = (D == 6) ? 7 : 3;
    DevBuf bad, b, sol, chi, scal, backup;
    double h_s[4] = {0, 0, 0, 0};
    int h_bad = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    double lambda = 0, ni = 2, chi_first = 0, chi_last = 0;
    int it = 0, trials = 0;
    HIPCHK(hipMalloc(&bad.p, sizeof(int)));
    HIPCHK(hipMalloc(&b.p, (size_t)nvec * 8));
    HIPCHK(hipMalloc(&sol.p, (size_t)nvec * 8));
    HIPCHK(hipMalloc(&chi.p, (size_t)std::max(in.ne, 1) * 8));
    HIPCHK(hipMalloc(&scal.p, 4 * 8));
    HIPCHK(hipMalloc(&backup.p, (size_t)std::max(in.nv, 1) * PS * 8));
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, s));
    {
        double *arena = (double *)const_cast<void *>(in.dev_arena);
        auto poses = [&](int mode) {
            if (D == 6) hipLaunchKernelGGL((pose_update_kernel<6>), dim3((in.nv + 63) / 64), dim3(64), 0, s, arena, gb.dev.vpo, gb.dev.pos, in.nv, (const double *)sol.p, (double *)backup.p, mode);
            else hipLaunchKernelGGL((pose_update_kernel<3>), dim3((in.nv + 63) / 64), dim3(64), 0, s, arena, gb.dev.vpo, gb.dev.pos, in.nv, (const double *)sol.p, (double *)backup.p, mode);
        };
        // chi2 of the current estimates into scal[slot] (the GLC kernel refreshes the weighted errors)
        auto chi2_into = [&](int slot, bool refresh_glc) {
            if (refresh_glc) {
                if (D == 6) launch_glc_jacobians<6>(gb, s);
                else launch_glc_jacobians<3>(gb, s);
            }
            if (in.ne > 0) {
                if (D == 6) hipLaunchKernelGGL((edge_chi2_kernel<6>), dim3((in.ne + 63) / 64), dim3(64), 0, s, gb.dev, (double *)chi.p);
                else hipLaunchKernelGGL((edge_chi2_kernel<3>), dim3((in.ne + 63) / 64), dim3(64), 0, s, gb.dev, (double *)chi.p);
            }
            hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, (const double *)chi.p, in.ne, (double *)scal.p, slot);
        };
        bool terminate = false;
        for (; it < iterations && !terminate; it++) {
            // buildSystem: H, b and chi2 at the current estimates
            HIPCHK(hipMemsetAsync(b.p, 0, (size_t)nvec * 8, s));
            if ((rc = lin.build(s, gb, (double *)b.p, (double *)scal.p, err, errlen))) goto done;
            chi2_into(0, false);
            HIPCHK(hipMemcpyAsync(h_s, scal.p, 2 * 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            double currentChi = h_s[0];
            if (it == 0) { chi_first = currentChi; lambda = 1e-5 * h_s[1]; ni = 2; }
            chi_last = currentChi;
            double rho = 0;
            int qmax = 0;
            bool lambda_ok = true;
            do {
                poses(1);                                                        // push()
                HIPCHK(hipMemsetAsync(bad.p, 0, sizeof(int), s));
                if ((rc = lin.solve(s, gb, lambda, (const double *)b.p, (double *)sol.p, (int *)bad.p, err, errlen))) goto done;
                HIPCHK(hipMemcpyAsync(&h_bad, bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                const bool ok2 = h_bad == 0;
                if (!ok2) HIPCHK(hipMemsetAsync(sol.p, 0, (size_t)nvec * 8, s));   // x = 0: the trial is rejected below
                poses(0);                                                        // update(x)
                chi2_into(2, true);
                hipLaunchKernelGGL(lm_scale_kernel, dim3(1), dim3(256), 0, s, (const double *)sol.p, (const double *)b.p, n, lambda, (double *)scal.p, 3);
                HIPCHK(hipMemcpyAsync(h_s, scal.p, 4 * 8, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                double tempChi = ok2 ? h_s[2] : std::numeric_limits<double>::max();
                rho = (currentChi - tempChi) / (h_s[3] + 1e-3);
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
                    poses(2);                                                    // pop()
                    if (!std::isfinite(lambda)) { lambda_ok = false; break; }
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0 || !lambda_ok) terminate = true;
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    stats[0] = it; stats[1] = trials; stats[2] = chi_first; stats[3] = chi_last; stats[4] = lambda;
    if (seconds) *seconds = 1e-3 * ms;
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

// Dense linear algebra for LM: H and a working copy A in HBM, the blocked fp64-MFMA Cholesky, two blocked
// triangular solves (one launch per 64-block). n <= 32 k unknowns (2 x 8 GB).
struct DenseLM : LMLinear {
    int D, n, N, nt;
    DevBuf H, A, linv, linv_all, rhs;
    int init(int D_, int n_, char *err, size_t errlen) {
        D = D_; n = n_; N = round_up(std::max(n, 1)); nt = N / TB;
        if (hipMalloc(&H.p, (size_t)N * N * 8) != hipSuccess || hipMalloc(&A.p, (size_t)N * N * 8) != hipSuccess) {
            snprintf(err, errlen, "hipMalloc of two %d x %d matrices failed", N, N);
            return SPG_ENOMEM;
        }
        if (hipMalloc(&linv.p, TB * TB * 8) != hipSuccess || hipMalloc(&linv_all.p, (size_t)nt * TB * TB * 8) != hipSuccess ||
            hipMalloc(&rhs.p, (size_t)N * 8) != hipSuccess) return SPG_ENOMEM;
        return 0;
    }
    int build(hipStream_t s, const GraphBufs &gb, double *b, double *scal, char *err, size_t errlen) override {
        int rc = 0;
        HIPCHK(hipMemsetAsync(H.p, 0, (size_t)N * N * 8, s));
        if (D == 6) launch_assemble<6>(gb, (double *)H.p, N, s, b);
        else launch_assemble<3>(gb, (double *)H.p, N, s, b);
        if (N > n) hipLaunchKernelGGL(pad_identity_kernel, dim3((N - n + 255) / 256), dim3(256), 0, s, (double *)H.p, N, n, N);
        hipLaunchKernelGGL(max_diag_kernel, dim3(1), dim3(256), 0, s, (const double *)H.p, N, n, scal, 1);
    done:
        return rc;
    }
    int solve(hipStream_t s, const GraphBufs &, double lambda, const double *b, double *sol, int *bad, char *err, size_t errlen) override {
        int rc = 0;
        HIPCHK(hipMemcpyAsync(A.p, H.p, (size_t)N * N * 8, hipMemcpyDeviceToDevice, s));
        if (n > 0) hipLaunchKernelGGL(add_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (double *)A.p, N, n, lambda);
        potrf_lower((double *)A.p, N, (double *)linv.p, bad, s);
        hipLaunchKernelGGL(diag_potrf_kernel, dim3(nt), dim3(64), 0, s, (double *)A.p, N, (double *)linv_all.p, bad, 0);
        HIPCHK(hipMemcpyAsync(rhs.p, b, (size_t)N * 8, hipMemcpyDeviceToDevice, s));
        for (int j = 0; j < nt; j++)
            hipLaunchKernelGGL(trsv_forward_step, dim3(nt - j), dim3(64), 0, s, (const double *)A.p, N, (const double *)linv_all.p, (double *)rhs.p, sol, j);
        HIPCHK(hipMemcpyAsync(rhs.p, sol, (size_t)N * 8, hipMemcpyDeviceToDevice, s));
        for (int j = nt - 1; j >= 0; j--)
            hipLaunchKernelGGL(trsv_backward_step, dim3(j + 1), dim3(64), 0, s, (const double *)A.p, N, (const double *)linv_all.p, (double *)rhs.p, sol, j);
    done:
        return rc;
    }
};

}  // namespace

namespace spg {

int hip_dense_optimize(void *stream, const DenseGraphIn &in, int n, int iterations, double *stats, double *seconds,
                       char *err, size_t errlen) {
    hipStream_t s = (hipStream_t)stream;
    GraphBufs gb;
    DenseLM lin;
    int rc = lin.init(in.D, n, err, errlen);
    if (rc) return rc;
    if ((rc = stage_graph(in, gb, s))) { snprintf(err, errlen, "staging the graph for the optimiser failed (%d)", rc); return rc; }
    return lm_run(s, in, gb, n, lin.N, iterations, lin, stats, seconds, err, errlen);
}

}  // namespace spg

#include "spg_sparse.inc"
