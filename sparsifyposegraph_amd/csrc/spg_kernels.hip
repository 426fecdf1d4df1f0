// csrc/spg_kernels.hip — the per-blanket marginalisation kernel for gfx950 and the HIP backend
// that launches one conflict-free round of blankets.
//
// One workgroup per Markov blanket (64 lanes = one wavefront for blankets whose tiles fit in LDS,
// 256 lanes with an L2-resident workspace for the rare large ones). The whole per-vertex iteration
// of the reference's VertexRemover::remove (src/vertex_remover.cpp:108-132) runs inside the kernel:
//
//   gather     poses + edge records of the blanket straight from the HBM-resident arena
//   assemble   H = sum_e J_e^T Omega_e J_e                   (g2o buildSystem, src/vertex_remover.cpp:397-402)
//   Schur      Lambda_t = H_kk - H_mk^T LLT(H_mm)^-1 H_mk    (src/vertex_remover.cpp:444-449)
//   topology   pseudo-Chow-Liu: (Lambda_t + I)^-1, pairwise log-det weights, Kruskal
//                                                             (src/pseudo_chow_liu.cpp:33-87,169-196,253-289)
//   skeleton   z_ab = x_a^-1 x_b, Jacobians at zero error     (src/topology_provider_binary.hpp:40-47,
//                                                              src/vertex_remover.cpp:466-498)
//   recover    eig(Lambda_t), gauge drop, Sigma = U S U^T, X_e = (J_e Sigma J_e^T)^-1
//                                                             (src/logdet_function.cpp:14-64,236-279)
//   KLD        1/2 (tr(S M) - logdet M - logdet S - r), M = U^T (J^T X J) U
//                                                             (src/logdet_function.cpp:119-133,281-323)
//   scatter    new edge records + per-blanket output record back into the arena
//
// 3x3 / 6x6 information blocks, Jacobians and the n x n tiles (n = d*k) live in LDS; HBM traffic is
// the algorithmic minimum: every pose and edge record is read once, every new record written once.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../include/spg.h"
#include "spg_dev_geom.hpp"
#include "spg_dev_la.hpp"
#include "spg_dev_wave.hpp"
#include "spg_internal.h"

using namespace spgdev;

namespace {

constexpr int EC = 8;  // edges whose Jacobians are staged per chunk
// kernel-internal third value of the ALG template parameter: NFR with the blanket-level LM of the Local
// linearisation point compiled in (its pose-update arithmetic costs ~160 VGPRs; the plain NFR kernel has 84)
constexpr int SPG_ALG_NFR_LM = 2;

// LDS / workspace carve-up for one blanket (all offsets in doubles). Monotone in k and m, so the
// layout of the largest blanket of a launch bounds every blanket in it.
struct Layout {
    int n, nm, ld, ldm, P, NE;
    int o_pose, o_red, o_cs, o_ev, o_S, o_w, o_ldb, o_Lb, o_nJ, o_X, o_eJ, o_eO, o_eT, o_Ng, o_tre, o_xch, o_int, small_doubles;
    int o_M1, o_M2, o_M3, o_Hmm, o_Hmk, mat_doubles;
    // int area offsets (in ints, relative to o_int)
    int i_perm, i_keep, i_sorted, i_pij, i_comp, i_pairs, i_ev, i_misc, int_count;
    // GLC tail: CNT problems of size S (tree: k-1 of 2d; dense / k==1: one of n), see glc section
    int gS, gCNT, gld, gstride;
    int o_gmeas, o_gev, o_gcs, o_gscr;          // small (LDS)
    int i_gperm, i_gdone, i_gmeta, i_gverts;    // ints
    int o_G, o_gA;                              // mat space: 4 batch buffers; GLC-edge assembly scratch
};

__host__ __device__ inline Layout make_layout(int D, int nt, int k, int m, int alg = SPG_ALG_NFR, int topo = SPG_TOPO_TREE,
                                              int scratch = 0) {
    Layout L;
    const bool glc = (alg == SPG_ALG_GLC);
    const bool single = (topo == SPG_TOPO_DENSE) || k <= 1;
    int DD = D * D;
    L.n = D * k; L.nm = D * m;
    L.ld = L.n | 1; L.ldm = L.nm | 1;
    L.P = k * (k - 1) / 2;
    L.NE = k > 0 ? k : 1;  // most new edges any algorithm emits (GLC tree: root + k-1)
    int psz = (D == 6) ? 12 : 3;
    int o = 0;
    L.o_pose = o; o += (k + m) * psz;
    L.o_red = o; o += nt;
    L.o_cs = o; o += L.n + 4;
    L.o_ev = o; o += L.n;
    L.o_S = o; o += L.n;
    L.o_w = o; o += (L.P > 0 ? L.P : 1);
    L.o_ldb = o; o += k + 1;
    L.o_Lb = o; o += k * DD;
    L.o_nJ = o; o += L.NE * 2 * DD;
    L.o_X = o; o += L.NE * DD;
    L.o_eJ = o; o += EC * 2 * DD;
    L.o_eO = o; o += EC * DD;
    L.o_eT = o; o += EC * 2 * DD;
    L.o_Ng = o; o += L.n * D;
    L.o_tre = o; o += L.NE;
    L.o_xch = o; o += 4;
    L.gS = single ? (k > 0 ? D * k : D) : 2 * D;
    L.gCNT = single ? 1 : (k - 1);
    L.gld = L.gS | 1;
    L.gstride = L.gS * L.gld;
    L.o_gmeas = o; if (glc) o += L.gCNT * L.gS;
    L.o_gev = o; if (glc) o += L.gCNT * L.gS;
    L.o_gcs = o; if (glc) o += L.gCNT * (L.gS + 4);
    L.o_gscr = o; if (glc) o += L.gCNT * (2 * L.gS + 2);
    L.o_int = o;
    int io = 0;
    L.i_perm = io; io += L.n;
    L.i_keep = io; io += L.n;
    L.i_sorted = io; io += (L.P > 0 ? L.P : 1);
    L.i_pij = io; io += 2 * (L.P > 0 ? L.P : 1);
    L.i_comp = io; io += k + 1;
    L.i_pairs = io; io += 2 * L.NE;
    L.i_ev = io; io += 2 * EC;
    L.i_misc = io; io += 12;
    L.i_gperm = io; if (glc) io += L.gCNT * L.gS;
    L.i_gdone = io; if (glc) io += L.gCNT;
    L.i_gmeta = io; if (glc) io += 3 * (L.NE + 1);
    L.i_gverts = io; if (glc) io += 2 * L.NE + k + 2;
    L.int_count = io;
    o += (io + 1) / 2;
    L.small_doubles = o;
    int mo = 0;
    L.o_M1 = mo; mo += L.n * L.ld;
    L.o_M2 = mo; mo += L.n * L.ld;
    L.o_M3 = mo; mo += L.n * L.ld;
    L.o_Hmm = mo; mo += L.nm * L.ldm;
    L.o_Hmk = mo; mo += L.nm * L.ld;
    L.o_G = mo; if (glc) mo += 4 * L.gCNT * L.gstride;
    L.o_gA = mo; if (glc) mo += scratch;
    L.mat_doubles = mo;
    return L;
}

struct KArgs {
    double *arena;
    const spg_blanket_desc *blk;
    const int64_t *vpo;
    const spg_edge_ref *er;
    const int32_t *ev;
    const int32_t *list;
    double *gws;
    int64_t gws_stride;
    int topology, algorithm, flags, lin_point, tag;
    double chord_ratio;
    double *mail;        // pinned host mailbox for out records (or nullptr)
    int64_t mail_base;   // arena offset that maps to mail[0]
};

// A value every lane holds identically: pin it to scalar registers. The descriptor fields steer every loop bound and
// tile offset of a blanket; where they arrive through LDS or a stack slot (persistent worker) the compiler would
// otherwise carry them — and everything derived from them — in vector registers.
__device__ __forceinline__ int uni32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffu)), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
template <class T>
__device__ __forceinline__ T *uniptr(T *p) { return reinterpret_cast<T *>(uni64(reinterpret_cast<unsigned long long>(p))); }

// upper-triangular (row-wise) index of (r,c), r <= c
__device__ __forceinline__ int utri(int r, int c, int D) { return r * D - (r * (r - 1)) / 2 + (c - r); }

// One blanket on one workgroup of NT lanes. bd = its descriptor; bvpo[0..n_vert) the arena offsets of its poses,
// ber[0..n_edge) its edge references (vbegin indexes bev[]); gws_slot = index of the workgroup's global workspace.
// Called by blanket_kernel (one launch per batch, descriptors in global memory) and by blanket_worker (persistent
// workgroups fed through a queue, descriptors staged in LDS).
// Edge records written by OTHER workgroups of the same kernel (persistent worker: the new edges of earlier blankets)
// are read with agent-scope loads: each XCD has its own L2, and a line that holds a neighbouring, earlier record may
// already sit there stale. Launched kernels read them plainly (a kernel boundary lies between writer and reader).
template <bool COH>
__device__ __forceinline__ double ldrec(const double *p) {
    if (COH) return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}

// ... and written with system-scope (write-through) stores there: the ready word may then follow after a plain
// s_waitcnt instead of a system-scope release fence, which writes back every dirty line of the XCD's L2 (measured:
// publish() 11.7k cycles of a 90k-cycle blanket with the fence).
template <bool COH>
__device__ __forceinline__ void strec(double *p, double v) {
    if (COH) __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *p = v;
}
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int D, int NT, bool GWS, int ALG, bool COH = false>
__device__ __forceinline__ void blanket_body(const KArgs &a_in, const spg_blanket_desc &bd_in, const int64_t *bvpo_in, const spg_edge_ref *ber_in,
                                             const int32_t *bev_in, const int gws_slot, double *smem) {
    KArgs a;
    a.arena = uniptr(a_in.arena); a.gws = uniptr(a_in.gws); a.mail = uniptr(a_in.mail);
    a.gws_stride = (int64_t)uni64((unsigned long long)a_in.gws_stride); a.mail_base = (int64_t)uni64((unsigned long long)a_in.mail_base);
    a.topology = uni32(a_in.topology); a.algorithm = uni32(a_in.algorithm); a.flags = uni32(a_in.flags);
    a.lin_point = uni32(a_in.lin_point); a.tag = uni32(a_in.tag);
    a.chord_ratio = __longlong_as_double((long long)uni64((unsigned long long)__double_as_longlong(a_in.chord_ratio)));
    a.blk = nullptr; a.vpo = nullptr; a.er = nullptr; a.ev = nullptr; a.list = nullptr;
    spg_blanket_desc bd;
    bd.vert_begin = 0; bd.edge_begin = 0; bd.new_len = 0;
    bd.n_vert = uni32(bd_in.n_vert); bd.n_remove = uni32(bd_in.n_remove); bd.n_edge = uni32(bd_in.n_edge);
    bd.n_new_max = uni32(bd_in.n_new_max); bd.n_new_vert_max = uni32(bd_in.n_new_vert_max); bd.pad_ = uni32(bd_in.pad_);
    bd.new_off = (int64_t)uni64((unsigned long long)bd_in.new_off); bd.out_off = (int64_t)uni64((unsigned long long)bd_in.out_off);
    bd.tinfo_off = (int64_t)uni64((unsigned long long)bd_in.tinfo_off);
    const int64_t *bvpo = uniptr(bvpo_in);
    const spg_edge_ref *ber = uniptr(ber_in);
    const int32_t *bev = uniptr(bev_in);
    constexpr int DD = D * D;
    constexpr int PS = (D == 6) ? 7 : 3;    // pose / measurement doubles in the arena
    constexpr int PSZ = (D == 6) ? 12 : 3;  // pose doubles in LDS
    constexpr int REC = PS + D * (D + 1) / 2;
    const int tid = threadIdx.x;
    const int nv = bd.n_vert, m = bd.n_remove, k = nv - m;
    const Layout L = make_layout(D, NT, k, m, ALG, a.topology, bd.pad_);
    const int n = L.n, nm = L.nm, ld = L.ld, ldm = L.ldm;
    double *mat = GWS ? (a.gws + (size_t)gws_slot * (size_t)a.gws_stride) : (smem + L.small_doubles);
    double *pose = smem + L.o_pose, *cs = smem + L.o_cs, *ev = smem + L.o_ev, *Sv = smem + L.o_S;
    double *w = smem + L.o_w, *ldb = smem + L.o_ldb, *Lb = smem + L.o_Lb, *nJ = smem + L.o_nJ, *X = smem + L.o_X;
    double *eJ = smem + L.o_eJ, *eO = smem + L.o_eO, *eT = smem + L.o_eT;
    int *ints = reinterpret_cast<int *>(smem + L.o_int);
    int *perm = ints + L.i_perm, *keep = ints + L.i_keep, *sorted = ints + L.i_sorted, *comp = ints + L.i_comp;
    int *pairs = ints + L.i_pairs, *echv = ints + L.i_ev, *misc = ints + L.i_misc, *pij = ints + L.i_pij;
    double *M1 = mat + L.o_M1, *M2 = mat + L.o_M2, *M3 = mat + L.o_M3, *Hmm = mat + L.o_Hmm, *Hmk = mat + L.o_Hmk;
    Team<NT> T{tid, smem + L.o_red, misc + 0};
    // register-resident single-wavefront SPD kernels (spg_dev_wave.hpp) when the tile fits
    const bool use_wave_hw = !GWS && (n <= kWaveMax) && !((a.flags >> 17) & 1);
    const bool use_wave = use_wave_hw && (NT == 64);
    // two wavefronts per blanket: the Chow-Liu chain and the gauge chain run side by side (NFR only)
    const bool split = (NT == 128) && (ALG != SPG_ALG_GLC) && use_wave_hw && !(a.flags & SPG_FLAG_FORCE_EIG);
    double *xch = smem + L.o_xch;
    double *arena = a.arena;
    double *orec = a.mail ? (a.mail + (bd.out_off - a.mail_base)) : (arena + bd.out_off);

    int status = SPG_OK, info = 0, n_new = 0;
    double kld = __builtin_nan(""), min_gap = __builtin_inf();

    // publish(): everything the host's graph update needs — status so far, n_new, the new-edge table —
    // followed by a system-scope release and the ready tag. finish() = publish (unless done) + KLD.
    bool published = false;
    // Worker (COH): the new edge records and the out record are staged in LDS (recbuf / orl) and leave as a few
    // coalesced write-through stores — one word per lane — instead of ~110 single stores by the lanes that compute them:
    // a system-scope store is acknowledged over the fabric (PCIe for the mailbox), and a lane that issues them one by one
    // pays that round trip per store.
    double *recbuf = smem + L.o_eJ;     // ne x REC doubles (eJ is free after the assembly)
    double *orl = smem + L.o_eO;        // the out record (<= 6 + 4 k + 2 k doubles; eO is free after the assembly)
    auto publish = [&]() {
        if (COH) {
            T.sync();   // recbuf is complete
            const int nrec = n_new * REC;
            for (int it = tid; it < nrec; it += NT) strec<true>(arena + bd.new_off + it, recbuf[it]);
            // compact out record (streaming driver, flags bit 20): ONE cache line — the six header words, then the new edges'
            // endpoint pairs as local vertex indices, 4 bits each, edge e in byte e of words [6] (e < 8) and [7]; the edge
            // table is implied (pose-pose records of REC doubles, back to back at new_off). The host polls, reads and later
            // re-reads (KLD, final word) a single line per blanket instead of five.
            const bool compact = (a.flags >> 20) & 1;
            if (tid == 0) {
                orl[0] = (double)status; orl[1] = (double)info; orl[2] = kld; orl[3] = min_gap; orl[4] = (double)n_new; orl[5] = 0.0;
                if (compact) {
                    unsigned long long w0 = 0, w1 = 0;
                    for (int e = 0; e < n_new; e++) {
                        const unsigned long long pr = (unsigned long long)((m + pairs[2 * e]) & 15) | ((unsigned long long)((m + pairs[2 * e + 1]) & 15) << 4);
                        if (e < 8) w0 |= pr << (8 * e); else w1 |= pr << (8 * (e - 8));
                    }
                    orl[6] = __longlong_as_double((long long)w0); orl[7] = __longlong_as_double((long long)w1);
                } else
                for (int e = 0; e < n_new; e++) {
                    orl[SPG_OUT_HDR + 4 * e + 0] = (double)SPG_EDGE_BINARY;
                    orl[SPG_OUT_HDR + 4 * e + 1] = (double)(e * REC);
                    orl[SPG_OUT_HDR + 4 * e + 2] = (double)REC;
                    orl[SPG_OUT_HDR + 4 * e + 3] = 2.0;
                    orl[SPG_OUT_HDR + 4 * bd.n_new_max + 2 * e + 0] = (double)(m + pairs[2 * e]);
                    orl[SPG_OUT_HDR + 4 * bd.n_new_max + 2 * e + 1] = (double)(m + pairs[2 * e + 1]);
                }
            }
            wait_stores();   // this lane's record stores have been acknowledged
            T.sync();        // ... everybody's; orl is complete
            const int olen = compact ? 8 : SPG_OUT_HDR + 4 * bd.n_new_max + 2 * n_new;
            if (tid < 64) {   // (wave 0; olen <= 42 for the blankets a worker takes)
                for (int t = tid; t < olen; t += 64) if (t != 5) strec<true>(orec + t, orl[t]);
                wait_stores();   // wave 0's stores are out before its lane 0 raises the ready word
                if (tid == 0) {
                    __hip_atomic_store(&orec[5], SPG_READY_WORD(a.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    orl[5] = (double)wall_clock64();   // (SPG_WORKER_STAMP diagnostics: when the ready word left)
                }
            }
            published = true;
            return;
        }
        T.sync();  // every lane's new-record stores precede the release below
        if (tid == 0) {
            orec[0] = (double)status; orec[1] = (double)info; orec[2] = kld; orec[3] = min_gap; orec[4] = (double)n_new;
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
        T.sync();
        if (tid == 0) {
            __hip_atomic_store(&orec[5], SPG_READY_WORD(a.tag), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        published = true;
    };
    auto finish = [&]() {
        if (!published) publish();
        else if (tid == 0) { strec<COH>(orec + 2, kld); strec<COH>(orec + 0, (double)status); }
        // the record is complete (KLD and a possible SPG_ST_KLD_NOT_PD included): final word, after a release
        if (tid == 0) {
            if (COH) { wait_stores(); __hip_atomic_store(&orec[5], SPG_FINAL_WORD(a.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            else { __threadfence_system(); __hip_atomic_store(&orec[5], SPG_FINAL_WORD(a.tag), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    };
    // diagnostic cycle stamps (flags bit 16, needs tinfo_off >= 0): written only to the debug region
    const bool stamping = ((a.flags >> 16) & 1) && bd.tinfo_off >= 0;
    auto STAMP = [&](int idx) {
        if (stamping) {
            T.sync();
            if (tid == 0) arena[bd.tinfo_off + idx] = (double)__builtin_amdgcn_s_memtime();
        }
    };
    // inside the two concurrent chains of the two-wavefront variant: no workgroup barrier, lane 0 of each wave stamps
    auto CSTAMP = [&](int idx) {
        if (stamping) {
            if (!split) { T.sync(); if (tid == 0) arena[bd.tinfo_off + idx] = (double)__builtin_amdgcn_s_memtime(); }
            else if ((tid & 63) == 0) arena[bd.tinfo_off + idx] = (double)__builtin_amdgcn_s_memtime();
        }
    };
    STAMP(0);  //
    // ---------------------------------------------------------------- gather poses, clear H
    if (tid == 0) { misc[0] = 0; misc[1] = 0; misc[2] = 0; misc[6] = 0; misc[7] = 0; misc[8] = 0; misc[9] = 0; }
    for (int v = tid; v < nv; v += NT) {
        const double *p = arena + bvpo[v];
        if (D == 6) iso_from_tq(p, pose + v * PSZ);
        else { pose[v * PSZ] = p[0]; pose[v * PSZ + 1] = p[1]; pose[v * PSZ + 2] = p[2]; }
    }
    for (int i = tid; i < n * ld; i += NT) M1[i] = 0.0;
    for (int i = tid; i < nm * ldm; i += NT) Hmm[i] = 0.0;
    for (int i = tid; i < nm * ld; i += NT) Hmk[i] = 0.0;
    T.sync();
    STAMP(23);  // gathered + cleared
    if (bd.n_edge == 0 || m < 1) { status = SPG_ST_EMPTY_BLANKET; finish(); return; }
    constexpr bool is_glc = (ALG == SPG_ALG_GLC);  // compile-time: the NFR instantiation carries no GLC code
    constexpr bool has_lm = (ALG == SPG_ALG_NFR_LM);
    if (is_glc && !(a.topology == SPG_TOPO_DENSE || a.topology == SPG_TOPO_TREE)) {
        status = SPG_ST_UNSUPPORTED;  // asserts at src/topology_provider_glc.cpp:107-111
        finish(); return;
    }

    // Local linearisation point without a closed form: lm_left LM iterations on the blanket (below)
    int lm_left = 0;
    // (a cluster with fewer than two kept vertices emits no edge whatever the linearisation point: it is not re-linearised;
    //  clusters with k >= 2 under Local go to the generic kernel behind the pre-pass of hip_run_round)
    if (a.lin_point != SPG_LIN_GLOBAL && !(m > 1 && k < 2)) {
        // Local linearisation point, closed-form branch of buildSubgraph (src/vertex_remover.cpp:304-381):
        // possible iff every vertex but the first removed one sits in exactly one (pose-pose) blanket edge;
        // then the removed vertex goes to the origin and each neighbour to z (or z^-1) of its edge.
        if (is_glc) { status = SPG_ST_UNSUPPORTED; finish(); return; }  // src/topology_provider_glc.cpp:110-111
        // the per-vertex edge counters below live in perm[] (n ints): a cluster with many removed and few kept vertices
        // (n < nv) would write past it, and clusters under a Local point are not built anyway — leave before any write
        if (n < nv || m != 1) { status = SPG_ST_NEEDS_LOCAL_OPTIMIZATION; finish(); return; }
        int *cntv = perm;
        for (int v = tid; v < nv; v += NT) cntv[v] = 0;
        T.sync();
        for (int e = tid; e < bd.n_edge; e += NT) {
            const spg_edge_ref er = ber[e];
            if (er.kind != SPG_EDGE_BINARY) { misc[8] = 1; continue; }
            int vi = bev[er.vbegin], vj = bev[er.vbegin + 1];
            if (vi != 0) atomicAdd(&cntv[vi], 1);
            if (vj != 0) atomicAdd(&cntv[vj], 1);
        }
        T.sync();
        for (int v = 1 + tid; v < nv; v += NT) if (cntv[v] > 1) misc[9] = 1;
        T.sync();
        // a GLC edge cannot propagate an estimate, and clusters (m > 1) with a Local point are not built
        if (misc[8] || n < nv || (misc[9] && m != 1)) { status = SPG_ST_NEEDS_LOCAL_OPTIMIZATION; finish(); return; }
        if (misc[9]) {
            // some kept vertex sits in several blanket edges: the reference fixes the removed vertex at its
            // current estimate and runs 10 LM iterations on the subgraph (src/vertex_remover.cpp:382-391)
            if constexpr (has_lm) lm_left = 10;
            else { status = SPG_ST_NEEDS_LOCAL_OPTIMIZATION; finish(); return; }   // (the host launches the LM variant for Local)
        } else {
        if (tid == 0) {
            if (D == 6) {
#pragma unroll
                for (int t_ = 0; t_ < kIso; t_++) pose[t_] = (t_ == 0 || t_ == 4 || t_ == 8) ? 1.0 : 0.0;
            }
            else { pose[0] = 0; pose[1] = 0; pose[2] = 0; }
        }
        for (int e = tid; e < bd.n_edge; e += NT) {
            const spg_edge_ref er = ber[e];
            int vi = bev[er.vbegin], vj = bev[er.vbegin + 1];
            const double *rec = arena + er.off;
            if (vi == 0 && vj == 0) continue;
            if (D == 6) {
                double Z[kIso];
                iso_from_tq(rec, Z);
                if (vi == 0) {
#pragma unroll
                    for (int t_ = 0; t_ < kIso; t_++) pose[vj * PSZ + t_] = Z[t_];
                }
                else {
                    double I12[kIso] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, Zi[kIso];
                    iso_inv_mul(Z, I12, Zi);
#pragma unroll
                    for (int t_ = 0; t_ < kIso; t_++) pose[vi * PSZ + t_] = Zi[t_];
                }
            } else {
                if (vi == 0) { pose[vj * PSZ] = rec[0]; pose[vj * PSZ + 1] = rec[1]; pose[vj * PSZ + 2] = normalize_theta(rec[2]); }
                else {
                    double c = cos(rec[2]), sn = sin(rec[2]);
                    pose[vi * PSZ] = -(c * rec[0] + sn * rec[1]);
                    pose[vi * PSZ + 1] = -(-sn * rec[0] + c * rec[1]);
                    pose[vi * PSZ + 2] = normalize_theta(-rec[2]);
                }
            }
        }
        }
        T.sync();
    }
    // ---------------------------------------------------------------- assemble H (a6)
    auto hadd = [&](int R, int Cc, double val) {
        if (R < nm) {
            if (Cc < nm) Hmm[R * ldm + Cc] += val;
            else Hmk[R * ld + (Cc - nm)] += val;
        } else if (Cc >= nm) {
            M1[(R - nm) * ld + (Cc - nm)] += val;
        }
    };
    // LM state (uniform over the workgroup; only used when lm_left > 0). The system of an LM iteration is
    // what the assembly produces anyway: H = M1 (kept block: the removed vertex is fixed), plus the
    // right-hand side b = -sum J^T Omega e and chi2, which the assembly adds in LM mode. M2 / Sv keep the
    // system of the last accepted estimates, M3 takes the factorisation, Ng the backup of the estimates.
    double *eE = nJ;                      // errors of the staged edges (EC x D)
    double *bnew = ev, *bcur = Sv, *xs = cs, *pbak = smem + L.o_Ng;
    double lm_lambda = 0, lm_ni = 2, lm_chi = 0, lm_scale = 0;
    int lm_it = 0, lm_q = 0, lm_phase = 0;
    bool lm_redo = false;                 // the tiles hold a rejected trial: assemble once more, then go on
    for (;;) {
    if (has_lm && (lm_left > 0 || lm_redo)) {
        if (lm_phase > 0 || lm_redo) {
            for (int i = tid; i < n * ld; i += NT) M1[i] = 0.0;
            for (int i = tid; i < nm * ldm; i += NT) Hmm[i] = 0.0;
            for (int i = tid; i < nm * ld; i += NT) Hmk[i] = 0.0;
        }
        for (int i = tid; i < n; i += NT) bnew[i] = 0.0;
        if (tid == 0) xch[0] = 0.0;
        T.sync();
    }
    for (int base = 0; base < bd.n_edge; base += EC) {
        int cnt = min(EC, bd.n_edge - base);
        if (tid < cnt) {
            const spg_edge_ref er = ber[base + tid];
            int vi = 0, vj = 0;
            if (er.kind == SPG_EDGE_BINARY) {
                vi = bev[er.vbegin]; vj = bev[er.vbegin + 1];
                double rec[PS];
#pragma unroll
                for (int i = 0; i < PS; i++) rec[i] = ldrec<COH>(arena + er.off + i);
                if (D == 6) {
                    double Z[kIso];
                    iso_from_tq(rec, Z);
                    se3_edge_jac(pose + vi * PSZ, pose + vj * PSZ, Z, eJ + tid * 2 * DD, eJ + tid * 2 * DD + DD, (has_lm && lm_left > 0) ? eE + tid * D : nullptr);
                } else {
                    se2_edge_jac(pose + vi * PSZ, pose + vj * PSZ, rec, eJ + tid * 2 * DD, eJ + tid * 2 * DD + DD, (has_lm && lm_left > 0) ? eE + tid * D : nullptr);
                }
            } else {
                if (!is_glc) misc[1] = 1;  // GLC edge inside an NFR blanket: no provider applies
                vi = -1;                    // handled by the n-ary loop below
            }
            echv[2 * tid] = vi; echv[2 * tid + 1] = vj;
        }
        for (int it = tid; it < cnt * DD; it += NT) {
            int e = it / DD, rc = it - e * DD, r = rc / D, c = rc - r * D;
            const spg_edge_ref er = ber[base + e];
            int lo = r < c ? r : c, hi = r < c ? c : r;
            eO[it] = (er.kind == SPG_EDGE_BINARY) ? ldrec<COH>(arena + er.off + PS + utri(lo, hi, D)) : 0.0;
        }
        T.sync();
        STAMP(24);  // jacobians + omega staged
        // T_e = Omega_e [Ji | Jj] for every staged edge at once
        for (int it = tid; it < cnt * 2 * DD; it += NT) {
            int e = it / (2 * DD), rem = it - e * 2 * DD, wch = rem / DD, rc = rem - wch * DD, r = rc / D, c = rc - r * D;
            const double *J = eJ + e * 2 * DD + wch * DD, *Om = eO + e * DD;
            double s = 0;
#pragma unroll
            for (int p = 0; p < D; p++) s += Om[r * D + p] * J[p * D + c];
            eT[it] = s;
        }
        T.sync();
        STAMP(25);  // T = Omega J
        if (has_lm && lm_left > 0) {
            // LM mode: b_v -= (Omega J_v)^T e and chi2 += e^T Omega e. Lanes own (edge, side, row); the edges
            // of the chunk are folded in serially so that the sums keep a fixed order.
            {
                double c2 = 0;
                for (int it = tid; it < cnt * D; it += NT) {
                    int e = it / D, p = it - e * D;
                    if (echv[2 * e] < 0) continue;
                    const double *Om = eO + e * DD, *er_ = eE + e * D;
                    double sacc = 0;
#pragma unroll
                    for (int q = 0; q < D; q++) sacc += Om[p * D + q] * er_[q];
                    c2 += er_[p] * sacc;
                }
                c2 = T.sum(c2);
                if (tid == 0) xch[0] += c2;
                for (int e = 0; e < cnt; e++) {
                    if (echv[2 * e] < 0) continue;   // (uniform: echv is shared)
                    for (int side = 0; side < 2; side++) {
                        const int v = echv[2 * e + side];
                        if (tid < D && v >= 1) {     // the removed vertex (local 0, m == 1) is fixed
                            const double *Te = eT + e * 2 * DD + side * DD, *er_ = eE + e * D;
                            double sacc = 0;
#pragma unroll
                            for (int p = 0; p < D; p++) sacc += Te[p * D + tid] * er_[p];
                            bnew[(v - 1) * D + tid] -= sacc;
                        }
                        T.sync();
                    }
                }
            }
            T.sync();
        }
        if (NT == 64) {
            // One wavefront: LDS atomics retire in program order and every ds_add_f64 below carries the
            // items of a single edge (128-slot stride, 108 used), so no two lanes of one instruction hit
            // the same address and contributions land in ascending edge order — deterministic, without a
            // barrier per edge.
            auto hadd_atomic = [&](int R, int Cc, double val) {
                double *dst = nullptr;
                if (R < nm) dst = (Cc < nm) ? &Hmm[R * ldm + Cc] : &Hmk[R * ld + (Cc - nm)];
                else if (Cc >= nm) dst = &M1[(R - nm) * ld + (Cc - nm)];
                if (dst) __hip_atomic_fetch_add(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            constexpr int SLOT = (3 * DD + 63) / 64 * 64;
            for (int it = tid; it < cnt * SLOT; it += NT) {
                int e = it / SLOT, loc = it - e * SLOT;
                if (loc >= 3 * DD || echv[2 * e] < 0) continue;
                const double *Ji = eJ + e * 2 * DD, *Jj = Ji + DD, *Te = eT + e * 2 * DD;
                int vi = echv[2 * e], vj = echv[2 * e + 1];
                int blk = loc / DD, rc = loc - blk * DD, r = rc / D, c = rc - r * D;
                const double *Ja = (blk == 2) ? Jj : Ji;
                const double *Tb = (blk == 0) ? Te : Te + DD;
                double s = 0;
#pragma unroll
                for (int p = 0; p < D; p++) s += Ja[p * D + r] * Tb[p * D + c];
                if (blk == 0) hadd_atomic(vi * D + r, vi * D + c, s);
                else if (blk == 2) hadd_atomic(vj * D + r, vj * D + c, s);
                else if (vi != vj) {
                    hadd_atomic(vi * D + r, vj * D + c, s);
                    hadd_atomic(vj * D + c, vi * D + r, s);
                }
            }
            T.sync();
        } else {
        // accumulate edge by edge (blocks of different edges overlap on the shared vertices)
        for (int e = 0; e < cnt; e++) {
            if (echv[2 * e] < 0) continue;
            const double *Ji = eJ + e * 2 * DD, *Jj = Ji + DD, *Te = eT + e * 2 * DD;
            int vi = echv[2 * e], vj = echv[2 * e + 1];
            for (int it = tid; it < 3 * DD; it += NT) {
                int blk = it / DD, rc = it - blk * DD, r = rc / D, c = rc - r * D;
                const double *Ja = (blk == 2) ? Jj : Ji;
                const double *Tb = (blk == 0) ? Te : Te + DD;
                double s = 0;
#pragma unroll
                for (int p = 0; p < D; p++) s += Ja[p * D + r] * Tb[p * D + c];
                if (blk == 0) hadd(vi * D + r, vi * D + c, s);
                else if (blk == 2) hadd(vj * D + r, vj * D + c, s);
                else if (vi != vj) { hadd(vi * D + r, vj * D + c, s); hadd(vj * D + c, vi * D + r, s); }
            }
            T.sync();
        }
        }
    }
    if (!has_lm || lm_redo || lm_left <= 0) break;   // ordinary case: one assembly
    if constexpr (has_lm)
    // ---- g2o Levenberg-Marquardt on the blanket, removed vertex fixed (OptimizationAlgorithmLevenberg:
    //      lambda_0 = 1e-5 max diag, <= 10 trials per iteration, rho = (chi2 - chi2') / (x.(lambda x + b) + 1e-3),
    //      good step: lambda *= clamp(1 - (2 rho - 1)^3, 1/3, 2/3); bad step: lambda *= ni, ni *= 2)
    {
        const double chi_new = xch[0];
        bool next_trial = false, done = false;
        if (lm_phase == 0) {
            // first system: the current estimates
            lm_chi = chi_new;
            if (tid == 0) { double md = 0; for (int i = 0; i < n; i++) md = fmax(md, fabs(M1[i * ld + i])); xch[1] = md; }
            T.sync();
            lm_lambda = 1e-5 * xch[1];
            lm_ni = 2; lm_it = 0; lm_q = 0;
            for (int i = tid; i < n * ld; i += NT) M2[i] = M1[i];
            for (int i = tid; i < n; i += NT) bcur[i] = bnew[i];
            T.sync();
            lm_phase = 1;
            next_trial = true;
        } else {
            const double rho = (lm_chi - chi_new) / (lm_scale + 1e-3);
            if (rho > 0 && isfinite(chi_new)) {
                double alpha = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                alpha = fmin(alpha, 2.0 / 3.0);
                lm_lambda *= fmax(1.0 / 3.0, alpha);
                lm_ni = 2;
                lm_chi = chi_new;
                for (int i = tid; i < n * ld; i += NT) M2[i] = M1[i];
                for (int i = tid; i < n; i += NT) bcur[i] = bnew[i];
                T.sync();
                lm_it++; lm_q = 0;
                if (lm_it >= lm_left) done = true;          // the tiles hold the system of the accepted estimates
                else next_trial = true;
            } else {
                lm_lambda *= lm_ni;
                lm_ni *= 2;
                for (int i = tid; i < nv * PSZ; i += NT) pose[i] = pbak[i];   // pop()
                T.sync();
                lm_q++;
                if (rho < 0 && lm_q < 10 && isfinite(lm_lambda)) next_trial = true;
                else { lm_redo = true; }                     // terminate: re-assemble at the restored estimates
            }
        }
        if (done) { lm_left = 0; break; }
        if (lm_redo) { lm_left = 0; continue; }
        if (next_trial) {
            // solve (H + lambda I) x = b with the system of the accepted estimates, then x -> estimates
            bool ok2 = true;
            for (;;) {
                for (int i = tid; i < nv * PSZ; i += NT) pbak[i] = pose[i];   // push()
                if (tid == 0) *T.flag = 0;
                if (use_wave_hw) {
                    // small system: (H + lambda I)^-1 by the register-resident Gauss-Jordan of one wavefront
                    T.sync();
                    if (tid < 64) {
                        double ldt_, tri_;
                        bool okw = wave_spd_inverse(M2, ld, n, tid, lm_lambda, M3, ldt_, tri_);
                        if (!okw && tid == 0) *T.flag = 1;
                    }
                    T.sync();
                } else {
                    for (int i = tid; i < n * ld; i += NT) { int r = i / ld, c = i - r * ld; M3[i] = M2[i] + ((r == c && c < n) ? lm_lambda : 0.0); }
                    T.sync();
                    chol_lower<NT>(T, M3, n, ld);
                }
                ok2 = (*T.flag == 0);
                T.sync();
                if (tid == 0) *T.flag = 0;
                if (ok2) break;
                // the factorisation failed: g2o rejects the step (chi2 = max) without looking at the estimates
                lm_lambda *= lm_ni; lm_ni *= 2; lm_q++;
                if (!(lm_q < 10 && isfinite(lm_lambda))) break;
                T.sync();
            }
            if (!ok2) { lm_redo = true; lm_left = 0; T.sync(); continue; }
            if (use_wave_hw) {
                for (int i = tid; i < n; i += NT) {
                    double sacc = 0;
                    for (int j = 0; j < n; j++) sacc += M3[i * ld + j] * bcur[j];
                    xs[i] = sacc;
                }
                T.sync();
                double sc = 0;
                for (int i = tid; i < n; i += NT) sc += xs[i] * (lm_lambda * xs[i] + bcur[i]);
                sc = T.sum(sc);
                if (tid == 0) xch[2] = sc;
            } else if (tid == 0) {
                for (int i = 0; i < n; i++) {
                    double sacc = bcur[i];
                    for (int kk = 0; kk < i; kk++) sacc -= M3[i * ld + kk] * xs[kk];
                    xs[i] = sacc / M3[i * ld + i];
                }
                for (int i = n - 1; i >= 0; i--) {
                    double sacc = xs[i];
                    for (int kk = i + 1; kk < n; kk++) sacc -= M3[kk * ld + i] * xs[kk];
                    xs[i] = sacc / M3[i * ld + i];
                }
                double sc = 0;
                for (int i = 0; i < n; i++) sc += xs[i] * (lm_lambda * xs[i] + bcur[i]);
                xch[2] = sc;
            }
            T.sync();
            lm_scale = xch[2];
            for (int v = 1 + tid; v < nv; v += NT) {
                const double *dx = xs + (v - 1) * D;
                double *pv = pose + v * PSZ;
                if (D == 6) {
                    // X <- X * fromVectorMQT(dx), renormalised through the quaternion (the estimates are stored as t + q)
                    double Dl[kIso], R[9], q[4];
                    iso_from_mqt(dx, Dl);
#pragma unroll
                    for (int i = 0; i < 3; i++) {
#pragma unroll
                        for (int c = 0; c < 3; c++) R[i * 3 + c] = pv[i * 3] * Dl[c] + pv[i * 3 + 1] * Dl[3 + c] + pv[i * 3 + 2] * Dl[6 + c];
                    }
                    double tn[3];
#pragma unroll
                    for (int i = 0; i < 3; i++) tn[i] = pv[9 + i] + pv[i * 3] * Dl[9] + pv[i * 3 + 1] * Dl[10] + pv[i * 3 + 2] * Dl[11];
                    R_to_quat(R, q);
                    quat_to_R(q, pv);
                    pv[9] = tn[0]; pv[10] = tn[1]; pv[11] = tn[2];
                } else {
                    pv[0] += dx[0]; pv[1] += dx[1]; pv[2] = normalize_theta(pv[2] + dx[2]);
                }
            }
            T.sync();
        }
    }
    }   // LM wrapper
    if (misc[1]) { status = SPG_ST_UNSUPPORTED; finish(); return; }
    if constexpr (is_glc) {
        // n-ary GLC edges already in the blanket (a14): H += (W Jr)^T (W Jr), Jr = reparametrisation
        // Jacobian at the current estimates (src/glc_edge.cpp:40-49, src/glc_reparam_binary.hpp:78-127)
        double *gA = mat + L.o_gA;
        for (int e = 0; e < bd.n_edge; e++) {
            const spg_edge_ref er = ber[e];
            if (er.kind != SPG_EDGE_GLC) continue;
            const int q = er.nv, dq = D * q, rr_ = (er.len - dq) / dq;
            const double *rec = arena + er.off;   // meas (dq) then W (rr_ x dq)
            double *Jb = gA;                      // q x (Ji0 | Jii), 2*DD each
            double *Aw = gA + q * 2 * DD;         // rr_ x dq
            for (int i = tid; i < q; i += NT) {
                int v0 = bev[er.vbegin], vi = bev[er.vbegin + i];
                if (D == 6) {
                    double Z[kIso], Xz[kIso] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
                    iso_from_mqt(rec + 6 * i, Z);
                    if (i == 0) se3_edge_jac(Xz, pose + v0 * PSZ, Z, Jb, Jb + DD, nullptr);
                    else se3_edge_jac(pose + v0 * PSZ, pose + vi * PSZ, Z, Jb + i * 2 * DD, Jb + i * 2 * DD + DD, nullptr);
                } else {
                    double xz[3] = {0, 0, 0};
                    if (i == 0) se2_edge_jac(xz, pose + v0 * PSZ, rec, Jb, Jb + DD, nullptr);
                    else se2_edge_jac(pose + v0 * PSZ, pose + vi * PSZ, rec + 3 * i, Jb + i * 2 * DD, Jb + i * 2 * DD + DD, nullptr);
                }
            }
            T.sync();
            for (int it = tid; it < rr_ * dq; it += NT) {
                int row = it / dq, col = it - row * dq, blk = col / D, c = col - blk * D;
                const double *Wr = rec + dq + (int64_t)row * dq;
                double sacc = 0;
                if (blk == 0) {
                    for (int p = 0; p < D; p++) sacc += Wr[p] * Jb[DD + p * D + c];            // W_0 * J00 (Jj of the mock edge)
                    for (int i = 1; i < q; i++)
                        for (int p = 0; p < D; p++) sacc += Wr[i * D + p] * Jb[i * 2 * DD + p * D + c];  // W_i * Ji0
                } else {
                    for (int p = 0; p < D; p++) sacc += Wr[blk * D + p] * Jb[blk * 2 * DD + DD + p * D + c];  // W_i * Jii
                }
                Aw[it] = sacc;
            }
            T.sync();
            for (int it = tid; it < dq * dq; it += NT) {
                int R = it / dq, Cc = it - R * dq;
                double sacc = 0;
                for (int p = 0; p < rr_; p++) sacc += Aw[p * dq + R] * Aw[p * dq + Cc];
                int vR = bev[er.vbegin + R / D], vC = bev[er.vbegin + Cc / D];
                hadd(vR * D + (R % D), vC * D + (Cc % D), sacc);
            }
            T.sync();
        }
    }
    STAMP(1);  // assembled
    const int stop_after = (a.flags >> 8) & 0xff;  // diagnostic: truncate the pipeline (timing breakdowns only)
    if (stop_after == 1) { finish(); return; }

    // ---------------------------------------------------------------- Schur complement (a7)
    if (m == 1) {
        // H_mm is a single d x d block: one lane inverts it in registers (LLT(H_mm).solve, :444-447),
        // then Lambda_t = H_kk - H_mk^T (H_mm^-1 H_mk) as two small products
        if (tid == 0) {
            double Ab[DD], Xr[DD];
#pragma unroll
            for (int i = 0; i < D; i++)
#pragma unroll
                for (int j = 0; j < D; j++) Ab[i * D + j] = Hmm[i * ldm + j];
            if (!chol_reg<D>(Ab)) misc[0] = 1;
            chol_inverse_reg<D>(Ab, Xr);
#pragma unroll
            for (int i = 0; i < D; i++)
#pragma unroll
                for (int j = 0; j < D; j++) Hmm[i * ldm + j] = Xr[i * D + j];
        }
        T.sync();
        STAMP(26);  // Hmm inverse
        if (misc[0]) { status = SPG_ST_HMM_NOT_PD; finish(); return; }
        // Y = H_mm^-1 H_mk into M2 rows 0..D-1 (M2 is free until Chow-Liu)
        for (int it = tid; it < D * n; it += NT) {
            int r = it / n, c = it - r * n;
            double s = 0;
#pragma unroll
            for (int p = 0; p < D; p++) s += Hmm[r * ldm + p] * Hmk[p * ld + c];
            M2[r * ld + c] = s;
        }
        T.sync();
        STAMP(27);  // Y
    } else {
        chol_lower<NT>(T, Hmm, nm, ldm);
        if (misc[0]) { status = SPG_ST_HMM_NOT_PD; finish(); return; }
        tri_solve_lower<NT>(T, Hmm, nm, ldm, Hmk, n, ld);
    }
    {
        int sh = ceil_log2(n), tot = n << sh;
        double bad = 0;
        const double *Yl = (m == 1) ? M2 : Hmk;   // m == 1: Hmk^T (Hmm^-1 Hmk) ; else (L^-1 Hmk)^T (L^-1 Hmk)
        for (int it = tid; it < tot; it += NT) {
            int i = it >> sh, j = it & ((1 << sh) - 1);
            if (j < n) {
                double s = 0;
                if (m == 1) {
#pragma unroll
                    for (int p = 0; p < D; p++) s += Hmk[p * ld + i] * Yl[p * ld + j];
                } else {
                    for (int p = 0; p < nm; p++) s += Hmk[p * ld + i] * Yl[p * ld + j];
                }
                double v = M1[i * ld + j] - s;
                M1[i * ld + j] = v;
                if (!isfinite(v)) bad = 1;
            }
        }
        double anybad = T.sum(bad);
        STAMP(28);  // Lambda update
        mirror_upper<NT>(T, M1, n, ld);
        if (anybad > 0) { status = SPG_ST_NONFINITE; finish(); return; }
    }
    if (bd.tinfo_off >= 0 && !stamping) {
        double *dst = arena + bd.tinfo_off;
        for (int it = tid; it < n * n; it += NT) { int i = it / n, j = it - i * n; dst[it] = M1[i * ld + j]; }
    }
    // pseudo-Chow-Liu tree of the kept vertices (a8): fills pairs[0..2(k-1)) in pop order, sets min_gap
    auto chow_liu_tree = [&](const auto TT) -> int {
        if (k == 2) {
            if (TT.tid == 0) { pairs[0] = 0; pairs[1] = 1; }
            TT.sync();
            return (int)SPG_OK;
        }
        // Sigma~ = (Lambda_t + 1 I)^-1 in M2
        if ((use_wave_hw && TT.size == 64)) {
            double ld_, tr_;
            bool ok_ = wave_spd_inverse(M1, ld, n, TT.tid, 1.0, M2, ld_, tr_);
            if (!ok_) return (int)SPG_ST_TIKHONOV_NOT_PD;
            TT.sync();
        } else {
            for (int it = TT.tid; it < n * ld; it += TT.size) M2[it] = M1[it];
            TT.sync();
            for (int i = TT.tid; i < n; i += TT.size) M2[i * ld + i] += 1.0;
            TT.sync();
            chol_lower(TT, M2, n, ld, ev);
            if ((*TT.flag)) return (int)SPG_ST_TIKHONOV_NOT_PD;
            CSTAMP(3);  // cl chol
            tri_inverse_lower(TT, M2, M3, n, ld, ev);
            CSTAMP(4);  // cl triinv
            gram_lower_inverse(TT, M3, M2, n, ld);
        }
        CSTAMP(5);  // cl gram
        // per-vertex diagonal blocks: Cholesky + log det
        for (int v = TT.tid; v < k; v += TT.size) {
            double Ab[DD];
#pragma unroll
            for (int r = 0; r < D; r++)
#pragma unroll
                for (int c = 0; c < D; c++) Ab[r * D + c] = M2[(v * D + r) * ld + v * D + c];
            if (!chol_reg<D>(Ab)) (*TT.flag) = 1;
            LogProd lp;
#pragma unroll
            for (int r = 0; r < D; r++) {
                lp.mul(Ab[r * D + r]);
                // keep 1/L_rr on the diagonal of the staged factor: the pair solves below multiply by it
#pragma unroll
                for (int c = 0; c < D; c++) Lb[v * DD + r * D + c] = (r == c) ? fast_rcp(Ab[r * D + r]) : Ab[r * D + c];
            }
            ldb[v] = 2.0 * lp.value();
        }
        TT.sync();
        CSTAMP(6);  // cl vertex chol
        // pair weights w = ld_i + ld_j - ld_{ij}, ld_{ij} = ld_i + logdet(S_jj - S_ji S_ii^-1 S_ij)
        for (int p = TT.tid; p < L.P; p += TT.size) {
            int i = 0, rem = p;
            while (rem >= k - 1 - i) { rem -= k - 1 - i; i++; }
            int j = i + 1 + rem;
            pij[2 * p] = i; pij[2 * p + 1] = j;
            const double *Li = Lb + i * DD;
            double Y[DD];
#pragma unroll
            for (int c = 0; c < D; c++) {
#pragma unroll
                for (int r = 0; r < D; r++) {
                    double s = M2[(i * D + r) * ld + j * D + c];
#pragma unroll
                    for (int q = 0; q < D; q++) if (q < r) s -= Li[r * D + q] * Y[q * D + c];
                    Y[r * D + c] = s * Li[r * D + r];  // diagonal holds the reciprocal
                }
            }
            double Sb[DD];
#pragma unroll
            for (int r = 0; r < D; r++)
#pragma unroll
                for (int c = 0; c < D; c++) if (c <= r) {
                    double s = M2[(j * D + r) * ld + j * D + c];
#pragma unroll
                    for (int q = 0; q < D; q++) s -= Y[q * D + r] * Y[q * D + c];
                    Sb[r * D + c] = s;
                }
            if (!chol_reg<D>(Sb)) (*TT.flag) = 1;
            LogProd lp;
#pragma unroll
            for (int r = 0; r < D; r++) lp.mul(Sb[r * D + r]);
            double lxy = ldb[i] + 2.0 * lp.value();
            w[p] = -((ldb[i] + ldb[j]) - lxy);  // stored negated: ascending sort == max-heap pop order
        }
        TT.sync();
        if ((*TT.flag)) return (int)SPG_ST_TIKHONOV_NOT_PD;
        CSTAMP(7);  // cl pair weights
        sort_ascending(TT, w, 1, L.P, sorted);
        CSTAMP(8);  // cl sort
        if (TT.tid == 0) {
            // Kruskal in pop order (src/pseudo_chow_liu.cpp:253-289); first `ne` of the bin are used
            for (int v = 0; v < k; v++) comp[v] = v;
            int nacc = 0, last = 0;
            // accepted edges first; rejected ones are only needed when ne > k-1, which has no closed form
            for (int s = 0; s < L.P && nacc < k - 1; s++) {
                int p = sorted[s];
                int i = pij[2 * p], j = pij[2 * p + 1];
                int ci = comp[i], cj = comp[j];
                if (ci != cj) {
                    pairs[2 * nacc] = i; pairs[2 * nacc + 1] = j;
                    nacc++;
                    for (int v = 0; v < k; v++) if (comp[v] == cj) comp[v] = ci;
                }
                last = s;
            }
            int upto = min(last + 1, L.P - 1);
            double g = __builtin_inf();
            for (int s = 0; s < upto; s++) {
                double x = -w[sorted[s]], y = -w[sorted[s + 1]];
                double den = fmax(fmax(fabs(x), fabs(y)), 1e-300);
                g = fmin(g, (x - y) * fast_rcp(den));
            }
            cs[0] = g;
        }
        TT.sync();
        min_gap = cs[0];
        TT.sync();
        CSTAMP(29);  // cl kruskal + gap done
        return (int)SPG_OK;
    };
    if constexpr (is_glc) {
#include "spg_glc_tail.inc"
    }
    STAMP(2);  // schur
    if (k < 2 || stop_after == 2) { finish(); return; }

    // ---------------------------------------------------------------- sparsity pattern (a8)
    int ne;
    {
        int msub = (int)((1 + a.chord_ratio) * (k - 1));
        bool full = msub >= k * (k - 1) / 2;
        if (k == 2) ne = 1;
        else if (a.topology == SPG_TOPO_TREE) ne = k - 1;
        else if (a.topology == SPG_TOPO_DENSE || (a.topology == SPG_TOPO_SUBGRAPH && full)) ne = k * (k - 1) / 2;
        else if (a.topology == SPG_TOPO_SUBGRAPH) ne = msub;
        else { status = SPG_ST_UNSUPPORTED; finish(); return; }
        if (ne * D != n - D) {
            // Chow-Liu still runs in the reference before optimizeInformation discovers that no closed
            // form exists (src/optimizer.cpp:21); the interior-point branch is out of scope.
            status = SPG_ST_NEEDS_INTERIOR_POINT; finish(); return;
        }
    }

    // ================================================================ information recovery (a11, a12)
    // Two routes to the same numbers.
    //  * eigen route — the reference's own: eig(Lambda_t), drop the d gauge directions, Sigma = U S U^T
    //    (src/logdet_function.cpp:14-64,236-279), KLD through M = U^T A U (src/logdet_function.cpp:119-133).
    //  * gauge route — Lambda_t of a blanket of relative-pose edges has an exactly known d-dimensional
    //    null space: the rigid motions of the whole blanket, N = [G_1; ...; G_k] in the vertices' update
    //    coordinates. With N^ an orthonormal basis of it and C = Lambda_t + N^ N^^T (SPD):
    //        U S U^T = C^-1 - N^ N^^T,   J_e N^ = 0  =>  J_e Sigma J_e^T = J_e C^-1 J_e^T,
    //        log det S = -log det C,  tr(S M) = tr(C^-1 A),  log det(U^T A U) = log det(A + N^ N^^T),
    //    so three Cholesky factorisations replace the Jacobi eigen-decomposition (~10x fewer dependent
    //    steps). It is taken only when ||C^-1||_F < 5e4, which proves lambda_{d+1}(Lambda_t) > 1e-5, i.e.
    //    the reference's `smalleigs <= dim` branch; anything else (rank-deficient blankets, failed
    //    factorisations) goes through the eigen route. tests/test_gpu_parity.py checks both routes
    //    against the oracle to 1e-9.
    const int r = n - D;
    double *Ng = smem + L.o_Ng;      // n x D orthonormal gauge basis
    double *tre = smem + L.o_tre;    // per-edge tr(X_e B_e)
    bool gauge_ok = false;
    double logdetS = 0.0;
    double *Sg = M1, *Scr = M3;      // Sigma and scratch for the closed form (swapped on the gauge route)
    auto gauge_chain = [&](const auto TT) {
        // ---- gauge basis
        for (int v = TT.tid; v < k; v += TT.size) {
            const double *X = pose + (m + v) * PSZ;
            double *Gv = Ng + v * DD;
            if (D == 6) {
#pragma unroll
                for (int rr = 0; rr < 3; rr++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        // R^T ; -R^T [t]x ; 0 ; 1/2 R^T   (column c of [t]x is t x e_c)
                        constexpr int A1[3] = {1, 2, 0}, B1[3] = {2, 0, 1};
                        const int ca = A1[c], cb = B1[c];
                        double rt = X[c * 3 + rr];
                        // (t x e_c): component cb = +t[ca]... derive: t x e_c = (t_a e_a + t_b e_b + t_c e_c) x e_c
                        //   e_a x e_c = -e_b , e_b x e_c = +e_a   (a = c+1, b = c+2 cyclic)
                        //   => t x e_c = t_b e_a - t_a e_b
                        double cx_a = X[9 + cb], cx_b = -X[9 + ca];
                        double val = -(X[ca * 3 + rr] * cx_a + X[cb * 3 + rr] * cx_b);  // -(R^T (t x e_c))[rr]
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
        TT.sync();
        CSTAMP(11);  // gauge basis
        // ---- orthonormalise: N^ = N L^-T with N^T N = L L^T (D x D, one lane, registers)
        if (TT.tid < DD) {
            int rr = TT.tid / D, c = TT.tid - rr * D;
            double s = 0;
            for (int i = 0; i < n; i++) s += Ng[i * D + rr] * Ng[i * D + c];
            eT[TT.tid] = s;
        }
        TT.sync();
        if (TT.tid == 0) {
            double Ab[DD], Li[DD];
#pragma unroll
            for (int i = 0; i < DD; i++) Ab[i] = eT[i];
            if (!chol_reg<D>(Ab)) (*TT.flag) = 1;
#pragma unroll
            for (int c = 0; c < D; c++)
#pragma unroll
                for (int i = 0; i < D; i++) {
                    if (i < c) Li[i * D + c] = 0.0;
                    else if (i == c) Li[i * D + c] = 1.0 / Ab[c * D + c];
                    else {
                        double s = 0;
#pragma unroll
                        for (int q = 0; q < D; q++) if (q >= c && q < i) s += Ab[i * D + q] * Li[q * D + c];
                        Li[i * D + c] = -s / Ab[i * D + i];
                    }
                }
#pragma unroll
            for (int i = 0; i < DD; i++) eT[DD + i] = Li[i];
        }
        TT.sync();
        {
            double nv_[D];
            for (int i = TT.tid; i < n; i += TT.size) {
#pragma unroll
                for (int c = 0; c < D; c++) {
                    double s = 0;
#pragma unroll
                    for (int q = 0; q < D; q++) if (q <= c) s += Ng[i * D + q] * eT[DD + c * D + q];
                    nv_[c] = s;
                }
#pragma unroll
                for (int c = 0; c < D; c++) Ng[i * D + c] = nv_[c];
            }
        }
        TT.sync();
        CSTAMP(12);  // orthonormalised
        // ---- C = Lambda_t + N^ N^^T into M3, Cholesky, inverse
        {
            int sh = ceil_log2(n), tot = n << sh;
            for (int it = TT.tid; it < tot; it += TT.size) {
                int i = it >> sh, j = it & ((1 << sh) - 1);
                if (j <= i) {
                    double s = M1[i * ld + j];
#pragma unroll
                    for (int q = 0; q < D; q++) s += Ng[i * D + q] * Ng[j * D + q];
                    M3[i * ld + j] = s;
                    M3[j * ld + i] = s;
                }
            }
            TT.sync();
        }
        CSTAMP(13);  // C formed
        if ((use_wave_hw && TT.size == 64)) {
            double ldC, trC;
            bool ok_ = wave_spd_inverse(M3, ld, n, TT.tid, 0.0, M3, ldC, trC);
            bool fail = !ok_ || ((*TT.flag) != 0);
            TT.sync();
            if (TT.tid == 0) { (*TT.flag) = 0; }
            TT.sync();
            // trace(C^-1) >= lambda_max(C^-1): below 5e4 proves lambda_{d+1}(Lambda_t) > 1e-5
            if (!fail && trC < 5e4 && isfinite(trC)) {
                if (TT.tid == 0) { xch[0] = 1.0; xch[1] = -ldC; }
            }
            CSTAMP(30);  // gauge inverse done
        } else {
            chol_lower(TT, M3, n, ld, Sv);
            CSTAMP(14);  // C chol
            bool fail = ((*TT.flag) != 0);
            TT.sync();
            if (TT.tid == 0) { (*TT.flag) = 0; }
            TT.sync();
            if (!fail) {
                double ldC = chol_logdet(TT, M3, n, ld);
                CSTAMP(15);  // logdet
                tri_inverse_lower(TT, M3, M2, n, ld, Sv);
                CSTAMP(16);  // triinv
                // C^-1 = Li^T Li into M3 with its Frobenius norm
                int sh = ceil_log2(n), tot = n << sh;
                double f2 = 0;
                for (int it = TT.tid; it < tot; it += TT.size) {
                    int i = it >> sh, j = it & ((1 << sh) - 1);
                    if (j <= i) {
                        double s = 0;
                        for (int q = i; q < n; q++) s += M2[q * ld + i] * M2[q * ld + j];
                        M3[i * ld + j] = s;
                        M3[j * ld + i] = s;
                        f2 += (i == j) ? s * s : 2.0 * s * s;
                    }
                }
                double fro2 = TT.sum(f2);
                if (fro2 < 5e4 * 5e4 && isfinite(fro2)) {
                    if (TT.tid == 0) { xch[0] = 1.0; xch[1] = -ldC; }
                }
            }
        }
    };
    // ---- run the two independent chains: Chow-Liu (Lambda_t + I)^-1 ... Kruskal, and the gauge route's
    // C^-1. With two wavefronts per blanket (NT == 128) they run side by side, each wavefront as its own
    // team with wave-local barriers; otherwise one after the other on the whole team.
    if (tid == 0) { xch[0] = 0.0; xch[1] = 0.0; }
    T.sync();
    {
        int st_ = SPG_OK;
        if (split) {
            Team<64> S{tid & 63, nullptr, misc + ((tid < 64) ? 6 : 2)};
            if (tid < 64) { int s2 = chow_liu_tree(S); if (S.tid == 0) misc[7] = s2; }
            else gauge_chain(S);
            T.sync();
            st_ = misc[7];
            min_gap = (k > 2) ? cs[0] : min_gap;
        } else {
            st_ = chow_liu_tree(T);
            if (st_ == SPG_OK && !(a.flags & SPG_FLAG_FORCE_EIG)) gauge_chain(T);
        }
        if (st_ != SPG_OK) { status = st_; finish(); return; }
        T.sync();
        if (xch[0] != 0.0) { gauge_ok = true; logdetS = xch[1]; Sg = M3; Scr = M1; }
        T.sync();
        if (tid == 0) { misc[0] = 0; misc[2] = 0; misc[6] = 0; }
        T.sync();
    }
    STAMP(9);  // both chains done
    if (stop_after == 3) { finish(); return; }
    // ---------------------------------------------------------------- new edge skeleton (a9, a10)
    for (int e = tid; e < ne; e += NT) {
        int va = m + pairs[2 * e], vb = m + pairs[2 * e + 1];
        double *rec = arena + bd.new_off + (int64_t)e * REC;
        if (D == 6) {
            double Z[kIso], q[4];
            iso_inv_mul(pose + va * PSZ, pose + vb * PSZ, Z);
            R_to_quat(Z, q);
            double *rw = COH ? (recbuf + e * REC) : rec;
            rw[0] = Z[9]; rw[1] = Z[10]; rw[2] = Z[11]; rw[3] = q[0]; rw[4] = q[1]; rw[5] = q[2]; rw[6] = q[3];
            se3_edge_jac(pose + va * PSZ, pose + vb * PSZ, Z, nJ + e * 2 * DD, nJ + e * 2 * DD + DD, nullptr);
        } else {
            double z[3];
            se2_between(pose + va * PSZ, pose + vb * PSZ, z);
            double *rw = COH ? (recbuf + e * REC) : rec;
            rw[0] = z[0]; rw[1] = z[1]; rw[2] = z[2];
            se2_edge_jac(pose + va * PSZ, pose + vb * PSZ, z, nJ + e * 2 * DD, nJ + e * 2 * DD + DD, nullptr);
        }
    }
    T.sync();
    if (ne == 1) {
        // single-measurement case goes through sparseJacobian(): entries below epsilon are dropped
        // (src/logdet_function.cpp:243-247,335)
        for (int it = tid; it < 2 * DD; it += NT) if (fabs(nJ[it]) < 2.220446049250313e-16) nJ[it] = 0.0;
        T.sync();
    }

    STAMP(10);  // new edges
    if (stop_after == 4) { finish(); return; }
    if (!gauge_ok) {
        // ------------------------------------------------------------ eigen route: spectrum of Lambda_t
        for (int it = tid; it < n * ld; it += NT) M3[it] = M1[it];
        T.sync();
        if (!jacobi_eigh<NT>(T, M3, M2, n, ld, cs)) { status = SPG_ST_EIG_FAIL; finish(); return; }
        for (int i = tid; i < n; i += NT) ev[i] = M3[i * ld + i];
        T.sync();
        if (stop_after == 5) { finish(); return; }
        sort_ascending<NT>(T, ev, 1, n, perm);
        {
            double cnt = 0;
            for (int i = tid; i < n; i += NT) cnt += (ev[i] < 1e-5) ? 1.0 : 0.0;
            int smalleigs = (int)T.sum(cnt);
            if (smalleigs <= D) {
                for (int j = tid; j < r; j += NT) { int idx = perm[D + j]; keep[j] = idx; Sv[j] = 1.0 / ev[idx]; }
            } else {
                info |= SPG_INFO_RANK_DEFICIENT;
                if (tid == 0) {
                    // chooseDimensions (src/logdet_function.cpp:66-81): drop the D candidates with the
                    // smallest ||J u||; S clamped as at src/logdet_function.cpp:55
                    double *nrm = cs;  // smalleigs <= n doubles
                    for (int c = 0; c < smalleigs; c++) {
                        int col = perm[c];
                        double s2 = 0;
                        for (int e = 0; e < ne; e++) {
                            int oa = pairs[2 * e] * D, ob = pairs[2 * e + 1] * D;
                            const double *Ja = nJ + e * 2 * DD, *Jb = Ja + DD;
                            for (int rr = 0; rr < D; rr++) {
                                double s = 0;
                                for (int p = 0; p < D; p++) {
                                    double ja = Ja[rr * D + p], jb = Jb[rr * D + p];
                                    if (fabs(ja) >= 2.220446049250313e-16) s += ja * M2[(oa + p) * ld + col];
                                    if (fabs(jb) >= 2.220446049250313e-16) s += jb * M2[(ob + p) * ld + col];
                                }
                                s2 += s * s;
                            }
                        }
                        nrm[c] = sqrt(s2);
                    }
                    // mark the D smallest (norm, then sorted position); keep[] doubles as the mark array
                    for (int i = 0; i < n; i++) keep[i] = 0;
                    for (int t = 0; t < D; t++) {
                        int best = -1;
                        for (int c = 0; c < smalleigs; c++) {
                            if (keep[c]) continue;
                            if (best < 0 || nrm[c] < nrm[best]) best = c;
                        }
                        keep[best] = 1;
                    }
                    double lmax = ev[perm[n - 1]];
                    int j = 0;
                    // compact in place: positions are visited in ascending order and j <= i always
                    for (int i = 0; i < n; i++) {
                        bool dropped = keep[i] != 0;
                        if (!dropped) {
                            int idx = perm[i];
                            keep[j] = idx;
                            Sv[j] = fmin(fabs(1.0 / ev[idx]), 1e6 / lmax);
                            j++;
                        }
                    }
                }
            }
            T.sync();
        }
        {
            double s = 0;
            for (int j = tid; j < r; j += NT) s += log(Sv[j]);
            logdetS = T.sum(s);
        }
        // Sigma = U S U^T into M1
        {
            int sh = ceil_log2(n), tot = n << sh;
            for (int it = tid; it < tot; it += NT) {
                int i = it >> sh, j = it & ((1 << sh) - 1);
                if (j <= i) {
                    double s = 0;
                    for (int q = 0; q < r; q++) { int c = keep[q]; s += M2[i * ld + c] * Sv[q] * M2[j * ld + c]; }
                    M1[i * ld + j] = s;
                    M1[j * ld + i] = s;
                }
            }
            T.sync();
        }
    }
    STAMP(17);  // Cinv+guard
    if (stop_after == 6) { finish(); return; }

    // ---------------------------------------------------------------- closed form X_e (a11)
    {
        double *Pw = Scr;               // ne x 3 x DD
        double *Bk = Scr + ne * 3 * DD; // ne x DD
        for (int it = tid; it < ne * 3 * DD; it += NT) {
            int e = it / (3 * DD), rem = it - e * 3 * DD, wch = rem / DD, rc = rem - wch * DD, rr = rc / D, c = rc - rr * D;
            int oa = pairs[2 * e] * D, ob = pairs[2 * e + 1] * D;
            const double *J = nJ + e * 2 * DD + (wch == 2 ? DD : 0);
            int ro = (wch == 2) ? ob : oa, co = (wch == 0) ? oa : ob;
            double s = 0;
#pragma unroll
            for (int p = 0; p < D; p++) s += J[rr * D + p] * Sg[(ro + p) * ld + co + c];
            Pw[it] = s;
        }
        T.sync();
        STAMP(31);  // closed form: J Sigma
        for (int it = tid; it < ne * DD; it += NT) {
            int e = it / DD, rc = it - e * DD, rr = rc / D, c = rc - rr * D;
            const double *Ja = nJ + e * 2 * DD, *Jb = Ja + DD;
            const double *P0 = Pw + e * 3 * DD, *P1 = P0 + DD, *P2 = P1 + DD;
            double t1rc = 0, t1cr = 0, t2rc = 0, t2cr = 0, t3rc = 0, t3cr = 0;
#pragma unroll
            for (int p = 0; p < D; p++) {
                t1rc += P0[rr * D + p] * Ja[c * D + p]; t1cr += P0[c * D + p] * Ja[rr * D + p];
                t2rc += P1[rr * D + p] * Jb[c * D + p]; t2cr += P1[c * D + p] * Jb[rr * D + p];
                t3rc += P2[rr * D + p] * Jb[c * D + p]; t3cr += P2[c * D + p] * Jb[rr * D + p];
            }
            double v = (ne == 1) ? (t1rc + t2rc + t2cr + t3rc)
                                 : (0.5 * (t1rc + t1cr) + (t2rc + t2cr) + 0.5 * (t3rc + t3cr));
            Bk[it] = v;
        }
        T.sync();
        STAMP(32);  // closed form: B_e
        for (int e = tid; e < ne; e += NT) {
            double Ab[DD], Xr[DD];
#pragma unroll
            for (int i = 0; i < DD; i++) Ab[i] = Bk[e * DD + i];
            double trs = 0;
            if (!chol_reg<D>(Ab)) misc[0] = 1;
            chol_inverse_reg<D>(Ab, Xr);
            double *rec = COH ? (recbuf + e * REC + PS) : (arena + bd.new_off + (int64_t)e * REC + PS);
            int pidx = 0;
#pragma unroll
            for (int i = 0; i < D; i++)
#pragma unroll
                for (int j = 0; j < D; j++) {
                    X[e * DD + i * D + j] = Xr[i * D + j];
                    trs += Xr[i * D + j] * Bk[e * DD + ((j <= i) ? (i * D + j) : (j * D + i))];
                    if (j >= i) rec[pidx++] = Xr[i * D + j];
                }
            tre[e] = trs;
        }
        T.sync();
        STAMP(33);  // closed form: X_e = B_e^-1, records written
        if (misc[0]) { status = SPG_ST_CLOSED_FORM_NOT_PD; finish(); return; }
    }
    n_new = ne;
    publish();  // the graph update can proceed; the KLD below is reported later
    STAMP(18);  // closed form
    if (stop_after == 7) { finish(); return; }

    // ---------------------------------------------------------------- per-blanket KLD (a12)
    {
        // A = J^T X J (upper blocks accumulated, then mirrored) into Am
        double *Am = gauge_ok ? M2 : M1;
        double *XJ = gauge_ok ? M1 : M3;  // ne x 2 x DD
        for (int it = tid; it < ne * 2 * DD; it += NT) {
            int e = it / (2 * DD), rem = it - e * 2 * DD, wch = rem / DD, rc = rem - wch * DD, rr = rc / D, c = rc - rr * D;
            const double *J = nJ + e * 2 * DD + wch * DD;
            double s = 0;
#pragma unroll
            for (int p = 0; p < D; p++) s += X[e * DD + rr * D + p] * J[p * D + c];
            XJ[it] = s;
        }
        for (int it = tid; it < n * ld; it += NT) Am[it] = 0.0;
        T.sync();
        for (int e = 0; e < ne; e++) {
            int oa = pairs[2 * e] * D, ob = pairs[2 * e + 1] * D;
            const double *Ja = nJ + e * 2 * DD, *Jb = Ja + DD;
            const double *XJa = XJ + e * 2 * DD, *XJb = XJa + DD;
            for (int it = tid; it < 3 * DD; it += NT) {
                int blk = it / DD, rc = it - blk * DD, rr = rc / D, c = rc - rr * D;
                const double *Jl = (blk == 2) ? Jb : Ja;
                const double *Xr_ = (blk == 0) ? XJa : XJb;
                double s = 0;
#pragma unroll
                for (int p = 0; p < D; p++) s += Jl[p * D + rr] * Xr_[p * D + c];
                int R = ((blk == 2) ? ob : oa) + rr, Cc = ((blk == 0) ? oa : ob) + c;
                Am[R * ld + Cc] += s;
            }
            T.sync();
        }
        mirror_upper<NT>(T, Am, n, ld);
        STAMP(19);  // A assembled
        if (gauge_ok) {
            // kld = 1/2 ( tr(C^-1 A) - log det(A + N^ N^^T) + log det C - r ), tr(C^-1 A) = sum_e tr(X_e B_e)
            int sh = ceil_log2(n), tot = n << sh;
            for (int it = tid; it < tot; it += NT) {
                int i = it >> sh, j = it & ((1 << sh) - 1);
                if (j < n) {
                    double s = 0;
#pragma unroll
                    for (int q = 0; q < D; q++) s += Ng[i * D + q] * Ng[j * D + q];
                    Am[i * ld + j] += s;
                }
            }
            T.sync();
            double tr;
            {
                double s = 0;
                for (int e = tid; e < ne; e += NT) s += tre[e];
                tr = T.sum(s);
            }
            STAMP(20);  // A + NN
            if (use_wave_hw && NT <= 128) {
                // the first wavefront factorises in registers; the result reaches every lane through LDS
                if (tid < 64) {
                    double ldA_;
                    bool ok_ = wave_spd_logdet(Am, ld, n, tid, ldA_);
                    if (tid == 0) { xch[2] = ok_ ? 1.0 : 0.0; xch[3] = ldA_; }
                }
                T.sync();
                if (xch[2] == 0.0) { kld = __builtin_inf(); status = SPG_ST_KLD_NOT_PD; }
                else kld = 0.5 * (tr - xch[3] - logdetS - (double)r);
            } else {
                chol_lower<NT>(T, Am, n, ld);
                if (misc[0]) {
                    kld = __builtin_inf();
                    status = SPG_ST_KLD_NOT_PD;
                } else {
                    double ldA = chol_logdet<NT>(T, Am, n, ld);
                    kld = 0.5 * (tr - ldA - logdetS - (double)r);
                }
            }
            STAMP(21);  // chol A
        } else {
            // Tm = A * U_kept  (n x r) into M3 ; XJ no longer needed
            int shr = ceil_log2(r > 0 ? r : 1);
            for (int it = tid; it < (n << shr); it += NT) {
                int i = it >> shr, j = it & ((1 << shr) - 1);
                if (j < r) {
                    int c = keep[j];
                    double s = 0;
                    for (int p = 0; p < n; p++) s += M1[i * ld + p] * M2[p * ld + c];
                    M3[i * ld + j] = s;
                }
            }
            T.sync();
            // M = U_kept^T Tm  (r x r) into M1 (upper needed, lower mirrored)
            for (int it = tid; it < (r << shr); it += NT) {
                int i = it >> shr, j = it & ((1 << shr) - 1);
                if (j < r && j >= i) {
                    int c = keep[i];
                    double s = 0;
                    for (int p = 0; p < n; p++) s += M2[p * ld + c] * M3[p * ld + j];
                    M1[i * ld + j] = s;
                    M1[j * ld + i] = s;
                }
            }
            T.sync();
            double tr;
            {
                double s = 0;
                for (int i = tid; i < r; i += NT) s += M1[i * ld + i] * Sv[i];
                tr = T.sum(s);
            }
            chol_lower<NT>(T, M1, r, ld);
            if (misc[0]) {
                kld = __builtin_inf();
                status = SPG_ST_KLD_NOT_PD;
            } else {
                double ldM = chol_logdet<NT>(T, M1, r, ld);
                kld = 0.5 * (tr - ldM - logdetS - (double)r);
            }
        }
    }
    STAMP(22);  // end
    finish();
}

template <int D, int NT, bool GWS, int ALG>
__global__ void __launch_bounds__(NT) blanket_kernel(KArgs a) {
    extern __shared__ double smem[];
    const int b = a.list[blockIdx.x];
    const spg_blanket_desc bd = a.blk[b];
    blanket_body<D, NT, GWS, ALG>(a, bd, a.vpo + bd.vert_begin, a.er + bd.edge_begin, a.ev, (int)blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------- Local linearisation point, pre-pass
// buildSubgraph with SparsityOptions::Local (src/vertex_remover.cpp:304-391) for the blankets the generic NFR kernel takes
// (interior point, correlated patterns, clusters): the blanket's estimates are copied into a scratch block; if every vertex
// but the first removed one sits in exactly one pose-pose edge they are re-initialised in closed form (the first removed
// vertex at the origin, each neighbour at z or z^-1 of its edge, edges in blanket order: :304-381) and flag = 1; otherwise
// flag = 0 and the host runs the 10 LM iterations with that vertex fixed (:382-391) on the scratch block. The kernels that
// follow read the blanket's poses from the scratch block (their pose offsets are redirected) and take them as they are.
__device__ __forceinline__ void tq_rotate(const double *q, const double *v, double *o) {
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    const double cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
    o[0] = v[0] + 2.0 * (w * cx + (uy * cz - uz * cy));
    o[1] = v[1] + 2.0 * (w * cy + (uz * cx - ux * cz));
    o[2] = v[2] + 2.0 * (w * cz + (ux * cy - uy * cx));
}
__device__ __forceinline__ void tq_compose(const double *a, const double *b, double *o) {   // o = a * b (may alias neither)
    double r[3];
    tq_rotate(a + 3, b, r);
    o[0] = a[0] + r[0]; o[1] = a[1] + r[1]; o[2] = a[2] + r[2];
    const double ax = a[3], ay = a[4], az = a[5], aw = a[6], bx = b[3], by = b[4], bz = b[5], bw = b[6];
    double x = aw * bx + ax * bw + ay * bz - az * by, y = aw * by - ax * bz + ay * bw + az * bx,
           z = aw * bz + ax * by - ay * bx + az * bw, w = aw * bw - ax * bx - ay * by - az * bz;
    const double nn = 1.0 / sqrt(x * x + y * y + z * z + w * w), sg = (w < 0) ? -nn : nn;
    o[3] = x * sg; o[4] = y * sg; o[5] = z * sg; o[6] = w * sg;
}
__device__ __forceinline__ void tq_inverse(const double *a, double *o) {
    double q[4] = {-a[3], -a[4], -a[5], a[6]}, r[3];
    tq_rotate(q, a, r);
    o[0] = -r[0]; o[1] = -r[1]; o[2] = -r[2]; o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
}

template <int D>
__global__ __launch_bounds__(64) void local_point_kernel(const double *arena, double *lpose, const spg_blanket_desc *blk, const int64_t *vpo,
                                                         const spg_edge_ref *er, const int32_t *ev, const int32_t *list, const int64_t *lp_off, int32_t *flag) {
    constexpr int PS = (D == 6) ? 7 : 3;
    __shared__ int cnt[1024];
    const int tid = threadIdx.x;
    const spg_blanket_desc bd = blk[list[blockIdx.x]];
    const int nv = bd.n_vert;
    double *P = lpose + lp_off[blockIdx.x];
    for (int it = tid; it < nv * PS; it += 64) P[it] = arena[vpo[bd.vert_begin + it / PS] + it % PS];
    for (int v = tid; v < nv && v < 1024; v += 64) cnt[v] = 0;
    __syncthreads();
    if (tid != 0) return;
    bool ok = nv <= 1024;
    for (int e = 0; ok && e < bd.n_edge; e++) {
        const spg_edge_ref r = er[bd.edge_begin + e];
        if (r.kind != SPG_EDGE_BINARY) { ok = false; break; }
        const int vi = ev[r.vbegin], vj = ev[r.vbegin + 1];
        if (vi != 0) cnt[vi]++;
        if (vj != 0) cnt[vj]++;
    }
    for (int v = 1; ok && v < nv; v++) if (cnt[v] > 1) ok = false;
    if (ok) {
        if (D == 3) { P[0] = 0; P[1] = 0; P[2] = 0; }
        else { P[0] = 0; P[1] = 0; P[2] = 0; P[3] = 0; P[4] = 0; P[5] = 0; P[6] = 1; }
        for (int e = 0; e < bd.n_edge; e++) {
            const spg_edge_ref r = er[bd.edge_begin + e];
            const int vi = ev[r.vbegin], vj = ev[r.vbegin + 1];
            if (vi == 0 && vj == 0) continue;
            const double *z = arena + r.off;
            if (D == 3) {
                if (vi == 0) {
                    const double c = cos(P[2]), sn = sin(P[2]);
                    double *o = P + vj * 3;
                    o[0] = P[0] + c * z[0] - sn * z[1]; o[1] = P[1] + sn * z[0] + c * z[1]; o[2] = normalize_theta(P[2] + z[2]);
                } else {
                    const double cz = cos(z[2]), sz = sin(z[2]);
                    const double zi[3] = {-(cz * z[0] + sz * z[1]), -(-sz * z[0] + cz * z[1]), normalize_theta(-z[2])};
                    const double *a = P + vj * 3;
                    const double c = cos(a[2]), sn = sin(a[2]);
                    const double ox = a[0] + c * zi[0] - sn * zi[1], oy = a[1] + sn * zi[0] + c * zi[1], ot = normalize_theta(a[2] + zi[2]);
                    double *o = P + vi * 3;
                    o[0] = ox; o[1] = oy; o[2] = ot;
                }
            } else {
                double zq[7], out[7];
                for (int i = 0; i < 7; i++) zq[i] = z[i];
                { const double nn = 1.0 / sqrt(zq[3] * zq[3] + zq[4] * zq[4] + zq[5] * zq[5] + zq[6] * zq[6]); for (int i = 3; i < 7; i++) zq[i] *= nn; }
                if (vi == 0) { tq_compose(P, zq, out); for (int i = 0; i < 7; i++) P[vj * 7 + i] = out[i]; }
                else { double zi[7]; tq_inverse(zq, zi); tq_compose(P + vj * 7, zi, out); for (int i = 0; i < 7; i++) P[vi * 7 + i] = out[i]; }
            }
        }
    }
    flag[blockIdx.x] = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------- persistent worker
// Narrow rounds (a few dozen blankets that depend on the previous few dozen) are bound by the launch-to-result latency
// of a batch, not by arithmetic. For them the device runs ONE long-lived kernel per marginalisation: kWorkerWGs
// workgroups that take tickets from a queue in fine-grained device memory which the host fills through the PCIe BAR
// (posted stores: blanket packets, then the item slots, then `tail`). A batch then costs the host a few cache-line
// writes instead of a kernel launch, and reaches a workgroup ~1-2 us after the store of `tail` instead of ~15 us after
// hipLaunchKernel. Results travel as before (out records in the pinned host mailbox, ready / final words).
//
// Packet of one blanket (8-byte words; written by the host, read with system-scope loads):
//   [0] arena  [1] mailbox (device address, 0 = none)  [2] mail_base  [3] out_off  [4] new_off  [5] tinfo_off
//   [6] n_vert | n_remove << 32   [7] n_edge | n_new_max << 32   [8] n_new_vert_max | scratch << 32
//   [9] topology | flags << 32    [10] lin_point | tag << 32     [11] chord_ratio (f64 bits)
//   [12] n_words | 0              then vpo[n_vert] (i64), er[n_edge] (spg_edge_ref, 3 words each, vbegin relative to the
//   packet's ev), ev (i32 pairs)
using spg::kQCap; using spg::kPktHdr; using spg::kPktWords; using spg::kPktStride; using spg::kWorkerMaxN; using spg::kBells; using spg::kBellStride;
using spg::WorkQ;

__device__ __forceinline__ unsigned long long load_sys(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}


template <int D, int ALG>
__global__ void __launch_bounds__(128) blanket_worker(WorkQ *q, unsigned long long *ticket, long long idle_ticks, int n_bells, int lazy) {
    extern __shared__ double smem[];
    __shared__ unsigned long long pk[kPktWords];
    __shared__ unsigned long long s_item;
    const int tid = threadIdx.x;
    for (;;) {
        if (tid == 0) {
            const unsigned long long my = atomicAdd(ticket, 1ULL);
            unsigned long long it = 0;
            long long t0 = wall_clock64();
            unsigned long long seen = ~0ULL;
            for (;;) {
                // (relaxed: an acquire here would invalidate this XCD's L1 / L2 on every poll of every idle workgroup — the item and
                //  the packet are read with system-scope loads that bypass the caches anyway)
                const unsigned long long tail = load_sys(&q->tail[(blockIdx.x % n_bells) * kBellStride]);
                if (tail > my) { it = load_sys(&q->item[my % kQCap]); break; }
                if (load_sys(&q->stop)) break;
                if (tail != seen) { seen = tail; t0 = wall_clock64(); }
                else if (wall_clock64() - t0 > idle_ticks) break;   // the host went away: every wave must be able to leave
                // the next few tickets poll back to back, the others doze (one poll is an uncached HBM read)
                const unsigned long long ahead = my - tail;
                if (lazy) {
                    if (ahead >= 2) for (int z = 0; z < lazy; z++) __builtin_amdgcn_s_sleep(32);   // lazy x ~0.9 us
                } else {
                    if (ahead >= 64) __builtin_amdgcn_s_sleep(64);
                    else if (ahead >= 4) __builtin_amdgcn_s_sleep(8);
                }
            }
            s_item = it;
        }
        __syncthreads();
        const unsigned long long it = s_item;
        if (!it) return;
        const long long t_pick = wall_clock64();
        const unsigned long long *gp = reinterpret_cast<const unsigned long long *>(it);
        if (tid < kPktStride) pk[tid] = load_sys(gp + tid);
        __syncthreads();
        const int n_words = (int)(uni64(pk[12]) & 0xffffffffu);
        for (int w = kPktStride + tid; w < n_words && w < kPktWords; w += 128) pk[w] = load_sys(gp + w);
        // (records written by other workgroups of this kernel are read with agent-scope loads inside the body — an
        //  agent-scope acquire fence here would also drop the kernel's code and the poses from this XCD's L2 per item)
        __syncthreads();
        unsigned long long h[kPktHdr];
#pragma unroll
        for (int w = 0; w < kPktHdr; w++) h[w] = uni64(pk[w]);
        KArgs a;
        a.arena = reinterpret_cast<double *>(h[0]);
        a.mail = reinterpret_cast<double *>(h[1]);
        a.mail_base = (int64_t)h[2];
        a.blk = nullptr; a.vpo = nullptr; a.er = nullptr; a.ev = nullptr; a.list = nullptr; a.gws = nullptr; a.gws_stride = 0;
        a.algorithm = ALG == SPG_ALG_GLC ? SPG_ALG_GLC : SPG_ALG_NFR;
        a.topology = (int)(h[9] & 0xffffffffu); a.flags = (int)(h[9] >> 32);
        a.lin_point = (int)(h[10] & 0xffffffffu); a.tag = (int)(h[10] >> 32);
        a.chord_ratio = __longlong_as_double((long long)h[11]);
        spg_blanket_desc bd;
        bd.vert_begin = 0; bd.edge_begin = 0;
        bd.out_off = (int64_t)h[3]; bd.new_off = (int64_t)h[4]; bd.tinfo_off = (int64_t)h[5]; bd.new_len = 0;
        bd.n_vert = (int)(h[6] & 0xffffffffu); bd.n_remove = (int)(h[6] >> 32);
        bd.n_edge = (int)(h[7] & 0xffffffffu); bd.n_new_max = (int)(h[7] >> 32);
        bd.n_new_vert_max = (int)(h[8] & 0xffffffffu); bd.pad_ = (int)(h[8] >> 32);
        const int64_t *vpo = reinterpret_cast<const int64_t *>(pk + kPktHdr);
        const spg_edge_ref *er = reinterpret_cast<const spg_edge_ref *>(pk + kPktHdr + bd.n_vert);
        const int32_t *ev = reinterpret_cast<const int32_t *>(pk + kPktHdr + bd.n_vert + 3 * bd.n_edge);
        const long long t_body = wall_clock64();
        // inlined: as an out-of-line call the body saved / restored 56 callee-saved VGPRs per blanket through scratch —
        // 28 KB per workgroup and item, 1.3 GB of HBM writes per step of the bench workload (rocprofv3 WRITE_SIZE). Inlined
        // it needs the whole register file of a SIMD (256 VGPRs + ~110 AGPRs as spill space, no scratch): one wave per
        // SIMD, which is what a grid of one 2-wave workgroup per CU uses anyway.
        blanket_body<D, 128, false, ALG, true>(a, bd, vpo, er, ev, 0, smem);
        __syncthreads();
        if ((h[12] >> 32) & 0x40000000u) {
            // diagnostic (SPG_WORKER_STAMP=1): min_gap slot <- ticks to the ready word + 1e-6 * ticks to the final word, from the pick (100 MHz wall clock); the final
            // word is rewritten after it so that the host's late harvest sees the stamp
            if (tid == 0 && a.mail) {
                double *orec = a.mail + (bd.out_off - a.mail_base);
                const double *orl_ = smem + make_layout(D, 128, bd.n_vert - bd.n_remove, bd.n_remove, ALG, a.topology, bd.pad_).o_eO;
                orec[3] = (double)((long long)orl_[5] - t_pick) + 1e-6 * (double)(wall_clock64() - t_pick);   // ticks of 10 ns: ready, final
                __threadfence_system();
            }
        }
    }
}

}  // namespace

// =================================================================================== HIP backend
namespace spg {

// (`err` is HipBackend::err, 512 bytes, as the member array or through a char * alias)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(err, (size_t)512, "%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return SPG_EHIP; } } while (0)

struct HipBackend {
    int device = 0;
    char err[512] = {0};
    struct Timed { hipEvent_t a, b; double bytes; int blankets; };
    // Independent launch slots (stream + descriptor buffers + pinned staging + pinned mailbox +
    // large-blanket workspace) so that the host can prepare and launch one batch of blankets while the
    // previous one is still running.
    struct Slot {
        hipStream_t stream = nullptr;
        void *d_desc = nullptr, *d_gws = nullptr, *d_ipws = nullptr, *d_lpose = nullptr, *d_lmeta = nullptr;
        size_t c_desc = 0, c_gws = 0, c_ipws = 0, c_lpose = 0, c_lmeta = 0;
        void *h_mail = nullptr, *d_mail = nullptr;   // pinned host mailbox the kernel writes out records into
        size_t c_mail = 0;
        void *h_stage = nullptr;                      // pinned host staging for the descriptor upload
        void *d_stage = nullptr;                      // the same buffer in the device's address space
        size_t c_stage = 0;
        bool busy = false;                            // work was queued on the stream since its last synchronisation
        std::vector<int32_t> bin_lists[5];            // scratch of hip_run_round
        // Waiting for a slot goes through the event recorded behind its last kernel: hipStreamSynchronize on
        // a stream that ends in a kernel has to submit a marker first and was measured at ~10 us per call,
        // hipEventSynchronize on an already recorded event at ~1 us.
        hipEvent_t done = nullptr, wait_ev = nullptr;
        void *d_bar = nullptr;                        // fine-grained device memory the host writes through the PCIe BAR
        size_t c_bar = 0;
        // blankets of the last batch that went through the persistent worker: addresses of their final words in the
        // pinned mailbox (wait_slot polls them: the slot's buffers may be rewritten once all of them are final)
        std::vector<const volatile double *> finals;
        double final_word = 0;
        void *d_pkt = nullptr;                        // fine-grained device memory for the packets (host writes, BAR)
        size_t c_pkt = 0;
        bool stream_dirty = false;                    // something was queued on `stream` since the last wait
        std::vector<Timed> pending;
        // per slot, so that a submission thread working on one slot and the graph thread draining
        // another never share state
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
        double prof_ms = 0, prof_bytes = 0;
        long long prof_launches = 0, prof_blankets = 0;
    };
    static constexpr int NSLOT = 8;
    Slot slots[NSLOT];
    // the persistent worker kernel (one per backend, alive between the first narrow batch of a marginalisation and
    // the next full synchronisation)
    struct Worker {
        hipStream_t stream = nullptr;
        WorkQ *q = nullptr;                           // fine-grained device memory; the host stores through the BAR
        unsigned long long *ticket = nullptr;         // device memory
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        bool running = false, disabled = false;
        int D = 0, alg = 0;
        unsigned long long tail = 0;                  // host copy of q->tail
        double bytes = 0;                             // algorithmic bytes / blankets handed over since it started
        long long blankets = 0;
        int bells = 1, lazy = 0;                     // (more doorbell copies / lazier polls: measured, no effect)
        int wgs = 256;                                // one per CU: launched kernels must always find room next to it
        std::chrono::steady_clock::time_point last_push;
    } worker;
    int batches_in_call = 0;                          // batches since the last full synchronisation
    // streaming driver (hip_stream_open): its own packet ring (fine-grained device memory) and pinned mailbox
    void *st_pkt = nullptr, *st_hmail = nullptr, *st_dmail = nullptr;
    size_t c_st_pkt = 0, c_st_mail = 0;
    int worker_cooldown = 0;                          // batches to go before the worker is considered again
    double prof_big_ms = 0, prof_big_flops = 0;       // large-blanket dense pipeline (always accumulated)
    long long prof_big_count = 0;
    int prof_big_nmax = 0;
    double prof_worker_ms = 0, prof_worker_bytes = 0;
    long long prof_worker_runs = 0, prof_worker_blankets = 0;
    int lds_limit = 160 * 1024;
    bool large_bar = false;       // the host can store straight into device memory (hipDeviceAttributeIsLargeBar)
    std::atomic<int> n_launches{0};
    bool force_one_wave = false;  // SPG_ONE_WAVE=1: never use the two-wavefront latency variant (A/B timing)
    // optional per-launch timing with HIP events on the launch stream (bench.py roofline leg)
    bool profiling = false;
    int prof_stride = 1, prof_tick = 0;   // time every prof_stride-th launch (1 = all)

    int ensure(Slot &S, void **p, size_t *cap, size_t need) {
        if (need <= *cap) return 0;
        size_t nc = std::max(need, *cap * 2);
        if (*p) { if (int rc = worker_stop()) return rc; HIPCHK(hipStreamSynchronize(S.stream)); HIPCHK(hipFree(*p)); *p = nullptr; }
        HIPCHK(hipMalloc(p, nc));
        *cap = nc;
        return 0;
    }
    int wait_slot(Slot &S) {
        if (S.wait_ev) { HIPCHK(hipEventSynchronize(S.wait_ev)); S.wait_ev = nullptr; }
        else if (S.stream_dirty) HIPCHK(hipStreamSynchronize(S.stream));
        S.stream_dirty = false;
        if (!S.finals.empty()) {
            const auto t0 = std::chrono::steady_clock::now();
            for (const volatile double *p : S.finals) {
                uint32_t spins = 0;
                while (*p != S.final_word) {
                    if ((++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
                        snprintf(err, sizeof err, "persistent worker: a blanket did not complete within 10 s");
                        return SPG_EHIP;
                    }
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            S.finals.clear();
        }
        S.busy = false;
        return 0;
    }
    int worker_stop();
    int worker_start(int D, int alg);
    int ensure_stage(Slot &S, size_t need) {
        if (need <= S.c_stage) return 0;
        size_t nc = std::max(need, S.c_stage * 2);
        if (S.h_stage) { if (int rc = worker_stop()) return rc; HIPCHK(hipStreamSynchronize(S.stream)); HIPCHK(hipHostFree(S.h_stage)); S.h_stage = nullptr; }
        HIPCHK(hipHostMalloc(&S.h_stage, nc, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer(&S.d_stage, S.h_stage, 0));
        S.c_stage = nc;
        return 0;
    }
};

template <int D, int NT, bool GWS, int ALG>
static int launch_bin(HipBackend *hb, HipBackend::Slot &S, const KArgs &ka, int nblocks, size_t lds_bytes, double alg_bytes) {
    char *err = hb->err;
    auto kern = blanket_kernel<D, NT, GWS, ALG>;
    if (lds_bytes > 64 * 1024)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    HipBackend::Timed t{};
    const bool timed = hb->profiling && (hb->prof_tick++ % hb->prof_stride == 0);
    if (timed) {
        if (S.pool.empty()) {
            HIPCHK(hipEventCreate(&t.a));
            HIPCHK(hipEventCreate(&t.b));
        } else { t.a = S.pool.back().first; t.b = S.pool.back().second; S.pool.pop_back(); }
        t.bytes = alg_bytes; t.blankets = nblocks;
        // one runtime call: the dispatch itself carries the two events (start / stop of this kernel)
        hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), (uint32_t)lds_bytes, S.stream, t.a, t.b, 0, ka);
        HIPCHK(hipGetLastError());
        S.pending.push_back(t);
        S.wait_ev = t.b;
    } else {
        hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), (uint32_t)lds_bytes, S.stream, nullptr, S.done, 0, ka);
        HIPCHK(hipGetLastError());
        S.wait_ev = S.done;
    }
    hb->n_launches++;
    S.stream_dirty = true;
    return 0;
}

// ---------------------------------------------------------------------------------- worker control (host)
static inline void bar_fence() {
    std::atomic_thread_fence(std::memory_order_release);
#if defined(__x86_64__)
    __builtin_ia32_sfence();   // write-combining buffers drained: stores through the BAR leave in program order
#endif
}

template <int D, int ALG>
static int launch_worker(HipBackend *hb, size_t lds) {
    char *err = hb->err;
    auto kern = blanket_worker<D, ALG>;
    if (lds > 48 * 1024)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long idle_ticks = 10LL * 100000000LL;   // 10 s of the 100 MHz wall clock
    hipExtLaunchKernelGGL(kern, dim3(hb->worker.wgs), dim3(128), (uint32_t)lds, hb->worker.stream, hb->worker.ev_a, hb->worker.ev_b, 0,
                          hb->worker.q, hb->worker.ticket, idle_ticks, hb->worker.bells, hb->worker.lazy);
    HIPCHK(hipGetLastError());
    return 0;
}

int HipBackend::worker_start(int D, int alg) {
    Worker &W = worker;
    if (W.running && W.D == D && W.alg == alg) {
        // an idle worker leaves by itself after 10 s without a new item (every wave needs an exit the host cannot
        // withhold): after a pause of more than 1 s retire it and start a fresh one rather than trust a half-gone grid
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - W.last_push).count() < 1.0) return 0;
    }
    if (W.running) if (int rc = worker_stop()) return rc;
    if (!W.stream) {
        HIPCHK(hipStreamCreateWithFlags(&W.stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&W.ev_a));
        HIPCHK(hipEventCreate(&W.ev_b));
        if (hipExtMallocWithFlags((void **)&W.q, sizeof(WorkQ), hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); W.disabled = true; return 1; }
        HIPCHK(hipMalloc((void **)&W.ticket, 64));
        const char *e = getenv("SPG_WORKER_WGS");
        if (e && atoi(e) > 0) W.wgs = atoi(e);
        if ((e = getenv("SPG_WORKER_BELLS")) && atoi(e) >= 1 && atoi(e) <= kBells) W.bells = atoi(e);
        if ((e = getenv("SPG_WORKER_LAZY")) && atoi(e) >= 0) W.lazy = atoi(e);
    }
    for (int c = 0; c < W.bells; c++) W.q->tail[c * kBellStride] = 0;     // through the BAR
    W.q->stop = 0;
    bar_fence();
    W.tail = 0; W.bytes = 0; W.blankets = 0;
    HIPCHK(hipMemsetAsync(W.ticket, 0, 64, W.stream));
    // LDS of the largest blanket a worker takes: n <= kWaveMax, one removed vertex, the two-wavefront carve-up
    Layout L = make_layout(D, 128, kWorkerMaxN / D, 1, alg, SPG_TOPO_TREE, 0);
    const size_t lds = (size_t)(L.small_doubles + L.mat_doubles) * 8;
    int rc = (D == 6) ? launch_worker<6, SPG_ALG_NFR>(this, lds) : launch_worker<3, SPG_ALG_NFR>(this, lds);
    if (rc) return rc;
    W.running = true; W.D = D; W.alg = alg;
    W.last_push = std::chrono::steady_clock::now();
    n_launches++;
    return 0;
}

int HipBackend::worker_stop() {
    Worker &W = worker;
    if (!W.running) return 0;
    W.q->stop = 1;
    bar_fence();
    HIPCHK(hipStreamSynchronize(W.stream));
    W.running = false;
    if (profiling) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, W.ev_a, W.ev_b) == hipSuccess) {
            prof_worker_ms += ms; prof_worker_bytes += W.bytes; prof_worker_runs++; prof_worker_blankets += W.blankets;
        }
    }
    return 0;
}

static void drain_profile(HipBackend *hb, HipBackend::Slot &S) {
    for (auto &t : S.pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            S.prof_ms += ms; S.prof_bytes += t.bytes; S.prof_launches++; S.prof_blankets += t.blankets;
        }
        S.pool.push_back({t.a, t.b});
    }
    S.pending.clear();
}

#ifdef SPG_LAUNCH_PROF
static double lp_t[6]; static long lp_n;
#define LP(i) do { auto n_ = std::chrono::steady_clock::now(); lp_t[i] += std::chrono::duration<double, std::micro>(n_ - lp0_).count(); lp0_ = n_; } while (0)
#define LP0 auto lp0_ = std::chrono::steady_clock::now(); lp_n++
#else
#define LP(i) do {} while (0)
#define LP0 do {} while (0)
#endif
static int hip_run_round(void *user, void *arena, const spg_round_desc *rd) {
    HipBackend *hb = (HipBackend *)user;
    LP0;
    char *err = hb->err;
    if (rd->count <= 0) return 0;
    HipBackend::Slot &S = hb->slots[rd->slot & (HipBackend::NSLOT - 1)];
    const spg_options &o = *rd->opts;
    const int D = o.pose_dim;
    if (D != 3 && D != 6) return SPG_EINVAL;
    // (the worker decision comes first: a batch that goes to the worker needs neither bins nor launch descriptors)
    {
        int cur = -1;   // (a thread-local read; hipSetDevice costs a microsecond per launch)
        if (hipGetDevice(&cur) != hipSuccess || cur != hb->device) HIPCHK(hipSetDevice(hb->device));
    }
    // the previous launch of this slot must have drained before its staging buffer is rewritten
    // (already the case when the host has just harvested the slot's late results)
    if (S.busy) if (int rcw = hb->wait_slot(S)) return rcw;
    drain_profile(hb, S);
    S.busy = true;
    LP(1);
    // ---- narrow batch: hand the blankets to the persistent worker instead of launching
    // While the worker runs nothing else is submitted to the device: HIP multiplexes streams onto a few hardware queues
    // and a dispatch queued behind the never-ending worker kernel would wait for it. So a batch goes to the worker
    // whole (every blanket eligible) or the worker is retired first and the batch is launched as before.
    static const bool worker_env = [] { const char *e = getenv("SPG_WORKER"); return !(e && e[0] == '0'); }();
    static const bool worker_stamp = [] { const char *e = getenv("SPG_WORKER_STAMP"); return e && e[0] == '1'; }();
    // (not for the first two batches after a synchronisation: a call that removes a handful of vertices, e.g. online
    //  decimation, is cheaper as a plain launch than as worker start + stop)
    hb->batches_in_call++;
    bool to_worker = worker_env && hb->large_bar && !hb->worker.disabled && rd->mail_len > 0 && o.algorithm == SPG_ALG_NFR && o.topology == SPG_TOPO_TREE &&
                     o.lin_point == SPG_LIN_GLOBAL && o.flags == 0 && rd->count <= 512 && !hb->force_one_wave &&
                     hb->worker_cooldown == 0 && (hb->batches_in_call > 2 || hb->worker.running);
    if (hb->worker_cooldown > 0) hb->worker_cooldown--;
    if (to_worker) {
        // eligible: pose-pose edges only, one removed vertex, n <= kWorkerMaxN, packet fits the staging area
        for (int b = rd->first; b < rd->first + rd->count && to_worker; b++) {
            const spg_blanket_desc &bd = rd->blankets[b];
            const int k = bd.n_vert - bd.n_remove;
            int nev = 0;
            bool bin = true;
            for (int e = bd.edge_begin; e < bd.edge_begin + bd.n_edge; e++) { nev += rd->edges[e].nv; bin &= (rd->edges[e].kind == SPG_EDGE_BINARY); }
            const int words = kPktHdr + bd.n_vert + 3 * bd.n_edge + (nev + 1) / 2;
            to_worker = bin && bd.n_remove == 1 && k >= 1 && D * k <= kWorkerMaxN && words <= kPktWords && bd.tinfo_off < 0;
        }
        if (!to_worker) hb->worker_cooldown = 8;   // mixed batches: stay with launches for a while rather than stop / start per batch
    }
    if (to_worker) {
        const size_t n_push = (size_t)rd->count;
        int wrc = 0;
        const size_t need = n_push * (size_t)kPktWords * 8, mneed = (size_t)rd->mail_len * 8;
        if (need > S.c_pkt || mneed > S.c_mail) {
            if (int rc2 = hb->worker_stop()) return rc2;   // hipFree waits for the device: nothing may be spinning on it
            if (need > S.c_pkt) {
                if (S.d_pkt) { HIPCHK(hipFree(S.d_pkt)); S.d_pkt = nullptr; S.c_pkt = 0; }
                size_t nc = std::max(need, (size_t)512 * kPktWords * 8);
                if (hipExtMallocWithFlags(&S.d_pkt, nc, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); hb->worker.disabled = true; }
                else S.c_pkt = nc;
            }
            if (mneed > S.c_mail) {
                if (S.h_mail) HIPCHK(hipHostFree(S.h_mail));
                size_t nc = std::max(mneed, S.c_mail * 2);
                HIPCHK(hipHostMalloc(&S.h_mail, nc, hipHostMallocMapped));
                HIPCHK(hipHostGetDevicePointer(&S.d_mail, S.h_mail, 0));
                memset(S.h_mail, 0, nc);
                S.c_mail = nc;
            }
        }
        if (!hb->worker.disabled) wrc = hb->worker_start(D, SPG_ALG_NFR);
        if (wrc < 0) return wrc;
        if (wrc == 0 && !hb->worker.disabled) {
            HipBackend::Worker &W = hb->worker;
            unsigned long long *pbase = (unsigned long long *)S.d_pkt;
            const double *hmail = (const double *)S.h_mail;
            S.final_word = SPG_FINAL_WORD(rd->tag);
            double wbytes = 0;
            for (size_t i = 0; i < n_push; i++) {
                const int32_t b = rd->first + (int32_t)i;
                const spg_blanket_desc &bd = rd->blankets[b];
                unsigned long long pkt[kPktWords];
                int nev = 0;
                for (int e = bd.edge_begin; e < bd.edge_begin + bd.n_edge; e++) nev += rd->edges[e].nv;
                const int words = kPktHdr + bd.n_vert + 3 * bd.n_edge + (nev + 1) / 2;
                auto pack = [](int lo, int hi) { return (unsigned long long)(uint32_t)lo | ((unsigned long long)(uint32_t)hi << 32); };
                pkt[0] = (unsigned long long)(uintptr_t)arena;
                pkt[1] = (unsigned long long)(uintptr_t)S.d_mail;
                pkt[2] = (unsigned long long)rd->mail_base;
                pkt[3] = (unsigned long long)bd.out_off; pkt[4] = (unsigned long long)bd.new_off; pkt[5] = (unsigned long long)bd.tinfo_off;
                pkt[6] = pack(bd.n_vert, bd.n_remove); pkt[7] = pack(bd.n_edge, bd.n_new_max); pkt[8] = pack(bd.n_new_vert_max, bd.pad_);
                pkt[9] = pack(o.topology, o.flags); pkt[10] = pack(o.lin_point, rd->tag);
                memcpy(&pkt[11], &o.chord_ratio, 8);
                pkt[12] = pack(words, nev | (worker_stamp ? 0x40000000 : 0));
                int w = kPktHdr;
                for (int v = 0; v < bd.n_vert; v++) pkt[w++] = (unsigned long long)rd->vert_pose_off[bd.vert_begin + v];
                int32_t *evp = (int32_t *)(pkt + kPktHdr + bd.n_vert + 3 * bd.n_edge);
                int evn = 0;
                double by = 8.0 * ((D == 6) ? 7 : 3) * bd.n_vert + 12.0 + 8.0 * bd.new_len;
                for (int e = bd.edge_begin; e < bd.edge_begin + bd.n_edge; e++) {
                    spg_edge_ref er = rd->edges[e];
                    for (int t = 0; t < er.nv; t++) evp[evn + t] = rd->edge_vert[er.vbegin + t];
                    er.vbegin = evn;
                    evn += er.nv;
                    memcpy(&pkt[w], &er, 24);
                    w += 3;
                    by += 4.0 * er.nv + 8.0 * er.len;
                }
                if (evn & 1) evp[evn] = 0;
                wbytes += by;
                unsigned long long *dst = pbase + i * (size_t)kPktWords;
                memcpy(dst, pkt, (size_t)words * 8);                       // through the BAR (write-combined)
                W.q->item[(W.tail + i) % kQCap] = (unsigned long long)(uintptr_t)dst;
                S.finals.push_back(hmail + (bd.out_off - rd->mail_base) + 5);
            }
            bar_fence();
            W.tail += n_push;
            for (int c = 0; c < W.bells; c++) W.q->tail[c * kBellStride] = W.tail;   // the doorbells
            bar_fence();
            W.last_push = std::chrono::steady_clock::now();
            W.bytes += wbytes; W.blankets += (long long)n_push;
            static const bool echo_test = [] { const char *e = getenv("SPG_WORKER_ECHO_TEST"); return e && e[0] == '1'; }();
            if (echo_test) {
                // diagnostic: doorbell -> ready word of the batch's first blanket, on the host clock
                const volatile double *rw = S.finals.front();
                const double want_r = SPG_READY_WORD(rd->tag), want_f = SPG_FINAL_WORD(rd->tag);
                auto t0 = std::chrono::steady_clock::now();
                while (*rw != want_r && *rw != want_f) { if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1.0) break; }
                static double acc = 0; static long cnt = 0;
                acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); cnt++;
                if (cnt % 252 == 0) { fprintf(stderr, "worker echo test: doorbell -> first blanket ready %.1f us (avg over %ld batches of ~%zu)\n", acc / cnt, cnt, n_push); acc = 0; cnt = 0; }
            }
            return 0;                                                      // the whole batch is with the worker
        }
    }
    if (int rcw = hb->worker_stop()) return rcw;                           // a launch follows: the worker must not be in its way
    // ---- bin this rank's blankets by the LDS their tiles need
    struct Bin { std::vector<int32_t> &list; int kmax = 0, mmax = 0, smax = 0; double bytes = 0; };
    const int NB = 5;
    const size_t lim[NB - 1] = {24 * 1024, 40 * 1024, 80 * 1024, (size_t)hb->lds_limit};
    for (auto &v : S.bin_lists) v.clear();   // per-slot scratch: no allocation per launch
    Bin bins[NB] = {{S.bin_lists[0]}, {S.bin_lists[1]}, {S.bin_lists[2]}, {S.bin_lists[3]}, {S.bin_lists[4]}};
    for (int b = rd->first; b < rd->first + rd->count; b++) {
        const spg_blanket_desc &bd = rd->blankets[b];
        int k = bd.n_vert - bd.n_remove, m = bd.n_remove;
        // (binned with the carve-up of the widest team any LDS variant uses, so that no variant outgrows its bin)
        Layout L = make_layout(D, 256, k, m, o.algorithm, o.topology, bd.pad_);
        size_t need = (size_t)(L.small_doubles + L.mat_doubles) * 8;
        int bi = NB - 1;
        for (int i = 0; i < NB - 1; i++) if (need <= lim[i]) { bi = i; break; }
        bins[bi].list.push_back(b);
        if (hb->profiling) {
            // algorithmic HBM bytes of this blanket: poses + (2 x i32 + record) per edge
            // + new records + (kld f64 + status i32)
            const int ps = (D == 6) ? 7 : 3;
            double by = 8.0 * ps * bd.n_vert + 12.0;
            for (int e = bd.edge_begin; e < bd.edge_begin + bd.n_edge; e++) by += 4.0 * rd->edges[e].nv + 8.0 * rd->edges[e].len;
            by += 8.0 * bd.new_len;
            bins[bi].bytes += by;
        }
        bins[bi].kmax = std::max(bins[bi].kmax, k);
        bins[bi].mmax = std::max(bins[bi].mmax, m);
        bins[bi].smax = std::max(bins[bi].smax, (int)bd.pad_);
    }
    // NFR blankets whose pattern (Dense / Subgraph with more than k-1 edges) has no closed form: interior point, its own
    // kernel (spg_nfr_ip.hip), one workgroup per blanket after the launches below
    std::vector<int32_t> ip_list;
    int ip_closed = 0;
    int64_t ip_stride = 0, ip_hot = 0;
    if (o.algorithm == SPG_ALG_NFR) {
        const bool cliquey = o.topology == SPG_TOPO_CLIQUEY_SUBGRAPH || o.topology == SPG_TOPO_CLIQUEY_DENSE;
        for (int i = 0; i < NB; i++) {
            size_t keep = 0;
            int kmax = 0, mmax = 0, smax = 0;
            for (int32_t b : bins[i].list) {
                const spg_blanket_desc &bd = rd->blankets[b];
                const int k = bd.n_vert - bd.n_remove, m = bd.n_remove;
                const int E = spg::nfr_ip_pattern_size(o.topology, o.chord_ratio, k);
                bool has_multi = false;
                for (int e = bd.edge_begin; e < bd.edge_begin + bd.n_edge; e++) has_multi |= rd->edges[e].kind == SPG_EDGE_MULTI;
                const bool ip = k >= 3 && !cliquey && E > k - 1;           // uncorrelated pattern without a closed form
                // correlated patterns (and any blanket that holds a correlated edge) take the same generic kernel's closed form
                // (clusters under the Local linearisation point too: the blanket kernel's own Local branch is for one removed vertex)
                const bool local_cluster = o.lin_point != SPG_LIN_GLOBAL && m > 1 && k >= 2;
                // (and blankets whose Chow-Liu pair tables and side buffers outgrow LDS even with the tiles in the L2 workspace —
                //  k beyond ~130: until round 3 SPG_ECAPACITY — the generic kernel keeps everything in its workspace, k <= 256)
                bool too_big = false;
                if (i == NB - 1 && k >= 2) {
                    const bool lm_ = o.lin_point != SPG_LIN_GLOBAL;
                    Layout Lw = make_layout(D, lm_ ? 256 : 1024, k, m, o.algorithm, o.topology, bd.pad_);
                    too_big = (size_t)Lw.small_doubles * 8 > (size_t)hb->lds_limit;
                }
                if (ip || (cliquey && k >= 3) || (has_multi && k >= 2) || local_cluster || too_big) {
                    const int msub = (int)((1 + o.chord_ratio) * (k - 1));
                    const bool masks = o.topology == SPG_TOPO_CLIQUEY_SUBGRAPH && msub < k * (k - 1) / 2;   // fillCliques on 64-bit vertex masks
                    // (interior point: Newton systems up to 2 048 variables in LDS-resident forms, up to spg::kIpMaxVars through the
                    //  blocked factorisation — one workgroup, 0.1 s per Newton step at 2 400 variables, 1 s at 4 900, 5 s at 8 300)
                    if ((ip && (int64_t)D * D * E > spg::kIpMaxVars) || (masks && k > 64) || k > 256) {
                        snprintf(err, sizeof hb->err, "interior-point / correlated NFR: a blanket with k=%d kept vertices and %d new measurements is beyond the generic kernel (Newton systems up to %d variables; k <= 64 for CliqueySubgraph, 256 otherwise)", k, E, spg::kIpMaxVars);
                        return SPG_ECAPACITY;
                    }
                    ip_list.push_back(b);
                    if (!ip) ip_closed++;        // (closed form: every correlated pattern, trees with correlated input edges)
                    int64_t hot = 0;
                    ip_stride = std::max(ip_stride, spg::nfr_ip_workspace(D, k, m, E, ip ? 0 : 1, &hot));
                    ip_hot = std::max(ip_hot, hot);
                    continue;
                }
                bins[i].list[keep++] = b;
                kmax = std::max(kmax, k); mmax = std::max(mmax, m); smax = std::max(smax, (int)bd.pad_);
            }
            if (keep != bins[i].list.size()) { bins[i].list.resize(keep); bins[i].kmax = kmax; bins[i].mmax = mmax; bins[i].smax = smax; }
        }
    }
    // Blankets whose side buffers (Chow-Liu pair tables, GLC batch buffers) exceed LDS even with the tiles in the L2
    // workspace: GLC Dense ones go through the dense HBM pipeline on the matrix cores (spg_dense.hip) after the
    // launches below, one at a time; for the others there is no path (SPG_ECAPACITY, as before).
    std::vector<int32_t> big_list;
    static const bool force_big = [] { const char *e = getenv("SPG_FORCE_BIG"); return e && e[0] == '1'; }();   // diagnostic / tests
    if (force_big && o.algorithm == SPG_ALG_GLC && o.topology == SPG_TOPO_DENSE && o.lin_point == SPG_LIN_GLOBAL) {
        // every blanket with at least one kept vertex takes the dense pipeline (parity of that path on small blankets)
        for (int i = 0; i < NB; i++) {
            size_t keep = 0;
            for (int32_t b : bins[i].list) {
                if (rd->blankets[b].n_vert - rd->blankets[b].n_remove >= 2 && rd->blankets[b].n_edge > 0) big_list.push_back(b);
                else bins[i].list[keep++] = b;
            }
            bins[i].list.resize(keep);
        }
        std::sort(big_list.begin(), big_list.end());
    } else {
        Bin &bb = bins[NB - 1];
        size_t keep = 0;
        int kmax = 0, mmax = 0, smax = 0;
        for (int32_t b : bb.list) {
            const spg_blanket_desc &bd = rd->blankets[b];
            const int k = bd.n_vert - bd.n_remove, m = bd.n_remove;
            Layout Lb = make_layout(D, 256, k, m, o.algorithm, o.topology, bd.pad_);
            if ((size_t)Lb.small_doubles * 8 <= (size_t)hb->lds_limit) {
                bb.list[keep++] = b;
                kmax = std::max(kmax, k); mmax = std::max(mmax, m); smax = std::max(smax, (int)bd.pad_);
                continue;
            }
            if (!(o.algorithm == SPG_ALG_GLC && o.topology == SPG_TOPO_DENSE && o.lin_point == SPG_LIN_GLOBAL)) {
                snprintf(err, sizeof hb->err, "blanket too large for LDS side buffers: k=%d m=%d (only GLC Dense blankets have a large-blanket path)", k, m);
                return SPG_ECAPACITY;
            }
            big_list.push_back(b);
        }
        if (!big_list.empty()) { bb.list.resize(keep); bb.kmax = kmax; bb.mmax = mmax; bb.smax = smax; }
    }
    LP(0);
    // ---- upload the round's descriptors (one pinned staging buffer, async copies)
    size_t s_blk = sizeof(spg_blanket_desc) * (size_t)rd->n_blankets;
    size_t s_vpo = sizeof(int64_t) * (size_t)rd->n_vert_total;
    size_t s_er = sizeof(spg_edge_ref) * (size_t)rd->n_edge_total;
    size_t s_ev = sizeof(int32_t) * (size_t)rd->n_edge_vert_total;
    size_t s_list = sizeof(int32_t) * (size_t)rd->count;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_blk = 0, o_vpo = o_blk + al(s_blk), o_er = o_vpo + al(s_vpo), o_ev = o_er + al(s_er), o_list = o_ev + al(s_ev);
    size_t tot = o_list + al(s_list);
    // Where the descriptors of this launch go:
    //  - small launch, large-BAR system: the host stores them straight into (fine-grained) device memory —
    //    posted writes ahead of the doorbell, no copy engine hop, and the kernel reads local HBM;
    //  - small launch otherwise: the kernel reads them from the mapped pinned staging buffer over PCIe;
    //  - large launch: one host->device copy from the staging buffer.
    static const int mapped_limit = [] { const char *e = getenv("SPG_MAPPED_DESC"); return e ? atoi(e) : 512; }();
    static const bool bar_ok = [] { const char *e = getenv("SPG_BAR_DESC"); return !(e && e[0] == '0'); }();
    const bool small = (long long)rd->count <= (long long)mapped_limit;
    bool via_bar = small && hb->large_bar && bar_ok;
    char *st;
    if (via_bar && tot > S.c_bar) {
        if (S.d_bar) { if (int rc = hb->worker_stop()) return rc; HIPCHK(hipFree(S.d_bar)); S.d_bar = nullptr; S.c_bar = 0; }
        size_t nc = std::max(tot, (size_t)1 << 16);
        if (hipExtMallocWithFlags(&S.d_bar, nc, hipDeviceMallocFinegrained) == hipSuccess) S.c_bar = nc;
        else { (void)hipGetLastError(); S.d_bar = nullptr; hb->large_bar = false; via_bar = false; }   // fall back to the mapped staging buffer
    }
    if (via_bar) {
        st = (char *)S.d_bar;   // write-only from the host
    } else {
        if (int rc = hb->ensure_stage(S, tot)) return rc;
        if (!small) if (int rc = hb->ensure(S, &S.d_desc, &S.c_desc, tot)) return rc;
        st = (char *)S.h_stage;
    }
    memcpy(st + o_blk, rd->blankets, s_blk);
    memcpy(st + o_vpo, rd->vert_pose_off, s_vpo);
    memcpy(st + o_er, rd->edges, s_er);
    if (s_ev) memcpy(st + o_ev, rd->edge_vert, s_ev);
    {
        int32_t *lst = (int32_t *)(st + o_list);
        size_t p = 0;
        for (int i = 0; i < NB; i++) for (int32_t b : bins[i].list) lst[p++] = b;
        for (int32_t b : ip_list) lst[p++] = b;      // (the lists together never exceed rd->count entries)
    }
    char *desc_base;
    if (via_bar) {
        std::atomic_thread_fence(std::memory_order_release);
#if defined(__x86_64__)
        __builtin_ia32_sfence();   // write-combining buffers drained before the launch rings the doorbell
#endif
        desc_base = (char *)S.d_bar;
    } else if (small) {
        desc_base = (char *)S.d_stage;
    } else {
        desc_base = (char *)S.d_desc;
        HIPCHK(hipMemcpyAsync(S.d_desc, st, tot, hipMemcpyHostToDevice, S.stream));
    }
    LP(2);
    double *mail_dev = nullptr;
    if (rd->mail_len > 0) {
        size_t need = (size_t)rd->mail_len * 8;
        if (need > S.c_mail) {
            if (int rc = hb->worker_stop()) return rc;
            if (S.h_mail) HIPCHK(hipHostFree(S.h_mail));
            size_t nc = std::max(need, S.c_mail * 2);
            HIPCHK(hipHostMalloc(&S.h_mail, nc, hipHostMallocMapped));
            HIPCHK(hipHostGetDevicePointer(&S.d_mail, S.h_mail, 0));
            memset(S.h_mail, 0, nc);   // a fresh mailbox never looks ready
            S.c_mail = nc;
        }
        mail_dev = (double *)S.d_mail;
    }
    // ---- launch each non-empty bin
    KArgs ka;
    ka.arena = (double *)arena;
    ka.blk = (const spg_blanket_desc *)(desc_base + o_blk);
    ka.vpo = (const int64_t *)(desc_base + o_vpo);
    ka.er = (const spg_edge_ref *)(desc_base + o_er);
    ka.ev = (const int32_t *)(desc_base + o_ev);
    ka.mail = mail_dev;
    ka.mail_base = rd->mail_base;
    ka.gws = nullptr; ka.gws_stride = 0;
    ka.topology = o.topology; ka.algorithm = o.algorithm; ka.flags = o.flags; ka.chord_ratio = o.chord_ratio; ka.lin_point = o.lin_point; ka.tag = rd->tag;
    // Local linearisation point: the kernel variant that carries the blanket-level LM
    const bool lm = (o.algorithm == SPG_ALG_NFR) && (o.lin_point != SPG_LIN_GLOBAL);
    size_t list_off = 0;
    for (int i = 0; i < NB; i++) {
        int nb = (int)bins[i].list.size();
        if (nb == 0) continue;
        ka.list = (const int32_t *)(desc_base + o_list) + list_off;
        list_off += nb;
        int rc;
        if (i < NB - 1) {
            Layout L = make_layout(D, 64, bins[i].kmax, bins[i].mmax, o.algorithm, o.topology, bins[i].smax);
            size_t lds = (size_t)(L.small_doubles + L.mat_doubles) * 8;
            // the (kmax, mmax, smax) envelope can exceed the device limit although every member fits
            // (each needs <= lim[i] <= lds_limit): clamp, the per-block carve-up uses its own k, m
            if (lds > (size_t)hb->lds_limit) lds = (size_t)hb->lds_limit;
            // latency mode: a launch that cannot fill the chip (<= 2 blankets per CU) gives every blanket
            // two wavefronts so the Chow-Liu and gauge chains overlap; throughput mode keeps one
            const bool two_waves = (o.algorithm == SPG_ALG_NFR) && nb <= 512 && D * bins[i].kmax <= kWaveMax && !hb->force_one_wave;
            if (two_waves) {
                Layout L2 = make_layout(D, 128, bins[i].kmax, bins[i].mmax, o.algorithm, o.topology, bins[i].smax);
                size_t lds2 = std::min((size_t)(L2.small_doubles + L2.mat_doubles) * 8, (size_t)hb->lds_limit);
                if (lm) rc = (D == 6) ? launch_bin<6, 128, false, SPG_ALG_NFR_LM>(hb, S, ka, nb, lds2, bins[i].bytes) : launch_bin<3, 128, false, SPG_ALG_NFR_LM>(hb, S, ka, nb, lds2, bins[i].bytes);
                else rc = (D == 6) ? launch_bin<6, 128, false, SPG_ALG_NFR>(hb, S, ka, nb, lds2, bins[i].bytes) : launch_bin<3, 128, false, SPG_ALG_NFR>(hb, S, ka, nb, lds2, bins[i].bytes);
            } else if ((o.algorithm == SPG_ALG_NFR) && !lm && nb <= 1024 && D * bins[i].kmax > kWaveMax && !hb->force_one_wave) {
                // tiles beyond the register-resident size (n > 24) run the LDS-cooperative routines: their O(n^2) inner
                // steps want lanes, and a launch this small leaves the chip empty anyway — four wavefronts per blanket
                // (parking.g2o: 0.70 -> ms per launch of such blankets)
                Layout L4 = make_layout(D, 256, bins[i].kmax, bins[i].mmax, o.algorithm, o.topology, bins[i].smax);
                size_t lds4 = std::min((size_t)(L4.small_doubles + L4.mat_doubles) * 8, (size_t)hb->lds_limit);
                rc = (D == 6) ? launch_bin<6, 256, false, SPG_ALG_NFR>(hb, S, ka, nb, lds4, bins[i].bytes) : launch_bin<3, 256, false, SPG_ALG_NFR>(hb, S, ka, nb, lds4, bins[i].bytes);
            } else if (o.algorithm == SPG_ALG_GLC)
                rc = (D == 6) ? launch_bin<6, 64, false, SPG_ALG_GLC>(hb, S, ka, nb, lds, bins[i].bytes) : launch_bin<3, 64, false, SPG_ALG_GLC>(hb, S, ka, nb, lds, bins[i].bytes);
            else if (lm)
                rc = (D == 6) ? launch_bin<6, 64, false, SPG_ALG_NFR_LM>(hb, S, ka, nb, lds, bins[i].bytes) : launch_bin<3, 64, false, SPG_ALG_NFR_LM>(hb, S, ka, nb, lds, bins[i].bytes);
            else
                rc = (D == 6) ? launch_bin<6, 64, false, SPG_ALG_NFR>(hb, S, ka, nb, lds, bins[i].bytes) : launch_bin<3, 64, false, SPG_ALG_NFR>(hb, S, ka, nb, lds, bins[i].bytes);
        } else {
            // NFR: eight wavefronts per blanket — the tiles sit in L2, every step of the cooperative routines is a round of
            // ~1 us accesses, and only lanes hide that (parking.g2o: 3.1 -> ms per launch of such blankets)
            const bool wide = (o.algorithm == SPG_ALG_NFR) && !lm && !hb->force_one_wave;
            const int ntg = wide ? 1024 : 256;
            Layout L = make_layout(D, ntg, bins[i].kmax, bins[i].mmax, o.algorithm, o.topology, bins[i].smax);
            size_t lds = (size_t)L.small_doubles * 8;
            if (lds > (size_t)hb->lds_limit) { snprintf(err, sizeof hb->err, "blanket too large for LDS side buffers: k=%d m=%d", bins[i].kmax, bins[i].mmax); return SPG_ECAPACITY; }
            size_t stride = ((size_t)L.mat_doubles + 31) & ~(size_t)31;
            if (int rc2 = hb->ensure(S, &S.d_gws, &S.c_gws, stride * 8 * (size_t)nb)) return rc2;
            ka.gws = (double *)S.d_gws;
            ka.gws_stride = (int64_t)stride;
            if (o.algorithm == SPG_ALG_GLC)
                rc = (D == 6) ? launch_bin<6, 256, true, SPG_ALG_GLC>(hb, S, ka, nb, lds, bins[i].bytes) : launch_bin<3, 256, true, SPG_ALG_GLC>(hb, S, ka, nb, lds, bins[i].bytes);
            else if (lm)
                rc = (D == 6) ? launch_bin<6, 256, true, SPG_ALG_NFR_LM>(hb, S, ka, nb, lds, bins[i].bytes) : launch_bin<3, 256, true, SPG_ALG_NFR_LM>(hb, S, ka, nb, lds, bins[i].bytes);
            else if (wide)
                rc = (D == 6) ? launch_bin<6, 1024, true, SPG_ALG_NFR>(hb, S, ka, nb, lds, bins[i].bytes) : launch_bin<3, 1024, true, SPG_ALG_NFR>(hb, S, ka, nb, lds, bins[i].bytes);
            else
                rc = (D == 6) ? launch_bin<6, 256, true, SPG_ALG_NFR>(hb, S, ka, nb, lds, bins[i].bytes) : launch_bin<3, 256, true, SPG_ALG_NFR>(hb, S, ka, nb, lds, bins[i].bytes);
        }
        if (rc) return rc;
    }
    // ---- interior-point NFR blankets
    if (!ip_list.empty()) {
        if (int rc2 = hb->ensure(S, &S.d_ipws, &S.c_ipws, (size_t)ip_stride * 8 * ip_list.size())) return rc2;
        // The workspace is reused from launch to launch and from graph to graph: cleared, so that what a blanket finds in its
        // slice never depends on what ran before it in the process (a run of several graphs through one context died with a
        // core dump in round 2 where each graph on its own passes; hipMalloc'ed memory is not zeroed either).
        HIPCHK(hipMemsetAsync(S.d_ipws, 0, (size_t)ip_stride * 8 * ip_list.size(), S.stream));
        spg::IpArgs ia{};
        ia.arena = (double *)arena; ia.blk = ka.blk; ia.vpo = ka.vpo; ia.er = ka.er; ia.ev = ka.ev;
        ia.list = (const int32_t *)(desc_base + o_list) + list_off;
        ia.ws = (double *)S.d_ipws; ia.ws_stride = ip_stride; ia.mail = mail_dev; ia.mail_base = rd->mail_base;
        ia.topology = o.topology; ia.lin_point = o.lin_point; ia.tag = rd->tag; ia.chord_ratio = o.chord_ratio;
        if (o.lin_point != SPG_LIN_GLOBAL) {
            // ---- Local linearisation point for these blankets (pre-pass, see local_point_kernel): scratch poses, closed-form
            // re-initialisation on the device, else the reference's 10 LM iterations with the first removed vertex fixed —
            // the dense LM of optimize() on the blanket as a small graph, host-driven, one blanket after the other — and a
            // second table of pose offsets that points the generic kernel at the scratch blocks.
            const int PSl = (D == 6) ? 7 : 3;
            const size_t nl = ip_list.size();
            std::vector<int64_t> lp_off(nl);
            int64_t lp_tot = 0;
            for (size_t i = 0; i < nl; i++) { lp_off[i] = lp_tot; lp_tot += (int64_t)rd->blankets[ip_list[i]].n_vert * PSl; }
            const size_t meta_bytes = nl * 8 + nl * 4 + (size_t)rd->n_vert_total * 8 + 64;
            if (int rc2 = hb->ensure(S, &S.d_lpose, &S.c_lpose, (size_t)lp_tot * 8 + 64)) return rc2;
            if (int rc2 = hb->ensure(S, &S.d_lmeta, &S.c_lmeta, meta_bytes)) return rc2;
            int64_t *d_lp_off = (int64_t *)S.d_lmeta, *d_vpo2 = d_lp_off + nl;
            int32_t *d_flag = (int32_t *)(d_vpo2 + rd->n_vert_total);
            const int64_t scratch0 = ((intptr_t)S.d_lpose - (intptr_t)arena) / 8;   // the scratch block as an "arena offset"
            std::vector<int64_t> vpo2(rd->vert_pose_off, rd->vert_pose_off + rd->n_vert_total);
            for (size_t i = 0; i < nl; i++) {
                const spg_blanket_desc &bd = rd->blankets[ip_list[i]];
                for (int v = 0; v < bd.n_vert; v++) vpo2[(size_t)bd.vert_begin + v] = scratch0 + lp_off[i] + (int64_t)v * PSl;
            }
            HIPCHK(hipMemcpyAsync(d_lp_off, lp_off.data(), nl * 8, hipMemcpyHostToDevice, S.stream));
            HIPCHK(hipMemcpyAsync(d_vpo2, vpo2.data(), (size_t)rd->n_vert_total * 8, hipMemcpyHostToDevice, S.stream));
            if (D == 6) hipLaunchKernelGGL((local_point_kernel<6>), dim3((unsigned)nl), dim3(64), 0, S.stream, (const double *)arena, (double *)S.d_lpose, ka.blk, ka.vpo, ka.er, ka.ev, ia.list, (const int64_t *)d_lp_off, d_flag);
            else hipLaunchKernelGGL((local_point_kernel<3>), dim3((unsigned)nl), dim3(64), 0, S.stream, (const double *)arena, (double *)S.d_lpose, ka.blk, ka.vpo, ka.er, ka.ev, ia.list, (const int64_t *)d_lp_off, d_flag);
            HIPCHK(hipGetLastError());
            std::vector<int32_t> h_flag(nl);
            HIPCHK(hipMemcpyAsync(h_flag.data(), d_flag, nl * 4, hipMemcpyDeviceToHost, S.stream));
            HIPCHK(hipStreamSynchronize(S.stream));
            for (size_t i = 0; i < nl; i++) {
                if (h_flag[i]) continue;
                const spg_blanket_desc &bd = rd->blankets[ip_list[i]];
                // the blanket as a small graph (local vertex = blanket-local index), vertex 0 fixed
                std::vector<int32_t> pos(bd.n_vert), rowptr(bd.n_vert + 1, 0), inc;
                std::vector<int64_t> lvpo(bd.n_vert);
                for (int l = 0; l < bd.n_vert; l++) { pos[l] = l == 0 ? -1 : (l - 1) * D; lvpo[l] = scratch0 + lp_off[i] + (int64_t)l * PSl; }
                std::vector<std::vector<int32_t>> per(bd.n_vert);
                for (int e = 0; e < bd.n_edge; e++) {
                    const spg_edge_ref &r = rd->edges[bd.edge_begin + e];
                    for (int t = 0; t < r.nv; t++) {
                        const int32_t l = rd->edge_vert[r.vbegin + t];
                        if (per[l].empty() || per[l].back() != e) per[l].push_back(e);
                    }
                }
                for (int l = 0; l < bd.n_vert; l++) { inc.insert(inc.end(), per[l].begin(), per[l].end()); rowptr[l + 1] = (int32_t)inc.size(); }
                spg::DenseGraphIn in;
                in.D = D; in.nv = bd.n_vert; in.ne = bd.n_edge;
                in.pos = pos.data(); in.vpo = lvpo.data(); in.rowptr = rowptr.data(); in.inc = inc.data();
                in.er = rd->edges + bd.edge_begin; in.ev = rd->edge_vert; in.n_ev = rd->n_edge_vert_total; in.dev_arena = arena;
                double lm_stats[8] = {0}, lm_secs = 0;
                if (int lrc = spg::hip_dense_optimize((void *)S.stream, in, D * (bd.n_vert - 1), 10, lm_stats, &lm_secs, hb->err, sizeof hb->err)) return lrc;
            }
            ia.vpo = (const int64_t *)d_vpo2;
            ia.lin_point = SPG_LIN_GLOBAL;   // the scratch poses ARE the local linearisation point: taken as they are
        }
        if (int rc2 = spg::hip_nfr_ip_launch((void *)S.stream, D, ia, (int)ip_list.size(), ip_closed, ip_hot)) { snprintf(err, sizeof hb->err, "launch of the interior-point kernel failed"); return rc2; }
        // (this kernel comes AFTER the event a bin launch left in wait_ev: waiting for the slot must mean the whole stream.
        //  Until round 3 wait_slot returned when the bin kernel was done — with a cluster of 150 vertices still running in
        //  this one, the commit read whatever the mailbox held at its record's place: silently wrong graphs on parking.g2o
        //  under CliqueyDense, and a segmentation fault when the stale words were not zeros.)
        S.wait_ev = nullptr;
        S.stream_dirty = true;
        S.busy = true;
    }
    // ---- large GLC Dense blankets: dense in HBM, O(n^3) parts on the fp64 matrix cores
    for (int32_t b : big_list) {
        const spg_blanket_desc &bd = rd->blankets[b];
        const int k = bd.n_vert - bd.n_remove, m = bd.n_remove;
        // the blanket as a local graph: vertices = blanket-local indices (removed first), edges as staged for the kernel
        const int nm = D * m, Nm = (nm + 63) / 64 * 64;
        std::vector<int32_t> pos(bd.n_vert), rowptr(bd.n_vert + 1, 0), inc;
        for (int l = 0; l < bd.n_vert; l++) pos[l] = l < m ? l * D : Nm + (l - m) * D;
        std::vector<std::vector<int32_t>> per(bd.n_vert);
        for (int e = 0; e < bd.n_edge; e++) {
            const spg_edge_ref &er = rd->edges[bd.edge_begin + e];
            for (int t = 0; t < er.nv; t++) {
                int32_t l = rd->edge_vert[er.vbegin + t];
                if (per[l].empty() || per[l].back() != e) per[l].push_back(e);
            }
        }
        for (int l = 0; l < bd.n_vert; l++) { inc.insert(inc.end(), per[l].begin(), per[l].end()); rowptr[l + 1] = (int32_t)inc.size(); }
        spg::DenseGraphIn in;
        in.D = D; in.nv = bd.n_vert; in.ne = bd.n_edge;
        in.pos = pos.data(); in.vpo = rd->vert_pose_off + bd.vert_begin; in.rowptr = rowptr.data(); in.inc = inc.data();
        in.er = rd->edges + bd.edge_begin; in.ev = rd->edge_vert; in.n_ev = rd->n_edge_vert_total; in.dev_arena = arena;
        double *orec = mail_dev ? (mail_dev + (bd.out_off - rd->mail_base)) : ((double *)arena + bd.out_off);
        double secs = 0, flops = 0;
        int brc = spg::hip_big_glc_dense((void *)S.stream, in, m, k, Nm, bd.new_off, orec, bd.n_new_max, rd->tag, &secs, &flops, hb->err, sizeof hb->err);
        if (brc) return brc;
        hb->prof_big_ms += 1e3 * secs; hb->prof_big_flops += flops; hb->prof_big_count++;
        hb->prof_big_nmax = std::max(hb->prof_big_nmax, D * (k + m));
        S.wait_ev = nullptr;   // (as above: the slot is done when the stream is)
        S.stream_dirty = true;
    }
    LP(3);
#ifdef SPG_LAUNCH_PROF
    if (lp_n % 1005 == 0) fprintf(stderr, "launch prof: bins %.2f sync+drain %.2f desc write %.2f launch %.2f us (avg over %ld)\n",
                                  lp_t[0] / lp_n, lp_t[1] / lp_n, lp_t[2] / lp_n, lp_t[3] / lp_n, lp_n);
#endif
    return 0;
}

static void *hip_alloc(void *user, int64_t doubles) {
    HipBackend *hb = (HipBackend *)user;
    void *p = nullptr;
    if (hipSetDevice(hb->device) != hipSuccess) return nullptr;
    if (hipMalloc(&p, (size_t)doubles * 8) != hipSuccess) return nullptr;
    return p;
}
static void hip_release(void *user, void *p) {
    HipBackend *hb = (HipBackend *)user;
    (void)hb->worker_stop();   // hipFree waits for the whole device
    (void)hipStreamSynchronize(hb->slots[0].stream);
    (void)hipFree(p);
}
static int hip_upload(void *user, void *dst, const double *src, int64_t doubles) {
    HipBackend *hb = (HipBackend *)user;
    char *err = hb->err;
    if (int rcw = hb->worker_stop()) return rcw;
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)doubles * 8, hipMemcpyHostToDevice, hb->slots[0].stream));
    HIPCHK(hipStreamSynchronize(hb->slots[0].stream));
    return 0;
}
static int hip_download(void *user, double *dst, const void *src, int64_t doubles) {
    HipBackend *hb = (HipBackend *)user;
    char *err = hb->err;
    if (int rcw = hb->worker_stop()) return rcw;
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)doubles * 8, hipMemcpyDeviceToHost, hb->slots[0].stream));
    HIPCHK(hipStreamSynchronize(hb->slots[0].stream));
    return 0;
}
static int hip_sync(void *user) {
    HipBackend *hb = (HipBackend *)user;
    char *err = hb->err;
    for (auto &S : hb->slots) {
        if (S.busy || S.wait_ev || !S.finals.empty()) { if (int rcw = hb->wait_slot(S)) return rcw; }
        else HIPCHK(hipStreamSynchronize(S.stream));   // copies issued outside hip_run_round
        drain_profile(hb, S);
    }
    // a full synchronisation leaves the device idle: the persistent worker retires (it restarts on demand)
    hb->batches_in_call = 0;
    return hb->worker_stop();
}
static int hip_sync_slot(void *user, int slot) {
    HipBackend *hb = (HipBackend *)user;
    char *err = hb->err;
    HipBackend::Slot &S = hb->slots[slot & (HipBackend::NSLOT - 1)];
    if (int rcw = hb->wait_slot(S)) return rcw;
    drain_profile(hb, S);
    return 0;
}

static const double *hip_mailbox(void *user) { return (const double *)((HipBackend *)user)->slots[0].h_mail; }
static const double *hip_mailbox_slot(void *user, int slot) { return (const double *)((HipBackend *)user)->slots[slot & (HipBackend::NSLOT - 1)].h_mail; }

int hip_backend_create(int device, spg_backend *out, char *errbuf, size_t errlen) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        snprintf(errbuf, errlen, "no HIP device available (count=%d, requested=%d): %s — libspg_hip has no CPU fallback",
                 ndev, device, e == hipSuccess ? "ok" : hipGetErrorString(e));
        return SPG_ENODEV;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { snprintf(errbuf, errlen, "hipGetDeviceProperties failed"); return SPG_ENODEV; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(errbuf, errlen, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
        return SPG_ENODEV;
    }
    HipBackend *hb = new HipBackend;
    hb->device = device;
    bool okc = hipSetDevice(device) == hipSuccess;
    for (auto &S : hb->slots) okc = okc && hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking) == hipSuccess &&
                                          hipEventCreateWithFlags(&S.done, hipEventDisableTiming) == hipSuccess;
    if (!okc) {
        snprintf(errbuf, errlen, "cannot create HIP stream on device %d", device);
        delete hb;
        return SPG_EHIP;
    }
    { const char *e1 = getenv("SPG_ONE_WAVE"); hb->force_one_wave = e1 && e1[0] == '1'; }
    hb->lds_limit = (int)prop.sharedMemPerBlock > 0 ? (int)std::min<size_t>(prop.sharedMemPerBlock, 160 * 1024) : 64 * 1024;
    {
        int lb = 0;
        hb->large_bar = hipDeviceGetAttribute(&lb, hipDeviceAttributeIsLargeBar, device) == hipSuccess && lb != 0;
    }
    out->user = hb;
    out->alloc = hip_alloc;
    out->release = hip_release;
    out->upload = hip_upload;
    out->download = hip_download;
    out->run_round = hip_run_round;
    out->synchronize = hip_sync;
    out->mailbox = hip_mailbox;
    out->synchronize_slot = hip_sync_slot;
    out->mailbox_slot = hip_mailbox_slot;
    return 0;
}

void hip_backend_destroy(spg_backend *b) {
    HipBackend *hb = (HipBackend *)b->user;
    if (!hb) return;
    (void)hipSetDevice(hb->device);
    (void)hb->worker_stop();
    if (hb->worker.q) (void)hipFree(hb->worker.q);
    if (hb->worker.ticket) (void)hipFree(hb->worker.ticket);
    if (hb->worker.ev_a) (void)hipEventDestroy(hb->worker.ev_a);
    if (hb->worker.ev_b) (void)hipEventDestroy(hb->worker.ev_b);
    if (hb->worker.stream) (void)hipStreamDestroy(hb->worker.stream);
    if (hb->st_pkt) (void)hipFree(hb->st_pkt);
    if (hb->st_hmail) (void)hipHostFree(hb->st_hmail);
    for (auto &S : hb->slots) {
        if (S.d_pkt) (void)hipFree(S.d_pkt);
        (void)hipStreamSynchronize(S.stream);
        if (S.d_desc) (void)hipFree(S.d_desc);
        if (S.d_gws) (void)hipFree(S.d_gws);
        if (S.d_lpose) (void)hipFree(S.d_lpose);
        if (S.d_lmeta) (void)hipFree(S.d_lmeta);
        if (S.h_mail) (void)hipHostFree(S.h_mail);
        if (S.h_stage) (void)hipHostFree(S.h_stage);
        if (S.d_bar) (void)hipFree(S.d_bar);
        if (S.done) (void)hipEventDestroy(S.done);
        for (auto &t : S.pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
        for (auto &pr : S.pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        (void)hipStreamDestroy(S.stream);
    }
    spg::hip_big_release_scratch();
    delete hb;
    b->user = nullptr;
}

void *hip_backend_stream(spg_backend *b) { return b->user ? (void *)((HipBackend *)b->user)->slots[0].stream : nullptr; }
const char *hip_backend_error(spg_backend *b) { return b->user ? ((HipBackend *)b->user)->err : ""; }
int hip_backend_device(spg_backend *b) { return b->user ? ((HipBackend *)b->user)->device : -1; }
int hip_backend_launches(spg_backend *b) { return b->user ? ((HipBackend *)b->user)->n_launches.load() : 0; }
void hip_backend_profile(spg_backend *b, int enable) {
    HipBackend *hb = (HipBackend *)b->user;
    if (!hb) return;
    hb->profiling = enable != 0;
    hb->prof_stride = enable > 1 ? enable : 1;   // enable = n > 1: HIP events around every n-th launch only
    hb->prof_tick = 0;
    for (auto &S : hb->slots) { S.prof_ms = S.prof_bytes = 0; S.prof_launches = S.prof_blankets = 0; }
    hb->prof_worker_ms = hb->prof_worker_bytes = 0; hb->prof_worker_runs = hb->prof_worker_blankets = 0;
    hb->prof_big_ms = hb->prof_big_flops = 0; hb->prof_big_count = 0; hb->prof_big_nmax = 0;
}
void hip_backend_profile_read_worker(spg_backend *b, double *ms, double *bytes, long long *runs, long long *blankets) {
    HipBackend *hb = (HipBackend *)b->user;
    if (!hb) return;
    *ms = hb->prof_worker_ms; *bytes = hb->prof_worker_bytes; *runs = hb->prof_worker_runs; *blankets = hb->prof_worker_blankets;
}
void hip_backend_profile_read_big(spg_backend *b, double *ms, double *flops, long long *count, int *nmax) {
    HipBackend *hb = (HipBackend *)b->user;
    if (!hb) return;
    *ms = hb->prof_big_ms; *flops = hb->prof_big_flops; *count = hb->prof_big_count; *nmax = hb->prof_big_nmax;
}
int hip_stream_open(spg_backend *b, int D, int slots, int mail_stride, StreamPort *out) {
    HipBackend *hb = (HipBackend *)b->user;
    if (!hb || !out || slots < 1 || slots > kQCap / 2 || (D != 3 && D != 6)) return SPG_EINVAL;
    char *err = hb->err;
    static const bool worker_env = [] { const char *e = getenv("SPG_WORKER"); return !(e && e[0] == '0'); }();
    if (!worker_env || !hb->large_bar || hb->worker.disabled || hb->force_one_wave) return 1;
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != hb->device) HIPCHK(hipSetDevice(hb->device));
    }
    // nothing of an earlier batch may still be running on the launch slots (their kernels would queue behind the worker)
    for (auto &S : hb->slots) if (S.busy || S.wait_ev || !S.finals.empty()) { if (int rcw = hb->wait_slot(S)) return rcw; drain_profile(hb, S); }
    const size_t need_pkt = (size_t)slots * kPktWords * 8, need_mail = (size_t)slots * (size_t)mail_stride * 8;
    if (need_pkt > hb->c_st_pkt || need_mail > hb->c_st_mail) {
        if (int rc2 = hb->worker_stop()) return rc2;   // hipFree waits for the device: nothing may be spinning on it
        if (need_pkt > hb->c_st_pkt) {
            if (hb->st_pkt) { HIPCHK(hipFree(hb->st_pkt)); hb->st_pkt = nullptr; hb->c_st_pkt = 0; }
            if (hipExtMallocWithFlags(&hb->st_pkt, need_pkt, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); hb->st_pkt = nullptr; return 1; }
            hb->c_st_pkt = need_pkt;
        }
        if (need_mail > hb->c_st_mail) {
            if (hb->st_hmail) { HIPCHK(hipHostFree(hb->st_hmail)); hb->st_hmail = nullptr; hb->c_st_mail = 0; }
            HIPCHK(hipHostMalloc(&hb->st_hmail, need_mail, hipHostMallocMapped));
            HIPCHK(hipHostGetDevicePointer(&hb->st_dmail, hb->st_hmail, 0));
            memset(hb->st_hmail, 0, need_mail);   // a fresh mailbox never looks ready
            hb->c_st_mail = need_mail;
        }
    }
    int wrc = hb->worker_start(D, SPG_ALG_NFR);
    if (wrc) return wrc;
    if (hb->worker.disabled) return 1;
    hb->batches_in_call += 3;   // (batches that follow in this call may go to the running worker right away)
    out->pkt = (unsigned long long *)hb->st_pkt;
    out->q = hb->worker.q;
    out->tail = hb->worker.tail;
    out->bells = hb->worker.bells;
    out->h_mail = (const double *)hb->st_hmail;
    out->d_mail = (unsigned long long)(uintptr_t)hb->st_dmail;
    out->mail_stride = mail_stride;
    out->slots = slots;
    return 0;
}

void hip_stream_close(spg_backend *b, const StreamPort *port, double alg_bytes, long long blankets) {
    HipBackend *hb = (HipBackend *)b->user;
    if (!hb || !port) return;
    hb->worker.tail = port->tail;
    hb->worker.bytes += alg_bytes;
    hb->worker.blankets += blankets;
    hb->worker.last_push = std::chrono::steady_clock::now();
}

int hip_backend_end_of_call(spg_backend *b) {
    HipBackend *hb = (HipBackend *)b->user;
    if (!hb) return 0;
    hb->batches_in_call = 0;
    return hb->worker_stop();
}
void hip_backend_profile_read(spg_backend *b, double *ms, double *bytes, long long *launches, long long *blankets) {
    HipBackend *hb = (HipBackend *)b->user;
    if (!hb) return;
    *ms = *bytes = 0; *launches = *blankets = 0;
    for (auto &S : hb->slots) { *ms += S.prof_ms; *bytes += S.prof_bytes; *launches += S.prof_launches; *blankets += S.prof_blankets; }
}

}  // namespace spg
