// csrc/spg_host.cpp — host side of libspg_hip.so: contexts, the device-resident pose graph, the
// conflict-free round scheduler and the C ABI of include/spg.h.
//
// What stays on the host is the integer graph work of the reference's loop:
//   markovBlanketVertices / extendedMarkovBlanketVertices / markovBlanketEdges
//                                             src/vertex_remover.cpp:142-251
//   buildSubgraph's vertex ordering           src/vertex_remover.cpp:349-356
//   updateInputGraph                          src/vertex_remover.cpp:500-546
//   GraphWrapperG2O bookkeeping               src/graph_wrapper_g2o.cpp:207-247,398-453
// Everything numeric runs in the HIP backend (spg_kernels.hip) on records that live in one HBM
// arena: [poses | edge records | per-round output regions]. The host keeps topology only.
//
// Sequential semantics. VertexRemover::remove mutates the graph after every vertex
// (src/vertex_remover.cpp:134). Two removals commute exactly when neither centre lies in the other's
// blanket and the blankets share at most one vertex (then no existing or future edge can belong to
// both). Each round scans the pending list in the reference's order and selects a vertex only if it
// commutes with every earlier vertex that is selected in this round or still deferred — a deferred
// vertex is represented by a superset D(u) of every vertex its blanket can reach before its turn.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif
#include <unistd.h>
#include <unordered_map>
#include <vector>
#include "../../include/spg.h"
#include "spg_internal.h"
#include "spg_sparse_plan.hpp"

namespace {
inline int pose_stride(int d) { return d == 3 ? 3 : 7; }
inline int info_len(int d) { return d * (d + 1) / 2; }
inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
}  // namespace

// Host mirror of the arena: grows without value-initialising (the regions the device produces are
// never read before they are downloaded), unlike std::vector<double>::resize.
struct HostMirror {
    double *p = nullptr;
    size_t n = 0, cap = 0;
    ~HostMirror() { free(p); }
    HostMirror() {}
    HostMirror(const HostMirror &) = delete;
    HostMirror &operator=(const HostMirror &) = delete;
    double *data() { return p; }
    const double *data() const { return p; }
    size_t size() const { return n; }
    double &operator[](size_t i) { return p[i]; }
    const double &operator[](size_t i) const { return p[i]; }
    void resize(size_t m) {
        if (m > cap) {
            size_t nc = std::max(m, cap * 2);
            p = (double *)realloc(p, nc * sizeof(double));
            if (!p) abort();
            cap = nc;
        }
        n = m;
    }
};

// Small vector of trivially copyable T with N inline slots sharing their storage with the heap pointer of
// the spilled form: sizeof == 8 + N*sizeof(T), so per-vertex adjacency (N = 6 edge ids) and owner lists
// (N = 3 references) are one dense 32-byte record each instead of a vector header plus a heap chunk — the
// scheduler and the graph update are bound by the cache misses on exactly these lists.
template <class T, int N>
struct InlVec {
    int32_t n = 0, cap = N;
    union { T inl[N]; T *ptr; };
    InlVec() {}
    InlVec(const InlVec &o) : n(o.n), cap(o.cap) {
        if (cap > N) { ptr = (T *)malloc(sizeof(T) * (size_t)cap); memcpy(ptr, o.ptr, sizeof(T) * (size_t)n); }
        else memcpy(inl, o.inl, sizeof inl);
    }
    InlVec(InlVec &&o) noexcept : n(o.n), cap(o.cap) {
        if (cap > N) { ptr = o.ptr; o.cap = N; o.n = 0; }
        else memcpy(inl, o.inl, sizeof inl);
    }
    InlVec &operator=(InlVec o) noexcept {
        if (cap > N) free(ptr);
        n = o.n; cap = o.cap;
        if (cap > N) { ptr = o.ptr; o.cap = N; o.n = 0; }
        else memcpy(inl, o.inl, sizeof inl);
        return *this;
    }
    ~InlVec() { if (cap > N) free(ptr); }
    T *data() { return cap > N ? ptr : inl; }
    const T *data() const { return cap > N ? ptr : inl; }
    T *begin() { return data(); }
    T *end() { return data() + n; }
    const T *begin() const { return data(); }
    const T *end() const { return data() + n; }
    size_t size() const { return (size_t)n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return data()[i]; }
    const T &operator[](size_t i) const { return data()[i]; }
    T &back() { return data()[n - 1]; }
    void pop_back() { n--; }
    void clear() { n = 0; }
    void push_back(const T &v) {
        if (n == cap) {
            int32_t nc = cap * 2;
            T *p = (T *)malloc(sizeof(T) * (size_t)nc);
            if (!p) abort();
            memcpy(p, data(), sizeof(T) * (size_t)n);
            if (cap > N) free(ptr);
            ptr = p; cap = nc;
        }
        data()[n++] = v;
    }
};

struct spg_ctx {
    spg_backend be{};
    bool is_hip = false;
    int rank = 0, nranks = 1;
    void *rccl = nullptr;         // communicator handle of csrc/spg_rccl.cpp (nullptr: single rank / no id given)
    int tag_counter = 0;          // ready tags are unique per context (its mailboxes are shared by all graphs)
    int linear_solver = 0;        // SPG_SOLVER_*: dense / block-sparse factorisation for optimize() and the global KLD
    spg::StreamPort *sim_port = nullptr;   // tools/host_sim.cpp only: a simulated persistent worker behind an injected backend
    char err[768] = {0};
};

// 32 bytes (two per cache line, never straddling one); a pose-pose edge carries its two endpoints inline (vtx[0],
// vtx[1]) — the scheduler walks edges by the million and would otherwise take a second cache miss per edge for the
// endpoint list; an n-ary GLC edge keeps its vertices in spg_graph::everts at index vtx[0].
// `key` orders the edges of a blanket for the Hessian sum: the position the edge has in the REFERENCE's sequential
// execution (insertion order; edges created by a removal come after everything that existed when the call began, in
// removal-list order of their root, then in emission order) — the order g2o's edge set hands them out in the
// sequential loop, and independent of the order in which this library happens to commit commuting removals.
struct GEdge {
    int64_t off;
    int64_t key;
    int32_t len;
    int32_t vtx[2];
    int16_t nv;
    int8_t kind;
    uint8_t alive;
};

struct RoundBlanket {
    int32_t root;                 // vertex index
    int32_t n_remove;
    // flat pools (spg_graph::rb_verts / rb_edges) instead of per-blanket vectors: no allocation per blanket
    int32_t vbeg = 0, nv = 0;     // vertex indices, removed first (asc id) then kept (asc id)
    int32_t ebeg = 0, ne = 0;     // edge ids, ascending
    int32_t rank = 0;
    int32_t owner = -1;           // id in spg_graph::owners while the blanket is scheduled / in flight
    spg_blanket_desc desc{};
};

struct BlanketLog { int32_t root_id, round, status, info; double kld, min_gap; };

// One batch of mutually independent blankets: what used to be "the round". Two of them can be in
// flight on the two launch slots of the backend (the host prepares / commits one while the device
// computes the other).
struct Batch {
    std::vector<RoundBlanket> rb;
    std::vector<int32_t> rb_verts, rb_edges;
    std::vector<spg_blanket_desc> h_blk;
    std::vector<int64_t> h_vpo;
    std::vector<spg_edge_ref> h_er;
    std::vector<int32_t> h_ev;
    std::vector<int64_t> chunk_hdr;   // per rank: doubles of out records at the start of its chunk
    spg_round_info rinfo{};
    bool round_open = false;
    bool used_mailbox = false;
    int eff_ranks = 1, eff_rank = 0;   // ranks the batch is split over (1 = computed whole by every rank)
    int slot = 0;
    int seq = 0;                       // launch order
    int round_no = 0;
    int tag = 0;                       // ready tag of the launch (out record word [5])
    double t_launch = 0;               // SPG_TRACE=1 diagnostics
    // blankets committed from the mailbox before their KLD tail finished: (log index, mailbox offset)
    std::vector<std::pair<int32_t, int64_t>> kld_pending;
    // hand-over to the submission thread (pipelined driver): 0 while the batch's descriptors are being written and
    // handed to the device, 1 once that is done (submit_rc = result); a commit waits for 1
    alignas(64) std::atomic<int> submitted{1};
    int submit_rc = 0;
    char pad_[56];
};

struct OwnRefT { int32_t oid; uint32_t gen; };

struct spg_graph {
    spg_ctx *ctx = nullptr;
    int d = 0, ps = 0, rec = 0;
    std::vector<int32_t> vid;
    std::unordered_map<int32_t, int32_t> vidx;
    // direct id -> index table next to the hash map, kept while the ids are small non-negative integers (g2o files number
    // their vertices 0, 1, 2, ...): the 50 000 lookups of a removal list cost 2 ms of a 19 ms marginalisation through the map
    std::vector<int32_t> vdirect;
    bool vdirect_ok = true;
    int32_t index_of(int32_t id) const {
        if (vdirect_ok) return (id >= 0 && (size_t)id < vdirect.size()) ? vdirect[(size_t)id] : -1;
        auto it = vidx.find(id);
        return it == vidx.end() ? -1 : it->second;
    }
    std::vector<uint8_t> valive;
    std::vector<int64_t> vpose;
    // Per vertex, two cache lines. Line 0: adjacency — live edge ids, each with the far endpoint of a pose-pose edge
    // (-1 for an n-ary edge), so that walking a neighbourhood reads no edge records. Line 1: the streaming driver's
    // state of the vertex and what a hand-over needs of it (copies of vid[] / vpose[]).
    struct AdjEnt { int32_t eid, other; operator int32_t() const { return eid; } };
    struct SVtx {                                     // 24 bytes
        int32_t nown;                                 // registered blankets (in flight or reserved) that contain the vertex: own[0 .. nown)
        int32_t own[4];
        int32_t slot;                                 // SV_STABLE / SV_INFLIGHT: the slot that holds the vertex's blanket
    };
    struct alignas(64) VRec {
        InlVec<AdjEnt, 7> adj;
        SVtx s;
        int32_t id, pad_;
        int64_t pose;
        char spare_[24];
        VRec() { s.nown = 0; s.slot = -1; id = 0; pad_ = 0; pose = 0; }
    };
    std::vector<VRec> vr;
    std::vector<InlVec<struct OwnRefT, 3>> vown;      // batch scheduler: owners whose vertex set holds the vertex
    // list position and state of every vertex in one small array (4 bytes per vertex: it stays in L2 while the per-vertex
    // records stream through): -1 = not in the removal list of the running call, else (position << 2) | SV_*
    std::vector<int32_t> cst;
    std::vector<GEdge> edges;
    std::vector<int32_t> everts;
    int n_live_v = 0, n_live_e = 0;
    // arena
    void *dev = nullptr;
    int64_t cap = 0, used = 0;
    HostMirror host;               // mirror of [0, used)
    int64_t dev_synced = 0;        // device holds [0, dev_synced)
    int64_t stale_lo = 0, stale_hi = 0;  // host mirror range that only the device holds
    // marginalisation state
    bool active = false;
    spg_options opts{};
    int rank = 0, nranks = 1;
    std::vector<int32_t> pending;   // removal list (vertex indices) in the caller's order; [pend_head, end) is still to do
    size_t pend_head = 0;
    std::vector<uint8_t> in_set;   // vertex index is in the removal list
    static constexpr int NB = 8;                      // batches that can be in flight (= backend launch slots)
    Batch bt[NB];
    Batch *B = &bt[0];                                // batch the round functions currently work on
    // Owner registry of the scheduler: every scheduled-but-uncommitted blanket and every vertex deferred
    // in the current scheduling pass "owns" a vertex set; vowners[x] lists the owners whose set holds x.
    // Blanket owners persist from the pass that selected them until their batch commits; entries die
    // lazily: freeing an owner bumps its generation and stale references are dropped when next seen.
    struct Owner { int32_t batch, off, len; uint32_t gen; };   // batch >= 0: bt[batch].rb_verts[off, off+len); -1: Dpool
    using OwnRef = OwnRefT;
    std::vector<Owner> owners;
    std::vector<int32_t> owner_free, transient;       // free ids; deferred-vertex owners of the last pass
    std::vector<int32_t> Dpool;                       // sets of the deferred-vertex owners, flat
    int shard_threshold = -1;                         // < 0: cost model (shard_pays); >= 0: minimum blankets
    int round_no = 0, launch_seq = 0;
    bool pipelined = false;                           // two batches in flight (single rank, backend with slots)
    spg_marg_stats stats{};
    double tr_age = 0, tr_wait = 0, tr_first = 0; long tr_n = 0;   // SPG_TRACE=1: launch->commit-start, wait inside commit, launch->first ready word
    std::vector<BlanketLog> log;
    std::vector<double> hdr_buf;
    // Submission thread of the pipelined driver: the graph thread selects and commits, this one writes the descriptors
    // of a selected batch and hands it to the device (descriptor work + device hand-over are ~25 % of the host time per
    // batch and need nothing the graph thread mutates: poses, edge records' locations and the batch's own lists)
    // (every word the two threads exchange sits on its own cache line: the submission thread polls sub_tail, and a line
    //  shared with anything the graph thread writes per blanket would bounce between the cores all the time)
    std::thread sub_thread;
    static constexpr uint32_t SUBQ = 8;
    bool sub_active = false;
    struct alignas(64) SubShared {
        alignas(64) std::atomic<uint32_t> tail{0};    // written by the graph thread
        alignas(64) Batch *q[SUBQ] = {nullptr};       // written by the graph thread
        alignas(64) std::atomic<uint32_t> head{0};    // written by the submission thread
        alignas(64) double seconds = 0;               // submission thread only: time spent (descriptors + hand-over)
        alignas(64) std::atomic<bool> run{false};
        char pad_[64];
    } sub;
    std::vector<int32_t> live_rank;                   // edge id -> index among live edges (spg_graph_vertex_edges)
    long n_mutations = 0, live_rank_stamp = -1;       // bumped whenever an edge is added or dies
    // canonical edge keys (GEdge::key): next key for an edge added by the caller; base of the running marginalisation
    // (new edge e of the removal at list position p gets key_base + p * kKeyStride + e)
    int64_t next_key = 0, key_base = 0;
    static constexpr int64_t kKeyStride = 65536;
    std::vector<int32_t> lpos;                        // vertex index -> position in the removal list of the running call, -1 otherwise
    // scheduler scratch
    std::vector<int32_t> vstamp, estamp;
    int32_t stamp = 0;
    std::vector<int32_t> ocnt;
    std::vector<int32_t> lidx;
    std::vector<int32_t> s_newpending, s_B, s_centres, s_Dv, s_tmp, s_work, s_seen, s_hit, s_vix;
    std::vector<double> s_cost;
    std::vector<int> s_first;
    std::vector<int64_t> s_chunk_len;
    // ---- streaming driver (stream_marginalize below): per-vertex / per-slot / per-position state, kept between calls
    static constexpr int kSOwn = 4, kSMaxV = 16, kSMaxE = 44;
    struct SSlot {                                    // one blanket in flight
        int32_t pos, root, nv, ne, n_new_max, tag, logi, bell;
        int32_t launched;                             // 0: a reservation (the blanket of a waiting entry), 1: in flight
        int32_t npend, pend[3];                       // the blanket's other vertices that are list entries (-1: more than 3, look at all)
        int32_t mcell;                                // its mailbox cell: cells are handed out in launch order, so the host polls and reads sequential memory
        int64_t new_off, out_off;                     // out_off: emulated port only (out record in the arena), else -1
        int32_t verts[kSMaxV];                        // removed vertex first, kept ones in ascending id
        int32_t edges[kSMaxE];                        // ascending key
    };
    std::vector<SSlot> sslots;
    std::vector<int32_t> s_free, s_fifo, s_fin, s_woken, s_ready, wl_next, wl_stable, wl_done;
    int64_t unsorted_from = -1;                       // edges[unsorted_from ..) were appended in commit order by the streaming driver: see canonicalize_edge_order
    bool layout_diverged = false;                     // the graph has streamed on one of several ranks: its arena layout is rank-specific, never shard it again
    int stream_emulation = -1;                        // tests (spg_graph_set_stream_emulation): >= 0 = completion-order seed
    int stream_disabled = 0;                          // SPG_STREAM=0 or spg_graph_set_stream_emulation(g, -2)
};

static inline const int32_t *edge_verts(const spg_graph *g, const GEdge &e) {
    return e.nv == 2 ? e.vtx : g->everts.data() + e.vtx[0];
}

static int set_err(spg_ctx *c, int code, const char *fmt, const char *a = "") {
    if (c) snprintf(c->err, sizeof c->err, fmt, a);
    return code;
}

// ================================================================================= context
// SPG_SEGV_BACKTRACE=1 (diagnostic): a SIGSEGV inside the process prints the native frames (addresses relative to the
// load address of this library, for addr2line on a -g build) before the default action runs.
#if defined(__linux__)
#include <execinfo.h>
#include <signal.h>
#include <dlfcn.h>
static void spg_segv_handler(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    Dl_info di;
    const char *base = nullptr;
    if (dladdr((void *)&spg_segv_handler, &di)) base = (const char *)di.dli_fbase;
    char line[160];
    int len = snprintf(line, sizeof line, "spg: signal %d; frames relative to libspg_hip.so (base %p):\n", sig, (const void *)base);
    if (write(2, line, (size_t)len) < 0) {}
    for (int i = 0; i < n; i++) {
        Dl_info fi;
        const bool ours = dladdr(frames[i], &fi) && fi.dli_fbase == (void *)base;
        len = snprintf(line, sizeof line, "  #%d %s0x%llx\n", i, ours ? "+" : "abs ", (unsigned long long)(ours ? (const char *)frames[i] - base : (const char *)frames[i] - (const char *)0));
        if (write(2, line, (size_t)len) < 0) {}
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
static void spg_install_segv_trace() {
    static bool done = false;
    const char *e = getenv("SPG_SEGV_BACKTRACE");
    if (done || !(e && e[0] == '1')) return;
    done = true;
    signal(SIGSEGV, spg_segv_handler);
    signal(SIGABRT, spg_segv_handler);
}
#else
static void spg_install_segv_trace() {}
#endif

extern "C" int spg_ctx_create(spg_ctx **out, int device) {
    spg_install_segv_trace();
    if (!out) return SPG_EINVAL;
    spg_ctx *c = new spg_ctx;
    int rc = spg::hip_backend_create(device, &c->be, c->err, sizeof c->err);
    if (rc != 0) {
        fprintf(stderr, "libspg_hip: %s\n", c->err);
        delete c;
        *out = nullptr;
        return rc;
    }
    c->is_hip = true;
    *out = c;
    return 0;
}

extern "C" int spg_ctx_create_injected(spg_ctx **out, const spg_backend *backend) {
    if (!out || !backend || !backend->alloc || !backend->run_round) return SPG_EINVAL;
    spg_ctx *c = new spg_ctx;
    c->be = *backend;
    c->is_hip = false;
    *out = c;
    return 0;
}

extern "C" int spg_get_unique_id(void *id_out) {
    if (!id_out) return SPG_EINVAL;
    char err[256];
    int rc = spg::rccl_get_unique_id(id_out, err, sizeof err);
    if (rc) fprintf(stderr, "libspg_hip: %s\n", err);
    return rc;
}

extern "C" int spg_ctx_create_ranks(spg_ctx **out, int device, int rank, int nranks, const void *nccl_unique_id) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return SPG_EINVAL;
    int rc = spg_ctx_create(out, device);
    if (rc) return rc;
    spg_ctx *c = *out;
    c->rank = rank; c->nranks = nranks;
    if (nccl_unique_id) {
        rc = spg::rccl_comm_create(device, rank, nranks, nccl_unique_id, &c->rccl, c->err, sizeof c->err);
        if (rc) {
            fprintf(stderr, "libspg_hip: %s\n", c->err);
            spg_ctx_destroy(c);
            *out = nullptr;
            return rc;
        }
    }
    return 0;
}
extern "C" int spg_ctx_set_linear_solver(spg_ctx *c, int solver) {
    if (!c || solver < SPG_SOLVER_AUTO || solver > SPG_SOLVER_SPARSE) return SPG_EINVAL;
    c->linear_solver = solver;
    return 0;
}

extern "C" int spg_ctx_rank(const spg_ctx *c) { return c ? c->rank : 0; }
extern "C" int spg_ctx_nranks(const spg_ctx *c) { return c ? c->nranks : 0; }

extern "C" int spg_allgather_region(spg_ctx *c, void *arena, int64_t region_off, int64_t chunk_len) {
    if (!c || !arena || region_off < 0 || chunk_len < 0) return SPG_EINVAL;
    if (!c->rccl) {
        // a multi-rank context without a communicator cannot exchange: committing un-gathered chunks would silently
        // diverge the replicas (single-rank contexts: nothing to do)
        if (c->nranks > 1) return set_err(c, SPG_ESTATE, "spg_allgather_region: the context has %s ranks but no RCCL communicator (spg_ctx_create_ranks without a unique id)", std::to_string(c->nranks).c_str());
        return 0;
    }
    return spg::rccl_allgather_f64(c->rccl, arena, region_off, chunk_len, spg::hip_backend_stream(&c->be), c->err, sizeof c->err);
}

extern "C" void spg_ctx_destroy(spg_ctx *c) {
    if (!c) return;
    if (c->rccl) spg::rccl_comm_destroy(c->rccl);
    if (c->is_hip) spg::hip_backend_destroy(&c->be);
    delete c;
}
extern "C" void spg_free(void *p) { free(p); }

extern "C" const char *spg_last_error(spg_ctx *c) {
    if (!c) return "";
    if (c->err[0]) return c->err;
    return c->is_hip ? spg::hip_backend_error(&c->be) : "";
}
extern "C" void *spg_ctx_stream(spg_ctx *c) { return (c && c->is_hip) ? spg::hip_backend_stream(&c->be) : nullptr; }
extern "C" int spg_ctx_synchronize(spg_ctx *c) { return c ? c->be.synchronize(c->be.user) : SPG_EINVAL; }

extern "C" int spg_ctx_profile(spg_ctx *c, int enable) {
    if (!c || !c->is_hip) return SPG_EINVAL;
    spg::hip_backend_profile(&c->be, enable);
    return 0;
}
extern "C" int spg_ctx_profile_read_worker(spg_ctx *c, double *kernel_ms, double *alg_bytes, int64_t *runs, int64_t *blankets) {
    if (!c || !c->is_hip || !kernel_ms || !alg_bytes || !runs || !blankets) return SPG_EINVAL;
    long long r = 0, b = 0;
    spg::hip_backend_profile_read_worker(&c->be, kernel_ms, alg_bytes, &r, &b);
    *runs = r; *blankets = b;
    return 0;
}
extern "C" int spg_ctx_profile_read_big(spg_ctx *c, double *kernel_ms, double *flops, int64_t *blankets, int32_t *n_max) {
    if (!c || !c->is_hip || !kernel_ms || !flops || !blankets || !n_max) return SPG_EINVAL;
    long long cnt = 0;
    int nm = 0;
    spg::hip_backend_profile_read_big(&c->be, kernel_ms, flops, &cnt, &nm);
    *blankets = cnt; *n_max = nm;
    return 0;
}
extern "C" int spg_ctx_profile_read(spg_ctx *c, double *kernel_ms, double *alg_bytes, int64_t *launches, int64_t *blankets) {
    if (!c || !c->is_hip || !kernel_ms || !alg_bytes || !launches || !blankets) return SPG_EINVAL;
    long long l = 0, b = 0;
    spg::hip_backend_profile_read(&c->be, kernel_ms, alg_bytes, &l, &b);
    *launches = l; *blankets = b;
    return 0;
}

// ================================================================================= decimation
// src/decimation.cpp:11-49
static int emit(const std::vector<int> &v, int32_t *out, int cap) {
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
    return (int)v.size();
}
extern "C" int spg_decimate_cluster(int last, int endvert, int sparsity, int clusterSize, int32_t *out, int cap) {
    std::vector<int> ret;
    if (clusterSize > 0 && (((last - 4) % clusterSize == 0 && last > 4) || last == endvert)) {
        for (int i = int(std::ceil((last - 5) / (double)clusterSize) - 1) * clusterSize + 5; i <= last; i++)
            if (i % sparsity > 0) ret.push_back(i);
    }
    return emit(ret, out, cap);
}
extern "C" int spg_decimate_online(int last, int, int sparsity, int, int32_t *out, int cap) {
    std::vector<int> ret;
    if (last % sparsity != 0) ret.push_back(last);
    return emit(ret, out, cap);
}
extern "C" int spg_decimate_global(int last, int endvert, int sparsity, int, int32_t *out, int cap) {
    std::vector<int> ret;
    if (last == endvert)
        for (int i = 4; i <= endvert; i++)
            if (i % sparsity != 0) ret.push_back(i);
    return emit(ret, out, cap);
}

// ================================================================================= arena
static int arena_ensure(spg_graph *g, int64_t need);

static int sync_host(spg_graph *g) {  // pull device-only ranges into the host mirror
    if (g->stale_hi > g->stale_lo) {
        if ((int64_t)g->host.size() < g->used) g->host.resize((size_t)g->used);
        int rc = g->ctx->be.download(g->ctx->be.user, g->host.data() + g->stale_lo,
                                     (char *)g->dev + g->stale_lo * 8, g->stale_hi - g->stale_lo);
        if (rc) return rc;
        g->stale_lo = g->stale_hi = 0;
    }
    return 0;
}

static int sync_device(spg_graph *g) {  // push host-only tail to the device
    if (g->dev_synced < g->used) {
        if (int rc = arena_ensure(g, g->used)) return rc;
        int64_t lo = g->dev_synced;
        // the tail may overlap a stale (device-only) range only if it was produced on the device,
        // in which case dev_synced already covers it
        int rc = g->ctx->be.upload(g->ctx->be.user, (char *)g->dev + lo * 8, g->host.data() + lo, g->used - lo);
        if (rc) return rc;
        g->dev_synced = g->used;
    }
    return 0;
}

static int arena_ensure(spg_graph *g, int64_t need) {
    if (need <= g->cap && g->dev) return 0;
    if (int rc = sync_host(g)) return rc;
    int64_t nc = std::max<int64_t>(need, std::max<int64_t>(g->cap * 2, 1 << 16));
    void *nd = g->ctx->be.alloc(g->ctx->be.user, nc);
    if (!nd) return set_err(g->ctx, SPG_ENOMEM, "arena allocation failed");
    if (g->dev) g->ctx->be.release(g->ctx->be.user, g->dev);
    g->dev = nd;
    g->cap = nc;
    g->dev_synced = 0;
    if ((int64_t)g->host.size() < g->used) g->host.resize((size_t)g->used);
    return 0;
}

static int64_t arena_push(spg_graph *g, const double *src, int64_t len) {
    int64_t off = g->used;
    if ((int64_t)g->host.size() < off + len) g->host.resize((size_t)std::max<int64_t>(off + len, (int64_t)g->host.size() * 2));
    if (src) memcpy(g->host.data() + off, src, (size_t)len * 8);
    g->used += len;
    return off;
}

// ================================================================================= graph basics
extern "C" int spg_graph_create(spg_ctx *ctx, int pose_dim, spg_graph **out) {
    if (!ctx || !out || (pose_dim != 3 && pose_dim != 6)) return SPG_EINVAL;
    spg_graph *g = new spg_graph;
    g->ctx = ctx;
    g->d = pose_dim;
    g->ps = pose_stride(pose_dim);
    g->rec = g->ps + info_len(pose_dim);
    *out = g;
    return 0;
}

extern "C" void spg_graph_destroy(spg_graph *g) {
    if (!g) return;
    if (g->dev) g->ctx->be.release(g->ctx->be.user, g->dev);
    delete g;
}

extern "C" int spg_graph_add_vertex(spg_graph *g, int id, const double *pose) {
    if (!g || !pose || g->active) return SPG_EINVAL;
    if (g->vidx.count(id)) return set_err(g->ctx, SPG_EINVAL, "duplicate vertex id");
    int32_t idx = (int32_t)g->vid.size();
    g->vid.push_back(id);
    g->vidx[id] = idx;
    if (g->vdirect_ok) {
        if (id < 0 || (size_t)id > 8 * g->vid.size() + 4096) { g->vdirect_ok = false; g->vdirect.clear(); g->vdirect.shrink_to_fit(); }
        else { if ((size_t)id >= g->vdirect.size()) g->vdirect.resize(std::max<size_t>((size_t)id + 1, g->vdirect.size() * 2), -1); g->vdirect[(size_t)id] = idx; }
    }
    g->valive.push_back(1);
    g->vr.emplace_back();
    g->vown.emplace_back();
    g->vpose.push_back(arena_push(g, pose, g->ps));
    g->vr.back().id = id;
    g->vr.back().pose = g->vpose.back();
    g->n_live_v++;
    return 0;
}

static void quiesce_submission(spg_graph *g);
static int add_edge_idx(spg_graph *g, int kind, int nv, const int32_t *vix, int64_t off, int32_t len, int64_t key = -1) {
    // the submission thread reads edges[] / everts[] of batches in its queue: never move them under it
    if (g->sub_active && (g->edges.size() == g->edges.capacity() || (nv != 2 && g->everts.size() + (size_t)nv > g->everts.capacity()))) {
        quiesce_submission(g);
        g->edges.reserve(std::max<size_t>(g->edges.capacity() * 2, 1024));
        g->everts.reserve(std::max<size_t>(g->everts.capacity() * 2 + (size_t)nv, 1024));
    }
    GEdge e;
    e.kind = (int8_t)kind; e.nv = (int16_t)nv; e.len = len; e.off = off; e.alive = 1;
    e.key = key >= 0 ? key : g->next_key++;
    if (nv == 2) { e.vtx[0] = vix[0]; e.vtx[1] = vix[1]; }
    else { e.vtx[0] = (int32_t)g->everts.size(); e.vtx[1] = 0; }
    int32_t eid = (int32_t)g->edges.size();
    if (nv != 2) for (int i = 0; i < nv; i++) g->everts.push_back(vix[i]);
    g->edges.push_back(e);
    g->n_mutations++;
    for (int i = 0; i < nv; i++) {
        bool dup = false;
        for (int j = 0; j < i; j++) dup |= (vix[j] == vix[i]);
        if (!dup) g->vr[vix[i]].adj.push_back({eid, (nv == 2 && kind == SPG_EDGE_BINARY) ? vix[1 - i] : -1});
    }
    g->n_live_e++;
    return eid;
}

extern "C" int spg_graph_add_edge(spg_graph *g, int from, int to, const double *meas, const double *info_upper) {
    if (!g || !meas || !info_upper || g->active) return SPG_EINVAL;
    auto a = g->vidx.find(from), b = g->vidx.find(to);
    if (a == g->vidx.end() || b == g->vidx.end() || !g->valive[a->second] || !g->valive[b->second])
        return set_err(g->ctx, SPG_EINVAL, "edge endpoint does not exist");
    int64_t off = arena_push(g, meas, g->ps);
    arena_push(g, info_upper, info_len(g->d));
    int32_t vix[2] = {a->second, b->second};
    add_edge_idx(g, SPG_EDGE_BINARY, 2, vix, off, g->rec);
    return 0;
}

extern "C" int spg_graph_add_glc_edge(spg_graph *g, int q, const int32_t *ids, int r, const double *meas, const double *W) {
    if (!g || q < 1 || r < 1 || !ids || !meas || !W || g->active) return SPG_EINVAL;
    std::vector<int32_t> vix(q);
    for (int i = 0; i < q; i++) {
        auto it = g->vidx.find(ids[i]);
        if (it == g->vidx.end() || !g->valive[it->second]) return set_err(g->ctx, SPG_EINVAL, "edge endpoint does not exist");
        vix[i] = it->second;
    }
    int n = g->d * q;
    int64_t off = arena_push(g, meas, n);
    arena_push(g, W, (int64_t)r * n);
    add_edge_idx(g, SPG_EDGE_GLC, q, vix.data(), off, n + r * n);
    return 0;
}

extern "C" int spg_graph_add_multi_edge(spg_graph *g, int q, const int32_t *ids, const double *record, int64_t len) {
    if (!g || q < 2 || !ids || !record || g->active) return SPG_EINVAL;
    const int nm = (int)record[0];
    if (nm < 1 || len != SPG_MULTI_LEN(g->d, nm)) return set_err(g->ctx, SPG_EINVAL, "multi edge record length does not match its measurement count");
    for (int i = 0; i < 2 * nm; i++) if (record[1 + i] < 0 || record[1 + i] >= q) return set_err(g->ctx, SPG_EINVAL, "multi edge: a measurement refers to a vertex outside the edge");
    std::vector<int32_t> vix(q);
    for (int i = 0; i < q; i++) {
        auto it = g->vidx.find(ids[i]);
        if (it == g->vidx.end() || !g->valive[it->second]) return set_err(g->ctx, SPG_EINVAL, "edge endpoint does not exist");
        vix[i] = it->second;
    }
    int64_t off = arena_push(g, record, len);
    add_edge_idx(g, SPG_EDGE_MULTI, q, vix.data(), off, (int32_t)len);
    return 0;
}

// bulk forms of addVertex / addEdge (same semantics, one call per array)
extern "C" int spg_graph_add_vertices(spg_graph *g, int n, const int32_t *ids, const double *poses) {
    if (!g || n < 0 || !ids || !poses) return SPG_EINVAL;
    g->vid.reserve(g->vid.size() + n);
    for (int i = 0; i < n; i++)
        if (int rc = spg_graph_add_vertex(g, ids[i], poses + (size_t)i * g->ps)) return rc;
    return 0;
}
extern "C" int spg_graph_add_edges(spg_graph *g, int n, const int32_t *ij, const double *records) {
    if (!g || n < 0 || !ij || !records) return SPG_EINVAL;
    g->edges.reserve(g->edges.size() + n);
    for (int i = 0; i < n; i++) {
        const double *r = records + (size_t)i * g->rec;
        if (int rc = spg_graph_add_edge(g, ij[2 * i], ij[2 * i + 1], r, r + g->ps)) return rc;
    }
    return 0;
}

extern "C" int spg_graph_pose_dim(const spg_graph *g) { return g ? g->d : 0; }
extern "C" int spg_graph_num_vertices(const spg_graph *g) { return g ? g->n_live_v : 0; }
extern "C" int spg_graph_num_edges(const spg_graph *g) { return g ? g->n_live_e : 0; }
// The streaming driver appends new edges in the order their blankets happen to complete, which varies from run to run.
// Results never depend on it (blanket edges are summed in key order), but everything that EXPOSES the edge array's order
// does: spg_graph_get_edges / the .g2o writer / clones, and the dense assembly, which sums a vertex's incident edges in
// array order. Before any of those the tail the stream appended is put into key order — the order the sequential loop
// would have inserted the edges in — so that two runs on the same input hand out byte-identical graphs. Lazy: it costs
// ~3 ms on the 100k-pose graph and a marginalisation that is only followed by another one never pays it.
static void next_stamp(spg_graph *g);
static void canonicalize_edge_order(spg_graph *g) {
    if (g->unsorted_from < 0 || g->active) return;
    const size_t from = (size_t)g->unsorted_from, n = g->edges.size() - from;
    g->unsorted_from = -1;
    if (n < 2) return;
    std::vector<uint32_t> perm(n);
    for (size_t i = 0; i < n; i++) perm[i] = (uint32_t)i;
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i++) sorted = g->edges[from + i - 1].key <= g->edges[from + i].key;
    if (sorted) return;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return g->edges[from + a].key < g->edges[from + b].key; });
    std::vector<int32_t> newid(n);
    std::vector<GEdge> tmp(n);
    for (size_t i = 0; i < n; i++) { tmp[i] = g->edges[from + perm[i]]; newid[perm[i]] = (int32_t)(from + i); }
    std::copy(tmp.begin(), tmp.end(), g->edges.begin() + (long)from);
    next_stamp(g);
    const int32_t st = g->stamp;
    for (size_t i = 0; i < n; i++) {
        const GEdge &e = g->edges[from + i];
        if (!e.alive) continue;
        for (int t = 0; t < e.nv; t++) {
            const int32_t v = edge_verts(g, e)[t];
            if (g->vstamp[v] == st) continue;
            g->vstamp[v] = st;
            for (auto &a : g->vr[v].adj) if ((size_t)a.eid >= from) a.eid = newid[(size_t)a.eid - from];
        }
    }
    g->n_mutations++;
}

extern "C" int64_t spg_graph_edge_data_size(const spg_graph *g) {
    int64_t s = 0;
    for (auto &e : g->edges) if (e.alive) s += e.len;
    return s;
}
extern "C" int64_t spg_graph_edge_vert_size(const spg_graph *g) {
    int64_t s = 0;
    for (auto &e : g->edges) if (e.alive) s += e.nv;
    return s;
}

extern "C" int spg_graph_get_vertices(spg_graph *g, int32_t *ids, double *poses) {
    if (!g) return SPG_EINVAL;
    if (int rc = sync_host(g)) return rc;
    std::vector<std::pair<int32_t, int32_t>> order;
    for (size_t i = 0; i < g->vid.size(); i++) if (g->valive[i]) order.push_back({g->vid[i], (int32_t)i});
    std::sort(order.begin(), order.end());
    for (size_t k = 0; k < order.size(); k++) {
        ids[k] = order[k].first;
        memcpy(poses + k * g->ps, g->host.data() + g->vpose[order[k].second], (size_t)g->ps * 8);
    }
    return (int)order.size();
}

extern "C" int spg_graph_get_edges(spg_graph *g, int32_t *kind, int32_t *vert_off, int32_t *vert_ids, int64_t *data_off, double *data) {
    if (!g) return SPG_EINVAL;
    canonicalize_edge_order(g);
    if (int rc = sync_host(g)) return rc;
    int ne = 0, nv = 0;
    int64_t nd = 0;
    vert_off[0] = 0; data_off[0] = 0;
    for (auto &e : g->edges) {
        if (!e.alive) continue;
        kind[ne] = e.kind;
        for (int i = 0; i < e.nv; i++) vert_ids[nv++] = g->vid[edge_verts(g, e)[i]];
        memcpy(data + nd, g->host.data() + e.off, (size_t)e.len * 8);
        nd += e.len;
        ne++;
        vert_off[ne] = nv; data_off[ne] = nd;
    }
    return ne;
}

extern "C" int spg_graph_set_estimate(spg_graph *g, int id, const double *pose) {
    if (!g || !pose || g->active) return SPG_EINVAL;
    auto it = g->vidx.find(id);
    if (it == g->vidx.end() || !g->valive[it->second]) return SPG_EINVAL;
    int64_t off = g->vpose[it->second];
    memcpy(g->host.data() + off, pose, (size_t)g->ps * 8);
    if (off < g->dev_synced && g->dev)
        return g->ctx->be.upload(g->ctx->be.user, (char *)g->dev + off * 8, pose, g->ps);
    return 0;
}

extern "C" void *spg_graph_arena(spg_graph *g, int64_t *capacity) {
    if (!g) return nullptr;
    if (capacity) *capacity = g->cap;
    return g->dev;
}

extern "C" int spg_graph_reserve(spg_graph *g, int64_t arena_doubles) {
    if (!g) return SPG_EINVAL;
    if (int rc = arena_ensure(g, std::max(arena_doubles, g->used))) return rc;
    // The host mirror is reserved AND touched up to the same size: the commit copies every batch's out records into it,
    // and a first touch there is a page fault inside the timed path — with transparent huge pages a 2 MB zero-fill
    // (~100 us) per fault, ~150 of them per 100 k-pose marginalisation on the mirrors the kernel happened to back with
    // huge pages (measured: 25 ms per call on some graphs, 55 ms on others, device time identical).
    if ((int64_t)g->host.size() < g->cap) g->host.resize((size_t)g->cap);
    // the same for the containers a marginalisation appends to: room for as many new edges as there are now, touched
    {
        const size_t ne = g->edges.size(), nl = g->log.size();
        g->edges.resize(2 * ne + 1024); g->edges.resize(ne);
        g->log.resize(g->vid.size() + 16); g->log.resize(nl);
        // and the scheduler's own scratch, to the sizes a few hundred blankets per batch need (they grow past that as ever)
        auto touch = [](auto &v, size_t n) { if (v.capacity() < n) { const size_t keep = v.size(); v.resize(n); v.resize(keep); } };
        const size_t nv = g->vid.size();
        for (int i = 0; i < spg_graph::NB; i++) {
            Batch &b = g->bt[i];
            touch(b.rb, 1024); touch(b.rb_verts, 16384); touch(b.rb_edges, 32768); touch(b.h_blk, 1024); touch(b.h_vpo, 16384);
            touch(b.h_er, 32768); touch(b.h_ev, 65536); touch(b.kld_pending, 1024);
        }
        touch(g->pending, nv); touch(g->in_set, nv); touch(g->vstamp, nv); touch(g->estamp, 2 * ne + 1024);
        touch(g->owners, 8192); touch(g->ocnt, 8192); touch(g->owner_free, 8192); touch(g->transient, 4096); touch(g->Dpool, 65536);
        touch(g->s_newpending, 4096); touch(g->hdr_buf, 65536);
    }
    return sync_device(g);
}

// ================================================================================= .g2o I/O
static void normalize_quat(double *q) {
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 0) for (int i = 0; i < 4; i++) q[i] /= n;
}

extern "C" int spg_graph_load_g2o(spg_ctx *ctx, const char *path, spg_graph **out) {
    if (!ctx || !path || !out) return SPG_EINVAL;
    FILE *f = fopen(path, "r");
    if (!f) return set_err(ctx, SPG_EIO, "cannot open %s", path);
    struct V { int id; double p[7]; };
    struct E { int a, b; double r[28]; };
    struct GE { std::vector<int32_t> ids; int r = 0, n = 0; std::vector<double> meas, W; };
    std::vector<V> vs;
    std::vector<E> es;
    std::vector<GE> ges;
    int d = 0;
    char *lbuf = nullptr;   // getline: a Dense GLC edge is one line of r*n numbers, far beyond any fixed buffer
    size_t lcap = 0;
    bool bad_glc = false, has_multi = false;
    while (getline(&lbuf, &lcap, f) >= 0) {
        char *s = lbuf;
        while (*s == ' ' || *s == '\t') s++;
        char tag[64];
        int adv = 0;
        if (sscanf(s, "%63s%n", tag, &adv) != 1) continue;
        s += adv;
        auto readn = [&](double *dst, int cnt) {
            for (int i = 0; i < cnt; i++) {
                char *end;
                dst[i] = strtod(s, &end);
                if (end == s) return false;
                s = end;
            }
            return true;
        };
        auto readi = [&](int &v) {
            char *end;
            long x = strtol(s, &end, 10);
            if (end == s) return false;
            v = (int)x; s = end;
            return true;
        };
        if (!strcmp(tag, "VERTEX_SE2")) {
            V v{};
            if (!readi(v.id) || !readn(v.p, 3)) continue;
            if (!d) d = 3;
            vs.push_back(v);
        } else if (!strcmp(tag, "VERTEX_SE3:QUAT")) {
            V v{};
            if (!readi(v.id) || !readn(v.p, 7)) continue;
            normalize_quat(v.p + 3);
            if (!d) d = 6;
            vs.push_back(v);
        } else if (!strcmp(tag, "EDGE_SE2")) {
            E e{};
            if (!readi(e.a) || !readi(e.b) || !readn(e.r, 9)) continue;
            es.push_back(e);
        } else if (!strcmp(tag, "EDGE_SE3:QUAT")) {
            E e{};
            if (!readi(e.a) || !readi(e.b) || !readn(e.r, 28)) continue;
            normalize_quat(e.r + 3);
            es.push_back(e);
        } else if (!strcmp(tag, "GLC_EDGE")) {
            // GLCEdge::read (src/glc_edge.cpp:65-93) behind g2o's hyper-edge prefix "id id ... ||":
            // <reparam tag> r n, measurement (n), W (r x n, row-major), information (upper triangle, r x r)
            GE ge;
            for (;;) {
                while (*s == ' ' || *s == '\t') s++;
                if (s[0] == '|' && s[1] == '|') { s += 2; break; }
                int id;
                if (!readi(id)) { bad_glc = true; break; }
                ge.ids.push_back(id);
            }
            char rtag[64];
            int adv2 = 0;
            if (bad_glc || sscanf(s, "%63s%n", rtag, &adv2) != 1) { bad_glc = true; continue; }
            s += adv2;
            if (!readi(ge.r) || !readi(ge.n) || ge.r < 1 || ge.n < 1 || ge.ids.empty() || ge.n % (int)ge.ids.size()) { bad_glc = true; continue; }
            ge.meas.resize(ge.n);
            ge.W.resize((size_t)ge.r * ge.n);
            std::vector<double> info((size_t)ge.r * (ge.r + 1) / 2);
            if (!readn(ge.meas.data(), ge.n) || !readn(ge.W.data(), ge.r * ge.n) || !readn(info.data(), (int)info.size())) { bad_glc = true; continue; }
            // the reference writes information = I_r; a general SPD information is folded into W (W <- L^T W, Omega = L L^T)
            bool ident = true;
            {
                size_t p2 = 0;
                for (int i = 0; i < ge.r; i++) for (int j = i; j < ge.r; j++) { ident &= (info[p2] == (i == j ? 1.0 : 0.0)); p2++; }
            }
            if (!ident) {
                std::vector<double> Lm((size_t)ge.r * ge.r, 0.0);
                size_t p2 = 0;
                for (int i = 0; i < ge.r; i++) for (int j = i; j < ge.r; j++) { Lm[(size_t)j * ge.r + i] = info[p2]; p2++; }   // lower triangle
                bool pd = true;
                for (int j = 0; j < ge.r && pd; j++) {
                    double dj = Lm[(size_t)j * ge.r + j];
                    for (int k = 0; k < j; k++) dj -= Lm[(size_t)j * ge.r + k] * Lm[(size_t)j * ge.r + k];
                    if (!(dj > 0)) { pd = false; break; }
                    double l = std::sqrt(dj);
                    Lm[(size_t)j * ge.r + j] = l;
                    for (int i = j + 1; i < ge.r; i++) {
                        double sacc = Lm[(size_t)i * ge.r + j];
                        for (int k = 0; k < j; k++) sacc -= Lm[(size_t)i * ge.r + k] * Lm[(size_t)j * ge.r + k];
                        Lm[(size_t)i * ge.r + j] = sacc / l;
                    }
                }
                if (!pd) { bad_glc = true; continue; }
                std::vector<double> W2((size_t)ge.r * ge.n, 0.0);
                for (int i = 0; i < ge.r; i++) for (int k = i; k < ge.r; k++) for (int c = 0; c < ge.n; c++) W2[(size_t)i * ge.n + c] += Lm[(size_t)k * ge.r + i] * ge.W[(size_t)k * ge.n + c];
                ge.W.swap(W2);
            }
            ges.push_back(std::move(ge));
        } else if (!strncmp(tag, "MULTI_EDGE_", 11)) {
            // MultiEdgeCorrelated::write (src/multi_edge_correlated.hpp:227-267) does not say which vertex pair each
            // measurement belongs to — the reference's own read() cannot restore the edge either. Dropping the line would
            // hand back a graph without its correlated constraints (possibly disconnected) and no error: refuse the file.
            has_multi = true;
        }
    }
    free(lbuf);
    fclose(f);
    if (has_multi) return set_err(ctx, SPG_EIO, "%s holds MULTI_EDGE_* records: that format omits the vertex pair of each measurement and cannot be read back", path);
    if (bad_glc) return set_err(ctx, SPG_EIO, "malformed GLC_EDGE record in %s", path);
    if (!d) return set_err(ctx, SPG_EIO, "no SE2/SE3 vertices in %s", path);
    std::stable_sort(vs.begin(), vs.end(), [](const V &a, const V &b) { return a.id < b.id; });
    spg_graph *g;
    if (int rc = spg_graph_create(ctx, d, &g)) return rc;
    for (auto &v : vs) if (int rc = spg_graph_add_vertex(g, v.id, v.p)) { spg_graph_destroy(g); return rc; }
    for (auto &e : es) if (int rc = spg_graph_add_edge(g, e.a, e.b, e.r, e.r + g->ps)) { spg_graph_destroy(g); return rc; }
    for (auto &ge : ges) {
        if (ge.n != d * (int)ge.ids.size()) { spg_graph_destroy(g); return set_err(ctx, SPG_EIO, "GLC_EDGE dimension does not match its vertices in %s", path); }
        if (int rc = spg_graph_add_glc_edge(g, (int)ge.ids.size(), ge.ids.data(), ge.r, ge.meas.data(), ge.W.data())) { spg_graph_destroy(g); return rc; }
    }
    *out = g;
    return 0;
}

static void write_g2o_stream(spg_graph *g, FILE *f);
extern "C" int spg_graph_write_g2o(spg_graph *g, const char *path) {
    if (!g || !path) return SPG_EINVAL;
    if (int rc = sync_host(g)) return rc;
    FILE *f = fopen(path, "w");
    if (!f) return set_err(g->ctx, SPG_EIO, "cannot open %s for writing", path);
    write_g2o_stream(g, f);
    fclose(f);
    return 0;
}
extern "C" int spg_graph_write_g2o_mem(spg_graph *g, char **text, size_t *len) {
    if (!g || !text || !len) return SPG_EINVAL;
    if (int rc = sync_host(g)) return rc;
    *text = nullptr; *len = 0;
    FILE *f = open_memstream(text, len);
    if (!f) return set_err(g->ctx, SPG_ENOMEM, "open_memstream failed");
    write_g2o_stream(g, f);
    fclose(f);   // finalises *text / *len (NUL-terminated)
    return 0;
}
static void write_g2o_stream(spg_graph *g, FILE *f) {
    canonicalize_edge_order(g);
    std::vector<std::pair<int32_t, int32_t>> order;
    for (size_t i = 0; i < g->vid.size(); i++) if (g->valive[i]) order.push_back({g->vid[i], (int32_t)i});
    std::sort(order.begin(), order.end());
    const char *vt = g->d == 3 ? "VERTEX_SE2" : "VERTEX_SE3:QUAT", *et = g->d == 3 ? "EDGE_SE2" : "EDGE_SE3:QUAT";
    for (auto &o : order) {
        fprintf(f, "%s %d", vt, o.first);
        for (int i = 0; i < g->ps; i++) fprintf(f, " %.17g", g->host[g->vpose[o.second] + i]);
        fputc('\n', f);
    }
    for (auto &e : g->edges) {
        if (!e.alive) continue;
        if (e.kind == SPG_EDGE_BINARY) {
            fprintf(f, "%s %d %d", et, g->vid[edge_verts(g, e)[0]], g->vid[edge_verts(g, e)[1]]);
            for (int i = 0; i < e.len; i++) fprintf(f, " %.17g", g->host[e.off + i]);
        } else if (e.kind == SPG_EDGE_MULTI) {
            // MultiEdgeCorrelated::write (src/multi_edge_correlated.hpp:227-267): "|| nmeas nrelevant meas... info(upper)". As in
            // the reference the vertex pair of each measurement is NOT part of the line (its own reader cannot restore it).
            const double *rec = g->host.data() + e.off;
            const int nm = (int)rec[0], r = g->d * nm;
            const double *meas = rec + 1 + 2 * nm, *W = meas + (size_t)nm * g->ps;
            fprintf(f, "%s", g->d == 3 ? "MULTI_EDGE_SE2" : "MULTI_EDGE_SE3");
            for (int i = 0; i < e.nv; i++) fprintf(f, " %d", g->vid[edge_verts(g, e)[i]]);
            fprintf(f, " || %d %d", nm, g->ps);
            for (int i = 0; i < nm * g->ps; i++) fprintf(f, " %.17g", meas[i]);
            for (int i = 0; i < r; i++) for (int j = i; j < r; j++) {
                double v = 0;
                for (int t = 0; t < r; t++) v += W[(size_t)t * r + i] * W[(size_t)t * r + j];
                fprintf(f, " %.17g", v);
            }
        } else {
            // GLCEdge::write (src/glc_edge.cpp:95-119): "|| <reparam tag> r dq meas W info(upper of I_r)"
            int n = g->d * e.nv, r = (e.len - n) / n;
            fprintf(f, "GLC_EDGE");
            for (int i = 0; i < e.nv; i++) fprintf(f, " %d", g->vid[edge_verts(g, e)[i]]);
            fprintf(f, " || %s %d %d", g->d == 3 ? "GLC_REPARAM_SE2_ISAM" : "GLC_REPARAM_SE3", r, n);
            for (int i = 0; i < e.len; i++) fprintf(f, " %.17g", g->host[e.off + i]);
            for (int i = 0; i < r; i++) for (int j = i; j < r; j++) fprintf(f, " %d", i == j ? 1 : 0);
        }
        fputc('\n', f);
    }
}

// GraphWrapperG2O::clonePortion (src/graph_wrapper_g2o.cpp:334-356)
extern "C" int spg_graph_clone_portion(spg_graph *g, int maxid, spg_graph **out) {
    if (!g || !out || g->active) return SPG_EINVAL;
    canonicalize_edge_order(g);
    if (int rc = sync_host(g)) return rc;
    spg_graph *c;
    if (int rc = spg_graph_create(g->ctx, g->d, &c)) return rc;
    std::vector<std::pair<int32_t, int32_t>> order;
    for (size_t i = 0; i < g->vid.size(); i++) if (g->valive[i] && g->vid[i] <= maxid) order.push_back({g->vid[i], (int32_t)i});
    std::sort(order.begin(), order.end());
    int rc = 0;
    for (auto &o : order) if ((rc = spg_graph_add_vertex(c, o.first, g->host.data() + g->vpose[o.second]))) break;
    std::vector<int32_t> ids;
    for (size_t ei = 0; ei < g->edges.size() && !rc; ei++) {
        const GEdge &e = g->edges[ei];
        if (!e.alive) continue;
        bool in = true;
        ids.clear();
        for (int i = 0; i < e.nv; i++) { int32_t id = g->vid[edge_verts(g, e)[i]]; in &= (id <= maxid); ids.push_back(id); }
        if (!in) continue;
        const double *rec = g->host.data() + e.off;
        if (e.kind == SPG_EDGE_BINARY) rc = spg_graph_add_edge(c, ids[0], ids[1], rec, rec + g->ps);
        else if (e.kind == SPG_EDGE_MULTI) rc = spg_graph_add_multi_edge(c, e.nv, ids.data(), rec, e.len);
        else { int n = g->d * e.nv; rc = spg_graph_add_glc_edge(c, e.nv, ids.data(), (e.len - n) / n, rec, rec + n); }
    }
    if (rc) { spg_graph_destroy(c); return rc; }
    *out = c;
    return 0;
}

// GraphWrapper::Vertex::edges() (src/graph_wrapper.h:26)
extern "C" int spg_graph_vertex_edges(spg_graph *g, int id, int32_t *edge_index, int cap) {
    if (!g || (cap > 0 && !edge_index)) return SPG_EINVAL;
    auto it = g->vidx.find(id);
    if (it == g->vidx.end() || !g->valive[it->second]) return set_err(g->ctx, SPG_EINVAL, "no such vertex");
    std::vector<int32_t> es(g->vr[it->second].adj.begin(), g->vr[it->second].adj.end());
    std::sort(es.begin(), es.end());
    // position of an edge in spg_graph_get_edges order = number of live edges before it
    if (g->live_rank_stamp != g->n_mutations || g->live_rank.size() != g->edges.size()) {
        g->live_rank.resize(g->edges.size());
        int32_t r = 0;
        for (size_t e = 0; e < g->edges.size(); e++) { g->live_rank[e] = r; r += g->edges[e].alive ? 1 : 0; }
        g->live_rank_stamp = g->n_mutations;
    }
    int n = 0;
    for (int32_t e : es) { if (n < cap) edge_index[n] = g->live_rank[e]; n++; }
    return n;
}

// ================================================================================= scheduler
static void next_stamp(spg_graph *g) {
    if (g->vstamp.size() < g->vid.size()) g->vstamp.resize(g->vid.size(), 0);
    if (g->estamp.size() < g->edges.size()) g->estamp.resize(g->edges.size(), 0);
    g->stamp++;
}

// N[v] including v (markovBlanketVertices, src/vertex_remover.cpp:197-215); unsorted, deduplicated
static void closed_neighbourhood(spg_graph *g, int32_t v, std::vector<int32_t> &out) {
    next_stamp(g);
    out.clear();
    out.push_back(v);
    g->vstamp[v] = g->stamp;
    for (int32_t eid : g->vr[v].adj) {
        const GEdge &e = g->edges[eid];
        for (int i = 0; i < e.nv; i++) {
            int32_t u = edge_verts(g, e)[i];
            if (g->vstamp[u] != g->stamp) { g->vstamp[u] = g->stamp; out.push_back(u); }
        }
    }
}

// extendedMarkovBlanketVertices (src/vertex_remover.cpp:142-195), literal: one ascending pass over
// the growing id-ordered set; pick bin = every vertex of the removal list that is still alive.
static void extended_blanket(spg_graph *g, int32_t root, std::vector<int32_t> &verts, std::vector<int32_t> &picked) {
    auto byid = [g](int32_t a, int32_t b) { return g->vid[a] < g->vid[b]; };
    std::set<int32_t, decltype(byid)> ret(byid), pk(byid);
    std::vector<int32_t> tmp;
    closed_neighbourhood(g, root, tmp);
    ret.insert(tmp.begin(), tmp.end());
    pk.insert(root);
    for (auto it = ret.begin(); it != ret.end(); ++it) {
        int32_t v = *it;
        if (g->in_set[v] && !pk.count(v)) {
            pk.insert(v);
            closed_neighbourhood(g, v, tmp);
            ret.insert(tmp.begin(), tmp.end());
        }
    }
    verts.assign(ret.begin(), ret.end());
    picked.assign(pk.begin(), pk.end());
}

static bool dense_mode(const spg_options &o) { return o.topology == SPG_TOPO_DENSE || o.topology == SPG_TOPO_CLIQUEY_DENSE; }

// markovBlanketEdges (src/vertex_remover.cpp:225-251) for a selected blanket. `verts` must be stamped.
static void collect_edges(spg_graph *g, const int32_t *verts, int nverts, const std::vector<int32_t> &centres,
                          bool intra, std::vector<int32_t> &out) {
    next_stamp(g);
    int32_t st = g->stamp;
    for (int i = 0; i < nverts; i++) g->vstamp[verts[i]] = st;
    out.clear();
    for (int vi_ = 0; vi_ < nverts; vi_++) {
        int32_t v = verts[vi_];
        for (int32_t eid : g->vr[v].adj) {
            if (g->estamp[eid] == st) continue;
            g->estamp[eid] = st;
            const GEdge &e = g->edges[eid];
            bool ok = true, hub = false;
            for (int i = 0; i < e.nv; i++) {
                int32_t u = edge_verts(g, e)[i];
                if (g->vstamp[u] != st) { ok = false; break; }
                if (!intra) for (int32_t c : centres) hub |= (c == u);
            }
            if (ok && (intra || hub)) out.push_back(eid);
        }
    }
    // (ascending key = the reference's sequential edge order, whatever order commuting removals were committed in)
    std::sort(out.begin(), out.end(), [g](int32_t a, int32_t b) { return g->edges[a].key < g->edges[b].key; });
}

static void new_edge_budget(const spg_options &o, int d, int k, int32_t &n_new_max, int32_t &n_new_vert_max, int64_t &new_len) {
    int ps = pose_stride(d);
    if (o.algorithm == SPG_ALG_NFR) {
        // pattern size (src/pseudo_chow_liu.cpp:33-87): a tree, or up to all pairs for Dense / Subgraph
        n_new_max = std::max(k - 1, 0);
        if (k > 2 && (o.topology == SPG_TOPO_DENSE || o.topology == SPG_TOPO_SUBGRAPH)) {
            const int msub = (int)((1 + o.chord_ratio) * (k - 1)), all = k * (k - 1) / 2;
            n_new_max = (o.topology == SPG_TOPO_DENSE || msub >= all) ? all : std::max(msub, k - 1);
        }
        n_new_vert_max = 2 * n_new_max;
        new_len = (int64_t)n_new_max * (ps + info_len(d));
        // correlated patterns: up to k - 1 measurements in all, possibly in one SPG_EDGE_MULTI record
        if (k > 2 && (o.topology == SPG_TOPO_CLIQUEY_SUBGRAPH || o.topology == SPG_TOPO_CLIQUEY_DENSE)) new_len += SPG_MULTI_LEN(d, k - 1);
    } else if (o.topology == SPG_TOPO_DENSE || k <= 1) {
        int64_t n = (int64_t)d * k;
        n_new_max = k > 0 ? 1 : 0;
        n_new_vert_max = k;
        new_len = n + n * n;
    } else {
        int64_t n2 = 2 * d;
        n_new_max = k;
        n_new_vert_max = 2 * k - 1;
        new_len = (d + (int64_t)d * d) + (int64_t)(k - 1) * (n2 + n2 * n2);
    }
}

static int32_t owner_acquire(spg_graph *g, int32_t batch, int32_t off, int32_t len) {
    int32_t oid;
    if (!g->owner_free.empty()) { oid = g->owner_free.back(); g->owner_free.pop_back(); }
    else { oid = (int32_t)g->owners.size(); g->owners.push_back({0, 0, 0, 0}); g->ocnt.push_back(0); }
    spg_graph::Owner &o = g->owners[oid];
    o.batch = batch; o.off = off; o.len = len;
    return oid;
}
static void owner_release(spg_graph *g, int32_t oid) {
    g->owners[oid].gen++;   // every reference to it in vowners[] is stale from now on
    g->owner_free.push_back(oid);
}
static const int32_t *owner_set(const spg_graph *g, const spg_graph::Owner &o) {
    return (o.batch >= 0 ? g->bt[o.batch].rb_verts.data() : g->Dpool.data()) + o.off;
}
static void release_batch_owners(spg_graph *g, Batch &bt) {
    for (RoundBlanket &r : bt.rb) if (r.owner >= 0) { owner_release(g, r.owner); r.owner = -1; }
}

// Select this round's mutually commuting blankets, in list order. Fills bt.rb; rewrites g->pending.
#ifdef SPG_SCHED_PROF
#include <x86intrin.h>
static unsigned long long prof_t[16], prof_n[16];
#define PT0 unsigned long long pt_ = __rdtsc()
#define PT(i) do { unsigned long long n_ = __rdtsc(); prof_t[i] += n_ - pt_; prof_n[i]++; pt_ = n_; } while (0)
#else
#define PT0 do {} while (0)
#define PT(i) do {} while (0)
#endif
static void schedule_round(spg_graph *g) {
    Batch &bt = *g->B;
    const spg_options &o = g->opts;
    const bool dense = dense_mode(o);
    const size_t DCAP = 512;
    bt.rb.clear();
    bt.rb_verts.clear();
    bt.rb_edges.clear();
    // the deferred-vertex owners of the previous pass are void; blanket owners of batches still in
    // flight stay registered (their removals are not in the host graph yet, so nothing that fails to
    // commute with them may be selected now)
    for (int32_t oid : g->transient) owner_release(g, oid);
    g->transient.clear();
    g->Dpool.clear();
    const int32_t my_batch = (int32_t)(&bt - g->bt);
    // scratch that keeps its capacity between passes (a pass runs a thousand times per marginalisation)
    std::vector<int32_t> &newpending = g->s_newpending, &B = g->s_B, &centres = g->s_centres, &Dv = g->s_Dv, &tmp = g->s_tmp,
                         &work = g->s_work, &seen_owner = g->s_seen, &hit = g->s_hit;
    newpending.clear();
    bool stop = false;
    size_t n_deferred = 0, consec = 0;
    auto reg = [&](int32_t batch, int32_t off, int32_t len) -> int32_t {
        int32_t oid = owner_acquire(g, batch, off, len);
        const uint32_t gen = g->owners[oid].gen;
        const int32_t *set = owner_set(g, g->owners[oid]);
        for (int32_t i = 0; i < len; i++) g->vown[set[i]].push_back({oid, gen});
        return oid;
    };
    // live owners of x, dropping stale references on the way
    auto for_owners = [&](int32_t x, auto &&fn) {
        auto &vo = g->vown[x];
        for (size_t i = 0; i < vo.size();) {
            const spg_graph::OwnRef r = vo[i];
            if (g->owners[r.oid].gen != r.gen) { vo[i] = vo.back(); vo.pop_back(); continue; }
            fn(r.oid);
            i++;
        }
    };
    bool inflight = false;
    for (int bi = 0; bi < spg_graph::NB; bi++) inflight |= (&g->bt[bi] != &bt && g->bt[bi].round_open);
    // The scan touches only a prefix of the pending list: entries that have to wait are written back
    // right in front of the untouched tail, so a call costs O(scanned), not O(pending).
    size_t pos = g->pend_head;
    for (; pos < g->pending.size() && !stop; pos++) {
        int32_t v = g->pending[pos];
        if (!g->valive[v]) continue;  // absorbed by an earlier cluster (`deleted`, src/vertex_remover.cpp:91)
        PT0;
        if (dense) extended_blanket(g, v, B, centres);
        else { closed_neighbourhood(g, v, B); centres.assign(1, v); }
        PT(0);
        bool inD = false, conflict = false;
        hit.clear();
        for (int32_t x : B) {
            bool is_c = false;
            for (int32_t c : centres) is_c |= (c == x);
            for_owners(x, [&](int32_t oid) {
                if (is_c) inD = true;
                if (g->ocnt[oid]++ == 0) hit.push_back(oid);
                if (g->ocnt[oid] >= 2) conflict = true;
            });
        }
        for (int32_t oid : hit) g->ocnt[oid] = 0;
        PT(1);
        if (!inD && !conflict) {
            RoundBlanket rbk;
            rbk.root = v;
            rbk.n_remove = (int32_t)centres.size();
            auto byid = [g](int32_t a, int32_t b) { return g->vid[a] < g->vid[b]; };
            std::sort(centres.begin(), centres.end(), byid);
            tmp.clear();
            for (int32_t x : B) {
                bool is_c = false;
                for (int32_t c : centres) is_c |= (c == x);
                if (!is_c) tmp.push_back(x);
            }
            std::sort(tmp.begin(), tmp.end(), byid);
            rbk.vbeg = (int32_t)bt.rb_verts.size();
            bt.rb_verts.insert(bt.rb_verts.end(), centres.begin(), centres.end());
            bt.rb_verts.insert(bt.rb_verts.end(), tmp.begin(), tmp.end());
            rbk.nv = (int32_t)(centres.size() + tmp.size());
            PT(5);
            collect_edges(g, bt.rb_verts.data() + rbk.vbeg, rbk.nv, centres, o.include_intra_clique != 0, work);
            rbk.ebeg = (int32_t)bt.rb_edges.size();
            rbk.ne = (int32_t)work.size();
            bt.rb_edges.insert(bt.rb_edges.end(), work.begin(), work.end());
            PT(6);
            rbk.owner = reg(my_batch, rbk.vbeg, rbk.nv);
            bt.rb.push_back(std::move(rbk));
            consec = 0;
            PT(2);
        } else {
            newpending.push_back(v);
            n_deferred++;
            // D(v): everything v's blanket can reach before its turn
            next_stamp(g);
            int32_t st = g->stamp;
            Dv.clear();
            auto addv = [&](int32_t x) { if (g->vstamp[x] != st) { g->vstamp[x] = st; Dv.push_back(x); } };
            for (int32_t x : B) addv(x);
            work.assign(centres.begin(), centres.end());
            size_t wi = 0;
            seen_owner.clear();
            while (wi < work.size() && Dv.size() <= DCAP) {
                int32_t c = work[wi++];
                for_owners(c, [&](int32_t oid) {
                    bool seen = false;
                    for (int32_t so : seen_owner) seen |= (so == oid);
                    if (seen) return;
                    seen_owner.push_back(oid);
                    const spg_graph::Owner ow = g->owners[oid];
                    for (int32_t yi = 0; yi < ow.len; yi++) {
                        int32_t y = owner_set(g, ow)[yi];   // (re-resolved: work/addv never touch the pools)
                        bool fresh = g->vstamp[y] != st;
                        addv(y);
                        // Dense: a newly reachable removable vertex is itself absorbed and brings its neighbourhood
                        if (dense && fresh && g->in_set[y] && g->valive[y]) work.push_back(y);
                    }
                });
                if (dense && c != v) {
                    // neighbourhood of an absorbed vertex (stamps are in use: gather without closed_neighbourhood)
                    for (int32_t eid : g->vr[c].adj) {
                        const GEdge &e = g->edges[eid];
                        for (int i = 0; i < e.nv; i++) {
                            int32_t y = edge_verts(g, e)[i];
                            bool fresh = g->vstamp[y] != st;
                            addv(y);
                            if (fresh && g->in_set[y] && g->valive[y]) work.push_back(y);
                        }
                    }
                }
            }
            if (Dv.size() > DCAP) { stop = true; continue; }  // (v is already in newpending; the loop ends here)
            {
                int32_t off = (int32_t)g->Dpool.size();
                g->Dpool.insert(g->Dpool.end(), Dv.begin(), Dv.end());
                g->transient.push_back(reg(-1, off, (int32_t)Dv.size()));
            }
            // stop scanning once a long run of list entries had to wait: whatever follows is
            // (almost always) waiting on them too, and not scanning only defers more
            // (with another batch in flight the blocked stretch is usually exactly the part of the list
            //  that waits for it: give up sooner, the next call comes right after that batch commits)
            static const int pat_inflight = [] { const char *e = getenv("SPG_PATIENCE"); return e ? atoi(e) : 16; }();
            size_t patience = inflight ? pat_inflight + bt.rb.size() / 16 : 48 + bt.rb.size() / 8;
            if (++consec > patience || n_deferred > 256 + 2 * bt.rb.size()) stop = true;
            PT(3);
        }
    }
    {
        size_t nd = newpending.size();
        size_t nh = pos - nd;
        for (size_t i = 0; i < nd; i++) g->pending[nh + i] = newpending[i];
        g->pend_head = nh;
    }
}

// ================================================================================= rounds
extern "C" int spg_graph_marginalize_begin(spg_graph *g, const int32_t *which, int n, const spg_options *opts, int rank, int nranks) {
    if (!g || !opts || (n > 0 && !which) || nranks < 1 || rank < 0 || rank >= nranks) return SPG_EINVAL;
    if (g->active) return set_err(g->ctx, SPG_ESTATE, "marginalize already in progress");
    if (opts->pose_dim != g->d) return set_err(g->ctx, SPG_EINVAL, "pose_dim mismatch");
    g->opts = *opts;
    g->rank = rank; g->nranks = nranks;
    g->pending.clear();
    g->pend_head = 0;
    g->in_set.assign(g->vid.size(), 0);
    g->lpos.assign(g->vid.size(), -1);
    for (int i = 0; i < n; i++) {
        const int32_t vi = g->index_of(which[i]);
        if (vi < 0 || !g->valive[vi])
            return set_err(g->ctx, SPG_EINVAL, "vertex needs to exist in order to be marginalized");
        if (g->in_set[vi]) continue;
        g->in_set[vi] = 1;
        g->lpos[vi] = (int32_t)g->pending.size();
        g->pending.push_back(vi);
    }
    // keys of the edges this call creates: after everything that exists, ordered by list position of their root
    g->key_base = g->next_key;
    g->next_key = g->key_base + ((int64_t)g->pending.size() + 1) * spg_graph::kKeyStride;
    // room for the regions of the rounds to come (grown later if this estimate is short). An arena that was reserved for
    // the job (spg_graph_reserve: 2.5x or more of what is in use) is left alone: growing means a new allocation, a device
    // synchronisation and the whole graph uploaded again (1.2 ms of a 19 ms marginalisation of the 100k-pose graph)
    if (g->cap < g->used * 5 / 2 + (1 << 16)) if (int rc = arena_ensure(g, g->used * 3 + (1 << 20))) return rc;
    if (int rc = sync_device(g)) return rc;
    g->active = true;
    for (int i = 0; i < spg_graph::NB; i++) { g->bt[i].round_open = false; g->bt[i].slot = i; }
    g->B = &g->bt[0];
    g->round_no = 0;
    g->launch_seq = 0;
    g->pipelined = false;
    g->stats = spg_marg_stats{};
    g->log.clear();
    return 0;
}

static int prepare_scheduled(spg_graph *g, spg_round_info *info, double t0);

extern "C" int spg_graph_set_shard_threshold(spg_graph *g, int min_blankets) {
    if (!g) return SPG_EINVAL;
    g->shard_threshold = min_blankets < 0 ? -1 : min_blankets;
    return 0;
}

// Sharding policy for one batch of mutually independent blankets (all ranks evaluate it on identical data, so
// they agree). Model, per GPU: a blanket of n = d*k target variables is one dependent chain of
//     t_b = T24 * max(1, n/24)^2.5     (T24 = 45 us: measured chain of a 24 x 24 blanket)
// and `cap` of them are resident at a time (LDS carve-up: floor(160 KB / tiles) workgroups per CU, at most 6, on
// 256 CUs; blankets whose tiles live in the L2 workspace: one per CU), so a batch takes
//     t_local = max(max_b t_b, sum_b t_b / cap_b),
// and sharded over nr ranks  t_shard = max(max_b t_b, sum_b t_b / (nr cap_b)) + T_x + bytes / BW_x
// with one all-gather of T_x = 30 us (small-message RCCL latency over xGMI) and BW_x = 100 GB/s towards each rank.
// Sharding pays iff t_shard < t_local: wide batches of many blankets; narrow ones (a few hundred blankets finish in
// one chain latency however they are split) are computed redundantly by every rank.
static bool shard_pays(const spg_graph *g, const Batch &bt) {
    const int B = (int)bt.rb.size(), nr = g->nranks;
    if (nr <= 1 || B == 0) return false;
    if (g->shard_threshold >= 0) return B >= g->shard_threshold;
    const int d = g->d;
    double t_sum = 0, t_max = 0, bytes = 0;
    for (const RoundBlanket &r : bt.rb) {
        const double n = (double)d * (r.nv - r.n_remove), nm = (double)d * r.n_remove;
        const double tb = 45e-6 * std::pow(std::max(1.0, n / 24.0), 2.5);
        const double lds = 8.0 * (3 * n * n + nm * nm + nm * n) + 4096.0;
        const double per_cu = lds > 160.0 * 1024 ? 1.0 : std::min(6.0, std::floor(160.0 * 1024 / lds));
        t_sum += tb / (256.0 * per_cu);
        t_max = std::max(t_max, tb);
        int32_t nn, nvv; int64_t nl;
        new_edge_budget(g->opts, d, r.nv - r.n_remove, nn, nvv, nl);
        bytes += 8.0 * (double)(SPG_OUT_LEN(nn, nvv) + nl);
    }
    const double t_local = std::max(t_max, t_sum);
    const double t_shard = std::max(t_max, t_sum / nr) + 30e-6 + bytes / 100e9;
    return t_shard < t_local;
}

extern "C" int spg_graph_round_prepare(spg_graph *g, spg_round_info *info) {
    if (!g || !g->active) return SPG_ESTATE;
    Batch &bt = *g->B;
    if (bt.round_open) return SPG_ESTATE;
    double t0 = now_s();
    schedule_round(g);
    g->stats.schedule_seconds += now_s() - t0;
    return prepare_scheduled(g, info, t0);
}

// descriptors + arena region for the blankets already selected into *g->B
static int prepare_scheduled(spg_graph *g, spg_round_info *info, double t0) {
    Batch &bt = *g->B;
    int B = (int)bt.rb.size();
    if (B == 0) { g->stats.host_seconds += now_s() - t0; return 0; }
    const spg_options &o = g->opts;
    // small rounds are latency-bound: every rank computes them whole, nothing is exchanged
    const bool sharded = shard_pays(g, bt);
    bt.eff_ranks = sharded ? g->nranks : 1;
    bt.eff_rank = sharded ? g->rank : 0;
    const int d = g->d, nr = bt.eff_ranks;
    // ---- contiguous, cost-balanced slices (cost ~ n^3 + E d^3)
    std::vector<double> &cost = g->s_cost;
    cost.assign(B, 0.0);
    double total = 0;
    for (int b = 0; b < B; b++) {
        RoundBlanket &r = bt.rb[b];
        double nn = (double)d * (r.nv - r.n_remove);
        cost[b] = nn * nn * nn + (double)r.ne * d * d * d + 1.0;
        total += cost[b];
    }
    std::vector<int> &first = g->s_first;
    first.assign(nr + 1, B);
    {
        double acc = 0;
        int q = 0;
        first[0] = 0;
        for (int b = 0; b < B; b++) {
            while (q + 1 < nr && acc >= total * (q + 1) / nr) first[++q] = b;
            acc += cost[b];
        }
        for (int qq = q + 1; qq <= nr; qq++) first[qq] = B;
        first[nr] = B;
    }
    // ---- descriptors
    bt.h_blk.resize(B);
    bt.h_vpo.clear(); bt.h_er.clear(); bt.h_ev.clear();
    bt.chunk_hdr.assign(nr, 0);
    std::vector<int64_t> &chunk_len = g->s_chunk_len;
    chunk_len.assign(nr, 0);
    if (g->lidx.size() < g->vid.size()) g->lidx.resize(g->vid.size(), -1);
    std::vector<int32_t> &lidx = g->lidx;
    for (int q = 0; q < nr; q++) {
        int64_t hdr = 0, body = 0;
        for (int b = first[q]; b < first[q + 1]; b++) {
            RoundBlanket &r = bt.rb[b];
            r.rank = q;
            int k = r.nv - r.n_remove;
            const int32_t *rverts = bt.rb_verts.data() + r.vbeg;
            const int32_t *redges = bt.rb_edges.data() + r.ebeg;
            spg_blanket_desc &bd = bt.h_blk[b];
            memset(&bd, 0, sizeof bd);
            bd.vert_begin = (int32_t)bt.h_vpo.size();
            bd.n_vert = r.nv;
            bd.n_remove = r.n_remove;
            for (int i = 0; i < r.nv; i++) { bt.h_vpo.push_back(g->vpose[rverts[i]]); lidx[rverts[i]] = (int32_t)i; }
            bd.edge_begin = (int32_t)bt.h_er.size();
            bd.n_edge = r.ne;
            int32_t scratch = 0;
            for (int ei_ = 0; ei_ < r.ne; ei_++) {
                int32_t eid = redges[ei_];
                const GEdge &e = g->edges[eid];
                if (e.kind == SPG_EDGE_GLC) scratch = std::max(scratch, e.len - d * e.nv + e.nv * 2 * d * d);
                spg_edge_ref er;
                er.off = e.off; er.len = e.len; er.kind = e.kind; er.vbegin = (int32_t)bt.h_ev.size(); er.nv = e.nv;
                for (int i = 0; i < e.nv; i++) bt.h_ev.push_back(lidx[edge_verts(g, e)[i]]);
                bt.h_er.push_back(er);
            }
            new_edge_budget(o, d, k, bd.n_new_max, bd.n_new_vert_max, bd.new_len);
            bd.pad_ = scratch;  // doubles of assembly scratch the blanket's n-ary edges need (r*dq + q*2*d*d)
            bd.out_off = hdr;  // relative for now
            hdr += SPG_OUT_LEN(bd.n_new_max, bd.n_new_vert_max);
            bd.new_off = body;
            body += bd.new_len;
            bd.tinfo_off = -1;
        }
        bt.chunk_hdr[q] = hdr;
        chunk_len[q] = hdr + body;
    }
    int64_t clen = 0;
    for (int q = 0; q < nr; q++) clen = std::max(clen, chunk_len[q]);
    clen = align_up(std::max<int64_t>(clen, 1), 32);
    int64_t region = align_up(g->used, 32);
    int64_t need = region + clen * nr;
    if (need > g->cap) {
        // grow: pull device-only ranges into the mirror, re-allocate, push the whole mirror back.
        // Not while another batch is running (it writes into the arena): tell the driver to commit it first.
        for (int bi = 0; bi < spg_graph::NB; bi++) if (&g->bt[bi] != &bt && g->bt[bi].round_open) return 2;
        if (int rc = arena_ensure(g, need + need / 2)) return rc;
        if (int rc = sync_device(g)) return rc;
    }
    for (int q = 0; q < nr; q++) {
        int64_t base = region + clen * q;
        for (int b = first[q]; b < first[q + 1]; b++) {
            spg_blanket_desc &bd = bt.h_blk[b];
            bd.out_off += base;
            bd.new_off += base + bt.chunk_hdr[q];
            bt.rb[b].desc = bd;
        }
    }
    if ((int64_t)g->host.size() < need) g->host.resize((size_t)need);
    g->used = need;
    g->dev_synced = need;  // the region is produced on the device
    bt.rinfo.n_blankets = B;
    bt.rinfo.my_first = first[bt.eff_rank];
    bt.rinfo.my_count = first[bt.eff_rank + 1] - first[bt.eff_rank];
    bt.rinfo.region_off = region;
    bt.rinfo.chunk_len = clen;
    bt.rinfo.exchange = sharded ? 1 : 0;
    bt.rinfo.pad_ = 0;
    if (info) *info = bt.rinfo;
    bt.round_open = true;
    g->stats.n_batches++;
    g->round_no++;
    bt.round_no = g->round_no;
    bt.seq = g->launch_seq++;
    g->stats.host_seconds += now_s() - t0;
    return 1;
}

// ---- the same preparation split in two for the pipelined single-rank driver ----------------------------------
// (1) graph thread: size and allocate the batch's output region (needs g->used / the arena), open the round;
// (2) submission thread: write the descriptors (fill_descriptors_single) and hand the batch to the device.
// Returns 1 = prepared, 0 = empty, 2 = the arena has to grow first (nothing may be in flight), < 0 error.
static int prepare_region_single(spg_graph *g, Batch &bt, double t0) {
    const int B = (int)bt.rb.size();
    if (B == 0) { g->stats.host_seconds += now_s() - t0; return 0; }
    const spg_options &o = g->opts;
    const int d = g->d;
    int64_t hdr = 0, body = 0;
    for (RoundBlanket &r : bt.rb) {
        // the part of the descriptor the commit needs is computed here, by the thread that commits (the submission
        // thread derives the same numbers for its own copy and never writes into bt.rb)
        spg_blanket_desc &bd = r.desc;
        new_edge_budget(o, d, r.nv - r.n_remove, bd.n_new_max, bd.n_new_vert_max, bd.new_len);
        bd.out_off = hdr;          // relative for now
        bd.new_off = body;
        hdr += SPG_OUT_LEN(bd.n_new_max, bd.n_new_vert_max);
        body += bd.new_len;
        r.rank = 0;
    }
    const int64_t clen = align_up(std::max<int64_t>(hdr + body, 1), 32);
    const int64_t region = align_up(g->used, 32), need = region + clen;
    if (need > g->cap) {
        for (int bi = 0; bi < spg_graph::NB; bi++) if (&g->bt[bi] != &bt && g->bt[bi].round_open) return 2;
        if (int rc = arena_ensure(g, need + need / 2)) return rc;
        if (int rc = sync_device(g)) return rc;
    }
    if ((int64_t)g->host.size() < need) {
        // the graph thread copies out records into the mirror while the submission thread works: grow it generously
        // here, never under a commit (HostMirror::resize may move the block; only this thread touches it)
        g->host.resize((size_t)std::max<int64_t>(need, g->cap));
    }
    g->used = need;
    g->dev_synced = need;
    for (RoundBlanket &r : bt.rb) { r.desc.out_off += region; r.desc.new_off += region + hdr; }
    bt.eff_ranks = 1; bt.eff_rank = 0;
    bt.chunk_hdr.assign(1, hdr);
    bt.rinfo.n_blankets = B; bt.rinfo.my_first = 0; bt.rinfo.my_count = B;
    bt.rinfo.region_off = region; bt.rinfo.chunk_len = clen; bt.rinfo.exchange = 0; bt.rinfo.pad_ = 0;
    bt.round_open = true;
    g->stats.n_batches++;
    g->round_no++;
    bt.round_no = g->round_no;
    bt.seq = g->launch_seq++;
    g->stats.host_seconds += now_s() - t0;
    return 1;
}

// (2) descriptors of a batch prepared by prepare_region_single. Reads only what the graph thread never changes
// while the batch is open: poses' offsets, the location / endpoints of existing edges, the batch's own lists.
static void fill_descriptors_single(spg_graph *g, Batch &bt) {
    const int B = (int)bt.rb.size(), d = g->d;
    const spg_options &o = g->opts;
    bt.h_blk.resize(B);
    bt.h_vpo.clear(); bt.h_er.clear(); bt.h_ev.clear();
    if (g->lidx.size() < g->vid.size()) g->lidx.resize(g->vid.size(), -1);
    std::vector<int32_t> &lidx = g->lidx;
    const int64_t base = bt.rinfo.region_off, hdr_total = bt.chunk_hdr[0];
    int64_t hdr = 0, body = 0;
    for (int b = 0; b < B; b++) {
        const RoundBlanket &r = bt.rb[b];
        const int k = r.nv - r.n_remove;
        const int32_t *rverts = bt.rb_verts.data() + r.vbeg;
        const int32_t *redges = bt.rb_edges.data() + r.ebeg;
        spg_blanket_desc &bd = bt.h_blk[b];
        memset(&bd, 0, sizeof bd);
        bd.vert_begin = (int32_t)bt.h_vpo.size();
        bd.n_vert = r.nv;
        bd.n_remove = r.n_remove;
        for (int i = 0; i < r.nv; i++) { bt.h_vpo.push_back(g->vpose[rverts[i]]); lidx[rverts[i]] = (int32_t)i; }
        bd.edge_begin = (int32_t)bt.h_er.size();
        bd.n_edge = r.ne;
        int32_t scratch = 0;
        for (int ei_ = 0; ei_ < r.ne; ei_++) {
            const GEdge &e = g->edges[redges[ei_]];
            if (e.kind == SPG_EDGE_GLC) scratch = std::max(scratch, e.len - d * e.nv + e.nv * 2 * d * d);
            spg_edge_ref er;
            er.off = e.off; er.len = e.len; er.kind = e.kind; er.vbegin = (int32_t)bt.h_ev.size(); er.nv = e.nv;
            for (int i = 0; i < e.nv; i++) bt.h_ev.push_back(lidx[edge_verts(g, e)[i]]);
            bt.h_er.push_back(er);
        }
        new_edge_budget(o, d, k, bd.n_new_max, bd.n_new_vert_max, bd.new_len);
        bd.pad_ = scratch;
        bd.out_off = base + hdr;
        hdr += SPG_OUT_LEN(bd.n_new_max, bd.n_new_vert_max);
        bd.new_off = base + hdr_total + body;
        body += bd.new_len;
        bd.tinfo_off = -1;
    }
}

static int harvest_kld(spg_graph *g, Batch &bt);
// graph-thread half of spg_graph_round_compute: launch tag, late results of the slot's previous launch
static int compute_prologue(spg_graph *g, Batch &bt) {
    bt.tag = ++g->ctx->tag_counter;
    return harvest_kld(g, bt);
}
// submission-thread half: the round descriptor and the hand-over to the backend
static int compute_submit(spg_graph *g, Batch &bt) {
    spg_round_desc rd{};
    rd.opts = &g->opts;
    rd.n_blankets = bt.rinfo.n_blankets;
    rd.first = bt.rinfo.my_first;
    rd.count = bt.rinfo.my_count;
    rd.blankets = bt.h_blk.data();
    rd.vert_pose_off = bt.h_vpo.data();
    rd.edges = bt.h_er.data();
    rd.edge_vert = bt.h_ev.data();
    rd.n_vert_total = (int64_t)bt.h_vpo.size();
    rd.n_edge_total = (int64_t)bt.h_er.size();
    rd.n_edge_vert_total = (int64_t)bt.h_ev.size();
    rd.mail_base = bt.rinfo.region_off + bt.rinfo.chunk_len * bt.eff_rank;
    rd.mail_len = (bt.eff_ranks == 1 && g->ctx->be.mailbox) ? bt.chunk_hdr[bt.eff_rank] : 0;
    rd.slot = g->pipelined ? bt.slot : 0;
    rd.tag = bt.tag;
    bt.used_mailbox = rd.mail_len > 0;
    bt.t_launch = now_s();
    return g->ctx->be.run_round(g->ctx->be.user, g->dev, &rd);
}

static void submission_main(spg_graph *g) {
    uint32_t idle = 0;
    while (g->sub.run.load(std::memory_order_acquire)) {
        const uint32_t h = g->sub.head.load(std::memory_order_relaxed);
        if (h == g->sub.tail.load(std::memory_order_acquire)) {
            if (++idle > 4096) std::this_thread::yield();
#if defined(__x86_64__)
            else __builtin_ia32_pause();
#endif
            continue;
        }
        idle = 0;
        Batch *b = g->sub.q[h % spg_graph::SUBQ];
        const double t0 = now_s();
        fill_descriptors_single(g, *b);
        b->submit_rc = compute_submit(g, *b);
        g->sub.seconds += now_s() - t0;
        g->sub.head.store(h + 1, std::memory_order_release);
        b->submitted.store(1, std::memory_order_release);
    }
}
static void submission_start(spg_graph *g) {
    if (g->sub_active) return;
    g->sub.head.store(0); g->sub.tail.store(0);
    g->sub.seconds = 0;
    g->sub.run.store(true, std::memory_order_release);
    g->sub_thread = std::thread(submission_main, g);
    g->sub_active = true;
#if defined(__linux__)
    // keep the two threads on neighbouring cores (same L3): the lists one writes and the other reads then move
    // through the shared cache instead of across the socket. Best effort; only CPUs this process may use.
    static const bool pin = [] { const char *e = getenv("SPG_PIN_THREADS"); return !(e && e[0] == '0'); }();
    if (pin) {
        int cpu = sched_getcpu();
        cpu_set_t allowed;
        if (cpu >= 0 && sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
            for (int cand : {cpu ^ 1, cpu + 1, cpu - 1}) {
                if (cand >= 0 && cand < CPU_SETSIZE && cand != cpu && CPU_ISSET(cand, &allowed)) {
                    cpu_set_t one;
                    CPU_ZERO(&one); CPU_SET(cand, &one);
                    (void)pthread_setaffinity_np(g->sub_thread.native_handle(), sizeof one, &one);
                    break;
                }
            }
        }
    }
#endif
}
static void wait_submitted(Batch &b) {
    uint32_t spins = 0;
    while (!b.submitted.load(std::memory_order_acquire)) {
        if (++spins > 4096) std::this_thread::yield();
#if defined(__x86_64__)
        else __builtin_ia32_pause();
#endif
    }
}
static void quiesce_submission(spg_graph *g) {
    for (int i = 0; i < spg_graph::NB; i++) wait_submitted(g->bt[i]);
}
static void submission_stop(spg_graph *g) {
    if (!g->sub_active) return;
    quiesce_submission(g);
    g->sub.run.store(false, std::memory_order_release);
    g->sub_thread.join();
    g->sub_active = false;
    g->stats.launch_seconds += g->sub.seconds;   // (the thread's time is reported as launch_seconds)
}
static void submission_push(spg_graph *g, Batch &b) {
    b.submitted.store(0, std::memory_order_relaxed);
    b.submit_rc = 0;
    const uint32_t t = g->sub.tail.load(std::memory_order_relaxed);
    g->sub.q[t % spg_graph::SUBQ] = &b;
    g->sub.tail.store(t + 1, std::memory_order_release);
}

// Late results of a batch that was committed by polling: once its launch has completed, pick up the
// per-blanket KLD (and a possible SPG_ST_KLD_NOT_PD) from the mailbox.
static int harvest_kld(spg_graph *g, Batch &bt) {
    if (bt.kld_pending.empty()) return 0;
    const bool slotted = g->ctx->be.synchronize_slot && g->ctx->be.mailbox_slot;
    int rc = slotted ? g->ctx->be.synchronize_slot(g->ctx->be.user, bt.slot) : g->ctx->be.synchronize(g->ctx->be.user);
    if (rc) return rc;
    const double *mail = slotted ? g->ctx->be.mailbox_slot(g->ctx->be.user, bt.slot) : g->ctx->be.mailbox(g->ctx->be.user);
    for (auto &pr : bt.kld_pending) {
        const double *rec = mail + pr.second;
        BlanketLog &lg = g->log[pr.first];
        lg.kld = rec[2];
        lg.min_gap = rec[3];
        lg.status = (int32_t)rec[0];
        if (std::isfinite(rec[2])) g->stats.kld_sum += rec[2];
    }
    bt.kld_pending.clear();
    return 0;
}

extern "C" int spg_graph_round_compute(spg_graph *g) {
    if (!g || !g->active) return SPG_ESTATE;
    Batch &bt = *g->B;
    if (!bt.round_open) return SPG_ESTATE;
    double t0 = now_s();
    // the mailbox of this slot is about to be rewritten: collect what the previous launch left in it
    if (int hrc = compute_prologue(g, bt)) return hrc;
    int rc = compute_submit(g, bt);
    g->stats.device_seconds += now_s() - t0;
    g->stats.launch_seconds += now_s() - t0;
    if (rc && g->ctx->is_hip) snprintf(g->ctx->err, sizeof g->ctx->err, "%s", spg::hip_backend_error(&g->ctx->be));
    return rc;
}

extern "C" int spg_graph_round_commit(spg_graph *g) {
    if (!g || !g->active) return SPG_ESTATE;
    Batch &bt = *g->B;
    if (!bt.round_open) return SPG_ESTATE;
    double t0 = now_s();
    wait_submitted(bt);   // (pipelined driver: the submission thread may still be handing the batch over)
    if (bt.submit_rc) {
        if (g->ctx->is_hip) snprintf(g->ctx->err, sizeof g->ctx->err, "%s", spg::hip_backend_error(&g->ctx->be));
        return bt.submit_rc;
    }
    const int nr = bt.eff_ranks;
    const bool slotted = g->pipelined && g->ctx->be.synchronize_slot && g->ctx->be.mailbox_slot;
    const double *mail = nullptr;
    if (bt.used_mailbox && g->ctx->be.mailbox)
        mail = slotted ? g->ctx->be.mailbox_slot(g->ctx->be.user, bt.slot) : g->ctx->be.mailbox(g->ctx->be.user);
    int rc = 0;
    bool polled = false;
    if (mail && nr == 1) {
        // Poll the ready tags the kernel writes (system-scope release) after each blanket's graph-update
        // data is complete; the launch itself may still be finishing KLD tails. Bounded spin: after
        // ~5 s fall back to a stream synchronisation, which also surfaces a faulted kernel.
        const int64_t base = bt.rinfo.region_off + bt.rinfo.chunk_len * bt.eff_rank;
        const double want = SPG_READY_WORD(bt.tag), want_final = SPG_FINAL_WORD(bt.tag);
        const double t_spin = now_s();
        // (SPG_POLL_SPIN_S: tests set 0 to force the synchronisation path on ordinary batches)
        const char *sl_env = getenv("SPG_POLL_SPIN_S");
        const double spin_limit = sl_env ? atof(sl_env) : 5.0;
        polled = true;
        bool first_seen = false;
        for (const RoundBlanket &r : bt.rb) {
            const volatile double *flag = mail + (r.desc.out_off - base) + 5;
            if (first_seen == false && &r != &bt.rb[0]) { g->tr_first += now_s() - bt.t_launch; first_seen = true; }
            uint32_t spins = 0;
            for (double fv = *flag; fv != want && fv != want_final; fv = *flag) {
                if ((++spins & 0x3fff) == 0 && now_s() - t_spin > spin_limit) { polled = false; break; }
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            if (!polled) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!polled) {
        rc = slotted ? g->ctx->be.synchronize_slot(g->ctx->be.user, bt.slot) : g->ctx->be.synchronize(g->ctx->be.user);
        if (rc) return rc;
        if (mail && nr == 1) {
            // the launch has completed: every record must carry this launch's tag now. One that does not was never written
            // (a blanket no kernel took, a kernel that died): an error, never a graph update from whatever the cell held.
            const int64_t base = bt.rinfo.region_off + bt.rinfo.chunk_len * bt.eff_rank;
            const double want = SPG_READY_WORD(bt.tag), want_final = SPG_FINAL_WORD(bt.tag);
            for (const RoundBlanket &r : bt.rb) {
                const double fv = mail[(r.desc.out_off - base) + 5];
                if (fv != want && fv != want_final)
                    return set_err(g->ctx, SPG_EHIP, "the blanket of vertex %s did not deliver its out record although its launch has completed", std::to_string(g->vid[r.root]).c_str());
            }
        }
    }
    g->tr_n++; g->tr_wait += now_s() - t0; g->tr_age += t0 - bt.t_launch;
    // read back the out-record part of every rank chunk (mailbox: already in host memory)
    for (int q = 0; q < nr; q++) {
        if (bt.chunk_hdr[q] == 0) continue;
        int64_t base = bt.rinfo.region_off + bt.rinfo.chunk_len * q;
        if (mail && q == bt.eff_rank) { memcpy(g->host.data() + base, mail, (size_t)bt.chunk_hdr[q] * 8); continue; }
        rc = g->ctx->be.download(g->ctx->be.user, g->host.data() + base, (char *)g->dev + base * 8, bt.chunk_hdr[q]);
        if (rc) return rc;
    }
    double t1 = now_s();
    g->stats.device_seconds += t1 - t0;
    // the payload part of the region stays device-only until someone asks for it
    {
        int64_t lo = bt.rinfo.region_off, hi = bt.rinfo.region_off + bt.rinfo.chunk_len * nr;
        if (g->stale_hi <= g->stale_lo) { g->stale_lo = lo; g->stale_hi = hi; }
        else { g->stale_lo = std::min(g->stale_lo, lo); g->stale_hi = std::max(g->stale_hi, hi); }
    }
    // updateInputGraph (src/vertex_remover.cpp:500-546), in list order
    std::vector<int32_t> &vix = g->s_vix;
    for (size_t b = 0; b < bt.rb.size(); b++) {
        RoundBlanket &r = bt.rb[b];
        const spg_blanket_desc &bd = r.desc;
        const double *rec = g->host.data() + bd.out_off;
        int status = (int)rec[0], inf = (int)rec[1], n_new = (int)rec[4];
        PT0;
        g->log.push_back({g->vid[r.root], bt.round_no, status, inf, rec[2], rec[3]});
        if (polled && (status == SPG_OK)) bt.kld_pending.push_back({(int32_t)g->log.size() - 1, bd.out_off - (bt.rinfo.region_off + bt.rinfo.chunk_len * bt.eff_rank)});
        g->stats.max_blanket = std::max(g->stats.max_blanket, r.nv);
        const int32_t *rverts = bt.rb_verts.data() + r.vbeg;
        const int32_t *redges = bt.rb_edges.data() + r.ebeg;
        bool fine = (status == SPG_OK || status == SPG_ST_KLD_NOT_PD);
        if (!fine) { g->stats.n_bad_status++; continue; }
        if (!polled && std::isfinite(rec[2])) g->stats.kld_sum += rec[2];
        PT(7);
        for (int ei_ = 0; ei_ < r.ne; ei_++) {
            int32_t eid = redges[ei_];
            GEdge &e = g->edges[eid];
            e.alive = 0;
            g->n_mutations++;
            g->n_live_e--;
            for (int i = 0; i < e.nv; i++) {
                auto &av = g->vr[edge_verts(g, e)[i]].adj;
                for (size_t j = 0; j < av.size(); j++) if (av[j] == eid) { av[j] = av.back(); av.pop_back(); break; }
            }
        }
        PT(8);
        for (int i = 0; i < r.n_remove; i++) {
            int32_t v = rverts[i];
            g->valive[v] = 0;
            g->vr[v].adj.clear();
            g->n_live_v--;
            g->stats.n_removed++;
        }
        PT(9);
        int vpos = 0;
        // (the record comes from the device: nothing in it is used as an index before it has been checked)
        {
            bool sane = n_new >= 0 && n_new <= bd.n_new_max;
            int vsum = 0, why = sane ? 0 : 1;
            double badv = 0;
            for (int e = 0; sane && e < n_new; e++) {
                const double nvd = rec[SPG_OUT_HDR + 4 * e + 3], reld = rec[SPG_OUT_HDR + 4 * e + 1], lend = rec[SPG_OUT_HDR + 4 * e + 2];
                sane = nvd >= 1 && nvd <= r.nv && reld >= 0 && lend >= 1 && reld + lend <= (double)bd.new_len;
                if (!sane) { why = 2; badv = (double)bd.new_len; }
                if (sane) { for (int i = 0; i < (int)nvd && sane; i++) { const double li = rec[SPG_OUT_HDR + 4 * bd.n_new_max + vsum + i]; sane = li >= 0 && li < r.nv; if (!sane) { why = 3; badv = li + 1e-3 * i; } } vsum += (int)nvd; }
                if (sane && vsum > bd.n_new_vert_max) { sane = false; why = 4; badv = bd.n_new_vert_max; }
            }
            if (!sane) {
                char msg[256];
                snprintf(msg, sizeof msg, "out record of the blanket of vertex %d is not well formed (status %d, %d new edges of at most %d, k + m = %d, m = %d; words %g %g %g %g %g %g | %g %g %g %g)", g->vid[r.root], status, n_new, bd.n_new_max, r.nv, r.n_remove,
                         rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], rec[6], rec[7], rec[8], rec[9]);
                return set_err(g->ctx, SPG_EHIP, "%s", msg);
            }
        }
        for (int e = 0; e < n_new; e++) {
            int kind = (int)rec[SPG_OUT_HDR + 4 * e + 0];
            int64_t rel = (int64_t)rec[SPG_OUT_HDR + 4 * e + 1];
            int32_t len = (int32_t)rec[SPG_OUT_HDR + 4 * e + 2];
            int nv = (int)rec[SPG_OUT_HDR + 4 * e + 3];
            vix.resize(nv);
            for (int i = 0; i < nv; i++) vix[i] = rverts[(int)rec[SPG_OUT_HDR + 4 * bd.n_new_max + vpos + i]];
            vpos += nv;
            add_edge_idx(g, kind, nv, vix.data(), bd.new_off + rel, len, g->key_base + (int64_t)g->lpos[r.root] * spg_graph::kKeyStride + e);
            g->stats.n_new_edges++;
        }
        PT(10);
    }
    bt.round_open = false;
    release_batch_owners(g, bt);
    g->stats.n_rounds = g->round_no;
    g->stats.host_seconds += now_s() - t1;
    g->stats.commit_seconds += now_s() - t1;
    return 0;
}

extern "C" int spg_graph_marginalize_end(spg_graph *g, spg_marg_stats *stats) {
    if (!g || !g->active) return SPG_ESTATE;
    submission_stop(g);
    g->active = false;
    for (int i = 0; i < spg_graph::NB; i++) (void)harvest_kld(g, g->bt[i]);
    for (int i = 0; i < spg_graph::NB; i++) { g->bt[i].round_open = false; release_batch_owners(g, g->bt[i]); g->bt[i].rb.clear(); }
    for (int32_t oid : g->transient) owner_release(g, oid);
    g->transient.clear();
    g->B = &g->bt[0];
    if (g->ctx->is_hip) (void)spg::hip_backend_end_of_call(&g->ctx->be);
    if (g->tr_n && getenv("SPG_TRACE"))
        fprintf(stderr, "spg trace: %ld batches; per batch: launch call -> commit start %.1f us, wait for the ready words %.1f us (launch call -> first blanket ready %.1f us)\n",
                g->tr_n, 1e6 * g->tr_age / g->tr_n, 1e6 * g->tr_wait / g->tr_n, 1e6 * g->tr_first / g->tr_n);
    g->tr_n = 0; g->tr_age = g->tr_wait = g->tr_first = 0;
    if (g->ctx->is_hip) g->stats.n_launches = spg::hip_backend_launches(&g->ctx->be);
#ifdef SPG_SCHED_PROF
    if (getenv("SPG_SCHED_PROF")) {
        const char *nm[12] = {"neighbourhood", "owner scan", "select:reg", "defer", "-", "select:sort", "select:edges", "commit:log", "commit:rm edges", "commit:rm verts", "commit:add", "prepare"};
        for (int i = 0; i < 12; i++) { fprintf(stderr, "sched %-14s %10llu calls %8.3f Mcycles\n", nm[i], prof_n[i], prof_t[i] * 1e-6); prof_t[i] = prof_n[i] = 0; }
    }
#endif
    if (stats) *stats = g->stats;
    return g->stats.n_bad_status ? SPG_EBLANKET : 0;
}

// Move blankets [from, to) of a freshly scheduled batch into another (idle) batch: all of them are
// mutually independent, so the parts can be launched back to back on different slots.
static void move_blankets(spg_graph *g, Batch &a, size_t from, size_t to, Batch &b) {
    b.rb.clear(); b.rb_verts.clear(); b.rb_edges.clear();
    const int32_t bidx = (int32_t)(&b - g->bt);
    for (size_t i = from; i < to; i++) {
        RoundBlanket r = a.rb[i];
        int32_t vb = (int32_t)b.rb_verts.size(), eb = (int32_t)b.rb_edges.size();
        b.rb_verts.insert(b.rb_verts.end(), a.rb_verts.begin() + r.vbeg, a.rb_verts.begin() + r.vbeg + r.nv);
        b.rb_edges.insert(b.rb_edges.end(), a.rb_edges.begin() + r.ebeg, a.rb_edges.begin() + r.ebeg + r.ne);
        r.vbeg = vb; r.ebeg = eb;
        if (r.owner >= 0) { g->owners[r.owner].batch = bidx; g->owners[r.owner].off = vb; }
        b.rb.push_back(r);
    }
}


// ================================================================================= streaming driver
// The batch driver above hands the device ~50 blankets at a time and commits them as a unit; on graphs whose removals
// form long dependent chains (ring lattices: 250 rounds of 200) the host then idles while a batch is in flight and the
// device idles while the host commits and selects. Here ONE blanket is the unit: it is handed to the persistent worker
// kernel as one queue item the moment the rule below allows it, and committed the moment its ready word arrives,
// whatever else is in flight. Nothing is rescanned: a vertex that cannot go yet is parked on the ONE event that blocks
// it (an earlier vertex being launched, or being committed) and looked at again when that event happens.
//
// Rule. Positions are indices into the removal list; a vertex is WAITING, INFLIGHT (handed over, not committed) or DONE
// (committed: the host graph holds its effect). In the current host graph G, with X = N[v], WAITING v may be launched iff
//   (A) no vertex of X \ {v} is an earlier list entry that is not DONE (and none is INFLIGHT at all);
//   (B) for every x in X: every in-flight blanket that contains x contains no other vertex of X (and none contains v);
//       and no neighbour of any x in X \ {v} is an earlier WAITING list entry;
//   (C) no in-flight blanket of (B) that belongs to an earlier list entry contains a WAITING list entry earlier than v.
// Why this is the sequential result (src/vertex_remover.cpp:83-140 removes in list order): two removals commute when
// neither centre is in the other's blanket and the blankets share at most one vertex (header of this file). Take the
// earlier not-DONE entries in list order and assume the first one, u, whose blanket AT ITS TURN meets X in a vertex it
// does not meet now. The edge u - x it needs is created by a still earlier not-DONE removal whose blanket holds u and
// x; that one meets X, so it is one of the in-flight blankets seen in (B); u is in it, earlier than v and not INFLIGHT
// (an INFLIGHT vertex had no earlier not-DONE neighbour, (A)) — which (C) excludes. Hence the earlier removals that ever
// touch X are exactly the in-flight ones of (B), with frozen blankets sharing one vertex with X, and nothing earlier
// touches v, so X and its edges are what they will be at v's turn. (A vertex in flight was launched under the same
// rule, so later entries that run before v were checked against v from their side.)
// The earliest WAITING entry is only ever blocked by in-flight blankets, so the stream always makes progress.
// tests/test_stream_scheduler.py drives this code on the CPU with adversarial completion orders against the oracle.
#if defined(__x86_64__)
#include <x86intrin.h>
static inline uint64_t ticks_now() { return __rdtsc(); }
#else
static inline uint64_t ticks_now() { return (uint64_t)(now_s() * 1e9); }
#endif

namespace {
enum : uint8_t { SV_WAITING = 0, SV_STABLE = 1, SV_INFLIGHT = 2, SV_DONE = 3 };
constexpr int kStreamSlots = 2048;       // blankets in flight at most (the worker has 256 workgroups; the rest queue)
constexpr int kStreamMailStride = 8;     // doubles per mailbox cell: the compact out record of the worker (flags bit 20, spg_kernels.hip publish()) is one cache line

struct Streamer {
    spg_graph *g;
    spg::StreamPort port;
    const bool emulate;
    const int D, ps, rec;
    const int32_t P;                     // list positions
    int32_t cursor = 0;                  // positions below it have been examined at least once
    int32_t prefetched_to = 0;
    int32_t n_done = 0, n_inflight = 0, pending_bell = 0, bell_no = 0;
    bool fallback = false;               // a blanket the worker does not take (or the arena is full): drain, then the batch driver
    int rc = 0;
    uint32_t launch_seq = 0, poll_seq = 0;  // HIP port: blankets launched so far; the oldest launch whose result has not been taken
    std::vector<int32_t> cell_slot;         // mailbox cell -> slot of the blanket whose record it holds / will hold, -1 = free, -2 = taken (committed, not harvested)
    size_t fifo_head = 0;
    int32_t cap_wait = -1;               // positions parked for a free slot
    uint64_t rng;
    double alg_bytes = 0;
    uint64_t t_idle = 0;
    void *arena_dev;

    Streamer(spg_graph *g_, bool emu) : g(g_), emulate(emu), D(g_->d), ps(g_->ps), rec(g_->rec), P((int32_t)g_->pending.size()), arena_dev(g_->dev) {}

    struct SvView { spg_graph::VRec *vr; spg_graph::SVtx &operator[](int32_t i) const { return vr[i].s; } };

    long n_exam = 0, n_park[8] = {0};
    // SPG_STREAM_PROF=1: TSC ticks per phase (poll, commit, examine -> parked, examine -> launch decision, packet, doorbell, late results)
    const bool prof = [] { const char *e = getenv("SPG_STREAM_PROF"); return e && e[0] == '1'; }();
    uint64_t pt[8] = {0}, pn[8] = {0}, pt_last = 0;
    inline void P0() { if (prof) pt_last = ticks_now(); }
    inline void P1(int i) { if (prof) { const uint64_t n = ticks_now(); pt[i] += n - pt_last; pn[i]++; pt_last = n; } }
    void park(std::vector<int32_t> &heads, int32_t on, int32_t p) { g->wl_next[p] = heads[on]; heads[on] = p; }
    void wake(std::vector<int32_t> &heads, int32_t on) {
        for (int32_t p = heads[on]; p >= 0;) { const int32_t nx = g->wl_next[p]; g->s_woken.push_back(p); p = nx; }
        heads[on] = -1;
    }

    // index of y in X[0 .. nX) or -1. (Branch-free forms of this search, of the two small sorts and of the list removals were
    // measured on the bench workload: every one of them slower than the early-exit loops — 19.4 -> 20.6 -> 22.7 ms per step.)
    static inline int find(const int32_t *X, const int nX, const int32_t y) { for (int i = 0; i < nX; i++) if (X[i] == y) return i; return -1; }
    static inline void adj_remove(InlVec<spg_graph::AdjEnt, 7> &av, const int32_t eid) {
        for (size_t j = 0; j < av.size(); j++) if (av[j].eid == eid) { av[j] = av.back(); av.pop_back(); break; }
    }
    inline void set_pend(spg_graph::SSlot &sl, const int32_t *X, const int nX) {
        const int32_t *const cst = g->cst.data();
        sl.npend = 0;
        for (int i = 1; i < nX; i++) if (cst[X[i]] >= 0) { if (sl.npend == 3) { sl.npend = -1; break; } sl.pend[sl.npend++] = X[i]; }
    }

    // v passed (A): its blanket X is final as a vertex set. If it has to wait all the same, the set is registered like
    // the blanket of a launched vertex (a reservation), so that later entries are checked against it instead of waiting
    // for it; the slot becomes the blanket's own when it is launched. Without a free slot the vertex simply stays WAITING.
    void reserve(const int32_t p, const int32_t *X, const int nX) {
        const SvView sv{g->vr.data()};
        const int32_t v = X[0];
        if ((g->cst[v] & 3) != SV_WAITING || g->s_free.size() < 64) return;
        for (int i = 0; i < nX; i++) if (sv[X[i]].nown == spg_graph::kSOwn) return;
        const int32_t s = g->s_free.back(); g->s_free.pop_back();
        spg_graph::SSlot &sl = g->sslots[s];
        sl.pos = p; sl.root = v; sl.nv = nX; sl.ne = 0; sl.launched = 0; sl.logi = -1;
        memcpy(sl.verts, X, sizeof(int32_t) * (size_t)nX);
        set_pend(sl, X, nX);
        for (int i = 0; i < nX; i++) { spg_graph::SVtx &sx = sv[X[i]]; sx.own[sx.nown++] = s; }
        g->cst[v] = (p << 2) | SV_STABLE; sv[v].slot = s;
        if (g->wl_stable[p] >= 0) wake(g->wl_stable, p);
    }

    // The queue item of the blanket in slot s (layout: spg_kernels.hip, blanket_worker), written through the BAR, and its
    // item word; `ahead` = items written since the last doorbell. Reads the slot, the poses' offsets and the blanket
    // edges' records' locations (all immutable once the slot is handed over).
    void build_packet(const int32_t s, const int ahead) {
        const spg_graph::SSlot &sl = g->sslots[s];
        const spg_graph::VRec *const vr = g->vr.data();
        const GEdge *const edges = g->edges.data();
        const int nX = sl.nv, ne = sl.ne, n_new_max = sl.n_new_max;
        const int words = spg::kPktHdr + nX + 4 * ne;
        unsigned long long pkt[spg::kPktWords];
        auto pack = [](int lo, int hi) { return (unsigned long long)(uint32_t)lo | ((unsigned long long)(uint32_t)hi << 32); };
        const spg_options &o = g->opts;
        pkt[0] = (unsigned long long)(uintptr_t)arena_dev;
        pkt[1] = port.d_mail;
        pkt[2] = 0;
        pkt[3] = (unsigned long long)((int64_t)sl.mcell * port.mail_stride); pkt[4] = (unsigned long long)sl.new_off; pkt[5] = (unsigned long long)(int64_t)-1;
        pkt[6] = pack(nX, 1); pkt[7] = pack(ne, n_new_max); pkt[8] = pack(2 * n_new_max, 0);
        pkt[9] = pack(o.topology, o.flags | (1 << 20)); pkt[10] = pack(o.lin_point, sl.tag);   // bit 20: compact out record
        memcpy(&pkt[11], &o.chord_ratio, 8);
        pkt[12] = pack(words, 2 * ne);
        int w = spg::kPktHdr;
        for (int i = 0; i < nX; i++) pkt[w++] = (unsigned long long)vr[sl.verts[i]].pose;
        int32_t *evp = (int32_t *)(pkt + spg::kPktHdr + nX + 3 * ne);
        double by = 8.0 * ps * nX + 12.0 + 8.0 * (double)n_new_max * rec;
        for (int i = 0; i < ne; i++) {
            const GEdge &e = edges[sl.edges[i]];
            spg_edge_ref er;
            er.off = e.off; er.len = e.len; er.kind = e.kind; er.vbegin = 2 * i; er.nv = 2;
            memcpy(&pkt[w], &er, 24);
            w += 3;
            evp[2 * i] = find(sl.verts, nX, e.vtx[0]); evp[2 * i + 1] = find(sl.verts, nX, e.vtx[1]);
            by += 8.0 + 8.0 * e.len;
        }
        alg_bytes += by;
        unsigned long long *dst = port.pkt + (size_t)s * spg::kPktWords;
        memcpy(dst, pkt, (size_t)words * 8);                                                  // through the BAR (write-combined)
        port.q->item[(port.tail + (unsigned long long)ahead) % spg::kQCap] = (unsigned long long)(uintptr_t)dst;
    }

    int examine(const int32_t p) {
        P0();
        const int r = examine_impl(p);
        P1(r ? 7 : 2);
        return r;
    }
    // 1 = launched, 0 = parked / nothing to do
    int examine_impl(const int32_t p) {
        const SvView sv{g->vr.data()};
        int32_t *const cst = g->cst.data();
        const int32_t v = g->pending[p];
        const int vstate = cst[v] & 3;
        if (vstate != SV_WAITING && vstate != SV_STABLE) return 0;
        n_exam++;
        spg_graph::VRec *const vr = g->vr.data();
        const GEdge *const edges = g->edges.data();
        const spg_graph::SSlot *const slots = g->sslots.data();
        int32_t X[spg_graph::kSMaxV];
        int nX = 1;
        int32_t mine = -1;                       // the slot of v's reservation
        if (vstate == SV_STABLE) {
            mine = sv[v].slot;
            nX = slots[mine].nv;
            memcpy(X, slots[mine].verts, sizeof(int32_t) * (size_t)nX);
        } else {
            X[0] = v;
            for (const spg_graph::AdjEnt &a : vr[v].adj) {
                const int32_t u = a.other;
                if (u < 0) { fallback = true; return 0; }   // an n-ary edge: not for the worker
                if (find(X, nX, u) < 0) {
                    if (nX == spg_graph::kSMaxV) { fallback = true; return 0; }
                    X[nX++] = u;
                }
            }
            const int k0 = nX - 1;
            if (k0 < 1 || D * k0 > spg::kWorkerMaxN) { fallback = true; return 0; }
            // (A)
            int32_t blocker = -1;   // of several blockers the LAST list entry: it is the one that finishes last, as a rule, and a wake-up by any other only parks v again
            for (int i = 1; i < nX; i++) {
                const int32_t c = cst[X[i]];
                // (a DONE entry that is still in the graph kept a status that forbids the graph update: it is inert)
                if (c >= 0 && (c & 3) != SV_DONE && ((c >> 2) < p || (c & 3) == SV_INFLIGHT)) blocker = std::max(blocker, c >> 2);
            }
            {
                if (blocker >= 0) {
                    n_park[0]++;
                    park(g->wl_done, blocker, p);
                    // first look at v (the list cursor runs well ahead of the results): pull what its launch will read — its
                    // neighbours' records and its edges' records, cold in DRAM until now — towards the shared cache
                    if (p >= prefetched_to) {
                        prefetched_to = p + 1;
                        __builtin_prefetch((const char *)&vr[v] + 64);
                        for (const spg_graph::AdjEnt &a : vr[v].adj) {
                            __builtin_prefetch(&edges[a.eid]);
                            __builtin_prefetch(&vr[a.other]);
                            __builtin_prefetch((const char *)&vr[a.other] + 64);
                        }
                    }
                    return 0;
                }
            }
            // kept vertices in ascending id (buildSubgraph's order, src/vertex_remover.cpp:349-356)
            for (int i = 2; i < nX; i++) {
                const int32_t x = X[i], idx = vr[x].id;
                int j = i - 1;
                for (; j >= 1 && vr[X[j]].id > idx; j--) X[j + 1] = X[j];
                X[j + 1] = x;
            }
        }
        const int k = nX - 1;
        // (B), registered blankets (in flight, or reserved by an earlier entry that is itself waiting), and (C)
        for (int j = 0; j < sv[v].nown; j++) if (sv[v].own[j] != mine) { n_park[1]++; park(g->wl_done, slots[sv[v].own[j]].pos, p); return 0; }
        int32_t hs[spg_graph::kSMaxV * spg_graph::kSOwn];
        int nh = 0;
        for (int i = 1; i < nX; i++) {
            const spg_graph::SVtx &sx = sv[X[i]];
            for (int j = 0; j < sx.nown; j++) {
                const int32_t s = sx.own[j];
                if (s == mine) continue;
                if (!slots[s].launched && slots[s].pos > p) continue;   // a later entry's reservation: it is checked against v, not v against it
                for (int h = 0; h < nh; h++) if (hs[h] == s) { n_park[2]++; reserve(p, X, nX); park(g->wl_done, slots[s].pos, p); return 0; }
                hs[nh++] = s;
            }
        }
        {
            int32_t blocker = -1;
            for (int h = 0; h < nh; h++) {
                const spg_graph::SSlot &o = slots[hs[h]];
                if (o.pos > p) continue;
                const int32_t *mem = o.npend >= 0 ? o.pend : o.verts + 1;
                const int nmem = o.npend >= 0 ? o.npend : o.nv - 1;
                for (int i = 0; i < nmem; i++) {
                    const int32_t c = cst[mem[i]];
                    if (c >= 0 && (c & 3) == SV_WAITING && (c >> 2) < p) { blocker = std::max(blocker, o.pos); break; }
                }
            }
            if (blocker >= 0) { n_park[3]++; reserve(p, X, nX); park(g->wl_done, blocker, p); return 0; }
        }
        // (B), entries without a final blanket, fused with markovBlanketEdges (src/vertex_remover.cpp:225-251): one pass over
        // the adjacency of X \ {v}; no edge record is read (the far endpoints are in the adjacency entries) and no
        // per-vertex record of a vertex outside X (list positions and states come from the compact array)
        int32_t E[spg_graph::kSMaxE];
        int ne = 0;
        for (int i = 1; i < nX; i++) {
            for (const spg_graph::AdjEnt &a : vr[X[i]].adj) {
                const int32_t y = a.other;
                if (y < 0) { fallback = true; return 0; }
                const int j = find(X, nX, y);
                if (j >= 0) {
                    if (j == 0 || j >= i) {   // every blanket edge once: from its kept end, or from the lower-numbered of two kept ends
                        if (ne == spg_graph::kSMaxE) { fallback = true; return 0; }
                        E[ne++] = a.eid;
                    }
                } else {
                    const int32_t c = cst[y];
                    if (c >= 0 && (c & 3) == SV_WAITING && (c >> 2) < p) { n_park[4]++; reserve(p, X, nX); park(g->wl_stable, c >> 2, p); return 0; }
                }
            }
        }
        const int words = spg::kPktHdr + nX + 4 * ne;
        if (words > spg::kPktWords) { fallback = true; return 0; }
        if (mine < 0) {
            for (int i = 0; i < nX; i++) if (sv[X[i]].nown == spg_graph::kSOwn) { park(g->wl_done, slots[sv[X[i]].own[0]].pos, p); return 0; }
            if (g->s_free.empty()) { g->wl_next[p] = cap_wait; cap_wait = p; return 0; }
        }
        if (!emulate && cell_slot[launch_seq & (uint32_t)(port.slots - 1)] != -1) { g->wl_next[p] = cap_wait; cap_wait = p; return 0; }   // the next mailbox cell still holds an unharvested record
        // ---- launch
        const int n_new_max = k - 1, n_new_vert_max = 2 * (k - 1);
        const int64_t new_len = (int64_t)n_new_max * rec, out_len = emulate ? SPG_OUT_LEN(n_new_max, n_new_vert_max) : 0;
        if (g->used + new_len + out_len > g->cap) { fallback = true; return 0; }
        // ascending key = the reference's sequential edge order
        for (int i = 1; i < ne; i++) {
            const int32_t eid = E[i]; const int64_t key = edges[eid].key;
            int j = i - 1;
            for (; j >= 0 && edges[E[j]].key > key; j--) E[j + 1] = E[j];
            E[j + 1] = eid;
        }
        int32_t s = mine;
        if (s < 0) { s = g->s_free.back(); g->s_free.pop_back(); }
        spg_graph::SSlot &sl = g->sslots[s];
        sl.pos = p; sl.root = v; sl.nv = nX; sl.ne = ne; sl.n_new_max = n_new_max; sl.logi = -1; sl.bell = bell_no + 1; sl.launched = 1;
        sl.tag = (++g->ctx->tag_counter & 0x3fffffff) + 1;
        sl.mcell = (int32_t)(launch_seq & (uint32_t)(port.slots - 1));
        if (!emulate) { cell_slot[sl.mcell] = s; launch_seq++; }
        sl.out_off = emulate ? g->used : -1;
        sl.new_off = g->used + out_len;
        g->used += new_len + out_len;
        memcpy(sl.edges, E, sizeof(int32_t) * (size_t)ne);
        if (mine < 0) {
            memcpy(sl.verts, X, sizeof(int32_t) * (size_t)nX);
            set_pend(sl, X, nX);
            for (int i = 0; i < nX; i++) { spg_graph::SVtx &sx = sv[X[i]]; sx.own[sx.nown++] = s; }
        }
        cst[v] = (p << 2) | SV_INFLIGHT; sv[v].slot = s;
        P1(3);
        if (!emulate) build_packet(s, pending_bell);
        if (helper) hp.cell_tag[sl.mcell] = (uint32_t)sl.tag;
        P1(4);
        pending_bell++;
        n_inflight++;
        if (emulate) g->s_fifo.push_back(s);
        if (g->wl_stable[p] >= 0) wake(g->wl_stable, p);
        return 1;
    }

    void ring() {
        if (!pending_bell) return;
        P0();
        if (!emulate) {
            std::atomic_thread_fence(std::memory_order_release);
#if defined(__x86_64__)
            __builtin_ia32_sfence();   // packets and item words have left the write-combining buffers before the doorbell
#endif
            port.tail += (unsigned long long)pending_bell;
            for (int c = 0; c < port.bells; c++) port.q->tail[c * spg::kBellStride] = port.tail;
#if defined(__x86_64__)
            __builtin_ia32_sfence();
#endif
        }
        pending_bell = 0;
        bell_no++;
        g->stats.n_batches++;
        if (helper) hp.launches.store(launch_seq, std::memory_order_release);
        P1(5);
    }

    // updateInputGraph (src/vertex_remover.cpp:500-546) for the blanket in slot s, whose out record is `recd`
    void commit(const int32_t s, const double *recd, const bool final_seen) {
        const SvView sv{g->vr.data()};
        spg_graph::SSlot &sl = g->sslots[s];
        const int32_t p = sl.pos, v = sl.root;
        const int status = (int)recd[0], inf = (int)recd[1], n_new = (int)recd[4];
        sl.logi = (int32_t)g->log.size();
        g->log.push_back({g->vid[v], sl.bell, status, inf, recd[2], recd[3]});
        g->stats.max_blanket = std::max(g->stats.max_blanket, sl.nv);
        const bool fine = (status == SPG_OK || status == SPG_ST_KLD_NOT_PD);
        if (!fine) g->stats.n_bad_status++;
        else {
            for (int i = 0; i < sl.ne; i++) {
                const int32_t eid = sl.edges[i];
                GEdge &e = g->edges[eid];
                e.alive = 0;
                adj_remove(g->vr[e.vtx[0]].adj, eid);
                if (e.vtx[1] != e.vtx[0]) adj_remove(g->vr[e.vtx[1]].adj, eid);
            }
            g->n_mutations += sl.ne + 1;
            g->n_live_e -= sl.ne;
            g->valive[v] = 0;
            g->vr[v].adj.clear();
            g->n_live_v--;
            g->stats.n_removed++;
            const int64_t key0 = g->key_base + (int64_t)p * spg_graph::kKeyStride;
            if (!emulate) {
                // compact record: endpoint pairs as 4-bit local indices, edge e in byte e of words [6] / [7]; records of `rec` doubles back to back
                uint64_t w[2];
                memcpy(w, recd + 6, 16);
                for (int e = 0; e < n_new; e++) {
                    const unsigned pr = (unsigned)(w[e >> 3] >> (8 * (e & 7))) & 0xffu;
                    const int32_t va = sl.verts[pr & 15u], vb = sl.verts[pr >> 4];
                    // (add_edge_idx for a pose-pose edge between two different vertices, without its general-case checks)
                    GEdge ge;
                    ge.off = sl.new_off + (int64_t)e * rec; ge.key = key0 + e; ge.len = rec; ge.vtx[0] = va; ge.vtx[1] = vb; ge.nv = 2; ge.kind = SPG_EDGE_BINARY; ge.alive = 1;
                    const int32_t eid = (int32_t)g->edges.size();
                    g->edges.push_back(ge);
                    g->vr[va].adj.push_back({eid, vb});
                    if (vb != va) g->vr[vb].adj.push_back({eid, va});
                }
            } else {
                int vpos = 0;
                for (int e = 0; e < n_new; e++) {
                    const int kind = (int)recd[SPG_OUT_HDR + 4 * e + 0];
                    const int64_t rel = (int64_t)recd[SPG_OUT_HDR + 4 * e + 1];
                    const int32_t len = (int32_t)recd[SPG_OUT_HDR + 4 * e + 2];
                    const int nv = (int)recd[SPG_OUT_HDR + 4 * e + 3];
                    int32_t vix[2] = {sl.verts[(int)recd[SPG_OUT_HDR + 4 * sl.n_new_max + vpos]], sl.verts[(int)recd[SPG_OUT_HDR + 4 * sl.n_new_max + vpos + 1]]};
                    vpos += nv;
                    add_edge_idx(g, kind, 2, vix, sl.new_off + rel, len, key0 + e);
                }
            }
            g->stats.n_new_edges += n_new;
            if (!emulate) { g->n_mutations += n_new; g->n_live_e += n_new; }
        }
        for (int i = 0; i < sl.nv; i++) {
            spg_graph::SVtx &sx = sv[sl.verts[i]];
            for (int j = 0; j < sx.nown; j++) if (sx.own[j] == s) { sx.own[j] = sx.own[--sx.nown]; break; }
        }
        g->cst[v] = (p << 2) | SV_DONE;
        n_done++;
        n_inflight--;
        if (final_seen) harvest(s, recd);
        else g->s_fin.push_back(s);
        if (g->wl_done[p] >= 0) wake(g->wl_done, p);
    }
    // the KLD tail of a committed blanket has landed (final word): late results, then the slot is free again
    void harvest(const int32_t s, const double *recd) {
        spg_graph::SSlot &sl = g->sslots[s];
        BlanketLog &lg = g->log[sl.logi];
        lg.kld = recd[2]; lg.min_gap = recd[3]; lg.status = (int32_t)recd[0];
        if (std::isfinite(recd[2])) g->stats.kld_sum += recd[2];
        if (!emulate) cell_slot[sl.mcell] = -1;
        g->s_free.push_back(s);
        for (int32_t q = cap_wait; q >= 0;) { const int32_t nx = g->wl_next[q]; g->s_woken.push_back(q); q = nx; }
        cap_wait = -1;
    }

    inline const double *cell(int32_t s) const { return port.h_mail + (size_t)g->sslots[s].mcell * (size_t)port.mail_stride; }

    // Blankets whose ready (or final) word has arrived. Mailbox cells are handed out in launch order and tickets are served
    // in that order, so results land nearly in sequence in sequential memory: the scan starts at the oldest launch not
    // taken yet and stops after `giveup` unfinished cells in a row (every 16th call looks at everything in flight).
    void poll_hip(int giveup) {
        std::vector<int32_t> &ready = g->s_ready;
        const uint32_t mask = (uint32_t)(port.slots - 1);
        while (poll_seq != launch_seq && cell_slot[poll_seq & mask] < 0) poll_seq++;
        int misses = 0;
        for (uint32_t q0 = poll_seq; q0 != launch_seq && misses < giveup;) {
            // the ready words of the next 16 cells are loaded before any is looked at: the lines the device has just written
            // miss the caches, and behind a branch per cell those misses would be taken one after the other
            const uint32_t nq = std::min<uint32_t>(16, launch_seq - q0);
            double w[16];
            for (uint32_t i = 0; i < nq; i++) w[i] = ((const volatile double *)(port.h_mail + (size_t)((q0 + i) & mask) * (size_t)port.mail_stride))[5];
            for (uint32_t i = 0; i < nq; i++) {
                const uint32_t q = q0 + i;
                const int32_t s = cell_slot[q & mask];
                if (s < 0) continue;
                const double tagd = (double)g->sslots[s].tag;
                if (w[i] == 4503599627370496.0 + tagd || w[i] == 4503599627370496.0 + 4294967296.0 + tagd) {
                    const double *c = port.h_mail + (size_t)(q & mask) * (size_t)port.mail_stride;
                    (void)c;
                    __builtin_prefetch(&g->sslots[s]); __builtin_prefetch((const char *)&g->sslots[s] + 64); __builtin_prefetch((const char *)&g->sslots[s] + 128);
                    ready.push_back(s);
                    cell_slot[q & mask] = -2;
                    misses = 0;
                } else misses++;
            }
            q0 += nq;
        }
        // the ready words the next call will look at first: on their way while this call's results are committed (a cell the
        // device has not written yet comes in stale and is invalidated by the write; one it has written is a hit next time)
        {
            uint32_t q = poll_seq;
            for (int n = 0; q != launch_seq && n < 12; q++) {
                if (cell_slot[q & mask] < 0) continue;
                __builtin_prefetch((const void *)(port.h_mail + (size_t)(q & mask) * (size_t)port.mail_stride));
                n++;
            }
        }
    }

    // Emulated device (injected backend; tests): "complete" a subset of the in-flight blankets, chosen and ordered by the
    // seed, by running them as one round of the backend; their out records land in the arena.
    int poll_emulated() {
        std::vector<int32_t> &fifo = g->s_fifo, &ready = g->s_ready;
        std::vector<int32_t> live;
        for (size_t i = fifo_head; i < fifo.size(); i++) if (fifo[i] >= 0) live.push_back((int32_t)i);
        if (live.empty()) { fifo.clear(); fifo_head = 0; return 0; }
        auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
        size_t take = live.size();
        if (g->stream_emulation > 0) {
            take = 1 + (size_t)(next() % live.size());
            for (size_t i = 0; i + 1 < live.size(); i++) std::swap(live[i], live[i + (size_t)(next() % (live.size() - i))]);
        }
        live.resize(take);
        std::vector<spg_blanket_desc> blk(take);
        std::vector<int64_t> vpo;
        std::vector<spg_edge_ref> er;
        std::vector<int32_t> ev;
        for (size_t t = 0; t < take; t++) {
            const spg_graph::SSlot &sl = g->sslots[fifo[live[t]]];
            spg_blanket_desc &bd = blk[t];
            memset(&bd, 0, sizeof bd);
            bd.vert_begin = (int32_t)vpo.size(); bd.n_vert = sl.nv; bd.n_remove = 1;
            bd.edge_begin = (int32_t)er.size(); bd.n_edge = sl.ne;
            bd.n_new_max = sl.n_new_max; bd.n_new_vert_max = 2 * sl.n_new_max;
            bd.new_off = sl.new_off; bd.new_len = (int64_t)sl.n_new_max * rec; bd.out_off = sl.out_off; bd.tinfo_off = -1;
            for (int i = 0; i < sl.nv; i++) { vpo.push_back(g->vpose[sl.verts[i]]); g->lidx[sl.verts[i]] = i; }
            for (int i = 0; i < sl.ne; i++) {
                const GEdge &e = g->edges[sl.edges[i]];
                spg_edge_ref r; r.off = e.off; r.len = e.len; r.kind = e.kind; r.vbegin = (int32_t)ev.size(); r.nv = 2;
                ev.push_back(g->lidx[e.vtx[0]]); ev.push_back(g->lidx[e.vtx[1]]);
                er.push_back(r);
            }
        }
        spg_round_desc rd{};
        rd.opts = &g->opts; rd.n_blankets = (int32_t)take; rd.first = 0; rd.count = (int32_t)take;
        rd.blankets = blk.data(); rd.vert_pose_off = vpo.data(); rd.edges = er.data(); rd.edge_vert = ev.data();
        rd.n_vert_total = (int64_t)vpo.size(); rd.n_edge_total = (int64_t)er.size(); rd.n_edge_vert_total = (int64_t)ev.size();
        rd.mail_base = 0; rd.mail_len = 0; rd.slot = 0; rd.tag = ++g->ctx->tag_counter;
        if (int r = g->ctx->be.run_round(g->ctx->be.user, g->dev, &rd)) return r;
        if (int r = g->ctx->be.synchronize(g->ctx->be.user)) return r;
        for (size_t t = 0; t < take; t++) {
            const int32_t s = fifo[live[t]];
            const spg_graph::SSlot &sl = g->sslots[s];
            const int64_t olen = SPG_OUT_LEN(sl.n_new_max, 2 * sl.n_new_max);
            if (int r = g->ctx->be.download(g->ctx->be.user, g->host.data() + sl.out_off, (char *)g->dev + sl.out_off * 8, olen)) return r;
            ready.push_back(s);
            fifo[live[t]] = -1;
        }
        while (fifo_head < fifo.size() && fifo[fifo_head] < 0) fifo_head++;
        return 0;
    }

    // ---- poll helper (on by default for lists of 4096 entries or more; SPG_STREAM_THREADS=1 switches it off): a second host thread does nothing but watch the
    // mailbox and copy each record that has arrived — one cache line — into a ring of ordinary memory, so that the graph
    // thread reads results from the neighbouring core's cache instead of taking a miss on device-written memory per poll.
    // It touches no graph data: all it needs is the number of launches so far and the tag each cell will show.
    struct Helper {
        static constexpr uint32_t RQ = 4096;
        alignas(64) std::atomic<uint32_t> launches{0};   // graph thread: cells [0, launches) have been handed out (their tags are in cell_tag)
        alignas(64) std::atomic<uint32_t> r_tail{0};     // helper: results published
        alignas(64) std::atomic<int> stop{0};
        alignas(64) uint32_t r_head = 0;                  // graph thread
        std::vector<double> rq;                           // RQ entries of 8 doubles: the compact record, word [5] = 2 * launch number + final flag
        std::vector<uint32_t> cell_tag;
        char pad_[64];
    } hp;
    bool helper = false;
    std::thread helper_thread;
    std::vector<const double *> ready_rec;               // helper mode: ring entry of each slot in g->s_ready

    void helper_main() {
        const uint32_t mask = (uint32_t)(port.slots - 1);
        std::vector<uint8_t> pend((size_t)port.slots, 0);
        uint32_t head = 0, seen = 0, r_tail = 0;
        unsigned n = 0;
        const double READY = 4503599627370496.0, FINAL = 4503599627370496.0 + 4294967296.0;
        while (!hp.stop.load(std::memory_order_acquire)) {
            const uint32_t lp = hp.launches.load(std::memory_order_acquire);
            while (seen != lp) { pend[seen & mask] = 1; seen++; }
            while (head != seen && !pend[head & mask]) head++;
            const int giveup = (++n & 15) ? 10 : 1 << 20;
            int misses = 0;
            bool got = false;
            for (uint32_t q0 = head; q0 != seen && misses < giveup;) {
                const uint32_t nq = std::min<uint32_t>(16, seen - q0);
                double w[16];
                for (uint32_t i = 0; i < nq; i++) w[i] = ((const volatile double *)(port.h_mail + (size_t)((q0 + i) & mask) * (size_t)port.mail_stride))[5];
                for (uint32_t i = 0; i < nq; i++) {
                    const uint32_t q = q0 + i;
                    if (!pend[q & mask]) continue;
                    const double tagd = (double)hp.cell_tag[q & mask];
                    if (w[i] == READY + tagd || w[i] == FINAL + tagd) {
                        double *e = hp.rq.data() + (size_t)(r_tail & (Helper::RQ - 1)) * 8;
                        memcpy(e, port.h_mail + (size_t)(q & mask) * (size_t)port.mail_stride, 64);
                        e[5] = (double)(2.0 * (double)q + (w[i] == FINAL + tagd ? 1.0 : 0.0));
                        r_tail++;
                        pend[q & mask] = 0;
                        misses = 0;
                        got = true;
                    } else misses++;
                }
                q0 += nq;
            }
            if (got) hp.r_tail.store(r_tail, std::memory_order_release);
        }
    }
    void take_helper_results() {
        const uint32_t rt = hp.r_tail.load(std::memory_order_acquire);
        const uint32_t mask = (uint32_t)(port.slots - 1);
        ready_rec.clear();
        while (hp.r_head != rt) {
            const double *e = hp.rq.data() + (size_t)(hp.r_head & (Helper::RQ - 1)) * 8;
            __builtin_prefetch(e + 8); __builtin_prefetch(e + 16);
            const uint32_t q = (uint32_t)((uint64_t)e[5] >> 1);
            const int32_t s = cell_slot[q & mask];
            g->s_ready.push_back(s);
            ready_rec.push_back(e);
            cell_slot[q & mask] = -2;
            __builtin_prefetch(&g->sslots[s]); __builtin_prefetch((const char *)&g->sslots[s] + 64); __builtin_prefetch((const char *)&g->sslots[s] + 128);
            hp.r_head++;
        }
    }
    // The helper is only used when it can sit on a core that shares an L3 with this thread's (cores of a group of 8 do on
    // the hosts this runs on) and both can be pinned for the duration of the call: across L3s the ring costs more than the
    // polls it saves. Returns false (nothing started) otherwise.
    bool helper_start() {
#if defined(__linux__)
        static const bool pin = [] { const char *e = getenv("SPG_PIN_THREADS"); return !(e && e[0] == '0'); }();
        if (!pin || sched_getaffinity(0, sizeof old_mask, &old_mask) != 0) return false;
        const int cpu = sched_getcpu();
        int cand = -1;
        for (int d = 1; cpu >= 0 && d < 8 && cand < 0; d++) {
            const int c = (cpu & ~7) | ((cpu + d) & 7);
            if (c < CPU_SETSIZE && CPU_ISSET(c, &old_mask)) cand = c;
        }
        if (cand < 0) return false;
        cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpu, &one);
        if (sched_setaffinity(0, sizeof one, &one) != 0) return false;
        repin = true;
        hp.rq.resize((size_t)Helper::RQ * 8);
        hp.cell_tag.assign((size_t)port.slots, 0);
        helper = true;
        helper_thread = std::thread([this] { helper_main(); });
        CPU_ZERO(&one); CPU_SET(cand, &one);
        (void)pthread_setaffinity_np(helper_thread.native_handle(), sizeof one, &one);
        return true;
#else
        return false;
#endif
    }
    void helper_stop() {
        if (!helper) return;
        hp.stop.store(1, std::memory_order_release);
        helper_thread.join();
        helper = false;
#if defined(__linux__)
        if (repin) (void)sched_setaffinity(0, sizeof old_mask, &old_mask);
#endif
    }
#if defined(__linux__)
    cpu_set_t old_mask;
    bool repin = false;
#endif

    int run() {
        const uint64_t t_begin = ticks_now();
        const double s_begin = now_s();
        uint64_t last_progress = t_begin, idle_since = 0;
        unsigned n_polls = 0;
        std::vector<int32_t> &woken = g->s_woken, &ready = g->s_ready, &fin = g->s_fin;
        size_t fin_head = 0;
        for (;;) {
            // ---- results
            ready.clear();
            P0();
            if (emulate) { if ((rc = poll_emulated()) != 0) return rc; }
            else if (n_inflight) {
                if (helper) take_helper_results(); else poll_hip((++n_polls & 15) ? 10 : 1 << 20);
            }
            P1(0);
            const bool got = !ready.empty();
            if (got && idle_since) { t_idle += ticks_now() - idle_since; idle_since = 0; }
            // every result that has arrived is committed before anything is examined: a woken entry often waits for two or
            // three of them (its column's predecessor and that one's neighbours), and looking at it between their commits
            // only parks it again
            for (size_t ri = 0; ri < ready.size(); ri++) {
                const int32_t s = ready[ri];
                const double *recd = emulate ? g->host.data() + g->sslots[s].out_off : (helper ? ready_rec[ri] : cell(s));
                const bool fin_now = emulate || (helper ? (((uint64_t)recd[5]) & 1) != 0 : recd[5] == 4503599627370496.0 + 4294967296.0 + (double)g->sslots[s].tag);
                P0();
                commit(s, recd, fin_now);
                P1(1);
            }
            if (!fallback) {
                // woken entries are examined in list order: an earlier one that launches (or gets its final blanket) is often
                // what a later one of the same wake-up waits for — the wait lists hand them out newest first
                if (woken.size() > 1) std::sort(woken.begin(), woken.end());
                for (size_t wi = 0; wi < woken.size(); wi++) {
                    examine(woken[wi]);
                    if (pending_bell >= 8) ring();
                }
            }
            woken.clear();
            // ---- late results (final words) of committed blankets, in bulk: each is a line the device has rewritten since the
            // commit read it (a miss), nothing waits for them, and taken 64 at a time the misses overlap
            if (!emulate && (fin.size() - fin_head >= 192 || g->s_free.size() < 256 || (cap_wait >= 0 && fin_head < fin.size()))) {
                P0();
                const size_t n = std::min<size_t>(64, fin.size() - fin_head);
                for (size_t i = 0; i < n; i++) __builtin_prefetch((const void *)cell(fin[fin_head + i]));
                for (size_t i = 0; i < n; i++) {
                    const int32_t s = fin[fin_head];
                    const volatile double *c = cell(s);
                    if (c[5] != 4503599627370496.0 + 4294967296.0 + (double)g->sslots[s].tag) break;
                    harvest(s, (const double *)c);
                    fin_head++;
                }
                if (fin_head > 8192) { fin.erase(fin.begin(), fin.begin() + (long)fin_head); fin_head = 0; }
                P1(6);
                if (!woken.empty()) {
                    if (!fallback) for (size_t wi = 0; wi < woken.size(); wi++) examine(woken[wi]);
                    woken.clear();
                }
            }
            // ---- list entries nobody has looked at yet: a few per turn, more when the device leaves the host idle
            if (!fallback && cursor < P) {
                int budget = got ? 4 : 32;
                while (cursor < P && budget-- > 0 && !fallback) {
                    examine(cursor++);
                    if (!woken.empty()) { for (size_t wi = 0; wi < woken.size() && !fallback; wi++) examine(woken[wi]); woken.clear(); }
                }
            }
            ring();
            if (n_done == P) break;
            if (fallback && n_inflight == 0) break;
            if (got || cursor < P) { last_progress = ticks_now(); continue; }
            // ---- nothing arrived and nothing to examine: the host waits for the device
            {
                const uint64_t t1 = ticks_now();
                if (!idle_since) idle_since = t1;
                if (n_inflight == 0 && fin_head == fin.size()) {
                    // nothing in flight, list not exhausted, nothing woken: every remaining entry is parked on an entry that never
                    // ran — cannot happen (the earliest WAITING entry only waits for blankets in flight); leave through the batch driver
                    fallback = true;
                    break;
                }
                if (((t1 - last_progress) >> 35) != 0) {   // ~ 10 s without a result
                    set_err(g->ctx, SPG_EHIP, "streaming driver: no blanket completed within 10 s");
                    return SPG_EHIP;
                }
            }
        }
        // ---- drain: every committed blanket's final word (its slot, packet and mailbox cell are reused by the next call)
        if (!emulate) {
            const double t0 = now_s();
            for (; fin_head < fin.size(); fin_head++) {
                const int32_t s = fin[fin_head];
                const volatile double *c = cell(s);
                uint32_t spins = 0;
                while (c[5] != 4503599627370496.0 + 4294967296.0 + (double)g->sslots[s].tag) {
                    if ((++spins & 0xfff) == 0 && now_s() - t0 > 10.0) { set_err(g->ctx, SPG_EHIP, "streaming driver: a blanket's KLD tail did not complete within 10 s"); return SPG_EHIP; }
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                }
                harvest(s, (const double *)c);
            }
            woken.clear();
        }
        fin.clear();
        if (idle_since) { t_idle += ticks_now() - idle_since; idle_since = 0; }
        const uint64_t t_end = ticks_now();
        const double secs = now_s() - s_begin;
        const double idle = (t_end > t_begin) ? secs * (double)t_idle / (double)(t_end - t_begin) : 0.0;
        g->stats.device_seconds += idle;
        g->stats.host_seconds += secs - idle;
        g->stats.schedule_seconds += secs - idle;   // (selection, commit and hand-over are one loop here; SPG_STREAM_PROF splits them)
        if (prof) {
            const char *nm[8] = {"poll", "commit", "examine: parked", "examine: launch decision", "packet", "doorbell", "late results", "launch tail"};
            const double tps = (double)(t_end - t_begin) / secs;
            for (int i = 0; i < 8; i++) fprintf(stderr, "stream prof %-26s %8llu x %8.1f ns = %8.3f ms\n", nm[i], (unsigned long long)pn[i], pn[i] ? 1e9 * (double)pt[i] / tps / (double)pn[i] : 0.0, 1e3 * (double)pt[i] / tps);
        }
        return 0;
    }
};
}  // namespace

// Runs the removal list of the open marginalisation (spg_graph_marginalize_begin) through the streaming driver.
// Returns 0 = list exhausted, 1 = the rest of the list (g->pending from g->pend_head) is left to the batch driver
// (a blanket the persistent worker does not take, arena full, or no streaming on this backend), < 0 error.
static int stream_marginalize(spg_graph *g, bool *started = nullptr) {
    const spg_options &o = g->opts;
    if (started) *started = false;
    static const bool env_off = [] { const char *e = getenv("SPG_STREAM"); return e && e[0] == '0'; }();
    spg::StreamPort *const sim = g->ctx->is_hip ? nullptr : g->ctx->sim_port;
    const bool emulate = !g->ctx->is_hip && !sim;
    if (env_off || g->stream_disabled) return 1;
    if (emulate && g->stream_emulation < 0) return 1;
    if (g->nranks != 1 || o.algorithm != SPG_ALG_NFR || o.topology != SPG_TOPO_TREE || o.lin_point != SPG_LIN_GLOBAL || o.flags != 0) return 1;
    const int32_t P = (int32_t)g->pending.size();
    if (P < (emulate ? 1 : 64)) return 1;   // a handful of removals (online decimation): one plain launch is cheaper than starting the worker
    Streamer S(g, emulate);
    if (sim) {
        if (sim->slots < kStreamSlots || sim->mail_stride < kStreamMailStride) return 1;
        S.port = *sim;
    } else if (!emulate) {
        int prc = spg::hip_stream_open(&g->ctx->be, g->d, kStreamSlots, kStreamMailStride, &S.port);
        if (prc < 0) { snprintf(g->ctx->err, sizeof g->ctx->err, "%s", spg::hip_backend_error(&g->ctx->be)); return prc; }
        if (prc > 0) return 1;
    }
    // state
    const size_t V = g->vid.size();
    if (g->cst.size() < V) g->cst.resize(V, -1);
    for (int32_t p = 0; p < P; p++) { g->cst[g->pending[p]] = (p << 2) | SV_WAITING; g->vr[g->pending[p]].s.nown = 0; }
    g->wl_next.assign((size_t)P, -1); g->wl_stable.assign((size_t)P, -1); g->wl_done.assign((size_t)P, -1);
    if (g->sslots.size() < (size_t)kStreamSlots) g->sslots.resize(kStreamSlots);
    g->s_free.clear();
    for (int s = kStreamSlots - 1; s >= 0; s--) g->s_free.push_back(s);
    g->s_fifo.clear(); g->s_fin.clear(); g->s_woken.clear(); g->s_ready.clear();
    next_stamp(g);
    if (g->lidx.size() < V) g->lidx.resize(V, -1);
    if ((int64_t)g->host.size() < g->cap) g->host.resize((size_t)g->cap);
    S.rng = 0x9E3779B97F4A7C15ULL ^ ((uint64_t)(g->stream_emulation > 0 ? g->stream_emulation : 1) * 0xD1B54A32D192ED03ULL);
    const int64_t used0 = g->used;
    if (started) *started = true;
    if (g->unsorted_from < 0) g->unsorted_from = (int64_t)g->edges.size();
    if (!emulate) S.cell_slot.assign((size_t)S.port.slots, -1);
    // second host thread that only polls the mailbox (SPG_STREAM_THREADS=1: none); the simulated port of tools/host_sim.cpp
    // gets one only on request (=2): its "device" is a thread as well
    static const int want_helper = [] { const char *e = getenv("SPG_STREAM_THREADS"); return e ? atoi(e) : 0; }();
    if (!emulate && (want_helper == 2 || (want_helper == 0 && !sim)) && P >= 4096) (void)S.helper_start();
    const double t_setup = now_s();
    const int rc = S.run();
    S.helper_stop();
    const double t_ran = now_s();
    if (sim) sim->tail = S.port.tail;
    else if (!emulate) spg::hip_stream_close(&g->ctx->be, &S.port, S.alg_bytes, (long long)S.n_done);
    // what the stream produced in the arena is device-only until someone asks for it
    if (g->used > used0) {
        g->dev_synced = g->used;
        if (!emulate && !sim) {
            if (g->stale_hi <= g->stale_lo) { g->stale_lo = used0; g->stale_hi = g->used; }
            else { g->stale_lo = std::min(g->stale_lo, used0); g->stale_hi = std::max(g->stale_hi, g->used); }
        } else {
            // injected backend: the mirror is filled from the backend's arena right away (tests read edges next)
            if (int r = g->ctx->be.download(g->ctx->be.user, g->host.data() + used0, (char *)g->dev + used0 * 8, g->used - used0)) return r;
        }
    }
    g->stats.n_rounds = S.bell_no;
    g->round_no = S.bell_no;
    // list entries the stream did not finish, in list order, for the batch driver
    size_t left = 0;
    for (int32_t p = 0; p < P; p++) {
        const int32_t v = g->pending[p];
        spg_graph::SVtx &sq = g->vr[v].s;
        const int vst = g->cst[v] & 3;
        if (vst == SV_STABLE) {   // a reservation that was never launched (the stream handed over to the batch driver)
            const spg_graph::SSlot &sl = g->sslots[sq.slot];
            for (int i = 0; i < sl.nv; i++) {
                spg_graph::SVtx &sx = g->vr[sl.verts[i]].s;
                for (int j = 0; j < sx.nown; j++) if (sx.own[j] == sq.slot) { sx.own[j] = sx.own[--sx.nown]; break; }
            }
        }
        const bool done = vst == SV_DONE;
        g->cst[v] = -1; sq.nown = 0;
        if (!done) g->pending[left++] = v;
    }
    g->pending.resize(left);
    g->pend_head = 0;
    if (getenv("SPG_TRACE"))
        fprintf(stderr, "spg trace: streaming driver: run %.3f ms, tear-down %.3f ms\n", 1e3 * (t_ran - t_setup), 1e3 * (now_s() - t_ran));
    if (getenv("SPG_TRACE"))
        fprintf(stderr, "spg trace: streaming driver: %d list entries, %d committed in %d doorbells, %zu left to the batch driver (examined up to entry %d)\n",
                P, S.n_done, S.bell_no, left, S.cursor);
    if (getenv("SPG_TRACE"))
        fprintf(stderr, "spg trace: streaming driver: %ld examinations; parked on: earlier neighbour %ld, own blanket %ld, two shared vertices %ld, waiting member of a touching blanket %ld, waiting 2-hop neighbour %ld\n",
                S.n_exam, S.n_park[0], S.n_park[1], S.n_park[2], S.n_park[3], S.n_park[4]);
    if (rc) return rc;
    return left ? 1 : 0;
}

extern "C" int spg_graph_marginalize(spg_graph *g, const int32_t *which, int n, const spg_options *opts, spg_marg_stats *stats) {
    return spg_graph_marginalize_ranks(g, which, n, opts, 0, 1, nullptr, nullptr, stats);
}

extern "C" int spg_graph_marginalize_ranks(spg_graph *g, const int32_t *which, int n, const spg_options *opts, int rank, int nranks,
                                           spg_exchange_fn exchange, void *exchange_user, spg_marg_stats *stats) {
    if (!g) return SPG_EINVAL;
    // no callback: the built-in RCCL all-gather of the context (spg_ctx_create_ranks with the same rank / nranks)
    const bool builtin = nranks > 1 && !exchange;
    if (builtin && (!g->ctx->rccl || g->ctx->nranks != nranks || g->ctx->rank != rank))
        return set_err(g->ctx, SPG_EINVAL, "spg_graph_marginalize_ranks: no exchange callback and the context has no matching RCCL communicator (spg_ctx_create_ranks)");
    int launches0 = (g && g->ctx->is_hip) ? spg::hip_backend_launches(&g->ctx->be) : 0;
    const double t_call = now_s();
    int rc = spg_graph_marginalize_begin(g, which, n, opts, rank, nranks);
    if (rc) return rc;
    const double t_begun = now_s();
    const char *env = getenv("SPG_NO_PIPELINE");
    g->pipelined = g->ctx->be.synchronize_slot && g->ctx->be.mailbox_slot && !(env && env[0] == '1');
    // single rank, NFR Tree at the stored estimates: blanket by blanket through the persistent worker (streaming driver);
    // whatever it leaves (rc 1: blankets the worker does not take) goes through the batch driver below
    int stream_rc = 1;
    if (nranks == 1 && !exchange) stream_rc = stream_marginalize(g);
    else if (nranks > 1) {
        // Several ranks, one replicated graph. A batch is worth sharding + one all-gather only when it is wide (shard_pays);
        // lists whose batches are narrow — the 100k-pose lattice: ~200 independent blankets at a time — are computed whole by
        // every rank with nothing exchanged, and then nothing ties the ranks to the same batches either: each rank may run
        // its own streaming driver. Results do not depend on the schedule (blanket edges are summed in key order), so the
        // replicas stay equal in content; their arena LAYOUTS diverge (records are placed in launch order), and regions are
        // exchanged by offset — so a graph that has streamed on several ranks never shards again (layout_diverged).
        // The decision is taken on the first pass of the batch scheduler, identically on every rank.
        bool independent = g->layout_diverged;
        if (!independent && g->shard_threshold < 0) {
            const std::vector<int32_t> saved = g->pending;
            g->B = &g->bt[0];
            schedule_round(g);
            independent = !g->bt[0].rb.empty() && !shard_pays(g, g->bt[0]);
            release_batch_owners(g, g->bt[0]);
            g->bt[0].rb.clear(); g->bt[0].rb_verts.clear(); g->bt[0].rb_edges.clear();
            for (int32_t oid : g->transient) owner_release(g, oid);
            g->transient.clear();
            g->pending = saved;
            g->pend_head = 0;
        }
        if (independent) {
            const int rank0 = g->rank, nranks0 = g->nranks;
            g->rank = 0; g->nranks = 1;
            bool started = false;
            stream_rc = stream_marginalize(g, &started);
            if (started || g->layout_diverged) g->layout_diverged = true;   // (and the batch driver below, if it gets the rest, runs without sharding)
            else { g->rank = rank0; g->nranks = nranks0; }
        }
    }
    if (stream_rc <= 0) {
        const double t_streamed = now_s();
        int rc2 = spg_graph_marginalize_end(g, stats);
        if (getenv("SPG_TRACE")) fprintf(stderr, "spg trace: marginalize: begin %.3f ms, stream %.3f ms, end %.3f ms\n", 1e3 * (t_begun - t_call), 1e3 * (t_streamed - t_begun), 1e3 * (now_s() - t_streamed));
        if (stats && g->ctx->is_hip) stats->n_launches -= launches0;
        return stream_rc < 0 ? stream_rc : rc2;
    }
    auto do_exchange = [&](Batch &b) -> int {
        if (!b.rinfo.exchange) return 0;
        double tx = now_s();
        int erc = g->ctx->be.synchronize(g->ctx->be.user);  // this rank's chunk is complete in memory
        if (erc) return erc;
        erc = builtin ? spg_allgather_region(g->ctx, g->dev, b.rinfo.region_off, b.rinfo.chunk_len)
                      : exchange(exchange_user, g->dev, b.rinfo.region_off, b.rinfo.chunk_len, g->nranks, g->rank);
        g->stats.n_exchanged++;
        g->stats.exchanged_bytes += 8.0 * (double)b.rinfo.chunk_len * g->nranks;
        g->stats.exchange_seconds += now_s() - tx;
        return erc;
    };
    if (!g->pipelined) {
        for (;;) {
            rc = spg_graph_round_prepare(g, nullptr);
            if (rc <= 0) break;
            if ((rc = spg_graph_round_compute(g)) != 0) break;
            if ((rc = do_exchange(*g->B)) != 0) break;
            if ((rc = spg_graph_round_commit(g)) != 0) break;
        }
    } else {
        // Up to NB batches in flight. After every commit one scheduling pass picks whatever commutes
        // with the batches still running (and with everything earlier in the list); a large pick is
        // cut into as many parts as there are idle slots so that the host work of one part overlaps the
        // device work of the others. With nothing schedulable, wait for the oldest batch and apply it.
        constexpr int NB = spg_graph::NB;
        rc = 0;
        const char *e1 = getenv("SPG_SPLIT_MIN"), *e2 = getenv("SPG_MAX_INFLIGHT");
        // defaults from sweeps on 100k-pose lattices with rings of 250 ... 1000: parts of >= 48 blankets,
        // four batches in flight (more only add passes), patience 16
        const size_t split_min = e1 ? (size_t)atoi(e1) : 48;
        const int max_inflight = e2 ? std::max(1, std::min(NB, atoi(e2))) : 4;
        auto commit_all = [&]() -> int {
            for (;;) {
                Batch *o = nullptr;
                for (int i = 0; i < NB; i++) if (g->bt[i].round_open && (!o || g->bt[i].seq < o->seq)) o = &g->bt[i];
                if (!o) return 0;
                g->B = o;
                if (int c = spg_graph_round_commit(g)) return c;
            }
        };
        // SPG_HOST_THREADS=2: descriptors + device hand-over on a second host thread. Off by default: measured on the
        // bench workload it does not pay (33.1 vs 31.9 ms per step with the two threads on neighbouring cores, 48 ms
        // on different L3s) — a batch waits for its round trip through the device, not for the graph thread.
        static const bool use_thread = [] { const char *e = getenv("SPG_HOST_THREADS"); return e && e[0] == '2'; }();
        auto launch_batch = [&](Batch &b, double t0) -> int {
            g->B = &b;
            const bool sharded = shard_pays(g, b);
            if (use_thread && !sharded) {
                // descriptors + hand-over on the submission thread; region, tag and late results here
                int prc = prepare_region_single(g, b, t0);
                if (prc == 2) {  // the arena has to grow: nothing may be running while it is re-allocated
                    if (int c = commit_all()) return c;
                    g->B = &b;
                    prc = prepare_region_single(g, b, now_s());
                }
                if (prc <= 0) return prc;
                if (int hrc = compute_prologue(g, b)) return hrc;
                submission_start(g);
                submission_push(g, b);
                return 0;
            }
            int prc = prepare_scheduled(g, nullptr, t0);
            if (prc == 2) {  // the arena has to grow: nothing may be running while it is re-allocated
                if (int c = commit_all()) return c;
                g->B = &b;
                prc = prepare_scheduled(g, nullptr, now_s());
            }
            if (prc < 0) return prc;
            return spg_graph_round_compute(g);
        };
        for (;;) {
            bool launched = false;
            int f = -1, nfree = 0;
            for (int i = 0; i < NB; i++) if (!g->bt[i].round_open) { if (f < 0) f = i; nfree++; }
            nfree -= NB - max_inflight;
            if (nfree <= 0) f = -1;
            if (f >= 0 && g->pend_head < g->pending.size()) {
                Batch &bt = g->bt[f];
                g->B = &bt;
                double t0 = now_s();
                schedule_round(g);
                g->stats.schedule_seconds += now_s() - t0;
                if (bt.rb.empty()) {
                    g->stats.host_seconds += now_s() - t0;
                } else if (shard_pays(g, bt)) {
                    // a wide batch: worth sharding over the ranks. Finish what is in flight, then run it
                    // as one exchanged round (compute own slice, all-gather, commit).
                    if ((rc = commit_all()) != 0) break;
                    if ((rc = launch_batch(bt, t0)) != 0) break;
                    if ((rc = do_exchange(bt)) != 0) break;
                    g->B = &bt;
                    if ((rc = spg_graph_round_commit(g)) != 0) break;
                    continue;
                } else {
                    size_t S = bt.rb.size();
                    int parts = (int)std::min<size_t>((size_t)nfree, std::max<size_t>(1, S / split_min));
                    // hand parts 1..parts-1 to other idle batches, keep part 0 here
                    std::vector<Batch *> tgt;
                    for (int i = 0; i < NB && (int)tgt.size() < parts - 1; i++) if (i != f && !g->bt[i].round_open) tgt.push_back(&g->bt[i]);
                    parts = (int)tgt.size() + 1;
                    size_t per = (S + parts - 1) / parts;
                    for (int pi = 1; pi < parts; pi++) move_blankets(g, bt, std::min(S, per * pi), std::min(S, per * (pi + 1)), *tgt[pi - 1]);
                    bt.rb.resize(std::min(S, per));
                    if ((rc = launch_batch(bt, t0)) != 0) break;
                    for (int pi = 1; pi < parts && rc == 0; pi++) if (!tgt[pi - 1]->rb.empty()) rc = launch_batch(*tgt[pi - 1], now_s());
                    if (rc != 0) break;
                    launched = true;
                }
            }
            Batch *oldest = nullptr;
            for (int i = 0; i < NB; i++) if (g->bt[i].round_open && (!oldest || g->bt[i].seq < oldest->seq)) oldest = &g->bt[i];
            if (!oldest) {
                if (!launched) break;  // nothing in flight, nothing schedulable: done
                continue;
            }
            // keep scheduling while slots are idle and the last pass found work; otherwise apply the oldest batch
            if (launched) {
                int nf2 = 0;
                for (int i = 0; i < NB; i++) nf2 += !g->bt[i].round_open;
                (void)nf2;
            }
            g->B = oldest;
            if ((rc = spg_graph_round_commit(g)) != 0) break;
        }
        // drain on error
        quiesce_submission(g);
        for (int s_ = 0; s_ < spg_graph::NB; s_++) if (g->bt[s_].round_open) { g->ctx->be.synchronize(g->ctx->be.user); g->bt[s_].round_open = false; }
        g->B = &g->bt[0];
    }
    int rc2 = spg_graph_marginalize_end(g, stats);
    if (stats && g->ctx->is_hip) stats->n_launches -= launches0;
    return rc < 0 ? rc : rc2;
}

// tools/host_sim.cpp (declared in csrc/spg_internal.h, not part of the public ABI): the streaming driver of a context
// with an injected backend talks to this port — host memory, with a thread of the tool playing the persistent worker.
extern "C" int spg_debug_la(int op, int M, int N, int K, int flags, int mode, double *A, int ra, int lda, double *B, int rb, int ldb, double *C, int rc, int ldc, int *ok) {
    if (!A || !B || !C || !ok || M <= 0 || lda <= 0 || ldb <= 0 || ldc <= 0) return SPG_EINVAL;
    return spg::hip_la_test(op, M, N, K, flags, mode, A, ra, lda, B, rb, ldb, C, rc, ldc, ok);
}

extern "C" int spg_debug_set_stream_port(spg_ctx *c, void *port) {
    if (!c || c->is_hip) return SPG_EINVAL;
    c->sim_port = (spg::StreamPort *)port;
    return 0;
}

extern "C" int spg_graph_set_stream_emulation(spg_graph *g, int seed) {
    if (!g) return SPG_EINVAL;
    g->stream_disabled = seed <= -2;
    g->stream_emulation = seed < 0 ? -1 : seed;
    return 0;
}

extern "C" int spg_graph_last_blanket_count(const spg_graph *g) { return g ? (int)g->log.size() : 0; }
extern "C" int spg_graph_last_blankets(const spg_graph *g, int32_t *root_id, int32_t *round, int32_t *status, int32_t *info, double *kld, double *min_gap) {
    if (!g) return SPG_EINVAL;
    for (size_t i = 0; i < g->log.size(); i++) {
        if (root_id) root_id[i] = g->log[i].root_id;
        if (round) round[i] = g->log[i].round;
        if (status) status[i] = g->log[i].status;
        if (info) info[i] = g->log[i].info;
        if (kld) kld[i] = g->log[i].kld;
        if (min_gap) min_gap[i] = g->log[i].min_gap;
    }
    return (int)g->log.size();
}

// ================================================================================= batch entry
// Self-contained batch: builds a temporary arena [poses | edge records | out records | new slots |
// target infos], runs ONE round through the same backend entry the graph uses, unpacks the result.
extern "C" int spg_marginalize_batch(spg_ctx *ctx, const spg_options *o, const spg_batch *bt, spg_result *r) {
    if (!ctx || !o || !bt || !r) return SPG_EINVAL;
    const int d = o->pose_dim;
    if (d != 3 && d != 6) return SPG_EINVAL;
    const int ps = pose_stride(d);
    const int B = bt->B;
    r->new_edge_off[0] = 0;
    r->new_edge_vert_off[0] = 0;
    r->new_edge_data_off[0] = 0;
    if (B == 0) return 0;
    const int V = bt->vert_off[B], E = bt->edge_off[B];
    const int64_t ED = bt->edge_data_off[E];
    std::vector<double> host;
    int64_t o_pose = 0, o_edge = (int64_t)V * ps;
    int64_t cur = align_up(o_edge + ED, 32);
    std::vector<spg_blanket_desc> blk(B);
    std::vector<int64_t> vpo(V);
    std::vector<spg_edge_ref> er(E);
    for (int v = 0; v < V; v++) vpo[v] = o_pose + (int64_t)v * ps;
    for (int e = 0; e < E; e++) {
        er[e].off = o_edge + bt->edge_data_off[e];
        er[e].len = (int32_t)(bt->edge_data_off[e + 1] - bt->edge_data_off[e]);
        er[e].kind = bt->edge_kind[e];
        er[e].vbegin = bt->edge_vert_off[e];
        er[e].nv = bt->edge_vert_off[e + 1] - bt->edge_vert_off[e];
    }
    for (int b = 0; b < B; b++) {
        spg_blanket_desc &bd = blk[b];
        memset(&bd, 0, sizeof bd);
        bd.vert_begin = bt->vert_off[b];
        bd.n_vert = bt->vert_off[b + 1] - bt->vert_off[b];
        bd.n_remove = bt->n_remove[b];
        bd.edge_begin = bt->edge_off[b];
        bd.n_edge = bt->edge_off[b + 1] - bt->edge_off[b];
        int k = bd.n_vert - bd.n_remove;
        new_edge_budget(*o, d, std::max(k, 0), bd.n_new_max, bd.n_new_vert_max, bd.new_len);
        for (int e = bd.edge_begin; e < bd.edge_begin + bd.n_edge; e++)
            if (er[e].kind == SPG_EDGE_GLC) bd.pad_ = std::max(bd.pad_, er[e].len - d * er[e].nv + er[e].nv * 2 * d * d);
        bd.out_off = cur; cur += SPG_OUT_LEN(bd.n_new_max, bd.n_new_vert_max);
        bd.new_off = cur; cur += bd.new_len;
        if (r->target_info) { int64_t n = (int64_t)d * std::max(k, 0); bd.tinfo_off = cur; cur += n * n; }
        else bd.tinfo_off = -1;
    }
    int64_t in_len = align_up(o_edge + ED, 32);
    host.assign((size_t)cur, 0.0);
    memcpy(host.data(), bt->pose, (size_t)V * ps * 8);
    if (ED) memcpy(host.data() + o_edge, bt->edge_data, (size_t)ED * 8);
    void *dev = ctx->be.alloc(ctx->be.user, cur);
    if (!dev) return set_err(ctx, SPG_ENOMEM, "arena allocation failed");
    int rc = ctx->be.upload(ctx->be.user, dev, host.data(), in_len);
    spg_round_desc rd{};
    rd.opts = o;
    rd.n_blankets = B; rd.first = 0; rd.count = B;
    rd.blankets = blk.data();
    rd.vert_pose_off = vpo.data();
    rd.edges = er.data();
    rd.edge_vert = bt->edge_vert;
    rd.n_vert_total = V; rd.n_edge_total = E; rd.n_edge_vert_total = bt->edge_vert_off[E];
    if (!rc) rc = ctx->be.run_round(ctx->be.user, dev, &rd);
    if (!rc) rc = ctx->be.synchronize(ctx->be.user);
    if (!rc && cur > in_len) rc = ctx->be.download(ctx->be.user, host.data() + in_len, (char *)dev + in_len * 8, cur - in_len);
    ctx->be.release(ctx->be.user, dev);
    if (rc) {
        if (ctx->is_hip) snprintf(ctx->err, sizeof ctx->err, "%s", spg::hip_backend_error(&ctx->be));
        return rc;
    }
    int32_t ne = 0, nev = 0;
    int64_t ned = 0;
    for (int b = 0; b < B; b++) {
        const spg_blanket_desc &bd = blk[b];
        const double *rec = host.data() + bd.out_off;
        r->status[b] = (int32_t)rec[0];
        if (r->info) r->info[b] = (int32_t)rec[1];
        r->kld[b] = rec[2];
        if (r->min_gap) r->min_gap[b] = rec[3];
        int n_new = (int)rec[4];
        int k = bd.n_vert - bd.n_remove;
        if (r->target_info && k > 0) {
            int64_t n = (int64_t)d * k;
            memcpy(r->target_info + r->target_info_off[b], host.data() + bd.tinfo_off, (size_t)(n * n) * 8);
        }
        int vpos = 0;
        for (int e = 0; e < n_new; e++) {
            int kind = (int)rec[SPG_OUT_HDR + 4 * e + 0];
            int64_t rel = (int64_t)rec[SPG_OUT_HDR + 4 * e + 1];
            int64_t len = (int64_t)rec[SPG_OUT_HDR + 4 * e + 2];
            int nv = (int)rec[SPG_OUT_HDR + 4 * e + 3];
            if (ne + 1 > r->new_edge_cap || nev + nv > r->new_edge_vert_cap || ned + len > r->new_edge_data_cap) return SPG_ECAPACITY;
            r->new_edge_kind[ne] = kind;
            for (int i = 0; i < nv; i++)
                r->new_edge_vert[nev++] = bt->vert_id[bd.vert_begin + (int)rec[SPG_OUT_HDR + 4 * bd.n_new_max + vpos + i]];
            vpos += nv;
            memcpy(r->new_edge_data + ned, host.data() + bd.new_off + rel, (size_t)len * 8);
            ned += len;
            ne++;
            r->new_edge_vert_off[ne] = nev;
            r->new_edge_data_off[ne] = ned;
        }
        r->new_edge_off[b + 1] = ne;
    }
    return 0;
}

// ================================================================================= substitute edge
// computeSubstituteEdge (src/compute_substitute_edge.cpp:13-96): online / cluster replay only. When a
// new edge points at a vertex that was already marginalised, walk breadth-first (ids <= maxid, never
// through vertex 0) to the nearest surviving vertex with the smallest id, compose the measurements
// along the way back and add the covariances. Host-side: 3x3 / 6x6 arithmetic once per such edge.
// Where the reference takes "the first edge of `reach` that touches the previous frontier" in g2o's
// pointer order, this build takes the lowest edge index.
namespace {
struct HPose { double t[3]; double q[4]; double th; };  // SE3: t,q ; SE2: t[0..1], th

void q_mul(const double *a, const double *b, double *o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
void q_rot(const double *q, const double *v, double *o) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]), tz = 2 * (x * v[1] - y * v[0]);
    o[0] = v[0] + w * tx + (y * tz - z * ty);
    o[1] = v[1] + w * ty + (z * tx - x * tz);
    o[2] = v[2] + w * tz + (x * ty - y * tx);
}
double wrap_theta(double th) {
    const double PI = 3.14159265358979323846;
    if (th >= -PI && th < PI) return th;
    double m = std::fmod(th, 2 * PI);
    if (m >= PI) m -= 2 * PI;
    if (m < -PI) m += 2 * PI;
    return m;
}
void pose_compose(int d, const double *a, const double *b, double *o) {  // o = a * b (o may alias neither)
    if (d == 3) {
        double c = std::cos(a[2]), s = std::sin(a[2]);
        o[0] = a[0] + c * b[0] - s * b[1]; o[1] = a[1] + s * b[0] + c * b[1]; o[2] = wrap_theta(a[2] + b[2]);
    } else {
        double r[3];
        q_rot(a + 3, b, r);
        o[0] = a[0] + r[0]; o[1] = a[1] + r[1]; o[2] = a[2] + r[2];
        q_mul(a + 3, b + 3, o + 3);
        double n = std::sqrt(o[3] * o[3] + o[4] * o[4] + o[5] * o[5] + o[6] * o[6]);
        for (int i = 3; i < 7; i++) o[i] /= n;
    }
}
void pose_inverse(int d, const double *a, double *o) {
    if (d == 3) {
        double c = std::cos(a[2]), s = std::sin(a[2]);
        o[0] = -(c * a[0] + s * a[1]); o[1] = -(-s * a[0] + c * a[1]); o[2] = wrap_theta(-a[2]);
    } else {
        double qi[4] = {-a[3], -a[4], -a[5], a[6]}, r[3];
        q_rot(qi, a, r);
        o[0] = -r[0]; o[1] = -r[1]; o[2] = -r[2];
        o[3] = qi[0]; o[4] = qi[1]; o[5] = qi[2]; o[6] = qi[3];
    }
}
bool dense_inverse(int n, std::vector<double> &A) {  // Gauss-Jordan, partial pivoting (Eigen .inverse())
    std::vector<double> X((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) X[(size_t)i * n + i] = 1.0;
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++) if (std::fabs(A[(size_t)i * n + k]) > std::fabs(A[(size_t)p * n + k])) p = i;
        if (A[(size_t)p * n + k] == 0.0) return false;
        if (p != k) for (int j = 0; j < n; j++) { std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]); std::swap(X[(size_t)k * n + j], X[(size_t)p * n + j]); }
        double ip = 1.0 / A[(size_t)k * n + k];
        for (int j = 0; j < n; j++) { A[(size_t)k * n + j] *= ip; X[(size_t)k * n + j] *= ip; }
        for (int i = 0; i < n; i++) {
            if (i == k) continue;
            double f = A[(size_t)i * n + k];
            if (f == 0.0) continue;
            for (int j = 0; j < n; j++) { A[(size_t)i * n + j] -= f * A[(size_t)k * n + j]; X[(size_t)i * n + j] -= f * X[(size_t)k * n + j]; }
        }
    }
    A.swap(X);
    return true;
}
}  // namespace

extern "C" int spg_graph_substitute_edge(spg_graph *g, const int32_t *marginalized, int n_marg, int maxid,
                                         int *from, int *to, double *meas, double *info_upper) {
    if (!g || !from || !to || !meas || !info_upper || (n_marg > 0 && !marginalized)) return SPG_EINVAL;
    if (int rc = sync_host(g)) return rc;
    const int d = g->d, ps = g->ps;
    const size_t NV = g->vid.size();
    auto slot_of = [&](int id) -> int32_t { auto it = g->vidx.find(id); return (it == g->vidx.end() || !g->valive[it->second]) ? -1 : it->second; };
    const int id_new = std::max(*from, *to), id_gone = std::min(*from, *to);   // the new vertex / its removed neighbour
    const int32_t s_new = slot_of(id_new), s_gone = slot_of(id_gone);
    if (s_new < 0 || s_gone < 0) return set_err(g->ctx, SPG_EINVAL, "substitute edge endpoint does not exist");
    // per-slot flags: bit 0 = removed so far, bit 1 = expanded by the search, bit 2 = queued for the next level
    std::vector<uint8_t> flag(NV, 0);
    for (int i = 0; i < n_marg; i++) { int32_t sl = slot_of(marginalized[i]); if (sl >= 0) flag[sl] |= 1; }
    // Breadth-first levels over vertex slots, every level in ascending id (the reference walks std::set<int>):
    // level[lvl_off[l] .. lvl_off[l+1]). A vertex that is removed (or an endpoint of the edge) is expanded; the
    // search stops at the first level that holds a surviving vertex and takes the smallest such id.
    std::vector<int32_t> level, lvl_off{0}, nextl;
    level.push_back(s_gone);
    lvl_off.push_back(1);
    flag[s_new] |= 2; flag[s_gone] |= 2;
    int32_t hit = -1;
    for (;;) {
        const int32_t lo = lvl_off[lvl_off.size() - 2], hi = lvl_off.back();
        nextl.clear();
        for (int32_t p = lo; p < hi; p++) {
            const int32_t v = level[p];
            const int id = g->vid[v];
            if (!(flag[v] & 1) && id != *from && id != *to) {
                if (hit < 0 || id < g->vid[hit]) hit = v;   // a survivor: candidate for the substitute endpoint
                continue;
            }
            flag[v] |= 2;
            for (int32_t eid : g->vr[v].adj) {
                const GEdge &e = g->edges[eid];
                if (e.nv != 2) continue;
                const int32_t u = (e.vtx[0] == v) ? e.vtx[1] : e.vtx[0];
                if ((flag[u] & 2) || g->vid[u] > maxid || g->vid[u] == 0 || (flag[u] & 4)) continue;
                flag[u] |= 4;
                nextl.push_back(u);
            }
        }
        if (hit >= 0) break;
        if (nextl.empty()) return set_err(g->ctx, SPG_EINVAL, "no surviving vertex reachable");
        std::sort(nextl.begin(), nextl.end(), [&](int32_t a, int32_t b) { return g->vid[a] < g->vid[b]; });
        for (int32_t u : nextl) flag[u] &= (uint8_t)~4;
        level.insert(level.end(), nextl.begin(), nextl.end());
        lvl_off.push_back((int32_t)level.size());
    }
    // Walk back from the survivor, level by level towards the new vertex: at each step the lowest-index pose-pose
    // edge of the current vertex that touches the previous level; measurements compose, covariances add.
    const int n_levels = (int)lvl_off.size() - 1;   // levels 0 .. n_levels-1; the survivor sits in the last one
    std::vector<double> cov((size_t)d * d, 0.0), acc(ps, 0.0), nxt(ps), zinv(ps), om((size_t)d * d);
    if (d == 6) acc[6] = 1.0;
    std::vector<uint8_t> in_prev(NV, 0);
    std::vector<int32_t> es;
    int32_t cur = hit;
    const bool new_is_from = (*from == id_new);
    for (int l = n_levels - 2; l >= -1; l--) {
        // previous level: l >= 0 -> the BFS level; l == -1 -> the new vertex alone
        if (l >= 0) for (int32_t p = lvl_off[l]; p < lvl_off[l + 1]; p++) in_prev[level[p]] = 1;
        else in_prev[s_new] = 1;
        es.assign(g->vr[cur].adj.begin(), g->vr[cur].adj.end());
        std::sort(es.begin(), es.end());
        bool stepped = false;
        for (int32_t eid : es) {
            const GEdge &e = g->edges[eid];
            if (e.nv != 2 || e.kind != SPG_EDGE_BINARY) continue;
            if (!(in_prev[e.vtx[0]] || in_prev[e.vtx[1]])) continue;
            const double *rec = g->host.data() + e.off;
            int q = 0;
            for (int i = 0; i < d; i++) for (int j = i; j < d; j++) { om[(size_t)i * d + j] = om[(size_t)j * d + i] = rec[ps + q]; q++; }
            if (!dense_inverse(d, om)) return set_err(g->ctx, SPG_EINVAL, "singular edge information on the substitute path");
            for (int i = 0; i < d * d; i++) cov[i] += om[i];
            const bool cur_is_head = (e.vtx[1] == cur);   // the edge points at the current vertex
            if (new_is_from) {
                if (cur_is_head) pose_compose(d, rec, acc.data(), nxt.data());
                else { pose_inverse(d, rec, zinv.data()); pose_compose(d, zinv.data(), acc.data(), nxt.data()); }
            } else {
                if (cur_is_head) { pose_inverse(d, rec, zinv.data()); pose_compose(d, acc.data(), zinv.data(), nxt.data()); }
                else pose_compose(d, acc.data(), rec, nxt.data());
            }
            acc.swap(nxt);
            cur = cur_is_head ? e.vtx[0] : e.vtx[1];
            stepped = true;
            break;
        }
        if (l >= 0) for (int32_t p = lvl_off[l]; p < lvl_off[l + 1]; p++) in_prev[level[p]] = 0;
        else in_prev[s_new] = 0;
        (void)stepped;   // as in the reference, a level without a matching edge is skipped
    }
    if (!dense_inverse(d, cov)) return set_err(g->ctx, SPG_EINVAL, "singular covariance sum");
    int q = 0;
    for (int i = 0; i < d; i++) for (int j = i; j < d; j++) info_upper[q++] = 0.5 * (cov[(size_t)i * d + j] + cov[(size_t)j * d + i]);
    for (int i = 0; i < ps; i++) meas[i] = acc[i];
    if (new_is_from) *to = g->vid[hit]; else *from = g->vid[hit];
    return 0;
}

// ================================================================================= global KLD (a18)
namespace {
struct DenseStage {
    std::vector<int32_t> pos, rowptr, inc, ev;
    std::vector<spg_edge_ref> er;
    spg::DenseGraphIn in;
};

// Live vertex indices in ascending id order.
std::vector<int32_t> live_vertices_by_id(const spg_graph *g) {
    std::vector<int32_t> v;
    for (size_t i = 0; i < g->vid.size(); i++) if (g->valive[i]) v.push_back((int32_t)i);
    std::sort(v.begin(), v.end(), [&](int32_t a, int32_t b) { return g->vid[a] < g->vid[b]; });
    return v;
}

// st.pos must be filled (size = number of vertex slots, -1 = not a variable).
void build_dense_stage(spg_graph *g, DenseStage &st) {
    canonicalize_edge_order(g);
    const int nv = (int)g->vid.size();
    std::vector<int32_t> remap(g->edges.size(), -1);
    for (size_t e = 0; e < g->edges.size(); e++) {
        const GEdge &ge = g->edges[e];
        if (!ge.alive) continue;
        remap[e] = (int32_t)st.er.size();
        st.er.push_back({ge.off, ge.len, ge.kind, (int32_t)st.ev.size(), ge.nv});
        for (int i = 0; i < ge.nv; i++) st.ev.push_back(edge_verts(g, ge)[i]);
    }
    st.rowptr.assign((size_t)nv + 1, 0);
    for (int v = 0; v < nv; v++) {
        if (g->valive[v]) {
            std::vector<int32_t> es;
            for (int32_t e : g->vr[v].adj) if (remap[e] >= 0) es.push_back(remap[e]);
            std::sort(es.begin(), es.end());
            es.erase(std::unique(es.begin(), es.end()), es.end());
            st.inc.insert(st.inc.end(), es.begin(), es.end());
        }
        st.rowptr[v + 1] = (int32_t)st.inc.size();
    }
    st.in.D = g->d; st.in.nv = nv; st.in.ne = (int)st.er.size();
    st.in.pos = st.pos.data(); st.in.vpo = g->vpose.data(); st.in.rowptr = st.rowptr.data(); st.in.inc = st.inc.data();
    st.in.er = st.er.data(); st.in.ev = st.ev.data(); st.in.n_ev = (int64_t)st.ev.size(); st.in.dev_arena = g->dev;
}

int resolve_fixed(const spg_graph *g, const std::vector<int32_t> &order, int32_t fixed_id) {
    if (order.empty()) return -1;
    if (fixed_id < 0) return order[0];   // the reference skips its first (smallest-id) vertex
    auto it = g->vidx.find(fixed_id);
    if (it == g->vidx.end() || !g->valive[it->second]) return -1;
    return it->second;
}
}  // namespace

extern "C" int64_t spg_graph_information(spg_graph *g, int32_t fixed_id, double *out, int64_t cap) {
    if (!g || g->active) return SPG_EINVAL;
    std::vector<int32_t> order = live_vertices_by_id(g);
    int fixed = resolve_fixed(g, order, fixed_id);
    if (fixed < 0) return set_err(g->ctx, SPG_EINVAL, "spg_graph_information: the fixed vertex is not in the graph");
    const int64_t n = (int64_t)g->d * ((int64_t)order.size() - 1);
    if (!out || cap < n * n) return n;
    if (!g->ctx->is_hip) return set_err(g->ctx, SPG_ESTATE, "spg_graph_information needs the HIP backend");
    if (int rc = sync_device(g)) return rc;
    if (int rc = g->ctx->be.synchronize(g->ctx->be.user)) return rc;
    DenseStage st;
    st.pos.assign(g->vid.size(), -1);
    int p = 0;
    for (int32_t v : order) if (v != fixed) { st.pos[v] = p; p += g->d; }
    build_dense_stage(g, st);
    g->ctx->err[0] = 0;
    int rc = spg::hip_dense_information(spg::hip_backend_stream(&g->ctx->be), st.in, (int)n, out, g->ctx->err, sizeof g->ctx->err);
    return rc ? rc : n;
}

extern "C" int64_t spg_graph_covariance(spg_graph *g, int32_t fixed_id, double *out, int64_t cap) {
    if (!g || g->active) return SPG_EINVAL;
    std::vector<int32_t> order = live_vertices_by_id(g);
    int fixed = resolve_fixed(g, order, fixed_id);
    if (fixed < 0) return set_err(g->ctx, SPG_EINVAL, "spg_graph_covariance: the fixed vertex is not in the graph");
    const int64_t n = (int64_t)g->d * ((int64_t)order.size() - 1);
    if (!out || cap < n * n) return n;
    if (!g->ctx->is_hip) return set_err(g->ctx, SPG_ESTATE, "spg_graph_covariance needs the HIP backend");
    if (n > 46000) return set_err(g->ctx, SPG_ECAPACITY, "spg_graph_covariance: dense formulation limited to 46k variables");
    if (int rc = sync_device(g)) return rc;
    if (int rc = g->ctx->be.synchronize(g->ctx->be.user)) return rc;
    DenseStage st;
    st.pos.assign(g->vid.size(), -1);
    int p = 0;
    for (int32_t v : order) if (v != fixed) { st.pos[v] = p; p += g->d; }
    build_dense_stage(g, st);
    g->ctx->err[0] = 0;
    int rc = spg::hip_dense_covariance(spg::hip_backend_stream(&g->ctx->be), st.in, (int)n, out, g->ctx->err, sizeof g->ctx->err);
    return rc ? rc : n;
}

extern "C" int spg_graph_kullback_leibler(spg_graph *base, spg_graph *other, int32_t fixed_id, spg_kld_terms *out) {
    if (!base || !other || !out || base->active || other->active) return SPG_EINVAL;
    spg_ctx *ctx = base->ctx;
    if (base->d != other->d) return set_err(ctx, SPG_EINVAL, "spg_graph_kullback_leibler: pose dimensions differ");
    if (!ctx->is_hip || !other->ctx->is_hip) return set_err(ctx, SPG_ESTATE, "spg_graph_kullback_leibler needs the HIP backend");
    if (spg::hip_backend_device(&ctx->be) != spg::hip_backend_device(&other->ctx->be))
        return set_err(ctx, SPG_EINVAL, "spg_graph_kullback_leibler: both graphs must live on the same device");
    const int d = base->d;
    std::vector<int32_t> ob = live_vertices_by_id(base), oo = live_vertices_by_id(other);
    int fb = resolve_fixed(base, ob, fixed_id);
    if (fb < 0) return set_err(ctx, SPG_EINVAL, "spg_graph_kullback_leibler: the fixed vertex is not in the baseline");
    const int32_t fid = base->vid[fb];
    int fo = resolve_fixed(other, oo, fid);
    if (fo < 0) return set_err(ctx, SPG_EINVAL, "spg_graph_kullback_leibler: the fixed vertex is not in the sparsified graph");
    // computeIndices (src/graph_wrapper_g2o.cpp:472-499): merge of the two id-sorted vertex lists
    std::vector<int32_t> kept_b, kept_o, marg_b;
    {
        size_t j = 0;
        for (int32_t v : ob) {
            if (v == fb) continue;
            while (j < oo.size() && (oo[j] == fo || other->vid[oo[j]] < base->vid[v])) {
                if (oo[j] != fo) return set_err(ctx, SPG_EINVAL, "spg_graph_kullback_leibler: the sparsified graph holds a vertex the baseline lacks");
                j++;
            }
            if (j < oo.size() && other->vid[oo[j]] == base->vid[v]) { kept_b.push_back(v); kept_o.push_back(oo[j]); j++; }
            else marg_b.push_back(v);
        }
        for (; j < oo.size(); j++)
            if (oo[j] != fo) return set_err(ctx, SPG_EINVAL, "spg_graph_kullback_leibler: the sparsified graph holds a vertex the baseline lacks");
    }
    if (kept_b.empty()) return set_err(ctx, SPG_EINVAL, "spg_graph_kullback_leibler: no common free vertex");
    const int64_t n_marg = (int64_t)d * marg_b.size(), n_keep = (int64_t)d * kept_b.size();
    const int64_t Nm = (n_marg + 63) / 64 * 64, Ng = (n_keep + 63) / 64 * 64;
    const bool sparse = ctx->linear_solver == SPG_SOLVER_SPARSE || (ctx->linear_solver == SPG_SOLVER_AUTO && Nm + Ng > 46000);
    if (!sparse && Nm + Ng > 46000) return set_err(ctx, SPG_ECAPACITY, "spg_graph_kullback_leibler: dense formulation limited to 46k variables (16 GB)");
    if (int rc = sync_device(base)) return rc;
    if (int rc = sync_device(other)) return rc;
    if (int rc = ctx->be.synchronize(ctx->be.user)) return rc;
    if (other->ctx != ctx) if (int rc = other->ctx->be.synchronize(other->ctx->be.user)) return rc;
    if (sparse) {
        // block-sparse multifrontal path: positions only number the blocks, the elimination order is the plan's
        DenseStage sb, so;
        sb.pos.assign(base->vid.size(), -1);
        so.pos.assign(other->vid.size(), -1);
        std::vector<uint8_t> is_marg(base->vid.size(), 0);
        std::vector<int64_t> kvb, kvo;
        int p = 0;
        for (int32_t v : ob) if (v != fb) sb.pos[v] = p++;
        for (int32_t v : marg_b) is_marg[v] = 1;
        p = 0;
        for (size_t i = 0; i < kept_b.size(); i++) {
            so.pos[kept_o[i]] = p++;
            kvb.push_back(base->vpose[kept_b[i]]);
            kvo.push_back(other->vpose[kept_o[i]]);
        }
        build_dense_stage(base, sb);
        build_dense_stage(other, so);
        double terms[6] = {0, 0, 0, 0, 0, 0}, secs = 0, info[4] = {0, 0, 0, 0};
        ctx->err[0] = 0;
        int rc = spg::hip_sparse_kld(spg::hip_backend_stream(&ctx->be), sb.in, so.in, is_marg.data(), kept_b.data(), kept_o.data(), (int)kept_b.size(),
                                     kvb.data(), kvo.data(), terms, &secs, info, ctx->err, sizeof ctx->err);
        if (rc) return rc;
        out->kld = terms[0]; out->innerprod = terms[1]; out->mahalanobis = terms[2]; out->logdetx = terms[3];
        out->logdety = terms[4]; out->n = (int64_t)terms[5]; out->n_marginalized = n_marg; out->device_seconds = secs;
        out->solver = SPG_SOLVER_SPARSE; out->supernodes = (int32_t)info[0]; out->front_bytes = info[2]; out->factor_flops = info[3];
        return 0;
    }
    DenseStage sb, so;
    sb.pos.assign(base->vid.size(), -1);
    so.pos.assign(other->vid.size(), -1);
    std::vector<int64_t> kvb, kvo;
    {
        int p = 0;
        for (int32_t v : marg_b) { sb.pos[v] = p; p += d; }
        p = (int)Nm;
        int q = 0;
        for (size_t i = 0; i < kept_b.size(); i++) {
            sb.pos[kept_b[i]] = p; p += d;
            so.pos[kept_o[i]] = q; q += d;
            kvb.push_back(base->vpose[kept_b[i]]);
            kvo.push_back(other->vpose[kept_o[i]]);
        }
    }
    build_dense_stage(base, sb);
    build_dense_stage(other, so);
    double terms[6] = {0, 0, 0, 0, 0, 0}, secs = 0;
    ctx->err[0] = 0;
    int rc = spg::hip_dense_kld(spg::hip_backend_stream(&ctx->be), sb.in, so.in, (int)n_marg, (int)n_keep, kvb.data(), kvo.data(),
                                terms, &secs, ctx->err, sizeof ctx->err);
    if (rc) return rc;
    out->kld = terms[0]; out->innerprod = terms[1]; out->mahalanobis = terms[2]; out->logdetx = terms[3];
    out->logdety = terms[4]; out->n = (int64_t)terms[5]; out->n_marginalized = n_marg; out->device_seconds = secs;
    out->solver = SPG_SOLVER_DENSE; out->supernodes = 0; out->front_bytes = 0; out->factor_flops = 0;
    return 0;
}

// ================================================================================= optimize() (8f.1)
static int optimize_with_fixed(spg_graph *g, int iterations, const std::vector<int32_t> &fixed_vertices, spg_optimize_stats *out) {
    spg_ctx *ctx = g->ctx;
    std::vector<int32_t> order = live_vertices_by_id(g);
    std::vector<uint8_t> is_fixed(g->vid.size(), 0);
    for (int32_t v : fixed_vertices) is_fixed[v] = 1;
    int64_t n = 0;
    for (int32_t v : order) if (!is_fixed[v]) n += g->d;
    // dense up to 12 k unknowns (two n^2 matrices, an n^3 / 3 factorisation per trial), block-sparse beyond
    const bool sparse = ctx->linear_solver == SPG_SOLVER_SPARSE || (ctx->linear_solver == SPG_SOLVER_AUTO && n > 12000);
    if (!sparse && n > 32000) return set_err(ctx, SPG_ECAPACITY, "spg_graph_optimize: dense formulation limited to 32k variables (2 x 8 GB)");
    if (int rc = sync_device(g)) return rc;
    if (int rc = ctx->be.synchronize(ctx->be.user)) return rc;
    DenseStage st;
    st.pos.assign(g->vid.size(), -1);
    int p = 0;
    for (int32_t v : order) if (!is_fixed[v]) { st.pos[v] = p; p += g->d; }
    build_dense_stage(g, st);
    double stats[5] = {0, 0, 0, 0, 0}, secs = 0, info[4] = {0, 0, 0, 0};
    ctx->err[0] = 0;
    int rc = (sparse && n > 0)
                 ? spg::hip_sparse_optimize(spg::hip_backend_stream(&ctx->be), st.in, (int)n, iterations, stats, &secs, info, ctx->err, sizeof ctx->err)
                 : spg::hip_dense_optimize(spg::hip_backend_stream(&ctx->be), st.in, (int)n, iterations, stats, &secs, ctx->err, sizeof ctx->err);
    // the estimates changed on the device: refresh the host mirror's copies
    if (int rc2 = sync_host(g)) return rc2;
    {
        // one download of the arena range that holds the free vertices' poses (a copy per vertex costs ~30 us each:
        // 3 s for a 100 k-pose graph), then only the pose slots are taken over
        int64_t lo = INT64_MAX, hi = -1;
        for (int32_t v : order) if (!is_fixed[v]) { lo = std::min(lo, g->vpose[v]); hi = std::max(hi, g->vpose[v] + g->ps); }
        if (hi > lo) {
            std::vector<double> tmp((size_t)(hi - lo));
            if (int rc2 = ctx->be.download(ctx->be.user, tmp.data(), (char *)g->dev + lo * 8, hi - lo)) return rc2;
            for (int32_t v : order) if (!is_fixed[v]) memcpy(g->host.data() + g->vpose[v], tmp.data() + (g->vpose[v] - lo), (size_t)g->ps * 8);
        }
    }
    if (rc) return rc;
    if (out) {
        out->iterations = (int32_t)stats[0]; out->trials = (int32_t)stats[1];
        out->chi2_initial = stats[2]; out->chi2_final = stats[3]; out->lambda_final = stats[4]; out->device_seconds = secs;
        out->n = n;
        out->solver = (sparse && n > 0) ? SPG_SOLVER_SPARSE : SPG_SOLVER_DENSE;
        out->supernodes = (int32_t)info[0]; out->front_bytes = info[2]; out->factor_flops = info[3];
    }
    return 0;
}

extern "C" int spg_sparse_plan(int n, const int32_t *ptr, const int32_t *adj, int pose_dim, const uint8_t *is_marg, int leaf,
                               spg_sparse_plan_info *info, int32_t *perm, int32_t *sn_first, int32_t *sn_parent, int32_t *sn_level,
                               int32_t *sn_rowptr, int32_t *rows, int32_t *rel, int64_t rows_cap) {
    if (n < 0 || !ptr || (pose_dim != 3 && pose_dim != 6) || !info) return SPG_EINVAL;
    // the row pointers come from the caller: 0-based, non-decreasing, non-negative total — before anything is read through them
    if (ptr[0] != 0) return SPG_EINVAL;
    for (int i = 0; i < n; i++) if (ptr[i + 1] < ptr[i]) return SPG_EINVAL;
    if (!adj && ptr[n] > 0) return SPG_EINVAL;
    spg::sparse::BlockGraph bg;
    bg.n = n;
    bg.ptr.assign(ptr, ptr + n + 1);
    bg.adj.assign(adj, adj + ptr[n]);
    for (int32_t u : bg.adj) if (u < 0 || u >= n) return SPG_EINVAL;
    spg::sparse::Plan P;
    spg::sparse::build_plan(bg, pose_dim, is_marg, leaf > 0 ? leaf : (pose_dim == 6 ? 32 : 64), P);
    info->n_supernodes = P.nsn; info->n_marg_supernodes = P.n_marg_sn; info->n_levels = P.nlevels; info->pad_ = 0;
    info->n_rows = (int64_t)P.rows.size(); info->front_bytes = 8.0 * (double)P.pool; info->flops = P.flops;
    if (perm) std::copy(P.perm.begin(), P.perm.end(), perm);
    if (sn_first) std::copy(P.first.begin(), P.first.end(), sn_first);
    if (sn_parent) std::copy(P.parent.begin(), P.parent.end(), sn_parent);
    if (sn_level) std::copy(P.level.begin(), P.level.end(), sn_level);
    if (sn_rowptr) std::copy(P.rowptr.begin(), P.rowptr.end(), sn_rowptr);
    if (rows_cap >= (int64_t)P.rows.size()) {
        if (rows) std::copy(P.rows.begin(), P.rows.end(), rows);
        if (rel) std::copy(P.rel.begin(), P.rel.end(), rel);
    }
    return 0;
}

extern "C" int spg_graph_optimize(spg_graph *g, int iterations, int32_t fixed_id, spg_optimize_stats *out) {
    if (!g || g->active || iterations < 0) return SPG_EINVAL;
    if (!g->ctx->is_hip) return set_err(g->ctx, SPG_ESTATE, "spg_graph_optimize needs the HIP backend");
    std::vector<int32_t> order = live_vertices_by_id(g);
    int fixed = resolve_fixed(g, order, fixed_id);
    if (fixed < 0) return set_err(g->ctx, SPG_EINVAL, "spg_graph_optimize: the fixed vertex is not in the graph");
    if (order.size() < 2) return set_err(g->ctx, SPG_EINVAL, "spg_graph_optimize: nothing to optimise");
    return optimize_with_fixed(g, iterations, std::vector<int32_t>{(int32_t)fixed}, out);
}

extern "C" int spg_graph_optimize_fixed(spg_graph *g, int iterations, const int32_t *fixed_ids, int n_fixed, spg_optimize_stats *out) {
    if (!g || g->active || iterations < 0 || n_fixed < 0 || (n_fixed > 0 && !fixed_ids)) return SPG_EINVAL;
    if (!g->ctx->is_hip) return set_err(g->ctx, SPG_ESTATE, "spg_graph_optimize_fixed needs the HIP backend");
    std::vector<int32_t> fx;
    for (int i = 0; i < n_fixed; i++) {
        auto it = g->vidx.find(fixed_ids[i]);
        if (it == g->vidx.end() || !g->valive[it->second]) return set_err(g->ctx, SPG_EINVAL, "spg_graph_optimize_fixed: a fixed vertex is not in the graph");
        fx.push_back(it->second);
    }
    return optimize_with_fixed(g, iterations, fx, out);
}

extern "C" int spg_graph_chi2(spg_graph *g, double *chi2) {
    if (!g || !chi2 || g->active) return SPG_EINVAL;
    // zero iterations with every vertex fixed: the optimiser's entry evaluates chi2 and returns
    std::vector<int32_t> all;
    for (size_t i = 0; i < g->vid.size(); i++) if (g->valive[i]) all.push_back((int32_t)i);
    if (!g->ctx->is_hip) return set_err(g->ctx, SPG_ESTATE, "spg_graph_chi2 needs the HIP backend");
    spg_optimize_stats st{};
    int rc = optimize_with_fixed(g, 1, all, &st);
    if (rc) return rc;
    *chi2 = st.chi2_initial;
    return 0;
}
