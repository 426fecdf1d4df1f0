// csrc/spg_internal.h — declarations shared between the HIP translation unit and the host code.
#pragma once
#include <cstddef>
#include <cstdint>
#include "../../include/spg.h"

namespace spg {

// HIP backend (spg_kernels.hip)
int hip_backend_create(int device, spg_backend *out, char *errbuf, size_t errlen);
void hip_backend_destroy(spg_backend *b);
void *hip_backend_stream(spg_backend *b);
const char *hip_backend_error(spg_backend *b);
int hip_backend_launches(spg_backend *b);
int hip_backend_device(spg_backend *b);
void hip_backend_profile(spg_backend *b, int enable);
void hip_backend_profile_read(spg_backend *b, double *ms, double *bytes, long long *launches, long long *blankets);
void hip_backend_profile_read_worker(spg_backend *b, double *ms, double *bytes, long long *runs, long long *blankets);
void hip_backend_profile_read_big(spg_backend *b, double *ms, double *flops, long long *count, int *nmax);
int hip_backend_end_of_call(spg_backend *b);   // the persistent worker retires (end of a marginalisation)

// ---- queue of the persistent worker kernel (blanket_worker, spg_kernels.hip): fine-grained device memory the host
// fills through the PCIe BAR. Shared by the batch hand-over of hip_run_round and by the streaming driver of spg_host.cpp.
constexpr int kQCap = 16384;          // queue slots (more items than this are never in flight)
constexpr int kPktHdr = 13;
constexpr int kPktWords = 192;        // largest packet a worker stages (1.5 KB of LDS); larger blankets are launched
constexpr int kPktStride = 64;        // the first pass reads 64 words blindly (packets are kPktWords apart)
constexpr int kWorkerMaxN = 36;       // largest target dimension n = d*k a worker takes (LDS of the worker is sized for it)
constexpr int kBells = 16, kBellStride = 512;   // doorbell copies, 4 KB apart (idle workgroups poll: spread them over channels)
struct WorkQ {
    unsigned long long stop;          // 1: leave when no published item is left (host)
    unsigned long long pad[7];
    unsigned long long tail[kBells * kBellStride];   // items published (host writes every copy in use)
    unsigned long long item[kQCap];   // device address of the packet of item i (mod kQCap)
};

// What the streaming driver (spg_host.cpp: one blanket = one queue item, committed as its ready word arrives) needs from
// a backend with a persistent worker: the packet ring and the queue (both written through the BAR) and a pinned host
// mailbox with one out-record cell per in-flight blanket.
struct StreamPort {
    unsigned long long *pkt = nullptr;        // `slots` packets of kPktWords words (device memory, host-writable)
    WorkQ *q = nullptr;                       // the worker's queue (device memory, host-writable)
    unsigned long long tail = 0;              // host copy of q->tail (items published so far)
    int bells = 1;
    const double *h_mail = nullptr;           // pinned host mailbox: cell s = h_mail + s * mail_stride
    unsigned long long d_mail = 0;            // the same buffer in the device's address space
    int mail_stride = 0, slots = 0;
};
// Starts (or keeps) the persistent worker for pose dimension D and sizes the stream's packet ring / mailbox.
// 0 = ready, 1 = no streaming on this backend (no large BAR, worker disabled): take the batch driver, < 0 error.
int hip_stream_open(spg_backend *b, int D, int slots, int mail_stride, StreamPort *out);
// the blankets the stream handed to the worker (profile accounting of the worker run) and the host's tail copy
void hip_stream_close(spg_backend *b, const StreamPort *port, double alg_bytes, long long blankets);

// Dense global KLD (spg_dense.hip). Host-staged description of one graph for the dense assembly.
struct DenseGraphIn {
    int D = 0, nv = 0, ne = 0;
    const int32_t *pos = nullptr;       // [nv] scalar offset of the vertex block in the dense matrix, -1 = not a variable
    const int64_t *vpo = nullptr;       // [nv] pose offset in the arena
    const int32_t *rowptr = nullptr;    // [nv+1] CSR of incident live edges (indices into er, ascending)
    const int32_t *inc = nullptr;
    const spg_edge_ref *er = nullptr;   // [ne] live edges; vbegin indexes ev
    const int32_t *ev = nullptr;        // vertex indices of the edges
    int64_t n_ev = 0;
    const void *dev_arena = nullptr;    // device arena of the graph
};
int hip_dense_information(void *stream, const DenseGraphIn &in, int n, double *out, char *err, size_t errlen);
int hip_dense_covariance(void *stream, const DenseGraphIn &in, int n, double *out, char *err, size_t errlen);
int hip_dense_kld(void *stream, const DenseGraphIn &base, const DenseGraphIn &other, int n_marg, int n_keep,
                  const int64_t *kept_vpo_base, const int64_t *kept_vpo_other, double *terms, double *seconds,
                  char *err, size_t errlen);

// One GLC Dense blanket too large for the LDS kernel, dense in HBM on the fp64 matrix cores (spg_dense.hip)
int hip_big_glc_dense(void *stream, const DenseGraphIn &local_graph, int m, int k, int Nm, int64_t new_off, double *orec, int n_new_max, int tag,
                      double *seconds, double *flops, char *err, size_t errlen);
// test harness of the generic kernel's workgroup linear algebra (csrc/spg_nfr_ip.hip: la_test_kernel), see spg_debug_la
int hip_la_test(int op, int M, int N, int K, int flags, int mode, double *A, int ra, int lda, double *B, int rb, int ldb, double *C, int rc, int ldc, int *ok);
// frees the scratch the large-blanket pipeline keeps between calls (device block + pinned staging); called when a backend goes
void hip_big_release_scratch();
int hip_dense_optimize(void *stream, const DenseGraphIn &in, int n, int iterations, double *stats, double *seconds,
                       char *err, size_t errlen);

// Block-sparse multifrontal path of the same two calls (spg_sparse.inc, symbolic phase in spg_sparse_plan.hpp).
// info[4]: supernodes, levels of the assembly tree, bytes of fronts, flops of one factorisation.
int hip_sparse_optimize(void *stream, const DenseGraphIn &in, int n, int iterations, double *stats, double *seconds, double *info,
                        char *err, size_t errlen);
int hip_sparse_kld(void *stream, const DenseGraphIn &base, const DenseGraphIn &other, const uint8_t *is_marg_vertex,
                   const int32_t *kept_b, const int32_t *kept_o, int nk, const int64_t *kept_vpo_base, const int64_t *kept_vpo_other,
                   double *terms, double *seconds, double *info, char *err, size_t errlen);

// Interior-point NFR (spg_nfr_ip.hip): blankets of the Dense / Subgraph patterns without a closed form, one workgroup
// each, everything in a per-blanket slice of a global workspace.
struct IpArgs {
    double *arena;
    const spg_blanket_desc *blk;
    const int64_t *vpo;
    const spg_edge_ref *er;
    const int32_t *ev;
    const int32_t *list;      // blanket indices of this launch (blockIdx.x -> list[blockIdx.x])
    double *ws;               // workspace, ws_stride doubles per blanket of the launch
    int64_t ws_stride;
    double *mail;             // pinned host mailbox for out records (or nullptr)
    int64_t mail_base;
    int topology, lin_point, tag;
    int lds_doubles;          // dynamic LDS of the launch: a blanket whose hot buffers fit keeps them there
    int eig_jacobi;           // diagnostic (SPG_EIG_JACOBI=1): Jacobi sweeps for the spectrum of large targets too
    int ip_untiled;           // diagnostic (SPG_IP_UNTILED=1): the column-at-a-time LDS factorisation instead of the register-tiled one
    double chord_ratio;
};
constexpr int kIpMaxVars = 8400;     // Newton systems of the interior point: d^2 E up to this (k = 22 SE3 / 43 SE2 poses under Dense: what
                                     // parking.g2o's largest blanket asks for); one workgroup factorises them: minutes per blanket at the top
int nfr_ip_pattern_size(int topology, double chord_ratio, int k);   // new edges of a blanket with k kept vertices (-1: correlated patterns)
int64_t nfr_ip_workspace(int D, int k, int m, int E, int closed, int64_t *hot);  // doubles of workspace one such blanket needs (*hot: its LDS-eligible part)
int hip_nfr_ip_launch(void *stream, int D, IpArgs a, int count, int n_closed, int64_t hot_max);   // n_closed of the blankets have a closed-form pattern

// RCCL binding (spg_rccl.cpp): librccl.so.1 is bound with dlopen when the first multi-rank context is created
int rccl_get_unique_id(void *id_out, char *err, size_t errlen);
int rccl_comm_create(int device, int rank, int nranks, const void *unique_id, void **handle, char *err, size_t errlen);
int rccl_allgather_f64(void *handle, void *arena, int64_t region_off, int64_t chunk_len, void *stream, char *err, size_t errlen);
void rccl_comm_destroy(void *handle);

}  // namespace spg

// tools/host_sim.cpp only (host-side timing of the streaming driver against a simulated device); not in include/spg.h
extern "C" int spg_debug_set_stream_port(spg_ctx *ctx, void *stream_port);
// (tests) one of the device linear-algebra routines of the generic NFR kernel on host arrays; device 0 must be usable
extern "C" int spg_debug_la(int op, int M, int N, int K, int flags, int mode, double *A, int ra, int lda, double *B, int rb, int ldb, double *C, int rc, int ldc, int *ok);
