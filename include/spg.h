/* include/spg.h — C ABI of the MI355X pose-graph sparsification hot path.
 *
 * The reference (Lecanyu/SparsifyPoseGraph) has no FFI: its seams are C++ virtual interfaces.
 * Each entry point below names the reference interface it stands in for (file:line under the
 * reference tree). The product library libspg_hip.so exports every symbol declared here; the CPU
 * oracle (oracle/libspg_ref.so, test infrastructure only) exports spg_marginalize_batch and
 * spg_run_round with the same signatures so parity tests call both through one binding.
 *
 * Conventions: plain pointers and sizes; caller owns every buffer it passes; the library owns the
 * context/graph objects it returns. All arithmetic is fp64, graph indices int32, arena offsets int64
 * (counted in doubles). Functions return 0 on success or a negative SPG_E* code; they never abort.
 */
#ifndef SPG_H_
#define SPG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums mirroring reference types ------------------------------------------------------- */
/* EvaluateInfo::algorithm (src/evaluate.h:20-22) / GraphWrapperG2O(useGLC) (src/graph_wrapper_g2o.cpp:102) */
enum { SPG_ALG_NFR = 0, SPG_ALG_GLC = 1 };
/* SparsityOptions::SparsityTopology (src/sparsity_options.h:12-14), same numeric order */
enum { SPG_TOPO_TREE = 0, SPG_TOPO_SUBGRAPH = 1, SPG_TOPO_CLIQUEY_SUBGRAPH = 2, SPG_TOPO_DENSE = 3,
       SPG_TOPO_CLIQUEY_DENSE = 4 };
/* SparsityOptions::LinearizationPoint (src/sparsity_options.h:16-18) */
enum { SPG_LIN_LOCAL = 0, SPG_LIN_GLOBAL = 1 };
/* edge kinds: pose-pose edge (EdgeSE2ISAM / EdgeSE3ISAM, src/se2_compatibility.h:20, src/se3_compatibility.h:25)
 * and n-ary GLC edge (GLCEdge, src/glc_edge.h:14) */
enum { SPG_EDGE_BINARY = 0, SPG_EDGE_GLC = 1, SPG_EDGE_MULTI = 2 };
/* SPG_EDGE_MULTI = MultiEdgeCorrelated<EdgeSE2 | EdgeSE3> (src/multi_edge_correlated.hpp:28-140): nm pose-pose measurements
 * between pairs of the edge's q vertices with ONE joint information matrix Omega (d nm x d nm), what the CliqueySubgraph /
 * CliqueyDense patterns emit (src/topology_provider_binary.hpp:48-67). Record (doubles):
 *   [0] nm | [1 + 2 i], [2 + 2 i]: the two vertices of measurement i as indices into the edge's vertex list |
 *   nm measurements (3 | 7 each) | W (r x r row-major, r = d nm) with Omega = W^T W.
 * The error is the stack of the pose-pose errors; chi2 = || W e ||^2. */
#define SPG_MULTI_LEN(d, nm) (1 + 2 * (nm) + (nm) * ((d) == 3 ? 3 : 7) + (int64_t)(d) * (nm) * (d) * (nm))

/* per-blanket status (replaces the reference's assert / exit(0) / NULL-edge failure modes,
 * src/vertex_remover.cpp:291,461, src/optimizer.cpp:75-77, src/topology_provider_glc.cpp:46-49,85-89) */
enum {
    SPG_OK = 0,
    SPG_ST_HMM_NOT_PD = 1,      /* LLT(H_mm) failed (src/vertex_remover.cpp:444) */
    SPG_ST_EIG_FAIL = 2,        /* eigen-decomposition did not converge */
    SPG_ST_NONFINITE = 3,       /* non-finite value in the target information */
    SPG_ST_TIKHONOV_NOT_PD = 4, /* LLT(Lambda + I) failed (src/pseudo_chow_liu.cpp:189) */
    SPG_ST_CLOSED_FORM_NOT_PD = 5, /* LLT(J Sigma J^T) failed (src/logdet_function.cpp:246,273) */
    SPG_ST_KLD_NOT_PD = 6,      /* value() returned +inf (src/logdet_function.cpp:130-131); edges still valid */
    SPG_ST_NEEDS_INTERIOR_POINT = 7, /* no closed form (src/optimizer.cpp:38-79): out of scope, no edges written */
    SPG_ST_MARGINAL_NOT_PD = 8, /* LLT inside PseudoChowLiu::marginal failed (src/pseudo_chow_liu.cpp:134) */
    SPG_ST_EMPTY_BLANKET = 9,   /* removed vertex without edges (assert at src/vertex_remover.cpp:291) */
    SPG_ST_UNSUPPORTED = 10,    /* option combination the reference asserts against or that is out of scope */
    SPG_ST_NEEDS_LOCAL_OPTIMIZATION = 11 /* Local linearisation point without a closed-form estimate where the 10 LM iterations of
                                   src/vertex_remover.cpp:382-391 cannot run: a blanket that holds a GLC edge (the GLC provider
                                   asserts the Global point, src/topology_provider_glc.cpp:110-111). Pose-pose and correlated
                                   blankets of every NFR pattern, clusters included, are re-linearised on the device. */
};
/* informational bits OR-ed into status << 8 are not used; see spg_result.info */
enum {
    SPG_INFO_RANK_DEFICIENT = 1, /* smalleigs > dim: chooseDimensions path taken (src/logdet_function.cpp:42-59) */
    SPG_INFO_GLC_ROOT_EDGE = 2,  /* a unary GLC root edge survived the 1e-8 cut (src/topology_provider_glc.cpp:134-140) */
    SPG_INFO_IP_HESSIAN_NOT_PD = 4 /* interior point: a Newton system was not positive definite and that barrier step was given up
                                    (src/pqn/pqn_optimizer.cpp:48-53 solves with the failed LLT unchecked; the loop over rho,
                                    src/optimizer.cpp:60-75, goes on from the same x either way) */
};

enum {
    SPG_EINVAL = -1, SPG_ENODEV = -2, SPG_ENOMEM = -3, SPG_ECAPACITY = -4, SPG_EHIP = -5, SPG_EIO = -6,
    SPG_ESTATE = -7, SPG_EBLANKET = -8, /* at least one blanket has status != OK that prevents graph update */
    SPG_ENOTPD = -9 /* a graph's information matrix is not positive definite (global KLD) */
};

/* SparsityOptions (src/sparsity_options.h:11-30) + algorithm selector + pose dimension */
typedef struct {
    int32_t pose_dim;             /* 3 (SE2) or 6 (SE3) */
    int32_t algorithm;            /* SPG_ALG_* */
    int32_t topology;             /* SPG_TOPO_* */
    int32_t lin_point;            /* SPG_LIN_* */
    int32_t include_intra_clique; /* reference default true; never changed by any caller */
    int32_t flags;                /* SPG_FLAG_* */
    double chord_ratio;           /* reference default 1 */
} spg_options;
enum {
    SPG_FLAG_RESERVED0 = 1, /* bit 0 is reserved (the oracle uses it for a private GLC diagnostic) */
    SPG_FLAG_FORCE_EIG = 2  /* always take the eigen-decomposition route of src/logdet_function.cpp:14-64
                               (default: the equivalent gauge/Cholesky route whenever its guard holds) */
    /* bits 8..15: diagnostic pipeline truncation used by tools/phase_bench.py */
};

/* One batch of mutually independent Markov blankets in CSR form: the data VertexRemover::remove
 * gathers per iteration (src/vertex_remover.cpp:93-108): blanket vertices (removed first, then kept
 * in ascending original id = g2o indexMapping order, src/vertex_remover.cpp:349-356), their
 * estimates, and every edge with all endpoints in the blanket (src/vertex_remover.cpp:225-251). */
typedef struct {
    int32_t B;
    const int32_t *vert_off;      /* B+1 */
    const int32_t *n_remove;      /* B : m_b >= 1 */
    const int32_t *vert_id;       /* V : original ids (only echoed into new_edge_vert) */
    const double *pose;           /* V x (3 | 7): (x y theta) | (tx ty tz qx qy qz qw) */
    const int32_t *edge_off;      /* B+1 */
    const int32_t *edge_kind;     /* E : SPG_EDGE_* */
    const int32_t *edge_vert_off; /* E+1 */
    const int32_t *edge_vert;     /* local vertex index inside the blanket */
    const int64_t *edge_data_off; /* E+1 */
    const double *edge_data;      /* binary: meas (3|7) + information upper triangle row-wise (6|21)
                                     GLC   : meas (d*q) + W row-major (r x d*q)  [r = (len-dq)/dq] */
} spg_batch;

/* Outputs of VertexRemover::remove for the batch: the new edges handed to updateInputGraph
 * (src/vertex_remover.cpp:500-546), plus the target information (src/vertex_remover.cpp:447-449)
 * and the per-blanket KLD (src/logdet_function.cpp:119-133). Capacities are supplied by the caller. */
typedef struct {
    double *target_info;          /* optional (NULL to skip): blanket b at target_info_off[b], n_b x n_b row-major */
    const int64_t *target_info_off; /* B+1 when target_info != NULL */
    int32_t *new_edge_off;        /* B+1, written */
    int32_t *new_edge_kind;       /* new_edge_cap */
    int32_t *new_edge_vert_off;   /* new_edge_cap+1 */
    int32_t *new_edge_vert;       /* ORIGINAL ids, new_edge_vert_cap */
    int64_t *new_edge_data_off;   /* new_edge_cap+1 */
    double *new_edge_data;        /* same record layout as spg_batch.edge_data */
    int32_t new_edge_cap, new_edge_vert_cap;
    int64_t new_edge_data_cap;
    double *kld;                  /* B */
    double *min_gap;              /* B, optional: smallest relative gap between consecutive Chow-Liu weights popped */
    int32_t *status;              /* B : SPG_ST_* */
    int32_t *info;                /* B, optional: SPG_INFO_* bits */
} spg_result;

/* ---- context ------------------------------------------------------------------------------ */
typedef struct spg_ctx spg_ctx;
/* one context per (host thread, device); owns a HIP stream and scratch. device = HIP ordinal.
 * Fails with SPG_ENODEV when no gfx950 device / code object is available: there is no CPU fallback. */
int spg_ctx_create(spg_ctx **out, int device);
/* Multi-GPU form (the reference has no counterpart: one process, no collective): one process per
 * GPU, rank r of nranks. nccl_unique_id = the SPG_UNIQUE_ID_BYTES bytes rank 0 obtained from spg_get_unique_id and
 * handed to every rank (any out-of-band channel: a file, MPI, torch.distributed's store); the call runs
 * ncclCommInitRank (RCCL, bound from librccl.so.1 at this call — single-GPU users never load it) and is
 * collective over the ranks. NULL id: no communicator (exchange through the caller's callback, or a single rank). */
#define SPG_UNIQUE_ID_BYTES 128
int spg_get_unique_id(void *id_out /* SPG_UNIQUE_ID_BYTES */);
int spg_ctx_create_ranks(spg_ctx **out, int device, int rank, int nranks, const void *nccl_unique_id);
int spg_ctx_rank(const spg_ctx *ctx);
int spg_ctx_nranks(const spg_ctx *ctx);
/* The built-in exchange of one sharded round: in-place ncclAllGather of the nranks equal chunks of
 * arena[region_off, region_off + nranks*chunk_len) (doubles; rank r owns chunk r) over RCCL / xGMI, on the
 * context's stream, then a stream synchronisation. What spg_graph_marginalize_ranks does per exchanged batch when
 * no callback is given. No-op for a context without a communicator. */
int spg_allgather_region(spg_ctx *ctx, void *arena, int64_t region_off, int64_t chunk_len);
void spg_ctx_destroy(spg_ctx *ctx);
const char *spg_last_error(spg_ctx *ctx);
/* HIP stream the context launches on (hipStream_t as void*), for event timing by the caller */
void *spg_ctx_stream(spg_ctx *ctx);
int spg_ctx_synchronize(spg_ctx *ctx);
/* per-launch timing of the blanket kernel with HIP events on the context stream (bench roofline leg):
 * enable/reset, then read the sums over the timed launches completed since: kernel milliseconds,
 * algorithmic HBM bytes (SURVEY.md 8d formula evaluated on the launched blankets), launches, blankets.
 * enable = 1 times every launch (what bench.py uses); enable = n > 1 only every n-th one; 0 switches timing
 * off. (Sampling does not pay on ROCm 7.2: launches without the trailing event record were measured to
 * cost more host time than the two records they save.) */
int spg_ctx_profile(spg_ctx *ctx, int enable);
int spg_ctx_profile_read(spg_ctx *ctx, double *kernel_ms, double *alg_bytes, int64_t *launches, int64_t *blankets);
/* the same for the persistent worker kernel (narrow batches are handed to one long-lived kernel per marginalisation
 * instead of being launched one by one): its runs, their total duration (HIP events around the kernel), the
 * algorithmic bytes and blankets it processed. Not included in spg_ctx_profile_read. */
int spg_ctx_profile_read_worker(spg_ctx *ctx, double *kernel_ms, double *alg_bytes, int64_t *runs, int64_t *blankets);
/* and for the large-blanket pipeline (GLC Dense blankets beyond the LDS kernel's capacity run dense in HBM on the fp64
 * matrix cores): blankets processed since the last spg_ctx_profile call, their total device time (HIP events), the
 * n^3-class flops of their factorisations, the largest (k + m) * pose_dim seen. */
int spg_ctx_profile_read_big(spg_ctx *ctx, double *kernel_ms, double *flops, int64_t *blankets, int32_t *n_max);

/* VertexRemover::remove restricted to its arithmetic, for B independent blankets at once
 * (replaces src/vertex_remover.cpp:108-132 + src/topology_provider_binary.hpp:23-70 +
 *  src/topology_provider_glc.cpp:100-185 + src/optimizer.cpp:16-22). Host pointers. */
int spg_marginalize_batch(spg_ctx *ctx, const spg_options *opts, const spg_batch *batch, spg_result *result);

/* ---- decimation (src/decimation.h:18-22, src/decimation.cpp:11-49) -------------------------- */
/* Each writes at most cap ids into out and returns the count (or the required count if > cap). */
int spg_decimate_global(int last, int endvert, int sparsity, int cluster_size, int32_t *out, int cap);
int spg_decimate_online(int last, int endvert, int sparsity, int cluster_size, int32_t *out, int cap);
int spg_decimate_cluster(int last, int endvert, int sparsity, int cluster_size, int32_t *out, int cap);

/* ---- device-resident graph: GraphWrapper (src/graph_wrapper.h:17-78) ------------------------- */
typedef struct spg_graph spg_graph;

/* GraphWrapperG2O(verbose,useGLC) (src/graph_wrapper_g2o.cpp:102): empty graph of SE2 (3) or SE3 (6) poses */
int spg_graph_create(spg_ctx *ctx, int pose_dim, spg_graph **out);
void spg_graph_destroy(spg_graph *g);
/* GraphWrapperG2O(fname, optimize=false, ...) (src/graph_wrapper_g2o.cpp:107-154), .g2o text:
 * VERTEX_SE2 / EDGE_SE2 / VERTEX_SE3:QUAT / EDGE_SE3:QUAT and GLC_EDGE (src/glc_edge.cpp:65-93). The estimates
 * are taken as stored (optimize=false); call spg_graph_optimize for the reference's optimize=true. */
int spg_graph_load_g2o(spg_ctx *ctx, const char *path, spg_graph **out);
/* GraphWrapper::write (src/graph_wrapper_g2o.cpp:467-470) incl. GLC_EDGE records (src/glc_edge.cpp:95-119) */
int spg_graph_write_g2o(spg_graph *g, const char *path);
/* GraphWrapper::write(std::ofstream &) (src/graph_wrapper.h:62): the same text into a malloc'ed buffer
 * (*text, *len bytes, NUL-terminated); release it with spg_free. */
int spg_graph_write_g2o_mem(spg_graph *g, char **text, size_t *len);
void spg_free(void *p);
/* GraphWrapperG2O::clonePortion(maxid) (src/graph_wrapper_g2o.cpp:334-356): a new graph on the same context with
 * the vertices of id <= maxid (estimates copied) and the edges whose endpoints all satisfy it. The reference then
 * calls optimize() on the clone; here that is the caller's next call (spg_graph_optimize). */
int spg_graph_clone_portion(spg_graph *g, int maxid, spg_graph **out);
/* GraphWrapperG2O::covariance() (src/graph_wrapper_g2o.cpp:368-373): inverse of spg_graph_information, dense on
 * the device (blocked fp64-MFMA Cholesky, triangular inverse, L^-T L^-1). Same conventions as
 * spg_graph_information: returns n; the n x n row-major matrix is written if cap >= n*n. */
int64_t spg_graph_covariance(spg_graph *g, int32_t fixed_id, double *out, int64_t cap);
/* GraphWrapper::Vertex::edges() (src/graph_wrapper.h:26): indices (into the order of spg_graph_get_edges) of the
 * live edges incident to vertex id, ascending. Returns the count (writes at most cap). */
int spg_graph_vertex_edges(spg_graph *g, int id, int32_t *edge_index, int cap);
/* addVertex / addEdge (src/graph_wrapper_g2o.cpp:214-247). pose/meas as in spg_batch; info = upper triangle */
int spg_graph_add_vertex(spg_graph *g, int id, const double *pose);
int spg_graph_add_edge(spg_graph *g, int from, int to, const double *meas, const double *info_upper);
/* bulk forms: poses n x (3|7); ij n x 2; records n x (meas + info upper triangle) */
int spg_graph_add_vertices(spg_graph *g, int n, const int32_t *ids, const double *poses);
int spg_graph_add_edges(spg_graph *g, int n, const int32_t *ij, const double *records);
/* n-ary GLC edge (GLCEdge::read, src/glc_edge.cpp:64-93) */
int spg_graph_add_glc_edge(spg_graph *g, int q, const int32_t *ids, int r, const double *meas, const double *W);
/* MultiEdgeCorrelated over q vertices (src/multi_edge_correlated.hpp:28-63): `record` in the SPG_EDGE_MULTI layout above,
 * len = SPG_MULTI_LEN(pose_dim, nm). */
int spg_graph_add_multi_edge(spg_graph *g, int q, const int32_t *ids, const double *record, int64_t len);
int spg_graph_pose_dim(const spg_graph *g);
int spg_graph_num_vertices(const spg_graph *g);
int spg_graph_num_edges(const spg_graph *g);
int64_t spg_graph_edge_data_size(const spg_graph *g); /* total doubles of all live edge records */
int64_t spg_graph_edge_vert_size(const spg_graph *g); /* total endpoints of all live edges */
/* vertices() / vertex(id)->estimate() (src/graph_wrapper.h:70-72): ids ascending, poses V x (3|7) */
int spg_graph_get_vertices(spg_graph *g, int32_t *ids, double *poses);
/* all live edges in insertion order; arrays sized from the three counters above (+1 for offsets) */
int spg_graph_get_edges(spg_graph *g, int32_t *kind, int32_t *vert_off, int32_t *vert_ids,
                        int64_t *data_off, double *data);
/* setEstimate (src/graph_wrapper_g2o.cpp:614-621) */
int spg_graph_set_estimate(spg_graph *g, int id, const double *pose);

typedef struct {
    int32_t n_removed;     /* vertices actually marginalised */
    int32_t n_rounds;      /* conflict-free rounds executed */
    int32_t n_new_edges;
    int32_t n_bad_status;  /* blankets with a status that is not SPG_OK / SPG_ST_KLD_NOT_PD */
    int32_t max_blanket;   /* largest k+m */
    int32_t n_launches;    /* kernel launches */
    double kld_sum;        /* sum of finite per-blanket KLD */
    double host_seconds;   /* host scheduling + graph update */
    double device_seconds; /* time blocked on the device (launch -> results visible) */
    double schedule_seconds; /* part of host_seconds: conflict-free round selection */
    double commit_seconds;   /* part of host_seconds: graph update */
    double launch_seconds;   /* part of device_seconds: descriptor upload + kernel launch calls */
    int32_t n_batches;         /* batches of blankets handed to the device (a round may be cut into several) */
    int32_t n_exchanged;       /* batches that were sharded over the ranks and all-gathered */
    double exchange_seconds;   /* time inside the exchange (collective + its synchronisation) */
    double exchanged_bytes;    /* bytes all-gathered (whole regions, all ranks' chunks) */
} spg_marg_stats;

/* GraphWrapperG2O::marginalizeNoOptimize (src/graph_wrapper_g2o.cpp:398-453): removes `which` with
 * the sequential semantics of VertexRemover::remove (src/vertex_remover.cpp:83-140), executed as
 * conflict-free rounds of independent blankets on the device. Does NOT run optimize()
 * (src/graph_wrapper_g2o.cpp:462); GraphWrapperG2O::marginalize = this + spg_graph_optimize. */
int spg_graph_marginalize(spg_graph *g, const int32_t *which, int n, const spg_options *opts,
                          spg_marg_stats *stats);
/* The same call for nranks cooperating processes (one per GPU), each holding an identical replica:
 * batches below the shard threshold are computed redundantly by every rank (no communication);
 * a batch at or above it is sharded, and `exchange` is called once for it — it must all-gather the
 * nranks equal chunks of arena[region_off, region_off + nranks*chunk_len) in place (rank r owns
 * chunk r), e.g. ncclAllGather / torch.distributed.all_gather_into_tensor, and return 0. */
typedef int (*spg_exchange_fn)(void *user, void *arena, int64_t region_off, int64_t chunk_len, int nranks, int rank);
/* exchange == NULL with nranks > 1: the built-in RCCL all-gather of the graph's context (spg_ctx_create_ranks with
 * the same rank / nranks); a callback overrides it (tests: gloo, host staging). */
int spg_graph_marginalize_ranks(spg_graph *g, const int32_t *which, int n, const spg_options *opts, int rank, int nranks,
                                spg_exchange_fn exchange, void *exchange_user, spg_marg_stats *stats);
/* Which batches are sharded. Default policy (threshold < 0): a cost model — a batch is sharded when the modelled
 * device time of this rank's slice plus one exchange is below the modelled time of the whole batch on one GPU
 * (per-blanket chain latency ~ n^2.5, blankets resident per GPU from the LDS carve-up, exchange = latency + bytes /
 * link rate; constants in csrc/spg_host.cpp). spg_graph_set_shard_threshold(g, n >= 0) replaces it by "at least n
 * blankets" (0 = always; used by tests). */
/* Which driver spg_graph_marginalize uses for a single-rank NFR Tree call at the stored estimates. Default (-1): the
 * streaming driver on the HIP backend (one blanket = one item of the persistent worker kernel's queue, committed as its
 * ready word arrives; blankets the worker does not take fall back to the batch driver), the batch driver on an injected
 * backend. seed >= 0: tests — the streaming driver also runs on an injected backend, its "device" completing the
 * blankets in flight in an order drawn from the seed (0: all, oldest first). seed = -2: never stream (A/B). */
int spg_graph_set_stream_emulation(spg_graph *g, int seed);
/* per-removed-vertex diagnostics of the last marginalize call, in processing order */
int spg_graph_last_blanket_count(const spg_graph *g);
int spg_graph_last_blankets(const spg_graph *g, int32_t *root_id, int32_t *round, int32_t *status,
                            int32_t *info, double *kld, double *min_gap);

/* computeSubstituteEdge (src/compute_substitute_edge.cpp:13-96), host-side: for the online / cluster
 * replay harness, when a new edge (*from,*to) points at an already marginalised vertex. g is the FULL
 * source graph; marginalized = ids removed so far; maxid = newest vertex. On return the marginalised
 * endpoint is replaced by the nearest surviving vertex, meas (3|7) and info_upper (6|21) are filled. */
int spg_graph_substitute_edge(spg_graph *g, const int32_t *marginalized, int n_marg, int maxid,
                              int *from, int *to, double *meas, double *info_upper);

/* ---- global Kullback-Leibler divergence ------------------------------------
 * other->information() / sparseInformation() (src/graph_wrapper_g2o.cpp:351-396): the dense
 * Gauss-Newton information of the graph at its stored estimates over all vertices except the fixed one
 * (fixed_id < 0: the smallest id, which is the vertex the reference fixes and skips), id order,
 * n = pose_dim * (V - 1). Returns n; the n x n row-major matrix is written if cap >= n*n. */
int64_t spg_graph_information(spg_graph *g, int32_t fixed_id, double *out, int64_t cap);
/* kullbackLeiblerDivergence(diff, infox, maty, InformationInformation) (src/utils.cpp:70-97):
 * kld = 0.5 * (innerprod + mahalanobis - logdetx - logdety - n), logdety = -sum log D(maty). */
typedef struct {
    double kld, innerprod, mahalanobis, logdetx, logdety;
    int64_t n;               /* variables compared: pose_dim * (kept vertices - fixed) */
    int64_t n_marginalized;  /* baseline variables marginalised out */
    double device_seconds;   /* HIP-event time of assembly + factorisations + solve */
    int32_t solver;          /* SPG_SOLVER_DENSE or SPG_SOLVER_SPARSE: what ran */
    int32_t supernodes;      /* sparse: fronts of the two factorisations' assembly trees */
    double front_bytes;      /* sparse: bytes of fronts in HBM */
    double factor_flops;     /* sparse: flops of the two factorisations */
} spg_kld_terms;
/* baseline->kullbackLeibler(other) (src/graph_wrapper_g2o.cpp:531-548): marginal of the baseline's
 * information onto other's vertices (computeIndices :472-499), estimateDifference (:550-575), then the
 * formula above. Dense on the device up to 46 k variables; beyond that (or when the context's linear solver is
 * SPG_SOLVER_SPARSE) the block-sparse multifrontal path: the baseline is factorised with its marginalised vertices
 * eliminated first, log det of the marginal from the kept supernodes, trace(maty^-1 infox) from the selected
 * inverse — the reference's dense n_g x n_g step (720 GB at 100 k poses) is never formed. other's vertices must
 * be a subset of the baseline's; both graphs must live on the same device. SPG_ENOTPD if either information
 * matrix is not PD. */
int spg_graph_kullback_leibler(spg_graph *baseline, spg_graph *other, int32_t fixed_id, spg_kld_terms *out);

/* ---- optimize() -------------------------------------------------------------
 * GraphWrapperG2O::optimize() (src/graph_wrapper_g2o.cpp:250-269): one vertex fixed (fixed_id < 0: the
 * smallest id), g2o's Levenberg-Marquardt for up to `iterations` iterations (the reference uses 50),
 * no robust kernel. On the device (Hessian assembly, Cholesky of H + lambda I, triangular solves, pose updates,
 * chi2): dense (blocked fp64-MFMA Cholesky) up to 12 k scalar variables, block-sparse multifrontal (nested
 * dissection on the host, fronts on the matrix cores: CHOLMOD's role) beyond — see spg_ctx_set_linear_solver.
 * The estimates of the graph are updated in place. */
enum { SPG_SOLVER_AUTO = 0, SPG_SOLVER_DENSE = 1, SPG_SOLVER_SPARSE = 2 };
/* Which factorisation spg_graph_optimize / _optimize_fixed / _kullback_leibler of graphs of this context use.
 * AUTO (default): by size. DENSE beyond its capacity (32 k / 46 k variables) returns SPG_ECAPACITY. */
int spg_ctx_set_linear_solver(spg_ctx *ctx, int solver);
typedef struct {
    int32_t iterations, trials;          /* LM iterations run; linear systems solved */
    double chi2_initial, chi2_final, lambda_final;
    int64_t n;                           /* scalar variables */
    double device_seconds;
    int32_t solver;                      /* SPG_SOLVER_DENSE or SPG_SOLVER_SPARSE: what ran */
    int32_t supernodes;                  /* sparse: fronts of the assembly tree */
    double front_bytes;                  /* sparse: bytes of fronts in HBM */
    double factor_flops;                 /* sparse: flops of one factorisation */
} spg_optimize_stats;
int spg_graph_optimize(spg_graph *g, int iterations, int32_t fixed_id, spg_optimize_stats *out);
/* Symbolic phase of the block-sparse solver on its own (host only; inspection and CPU tests): nested-dissection
 * elimination order and assembly tree of a block graph given as symmetric CSR adjacency. is_marg (may be NULL): blocks
 * to eliminate first (the global KLD's marginalised vertices). leaf <= 0: default leaf size. Every output array may be
 * NULL; rows / rel are written only if rows_cap >= info->n_rows. perm[position] = block; supernode s owns positions
 * [sn_first[s], sn_first[s+1]); rows[sn_rowptr[s] ..) = its boundary positions (ascending); rel = scalar offset of each
 * boundary block inside the PARENT's front ([pivot columns padded to 64 | boundary rows]). */
typedef struct {
    int32_t n_supernodes, n_marg_supernodes, n_levels, pad_;
    int64_t n_rows;
    double front_bytes, flops;
} spg_sparse_plan_info;
int spg_sparse_plan(int n_blocks, const int32_t *adj_ptr, const int32_t *adj, int pose_dim, const uint8_t *is_marg, int leaf,
                    spg_sparse_plan_info *info, int32_t *perm, int32_t *sn_first, int32_t *sn_parent, int32_t *sn_level,
                    int32_t *sn_rowptr, int32_t *rows, int32_t *rel, int64_t rows_cap);
/* The same with a set of vertices held fixed at their current estimates — the inner step of
 * GraphWrapperG2O::chi2(other) (src/graph_wrapper_g2o.cpp:503-529): fix other's vertices at other's
 * estimates, optimise the rest, read chi2. n_fixed may cover every vertex (chi2 is evaluated, nothing moves). */
int spg_graph_optimize_fixed(spg_graph *g, int iterations, const int32_t *fixed_ids, int n_fixed, spg_optimize_stats *out);
/* _so->chi2(): sum of e^T Omega e over all edges at the current estimates (src/graph_wrapper_g2o.cpp:501,519) */
int spg_graph_chi2(spg_graph *g, double *chi2);

/* ---- round-stepping form of the same call, for multi-GPU sharding ---------------------------
 * All ranks hold a replica and run the same deterministic scheduler; rank r computes its slice of
 * each round's blankets; the caller exchanges the round's output region of the arena between
 * ranks (one all-gather over RCCL/xGMI) between compute and commit. */
typedef struct {
    int32_t n_blankets;       /* blankets scheduled in this round (all ranks) */
    int32_t my_first, my_count; /* this rank's slice */
    int64_t region_off;       /* arena offset (doubles) of the round's output region */
    int64_t chunk_len;        /* doubles per rank chunk; region = nranks * chunk_len, rank r owns chunk r */
    int32_t exchange;         /* 1: blankets are sharded, the caller must all-gather the region before commit;
                                 0: the round is small (latency-bound), every rank computes all of it redundantly
                                    (bit-identical: no atomics, fixed reduction orders) and nothing is exchanged */
    int32_t pad_;
} spg_round_info;
int spg_graph_marginalize_begin(spg_graph *g, const int32_t *which, int n, const spg_options *opts,
                                int rank, int nranks);
/* n >= 0: rounds with fewer blankets than n are computed redundantly on every rank instead of being sharded +
 * exchanged (0 = always shard); n < 0 (default): the cost model described at spg_graph_marginalize_ranks */
int spg_graph_set_shard_threshold(spg_graph *g, int min_blankets);
/* returns 1 if a round was prepared (info filled), 0 when the removal list is exhausted */
int spg_graph_round_prepare(spg_graph *g, spg_round_info *info);
int spg_graph_round_compute(spg_graph *g);   /* asynchronous on the context stream */
int spg_graph_round_commit(spg_graph *g);    /* waits, reads back the region, applies the graph update */
int spg_graph_marginalize_end(spg_graph *g, spg_marg_stats *stats);
/* device (or, for an injected backend, host) address of the arena and its capacity in doubles;
 * the arena may be re-allocated by add_* calls but never between begin and end */
void *spg_graph_arena(spg_graph *g, int64_t *capacity);
/* reserve arena capacity up front (doubles) so the address stays fixed. Also sizes and touches the host-side buffers a
 * marginalisation writes (the host mirror of the arena, the edge / log containers, the scheduler's scratch): a first touch
 * inside the call is a page fault in the commit path — measured as 55 ms instead of 25 ms per 100 k-pose marginalisation on
 * graphs whose mirror had never been touched. Call it once after loading a graph that will be sparsified (3x the loaded
 * size covers sparsity 2). */
int spg_graph_reserve(spg_graph *g, int64_t arena_doubles);

/* ---- compute backend ------------------------------------------------------------------------
 * The device side of a round, as the product's HIP backend implements it and as a test may inject
 * it (spg_ctx_create_injected): tests/ use this to run the host scheduler and the multi-rank
 * exchange on CPU-only machines with oracle/libspg_ref.so as the arithmetic. The product never
 * injects anything: spg_ctx_create always binds the HIP backend and fails without a device. */
typedef struct {
    int32_t vert_begin, n_vert, n_remove; /* into vert_pose_off[] */
    int32_t edge_begin, n_edge;           /* into edge refs */
    int32_t n_new_max;                    /* most new edges this blanket can emit */
    int32_t n_new_vert_max;               /* most endpoints over all its new edges */
    int32_t pad_;                         /* scratch doubles the blanket's n-ary (GLC) edges need during assembly */
    int64_t new_off;                      /* arena offset where new-edge records are packed */
    int64_t new_len;                      /* doubles reserved at new_off */
    int64_t out_off;                      /* arena offset of the per-blanket output record */
    int64_t tinfo_off;                    /* arena offset for the target information, or -1 */
} spg_blanket_desc;
typedef struct {
    int64_t off;    /* arena offset of the edge record */
    int32_t len;    /* record length in doubles */
    int32_t kind;   /* SPG_EDGE_* */
    int32_t vbegin; /* into edge_vert[] (local blanket indices) */
    int32_t nv;
} spg_edge_ref;
/* per-blanket output record layout (doubles) at out_off:
 *   [0] status  [1] info bits  [2] kld  [3] min_gap  [4] n_new  [5] ready tag
 *   then per new edge e < n_new: [6+4e] kind  [7+4e] record offset relative to new_off
 *                                [8+4e] record length  [9+4e] nv
 *   then, starting at 6 + 4*n_new_max: the local vertex indices of the new edges, concatenated
 * record length = SPG_OUT_LEN(n_new_max, n_new_vert_max).
 * [5] is written LAST (after a system-scope release) with SPG_READY_WORD(spg_round_desc.tag) — 2^52 + tag,
 * a value no other word of a record can hold, so stale mailbox contents never look ready — once everything the graph
 * update needs (status, n_new, the new-edge table and the new records in the arena) is in place; the
 * per-blanket KLD [2] — and a status change to SPG_ST_KLD_NOT_PD — may land later, at the latest when
 * the launch has completed. A host that polls the mailbox can therefore commit a batch while its
 * KLD tails are still running.
 * When the record is complete (KLD included) the device overwrites [5] with SPG_FINAL_WORD(tag): a poller treats
 * either value as "ready", and FINAL as "this record will not change any more". */
#define SPG_OUT_HDR 6
#define SPG_READY_WORD(tag) (4503599627370496.0 + (double)(tag))
#define SPG_FINAL_WORD(tag) (4503599627370496.0 + 4294967296.0 + (double)(tag))
#define SPG_OUT_LEN(n_new_max, n_new_vert_max) (SPG_OUT_HDR + 4 * (n_new_max) + (n_new_vert_max))
typedef struct {
    const spg_options *opts;
    int32_t n_blankets, first, count;    /* compute blankets [first, first+count) */
    const spg_blanket_desc *blankets;    /* n_blankets */
    const int64_t *vert_pose_off;        /* arena offsets of the blanket vertices' poses */
    const spg_edge_ref *edges;
    const int32_t *edge_vert;
    int64_t n_vert_total, n_edge_total, n_edge_vert_total;
    /* optional host "mailbox": when mail_len > 0 and the backend has one, the out records of the
     * computed blankets are ALSO delivered to backend->mailbox()[out_off - mail_base] (pinned host
     * memory written by the kernel), so the host needs no device->host copy to read them */
    int64_t mail_base, mail_len;
    int32_t slot;   /* launch slot: batches in different slots may be in flight at the same time */
    int32_t tag;    /* launch tag: SPG_READY_WORD(tag) lands in word [5] of every out record of this launch */
} spg_round_desc;
typedef struct {
    void *user;
    void *(*alloc)(void *user, int64_t doubles);
    void (*release)(void *user, void *p);
    int (*upload)(void *user, void *dst, const double *src, int64_t doubles);   /* host -> arena */
    int (*download)(void *user, double *dst, const void *src, int64_t doubles); /* arena -> host */
    int (*run_round)(void *user, void *arena, const spg_round_desc *round);     /* may be asynchronous */
    int (*synchronize)(void *user);
    const double *(*mailbox)(void *user);   /* may be NULL: no mailbox, out records are downloaded */
    /* optional: two launch slots for overlapping the host work of one batch with the device work of
     * another. NULL = the backend runs one batch at a time */
    int (*synchronize_slot)(void *user, int slot);
    const double *(*mailbox_slot)(void *user, int slot);
} spg_backend;
int spg_ctx_create_injected(spg_ctx **out, const spg_backend *backend);

/* oracle-side twin of run_round on host memory (exported by oracle/libspg_ref.so only) */
int spg_run_round(double *arena, const spg_round_desc *round);

#ifdef __cplusplus
}
#endif
#endif /* SPG_H_ */
