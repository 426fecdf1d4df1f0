// tests/cpp/ranks_demo.cpp — the multi-GPU driver from C++ (one process per rank), as a C++ caller of the
// reference would use it: spg_ctx_create_ranks + spg_graph_marginalize_ranks.
//
//   ranks_demo <mode> <rank> <nranks> <graph.g2o> <rendezvous-file> [device]
//     mode rccl : the library's built-in exchange (ncclAllGather over RCCL/xGMI). Rank 0 writes the 128-byte unique id
//                 into <rendezvous-file>, the others wait for it. One GPU per rank (RCCL refuses two ranks on one
//                 device).
//     mode shm  : the same driver with a caller-supplied exchange: chunks staged through a shared file mapping, a
//                 sense-reversing barrier in it. Lets several ranks share ONE GPU (the test box has one).
//   Every rank removes the odd vertices >= 4 with every batch sharded (threshold 0), then repeats the job alone on a
//   second copy of the graph and compares the two results record by record.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spg.h"

#define REQ(x) do { int rc_ = (x); if (rc_ < 0) { std::fprintf(stderr, "rank %d: %s failed (%d): %s\n", g_rank, #x, rc_, g_ctx ? spg_last_error(g_ctx) : ""); return 10; } } while (0)

static int g_rank = 0;
static spg_ctx *g_ctx = nullptr;

struct Shm {
    std::atomic<int> arrived, sense;
    char pad[56];
    double data[1];   // nranks * chunk doubles (mapping is sized generously)
};
struct ShmExchange { Shm *shm; size_t cap_doubles; int nranks; int local_sense = 0; };

static void barrier(ShmExchange *x) {
    x->local_sense ^= 1;
    if (x->shm->arrived.fetch_add(1) + 1 == x->nranks) { x->shm->arrived.store(0); x->shm->sense.store(x->local_sense); }
    else while (x->shm->sense.load() != x->local_sense) std::this_thread::yield();
}

// spg_exchange_fn: all-gather the nranks chunks of arena[region_off ...) in place, staged through the mapping
static int shm_exchange(void *user, void *arena, int64_t region_off, int64_t chunk_len, int nranks, int rank) {
    ShmExchange *x = (ShmExchange *)user;
    if ((size_t)(chunk_len * nranks) > x->cap_doubles) return SPG_ECAPACITY;
    double *dev = (double *)arena + region_off;
    if (hipMemcpy(x->shm->data + (size_t)rank * chunk_len, dev + (size_t)rank * chunk_len, (size_t)chunk_len * 8, hipMemcpyDeviceToHost) != hipSuccess) return SPG_EHIP;
    barrier(x);
    for (int r = 0; r < nranks; r++)
        if (r != rank && hipMemcpy(dev + (size_t)r * chunk_len, x->shm->data + (size_t)r * chunk_len, (size_t)chunk_len * 8, hipMemcpyHostToDevice) != hipSuccess) return SPG_EHIP;
    barrier(x);   // nobody overwrites the staging area before everyone has read it
    return 0;
}

// live edges as (vertex ids, record), sorted: batch composition (hence insertion order) differs between the two runs
typedef std::pair<std::vector<int32_t>, std::vector<double>> EdgeRec;
static std::vector<EdgeRec> dump_edges(spg_graph *g) {
    int ne = spg_graph_num_edges(g);
    std::vector<int32_t> kind(ne + 1), voff(ne + 2), vids((size_t)spg_graph_edge_vert_size(g) + 1);
    std::vector<int64_t> doff(ne + 2);
    std::vector<double> data((size_t)spg_graph_edge_data_size(g) + 1);
    std::vector<EdgeRec> out;
    if (spg_graph_get_edges(g, kind.data(), voff.data(), vids.data(), doff.data(), data.data()) != ne) return out;
    for (int e = 0; e < ne; e++)
        out.push_back({std::vector<int32_t>(vids.begin() + voff[e], vids.begin() + voff[e + 1]), std::vector<double>(data.begin() + doff[e], data.begin() + doff[e + 1])});
    std::sort(out.begin(), out.end());
    return out;
}

int main(int argc, char **argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: ranks_demo <rccl|shm> <rank> <nranks> <graph.g2o> <rendezvous-file> [device]\n"); return 2; }
    const std::string mode = argv[1];
    const int rank = std::atoi(argv[2]), nranks = std::atoi(argv[3]);
    const char *path = argv[4], *rdv = argv[5];
    const int device = argc > 6 ? std::atoi(argv[6]) : 0;
    g_rank = rank;
    ShmExchange xs{nullptr, 0, nranks};
    if (mode == "rccl") {
        unsigned char id[SPG_UNIQUE_ID_BYTES];
        if (rank == 0) {
            REQ(spg_get_unique_id(id));
            std::string tmp = std::string(rdv) + ".tmp";
            FILE *f = std::fopen(tmp.c_str(), "wb");
            std::fwrite(id, 1, sizeof id, f);
            std::fclose(f);
            std::rename(tmp.c_str(), rdv);
        } else {
            FILE *f = nullptr;
            for (int i = 0; i < 3000 && !(f = std::fopen(rdv, "rb")); i++) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            if (!f || std::fread(id, 1, sizeof id, f) != sizeof id) { std::fprintf(stderr, "rank %d: no unique id\n", rank); return 3; }
            std::fclose(f);
        }
        int rc = spg_ctx_create_ranks(&g_ctx, device, rank, nranks, id);
        if (rc) { std::printf("rank %d: communicator refused (%d)\n", rank, rc); return 4; }
    } else {
        REQ(spg_ctx_create_ranks(&g_ctx, device, rank, nranks, nullptr));
        const size_t bytes = (size_t)64 << 20;
        int fd = open(rdv, O_RDWR | O_CREAT, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { std::perror("rendezvous file"); return 3; }
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { std::perror("mmap"); return 3; }
        xs.shm = (Shm *)m;   // a fresh file is zero-filled: arrived = sense = 0
        xs.cap_doubles = (bytes - sizeof(Shm)) / 8;
    }
    if (spg_ctx_rank(g_ctx) != rank || spg_ctx_nranks(g_ctx) != nranks) return 5;
    spg_graph *g = nullptr, *solo = nullptr;
    REQ(spg_graph_load_g2o(g_ctx, path, &g));
    REQ(spg_graph_load_g2o(g_ctx, path, &solo));
    const int d = spg_graph_pose_dim(g), nv = spg_graph_num_vertices(g);
    std::vector<int32_t> ids(nv);
    std::vector<double> poses((size_t)nv * (d == 3 ? 3 : 7));
    REQ(spg_graph_get_vertices(g, ids.data(), poses.data()));
    std::vector<int32_t> which;
    for (int32_t i : ids) if (i >= 4 && i % 2) which.push_back(i);
    spg_options o{d, SPG_ALG_NFR, SPG_TOPO_TREE, SPG_LIN_GLOBAL, 1, 0, 1.0};
    spg_marg_stats st{}, st1{};
    REQ(spg_graph_set_shard_threshold(g, 0));   // every batch is sharded and exchanged
    REQ(spg_graph_marginalize_ranks(g, which.data(), (int)which.size(), &o, rank, nranks,
                                    mode == "rccl" ? nullptr : shm_exchange, mode == "rccl" ? nullptr : &xs, &st));
    REQ(spg_graph_marginalize(solo, which.data(), (int)which.size(), &o, &st1));
    std::vector<EdgeRec> ea = dump_edges(g), eb = dump_edges(solo);
    const int na = (int)ea.size(), nb = (int)eb.size();
    double worst = 0;
    bool same = na == nb && na > 0;
    for (int e = 0; same && e < na; e++) {
        same = ea[e].first == eb[e].first && ea[e].second.size() == eb[e].second.size();
        for (size_t i = 0; same && i < ea[e].second.size(); i++)
            worst = std::max(worst, std::fabs(ea[e].second[i] - eb[e].second[i]) / std::max(1.0, std::fabs(eb[e].second[i])));
    }
    std::printf("rank %d/%d mode %s: removed %d in %d batches, %d exchanged (%.0f bytes, %.3f ms); edges %d vs %d alone, worst rel diff %.2e, kld %.12g vs %.12g\n",
                rank, nranks, mode.c_str(), st.n_removed, st.n_batches, st.n_exchanged, st.exchanged_bytes, 1e3 * st.exchange_seconds, na, nb, worst, st.kld_sum, st1.kld_sum);
    if (!same || worst > 1e-10 || st.n_removed != st1.n_removed || (nranks > 1 && st.n_exchanged == 0)) { std::printf("rank %d: MISMATCH\n", rank); return 6; }
    if (mode == "rccl" && nranks == 1) {
        // the collective itself at world size 1, on the live arena: ncclAllGather in place is the identity
        int64_t cap = 0;
        void *arena = spg_graph_arena(g, &cap);
        REQ(spg_allgather_region(g_ctx, arena, 0, cap));
        if (dump_edges(g) != ea) { std::printf("rank %d: allgather changed the arena\n", rank); return 7; }
        std::printf("rank %d: ncclAllGather of %lld doubles in place ok\n", rank, (long long)cap);
    }
    spg_graph_destroy(g);
    spg_graph_destroy(solo);
    spg_ctx_destroy(g_ctx);
    std::printf("rank %d ok\n", rank);
    return 0;
}
