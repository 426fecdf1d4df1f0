// tools/host_sim.cpp — host-side timing of the streaming driver without a GPU.
//
// The driver of csrc/spg_host.cpp (selection rule, commit, packets, doorbells, mailbox polling) runs unchanged; behind
// it a thread of this tool plays the persistent worker kernel: it takes the queue items the driver publishes, waits a
// fixed latency (the measured ticket -> ready word time of one blanket on MI355X) and writes the blanket's out record
// and new edge records — recorded beforehand from the CPU oracle (test infrastructure, dlopen'ed here as such) — into
// the mailbox and the arena. What comes out is the host's share of a marginalisation: busy time per removed vertex and
// the step time a device of that latency and unlimited width would allow. A development tool; never part of the product.
//
//   build:  g++ -O2 -std=c++17 -pthread tools/host_sim.cpp -o /tmp/host_sim -Lsparsifyposegraph_amd -lspg_hip -ldl \
//               -Wl,-rpath,$PWD/sparsifyposegraph_amd
//   run:    /tmp/host_sim graph.g2o [latency_us=40] [reps=3] [final_lag_us=10]
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../include/spg.h"
#include "../sparsifyposegraph_amd/csrc/spg_internal.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Rec { std::vector<double> out, recs; };
static std::unordered_map<int64_t, Rec> g_table;   // root pose offset -> results of its blanket
static int (*oracle_run_round)(double *, const spg_round_desc *) = nullptr;
static bool g_record = true;

static void *be_alloc(void *, int64_t n) { void *p = nullptr; if (posix_memalign(&p, 4096, (size_t)n * 8)) return nullptr; memset(p, 0, (size_t)n * 8); return p; }
static void be_release(void *, void *p) { free(p); }
static int be_upload(void *, void *d, const double *s, int64_t n) { memcpy(d, s, (size_t)n * 8); return 0; }
static int be_download(void *, double *d, const void *s, int64_t n) { memcpy(d, s, (size_t)n * 8); return 0; }
static int be_sync(void *) { return 0; }
static int be_run_round(void *, void *arena, const spg_round_desc *rd) {
    double *a = (double *)arena;
    if (!g_record) { fprintf(stderr, "host_sim: a batch reached run_round in replay mode (the stream fell back)\n"); return SPG_ESTATE; }
    int rc = oracle_run_round(a, rd);
    if (rc) return rc;
    for (int b = rd->first; b < rd->first + rd->count; b++) {
        const spg_blanket_desc &bd = rd->blankets[b];
        Rec r;
        r.out.assign(a + bd.out_off, a + bd.out_off + SPG_OUT_LEN(bd.n_new_max, bd.n_new_vert_max));
        r.recs.assign(a + bd.new_off, a + bd.new_off + bd.new_len);
        g_table[rd->vert_pose_off[bd.vert_begin]] = std::move(r);
    }
    return 0;
}

struct Sim {
    spg::StreamPort port;
    std::thread th;
    std::atomic<bool> run{false};
    double latency = 40e-6, final_lag = 10e-6;
    unsigned long long consumed = 0;
    long n_items = 0;
    struct Item { double t_ready, t_final; unsigned long long *pkt; bool ready_done; };
    std::deque<Item> q;
    void complete(Item &it, bool fin) {
        unsigned long long *pk = it.pkt;
        double *arena = (double *)(uintptr_t)pk[0];
        double *cell = (double *)(uintptr_t)pk[1] + ((int64_t)pk[3] - (int64_t)pk[2]);
        const int tag = (int)(pk[10] >> 32);
        if (!fin) {
            const int64_t key = (int64_t)pk[spg::kPktHdr];
            auto f = g_table.find(key);
            if (f == g_table.end()) { fprintf(stderr, "host_sim: no recorded result for root pose offset %lld\n", (long long)key); abort(); }
            const Rec &r = f->second;
            memcpy(arena + (int64_t)pk[4], r.recs.data(), r.recs.size() * 8);
            // the worker's compact record (flags bit 20): six header words + the endpoint pairs as 4-bit local indices
            const int n_new = (int)r.out[4], n_new_max = (int)((pk[7] >> 32) & 0xffffffffu);
            unsigned long long w[2] = {0, 0};
            for (int e = 0; e < n_new; e++) {
                const unsigned long long pr = (unsigned long long)((int)r.out[SPG_OUT_HDR + 4 * n_new_max + 2 * e] & 15) | ((unsigned long long)((int)r.out[SPG_OUT_HDR + 4 * n_new_max + 2 * e + 1] & 15) << 4);
                w[e >> 3] |= pr << (8 * (e & 7));
            }
            for (int i = 0; i < 5; i++) cell[i] = r.out[i];
            memcpy(cell + 6, w, 16);
            std::atomic_thread_fence(std::memory_order_release);
            ((volatile double *)cell)[5] = SPG_READY_WORD(tag);
        } else {
            std::atomic_thread_fence(std::memory_order_release);
            ((volatile double *)cell)[5] = SPG_FINAL_WORD(tag);
        }
    }
    void loop() {
        while (run.load(std::memory_order_acquire)) {
            const unsigned long long tail = ((volatile unsigned long long *)port.q->tail)[0];
            std::atomic_thread_fence(std::memory_order_acquire);
            const double t = now_s();
            while (consumed < tail) {
                unsigned long long *pk = (unsigned long long *)(uintptr_t)port.q->item[consumed % spg::kQCap];
                q.push_back({t + latency, t + latency + final_lag, pk, false});
                consumed++; n_items++;
            }
            for (auto &it : q) { if (it.ready_done) continue; if (it.t_ready > t) break; complete(it, false); it.ready_done = true; }
            while (!q.empty() && q.front().ready_done && q.front().t_final <= t) { complete(q.front(), true); q.pop_front(); }
        }
    }
};

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: host_sim graph.g2o [latency_us] [reps] [final_lag_us]\n"); return 2; }
    const double lat = argc > 2 ? atof(argv[2]) * 1e-6 : 40e-6;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    const double flag = argc > 4 ? atof(argv[4]) * 1e-6 : 10e-6;
    void *oh = dlopen("oracle/libspg_ref.so", RTLD_NOW | RTLD_LOCAL);
    if (!oh) { fprintf(stderr, "dlopen oracle/libspg_ref.so: %s (run from the repository root)\n", dlerror()); return 1; }
    oracle_run_round = (int (*)(double *, const spg_round_desc *))dlsym(oh, "spg_run_round");
    spg_backend be{};
    be.alloc = be_alloc; be.release = be_release; be.upload = be_upload; be.download = be_download; be.run_round = be_run_round; be.synchronize = be_sync;
    spg_ctx *ctx = nullptr;
    if (spg_ctx_create_injected(&ctx, &be)) return 1;
    spg_options o{};
    o.pose_dim = 6; o.algorithm = SPG_ALG_NFR; o.topology = SPG_TOPO_TREE; o.lin_point = SPG_LIN_GLOBAL; o.include_intra_clique = 1; o.chord_ratio = 1.0;
    auto load = [&](spg_graph **g, std::vector<int32_t> &which) {
        if (spg_graph_load_g2o(ctx, argv[1], g)) { fprintf(stderr, "load failed: %s\n", spg_last_error(ctx)); exit(1); }
        const int nv = spg_graph_num_vertices(*g);
        std::vector<int32_t> ids(nv); std::vector<double> poses((size_t)nv * 7);
        spg_graph_get_vertices(*g, ids.data(), poses.data());
        which.clear();
        const int last = ids.back();
        for (int i = 4; i <= last; i++) if (i % 2) which.push_back(i);
        int64_t cap = 0; spg_graph_arena(*g, &cap);
        spg_graph_reserve(*g, (int64_t)nv * 7 * 3 + (int64_t)spg_graph_num_edges(*g) * 28 * 3);
    };
    // ---- record: the batch driver with the oracle as the arithmetic
    {
        spg_graph *g; std::vector<int32_t> which;
        load(&g, which);
        spg_marg_stats st{};
        double t0 = now_s();
        int rc = spg_graph_marginalize(g, which.data(), (int)which.size(), &o, &st);
        fprintf(stderr, "record: rc %d, %d removed in %d rounds, %.2f s (oracle arithmetic), %zu results; batch driver host %.3f ms (schedule %.3f, commit %.3f) = %.1f ns per vertex\n", rc, st.n_removed, st.n_rounds, now_s() - t0, g_table.size(),
                1e3 * st.host_seconds, 1e3 * st.schedule_seconds, 1e3 * st.commit_seconds, 1e9 * st.host_seconds / (st.n_removed ? st.n_removed : 1));
        spg_graph_destroy(g);
    }
    g_record = false;
    // ---- replay through the streaming driver against the simulated worker
    Sim sim;
    sim.latency = lat; sim.final_lag = flag;
    const int slots = 2048, stride = 8;
    static spg::WorkQ wq;
    memset(&wq, 0, sizeof wq);
    std::vector<unsigned long long> pkt((size_t)slots * spg::kPktWords);
    std::vector<double> mail((size_t)slots * stride, 0.0);
    sim.port.pkt = pkt.data(); sim.port.q = &wq; sim.port.tail = 0; sim.port.bells = 1;
    sim.port.h_mail = mail.data(); sim.port.d_mail = (unsigned long long)(uintptr_t)mail.data(); sim.port.mail_stride = stride; sim.port.slots = slots;
    spg_debug_set_stream_port(ctx, &sim.port);
    sim.run.store(true);
    sim.th = std::thread([&] { sim.loop(); });
    for (int r = 0; r < reps; r++) {
        spg_graph *g; std::vector<int32_t> which;
        load(&g, which);
        spg_marg_stats st{};
        const double t0 = now_s();
        int rc = spg_graph_marginalize(g, which.data(), (int)which.size(), &o, &st);
        const double dt = now_s() - t0;
        printf("{\"rc\": %d, \"removed\": %d, \"ms\": %.3f, \"host_ms\": %.3f, \"idle_ms\": %.3f, \"doorbells\": %d, \"host_ns_per_vertex\": %.1f, \"latency_us\": %.1f, \"kld_sum\": %.9g}\n",
               rc, st.n_removed, 1e3 * dt, 1e3 * st.host_seconds, 1e3 * st.device_seconds, st.n_batches, 1e9 * st.host_seconds / (st.n_removed ? st.n_removed : 1), 1e6 * lat, st.kld_sum);
        fflush(stdout);
        spg_graph_destroy(g);
    }
    sim.run.store(false);
    sim.th.join();
    return 0;
}
