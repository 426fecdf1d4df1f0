#!/usr/bin/env python3
"""bench.py — nodes marginalised per second on the synthetic 100k-pose SE3 graph (BASELINE.json).

One "step" = one full GraphWrapper::marginalizeNoOptimize of the globalDecimate(sparsity 2) removal
list (49 998 vertices) on a fresh, HBM-resident replica of the graph: host scheduling of the
conflict-free rounds + the per-blanket HIP kernel + graph update. Inputs (poses + edge records) are
resident in HBM before the timed region starts; per batch only int descriptors go up and the
per-blanket output records come back.

N > 1 (launched by torch.distributed.run, one rank per GPU): `value` is the ONE-graph figure the north star names — the
same 100k-pose graph replicated on every rank, marginalised through spg_graph_marginalize_ranks with the library's own
policy and its built-in RCCL all-gather (spg_ctx_create_ranks). On this workload the policy never exchanges: its batches
are ~200 independent blankets, each a 40 us dependent chain however they are split, so every rank computes the whole
graph (its own streaming driver) and the figure is FLAT in N by design ("scaling": "strong"). What N GPUs do buy —
N graphs at once, the reference's own job-level parallelism (src/evaluate.cpp:413-433) — is reported as
`multi_gpu.independent_graphs`; `multi_gpu.wide_rounds` is a graph whose batches ARE wide enough to be sharded
(~50 000 blankets each) and drives the RCCL exchange, with the ranks the communicator saw.

`--config parking`: BASELINE config 4 (parking.g2o, NFR Tree, sparsity 2; graph from tests/golden) with the same
contract; at N > 1 the ranks share one replicated graph and the line carries `exchanged_batches` / bytes all-gathered.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel — the persistent
worker kernel that processes the narrow batches, HIP-event timed on its own stream) and, at N = 1, `cpu_baseline` (the
CPU oracle: strictly sequential on 1 core = the reference's execution model, plus `rounds_all_cores`) and `parity` (the
device result of the last step against that sequential oracle run).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X datasheet fp64 vector
FLOP_PER_NODE_K4 = 240e3   # SURVEY.md 8d estimate for the k=4 SE3 blanket


def pin_to_gpu_numa_node(torch, index):
    """Run this process on the CPUs of the NUMA node the GPU hangs off (sysfs; silently skipped if anything is
    missing). The per-round host work talks to the GPU through PCIe stores / pinned-memory polls: from the
    far socket of the 2-socket MI355X hosts a step was measured 5 % slower."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return f"process pinned to the {len(cpus)} CPUs of NUMA node {node} (GPU {bdf})"
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--poses", type=int, default=100000)
    ap.add_argument("--ring", type=int, default=400)
    ap.add_argument("--sparsity", type=int, default=2)
    ap.add_argument("--lin-point", choices=["global", "local"], default="global",
                    help="linearisation point (BASELINE config 5 is Global; local = the reference's default, LM on the blankets)")
    ap.add_argument("--config", choices=["synthetic", "parking"], default="synthetic",
                    help="synthetic = BASELINE config 5 (the headline metric); parking = config 4 (parking.g2o from tests/golden)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-poses", type=int, default=0, help="0 = full workload")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal mode for a one-GPU box (never used by the driver): SPG_BENCH_REHEARSE=1 runs the N > 1 code
    # path with every rank on cuda:0 and the exchange staged through host memory over gloo.
    rehearse = os.environ.get("SPG_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = f"cuda:{local_rank}"
    numa_note = pin_to_gpu_numa_node(torch, local_rank)

    from sparsifyposegraph_amd import abi, g2o_io
    from sparsifyposegraph_amd.graph import DecimateOptions, GraphWrapperHIP, globalDecimate
    from sparsifyposegraph_amd.lib import Context
    from sparsifyposegraph_amd.parallel import marginalize_sharded

    ctx = Context(local_rank)  # raises without a gfx950 device: no CPU fallback

    if numa_note and os.environ.get("SPG_BENCH_PIN_CORE", "0") == "1":
        # optional (measured inconclusive on shared hosts, hence off): the runtime's helper threads exist now and
        # keep the whole node; the graph thread itself stays on one core of it (rank-dependent)
        try:
            node_cpus = sorted(os.sched_getaffinity(0))
            os.sched_setaffinity(0, {node_cpus[(2 * local_rank + 1) % len(node_cpus)]})
            numa_note += ", graph thread on one core"
        except Exception:
            pass
    if args.config == "parking":
        from tests import util as _util
        g, which, *_ = _util.load_golden("parking_full_nfr_tree")
        which = np.asarray(which, np.int32)
    else:
        # N > 1: the SAME graph on every rank (replicas of one graph)
        g = g2o_io.synth_sphere(n_poses=args.poses, ring=args.ring)
        last = int(g["ids"][-1])
        which = np.array(globalDecimate(last, last, DecimateOptions(args.sparsity)), np.int32)
    opts = abi.make_options(6, abi.ALG_NFR, abi.TOPO_TREE, abi.LIN_GLOBAL if args.lin_point == "global" else abi.LIN_LOCAL)

    n_rep = args.warmup + args.steps
    arena_need = int(len(g["ids"]) * 7 + len(g["edge_ij"]) * 28) * 3
    replicas = []
    for _ in range(n_rep):
        hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
        hg.reserve(arena_need)  # uploads poses + edge records: resident in HBM before timing
        replicas.append(hg)

    shared_graph = world > 1   # one replicated graph, the library's policy decides what is sharded + exchanged

    def run_on(hg, w, builtin=False):
        if world == 1:
            return hg.marginalizeNoOptimize(w, opts)
        if builtin:   # the library's own RCCL communicator (context from spg_ctx_create_ranks)
            return hg.marginalize_ranks(w, opts, rank, world)
        return marginalize_sharded(hg, w, opts, device=device)   # exchange (if the policy ever asks for one) through torch.distributed

    def run(hg):
        return run_on(hg, which)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        run(replicas[i])
    fence()
    EVENT_STRIDE = 1   # HIP events around every launch (sampling every n-th launch made the untimed launches slower)
    ctx.profile(0 if os.environ.get("SPG_BENCH_NO_EVENTS") == "1" else EVENT_STRIDE)   # (=1: A/B of the event overhead; no roofline then)
    t0 = time.perf_counter()
    stats = None
    for i in range(args.steps):
        stats = run(replicas[args.warmup + i])
    fence()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    wprof = ctx.profile_read_worker()
    ctx.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # N > 1 only, outside the timed region of `value`: the one-graph views, each guarded so that a failure costs a
    # field, not the bench line.
    multi = {}
    extras_guard = None
    if world > 1 and args.config == "synthetic":
        # The extra run below drives RCCL inside the library across GPUs — a path no one-GPU box can rehearse. If it hangs,
        # the headline measured above must not be lost with it: after 5 minutes rank 0 prints the line without the extra
        # and every rank leaves.
        import threading

        def _bail():
            if rank == 0:
                removed0 = stats["n_removed"]
                print(json.dumps({"metric": "nodes_marginalized_per_s", "value": removed0 * args.steps / dt, "unit": "nodes/s", "n_gpus": world,
                                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True,
                                  "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                                  "config": {"workload": f"synthetic SE3 sphere-spiral pose graph, {args.poses} poses, NFR Tree, globalDecimate sparsity {args.sparsity}, one replicated graph on {world} GPUs",
                                             "note": "multi_gpu.wide_rounds did not finish within 300 s and was abandoned; roofline / cpu_baseline are N = 1 fields"},
                                  "multi_gpu": {"wide_rounds": {"error": "timeout after 300 s (RCCL inside the library across GPUs: never rehearsed, one-GPU boxes)"}}}), flush=True)
            os._exit(0)
        extras_guard = threading.Timer(300.0, _bail)
        extras_guard.daemon = True
        extras_guard.start()

        def timed(fn, reps):
            fence()
            t1 = time.perf_counter()
            res = None
            for _ in range(reps):
                res = fn()
            fence()
            tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cpu" if rehearse else device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item()), res
        try:
            # (2) a graph whose rounds are wide (2 rings of 100 000 poses: ~50k independent blankets per
            #     batch), so that every batch IS sharded over the ranks and all-gathered (RCCL over xGMI)
            # ... through the library's OWN communicator (spg_ctx_create_ranks: ncclCommInitRank inside libspg_hip.so, rank 0's
            # unique id handed round through torch's store), falling back to torch's all-gather behind the exchange callback
            ctx2, rccl_note = ctx, None
            if not rehearse:
                from sparsifyposegraph_amd.lib import get_unique_id
                try:
                    store = dist.distributed_c10d._get_default_store()
                    if rank == 0:
                        store.set("spg_unique_id", get_unique_id())
                    ctx2 = Context.ranks(local_rank, rank, world, bytes(store.get("spg_unique_id")))
                except Exception as e:
                    rccl_note = f"spg_ctx_create_ranks failed ({str(e)[:120]}): exchange through torch.distributed instead"
            builtin = ctx2 is not ctx
            gw = g2o_io.synth_sphere(n_poses=200000, ring=100000)
            ww = np.array([i for i in range(4, 200000) if i % 2], np.int32)
            hw = [GraphWrapperHIP.from_dict(gw, ctx=ctx2) for _ in range(2)]
            for hg in hw:
                hg.reserve(int(len(gw["ids"]) * 7 + len(gw["edge_ij"]) * 28) * 3)
            itw = iter(hw)
            secs, stw = timed(lambda: run_on(next(itw), ww, builtin), len(hw))
            multi["wide_rounds"] = {"value": stw["n_removed"] * len(hw) / secs, "unit": "nodes/s", "scaling": "strong",
                                    "removed": stw["n_removed"], "batches": stw["n_batches"], "exchanged_batches": stw["n_exchanged"],
                                    "exchanged_bytes": stw["exchanged_bytes"], "exchange_seconds": stw["exchange_seconds"], "kld_sum": stw["kld_sum"],
                                    "ranks_seen_by_rccl": ctx2.nranks() if builtin else None,
                                    "exchange": "library: spg_ctx_create_ranks + in-place ncclAllGather" if builtin else "torch.distributed all_gather_into_tensor behind the exchange callback", "note": rccl_note,
                                    "what": "synthetic SE3 graph of 2 rings x 100 000 poses: batches of ~50k blankets sharded over the ranks + one all-gather each"}
        except Exception as e:
            multi["wide_rounds"] = {"error": str(e)[:200]}

    if extras_guard is not None:
        extras_guard.cancel()
    removed = stats["n_removed"]
    weak = False
    value = removed * args.steps / dt
    if world > 1:
        multi["independent_graphs"] = {"value": world * value, "unit": "nodes/s", "scaling": "weak",
                                       "what": f"in the timed region every rank marginalised a whole replica by itself (nothing was exchanged: exchanged_batches = {stats['n_exchanged']}), "
                                               f"i.e. {world} graphs of this size were sparsified in the time of one: the job-level parallelism of the reference (src/evaluate.cpp:413-433)"}
    out = {
        "metric": "nodes_marginalized_per_s", "value": value, "unit": "nodes/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1),
        "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if args.config == "synthetic" else "parking.g2o (reference dataset, via tests/golden)",
        "config": {
            "workload": (f"synthetic SE3 sphere-spiral pose graph, {args.poses} poses ({args.poses // args.ring} rings x {args.ring}), " if args.config == "synthetic"
                         else f"parking.g2o SE3, {len(g['ids'])} poses, ") +
                        f"{len(g['edge_ij'])} edges, NFR Tree, " + ("Global linearisation point = stored estimates, " if args.lin_point == "global" else "Local linearisation point (10 LM iterations per blanket), ") +
                        f"globalDecimate sparsity {args.sparsity} ({len(which)} removals), marginalizeNoOptimize only",
            "parallelism": "single GPU" if world == 1 else
                           f"one replicated graph on {world} GPUs; a batch is sharded + all-gathered (RCCL) when the cost model says it pays, otherwise computed by every rank "
                           "(this workload: never - the figure is flat in N by design, see multi_gpu)",
            "exchanged_batches": stats["n_exchanged"], "exchanged_bytes": stats["exchanged_bytes"], "exchange_seconds": stats["exchange_seconds"],
            "rounds": stats["n_rounds"], "removed": removed, "max_blanket": stats["max_blanket"],
            "kld_sum": stats["kld_sum"], "host_seconds_per_step": stats["host_seconds"], "device_wait_seconds_per_step": stats["device_seconds"],
            "schedule_seconds_per_step": stats["schedule_seconds"], "commit_seconds_per_step": stats["commit_seconds"],
            "launch_seconds_per_step": stats["launch_seconds"], "cpu_affinity": numa_note,
            "batches": stats["n_batches"], "kernel_launches": stats["n_launches"], "worker": wprof,
        },
    }
    # The dominant kernel. Narrow batches (this workload: all of them after the first two) run inside ONE persistent
    # worker kernel per marginalisation (blanket_worker, csrc/spg_kernels.hip): "a launch" is one such run = one step,
    # its duration (HIP events on its stream) includes the time its workgroups wait for the host's next batch.
    traffic, traffic_note = None, None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_summary.json")))
        if world == 1 and args.config == "synthetic" and args.poses == 100000 and args.ring == 400:
            traffic = pm["traffic_bytes_per_launch"]
            traffic_note = ("NOT measured in this run: from the committed rocprofv3 --pmc passes of this same command (profiles/r03_pmc_summary.json: "
                            "FETCH_SIZE + WRITE_SIZE of the worker kernel, separate passes, per launch)")
    except Exception:
        pass
    if wprof["runs"] > 0 and wprof["kernel_ms"] > 0 and wprof["blankets"] >= 0.5 * removed * args.steps:
        per_launch_bytes = wprof["alg_bytes"] / wprof["runs"]
        per_launch_s = 1e-3 * wprof["kernel_ms"] / wprof["runs"]
        achieved = per_launch_bytes / per_launch_s / 1e9
        flops = FLOP_PER_NODE_K4 * wprof["blankets"] / (1e-3 * wprof["kernel_ms"]) / 1e12
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_note": traffic_note,
            "kernel": "blanket_worker<6,NFR> (persistent: one launch per marginalisation, fed through a queue)",
            "launches": wprof["runs"], "event_sampling": "HIP events around every run of the worker kernel in the timed region",
            "avg_launch_us": 1e6 * per_launch_s, "alg_bytes_per_launch": per_launch_bytes,
            "blankets_per_launch": wprof["blankets"] / wprof["runs"],
            "other_launches": {"launches": prof["launches"], "kernel_ms": prof["kernel_ms"], "blankets": prof["blankets"],
                               "what": "batches launched the ordinary way (first two of a call, blankets the worker does not take)"},
            "note": "the path is latency-bound, not HBM-bound: ~250 dependent rounds of ~200 blankets, one blanket = one ~45 us dependent chain "
                    "(SURVEY.md 8d: ~110 flop/B on paper); fp64 vector fraction alongside",
            "fp64_vector_tflops_est": flops, "fp64_vector_frac_est": flops / FP64_VECTOR_PEAK_TFLOPS,
        }
    elif prof["launches"] > 0 and prof["kernel_ms"] > 0:
        per_launch_bytes = prof["alg_bytes"] / prof["launches"]
        per_launch_s = 1e-3 * prof["kernel_ms"] / prof["launches"]
        achieved = per_launch_bytes / per_launch_s / 1e9
        flops = FLOP_PER_NODE_K4 * prof["blankets"] / (1e-3 * prof["kernel_ms"]) / 1e12
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None, "traffic_note": None,
            "kernel": "blanket_kernel<6,128,false,NFR>" if prof["blankets"] / prof["launches"] <= 512 else "blanket_kernel<6,64,false,NFR>",
            "launches": prof["launches"], "event_sampling": "HIP events around every launch of the timed region",
            "avg_launch_us": 1e6 * per_launch_s, "alg_bytes_per_launch": per_launch_bytes,
            "blankets_per_launch": prof["blankets"] / prof["launches"],
            "note": "path is fp64-ALU/latency-bound on paper (SURVEY.md 8d: ~110 flop/B); fp64 vector fraction alongside",
            "fp64_vector_tflops_est": flops, "fp64_vector_frac_est": flops / FP64_VECTOR_PEAK_TFLOPS,
        }
    if multi:
        out["multi_gpu"] = multi
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from tests import oracle_lib, util
        gs = g
        ws = which
        sample = "full workload, strictly sequential (the reference's execution model), 1 core"
        if args.cpu_sample_poses and args.cpu_sample_poses < args.poses:
            n = args.cpu_sample_poses
            keep = (g["edge_ij"][:, 0] < n) & (g["edge_ij"][:, 1] < n)
            gs = {"pose_dim": 6, "ids": g["ids"][:n], "poses": g["poses"][:n], "edge_ij": g["edge_ij"][keep], "edge_data": g["edge_data"][keep]}
            ws = which[which < n]
            sample = f"vertex prefix of {n} poses of the same graph ({len(ws)} removals), strictly sequential, 1 core"
        og = oracle_lib.OracleGraph.from_dict(gs)
        rc = og.marginalize(ws, opts)
        secs = og.seconds()
        bl = og.blankets()
        out["cpu_baseline"] = {"value": len(bl["root"]) / secs, "unit": "nodes/s", "cores": 1, "kind": "port",
                               "sample": sample, "seconds": secs, "rc": rc,
                               "what": "CPU restatement of the reference's Eigen/CHOLMOD path (oracle/libspg_ref.so), not the reference"}
        # (b) SURVEY.md 8d: the same conflict-free rounds (the product's host scheduler) with the oracle as
        # the arithmetic, the blankets of a round spread over the host cores this process may use
        try:
            import time as _time
            cores = max(1, len(os.sched_getaffinity(0)))   # every core this process may use
            ictx = oracle_lib.injected_context(threads=cores)
            from sparsifyposegraph_amd.graph import GraphWrapperHIP as _GW
            hg = _GW.from_dict(gs, ctx=ictx)
            t0 = _time.perf_counter()
            st_mt = hg.marginalizeNoOptimize(ws, opts)
            secs_mt = _time.perf_counter() - t0
            out["cpu_baseline"]["rounds_all_cores"] = {
                "value": st_mt["n_removed"] / secs_mt, "unit": "nodes/s", "cores": cores, "seconds": secs_mt, "rounds": st_mt["n_rounds"],
                "what": "same restatement, independent blankets of each round on std::threads (the reference itself is single-threaded per graph)"}
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out["cpu_baseline"]["rounds_all_cores"] = {"error": str(e)[:200]}
        if gs is g:
            kref = float(np.nansum(bl["kld"]))
            out["parity"] = {"kld_sum_ref": kref, "kld_sum_rel_err": abs(kref - stats["kld_sum"]) / max(abs(kref), 1e-300)}
            try:
                out["parity"]["max_edge_rel_err"] = util.compare_edge_sets(6, og.edges(), replicas[-1].edges(), rtol=1.0)
                out["parity"]["topology_identical"] = True
            except AssertionError as e:
                out["parity"]["topology_identical"] = False
                out["parity"]["error"] = str(e)[:200]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
