import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
ctx = Context(0)
g = g2o_io.synth_sphere(100000, 400)
which = np.array([i for i in range(4, 100000) if i % 2], np.int32)
opts = abi.make_options(6)
reps = [GraphWrapperHIP.from_dict(g, ctx=ctx) for _ in range(int(os.environ.get("NREP", "8")))]
need = int(len(g["ids"]) * 7 + len(g["edge_ij"]) * 28) * 3
if os.environ.get("ARENA_MIB"): need = int(os.environ["ARENA_MIB"]) * 2**20 // 8
for r in reps: r.reserve(need)
prof = len(sys.argv) > 1 and sys.argv[1] == "prof"
if os.environ.get("KEEPALIVE"):
    import threading, torch
    mb = int(os.environ["KEEPALIVE"])
    x = torch.empty(mb * 2**20 // 4, dtype=torch.float32, device="cuda:0"); y = torch.empty_like(x)
    side = torch.cuda.Stream()
    stop = False
    def spin():
        with torch.cuda.stream(side):
            while not stop:
                for _ in range(4): y.copy_(x, non_blocking=True)
                side.synchronize()
    th = threading.Thread(target=spin, daemon=True); th.start()
if prof: ctx.profile(1)
small = g2o_io.synth_sphere(2000, 50)
sw = np.array([i for i in range(4, 2000) if i % 2], np.int32)
for i, r in enumerate(reps):
    if os.environ.get("INTERLEAVE"):
        for _ in range(int(os.environ["INTERLEAVE"])): GraphWrapperHIP.from_dict(small, ctx=ctx).marginalizeNoOptimize(sw, opts)
    if os.environ.get("SLEEP_MS"): time.sleep(float(os.environ["SLEEP_MS"]) * 1e-3)
    ctx.synchronize()
    import resource
    f0 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    t0 = time.perf_counter()
    st = r.marginalizeNoOptimize(which, opts)
    t1 = time.perf_counter()
    print(f"  minor page faults during the call: {resource.getrusage(resource.RUSAGE_SELF).ru_minflt - f0}", flush=True)
    ctx.synchronize()
    t2 = time.perf_counter()
    import ctypes as C
    cap = C.c_int64()
    ptr = ctx.L.spg_graph_arena(r.h, C.byref(cap))
    print(f"  arena {int(ptr or 0):#x} cap {cap.value * 8 / 2**20:.1f} MiB  mod2M {int(ptr or 0) % (2 << 20):#x}", flush=True)
    if os.environ.get("SPG_WORKER_STAMP") == "1":
        mg = r.blankets()["min_gap"]
        ready = np.floor(mg) * 0.01
        final = (mg - np.floor(mg)) * 1e6 * 0.01
        ok = (mg >= 1) & (final > 1) & (ready > 1) & (ready < 1e4)
        print(f"  device-side pick -> ready us: median {np.median(ready[ok]):.1f} p90 {np.percentile(ready[ok], 90):.1f}; -> final median {np.median(final[ok]):.1f}  ({ok.sum()} stamped)", flush=True)
    print(f"step {i}: wall {1e3*(t1-t0):.2f} ms (+sync {1e3*(t2-t1):.2f}) host {1e3*st['host_seconds']:.2f} wait {1e3*st['device_seconds']:.2f} sched {1e3*st['schedule_seconds']:.2f} commit {1e3*st['commit_seconds']:.2f} launch {1e3*st['launch_seconds']:.2f} -> unaccounted {1e3*(t1-t0-st['host_seconds']-st['device_seconds']):.2f} ms", flush=True)
