"""Diagnostic: device-side time per blanket inside the persistent worker (SPG_WORKER_STAMP=1): staging (ticket taken ->
descriptors in LDS) and body (-> final word), from the 100 MHz wall clock, on the bench workload."""
import os, sys
os.environ["SPG_WORKER_STAMP"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
ctx = Context(0)
g = g2o_io.synth_sphere(40000, 400)
which = np.array([i for i in range(4, 40000) if i % 2], np.int32)
for rep in range(2):
    hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
    st = hg.marginalizeNoOptimize(which, abi.make_options(6))
mg = hg.blankets()["min_gap"]
stage = np.floor(mg) * 0.01
body = (mg - np.floor(mg)) * 1e6 * 0.01
ok = (mg >= 1) & (body > 1)
print(f"blankets {len(mg)} stamped {ok.sum()}  staging us: median {np.median(stage[ok]):.2f} p90 {np.percentile(stage[ok],90):.2f}  body us: median {np.median(body[ok]):.2f} p90 {np.percentile(body[ok],90):.2f}")
print(st)
