"""Diagnostic: device-side time per blanket inside the persistent worker (SPG_WORKER_STAMP=1): ticket taken -> ready word and -> final word, from the 100 MHz wall clock, on the bench workload."""
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
ready = np.floor(mg) * 0.01
final = (mg - np.floor(mg)) * 1e6 * 0.01
ok = (mg >= 1) & (final > 1) & (ready > 1) & (ready < 1e4)
print(f"blankets {len(mg)} stamped {ok.sum()}  pick -> ready us: median {np.median(ready[ok]):.2f} p90 {np.percentile(ready[ok],90):.2f}  pick -> final us: median {np.median(final[ok]):.2f} p90 {np.percentile(final[ok],90):.2f}")
print(st)
