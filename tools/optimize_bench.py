"""Diagnostic: dense device LM (spg_graph_optimize) on synthetic SE3 sphere graphs with perturbed
estimates; HIP-event time per linear solve. Not part of the product or the tests."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context

ctx = Context(0)
for n in [int(x) for x in sys.argv[1:]] or [1000, 2500]:
    g = g2o_io.synth_sphere(n, 50)
    rng = np.random.default_rng(0)
    P = np.array(g["poses"], float)
    P[1:, :3] += 0.05 * rng.standard_normal((n - 1, 3))
    g["poses"] = P
    hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
    st = hg.optimize(50)
    st["poses"] = n
    st["ms_per_solve"] = 1e3 * st["device_seconds"] / max(st["trials"], 1)
    print(json.dumps(st), flush=True)
