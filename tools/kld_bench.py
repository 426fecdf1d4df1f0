"""Diagnostic: dense global KLD (spg_graph_kullback_leibler) on synthetic SE3 sphere graphs of
growing size, NFR Tree sparsification at sparsity 2; prints HIP-event time and the fp64 rate of the
O(n^3) part (Cholesky of the [marginalised | kept] baseline, Cholesky of the sparsified graph,
triangular solve). Not part of the product or the tests."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context

sizes = [int(x) for x in sys.argv[1:]] or [1000, 2500, 5000]
ctx = Context(0)
for n in sizes:
    g = g2o_io.synth_sphere(n, 50)
    which = np.array([i for i in range(4, n) if i % 2], np.int32)
    base = GraphWrapperHIP.from_dict(g, ctx=ctx)
    sp = GraphWrapperHIP.from_dict(g, ctx=ctx)
    sp.marginalizeNoOptimize(which, abi.make_options(6, abi.ALG_NFR, abi.TOPO_TREE))
    base.kullbackLeibler(sp)  # warm-up (allocations, code load)
    t0 = time.time()
    kld = base.kullbackLeibler(sp)
    wall = time.time() - t0
    t = base.last_kld_terms
    N = (t["n_marginalized"] + 63) // 64 * 64 + (t["n"] + 63) // 64 * 64
    Ng = (t["n"] + 63) // 64 * 64
    flops = N ** 3 / 3 + Ng ** 3 / 3 + Ng ** 3 / 3
    print(json.dumps({"poses": n, "N": int(N), "Ng": int(Ng), "kld": kld, "device_ms": 1e3 * t["device_seconds"], "wall_ms": 1e3 * wall,
                      "fp64_tflops": flops / t["device_seconds"] / 1e12, "frac_of_78.6": flops / t["device_seconds"] / 78.6e12}), flush=True)
