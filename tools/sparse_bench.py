"""Diagnostic: the block-sparse multifrontal path at the size of the headline workload (SURVEY.md §8d config 5:
100 000 SE3 poses, 250 rings x 400): GraphWrapperG2O::marginalize = marginalizeNoOptimize + optimize()
(src/graph_wrapper_g2o.cpp:455-463), optimize() of the perturbed baseline, and baseline.kullbackLeibler(sparsified)
(:531-548) — the two calls the dense formulation cannot reach. Prints one JSON line per step. Not part of the product
or the tests."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ring = int(sys.argv[2]) if len(sys.argv) > 2 else 400
what = sys.argv[3] if len(sys.argv) > 3 else "all"
ctx = Context(0)
g = g2o_io.synth_sphere(n, ring)
which = np.array([i for i in range(4, n) if i % 2], np.int32)


def line(name, wall, st):
    keep = {k: st[k] for k in st if k in ("iterations", "trials", "chi2_initial", "chi2_final", "n", "n_marginalized", "device_seconds", "supernodes",
                                          "front_bytes", "factor_flops", "kld", "innerprod", "mahalanobis", "logdetx", "logdety", "solver")}
    keep.update(step=name, poses=n, wall_s=round(wall, 3))
    if st.get("factor_flops") and st.get("device_seconds"):
        per = st.get("trials", 1) or 1
        keep["tflops_factor_only_lower_bound"] = st["factor_flops"] * per / st["device_seconds"] / 1e12
    print(json.dumps(keep), flush=True)


if what in ("all", "optimize"):
    # optimize() of the baseline from perturbed estimates (what the reference does at load, src/evaluate.cpp:404-405)
    gp = dict(g)
    rng = np.random.default_rng(0)
    P = np.array(g["poses"], float)
    P[1:, :3] += 0.05 * rng.standard_normal((n - 1, 3))
    gp["poses"] = P
    hg = GraphWrapperHIP.from_dict(gp, ctx=ctx)
    t0 = time.time()
    st = hg.optimize(50)
    line("optimize(baseline, perturbed)", time.time() - t0, st)
    del hg

base = GraphWrapperHIP.from_dict(g, ctx=ctx)
sp = GraphWrapperHIP.from_dict(g, ctx=ctx)
t0 = time.time()
sp.marginalizeNoOptimize(which, abi.make_options(6))
t1 = time.time()
if what in ("all", "marginalize"):
    st = sp.optimize(50)
    line("marginalize = marginalizeNoOptimize (%.3f s) + optimize" % (t1 - t0), time.time() - t0, st)
if what in ("all", "kld"):
    t0 = time.time()
    kld = base.kullbackLeibler(sp)
    line("kullbackLeibler(baseline, sparsified)", time.time() - t0, base.last_kld_terms)
    t0 = time.time()
    kld = base.kullbackLeibler(sp)
    line("kullbackLeibler again (warm)", time.time() - t0, base.last_kld_terms)
