"""Diagnostic / profile target: GLC Dense on sphere.g2o at full size (graph from tests/golden) — 25 clusters of 50 removed +
~100 kept SE3 vertices each through the dense HBM pipeline (csrc/spg_dense.hip). Prints one JSON line: device time and
fp64 rate of the n^3-class work against the fp64 matrix-core peak."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
from tests import util
ctx = Context(0)
g, which, *_ = util.load_golden("sphere_full_nfr_tree")
opts = abi.make_options(6, abi.ALG_GLC, abi.TOPO_DENSE)
out = None
for rep in range(3):
    hg = GraphWrapperHIP.from_dict(g, ctx=ctx, useGLC=True)
    ctx.profile(True)
    t0 = time.perf_counter()
    st = hg.marginalizeNoOptimize(which, opts)
    wall_ms = 1e3 * (time.perf_counter() - t0)
    out = ctx.profile_read_big()
    ctx.profile(False)
tf = 1e-9 * out["flops"] / max(out["kernel_ms"], 1e-9)
print(json.dumps({"workload": "sphere.g2o GLC Dense sparsity 2, full size", "large_blankets": out["blankets"], "largest_n_plus_nm": out["n_max"],
                  "device_ms": out["kernel_ms"], "marginalize_wall_ms": wall_ms, "n3_flops": out["flops"], "fp64_tflops": tf, "fp64_mfma_peak_tflops_datasheet": 78.6,
                  "frac_of_datasheet": tf / 78.6, "removed": st["n_removed"], "batches": st["n_batches"]}))
