"""Diagnostic / profile target: the generic NFR kernel on clusters — sphere.g2o under CliqueyDense at full size (25 blankets
of 100 kept + 50 removed SE3 vertices, each consuming the previous one's 99-measurement correlated edge), optionally
parking.g2o (clusters of 150-196 vertices, rank-deficient targets). Prints one JSON line per dataset.
Usage: python tools/cluster_bench.py [sphere] [parking]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
from tests import util
ctx = Context(0)
for name in (sys.argv[1:] or ["sphere"]):
    g, which, *_ = util.load_golden({"sphere": "sphere_full_nfr_tree", "parking": "parking_full_nfr_tree"}[name])
    o = abi.make_options(6, abi.ALG_NFR, abi.TOPO_CLIQUEY_DENSE)
    o.chord_ratio = 1.0
    base = GraphWrapperHIP.from_dict(g, ctx=ctx)
    hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
    t0 = time.perf_counter()
    st = hg.marginalizeNoOptimize(which, o)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    kld = base.kullbackLeibler(hg)
    dk = time.perf_counter() - t0
    b = hg.blankets()
    print(json.dumps({"workload": f"{name}.g2o NFR CliqueyDense, full size", "removed": int(st["n_removed"]), "bad_status": int(st["n_bad_status"]),
                      "max_blanket": int(st["max_blanket"]), "rank_deficient_blankets": int((b["info"] & 1).sum()),
                      "marginalize_s": dt, "kld_sum": float(st["kld_sum"]), "global_kld": float(kld), "global_kld_s": dk}), flush=True)
