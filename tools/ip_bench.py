"""Diagnostic: interior-point NFR (csrc/spg_nfr_ip.hip) on BASELINE config 3's graph (sphere.g2o at full size, 1 248
removals) with the Subgraph pattern (chord ratio 0.5): wall time of marginalizeNoOptimize, blankets through the
interior point, Newton steps. Not part of the product or the tests."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
from tests import util

ctx = Context(0)
g, which, opts, *_ = util.load_golden("sphere_full_nfr_tree")
for topo, chord in ((abi.TOPO_TREE, 1.0), (abi.TOPO_SUBGRAPH, 0.5)):
    o = abi.make_options(6, abi.ALG_NFR, topo)
    o.chord_ratio = chord
    hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
    t0 = time.time()
    st = hg.marginalizeNoOptimize(which, o)
    wall = time.time() - t0
    b = hg.blankets()
    steps = b["info"] >> 8
    print(json.dumps({"topology": "Tree" if topo == abi.TOPO_TREE else f"Subgraph({chord})", "removed": int(st["n_removed"]), "rounds": int(st["n_rounds"]),
                      "bad_status": int(st["n_bad_status"]), "kld_sum": st["kld_sum"], "wall_s": round(wall, 4),
                      "interior_point_blankets": int((steps > 0).sum()), "newton_steps_total": int(steps.sum()),
                      "ms_per_ip_blanket_wall": round(1e3 * wall / max(int((steps > 0).sum()), 1), 3)}), flush=True)
