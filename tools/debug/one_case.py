"""Diagnostic: one (golden graph, NFR topology, chord ratio) through marginalizeNoOptimize with statuses printed:
    python -X faulthandler tools/debug/one_case.py parking_full_nfr_tree 4 1.0"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context, SpgError
from tests import util

case, topo, chord = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
g, which, opts, *_ = util.load_golden(case)
o = abi.make_options(opts.pose_dim, abi.ALG_NFR, topo)
o.chord_ratio = chord
hg = GraphWrapperHIP.from_dict(g, ctx=Context(0))
t0 = time.time()
try:
    st = hg.marginalizeNoOptimize(which, o)
    b = hg.blankets()
    print(case, topo, "ok removed", st["n_removed"], "bad", st["n_bad_status"], "max_blanket", st["max_blanket"], "rounds", st["n_rounds"],
          "status", np.unique(b["status"], return_counts=True), "%.2fs" % (time.time() - t0), flush=True)
except SpgError as e:
    b = hg.blankets()
    print(case, topo, "ERROR", str(e)[:300], "after", len(b["root"]), "blankets", flush=True)
