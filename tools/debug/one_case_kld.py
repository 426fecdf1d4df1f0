"""Diagnostic: one (golden graph, NFR topology, chord ratio) through marginalizeNoOptimize AND the global KLD against the baseline,
each step announced before it starts (so that a crash names its step):
    python -X faulthandler tools/debug/one_case_kld.py parking_full_nfr_tree 4 1.0"""
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
ctx = Context(0)
hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
base = GraphWrapperHIP.from_dict(g, ctx=ctx)
t0 = time.time()
try:
    print("marginalize ...", flush=True)
    st = hg.marginalizeNoOptimize(which, o)
    print(case, topo, "removed", st["n_removed"], "bad", st["n_bad_status"], "max_blanket", st["max_blanket"], "%.2fs" % (time.time() - t0), flush=True)
    e = hg.edges()
    kinds, counts = np.unique(e["kind"], return_counts=True)
    lens = np.diff(e["data_off"])
    print("edges by kind", dict(zip(kinds.tolist(), counts.tolist())), "largest record", int(lens.max()), "doubles; edge data", float(lens.sum()) * 8 / 2**20, "MiB", flush=True)
    print("global KLD ...", flush=True)
    t1 = time.time()
    kld = base.kullbackLeibler(hg)
    print("global KLD", kld, "%.2fs" % (time.time() - t1), base.last_kld if hasattr(base, "last_kld") else "", flush=True)
except SpgError as ex:
    print(case, topo, "ERROR", str(ex)[:300], flush=True)
