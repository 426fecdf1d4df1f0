import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context, SpgError
from tests import util
ctx = Context(0)
for case in ("manhattan_full_glc_tree", "sphere_full_nfr_tree", "parking_full_nfr_tree", "intel_nfr_tree_sp3"):
    g, which, opts, *_ = util.load_golden(case)
    d = opts.pose_dim
    for topo, chord in ((abi.TOPO_CLIQUEY_SUBGRAPH, 0.5), (abi.TOPO_CLIQUEY_DENSE, 1.0), (abi.TOPO_SUBGRAPH, 0.3)):
        o = abi.make_options(d, abi.ALG_NFR, topo); o.chord_ratio = chord
        hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
        base = GraphWrapperHIP.from_dict(g, ctx=ctx)
        t0 = time.time()
        try:
            st = hg.marginalizeNoOptimize(which, o)
            dt = time.time() - t0
            kld = base.kullbackLeibler(hg)
            b = hg.blankets()
            print(case, topo, "ok removed", st["n_removed"], "bad", st["n_bad_status"], "status", np.unique(b["status"], return_counts=True), "max_blanket", st["max_blanket"], "kld_sum %.4g global %.4g" % (st["kld_sum"], kld), "%.2fs" % dt, flush=True)
        except SpgError as e:
            b = hg.blankets()
            print(case, topo, "ERROR", str(e)[:160], "after", len(b["root"]), "blankets; statuses", np.unique(b["status"], return_counts=True), flush=True)
