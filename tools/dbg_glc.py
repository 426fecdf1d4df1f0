import sys, numpy as np
sys.path.insert(0, '.')
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context, SpgError
from tests import util
ctx = Context(0)
for case in ["manhattan_glc_tree", "intel_glc_tree_10pct", "sphere_glc_tree", "manhattan_glc_dense"]:
    g, which, opts, gold_edges, gold_bl, gold_vids = util.load_golden(case)
    hg = GraphWrapperHIP.from_dict(g, ctx=ctx, useGLC=True)
    try:
        st = hg.marginalizeNoOptimize(which, opts)
        print(case, "ok", st["n_rounds"])
    except SpgError as e:
        bl = hg.blankets()
        bad = np.nonzero((bl["status"] != 0) & (bl["status"] != 6))[0]
        print(case, "FAIL", str(e)[:60], "statuses", np.bincount(bl["status"]).tolist(), "first bad", [(int(bl["root"][i]), int(bl["round"][i]), int(bl["status"][i])) for i in bad[:5]], hg.last_stats)
