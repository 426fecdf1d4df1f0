"""Debug helper: run one golden case on the device and print the non-OK blanket statuses."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import util
from sparsifyposegraph_amd import abi, lib
from sparsifyposegraph_amd.graph import GraphWrapperHIP

case = sys.argv[1] if len(sys.argv) > 1 else "intel_nfr_tree_sp3"
ctx = lib.Context(0)
g, which, opts, gold_edges, gold_bl, gold_vids = util.load_golden(case)
hg = GraphWrapperHIP.from_dict(g, ctx=ctx, useGLC=False)
hopts = abi.make_options(opts.pose_dim, opts.algorithm, opts.topology, opts.lin_point, 0)
try:
    st = hg.marginalizeNoOptimize(which, hopts)
    print("ok", st)
except Exception as e:
    print("ERR", e)
    print(hg.last_stats)
bl = hg.blankets()
bad = np.nonzero(bl["status"] != 0)[0]
print("n blankets", len(bl["root"]), "bad", len(bad))
for i in bad[:20]:
    print({k: bl[k][i] for k in bl})
