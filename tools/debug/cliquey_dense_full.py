import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context, SpgError
from tests import util
ctx = Context(0)
for case in sys.argv[1:] or ("sphere_full_nfr_tree", "intel_nfr_tree_sp3", "parking_full_nfr_tree"):
    g, which, opts, *_ = util.load_golden(case)
    d = opts.pose_dim
    o = abi.make_options(d, abi.ALG_NFR, abi.TOPO_CLIQUEY_DENSE)
    hg, base = GraphWrapperHIP.from_dict(g, ctx=ctx), GraphWrapperHIP.from_dict(g, ctx=ctx)
    t0 = time.time()
    try:
        st = hg.marginalizeNoOptimize(which, o)
        dt = time.time() - t0
        kld = base.kullbackLeibler(hg)
        print(case, "CliqueyDense ok: removed", st["n_removed"], "blankets", len(hg.blankets()["root"]), "max_blanket", st["max_blanket"],
              "kld_sum %.3g global KLD %.3g" % (st["kld_sum"], kld), "%.2f s" % dt, flush=True)
    except SpgError as e:
        print(case, "ERROR", str(e)[:200], flush=True)
