import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context, SpgError
from tests import oracle_lib, util
ctx = Context(0)
case, topo, chord = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
g, which, opts, *_ = util.load_golden(case)
d = opts.pose_dim
o = abi.make_options(d, abi.ALG_NFR, topo); o.chord_ratio = chord
hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
try:
    st = hg.marginalizeNoOptimize(which, o)
except SpgError as e:
    print("device:", str(e)[:120])
og = oracle_lib.OracleGraph.from_dict(g)
rc = og.marginalize(which, o)
hb, ob = hg.blankets(), og.blankets()
at = {int(r): i for i, r in enumerate(hb["root"])}
idx = np.array([at[int(r)] for r in ob["root"]])
print(case, topo, chord, "oracle rc", rc, "blankets", len(ob["root"]), "status oracle", np.unique(ob["status"], return_counts=True), "device", np.unique(hb["status"], return_counts=True))
print("rank-deficient flags: oracle", int((ob["info"] & 1).sum()), "device", int((hb["info"][idx] & 1).sum()), "status equal", bool(np.array_equal(ob["status"], hb["status"][idx])))
fin = np.isfinite(ob["kld"]) & np.isfinite(hb["kld"][idx])
print("max |kld diff|", np.abs(ob["kld"][fin] - hb["kld"][idx][fin]).max(), "on", int(fin.sum()))
rd = (ob["info"] & 1) != 0
if rd.any(): print("rank-deficient blankets kld oracle", ob["kld"][rd][:6], "device", hb["kld"][idx][rd][:6])
try:
    print("worst edge rel err", util.compare_edge_sets(d, og.edges(), hg.edges(), rtol=1e-6))
except AssertionError as e:
    print("edge compare:", str(e)[:100])
