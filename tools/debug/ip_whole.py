import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
from tests import oracle_lib, util
ctx = Context(0)
case = sys.argv[2] if len(sys.argv) > 2 else "manhattan_nfr_tree"
g, which, opts, *_ = util.load_golden(case)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
sub, w = util.prefix_graph(g, which, n)
o = abi.make_options(opts.pose_dim, abi.ALG_NFR, int(sys.argv[3]) if len(sys.argv) > 3 else abi.TOPO_DENSE)
o.chord_ratio = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
print("==", case, n, "topology", o.topology, "chord", o.chord_ratio)
hg = GraphWrapperHIP.from_dict(sub, ctx=ctx)
st = hg.marginalizeNoOptimize(w, o)
og = oracle_lib.OracleGraph.from_dict(sub)
assert og.marginalize(w, o) == 0
hb, ob = hg.blankets(), og.blankets()
ho = {int(r): i for i, r in enumerate(hb["root"])}
bad = []
for j, r in enumerate(ob["root"]):
    i = ho[int(r)]
    dk = abs(hb["kld"][i] - ob["kld"][j])
    if not (dk <= 1e-9) or hb["status"][i] != ob["status"][j] or (hb["info"][i] >> 8) != (ob["info"][j] >> 8):
        bad.append((j, int(r), int(hb["round"][i]), hb["kld"][i], ob["kld"][j], int(hb["info"][i] >> 8), int(ob["info"][j] >> 8), int(ob["k"][j]), int(hb["status"][i]), int(ob["status"][j])))
print("blankets", len(ob["root"]), "mismatching", len(bad), "max k", int(ob["k"].max()), "wall", st.get("seconds"))
for b in bad[:15]:
    print("seq %d root %d round %d kld dev %.12g oracle %.12g steps %d/%d k=%d status %d/%d" % b)
ca, cb = util.canonical(og.edges()), util.canonical(hg.edges())
cnt = 0
for (k, ids, xa), (_, _, xb) in zip(ca, cb):
    ps_ = abi.pose_stride(opts.pose_dim)
    e = max(util.rel_err(xa[:ps_], xb[:ps_]), util.rel_err(xa[ps_:], xb[ps_:]))
    if e > 1e-6:
        cnt += 1
        if cnt < 2: print("edge", ids, "err", e)
print("edges", len(ca), "mismatching", cnt)
