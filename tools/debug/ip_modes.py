"""Diagnostic: dump the records the interior-point path writes for one golden graph, so that two runs with different
factorisation modes (SPG_IP_UNTILED=1, SPG_IP_NO_LDS_HESSIAN=1, default) can be compared bit for bit:
    python tools/debug/ip_modes.py out_a.npz ; SPG_IP_UNTILED=1 python tools/debug/ip_modes.py out_b.npz
    python tools/debug/ip_modes.py --cmp out_a.npz out_b.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
        if not same:
            bad += 1
            d = np.abs(a[k].astype(float) - b[k].astype(float)).max() if a[k].shape == b[k].shape else float("nan")
            print(f"{k}: differs (max abs diff {d:.3g})")
    print("bit-identical" if bad == 0 else f"{bad} arrays differ")
    sys.exit(1 if bad else 0)

from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
from tests import util

name = sys.argv[2] if len(sys.argv) > 2 else "sphere_nfr_tree"
g, which, opts, *_ = util.load_golden(name)
o = abi.make_options(g["pose_dim"], abi.ALG_NFR, abi.TOPO_SUBGRAPH)
o.chord_ratio = 0.5
hg = GraphWrapperHIP.from_dict(g, ctx=Context(0))
st = hg.marginalizeNoOptimize(which, o)
b = hg.blankets()
e = hg.edges()
out = {"info": b["info"], "kld": b["kld"]}
for k, v in e.items():
    if isinstance(v, np.ndarray): out["e_" + k] = v
np.savez(sys.argv[1], **out)
print("steps", int((b["info"] >> 8).sum()), "removed", int(st["n_removed"]), "bad", int(st["n_bad_status"]))
