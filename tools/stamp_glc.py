"""Diagnostic: per-phase cycle stamps of the GLC Tree blanket kernel (one wavefront per blanket) on first-round blankets of
the synthetic SE3 graph. flags bit 16; stamps land in the target-info region."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.lib import Context
from tests import util
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = g2o_io.synth_sphere(40000, 400)
which = np.array([i for i in range(405, 40000 - 405) if i % 2], np.int32)
batch, roots = util.first_round_batch(g, which, None, limit=B)
ctx = Context(0)
opts = abi.make_options(6, abi.ALG_GLC, abi.TOPO_TREE, flags=(1 << 16))
for rep in range(3):
    out = ctx.marginalize_batch(opts, batch, want_target=True)
ti, off = out["target_info"], out["target_info_off"]
S = np.array([ti[off[b]:off[b] + 50] for b in range(len(roots))])
def seg(name, a, b):
    ok = (S[:, a] > 0) & (S[:, b] > 0)
    if ok.any():
        print(f"{name:40s} {np.median(S[ok, b] - S[ok, a]):9.0f} cycles")
seg("prefix: gather .. Schur", 0, 40)
seg("Chow-Liu tree", 40, 41)
seg("root edge (marginal + getEdge)", 41, 42)
seg("joint marginals of the tree pairs", 42, 43)
seg("conditional targets (pinv, products)", 43, 44)
seg("getEdge x (k-1): reparam + J", 44, 45)
seg("getEdge: J inverse (batched GJ)", 45, 46)
seg("getEdge: M = invJ^T T invJ", 46, 47)
seg("getEdge: eig (batched Jacobi)", 47, 48)
seg("getEdge: cut, records", 48, 49)
seg("TOTAL", 0, 49)
