"""Diagnostic: per-phase cycle stamps (s_memtime, 100 MHz ticks -> us) of the NFR blanket kernel on
B first-round blankets of the synthetic SE3 graph. flags bit 16; stamps land in the target-info region."""
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
opts = abi.make_options(6, flags=(1 << 16) | (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
names = ["start", "assemble", "schur", "cl chol", "cl triinv", "cl gram", "cl vertex chol", "cl pair weights", "cl sort", "cl kruskal",
         "new edges", "gauge basis", "orthonorm", "C formed", "C chol", "C logdet", "C triinv", "Cinv+guard", "closed form",
         "A assembled", "A+NN", "chol A", "end", "  gather+clear", "  jac+omega", "  T=OmJ", "  Hmm inv", "  Y", "  Lambda upd"]
order = [0, 23, 24, 25, 1, 26, 27, 28] + list(range(2, 23))
for rep in range(3):
    out = ctx.marginalize_batch(opts, batch, want_target=True)
ti = out["target_info"]
off = out["target_info_off"]
S = np.array([ti[off[b]:off[b] + len(names)] for b in range(len(roots))])
# s_memtime counts shader cycles; stamps a path does not execute stay 0 and are skipped
last = S[:, 0].copy()
tot = np.zeros(len(S))
for i in order[1:]:
    n_ = names[i]
    cur = S[:, i]
    ok = cur > 0
    if not ok.any():
        continue
    dcy = np.where(ok, cur - last, 0.0)
    print(f"{n_:18s} {np.median(dcy[ok]):10.0f} cycles")
    last = np.where(ok, cur, last)
print(f"{'total':18s} {np.median(S[:, 22] - S[:, 0]):10.0f} cycles")
