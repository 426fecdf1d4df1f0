"""Diagnostic: per-phase cycle stamps of the NFR blanket kernel in its two-wavefront form (the one the worker runs):
the common prefix, the two concurrent chains (Chow-Liu on wave 0, gauge route on wave 1), the common tail.
flags bit 16; stamps land in the target-info region. Usage: python tools/stamp_split.py [blankets]"""
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
opts = abi.make_options(6, flags=(1 << 16))
for rep in range(3):
    out = ctx.marginalize_batch(opts, batch, want_target=True)
ti, off = out["target_info"], out["target_info_off"]
S = np.array([ti[off[b]:off[b] + 36] for b in range(len(roots))])
def seg(name, a, b):
    ok = (S[:, a] > 0) & (S[:, b] > 0)
    if ok.any():
        print(f"{name:34s} {np.median(S[ok, b] - S[ok, a]):9.0f} cycles")
print("--- prefix")
seg("gather + clear", 0, 23); seg("jacobians + omega staged", 23, 24); seg("T = Omega J", 24, 25); seg("accumulate H", 25, 1)
seg("Hmm inverse", 1, 26); seg("Y", 26, 27); seg("Lambda update", 27, 28); seg("mirror (schur done)", 28, 2)
print("--- wave 0: Chow-Liu")
seg("Tikhonov inverse", 2, 5); seg("vertex chol", 5, 6); seg("pair weights", 6, 7); seg("sort", 7, 8); seg("kruskal + gap", 8, 29)
seg("CHAIN 0 total", 2, 29)
print("--- wave 1: gauge")
seg("gauge basis", 2, 11); seg("orthonormalise", 11, 12); seg("C formed", 12, 13); seg("C inverse + guard", 13, 30)
seg("CHAIN 1 total", 2, 30)
print("--- tail")
seg("join (both chains done)", 2, 9); seg("new edges", 9, 10); seg("(gauge accepted)", 10, 17); seg("  J Sigma", 17, 31); seg("  B_e", 31, 32); seg("  X_e = B_e^-1 + records", 32, 33); seg("  publish", 33, 18)
seg("closed form -> publish", 17, 18)
seg("A assembled", 18, 19); seg("A + NN", 19, 20); seg("chol A", 20, 21); seg("end", 21, 22)
seg("TOTAL to publish", 0, 18); seg("TOTAL", 0, 22)
