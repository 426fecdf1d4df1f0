"""Diagnostic: 40 marginalisations of the 100k-pose workload must be bit-identical (KLD sum, edge count,
sum of all edge payloads): guards the early-publish / polling protocol and the BAR descriptor path
against races. Not part of the product or the tests."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
ctx = Context(0)
g = g2o_io.synth_sphere(100000, 400)
which = np.array([i for i in range(4, 100000) if i % 2], np.int32)
opts = abi.make_options(6)
ref = None
for it in range(40):
    hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
    st = hg.marginalizeNoOptimize(which, opts)
    key = (st["kld_sum"], st["n_new_edges"], st["n_removed"], st["n_bad_status"])
    if ref is None: ref = key
    assert key == ref, (it, key, ref)
    if it % 10 == 0:
        e = hg.edges()
        chk = float(np.sum(e["data"]))
        if it == 0: chk0 = chk
        assert chk == chk0, (it, chk, chk0)
print("stress ok", ref)
