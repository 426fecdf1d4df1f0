"""Diagnostic: the marginalisation at 10x the headline size (1M poses, rings of 400 and of 4000) — the rate
must hold and no blanket may fail. Not part of the product or the tests."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
ctx = Context(0)
for n, ring in ((1000000, 400), (1000000, 4000)):
    t0 = time.time()
    g = g2o_io.synth_sphere(n, ring)
    which = np.array([i for i in range(4, n) if i % 2], np.int32)
    hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
    t1 = time.time()
    st = hg.marginalizeNoOptimize(which, abi.make_options(6))
    t2 = time.time()
    print(f"poses {n} ring {ring}: build {t1-t0:.1f}s, marginalize {t2-t1:.3f}s = {st['n_removed']/(t2-t1):.0f} nodes/s, batches {st['n_rounds']}, bad {st['n_bad_status']}, kld_sum {st['kld_sum']:.6g}, V {hg.numVertices()} E {hg.numEdges()}", flush=True)
