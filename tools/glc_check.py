"""Diagnostic: NFR Tree vs GLC Tree on the 100k-pose workload (un-timed launches, host wall clock).
Not part of the product or the tests."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
ctx = Context(0)
n = 100000
g = g2o_io.synth_sphere(n, 400)
which = np.array([i for i in range(4, n) if i % 2], np.int32)
for name, alg, topo, glc in (("NFR tree", abi.ALG_NFR, abi.TOPO_TREE, False), ("GLC tree", abi.ALG_GLC, abi.TOPO_TREE, True)):
    for rep in range(3):
        hg = GraphWrapperHIP.from_dict(g, ctx=ctx, useGLC=glc)
        hg.reserve(int(len(g["ids"]) * 7 + len(g["edge_ij"]) * 28) * 6)
        t1 = time.perf_counter()
        st = hg.marginalizeNoOptimize(which, abi.make_options(6, alg, topo))
        t2 = time.perf_counter()
    print(f"{name}: {t2-t1:.4f}s = {st['n_removed']/(t2-t1):.0f} nodes/s, batches {st['n_rounds']}, bad {st['n_bad_status']}, E {hg.numEdges()}", flush=True)
