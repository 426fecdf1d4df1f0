"""Diagnostic: one NFR blanket of k kept SE3 vertices (hub of a hub graph) through the blanket kernel (Tree) and through the
generic kernel's closed form (CliqueySubgraph with chord ratio 0 = the same tree, one measurement per group). Wall time per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
from tests.test_big_blankets import _star_graph
ctx = Context(0)
for k in [int(a) for a in sys.argv[1:]] or [22]:
    g = _star_graph(k, seed=5)
    for name, topo, chord in (("Tree (blanket kernel)", abi.TOPO_TREE, 0.0), ("CliqueySubgraph(0) (generic kernel)", abi.TOPO_CLIQUEY_SUBGRAPH, 0.0)):
        o = abi.make_options(6, abi.ALG_NFR, topo); o.chord_ratio = chord
        ts = []
        for rep in range(4):
            hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
            t0 = time.perf_counter()
            st = hg.marginalizeNoOptimize(np.array([0], np.int32), o)
            ts.append(time.perf_counter() - t0)
        b = hg.blankets()
        print(f"k={k} {name}: {1e3 * min(ts):.2f} ms, status {b['status']}, KLD {b['kld']}, edges {hg.numEdges()}")
