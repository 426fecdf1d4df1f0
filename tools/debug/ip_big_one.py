"""Diagnostic: the hub of an SE3 hub graph with k neighbours under NFR Dense on the device — one interior-point blanket of
36 k (k - 1) / 2 variables. Usage: python tools/debug/ip_big_one.py [k=12]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsifyposegraph_amd import abi
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from sparsifyposegraph_amd.lib import Context
from tests.test_big_blankets import _star_graph
k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
g = _star_graph(k, seed=5)
ctx = Context(0)
hg = GraphWrapperHIP.from_dict(g, ctx=ctx)
t0 = time.perf_counter()
st = hg.marginalizeNoOptimize(np.array([0], np.int32), abi.make_options(6, abi.ALG_NFR, abi.TOPO_DENSE))
dt = time.perf_counter() - t0
b = hg.blankets()
print(f"k={k}: {36 * k * (k - 1) // 2} variables, status {b['status']}, info flags {b['info'] & 255}, Newton steps {b['info'] >> 8}, KLD {b['kld']}, {dt:.2f} s, edges {hg.numEdges()}")
