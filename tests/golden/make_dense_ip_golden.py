"""Generates tests/golden/digest_hub_dense_ip.npz: the sequential oracle's result for the removal of the hub of an SE3 hub graph
with 12 neighbours under NFR Dense — one blanket whose interior point has 66 new edges = 2 376 variables, a Newton system
beyond the sizes the LDS-resident factorisations take (path C of csrc/spg_nfr_ip.hip). The oracle needs many minutes for it
(a 2 376^2 Cholesky per Newton step in plain C), which is why the expected output is a committed fixture. Inputs come from
tests/test_big_blankets.py::_star_graph (seeded).
    python tests/golden/make_dense_ip_golden.py"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsifyposegraph_amd import abi
from tests import oracle_lib

spec = importlib.util.spec_from_file_location("tb", os.path.join(ROOT, "tests", "test_big_blankets.py"))
tb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tb)
g = tb._star_graph(12, seed=5)
og = oracle_lib.OracleGraph.from_dict(g)
t0 = time.time()
assert og.marginalize(np.array([0], np.int32), abi.make_options(6, abi.ALG_NFR, abi.TOPO_DENSE)) == 0
e, b = og.edges(), og.blankets()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "digest_hub_dense_ip.npz"), kind=e["kind"], vert_off=e["vert_off"], vert_ids=e["vert_ids"],
                    data_off=e["data_off"], data=e["data"], kld=b["kld"], status=b["status"], info=b["info"])
print("hub12 dense: kld", b["kld"], "status", b["status"], "Newton steps", b["info"] >> 8, "edges", len(e["kind"]), f"{time.time() - t0:.0f} s")
