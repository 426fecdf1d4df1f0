"""Generates the committed fixtures under tests/golden/ (run in the build container, where
/root/reference exists):

  <name>.npz  = input pose graph (a vertex-prefix of a reference dataset, plain data) + the removal
                list + options + the CPU oracle's outputs on it (final edge set, per-blanket log).

The inputs are data files of the reference (datasets/*.g2o); the expected
outputs come from oracle/libspg_ref.so, NOT from the reference (which cannot be built here), so
they pin the oracle against regressions and carry the parity bar to the GPU box, where
/root/reference does not exist.  Usage: python tests/golden/make_golden.py [case ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsifyposegraph_amd import abi, g2o_io  # noqa: E402
from tests.oracle_lib import OracleGraph, canonical_edges  # noqa: E402

DATASETS = "/root/reference/datasets"
CASES = [
    # name, file, max vertex id kept, algorithm, topology, removal rule
    ("sphere_nfr_tree", "sphere.g2o", 699, abi.ALG_NFR, abi.TOPO_TREE, ("sparsity", 2)),
    ("parking_nfr_tree", "parking.g2o", 599, abi.ALG_NFR, abi.TOPO_TREE, ("sparsity", 2)),
    ("manhattan_nfr_tree", "manhattan.g2o", 1199, abi.ALG_NFR, abi.TOPO_TREE, ("sparsity", 2)),
    ("manhattan_glc_tree", "manhattan.g2o", 1199, abi.ALG_GLC, abi.TOPO_TREE, ("sparsity", 2)),
    ("manhattan_glc_dense", "manhattan.g2o", 1199, abi.ALG_GLC, abi.TOPO_DENSE, ("sparsity", 2)),
    ("intel_glc_tree_10pct", "intel.g2o", 942, abi.ALG_GLC, abi.TOPO_TREE, ("mod10", 5)),
    ("intel_nfr_tree_sp3", "intel.g2o", 942, abi.ALG_NFR, abi.TOPO_TREE, ("sparsity", 3)),
    ("sphere_glc_tree", "sphere.g2o", 399, abi.ALG_GLC, abi.TOPO_TREE, ("sparsity", 2)),
    # BASELINE.json config 2 at full size: all 3500 vertices, Dense clustering (blankets up to k+m = 33)
    ("manhattan_full_glc_dense", "manhattan.g2o", 3499, abi.ALG_GLC, abi.TOPO_DENSE, ("sparsity", 2)),
    ("manhattan_full_glc_tree", "manhattan.g2o", 3499, abi.ALG_GLC, abi.TOPO_TREE, ("sparsity", 2)),
    # configs 3 and 4 at full size
    ("sphere_full_nfr_tree", "sphere.g2o", 2499, abi.ALG_NFR, abi.TOPO_TREE, ("sparsity", 2)),
    ("parking_full_nfr_tree", "parking.g2o", 1660, abi.ALG_NFR, abi.TOPO_TREE, ("sparsity", 2)),
    # correlated NFR patterns (MultiEdgeCorrelated edges, SPG_EDGE_MULTI): one fully correlated edge per (clustered) blanket,
    # and the partially correlated cliques of fillCliques at chord ratio 1
    ("manhattan_cliquey_dense", "manhattan.g2o", 499, abi.ALG_NFR, abi.TOPO_CLIQUEY_DENSE, ("sparsity", 2)),
    ("sphere_cliquey_subgraph", "sphere.g2o", 399, abi.ALG_NFR, abi.TOPO_CLIQUEY_SUBGRAPH, ("sparsity", 2)),
]


def prefix(g, maxid):
    keep = g["ids"] <= maxid
    ek = (g["edge_ij"][:, 0] <= maxid) & (g["edge_ij"][:, 1] <= maxid)
    return {"pose_dim": g["pose_dim"], "ids": g["ids"][keep], "poses": g["poses"][keep],
            "edge_ij": g["edge_ij"][ek], "edge_data": g["edge_data"][ek]}


def removal_list(rule, last):
    if rule[0] == "sparsity":
        return np.array([i for i in range(4, last + 1) if i % rule[1] != 0], np.int32)
    return np.array([i for i in range(4, last + 1) if i % 10 == rule[1]], np.int32)


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = set(sys.argv[1:])   # optional: regenerate just the named cases
    for name, fname, maxid, alg, topo, rule in CASES:
        if only and name not in only:
            continue
        g = prefix(g2o_io.load_g2o(os.path.join(DATASETS, fname)), maxid)
        which = removal_list(rule, int(g["ids"][-1]))
        opts = abi.make_options(g["pose_dim"], alg, topo)
        og = OracleGraph.from_dict(g)
        rc = og.marginalize(which, opts)
        assert rc == 0, (name, rc)
        e = og.edges()
        b = og.blankets()
        ids, _ = og.vertices()
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            pose_dim=g["pose_dim"], ids=g["ids"], poses=g["poses"], edge_ij=g["edge_ij"], edge_data=g["edge_data"],
            which=which, algorithm=alg, topology=topo,
            out_vertex_ids=ids, out_kind=e["kind"], out_vert_off=e["vert_off"], out_vert_ids=e["vert_ids"],
            out_data_off=e["data_off"], out_data=e["data"],
            bl_root=b["root"], bl_status=b["status"], bl_info=b["info"], bl_kld=b["kld"], bl_min_gap=b["min_gap"], bl_k=b["k"])
        print(name, "V", len(g["ids"]), "E", len(g["edge_ij"]), "removed", len(b["root"]), "out edges", len(e["kind"]),
              "kld", float(np.nansum(b["kld"])), "min gap", float(b["min_gap"].min()))


if __name__ == "__main__":
    main()
