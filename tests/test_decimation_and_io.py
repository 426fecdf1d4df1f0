"""CPU: decimation.h functions against a line-by-line Python restatement of src/decimation.cpp:11-49;
native .g2o reader/writer against the Python parser."""
import math
import os

import numpy as np
import pytest

from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import (DecimateOptions, GraphWrapperHIP, clusterDecimate, globalDecimate,
                                         onlineDecimate)
from tests import oracle_lib, util


def ref_cluster(last, endvert, sparsity, cs):
    if ((last - 4) % cs == 0 and last > 4) or last == endvert:
        start = int(math.ceil((last - 5) / float(cs)) - 1) * cs + 5
        return [i for i in range(start, last + 1) if i % sparsity > 0]
    return []


def ref_online(last, endvert, sparsity, cs):
    return [] if last % sparsity == 0 else [last]


def ref_global(last, endvert, sparsity, cs):
    return [i for i in range(4, endvert + 1) if i % sparsity != 0] if last == endvert else []


@pytest.mark.parametrize("sparsity", [2, 3, 5])
@pytest.mark.parametrize("cs", [1, 4, 10])
def test_decimation_functions(sparsity, cs):
    o = DecimateOptions(sparsity, cs)
    end = 57
    for last in range(4, end + 1):
        assert globalDecimate(last, end, o) == ref_global(last, end, sparsity, cs)
        assert onlineDecimate(last, end, o) == ref_online(last, end, sparsity, cs)
        assert clusterDecimate(last, end, o) == ref_cluster(last, end, sparsity, cs)
    assert len(globalDecimate(99999, 99999, DecimateOptions(2))) == 49998  # BASELINE.json config 5


@pytest.mark.parametrize("d", [3, 6])
def test_g2o_roundtrip_native_vs_python(d, tmp_path):
    ictx = oracle_lib.injected_context()
    g = g2o_io.synth_sphere(300, 20) if d == 6 else g2o_io.synth_manhattan(300, 20)
    p = str(tmp_path / "in.g2o")
    g2o_io.write_g2o(p, g)
    gp = g2o_io.load_g2o(p)
    hg = GraphWrapperHIP.load(p, ctx=ictx)
    ids, poses = hg.vertices()
    assert np.array_equal(ids, gp["ids"]) and np.allclose(poses, gp["poses"], rtol=0, atol=1e-15)
    e = hg.edges()
    assert np.array_equal(e["vert_ids"].reshape(-1, 2), gp["edge_ij"])
    assert np.allclose(e["data"].reshape(len(gp["edge_ij"]), -1), gp["edge_data"], rtol=0, atol=1e-15)
    # marginalise (GLC leaves n-ary edges in the graph), write, and re-read the binary part
    which = np.array([i for i in range(4, 300) if i % 2], np.int32)
    hg.marginalizeNoOptimize(which, abi.make_options(d, abi.ALG_GLC if d == 3 else abi.ALG_NFR))
    out = str(tmp_path / "out.g2o")
    hg.write(out)
    txt = open(out).read().splitlines()
    assert sum(l.startswith("VERTEX") for l in txt) == hg.numVertices()
    assert sum(l.startswith(("EDGE", "GLC_EDGE")) for l in txt) == hg.numEdges()
    if d == 3:
        glc = [l for l in txt if l.startswith("GLC_EDGE")]
        assert glc and all(" || GLC_REPARAM_SE2_ISAM " in l for l in glc)
        t = glc[0].split()
        bar = t.index("||")
        q, r, n = bar - 1, int(t[bar + 2]), int(t[bar + 3])
        assert n == 3 * q and len(t) == bar + 4 + n + r * n + r * (r + 1) // 2
    # what was written loads back as the same graph, GLC_EDGE records included (GLCEdge::read, src/glc_edge.cpp:65-93)
    back = GraphWrapperHIP.load(out, ctx=ictx, useGLC=(d == 3))
    assert back.numVertices() == hg.numVertices() and back.numEdges() == hg.numEdges()
    util.compare_edge_sets(d, hg.edges(), back.edges(), rtol=1e-15)
    i1, p1 = hg.vertices()
    i2, p2 = back.vertices()
    assert np.array_equal(i1, i2) and np.allclose(p1, p2, rtol=0, atol=1e-15)


def test_glc_edge_with_general_information_is_folded_into_w(tmp_path):
    """A GLC_EDGE whose information is not the identity (the reference always writes I_r) loads as the
    equivalent edge W' = L^T W with Omega = L L^T, i.e. the same W^T Omega W."""
    ictx = oracle_lib.injected_context()
    rng = np.random.default_rng(2)
    W = rng.standard_normal((2, 6))
    A = rng.standard_normal((2, 2))
    Om = A @ A.T + np.eye(2)
    meas = rng.standard_normal(6)
    path = str(tmp_path / "g.g2o")
    with open(path, "w") as f:
        f.write("VERTEX_SE2 0 0 0 0\nVERTEX_SE2 1 1 0 0.1\nVERTEX_SE2 2 2 0.5 0.2\n")
        f.write("EDGE_SE2 0 1 1 0 0.1 10 0 0 10 0 5\n")
        nums = list(meas) + list(W.reshape(-1)) + [Om[0, 0], Om[0, 1], Om[1, 1]]
        f.write("GLC_EDGE 1 2 || GLC_REPARAM_SE2_ISAM 2 6 " + " ".join(repr(float(x)) for x in nums) + "\n")
    hg = GraphWrapperHIP.load(path, ctx=ictx, useGLC=True)
    e = hg.edges()
    k = int(np.nonzero(e["kind"] == abi.EDGE_GLC)[0][0])
    rec = e["data"][e["data_off"][k]:e["data_off"][k + 1]]
    assert np.allclose(rec[:6], meas, rtol=0, atol=1e-15)
    W2 = rec[6:].reshape(2, 6)
    assert np.allclose(W2.T @ W2, W.T @ Om @ W, rtol=1e-13, atol=1e-13)


def test_reference_dataset_loads_if_present():
    path = "/root/reference/datasets/sphere.g2o"
    if not os.path.exists(path):
        pytest.skip("reference datasets exist only in the build container")
    g = g2o_io.load_g2o(path)
    assert len(g["ids"]) == 2500 and len(g["edge_ij"]) == 4948 and g["pose_dim"] == 6
    hg = GraphWrapperHIP.load(path, ctx=oracle_lib.injected_context())
    assert hg.numVertices() == 2500 and hg.numEdges() == 4948
