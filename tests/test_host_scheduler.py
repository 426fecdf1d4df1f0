"""CPU: the product's HOST logic (blanket extraction, conflict-free round scheduler, arena / round
protocol, graph update) driven through the C ABI with the oracle injected as the compute backend,
checked against the oracle's strictly sequential loop. No GPU, no product arithmetic involved."""
import numpy as np
import pytest

from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from tests import oracle_lib, util


@pytest.fixture(scope="module")
def ictx():
    return oracle_lib.injected_context()


@pytest.fixture(scope="module")
def pctx():
    """injected backend with launch slots and mailboxes: the library takes its pipelined driver"""
    return oracle_lib.injected_context(slots=True)


@pytest.mark.parametrize("case", util.golden_cases())
def test_pipelined_driver_equals_sequential(case, pctx):
    """Several batches in flight (scheduling against uncommitted batches, batch splitting over launch
    slots, commit by polling the mailbox ready words, late KLD harvest) == the sequential loop."""
    g, which, opts, gold_edges, gold_bl, gold_vids = util.load_golden(case)
    hg = GraphWrapperHIP.from_dict(g, ctx=pctx, useGLC=bool(opts.algorithm))
    st = hg.marginalizeNoOptimize(which, opts)
    assert st["n_bad_status"] == 0
    ids, _ = hg.vertices()
    assert np.array_equal(ids, gold_vids)
    util.compare_edge_sets(g["pose_dim"], gold_edges, hg.edges(), rtol=1e-11)
    bl = hg.blankets()
    order, gorder = np.argsort(bl["root"], kind="stable"), np.argsort(gold_bl["root"], kind="stable")
    assert np.array_equal(bl["root"][order], gold_bl["root"][gorder])
    assert np.array_equal(bl["status"][order], gold_bl["status"][gorder])
    k1, k2 = bl["kld"][order], gold_bl["kld"][gorder]
    fin = np.isfinite(k2)
    assert np.array_equal(fin, np.isfinite(k1))
    if fin.any():
        assert np.max(np.abs(k1[fin] - k2[fin]) / np.maximum(np.abs(k2[fin]), 1.0)) <= 1e-11
        assert abs(st["kld_sum"] - np.sum(k2[fin])) <= 1e-9 * max(1.0, abs(np.sum(k2[fin])))


def test_pipelined_driver_other_orders(pctx):
    g = g2o_io.synth_sphere(n_poses=3000, ring=60)
    rng = np.random.default_rng(11)
    for which in (np.array([i for i in range(4, 3000) if i % 2]), rng.permutation(np.arange(4, 3000))[:1400],
                  np.array([i for i in range(4, 3000) if i % 3])):
        which = which.astype(np.int32)
        opts = abi.make_options(6)
        og = oracle_lib.OracleGraph.from_dict(g)
        assert og.marginalize(which, opts) == 0
        hg = GraphWrapperHIP.from_dict(g, ctx=pctx)
        st = hg.marginalizeNoOptimize(which, opts)
        assert st["n_removed"] == len(which) and st["n_bad_status"] == 0
        util.compare_edge_sets(6, og.edges(), hg.edges(), rtol=1e-11)


@pytest.mark.parametrize("case", util.golden_cases())
def test_rounds_equal_sequential(case, ictx):
    g, which, opts, gold_edges, gold_bl, gold_vids = util.load_golden(case)
    hg = GraphWrapperHIP.from_dict(g, ctx=ictx, useGLC=bool(opts.algorithm))
    st = hg.marginalizeNoOptimize(which, opts)
    assert st["n_bad_status"] == 0
    assert st["n_removed"] == int(np.sum(gold_bl["k"] >= 0)) or opts.topology in (abi.TOPO_DENSE, abi.TOPO_CLIQUEY_DENSE)   # (clustered blankets remove several vertices each)
    ids, _ = hg.vertices()
    assert np.array_equal(ids, gold_vids)
    util.compare_edge_sets(g["pose_dim"], gold_edges, hg.edges(), rtol=1e-11)
    assert st["n_rounds"] < len(gold_bl["root"])  # it did batch


def test_rounds_equal_sequential_other_orders(ictx):
    """Removal lists that are not ascending / not sparsity patterns, and sparsity > 2 (chains of
    adjacent removed vertices)."""
    g = g2o_io.synth_sphere(n_poses=1500, ring=50)
    rng = np.random.default_rng(4)
    for which in (rng.permutation(np.arange(4, 1500))[:700], np.array([i for i in range(4, 1500) if i % 4]),
                  np.arange(1499, 3, -1)[:900]):
        which = which.astype(np.int32)
        opts = abi.make_options(6)
        og = oracle_lib.OracleGraph.from_dict(g)
        assert og.marginalize(which, opts) == 0
        hg = GraphWrapperHIP.from_dict(g, ctx=ictx)
        st = hg.marginalizeNoOptimize(which, opts)
        assert st["n_removed"] == len(which)
        util.compare_edge_sets(6, og.edges(), hg.edges(), rtol=1e-11)


def test_invalid_arguments(ictx):
    g = g2o_io.synth_sphere(n_poses=100, ring=10)
    hg = GraphWrapperHIP.from_dict(g, ctx=ictx)
    from sparsifyposegraph_amd.lib import SpgError
    with pytest.raises(SpgError):
        hg.marginalizeNoOptimize(np.array([5000], np.int32), abi.make_options(6))  # "vertex needs to exist"
    with pytest.raises(SpgError):
        hg.marginalizeNoOptimize(np.array([5], np.int32), abi.make_options(3))     # pose_dim mismatch
    with pytest.raises(SpgError):
        hg.addEdge(1, 9999, np.zeros(7), np.eye(6))
    st = hg.marginalizeNoOptimize(np.zeros(0, np.int32), abi.make_options(6))      # empty list is a no-op
    assert st["n_removed"] == 0 and st["n_rounds"] == 0


def test_leaf_and_duplicate_edges(ictx):
    """Ragged inputs: a leaf (k = 1: vertex deleted, no new edge), duplicate i-j edges (manhattan has
    145), and a vertex listed twice."""
    d = 3
    ids = np.arange(6, dtype=np.int32)
    poses = np.array([[0, 0, 0], [1, 0, 0.1], [2, 0.2, 0.2], [3, 0.1, 0.1], [1, 1, 1.0], [2, 2, -1.0]], float)
    ij = np.array([[0, 1], [1, 2], [1, 2], [2, 3], [1, 4], [3, 5]], np.int32)
    info = np.array([50, 0, 0, 50, 0, 100.0])
    data = []
    for a, b in ij:
        c, s = np.cos(poses[a, 2]), np.sin(poses[a, 2])
        dx, dy = poses[b, :2] - poses[a, :2]
        data.append(np.concatenate([[c * dx + s * dy, -s * dx + c * dy, poses[b, 2] - poses[a, 2]], info]))
    g = {"pose_dim": d, "ids": ids, "poses": poses, "edge_ij": ij, "edge_data": np.array(data)}
    which = np.array([5, 2, 2, 4], np.int32)
    opts = abi.make_options(3)
    og = oracle_lib.OracleGraph.from_dict(g)
    assert og.marginalize(np.array([5, 2, 4], np.int32), opts) == 0
    hg = GraphWrapperHIP.from_dict(g, ctx=ictx)
    st = hg.marginalizeNoOptimize(which, opts)
    assert st["n_removed"] == 3
    util.compare_edge_sets(3, og.edges(), hg.edges(), rtol=1e-11)
    ids_left, _ = hg.vertices()
    assert list(ids_left) == [0, 1, 3]


def test_reserve_changes_nothing_but_memory(ictx):
    """spg_graph_reserve sizes and touches the host-side buffers of a marginalisation (include/spg.h); the result is the same
    graph as without it, and the arena's capacity is what was asked for."""
    g, which, opts, gold_edges, gold_bl, gold_vids = util.load_golden("sphere_nfr_tree")
    a = GraphWrapperHIP.from_dict(g, ctx=ictx)
    b = GraphWrapperHIP.from_dict(g, ctx=ictx)
    need = int(len(g["ids"]) * 7 + len(g["edge_ij"]) * 28) * 3
    b.reserve(need)
    _, cap = b.arena()
    assert cap >= need
    sa, sb = a.marginalizeNoOptimize(which, opts), b.marginalizeNoOptimize(which, opts)
    assert sa["n_removed"] == sb["n_removed"] and sa["kld_sum"] == sb["kld_sum"]
    util.compare_edge_sets(g["pose_dim"], a.edges(), b.edges(), rtol=0)
    util.compare_edge_sets(g["pose_dim"], gold_edges, b.edges(), rtol=1e-11)
