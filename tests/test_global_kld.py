"""Global Kullback-Leibler divergence: the oracle restatement of
GraphWrapperG2O::kullbackLeibler (src/graph_wrapper_g2o.cpp:531-548) + kullbackLeiblerDivergence
(src/utils.cpp:70-97) pinned against numpy/LAPACK and against the invariants of the formula.
PARITY UNPINNED against the reference itself (no golden vectors, reference not buildable here)."""
import numpy as np
import pytest

from tests import oracle_lib, util
from sparsifyposegraph_amd import abi

PREFIX = 140


def _pair(case, n=PREFIX):
    g, which, opts, *_ = util.load_golden(case)
    sub, w = util.prefix_graph(g, which, n)
    base = oracle_lib.OracleGraph.from_dict(sub)
    other = oracle_lib.OracleGraph.from_dict(sub)
    assert other.marginalize(w, opts) == 0
    return sub, w, opts, base, other


@pytest.mark.parametrize("case", ["intel_nfr_tree_sp3", "sphere_nfr_tree", "manhattan_glc_tree", "sphere_glc_tree"])
def test_oracle_global_kld_matches_numpy(case):
    sub, w, opts, base, other = _pair(case)
    fid = int(min(sub["ids"]))
    d = sub["pose_dim"]
    Hb, Ho = base.information(fid), other.information(fid)
    assert np.abs(Hb - Hb.T).max() <= 1e-12 * np.abs(Hb).max()
    ids_b = [int(i) for i in sorted(sub["ids"]) if int(i) != fid]
    ids_o = [int(i) for i in sorted(other.vertices()[0]) if int(i) != fid]
    keep = [ids_b.index(i) * d + a for i in ids_o for a in range(d)]
    marg = [i * d + a for i, v in enumerate(ids_b) if v not in set(ids_o) for a in range(d)]
    S = Hb[np.ix_(keep, keep)] - Hb[np.ix_(marg, keep)].T @ np.linalg.solve(Hb[np.ix_(marg, marg)], Hb[np.ix_(marg, keep)])
    n = len(keep)
    inner = np.trace(np.linalg.solve(S, Ho))
    ldx, ldy = np.linalg.slogdet(Ho)[1], np.linalg.slogdet(S)[1]
    kld = 0.5 * (inner - ldx + ldy - n)   # same estimates in both graphs: Mahalanobis term is 0
    t = base.kullback_leibler(other, fid)
    assert t["n"] == n and t["mahalanobis"] == 0.0
    assert abs(t["innerprod"] - inner) <= 1e-9 * n
    assert abs(t["logdetx"] - ldx) <= 1e-10 * abs(ldx)
    assert abs(t["logdety"] + ldy) <= 1e-10 * abs(ldy)   # reference sign: logdety = -sum log D(maty)
    assert abs(t["kld"] - kld) <= 1e-9 * n
    assert t["kld"] > 0


def test_oracle_global_kld_invariants():
    # a graph against itself: exactly the same information, KLD = 0
    sub, w, opts, base, other = _pair("sphere_nfr_tree", 80)
    fid = int(min(sub["ids"]))
    t = base.kullback_leibler(base, fid)
    assert abs(t["kld"]) <= 1e-9 and t["n"] == 6 * (len(sub["ids"]) - 1)
    # Dense GLC reproduces the marginal exactly (up to the 1e-8 eigenvalue cut): KLD ~ 0
    sub, w, opts, base, other = _pair("manhattan_glc_dense", 120)
    t = base.kullback_leibler(other, int(min(sub["ids"])))
    assert abs(t["kld"]) <= 1e-6
    # GLC Tree and NFR Tree carry the same Chow-Liu tree approximation: equal global KLD
    g, which, o_nfr, *_ = util.load_golden("manhattan_nfr_tree")
    sub, w = util.prefix_graph(g, which, 120)
    base = oracle_lib.OracleGraph.from_dict(sub)
    a, b = oracle_lib.OracleGraph.from_dict(sub), oracle_lib.OracleGraph.from_dict(sub)
    assert a.marginalize(w, o_nfr) == 0
    assert b.marginalize(w, abi.make_options(3, abi.ALG_GLC, abi.TOPO_TREE)) == 0
    fid = int(min(sub["ids"]))
    ka, kb = base.kullback_leibler(a, fid)["kld"], base.kullback_leibler(b, fid)["kld"]
    assert abs(ka - kb) <= 1e-6 * max(1.0, abs(ka))


def test_oracle_global_kld_mahalanobis_term():
    # moving one estimate of the sparsified graph adds diff^T infox diff (estimateDifference)
    g, which, opts, *_ = util.load_golden("intel_nfr_tree_sp3")
    sub, w = util.prefix_graph(g, which, 60)
    base = oracle_lib.OracleGraph.from_dict(sub)
    sub2 = dict(sub)
    sub2["poses"] = np.array(sub["poses"], float).copy()
    sub2["poses"][7] += [0.01, -0.02, 0.003]
    moved = oracle_lib.OracleGraph.from_dict(sub2)
    fid = int(min(sub["ids"]))
    t = base.kullback_leibler(moved, fid)
    Hm = moved.information(fid)
    ids = [int(i) for i in sorted(sub["ids"]) if int(i) != fid]
    diff = np.zeros(len(ids) * 3)
    k = ids.index(int(sub["ids"][7]))
    diff[3 * k:3 * k + 3] = [-0.01, 0.02, -0.003]     # baseline minus other
    assert abs(t["mahalanobis"] - diff @ Hm @ diff) <= 1e-9 * max(1.0, t["mahalanobis"])


def test_product_global_kld_needs_the_device():
    """The product has no CPU path for the dense KLD: with an injected (CPU) backend the entry points
    fail loudly instead of falling back."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    from sparsifyposegraph_amd.lib import SpgError
    g, which, opts, *_ = util.load_golden("intel_nfr_tree_sp3")
    sub, w = util.prefix_graph(g, which, 30)
    ictx = oracle_lib.injected_context()
    a, b = GraphWrapperHIP.from_dict(sub, ctx=ictx), GraphWrapperHIP.from_dict(sub, ctx=ictx)
    with pytest.raises(SpgError, match="HIP backend"):
        a.kullbackLeibler(b)
    with pytest.raises(SpgError, match="HIP backend"):
        a.information()


def test_product_optimize_and_chi2_need_the_device():
    """optimize() / chi2() are device paths too: no CPU fallback behind an injected backend, and invalid
    arguments are reported, not asserted."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    from sparsifyposegraph_amd.lib import SpgError
    g, which, opts, *_ = util.load_golden("intel_nfr_tree_sp3")
    sub, w = util.prefix_graph(g, which, 20)
    a = GraphWrapperHIP.from_dict(sub, ctx=oracle_lib.injected_context())
    with pytest.raises(SpgError, match="HIP backend"):
        a.optimize()
    with pytest.raises(SpgError, match="HIP backend"):
        a.chi2()
    with pytest.raises(SpgError):
        a.optimize(iterations=-1)
