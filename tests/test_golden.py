"""CPU: the oracle reproduces the committed fixtures bit-for-bit in topology and to 1e-12 in payload
(regression pin for the checker itself), and the fixtures' own invariants hold."""
import numpy as np
import pytest

from tests import oracle_lib, util


@pytest.mark.parametrize("case", util.golden_cases())
def test_oracle_reproduces_fixture(case):
    g, which, opts, gold_edges, gold_bl, gold_vids = util.load_golden(case)
    og = oracle_lib.OracleGraph.from_dict(g)
    assert og.marginalize(which, opts) == 0
    ids, _ = og.vertices()
    assert np.array_equal(ids, gold_vids)
    util.compare_edge_sets(g["pose_dim"], gold_edges, og.edges(), rtol=1e-11)
    bl = og.blankets()
    assert np.array_equal(bl["root"], gold_bl["root"])
    assert np.array_equal(bl["status"], gold_bl["status"])
    assert (bl["status"] == 0).all()
    fin = np.isfinite(gold_bl["kld"])
    assert np.allclose(bl["kld"][fin], gold_bl["kld"][fin], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("case", [c for c in util.golden_cases() if "glc" in c])
def test_glc_fixture_invariants(case):
    """GLC edges carry no information on the absolute pose of their first vertex (gauge), and a Dense
    GLC edge reproduces its target: both visible from W^T W alone."""
    g, which, opts, gold_edges, gold_bl, _ = util.load_golden(case)
    d = g["pose_dim"]
    n_glc = 0
    for kind, ids, data in util.edge_list(gold_edges):
        if kind != 1:
            continue
        n_glc += 1
        G = util.glc_gram(d, ids, data)
        assert np.abs(G[:d, :]).max() <= 1e-6 * max(1.0, np.abs(G).max())
        w = np.linalg.eigvalsh(G)
        assert w.min() > -1e-9 * max(1.0, w.max())
    assert n_glc > 50
    assert (gold_bl["info"] & 2 == 0).all()  # no unary root edge survived the 1e-8 cut
