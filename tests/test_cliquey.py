"""Correlated NFR patterns: SparsityOptions::CliqueySubgraph / ::CliqueyDense group the Chow-Liu tree's
measurements into MultiEdgeCorrelated edges (src/pseudo_chow_liu.cpp:62-85,198-251, src/topology_provider_binary.hpp:48-67,
src/multi_edge_correlated.hpp:65-140) whose joint information has a closed form (src/logdet_function.cpp:236-279).
CPU: the oracle through invariants — CliqueyDense reproduces the target exactly (per-blanket KLD 0, global KLD of the
sparsified graph 0: the same statement GLC Dense makes), CliqueySubgraph sits between the tree and that. GPU: the
generic NFR kernel (csrc/spg_nfr_ip.hip) and the assembly kernels with SPG_EDGE_MULTI edges against the oracle."""
import os
import numpy as np
import pytest

from sparsifyposegraph_amd import abi
from tests import oracle_lib, util


def _opts(d, topo, chord=1.0):
    o = abi.make_options(d, abi.ALG_NFR, topo)
    o.chord_ratio = chord
    return o


CASES = [("sphere_nfr_tree", abi.TOPO_CLIQUEY_DENSE, 1.0), ("sphere_nfr_tree", abi.TOPO_CLIQUEY_SUBGRAPH, 0.5),
         ("manhattan_nfr_tree", abi.TOPO_CLIQUEY_DENSE, 1.0), ("manhattan_nfr_tree", abi.TOPO_CLIQUEY_SUBGRAPH, 0.5),
         ("parking_nfr_tree", abi.TOPO_CLIQUEY_SUBGRAPH, 0.3), ("intel_nfr_tree_sp3", abi.TOPO_CLIQUEY_DENSE, 1.0)]


@pytest.mark.parametrize("case,topo,chord", CASES)
def test_oracle_correlated_patterns(case, topo, chord, oracle):
    g, which, opts, *_ = util.load_golden(case)
    d = opts.pose_dim
    batch, roots = util.first_round_batch(g, which, _opts(d, topo, chord))
    ref = abi.marginalize_batch(oracle, None, _opts(d, topo, chord), batch)
    tree = abi.marginalize_batch(oracle, None, _opts(d, abi.TOPO_TREE), batch)
    assert (ref["status"] == 0).all()
    fin = np.isfinite(ref["kld"])
    ne = ref["new_edge_off"][-1]
    kinds = ref["new_edge_kind"][:ne]
    assert (kinds == abi.EDGE_MULTI).sum() >= 5 and set(kinds) <= {abi.EDGE_BINARY, abi.EDGE_MULTI}
    if topo == abi.TOPO_CLIQUEY_DENSE:
        assert np.abs(ref["kld"][fin]).max() <= 1e-10          # one fully correlated edge carries the whole target
        assert (np.diff(ref["new_edge_off"])[fin] == 1).all()
    else:
        assert (ref["kld"][fin] <= tree["kld"][fin] + 1e-9).all() and (ref["kld"][fin] >= -1e-9).all()
    # every measurement of a correlated edge is a tree measurement, its information is PD
    for e in np.nonzero(kinds == abi.EDGE_MULTI)[0][:30]:
        data = ref["new_edge_data"][ref["new_edge_data_off"][e]:ref["new_edge_data_off"][e + 1]]
        pairs, meas, om = util.multi_parts(d, data)
        nv = ref["new_edge_vert_off"][e + 1] - ref["new_edge_vert_off"][e]
        assert len(data) == 1 + 2 * len(pairs) + len(pairs) * abi.pose_stride(d) + (d * len(pairs)) ** 2
        assert pairs.max() == nv - 1 and len(pairs) == nv - 1 and np.linalg.eigvalsh(om).min() > 0


@pytest.mark.parametrize("case,n,topo,chord", [("sphere_nfr_tree", 150, abi.TOPO_CLIQUEY_DENSE, 1.0), ("manhattan_nfr_tree", 250, abi.TOPO_CLIQUEY_DENSE, 1.0),
                                               ("manhattan_nfr_tree", 250, abi.TOPO_CLIQUEY_SUBGRAPH, 0.5)])
def test_oracle_whole_graph_with_correlated_edges(case, n, topo, chord):
    """Later blankets contain the correlated edges earlier ones produced. CliqueyDense keeps the marginal exactly: the
    global KLD of the sparsified graph against its baseline is 0 — which exercises the multi edges' Jacobians and
    information in assembly, Schur complement and KLD at once."""
    g, which, opts, *_ = util.load_golden(case)
    sub, w = util.prefix_graph(g, which, n)
    fid = int(min(sub["ids"]))
    ob, og = oracle_lib.OracleGraph.from_dict(sub), oracle_lib.OracleGraph.from_dict(sub)
    assert og.marginalize(w, _opts(opts.pose_dim, topo, chord)) == 0
    b = og.blankets()
    assert (b["status"] == 0).all() and (og.edges()["kind"] == abi.EDGE_MULTI).sum() >= 2
    r = ob.kullback_leibler(og, fid)
    ot = oracle_lib.OracleGraph.from_dict(sub)
    assert ot.marginalize(w, _opts(opts.pose_dim, abi.TOPO_TREE)) == 0
    rt = ob.kullback_leibler(ot, fid)
    if topo == abi.TOPO_CLIQUEY_DENSE:
        assert abs(r["kld"]) <= 1e-9 * r["n"]
    else:
        assert -1e-9 <= r["kld"] <= rt["kld"]


def test_fill_cliques_keeps_every_tree_edge_in_one_clique():
    """fillCliques (src/pseudo_chow_liu.cpp:198-251) merges cliques that start as the edges of a TREE, so they stay
    edge-disjoint connected subtrees: every tree edge ends in exactly one clique, the measurements of the pattern add up to
    the rank (hasClosedFormSolution, src/logdet_function.cpp:83-86) and the interior point over correlated blocks
    (src/logdet_function.cpp:135-214) is never reached from CliqueySubgraph. Random trees of 3-40 vertices, every budget m."""
    import ctypes as C
    L = oracle_lib.lib()
    L.spgref_fill_cliques.restype = C.c_int
    rng = np.random.default_rng(5)
    merged = 0
    for trial in range(300):
        k = int(rng.integers(3, 41))
        perm = rng.permutation(k)
        pairs = np.array([(perm[i], perm[rng.integers(0, i)]) for i in range(1, k)], np.int32)
        pairs = np.ascontiguousarray(pairs[rng.permutation(k - 1)])
        for chord in (0.0, 0.25, 0.5, 1.0, 3.0):
            m = int((1 + chord) * (k - 1))
            clique_of, count = np.zeros(k - 1, np.int32), np.zeros(k - 1, np.int32)
            nc = L.spgref_fill_cliques(k, m, pairs.ctypes.data_as(C.POINTER(C.c_int32)), clique_of.ctypes.data_as(C.POINTER(C.c_int32)),
                                       count.ctypes.data_as(C.POINTER(C.c_int32)))
            assert (count == 1).all(), (k, m, pairs.tolist(), count.tolist())
            assert 1 <= nc <= k - 1
            merged += nc < k - 1
    assert merged > 300


@pytest.mark.gpu
@pytest.mark.parametrize("case,topo,chord", CASES)
def test_device_correlated_patterns_match_oracle(case, topo, chord, hip_ctx, oracle):
    g, which, opts, *_ = util.load_golden(case)
    d = opts.pose_dim
    o = _opts(d, topo, chord)
    batch, roots = util.first_round_batch(g, which, o)
    ref = abi.marginalize_batch(oracle, None, o, batch)
    got = hip_ctx.marginalize_batch(o, batch)
    assert np.array_equal(ref["status"], got["status"])
    for key in ("new_edge_off", "new_edge_kind", "new_edge_vert_off", "new_edge_vert", "new_edge_data_off"):
        ne = ref["new_edge_off"][-1]
        n = {"new_edge_off": len(ref[key]), "new_edge_kind": ne, "new_edge_vert_off": ne + 1, "new_edge_vert": ref["new_edge_vert_off"][ne], "new_edge_data_off": ne + 1}[key]
        assert np.array_equal(ref[key][:n], got[key][:n]), key
    worst = 0.0
    ps = abi.pose_stride(d)
    for e in range(ref["new_edge_off"][-1]):
        xa = ref["new_edge_data"][ref["new_edge_data_off"][e]:ref["new_edge_data_off"][e + 1]]
        xb = got["new_edge_data"][got["new_edge_data_off"][e]:got["new_edge_data_off"][e + 1]]
        if ref["new_edge_kind"][e] == abi.EDGE_MULTI:
            (pa, ma, oa), (pb, mb, ob) = util.multi_parts(d, xa), util.multi_parts(d, xb)
            assert np.array_equal(pa, pb)
            worst = max(worst, util.rel_err(ma, mb), util.rel_err(oa, ob))
        else:
            worst = max(worst, util.rel_err(xa[:ps], xb[:ps]), util.rel_err(xa[ps:], xb[ps:]))
    fin = np.isfinite(ref["kld"])
    kerr = np.abs(ref["kld"][fin] - got["kld"][fin]).max()
    print(f"{case} topo={topo}: {int((ref['new_edge_kind'][:ref['new_edge_off'][-1]] == abi.EDGE_MULTI).sum())} correlated edges, worst rel err {worst:.1e}, KLD abs err {kerr:.1e}")
    assert worst <= 1e-9 and kerr <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("case,n,topo,chord", [("sphere_nfr_tree", 200, abi.TOPO_CLIQUEY_DENSE, 1.0), ("manhattan_nfr_tree", 300, abi.TOPO_CLIQUEY_DENSE, 1.0),
                                               ("manhattan_nfr_tree", 300, abi.TOPO_CLIQUEY_SUBGRAPH, 0.5), ("parking_nfr_tree", 200, abi.TOPO_CLIQUEY_SUBGRAPH, 0.3),
                                               ("intel_nfr_tree_sp3", 300, abi.TOPO_CLIQUEY_SUBGRAPH, 0.4)])
def test_device_whole_graph_with_correlated_edges(case, n, topo, chord, hip_ctx):
    """Through the round scheduler: graph identical to the sequential oracle's (edges to 1e-9), and the device's global KLD
    of the result — dense assembly with SPG_EDGE_MULTI edges — equal to the oracle's (0 for CliqueyDense)."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    g, which, opts, *_ = util.load_golden(case)
    d = opts.pose_dim
    sub, w = util.prefix_graph(g, which, n)
    o = _opts(d, topo, chord)
    hg = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx)
    st = hg.marginalizeNoOptimize(w, o)
    og = oracle_lib.OracleGraph.from_dict(sub)
    assert og.marginalize(w, o) == 0 and st["n_bad_status"] == 0
    worst = util.compare_edge_sets(d, og.edges(), hg.edges(), rtol=1e-9)
    kref = float(np.nansum(og.blankets()["kld"]))
    assert abs(st["kld_sum"] - kref) <= 1e-9 * max(1.0, abs(kref))
    fid = int(min(sub["ids"]))
    hb, ob = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx), oracle_lib.OracleGraph.from_dict(sub)
    kld = hb.kullbackLeibler(hg)
    r = ob.kullback_leibler(og, fid)
    assert abs(kld - r["kld"]) <= 1e-9 * r["n"]
    if topo == abi.TOPO_CLIQUEY_DENSE:
        assert abs(kld) <= 1e-9 * r["n"]
    # chi2 / optimize see the correlated edges too
    c_dev, c_ref = hg.chi2(), og.chi2(fid)
    assert c_dev == pytest.approx(c_ref, rel=1e-9, abs=1e-12)
    n_multi = int((og.edges()["kind"] == abi.EDGE_MULTI).sum())
    print(f"{case} topo={topo}: {n_multi} correlated edges in the result, worst edge rel err {worst:.1e}, global KLD {kld:.3e} (oracle {r['kld']:.3e})")


@pytest.mark.parametrize("case,d", [("manhattan_nfr_tree", 3), ("sphere_nfr_tree", 6)])
def test_host_path_with_correlated_edges(case, d, tmp_path):
    """CPU: the product's host code (round scheduler, budgets, commit, clonePortion, .g2o writer) with the oracle injected as
    the arithmetic: CliqueyDense through spg_graph_marginalize equals the oracle's own sequential run, the written file
    holds MULTI_EDGE_* lines in the reference's format (src/multi_edge_correlated.hpp:227-267: '|| nmeas nrelevant
    measurements information-upper'), a clone carries the correlated edges, spg_graph_add_multi_edge validates its input."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    from sparsifyposegraph_amd.lib import SpgError
    g, which, opts, *_ = util.load_golden(case)
    sub, w = util.prefix_graph(g, which, 120)
    o = _opts(d, abi.TOPO_CLIQUEY_DENSE)
    hg = GraphWrapperHIP.from_dict(sub, ctx=oracle_lib.injected_context())
    st = hg.marginalizeNoOptimize(w, o)
    og = oracle_lib.OracleGraph.from_dict(sub)
    assert og.marginalize(w, o) == 0 and st["n_bad_status"] == 0
    util.compare_edge_sets(d, og.edges(), hg.edges(), rtol=1e-12)
    e = hg.edges()
    multi = np.nonzero(e["kind"] == abi.EDGE_MULTI)[0]
    assert len(multi) >= 2
    lines = [l for l in hg.writeString().splitlines() if l.startswith("MULTI_EDGE")]
    assert len(lines) == len(multi)
    ps = abi.pose_stride(d)
    for l in lines:
        t = l.split()
        assert t[0] == ("MULTI_EDGE_SE2" if d == 3 else "MULTI_EDGE_SE3")
        bar = t.index("||")
        q, nm, nrel = bar - 1, int(t[bar + 1]), int(t[bar + 2])
        r = d * nm
        assert nrel == ps and nm == q - 1 and len(t) == bar + 3 + nm * ps + r * (r + 1) // 2
    # ... and such a file is REFUSED on load (the format omits which vertex pair a measurement belongs to: silently dropping
    # the lines would hand back a graph without its correlated constraints)
    path = tmp_path / "cliquey.g2o"
    path.write_text(hg.writeString())
    with pytest.raises(SpgError, match="MULTI_EDGE"):
        GraphWrapperHIP.load(str(path), ctx=oracle_lib.injected_context())
    # clonePortion keeps the correlated edges whose vertices survive the cut
    top = int(max(hg.vertices()[0]))
    clone = hg.clonePortion(top, optimize=False)     # (optimize() is a device path)
    util.compare_edge_sets(d, hg.edges(), clone.edges(), rtol=0)
    # add_multi_edge: a record of the wrong length / a measurement that points outside the edge is refused
    i0 = multi[0]
    ids = e["vert_ids"][e["vert_off"][i0]:e["vert_off"][i0 + 1]]
    rec = e["data"][e["data_off"][i0]:e["data_off"][i0 + 1]].copy()
    fresh = GraphWrapperHIP.from_dict(sub, ctx=oracle_lib.injected_context())
    fresh.addMultiEdge(ids, rec)
    assert fresh.numEdges() == len(sub["edge_ij"]) + 1
    with pytest.raises(SpgError):
        fresh.addMultiEdge(ids, rec[:-1])
    bad = rec.copy()
    bad[1] = len(ids)
    with pytest.raises(SpgError):
        fresh.addMultiEdge(ids, bad)


@pytest.mark.gpu
def test_device_rank_deficient_blankets_choose_dimensions(hip_ctx):
    """parking.g2o at full size under CliqueySubgraph(0.5): six of the 828 blankets have more than d eigenvalues below the
    cutoff, the reference's chooseDimensions branch (src/logdet_function.cpp:40-59,66-81) — the generic kernel's eigen route;
    everything else takes its gauge route. Statuses, rank-deficiency flags, every KLD and the whole graph equal the oracle's."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    g, which, opts, *_ = util.load_golden("parking_full_nfr_tree")
    o = _opts(6, abi.TOPO_CLIQUEY_SUBGRAPH, 0.5)
    hg = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    st = hg.marginalizeNoOptimize(which, o)
    og = oracle_lib.OracleGraph.from_dict(g)
    assert og.marginalize(which, o) == 0 and st["n_bad_status"] == 0
    hb, ob = hg.blankets(), og.blankets()
    at = {int(r): i for i, r in enumerate(hb["root"])}
    idx = np.array([at[int(r)] for r in ob["root"]])
    assert np.array_equal(ob["status"], hb["status"][idx])
    assert int((ob["info"] & 1).sum()) >= 3 and np.array_equal(ob["info"] & 1, hb["info"][idx] & 1)
    fin = np.isfinite(ob["kld"])
    assert np.abs(ob["kld"][fin] - hb["kld"][idx][fin]).max() <= 1e-9
    worst = util.compare_edge_sets(6, og.edges(), hg.edges(), rtol=1e-9)
    print(f"parking full, CliqueySubgraph(0.5): {int((ob['info'] & 1).sum())} rank-deficient blankets, worst edge rel err {worst:.1e}")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["manhattan_cliquey_dense", "sphere_cliquey_subgraph"])
def test_device_reproduces_correlated_fixtures(case, hip_ctx):
    """The committed fixtures (tests/golden/make_golden.py: oracle outputs on dataset prefixes) carry the bar to the GPU box."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    g, which, opts, gold_edges, gold_bl, gold_vids = util.load_golden(case)
    hg = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    st = hg.marginalizeNoOptimize(which, opts)
    assert st["n_bad_status"] == 0 and np.array_equal(hg.vertices()[0], gold_vids)
    worst = util.compare_edge_sets(g["pose_dim"], gold_edges, hg.edges(), rtol=1e-9)
    assert abs(st["kld_sum"] - float(np.nansum(gold_bl["kld"]))) <= 1e-9 * max(1.0, abs(float(np.nansum(gold_bl["kld"]))))
    print(f"{case}: {int((gold_edges['kind'] == abi.EDGE_MULTI).sum())} correlated edges, worst rel err {worst:.1e}")


def _hub_graph_se2(k=45, weak=(7, 21, 40), seed=11):
    """SE2: a hub with k neighbours, those in `weak` measured in translation only (information of rank 2), a chain among the
    others: the hub's blanket has 3 k variables and a target that lacks the headings of the weak neighbours."""
    rng = np.random.default_rng(seed)
    n = k + 1
    poses = np.zeros((n, 3))
    for i in range(1, n):
        a = 2 * np.pi * i / k
        poses[i] = [8 * np.cos(a), 8 * np.sin(a), a + rng.normal(scale=0.3)]

    def rel(a, b):
        c, s = np.cos(poses[a, 2]), np.sin(poses[a, 2])
        dx, dy = poses[b, :2] - poses[a, :2]
        return np.array([c * dx + s * dy, -s * dx + c * dy, poses[b, 2] - poses[a, 2]]) + rng.normal(scale=0.01, size=3)
    full = np.diag([50.0, 50.0, 200.0])[np.triu_indices(3)]
    trans = np.diag([50.0, 50.0, 0.0])[np.triu_indices(3)]
    ij, data = [], []
    for i in range(1, n):
        ij.append((0, i)); data.append(np.concatenate([rel(0, i), trans if i in weak else full]))
    for i in range(1, n - 1):
        if i in weak or i + 1 in weak:
            continue
        ij.append((i, i + 1)); data.append(np.concatenate([rel(i, i + 1), full]))
    return {"pose_dim": 3, "ids": np.arange(n, dtype=np.int32), "poses": poses, "edge_ij": np.array(ij, np.int32), "edge_data": np.array(data)}


@pytest.mark.gpu
@pytest.mark.parametrize("topo", [abi.TOPO_CLIQUEY_DENSE, abi.TOPO_CLIQUEY_SUBGRAPH])
def test_device_large_rank_deficient_blanket_takes_the_ql_eigen_route(topo, hip_ctx):
    """A blanket of 135 variables whose target is rank-deficient beyond the gauge (three neighbours of the hub measured in
    translation only): more than d eigenvalues below the cutoff — the chooseDimensions branch
    (src/logdet_function.cpp:40-59,66-81) — at a size where the generic kernel's eigen route is the tridiagonal QL solver
    (from 128 variables on) instead of the Jacobi sweeps the oracle and the small blankets use. Status, flag and topology
    equal the oracle's. The payload is compared at 1e-4, not 1e-9: the null space of this target is exactly degenerate (three
    gauge directions + three unobserved headings), an eigen-solver returns an arbitrary basis of it, and chooseDimensions
    picks individual vectors of that basis — the result depends on the solver at the 1e-8 ... 1e-5 level (the reference's is
    Eigen's tridiagonal QR, a third basis). With the Jacobi route on both sides (SPG_EIG_JACOBI=1) the same blanket agrees
    to 1e-13; measured with the QL route: 1.1e-8 (CliqueyDense), 9.0e-6 (CliqueySubgraph). parking.g2o under CliqueyDense at full
    size (11 blankets on this route) gives the same global KLD with either solver to 1e-10 (23.17327419)."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    g = _hub_graph_se2()
    which = np.array([0], np.int32)
    o = _opts(3, topo, 1.0 if topo == abi.TOPO_CLIQUEY_DENSE else 0.5)
    og = oracle_lib.OracleGraph.from_dict(g)
    assert og.marginalize(which, o) == 0
    hg = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    st = hg.marginalizeNoOptimize(which, o)
    hb, ob = hg.blankets(), og.blankets()
    assert np.array_equal(hb["status"], ob["status"]) and (ob["status"] == 0).all() and st["n_bad_status"] == 0
    assert (ob["info"] & abi.INFO_RANK_DEFICIENT).all() and np.array_equal(hb["info"] & 1, ob["info"] & 1)
    kerr = float(np.abs(hb["kld"] - ob["kld"]).max())
    worst = util.compare_edge_sets(3, og.edges(), hg.edges(), rtol=1e-4)
    print(f"hub blanket, 135 variables, topology {topo}: KLD {hb['kld'][0]:.9g} (oracle {ob['kld'][0]:.9g}), worst edge rel err {worst:.2e}")
    # (the blanket's KLD is taken over the chosen dimensions with the clamped 1 / lambda of the kept near-null ones: it moves with
    #  the basis far more than the edges do — 18.003 against the oracle's 17.934 under CliqueySubgraph, equal under CliqueyDense)
    assert kerr <= 1e-2 * max(1.0, abs(float(ob["kld"][0])))


@pytest.mark.gpu
def test_device_slow_blanket_behind_fast_ones_is_waited_for(hip_ctx, monkeypatch):
    """The bug behind round 2's core dump. A batch's narrow blankets go to a bin kernel, its clusters to the generic NFR
    kernel launched behind it on the same stream. The commit polls the ready words for 5 s and then falls back to waiting for
    the launch slot — which used to mean the event behind the BIN kernel only: with a cluster still running (parking.g2o under
    CliqueyDense: clusters of 150-190 vertices take tens of seconds) the commit took whatever the mailbox held at the cluster's
    record — zeros in a fresh process (a blanket "without new edges": a silently wrong, disconnected graph and SPG_ENOTPD from
    the global KLD), a stale record otherwise (indices out of range: a segmentation fault). Here the fallback is forced
    (SPG_POLL_SPIN_S=0) on sphere.g2o prefixes whose batches mix both kinds, on a context that has run another graph before;
    CliqueyDense must leave the global KLD at zero, which a single dropped or stale record destroys."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    g0, which0, opts0, *_ = util.load_golden("manhattan_cliquey_dense")
    warm = GraphWrapperHIP.from_dict(g0, ctx=hip_ctx)
    assert warm.marginalizeNoOptimize(which0, opts0)["n_bad_status"] == 0
    monkeypatch.setenv("SPG_POLL_SPIN_S", "0")
    g, which, opts, *_ = util.load_golden("sphere_full_nfr_tree")
    sub, w = util.prefix_graph(g, which, 700)
    o = _opts(6, abi.TOPO_CLIQUEY_DENSE, 1.0)
    for _ in range(2):
        hg = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx)
        base = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx)
        st = hg.marginalizeNoOptimize(w, o)
        assert st["n_bad_status"] == 0 and st["n_removed"] == len(w) and st["max_blanket"] >= 40
        kld = base.kullbackLeibler(hg)
        assert abs(kld) <= 1e-7 * len(sub["ids"]), kld


@pytest.mark.gpu
def test_device_parking_full_cliquey_dense_then_global_kld(hip_ctx):
    """parking.g2o at full size under CliqueyDense, then the global KLD of the result (correlated edges of up to ~185
    measurements), after another graph on the same context: the sequence of round 2's core dump, end to end. Every call has to
    come back with a result or an SPG_E* code. Measured in round 3: 828 vertices removed, no bad status, largest blanket 196
    vertices, global KLD 23.1732742 — not zero: parking.g2o has blankets whose target is rank deficient beyond the gauge
    (chooseDimensions, src/logdet_function.cpp:40-59), which one correlated edge does not reproduce; the same value to 1e-10
    with the Jacobi and with the QL eigen route. (Parity of that number is unpinned: the oracle needs hours for this run.) 363 s
    when the test was written and kept behind SPG_SLOW_TESTS; 9 s since the cluster path runs on the matrix cores."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    from sparsifyposegraph_amd.lib import SpgError
    g0, which0, opts0, *_ = util.load_golden("manhattan_cliquey_dense")
    warm = GraphWrapperHIP.from_dict(g0, ctx=hip_ctx)
    assert warm.marginalizeNoOptimize(which0, opts0)["n_bad_status"] == 0
    g, which, opts, *_ = util.load_golden("parking_full_nfr_tree")
    o = _opts(6, abi.TOPO_CLIQUEY_DENSE, 1.0)
    hg = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    base = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    st = hg.marginalizeNoOptimize(which, o)
    assert st["n_bad_status"] == 0 and st["n_removed"] == len(which)
    e = hg.edges()
    assert set(np.unique(e["kind"]).tolist()) <= {abi.EDGE_BINARY, abi.EDGE_MULTI}
    try:
        kld = base.kullbackLeibler(hg)
        assert np.isfinite(kld) and kld > -1e-6 * len(g["ids"])
        print(f"parking.g2o CliqueyDense at full size: largest blanket {st['max_blanket']}, global KLD {kld:.9g}")
        assert abs(kld - 23.1732742) <= 1e-4          # (the value every build of round 3 has produced, either eigen route)
    except SpgError as ex:
        assert "code -9" in str(ex), str(ex)
