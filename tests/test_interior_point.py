"""Interior-point NFR: SparsityOptions::Dense / ::Subgraph patterns with more than k-1 pose-pose
edges have no closed form; the reference runs a log-barrier Newton method over the edge informations
(src/optimizer.cpp:38-79, src/logdet_function.cpp:87-214,348-427, src/pqn/pqn_optimizer.cpp:29-126,
src/pqn/line_search.cpp:12-37). PARITY UNPINNED against the reference itself (no vectors, not buildable here).
CPU: the oracle's restatement against an INDEPENDENT solution of the same convex problem (scipy, X_e = L_e L_e^T) and
against the properties of the optimum. GPU: csrc/spg_nfr_ip.hip against the oracle."""
import numpy as np
import pytest

from sparsifyposegraph_amd import abi
from tests import oracle_lib, util


def _opts(d, topo, chord=1.0):
    o = abi.make_options(d, abi.ALG_NFR, topo)
    o.chord_ratio = chord
    return o


def _blocks(d, ref, b):
    """(pairs, X blocks, target) of blanket b of a marginalize_batch result"""
    ps, il = abi.pose_stride(d), d * (d + 1) // 2
    e0, e1 = ref["new_edge_off"][b], ref["new_edge_off"][b + 1]
    out = []
    for e in range(e0, e1):
        v = ref["new_edge_vert"][ref["new_edge_vert_off"][e]:ref["new_edge_vert_off"][e + 1]]
        data = ref["new_edge_data"][ref["new_edge_data_off"][e]:ref["new_edge_data_off"][e + 1]]
        X = np.zeros((d, d))
        X[np.triu_indices(d)] = data[ps:ps + il]
        X = X + np.triu(X, 1).T
        out.append((tuple(int(x) for x in v), X))
    return out


@pytest.mark.parametrize("case,topo,chord", [("sphere_nfr_tree", abi.TOPO_DENSE, 1.0), ("manhattan_nfr_tree", abi.TOPO_DENSE, 1.0),
                                             ("manhattan_nfr_tree", abi.TOPO_SUBGRAPH, 0.5), ("intel_nfr_tree_sp3", abi.TOPO_SUBGRAPH, 0.4)])
def test_oracle_interior_point_properties(case, topo, chord, oracle):
    """More edges never fit worse than the Chow-Liu tree; every information is PD; the tree part of a Subgraph pattern
    is the Chow-Liu tree; blankets whose pattern is a tree anyway take the closed form (identical results)."""
    g, which, opts, *_ = util.load_golden(case)
    d = opts.pose_dim
    batch, roots = util.first_round_batch(g, which, _opts(d, topo, chord))
    ref = abi.marginalize_batch(oracle, None, _opts(d, topo, chord), batch)
    tree = abi.marginalize_batch(oracle, None, _opts(d, abi.TOPO_TREE), batch)
    assert (ref["status"] == 0).all() and (tree["status"] == 0).all()
    ne, nt = np.diff(ref["new_edge_off"]), np.diff(tree["new_edge_off"])
    more = ne > nt
    assert more.sum() >= 5
    assert (ref["kld"][more] <= tree["kld"][more] + 1e-9).all() and (ref["kld"][more] >= -1e-9).all()
    same = ~more
    assert np.allclose(ref["kld"][same], tree["kld"][same], rtol=0, atol=1e-12, equal_nan=True)   # (k < 2: no KLD)
    for b in np.nonzero(more)[0][:20]:
        blocks = _blocks(d, ref, b)
        for _, X in blocks:
            assert np.linalg.eigvalsh(X).min() > 0
        if topo == abi.TOPO_SUBGRAPH:
            tpairs = [p for p, _ in _blocks(d, tree, b)]
            got = [p for p, _ in blocks]
            full = len(got) == len(set(sum(got, ()))) * (len(set(sum(got, ()))) - 1) // 2    # all pairs: listed in (i, j) order
            assert set(tpairs) <= set(got) and (full or got[:len(tpairs)] == tpairs)
    print(f"{case} topo={topo}: {more.sum()} blankets through the interior point, KLD {ref['kld'][more].mean():.3g} vs tree {tree['kld'][more].mean():.3g}, "
          f"Newton steps {np.mean(ref['info'][more] >> 8):.0f}")


def test_oracle_interior_point_finds_the_minimiser(oracle):
    """Independent check of the optimum: minimise KLD(X) = 1/2 (tr(S M) - log det M - log det S - r), M = U^T J^T X J U,
    over X_e = L_e L_e^T with scipy (BFGS on the Cholesky parameters: no barrier, no Newton, none of the oracle's code)
    from the oracle's Lambda_t and new-edge Jacobians (central differences of the measurement function). The
    barrier leaves the interior-point value at most ~ rho_final * dim above the optimum."""
    from scipy.optimize import minimize
    g, which, opts, *_ = util.load_golden("manhattan_nfr_tree")
    d = 3
    o = _opts(d, abi.TOPO_DENSE)
    batch, roots = util.first_round_batch(g, which, o)
    ref = abi.marginalize_batch(oracle, None, o, batch)
    ne = np.diff(ref["new_edge_off"])
    done = 0
    for b in np.nonzero(ne >= 3)[0][:4]:
        v0, v1 = batch["vert_off"][b], batch["vert_off"][b + 1]
        m = batch["n_remove"][b]
        poses = np.asarray(batch["pose"]).reshape(-1, 3)[v0:v1]
        k = (v1 - v0) - m
        n = d * k
        T = ref["target_info"][ref["target_info_off"][b]:ref["target_info_off"][b] + n * n].reshape(n, n)
        w, V = np.linalg.eigh(T)
        assert (w < 1e-5).sum() <= d
        S, U = 1.0 / w[d:], V[:, d:]

        def between(xa, xb):
            c, s = np.cos(xa[2]), np.sin(xa[2])
            dx = xb[:2] - xa[:2]
            return np.array([c * dx[0] + s * dx[1], -s * dx[0] + c * dx[1], xb[2] - xa[2]])

        local = {int(v): i for i, v in enumerate(np.asarray(batch["vert_id"])[v0:v1])}    # new_edge_vert holds original ids
        blocks = [((local[a], local[c]), X) for (a, c), X in _blocks(d, ref, b)]
        Js = []
        for (a, c), _ in blocks:
            J = np.zeros((d, n))
            for col in range(2 * d):
                vtx, comp = (a, col) if col < d else (c, col - d)
                P1, P2 = poses.copy(), poses.copy()
                P1[vtx, comp] += 1e-6; P2[vtx, comp] -= 1e-6
                J[:, (vtx - m) * d + comp] = (between(P1[a], P1[c]) - between(P2[a], P2[c])) / 2e-6
            Js.append(J @ U)
        E = len(Js)
        tri = np.tril_indices(d)

        def f(p):
            M = np.zeros((n - d, n - d))
            for e in range(E):
                L = np.zeros((d, d))
                L[tri] = p[e * len(tri[0]):(e + 1) * len(tri[0])]
                M += Js[e].T @ (L @ L.T) @ Js[e]
            sign, ld = np.linalg.slogdet(M)
            if sign <= 0:
                return 1e30
            return 0.5 * (np.sum(np.diag(M) * S) - ld - np.log(S).sum() - (n - d))

        p0 = np.concatenate([np.eye(d)[tri]] * E)
        best = minimize(f, p0, method="BFGS", options={"gtol": 1e-9, "maxiter": 4000})
        assert best.fun <= ref["kld"][b] + 1e-7, (best.fun, ref["kld"][b])
        assert ref["kld"][b] - best.fun <= 5e-6, (best.fun, ref["kld"][b])        # barrier bias: rho_final * d * E ~ 5e-7
        done += 1
    assert done >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("case,topo,chord", [("sphere_nfr_tree", abi.TOPO_DENSE, 1.0), ("manhattan_nfr_tree", abi.TOPO_DENSE, 1.0),
                                             ("manhattan_nfr_tree", abi.TOPO_SUBGRAPH, 0.5), ("intel_nfr_tree_sp3", abi.TOPO_SUBGRAPH, 0.4),
                                             ("parking_nfr_tree", abi.TOPO_SUBGRAPH, 0.3)])
def test_device_interior_point_matches_oracle(case, topo, chord, hip_ctx, oracle):
    """First-round blankets: same patterns, statuses and Newton-step counts (exactly); informations and KLD to 1e-9 of their
    scale (the last barrier problem is solved to 1e-12 by both; the path there goes through ~50 tolerance-terminated Newton
    runs, which the device reproduces step for step)."""
    g, which, opts, *_ = util.load_golden(case)
    d = opts.pose_dim
    o = _opts(d, topo, chord)
    batch, roots = util.first_round_batch(g, which, o)
    ref = abi.marginalize_batch(oracle, None, o, batch)
    got = hip_ctx.marginalize_batch(o, batch)
    assert np.array_equal(ref["status"], got["status"]) and (ref["status"] == 0).sum() > 10
    assert np.array_equal(ref["new_edge_off"], got["new_edge_off"])
    assert np.array_equal(ref["new_edge_vert"], got["new_edge_vert"])
    ne = np.diff(ref["new_edge_off"])
    k = np.diff(batch["vert_off"]) - batch["n_remove"]
    ip = ne > np.maximum(k - 1, 0)
    assert ip.sum() >= 5
    steps_r, steps_g = ref["info"][ip] >> 8, got["info"][ip] >> 8
    worst = 0.0
    for b in np.nonzero(ip)[0]:
        for (pr, Xr), (pg, Xg) in zip(_blocks(d, ref, b), _blocks(d, got, b)):
            assert pr == pg
            worst = max(worst, np.abs(Xr - Xg).max() / np.abs(Xr).max())
    kerr = np.max(np.abs(ref["kld"][ip] - got["kld"][ip]))
    print(f"{case} topo={topo}: {ip.sum()} interior-point blankets, worst information rel err {worst:.1e}, worst KLD abs err {kerr:.1e}, "
          f"Newton steps oracle {steps_r.mean():.1f} device {steps_g.mean():.1f}")
    assert worst <= 1e-9 and kerr <= 1e-9      # (the north star's bar; measured ~5e-12)
    # the target information of the batch interface comes from the generic kernel too
    for b in np.nonzero(ip)[0]:
        lo, hi = ref["target_info_off"][b], ref["target_info_off"][b + 1]
        assert util.rel_err(ref["target_info"][lo:hi], got["target_info"][lo:hi]) <= 1e-9
    assert np.array_equal(steps_r, steps_g)    # where a tolerance-terminated Newton run stops is part of the result
    assert np.array_equal(ref["info"][ip] & abi.INFO_IP_HESSIAN_NOT_PD, got["info"][ip] & abi.INFO_IP_HESSIAN_NOT_PD)   # a given-up barrier step is reported alike
    fin = ~ip & np.isfinite(ref["kld"])
    assert util.rel_err(ref["kld"][fin], got["kld"][fin]) <= 1e-9 or np.abs(ref["kld"][fin] - got["kld"][fin]).max() <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("case,k_range", [("parking_full_nfr_tree", (5, 7)), ("manhattan_full_glc_tree", (8, 10))])
def test_device_interior_point_streamed_sizes_match_oracle(case, k_range, hip_ctx, oracle):
    """Barrier problems of 300 to 800 variables (Dense pattern on blankets of 5-7 SE3 / 8-10 SE2 kept vertices: d^2 k (k-1)/2
    unknowns) — the sizes whose Newton systems take the streamed tiled Cholesky out of the L2 workspace (csrc/spg_nfr_ip.hip,
    path B'), which the bit-identity test of the factorisation paths only reaches at 252 variables. Against the oracle, per
    blanket: informations and KLD to the north star's 1e-9, Newton-step counts equal."""
    g, which, opts, *_ = util.load_golden(case)
    d = opts.pose_dim
    o = _opts(d, abi.TOPO_DENSE, 1.0)
    batch, roots = util.first_round_batch(g, which, o, limit=6, k_range=k_range)
    k = np.diff(batch["vert_off"]) - batch["n_remove"]
    nvar = d * d * k * (k - 1) // 2
    assert len(roots) >= 3 and nvar.min() >= 250 and nvar.max() <= 895 and nvar.max() >= 300
    ref = abi.marginalize_batch(oracle, None, o, batch)
    got = hip_ctx.marginalize_batch(o, batch)
    assert np.array_equal(ref["status"], got["status"]) and (ref["status"] == 0).all()
    assert np.array_equal(ref["new_edge_off"], got["new_edge_off"]) and np.array_equal(ref["new_edge_vert"], got["new_edge_vert"])
    worst = 0.0
    for b in range(len(roots)):
        for (pr, Xr), (pg, Xg) in zip(_blocks(d, ref, b), _blocks(d, got, b)):
            assert pr == pg
            worst = max(worst, np.abs(Xr - Xg).max() / np.abs(Xr).max())
    kerr = np.max(np.abs(ref["kld"] - got["kld"]))
    print(f"{case}: {len(roots)} blankets, {nvar.min()}-{nvar.max()} variables, worst information rel err {worst:.1e}, worst KLD abs err {kerr:.1e}, "
          f"Newton steps {(ref['info'] >> 8).tolist()}")
    assert worst <= util.RTOL and kerr <= 1e-9
    assert np.array_equal(ref["info"] >> 8, got["info"] >> 8)
    assert np.array_equal(ref["info"] & abi.INFO_IP_HESSIAN_NOT_PD, got["info"] & abi.INFO_IP_HESSIAN_NOT_PD)


@pytest.mark.gpu
@pytest.mark.parametrize("case,n,chord", [("sphere_nfr_tree", 300, 0.5), ("manhattan_nfr_tree", 400, 0.34), ("intel_nfr_tree_sp3", 300, 0.4),
                                          ("parking_nfr_tree", 250, 0.2)])
def test_device_interior_point_whole_graph(case, n, chord, hip_ctx):
    """Subgraph NFR on fixture prefixes through the round scheduler (the new edges of one blanket feed the next; blankets
    reach k = 16, 20 new edges): topology identical to the sequential oracle, every blanket's KLD and Newton-step count
    equal, payload to 1e-7."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    g, which, opts, *_ = util.load_golden(case)
    d = opts.pose_dim
    sub, w = util.prefix_graph(g, which, n)
    o = _opts(d, abi.TOPO_SUBGRAPH, chord)
    hg = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx)
    st = hg.marginalizeNoOptimize(w, o)
    og = oracle_lib.OracleGraph.from_dict(sub)
    assert og.marginalize(w, o) == 0
    assert st["n_bad_status"] == 0
    worst = util.compare_edge_sets(d, og.edges(), hg.edges(), rtol=util.RTOL)
    hb, ob = hg.blankets(), og.blankets()
    at = {int(r): i for i, r in enumerate(hb["root"])}
    idx = np.array([at[int(r)] for r in ob["root"]])
    assert np.array_equal(hb["info"][idx] >> 8, ob["info"] >> 8)
    fin = np.isfinite(ob["kld"])
    assert np.abs(hb["kld"][idx][fin] - ob["kld"][fin]).max() <= 1e-9
    n_ip = int(((ob["info"] >> 8) > 0).sum())
    assert n_ip >= 3
    kref = float(np.nansum(ob["kld"]))
    ht = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx)
    stt = ht.marginalizeNoOptimize(w, _opts(d, abi.TOPO_TREE))
    assert st["kld_sum"] < stt["kld_sum"]
    print(f"{case}: Subgraph({chord}) NFR, {n_ip} of {len(ob['root'])} blankets through the interior point (max k {int(ob['k'].max())}), "
          f"kld_sum {st['kld_sum']:.6g} (oracle {kref:.6g}; Tree {stt['kld_sum']:.6g}), worst edge rel err {worst:.1e}")


@pytest.mark.gpu
def test_device_interior_point_dense_whole_graph(hip_ctx):
    """Dense NFR fills the graph in: blankets grow to k = 16 with 120 new edges (a 1080-variable barrier problem whose
    Newton runs end on the reference's stall tests, not at the optimum), so beyond such a blanket device and oracle
    agree in topology only — the same conditioning limit as the Local LM (tests/test_local_conditioning.py). Asserted:
    no bad status, identical topology, at least 90 % of the blankets with equal KLD (1e-9) and equal Newton-step
    counts, a summed KLD far below the tree's."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    g, which, opts, *_ = util.load_golden("manhattan_nfr_tree")
    sub, w = util.prefix_graph(g, which, 500)
    o = _opts(3, abi.TOPO_DENSE)
    hg = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx)
    st = hg.marginalizeNoOptimize(w, o)
    og = oracle_lib.OracleGraph.from_dict(sub)
    assert og.marginalize(w, o) == 0 and st["n_bad_status"] == 0
    ca, cb = util.canonical(og.edges()), util.canonical(hg.edges())
    assert [(k, i) for k, i, _ in ca] == [(k, i) for k, i, _ in cb]
    hb, ob = hg.blankets(), og.blankets()
    at = {int(r): i for i, r in enumerate(hb["root"])}
    idx = np.array([at[int(r)] for r in ob["root"]])
    fin = np.isfinite(ob["kld"])
    same = (np.abs(hb["kld"][idx][fin] - ob["kld"][fin]) <= 1e-9) & ((hb["info"][idx][fin] >> 8) == (ob["info"][fin] >> 8))
    assert same.mean() >= 0.9
    first_bad = int(np.argmin(same)) if not same.all() else -1
    stt = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx).marginalizeNoOptimize(w, _opts(3, abi.TOPO_TREE))
    assert st["kld_sum"] < 0.5 * stt["kld_sum"]
    print(f"manhattan prefix, Dense NFR: {same.sum()} of {fin.sum()} blankets identical (first other: k = {int(ob['k'][fin][first_bad]) if first_bad >= 0 else 0}), "
          f"kld_sum {st['kld_sum']:.6g} (Tree {stt['kld_sum']:.6g})")


@pytest.mark.gpu
def test_factorisation_paths_are_bit_identical(tmp_path):
    """The Newton systems are factorised by one of three paths, chosen by size: register tiles (up to 247 variables), the
    packed Hessian in LDS, panels out of the L2 workspace. They perform the same operations in the same order — the
    records of a whole sphere.g2o run (1 248 blankets, ~75 k Newton steps, 108 to 252 variables) must agree byte for
    byte between the default selection and the column-at-a-time / out-of-L2 paths. The selection is read once per
    process, so each variant runs in a child process (tools/debug/ip_modes.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "debug", "ip_modes.py")
    outs = {}
    for name, env in (("default", {}), ("untiled", {"SPG_IP_UNTILED": "1"})):
        out = str(tmp_path / f"{name}.npz")
        r = subprocess.run([sys.executable, tool, out, "sphere_full_nfr_tree"], env={**os.environ, **env}, cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[name] = np.load(out)
    a, b = outs["default"], outs["untiled"]
    assert set(a.files) == set(b.files)
    assert int((a["info"] >> 8).sum()) > 50000
    for k in a.files:
        assert a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), k


def _hub_blanket_against(ref_edges, ref_b, k, hip_ctx):
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    from tests.test_big_blankets import _star_graph
    g = _star_graph(k, seed=5)
    hg = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    st = hg.marginalizeNoOptimize(np.array([0], np.int32), abi.make_options(6, abi.ALG_NFR, abi.TOPO_DENSE))
    hb = hg.blankets()
    assert st["n_bad_status"] == 0 and np.array_equal(hb["status"], ref_b["status"])
    worst = util.compare_edge_sets(6, ref_edges, hg.edges())            # 1e-9
    kerr = float(np.abs(hb["kld"] - ref_b["kld"]).max())
    print(f"hub with {k} neighbours under Dense: {36 * k * (k - 1) // 2} variables, Newton steps {hb['info'] >> 8} (oracle {ref_b['info'] >> 8}), "
          f"KLD {hb['kld'][0]:.9g}, worst edge rel err {worst:.2e}, KLD abs err {kerr:.1e}")
    assert np.array_equal(hb["info"], ref_b["info"])                    # Newton-step count and flags
    assert kerr <= 1e-9
    return hg


@pytest.mark.gpu
def test_device_interior_point_blocked_factorisation_matches_oracle(hip_ctx):
    """Path (C) of the interior point — Newton systems beyond the streamed sizes, factorised in 64-wide block columns on the
    matrix cores — against the oracle run here: the hub of an SE3 hub graph with 8 neighbours under Dense, 28 new edges =
    1 008 variables (the oracle needs ~15 s for it). Informations and KLD to 1e-9, the Newton-step count equal."""
    from tests.test_big_blankets import _star_graph
    og = oracle_lib.OracleGraph.from_dict(_star_graph(8, seed=5))
    assert og.marginalize(np.array([0], np.int32), abi.make_options(6, abi.ALG_NFR, abi.TOPO_DENSE)) == 0
    _hub_blanket_against(og.edges(), og.blankets(), 8, hip_ctx)


@pytest.mark.gpu
def test_device_interior_point_beyond_2048_variables(hip_ctx):
    """The size round 2 refused with SPG_ECAPACITY (parking.g2o under Dense fails there at k = 12): 12 SE3 neighbours, 66 new
    edges = 2 376 variables, a 2 376^2 Newton system per step. Against the committed oracle result (the oracle needs three
    minutes for this blanket: tests/golden/make_dense_ip_golden.py): 79 Newton steps on both sides, informations and KLD 1e-9."""
    import os
    z = np.load(os.path.join(util.GOLDEN_DIR, "digest_hub_dense_ip.npz"))
    ref_e = {k2: z[k2] for k2 in ("kind", "vert_off", "vert_ids", "data_off", "data")}
    ref_b = {"status": z["status"], "kld": z["kld"], "info": z["info"]}
    _hub_blanket_against(ref_e, ref_b, 12, hip_ctx)
