"""CPU: pins the oracle's building blocks against independent implementations (numpy/LAPACK,
central differences) and against the invariants SURVEY.md §8c lists. The reference owns no golden
vectors for this path ("parity unpinned"), so these are what stands behind the oracle."""
import ctypes as C

import numpy as np
import pytest

from sparsifyposegraph_amd import abi, g2o_io
from tests import oracle_lib, util

f64p = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(f64p)


def rand_spd(rng, n, cond=1e3):
    q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    w = np.exp(rng.uniform(0, np.log(cond), n))
    return (q * w) @ q.T


@pytest.mark.parametrize("n", [1, 3, 6, 12, 24, 37])
def test_la_against_lapack(oracle, n):
    rng = np.random.default_rng(n)
    A = rand_spd(rng, n)
    L = np.zeros((n, n))
    assert oracle.spgref_chol(n, P(A), P(L)) == 0
    assert np.allclose(L, np.linalg.cholesky(A), rtol=1e-12, atol=1e-12)
    assert abs(oracle.spgref_spd_logdet(n, P(A)) - np.linalg.slogdet(A)[1]) < 1e-10
    w, V = np.zeros(n), np.zeros((n, n))
    assert oracle.spgref_eigh(n, P(A), P(w), P(V)) == 0
    wr = np.linalg.eigvalsh(A)
    assert np.allclose(w, wr, rtol=1e-12)
    assert np.allclose(V @ np.diag(w) @ V.T, A, rtol=1e-11, atol=1e-11)
    assert np.allclose(V.T @ V, np.eye(n), atol=1e-12)
    B = rng.normal(size=(n, n)) + 3 * np.eye(n)
    X = np.zeros((n, n))
    assert oracle.spgref_lu_inverse(n, P(B), P(X)) == 0
    assert np.allclose(X @ B, np.eye(n), atol=1e-10)


def test_eigh_on_singular_gauge_like_matrix(oracle):
    rng = np.random.default_rng(5)
    n, d = 24, 6
    U, _ = np.linalg.qr(rng.normal(size=(n, n)))
    w = np.concatenate([np.zeros(d), np.exp(rng.uniform(0, 6, n - d))])
    A = (U * w) @ U.T
    A = 0.5 * (A + A.T)
    wv, V = np.zeros(n), np.zeros((n, n))
    assert oracle.spgref_eigh(n, P(A), P(wv), P(V)) == 0
    assert np.abs(wv[:d]).max() < 1e-10 * w.max()
    assert np.allclose(wv[d:], np.sort(w[d:]), rtol=1e-11)


def _rq(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def test_se3_jacobians_by_central_differences(oracle):
    rng = np.random.default_rng(0)
    worst = 0
    for trial in range(300):
        xi = np.concatenate([rng.normal(size=3) * 3, _rq(rng)])
        xj = np.concatenate([rng.normal(size=3) * 3, _rq(rng)])
        z = np.concatenate([rng.normal(size=3) * 3, _rq(rng)])
        if trial % 2 == 0:  # zero-error linearisation, the new-edge case
            z = np.zeros(7)
            oracle.spgref_se3_between(P(xi), P(xj), P(z))
        err, Ji, Jj = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
        oracle.spgref_se3_edge(P(xi), P(xj), P(z), P(err), P(Ji), P(Jj))
        if trial % 2 == 0:
            assert np.abs(err).max() < 1e-12
        h, z0 = 1e-6, np.zeros(6)
        Ni, Nj = np.zeros((6, 6)), np.zeros((6, 6))
        for c in range(6):
            d = np.zeros(6)
            d[c] = h
            ep, em = np.zeros(6), np.zeros(6)
            oracle.spgref_se3_error_perturbed(P(xi), P(xj), P(z), P(d), P(z0), P(ep))
            oracle.spgref_se3_error_perturbed(P(xi), P(xj), P(z), P(-d), P(z0), P(em))
            Ni[:, c] = (ep - em) / (2 * h)
            oracle.spgref_se3_error_perturbed(P(xi), P(xj), P(z), P(z0), P(d), P(ep))
            oracle.spgref_se3_error_perturbed(P(xi), P(xj), P(z), P(z0), P(-d), P(em))
            Nj[:, c] = (ep - em) / (2 * h)
        worst = max(worst, np.abs(Ni - Ji).max(), np.abs(Nj - Jj).max())
    assert worst < 5e-8


def test_se2_jacobians_by_central_differences(oracle):
    rng = np.random.default_rng(1)
    for _ in range(100):
        xi, xj, z = rng.normal(size=3) * 2, rng.normal(size=3) * 2, rng.normal(size=3)
        err, Ji, Jj = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 3))
        oracle.spgref_se2_edge(P(xi), P(xj), P(z), P(err), P(Ji), P(Jj))
        h = 1e-6
        for c in range(3):
            d = np.zeros(3)
            d[c] = h
            ep, em = np.zeros(3), np.zeros(3)
            oracle.spgref_se2_edge(P(xi + d), P(xj), P(z), P(ep), None, None)
            oracle.spgref_se2_edge(P(xi - d), P(xj), P(z), P(em), None, None)
            de = ep - em
            de[2] = (de[2] + np.pi) % (2 * np.pi) - np.pi
            assert np.abs(de / (2 * h) - Ji[:, c]).max() < 1e-6
            oracle.spgref_se2_edge(P(xi), P(xj + d), P(z), P(ep), None, None)
            oracle.spgref_se2_edge(P(xi), P(xj - d), P(z), P(em), None, None)
            de = ep - em
            de[2] = (de[2] + np.pi) % (2 * np.pi) - np.pi
            assert np.abs(de / (2 * h) - Jj[:, c]).max() < 1e-6


def test_se2_between_matches_edge_error(oracle):
    rng = np.random.default_rng(2)
    xi, xj = rng.normal(size=3), rng.normal(size=3)
    z = np.zeros(3)
    oracle.spgref_se2_between(P(xi), P(xj), P(z))
    err = np.zeros(3)
    oracle.spgref_se2_edge(P(xi), P(xj), P(z), P(err), None, None)
    assert np.abs(err).max() < 1e-14


# --------------------------------------------------------------------------- blanket level
def numpy_nfr_blanket(d, poses, m, edges, pairs, oracle):
    """Independent restatement of computeTargetInformation + closedFormSolution + value with
    numpy/LAPACK linear algebra (Jacobians through the oracle's unit function, itself pinned above)."""
    nv = len(poses)
    N = nv * d
    ps = abi.pose_stride(d)
    H = np.zeros((N, N))
    iu = np.triu_indices(d)

    def jac(xi, xj, z):
        Ji, Jj = np.zeros((d, d)), np.zeros((d, d))
        if d == 6:
            oracle.spgref_se3_edge(P(xi), P(xj), P(z), None, P(Ji), P(Jj))
        else:
            oracle.spgref_se2_edge(P(xi), P(xj), P(z), None, P(Ji), P(Jj))
        return Ji, Jj
    for (a, b, rec) in edges:
        Om = np.zeros((d, d))
        Om[iu] = rec[ps:]
        Om = Om + Om.T - np.diag(np.diag(Om))
        Ji, Jj = jac(poses[a], poses[b], np.ascontiguousarray(rec[:ps]))
        J = np.zeros((d, N))
        J[:, a * d:(a + 1) * d] = Ji
        J[:, b * d:(b + 1) * d] = Jj
        H += J.T @ Om @ J
    nm = m * d
    T = H[nm:, nm:] - H[:nm, nm:].T @ np.linalg.solve(H[:nm, :nm], H[:nm, nm:])
    T = np.triu(T) + np.triu(T, 1).T
    n = N - nm
    w, V = np.linalg.eigh(T)
    U, S = V[:, d:], 1.0 / w[d:]
    Sigma = (U * S) @ U.T
    Xs, A = [], np.zeros((n, n))
    for (a, b) in pairs:
        z = np.zeros(ps)
        if d == 6:
            oracle.spgref_se3_between(P(poses[m + a]), P(poses[m + b]), P(z))
        else:
            oracle.spgref_se2_between(P(poses[m + a]), P(poses[m + b]), P(z))
        Ja, Jb = jac(poses[m + a], poses[m + b], z)
        J = np.zeros((d, n))
        J[:, a * d:(a + 1) * d] = Ja
        J[:, b * d:(b + 1) * d] = Jb
        X = np.linalg.inv(J @ Sigma @ J.T)
        Xs.append(X)
        A += J.T @ X @ J
    M = U.T @ A @ U
    kld = 0.5 * (np.sum(np.diag(M) * S) - np.linalg.slogdet(M)[1] - np.sum(np.log(S)) - (n - d))
    return T, Xs, kld


@pytest.mark.parametrize("case", ["sphere_nfr_tree", "manhattan_nfr_tree", "parking_nfr_tree"])
def test_nfr_blanket_against_numpy_restatement(case, oracle):
    g, which, opts, _, _, _ = util.load_golden(case)
    d = g["pose_dim"]
    batch, roots = util.first_round_batch(g, which, opts, limit=40)
    ref = abi.marginalize_batch(oracle, None, opts, batch)
    ps, rec = abi.pose_stride(d), abi.binary_record_len(d)
    iu = np.triu_indices(d)
    checked = 0
    for b in range(len(roots)):
        if ref["status"][b] != 0 or ref["info"][b] != 0:
            continue
        v0, v1 = batch["vert_off"][b], batch["vert_off"][b + 1]
        poses = [np.ascontiguousarray(batch["pose"][v * ps:(v + 1) * ps]) for v in range(v0, v1)]
        ids = list(batch["vert_id"][v0:v1])
        edges = []
        for e in range(batch["edge_off"][b], batch["edge_off"][b + 1]):
            a, bb = batch["edge_vert"][2 * e], batch["edge_vert"][2 * e + 1]
            edges.append((a, bb, batch["edge_data"][e * rec:(e + 1) * rec]))
        e0, e1 = ref["new_edge_off"][b], ref["new_edge_off"][b + 1]
        pairs = []
        for e in range(e0, e1):
            va, vb = ref["new_edge_vert"][2 * e], ref["new_edge_vert"][2 * e + 1]
            pairs.append((ids.index(va) - 1, ids.index(vb) - 1))
        if len(pairs) == 0:
            continue
        T, Xs, kld = numpy_nfr_blanket(d, poses, 1, edges, pairs, oracle)
        lo, hi = ref["target_info_off"][b], ref["target_info_off"][b + 1]
        n = int(round(np.sqrt(hi - lo)))
        To = ref["target_info"][lo:hi].reshape(n, n)
        assert util.rel_err(To, T) < 1e-9
        # invariants: symmetric, PSD, exact d-dimensional gauge null space
        assert np.abs(To - To.T).max() == 0
        w = np.linalg.eigvalsh(To)
        assert w[0] > -1e-9 * w[-1] and np.abs(w[:d]).max() < 1e-9 * w[-1] and w[d] > 1e-9 * w[-1]
        for e, X in zip(range(e0, e1), Xs):
            Xo = np.zeros((d, d))
            Xo[iu] = ref["new_edge_data"][e * rec + ps:(e + 1) * rec]
            Xo = Xo + Xo.T - np.diag(np.diag(Xo))
            assert util.rel_err(Xo, X) < 1e-8
            assert np.linalg.eigvalsh(Xo).min() > 0
        assert abs(ref["kld"][b] - kld) < 1e-7 * max(1.0, abs(kld))
        assert ref["kld"][b] > -1e-9
        if len(pairs) == 1:  # k = 2: the single edge reproduces the target exactly
            assert abs(ref["kld"][b]) < 1e-8
        # spanning tree
        assert len(pairs) == (v1 - v0) - 2
        comp = list(range(v1 - v0 - 1))
        for a, bb in pairs:
            ca, cb = comp[a], comp[bb]
            assert ca != cb
            comp = [ca if c == cb else c for c in comp]
        checked += 1
    assert checked >= 20


def test_glc_reparam_jacobian_by_central_differences(oracle):
    rng = np.random.default_rng(3)
    for d in (3, 6):
        ps = abi.pose_stride(d)
        q = 3
        poses = np.zeros((q, ps))
        for i in range(q):
            poses[i, :3 if d == 6 else 2] = rng.normal(size=3 if d == 6 else 2)
            if d == 6:
                poses[i, 3:] = _rq(rng)
            else:
                poses[i, 2] = rng.uniform(-3, 3)
        meas, J = np.zeros(d * q), np.zeros((d * q, d * q))
        oracle.spgref_glc_reparam(d, q, P(poses), None, P(meas), P(J))
        err = np.zeros(d * q)
        oracle.spgref_glc_reparam(d, q, P(poses), P(meas), P(err), P(J))
        assert np.abs(err).max() < 1e-12
        # block structure: first block absolute, others relative to the first
        assert np.abs(J[:d, d:]).max() == 0 and np.abs(J[d:2 * d, 2 * d:]).max() == 0
        if d == 3:
            h = 1e-6
            for v in range(q):
                for c in range(3):
                    pp, pm = poses.copy(), poses.copy()
                    pp[v, c] += h
                    pm[v, c] -= h
                    ep, em = np.zeros(9), np.zeros(9)
                    Jd = np.zeros((9, 9))
                    oracle.spgref_glc_reparam(3, q, P(pp), P(meas), P(ep), P(Jd))
                    oracle.spgref_glc_reparam(3, q, P(pm), P(meas), P(em), P(Jd))
                    assert np.abs((ep - em) / (2 * h) - J[:, v * 3 + c]).max() < 1e-6


# The only numbers the reference ships for this path: the GLC_EDGE line quoted at
# src/test_marginalize_within_window.cpp:198-206 (an edge its own sliding-window run wrote).
REF_GLC_LINE = ("GLC_EDGE 10 16 || GLC_REPARAM_SE2_ISAM 3 6 10 0 0 6 0 0 0 6.44942e-14 7.37069e-14 -5.38607e-12 -0.0781786 -2.09314e-05 "
                "-0 -2.5976e-16 -3.96149e-14 0.0782534 -5.39122e-12 -1.42468e-15 0 5.67138e-15 -1.09704e-14 -1.87717e-15 -0.00209596 7.82839 "
                "1 0 0 1 0 1")
REF_GLC_W = np.array([[0, 6.44942e-14, 7.37069e-14, -5.38607e-12, -0.0781786, -2.09314e-05],
                      [-0, -2.5976e-16, -3.96149e-14, 0.0782534, -5.39122e-12, -1.42468e-15],
                      [0, 5.67138e-15, -1.09704e-14, -1.87717e-15, -0.00209596, 7.82839]])
REF_GLC_MEAS = np.array([10.0, 0, 0, 6, 0, 0])


def test_known_glc_edge_from_reference_comment(oracle):
    """Pins the oracle's GLC machinery to the one edge the reference quotes.
    (1) reparametrize(vertices 10 at (10,0,0), 16 at (16,0,0)) IS the quoted measurement `10 0 0 6 0 0`
        (first pose absolute, second relative to the first: src/glc_reparam_binary.hpp:35-75).
    (2) The quoted W is what glc_chol produces: orthogonal rows sqrt(lambda) v^T in ascending eigenvalue order, no
        information on the absolute block (src/topology_provider_glc.cpp:59-71).
    (3) Fixed point: a graph that holds exactly this GLC edge plus a third vertex tied to 10 and 16 by edges of negligible
        information (1e-12) is sparsified by removing that vertex. The blanket's target is then the quoted edge's own
        information (n-ary edge Jacobian W J_reparam, src/glc_edge.cpp:40-49, Schur, reparametrisation, LU inverse, eig,
        1e-8 cut, sqrt scaling — the whole GLC path), so the oracle must hand back the quoted record: same measurement,
        3 x 6 W equal to the quoted one up to the sign of each row, to the 6 digits the reference printed."""
    poses = np.array([[10.0, 0, 0], [16.0, 0, 0]])
    meas, J = np.zeros(6), np.zeros((6, 6))
    oracle.spgref_glc_reparam(3, 2, P(poses), None, P(meas), P(J))
    assert np.array_equal(meas, REF_GLC_MEAS)
    G = REF_GLC_W.T @ REF_GLC_W
    assert np.abs(G[:3, :]).max() < 1e-10 and np.linalg.matrix_rank(G[3:, 3:], tol=1e-12) == 3
    WWt = REF_GLC_W @ REF_GLC_W.T
    assert np.abs(WWt - np.diag(np.diag(WWt))).max() < 2e-6 * WWt.max()         # rows orthogonal (6 printed digits)
    assert np.all(np.diff(np.diag(WWt)) > 0)                                      # ascending eigenvalues
    og = oracle_lib.OracleGraph(3)
    for vid, p in ((10, [10.0, 0, 0]), (13, [13.0, 0.4, 0.1]), (16, [16.0, 0, 0])):
        og.L.spgref_graph_add_vertex(og.h, vid, P(np.array(p)))
    assert og.add_edge(abi.EDGE_GLC, [10, 16], np.concatenate([REF_GLC_MEAS, REF_GLC_W.ravel()])) == 0
    weak = 1e-12 * np.eye(3)[np.triu_indices(3)]
    for a, b, pa, pb in ((10, 13, [10.0, 0, 0], [13.0, 0.4, 0.1]), (13, 16, [13.0, 0.4, 0.1], [16.0, 0, 0])):
        z = np.zeros(3)
        oracle.spgref_se2_between(P(np.array(pa)), P(np.array(pb)), P(z))
        assert og.add_edge(abi.EDGE_BINARY, [a, b], np.concatenate([z, weak])) == 0
    for topo in (abi.TOPO_DENSE,):
        assert og.marginalize(np.array([13], np.int32), abi.make_options(3, abi.ALG_GLC, topo)) == 0
    (kind, ids, data), = util.edge_list(og.edges())
    assert kind == abi.EDGE_GLC and ids == (10, 16) and len(data) == 6 + 3 * 6
    assert np.abs(data[:6] - REF_GLC_MEAS).max() < 1e-12
    W = data[6:].reshape(3, 6)
    assert np.abs(W[:, :3]).max() < 1e-9
    for r in range(3):
        s = np.sign(W[r] @ REF_GLC_W[r])
        assert np.abs(s * W[r, 3:] - REF_GLC_W[r, 3:]).max() <= 2e-6 * np.abs(REF_GLC_W[r]).max(), (r, W[r], REF_GLC_W[r])


def test_reference_glc_line_through_the_product_reader_and_writer(tmp_path):
    """f3: the product's .g2o reader takes the reference's own GLC_EDGE sample line (src/test_marginalize_within_window.cpp:
    198-206; grammar of GLCEdge::read, src/glc_edge.cpp:65-93) and its writer gives back the same grammar: tag, the `||`
    separator, reparametrisation tag, r, d*q, measurement, W row by row, the upper triangle of the identity."""
    from sparsifyposegraph_amd.graph import GraphWrapperHIP
    path = tmp_path / "ref_line.g2o"
    path.write_text("VERTEX_SE2 10 10 0 0\nVERTEX_SE2 16 16 0 0\n" + REF_GLC_LINE + "\n")
    ctx = oracle_lib.injected_context()
    g = GraphWrapperHIP.load(str(path), ctx=ctx, useGLC=True)
    e = g.edges()
    assert list(e["kind"]) == [abi.EDGE_GLC] and list(e["vert_ids"]) == [10, 16]
    assert np.array_equal(e["data"][:6], REF_GLC_MEAS) and np.array_equal(e["data"][6:].reshape(3, 6), REF_GLC_W)
    tokens = [l for l in g.writeString().splitlines() if l.startswith("GLC_EDGE")][0].split()
    ref = REF_GLC_LINE.split()
    assert tokens[:8] == ref[:8]                                   # GLC_EDGE 10 16 || GLC_REPARAM_SE2_ISAM 3 6 10
    assert [float(t) for t in tokens[7:]] == [float(t) for t in ref[7:]]   # measurement, W, information: the same numbers
    assert tokens[-6:] == ["1", "0", "0", "1", "0", "1"]


def numpy_chow_liu(d, T):
    """pseudo-Chow-Liu tree of src/pseudo_chow_liu.cpp:169-196,253-289 in numpy: Tikhonov inverse, pairwise mutual
    information from log-determinants, Kruskal on descending weight with the (weight, i, j) tie-break."""
    k = T.shape[0] // d
    S = np.linalg.inv(T + np.eye(d * k))
    ld = lambda idx: np.linalg.slogdet(S[np.ix_(idx, idx)])[1]   # noqa: E731
    blk = lambda i: list(range(i * d, (i + 1) * d))               # noqa: E731
    w = sorted(((-(ld(blk(i)) + ld(blk(j)) - ld(blk(i) + blk(j))), i, j) for i in range(k) for j in range(i + 1, k)))
    comp, tree = list(range(k)), []
    for _, i, j in w:
        if comp[i] != comp[j]:
            tree.append((i, j))
            ci, cj = comp[i], comp[j]
            comp = [ci if c == cj else c for c in comp]
    return tree


def numpy_glc_edge(d, T, poses, oracle):
    """TopologyProviderGLC::getEdge + glc_chol (src/topology_provider_glc.cpp:59-98) in numpy: returns (measurement,
    W^T W) — W itself is defined up to the eigenvectors' signs. The reparametrisation is the oracle's unit function,
    pinned above by central differences."""
    q = len(poses)
    pp = np.ascontiguousarray(np.stack(poses))
    meas, J = np.zeros(d * q), np.zeros((d * q, d * q))
    oracle.spgref_glc_reparam(d, q, P(pp), None, P(meas), P(J))
    oracle.spgref_glc_reparam(d, q, P(pp), P(meas), P(np.zeros(d * q)), P(J))
    iJ = np.linalg.inv(J)
    M = iJ.T @ T @ iJ
    M = 0.5 * (M + M.T)
    w, V = np.linalg.eigh(M)
    keep = w >= 1e-8
    return meas, (V[:, keep] * w[keep]) @ V[:, keep].T, int(keep.sum())


def numpy_schur_onto(T, keep):
    """PseudoChowLiu::marginal (src/pseudo_chow_liu.cpp:130-138)"""
    rest = [i for i in range(T.shape[0]) if i not in set(keep)]
    if not rest:
        return T[np.ix_(keep, keep)]
    return T[np.ix_(keep, keep)] - T[np.ix_(keep, rest)] @ np.linalg.solve(T[np.ix_(rest, rest)], T[np.ix_(rest, keep)])


def numpy_posdef_pinv(a):
    """posdef_pinv (src/topology_provider_glc.cpp:42-56)"""
    w, V = np.linalg.eigh(a)
    tol = np.finfo(float).eps * a.shape[0] * np.abs(w).max()
    inv = np.where(w > tol, 1.0 / np.where(w > tol, w, 1.0), 0.0)
    return (V * inv) @ V.T


@pytest.mark.parametrize("case", ["manhattan_glc_tree", "sphere_glc_tree", "intel_glc_tree_10pct"])
def test_glc_tree_blanket_against_numpy_restatement(case, oracle):
    """The GLC Tree tail restated independently in numpy/LAPACK on first-round blankets: the Chow-Liu tree (topology
    bit-identical), the root marginal, the conditional targets [[jaa, jab],[jba, jba pinv(jaa) jab]]
    (src/topology_provider_glc.cpp:134-176) and getEdge; measurements and W^T W of every edge the oracle emits."""
    g, which, opts, _, _, _ = util.load_golden(case)
    d = g["pose_dim"]
    ps = abi.pose_stride(d)
    batch, roots = util.first_round_batch(g, which, opts, limit=40)
    ref = abi.marginalize_batch(oracle, None, opts, batch)
    checked = 0
    for b in range(len(roots)):
        v0, v1 = batch["vert_off"][b], batch["vert_off"][b + 1]
        k = (v1 - v0) - 1
        if ref["status"][b] != 0 or k < 2:
            continue
        ids = list(batch["vert_id"][v0:v1])
        poses = [np.ascontiguousarray(batch["pose"][v * ps:(v + 1) * ps]) for v in range(v0 + 1, v1)]   # kept, ascending id
        lo, hi = ref["target_info_off"][b], ref["target_info_off"][b + 1]
        T = ref["target_info"][lo:hi].reshape(d * k, d * k)
        tree = numpy_chow_liu(d, T)
        expect = []
        root = tree[0][0]
        rblk = list(range(root * d, (root + 1) * d))
        m_, G_, r_ = numpy_glc_edge(d, numpy_schur_onto(T, rblk), [poses[root]], oracle)
        if r_ > 0:
            expect.append(((ids[1 + root],), m_, G_))
        for (i, j) in tree:
            idx = list(range(i * d, (i + 1) * d)) + list(range(j * d, (j + 1) * d))
            jm = numpy_schur_onto(T, idx)
            jaa, jab, jba = jm[:d, :d], jm[:d, d:], jm[d:, :d]
            tgt = np.block([[jaa, jab], [jba, jba @ numpy_posdef_pinv(jaa) @ jab]])
            tgt = np.triu(tgt) + np.triu(tgt, 1).T
            m_, G_, r_ = numpy_glc_edge(d, tgt, [poses[i], poses[j]], oracle)
            if r_ > 0:
                expect.append(((ids[1 + i], ids[1 + j]), m_, G_))
        got = []
        for e in range(ref["new_edge_off"][b], ref["new_edge_off"][b + 1]):
            vs = tuple(int(x) for x in ref["new_edge_vert"][ref["new_edge_vert_off"][e]:ref["new_edge_vert_off"][e + 1]])
            data = ref["new_edge_data"][ref["new_edge_data_off"][e]:ref["new_edge_data_off"][e + 1]]
            n = d * len(vs)
            W = data[n:].reshape(-1, n)
            got.append((vs, data[:n], W.T @ W))
        if ref["min_gap"][b] < 1e-9:
            continue   # a near-tie between two Chow-Liu weights: the tree is not determined at this precision
        assert [v for v, *_ in got] == [v for v, *_ in expect], (b, [v for v, *_ in got], [v for v, *_ in expect])
        for (_, ma, Ga), (_, mb, Gb) in zip(got, expect):
            assert util.rel_err(ma, mb) < 1e-10
            assert util.rel_err(Ga, Gb) < 1e-8
        checked += 1
    assert checked >= 15


def test_local_linearization_point_chain(oracle):
    """src/test_marginalize_se3.cpp:20-48 with fixed numbers: 3-pose SE3 chain, middle pose removed,
    {Local, Tree}. Closed-form local estimate: the removed pose at the origin, neighbours at z^-1 / z,
    so the new edge's measurement is exactly z01 * z12 and (k = 2) the KLD is 0."""
    def q(v):
        v = np.asarray(v, float)
        return v / np.linalg.norm(v)
    z01 = np.concatenate([[0.10, 0.11, 0.12], q([0.05, 0.04, 0.06, 0.99])])
    z12 = np.concatenate([[0.09, 0.13, 0.08], q([0.03, 0.06, 0.02, 0.99])])
    x0 = np.array([0, 0, 0, 0, 0, 0, 1.0])
    x1 = np.concatenate([[0.3, -0.2, 0.1], q([0.1, 0.0, 0.2, 0.97])])   # far from z01: Local must ignore it
    x2 = np.concatenate([[0.7, 0.1, 0.3], q([0.2, 0.1, 0.1, 0.96])])
    info = np.eye(6)[np.triu_indices(6)]
    batch = {"vert_off": np.array([0, 3], np.int32), "n_remove": np.array([1], np.int32), "vert_id": np.array([1, 0, 2], np.int32),
             "pose": np.concatenate([x1, x0, x2]), "edge_off": np.array([0, 2], np.int32), "edge_kind": np.zeros(2, np.int32),
             "edge_vert_off": np.array([0, 2, 4], np.int32), "edge_vert": np.array([1, 0, 0, 2], np.int32),
             "edge_data_off": np.array([0, 28, 56], np.int64), "edge_data": np.concatenate([z01, info, z12, info])}
    opts = abi.make_options(6, abi.ALG_NFR, abi.TOPO_TREE, abi.LIN_LOCAL)
    out = abi.marginalize_batch(oracle, None, opts, batch)
    assert out["status"][0] == 0 and list(out["new_edge_vert"]) == [0, 2]
    z02 = np.zeros(7)
    from sparsifyposegraph_amd.g2o_io import quat_mul, quat_rotate
    t = z01[:3] + quat_rotate(z01[3:], z12[:3])
    qq = quat_mul(z01[3:], z12[3:])
    assert np.allclose(out["new_edge_data"][:3], t, atol=1e-14)
    assert np.allclose(np.abs(out["new_edge_data"][3:7] @ qq), 1.0, atol=1e-14)
    assert abs(out["kld"][0]) < 1e-9
    # Global at the (inconsistent) stored estimates gives a different measurement
    outg = abi.marginalize_batch(oracle, None, abi.make_options(6), batch)
    assert not np.allclose(outg["new_edge_data"][:3], t, atol=1e-3)
    # an extra edge between the two neighbours breaks the closed form: the reference would run LM
    batch2 = dict(batch)
    batch2["edge_off"] = np.array([0, 3], np.int32)
    batch2["edge_kind"] = np.zeros(3, np.int32)
    batch2["edge_vert_off"] = np.array([0, 2, 4, 6], np.int32)
    batch2["edge_vert"] = np.array([1, 0, 0, 2, 1, 2], np.int32)
    batch2["edge_data_off"] = np.array([0, 28, 56, 84], np.int64)
    batch2["edge_data"] = np.concatenate([z01, info, z12, info, z12, info])
    out2 = abi.marginalize_batch(oracle, None, opts, batch2)
    # ... which the oracle restates (10 LM iterations on the blanket, removed vertex fixed): the estimate of
    # vertex 2 is now a compromise between the chain 1-0-2 and the direct edge 1-2, so the recovered
    # measurement 0->2 moves away from the pure composition
    assert out2["status"][0] == 0 and list(out2["new_edge_vert"]) == [0, 2]
    assert not np.allclose(out2["new_edge_data"][:3], t, atol=1e-3)
    # a cluster of two removed vertices under Local (restated since round 3): only the FIRST removed vertex is fixed
    # (src/vertex_remover.cpp:382-391), the other removed vertex moves with the kept one; one kept vertex left: no new edge
    batch3 = dict(batch2)
    batch3["n_remove"] = np.array([2], np.int32)
    out3 = abi.marginalize_batch(oracle, None, opts, batch3)
    assert out3["status"][0] == 0 and len(out3["new_edge_vert"]) == 0
