"""CPU: the symbolic phase of the block-sparse solver (csrc/spg_sparse_plan.hpp through spg_sparse_plan).
A numpy multifrontal Cholesky driven ONLY by the plan's arrays (order, supernodes, boundary rows, child -> parent
maps, the padded front layout) must reproduce the dense factorisation of a random SPD block matrix with the given
pattern: log det, a solve, the log det of the marginal when marginalised blocks go first, and the selected inverse
by the recurrence the device code uses. That pins every structure the device kernels index with."""
import numpy as np
import pytest

from sparsifyposegraph_amd import lib


def lattice(R, Cc, extra=0, seed=0):
    """R x Cc lattice in id order (i, i+1) and (i, i+Cc), plus `extra` random chords; returns CSR adjacency."""
    n = R * Cc
    pairs = set()
    for i in range(n):
        if i + 1 < n:
            pairs.add((i, i + 1))
        if i + Cc < n:
            pairs.add((i, i + Cc))
    rng = np.random.default_rng(seed)
    for _ in range(extra):
        a, b = rng.integers(0, n, 2)
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    adj = [[] for _ in range(n)]
    for a, b in pairs:
        adj[a].append(b)
        adj[b].append(a)
    ptr = np.zeros(n + 1, np.int32)
    for i in range(n):
        ptr[i + 1] = ptr[i] + len(adj[i])
    return ptr, np.array([u for a in adj for u in sorted(a)], np.int32), sorted(pairs)


def random_spd(n, D, pairs, seed):
    rng = np.random.default_rng(seed)
    H = np.zeros((n * D, n * D))
    for a, b in pairs:                       # J^T J of a random "edge" on blocks a, b
        J = rng.standard_normal((D, 2 * D))
        M = J.T @ J
        ia, ib = slice(a * D, a * D + D), slice(b * D, b * D + D)
        H[ia, ia] += M[:D, :D]; H[ib, ib] += M[D:, D:]
        H[ia, ib] += M[:D, D:]; H[ib, ia] += M[D:, :D]
    H += 0.5 * np.eye(n * D)
    return H


def multifrontal(plan, H, D, selinv_from=None):
    """Numpy restatement of csrc/spg_sparse.inc on the plan's arrays. Returns per-supernode (L11, L21), logdiag sums
    per supernode and, if selinv_from is not None, the full Sigma fronts of supernodes >= selinv_from."""
    perm, first, parent, rowptr, rows, rel = (plan[k] for k in ("perm", "first", "parent", "rowptr", "rows", "rel"))
    nsn = len(first) - 1
    pad = lambda x: (x + 63) // 64 * 64  # noqa: E731
    NP = [pad(D * (first[s + 1] - first[s])) for s in range(nsn)]
    NB = [pad(D * (rowptr[s + 1] - rowptr[s])) for s in range(nsn)]
    iperm = np.empty_like(perm)
    iperm[perm] = np.arange(len(perm))
    fronts, children = [], [[] for _ in range(nsn)]
    for s in range(nsn):
        if parent[s] >= 0:
            children[parent[s]].append(s)
    logdiag = np.zeros(nsn)
    for s in range(nsn):
        ld = NP[s] + NB[s]
        F = np.zeros((ld, ld))
        cols = list(range(first[s], first[s + 1]))
        brow = list(rows[rowptr[s]:rowptr[s + 1]])
        loc = {q: D * i for i, q in enumerate(cols)}
        loc.update({q: NP[s] + D * i for i, q in enumerate(brow)})
        for q in cols:                       # original entries of the supernode's columns (lower part)
            v = perm[q]
            for q2, o2 in loc.items():
                if q2 < q:
                    continue
                u = perm[q2]
                F[o2:o2 + D, loc[q]:loc[q] + D] = H[u * D:u * D + D, v * D:v * D + D]
        ns = D * len(cols)
        F[np.arange(ns, NP[s]), np.arange(ns, NP[s])] = 1.0
        for c in children[s]:                # extend-add through rel
            U = fronts[c][NP[c]:, NP[c]:]
            r = rel[rowptr[c]:rowptr[c + 1]]
            assert (r >= 0).all()
            for t1 in range(len(r)):
                for t2 in range(t1 + 1):
                    F[r[t1]:r[t1] + D, r[t2]:r[t2] + D] += U[D * t1:D * t1 + D, D * t2:D * t2 + D]
        F = np.tril(F) + np.tril(F, -1).T
        L11 = np.linalg.cholesky(F[:NP[s], :NP[s]])
        L21 = np.linalg.solve(L11, F[:NP[s], NP[s]:]).T
        F[:NP[s], :NP[s]] = L11
        F[NP[s]:, :NP[s]] = L21
        F[NP[s]:, NP[s]:] -= L21 @ L21.T
        logdiag[s] = np.log(np.diag(L11)[:ns]).sum()
        fronts.append(F)
    sig = {}
    if selinv_from is not None:
        for s in range(nsn - 1, selinv_from - 1, -1):
            F = fronts[s]
            np_, nb_ = NP[s], NB[s]
            S = np.zeros_like(F)
            if parent[s] >= 0:
                Pn = sig[parent[s]]
                r = rel[rowptr[s]:rowptr[s + 1]]
                idx = np.concatenate([np.arange(x, x + D) for x in r]) if len(r) else np.zeros(0, int)
                S[np_:np_ + len(idx), np_:np_ + len(idx)] = Pn[np.ix_(idx, idx)]
            L11, L21 = F[:np_, :np_], F[np_:, :np_]
            Z = np.linalg.inv(np.tril(L11))
            Y = L21 @ Z
            Ssb = -Y.T @ S[np_:, np_:]
            S[:np_, np_:] = Ssb
            S[np_:, :np_] = Ssb.T
            S[:np_, :np_] = Z.T @ Z - Ssb @ Y
            sig[s] = S
    return fronts, logdiag, NP, NB, iperm, sig


def check_plan_basics(plan, n, ptr, adj):
    perm, first, parent, level = plan["perm"], plan["first"], plan["parent"], plan["level"]
    assert sorted(perm) == list(range(n))
    assert first[0] == 0 and first[-1] == n and (np.diff(first) > 0).all()
    nsn = len(first) - 1
    sn_of = np.repeat(np.arange(nsn), np.diff(first))
    iperm = np.empty(n, int)
    iperm[perm] = np.arange(n)
    for s in range(nsn):
        r = plan["rows"][plan["rowptr"][s]:plan["rowptr"][s + 1]]
        assert (np.diff(r) > 0).all() and (len(r) == 0 or r[0] >= first[s + 1])
        if len(r):
            assert parent[s] == sn_of[r[0]] and level[parent[s]] > level[s]
        else:
            assert parent[s] == -1
    # every edge of the graph is inside some front: row position in the boundary of the column's supernode
    for v in range(n):
        for u in adj[ptr[v]:ptr[v + 1]]:
            a, b = sorted((iperm[v], iperm[u]))
            s = sn_of[a]
            assert b < first[s + 1] or b in plan["rows"][plan["rowptr"][s]:plan["rowptr"][s + 1]]


@pytest.mark.parametrize("R,Cc,D,extra,leaf", [(12, 14, 3, 0, 8), (10, 10, 6, 15, 6), (1, 40, 3, 0, 4), (6, 7, 6, 0, 100)])
def test_multifrontal_on_plan_matches_dense(R, Cc, D, extra, leaf):
    ptr, adj, pairs = lattice(R, Cc, extra, seed=R)
    n = R * Cc
    plan = lib.sparse_plan(ptr, adj, D, leaf=leaf)
    check_plan_basics(plan, n, ptr, adj)
    H = random_spd(n, D, pairs, seed=3)
    fronts, logdiag, NP, NB, iperm, _ = multifrontal(plan, H, D)
    sign, ref = np.linalg.slogdet(H)
    assert sign > 0 and abs(2 * logdiag.sum() - ref) <= 1e-9 * abs(ref)
    # solve through the fronts, as sp_forward_kernel / sp_backward_kernel do
    first, rowptr, rows, rel, parent = (plan[k] for k in ("first", "rowptr", "rows", "rel", "parent"))
    nsn = len(first) - 1
    b = np.random.default_rng(5).standard_normal(n * D)
    x = np.zeros(n * D)
    for q in range(n):
        x[q * D:q * D + D] = b[plan["perm"][q] * D:plan["perm"][q] * D + D]
    upd = [None] * nsn
    for s in range(nsn):
        ns = D * (first[s + 1] - first[s])
        r = np.zeros(NP[s] + NB[s])
        r[:ns] = x[D * first[s]:D * first[s] + ns]
        for c in range(s):
            if parent[c] == s:
                rc = rel[rowptr[c]:rowptr[c + 1]]
                for t, o in enumerate(rc):
                    r[o:o + D] += upd[c][D * t:D * t + D]
        F = fronts[s]
        y = np.linalg.solve(np.tril(F[:NP[s], :NP[s]]), r[:NP[s]])
        upd[s] = r[NP[s]:] - F[NP[s]:, :NP[s]] @ y
        x[D * first[s]:D * first[s] + ns] = y[:ns]
    for s in range(nsn - 1, -1, -1):
        ns = D * (first[s + 1] - first[s])
        F = fronts[s]
        xb = np.zeros(NB[s])
        for t, q in enumerate(rows[rowptr[s]:rowptr[s + 1]]):
            xb[D * t:D * t + D] = x[D * q:D * q + D]
        y = np.zeros(NP[s])
        y[:ns] = x[D * first[s]:D * first[s] + ns]
        sol = np.linalg.solve(np.tril(F[:NP[s], :NP[s]]).T, y - F[NP[s]:, :NP[s]].T @ xb)
        x[D * first[s]:D * first[s] + ns] = sol[:ns]
    xs = np.zeros(n * D)
    for q in range(n):
        xs[plan["perm"][q] * D:plan["perm"][q] * D + D] = x[q * D:q * D + D]
    ref = np.linalg.solve(H, b)
    assert np.abs(xs - ref).max() <= 1e-9 * np.abs(ref).max()


@pytest.mark.parametrize("R,Cc,D,leaf", [(10, 12, 3, 6), (8, 8, 6, 5), (4, 30, 3, 1000)])
def test_marginalised_first_and_selected_inverse(R, Cc, D, leaf):
    ptr, adj, pairs = lattice(R, Cc, 5, seed=7)
    n = R * Cc
    is_marg = np.array([1 if (i % 2 == 1 and i >= 4) else 0 for i in range(n)], np.uint8)
    plan = lib.sparse_plan(ptr, adj, D, is_marg=is_marg, leaf=leaf)
    check_plan_basics(plan, n, ptr, adj)
    nm = plan["n_marg_supernodes"]
    first, perm = plan["first"], plan["perm"]
    assert all(is_marg[perm[q]] == (1 if q < first[nm] else 0) for q in range(n))
    H = random_spd(n, D, pairs, seed=11)
    fronts, logdiag, NP, NB, iperm, sig = multifrontal(plan, H, D, selinv_from=nm)
    # log det of the marginal information of the kept blocks
    kept = np.array([i for i in range(n) if not is_marg[i]])
    kidx = np.concatenate([np.arange(i * D, i * D + D) for i in kept])
    midx = np.setdiff1d(np.arange(n * D), kidx)
    schur = H[np.ix_(kidx, kidx)] - H[np.ix_(kidx, midx)] @ np.linalg.solve(H[np.ix_(midx, midx)], H[np.ix_(midx, kidx)])
    ref = np.linalg.slogdet(schur)[1]
    assert abs(2 * logdiag[nm:].sum() - ref) <= 1e-9 * abs(ref)
    # selected inverse: every entry held in a kept front equals the entry of the marginal covariance
    Sig = np.linalg.inv(schur)
    pos_in_kept = {int(b): i for i, b in enumerate(kept)}
    rowptr, rows = plan["rowptr"], plan["rows"]
    worst = 0.0
    for s in range(nm, len(first) - 1):
        cols = list(range(first[s], first[s + 1]))
        loc = [(q, D * i) for i, q in enumerate(cols)] + [(q, NP[s] + D * i) for i, q in enumerate(rows[rowptr[s]:rowptr[s + 1]])]
        for q1, o1 in loc:
            for q2, o2 in loc[:len(cols)]:
                a, b = pos_in_kept[int(perm[q1])], pos_in_kept[int(perm[q2])]
                ref_blk = Sig[a * D:a * D + D, b * D:b * D + D]
                worst = max(worst, np.abs(sig[s][o1:o1 + D, o2:o2 + D] - ref_blk).max())
                worst = max(worst, np.abs(sig[s][o2:o2 + D, o1:o1 + D] - ref_blk.T).max())
    assert worst <= 1e-9 * np.abs(Sig).max()
    # the kept part of the pattern contains the Schur complement's pattern (every kept-kept coupling through
    # marginalised blocks lies inside a front), which is what trace(Sigma Lambda_x) relies on
    nsn = len(first) - 1
    sn_of = np.repeat(np.arange(nsn), np.diff(first))
    nz = np.abs(schur) > 1e-12
    for i, bi in enumerate(kept):
        for j, bj in enumerate(kept[:i]):
            if nz[i * D:i * D + D, j * D:j * D + D].any():
                a, b = sorted((iperm[bi], iperm[bj]))
                s = sn_of[a]
                assert b < first[s + 1] or b in rows[rowptr[s]:rowptr[s + 1]]


def test_plan_of_the_benchmark_lattice_is_small():
    """100 000-block lattice (250 rings x 400): the plan is built in well under a second and its fronts fit HBM."""
    ptr, adj, _ = lattice(250, 400)
    plan = lib.sparse_plan(ptr, adj, 6)
    assert plan["front_bytes"] < 20e9 and plan["n_levels"] < 40
    assert sorted(plan["perm"]) == list(range(100000))


def test_plan_rejects_malformed_row_pointers():
    """spg_sparse_plan reads the caller's CSR through its row pointers: they are validated first (0-based, non-decreasing)."""
    from sparsifyposegraph_amd.lib import SpgError, sparse_plan
    ptr, adj = np.array([0, 1, 2], np.int32), np.array([1, 0], np.int32)
    sparse_plan(ptr, adj, 6)
    for bad in (np.array([1, 1, 2], np.int32), np.array([0, 2, 1], np.int32), np.array([0, 1, -1], np.int32)):
        with pytest.raises(SpgError):
            sparse_plan(bad, adj, 6)
    with pytest.raises(SpgError):
        sparse_plan(ptr, np.array([1, 5], np.int32), 6)
