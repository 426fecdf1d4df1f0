"""GPU: the block-sparse multifrontal path of optimize() and of the global KLD (SURVEY.md 8f.1,
src/graph_wrapper_g2o.cpp:250-269 / :531-548 — CHOLMOD's and SimplicialLLT's role) against the dense path of the
same library, which tests/test_optimize.py and tests/test_gpu_parity.py hold against the oracle. Same algorithm on
top (g2o's LM, the KLD formula), different factorisation underneath: results must agree to 1e-9."""
import os

import numpy as np
import pytest

from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.graph import GraphWrapperHIP
from tests import util
from tests.test_optimize import _perturbed

pytestmark = pytest.mark.gpu


class solver:
    """with solver(ctx, abi.SOLVER_SPARSE, leaf=4): ... — the context's factorisation, restored on exit"""

    def __init__(self, ctx, which, leaf=None):
        self.ctx, self.which, self.leaf = ctx, which, leaf

    def __enter__(self):
        self.ctx.set_linear_solver(self.which)
        if self.leaf:
            os.environ["SPG_SPARSE_LEAF"] = str(self.leaf)

    def __exit__(self, *a):
        self.ctx.set_linear_solver(abi.SOLVER_AUTO)
        os.environ.pop("SPG_SPARSE_LEAF", None)


def _poses_close(d, pa, pb, tol):
    if d == 6:
        sign = np.sign(np.sum(pa[:, 3:] * pb[:, 3:], axis=1))[:, None]
        pa = np.concatenate([pa[:, :3], pa[:, 3:] * sign], axis=1)
    return np.abs(pa - pb).max() <= tol * max(1.0, np.abs(pb).max())


@pytest.mark.parametrize("case,n,glc,leaf", [("intel_nfr_tree_sp3", 300, False, 8), ("sphere_nfr_tree", 200, False, 4),
                                             ("manhattan_glc_tree", 300, True, 16), ("sphere_glc_tree", 150, True, None)])
def test_sparse_lm_matches_dense_lm(case, n, glc, leaf, hip_ctx):
    """Same LM, same inputs, dense vs multifrontal factorisation: identical accept / reject decisions, chi2 and
    estimates to 1e-9 — on the perturbed baseline and on its sparsified graph (binary or n-ary GLC edges)."""
    sub, w, opts = _perturbed(case, n)
    fid = int(sub["ids"][0])
    d = sub["pose_dim"]
    for sparsify in (False, True):
        # 3 iterations: far from convergence, every accept / reject decision is rounding-proof -> strict comparison;
        # 50 iterations: at the fixed point the last trials' gain ratios are rounding noise (as in tests/test_optimize.py),
        # the fixed point itself is not
        for iters, tol in ((3, 1e-9), (50, 1e-7)):
            graphs = []
            for _ in range(2):
                hg = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx, useGLC=glc)
                if sparsify:
                    base, _, _ = _perturbed(case, n, sigma=0.0)
                    hg = GraphWrapperHIP.from_dict(base, ctx=hip_ctx, useGLC=glc)
                    hg.marginalizeNoOptimize(w, opts)
                    keep = {int(i) for i in hg.vertices()[0]}
                    for i, vid in enumerate(sub["ids"]):
                        if int(vid) in keep:
                            hg.setEstimate(int(vid), sub["poses"][i])
                graphs.append(hg)
            with solver(hip_ctx, abi.SOLVER_DENSE):
                ref = graphs[0].optimize(iters, fid)
            with solver(hip_ctx, abi.SOLVER_SPARSE, leaf):
                got = graphs[1].optimize(iters, fid)
            assert ref["solver"] == abi.SOLVER_DENSE and got["solver"] == abi.SOLVER_SPARSE and got["supernodes"] >= 1
            assert got["chi2_initial"] == pytest.approx(ref["chi2_initial"], rel=1e-12)
            if iters == 3:
                assert (got["iterations"], got["trials"]) == (ref["iterations"], ref["trials"])
            assert got["chi2_final"] == pytest.approx(ref["chi2_final"], rel=tol, abs=1e-12)
            assert got["chi2_final"] < got["chi2_initial"]
            (ids_a, pa), (ids_b, pb) = graphs[1].vertices(), graphs[0].vertices()
            assert np.array_equal(ids_a, ids_b) and _poses_close(d, pa, pb, tol)
            print(f"{case} sparsified={sparsify} it={iters}: chi2 {got['chi2_initial']:.6g} -> {got['chi2_final']:.6g}, {got['trials']} solves "
                  f"(dense {ref['trials']}), {got['supernodes']} supernodes, sparse {got['device_seconds'] * 1e3:.1f} ms / dense {ref['device_seconds'] * 1e3:.1f} ms")


def _kld_both(hip_ctx, base, sp, leaf=None, fid=-1):
    with solver(hip_ctx, abi.SOLVER_DENSE):
        base.kullbackLeibler(sp, fid)
        ref = dict(base.last_kld_terms)
    with solver(hip_ctx, abi.SOLVER_SPARSE, leaf):
        base.kullbackLeibler(sp, fid)
        got = dict(base.last_kld_terms)
    assert ref["solver"] == abi.SOLVER_DENSE and got["solver"] == abi.SOLVER_SPARSE
    n = ref["n"]
    assert got["n"] == n and got["n_marginalized"] == ref["n_marginalized"]
    assert abs(got["innerprod"] - ref["innerprod"]) <= 1e-9 * n
    assert abs(got["logdetx"] - ref["logdetx"]) <= 1e-9 * max(abs(ref["logdetx"]), n)
    assert abs(got["logdety"] - ref["logdety"]) <= 1e-9 * max(abs(ref["logdety"]), n)
    assert abs(got["mahalanobis"] - ref["mahalanobis"]) <= 1e-9 * max(abs(ref["mahalanobis"]), 1e-9 * n)
    assert abs(got["kld"] - ref["kld"]) <= 1e-9 * n
    return ref, got


@pytest.mark.parametrize("case,n,leaf", [("sphere_nfr_tree", 300, 4), ("manhattan_glc_tree", 400, 8), ("manhattan_glc_dense", 400, 16),
                                         ("intel_nfr_tree_sp3", 400, 6), ("parking_nfr_tree", 250, 4)])
def test_sparse_kld_matches_dense_kld(case, n, leaf, hip_ctx):
    """Every term of the formula from the multifrontal factorisation with the marginalised vertices first (log det of
    the marginal from the kept supernodes, the trace from the selected inverse) vs the dense Schur complement."""
    g, which, opts, *_ = util.load_golden(case)
    sub, w = util.prefix_graph(g, which, n)
    glc = opts.algorithm == abi.ALG_GLC
    base = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx)
    sp = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx, useGLC=glc)
    sp.marginalizeNoOptimize(w, opts)
    ref, got = _kld_both(hip_ctx, base, sp, leaf)
    print(f"{case}: n={ref['n']} kld dense {ref['kld']:.9g} sparse {got['kld']:.9g}, {got['supernodes']} supernodes")


def test_sparse_kld_mahalanobis_term(hip_ctx):
    g, which, opts, *_ = util.load_golden("sphere_nfr_tree")
    sub, w = util.prefix_graph(g, which, 200)
    base = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx)
    sp = GraphWrapperHIP.from_dict(sub, ctx=hip_ctx)
    sp.marginalizeNoOptimize(w, opts)
    rng = np.random.default_rng(5)
    ids, P = sp.vertices()
    for i in range(3, len(ids), 2):
        p = P[i].copy()
        p[:3] += 0.01 * rng.standard_normal(3)
        q = p[3:] + 0.002 * rng.standard_normal(4)
        p[3:] = q / np.linalg.norm(q)
        sp.setEstimate(int(ids[i]), p)
    ref, got = _kld_both(hip_ctx, base, sp, 4, fid=int(sub["ids"][0]))
    assert ref["mahalanobis"] > 0.1


def test_sparse_paths_on_full_sphere(hip_ctx):
    """BASELINE config 3 at full size (sphere.g2o, 2 500 poses, NFR Tree): VERDICT r1 item 7's bar — the sparse
    optimize() and global KLD within 1e-9 of the dense path."""
    g, which, opts, *_ = util.load_golden("sphere_full_nfr_tree")
    base = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    sp = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    sp.marginalizeNoOptimize(which, opts)
    ref, got = _kld_both(hip_ctx, base, sp)
    print(f"sphere full: kld dense {ref['kld']:.9g} ({ref['device_seconds'] * 1e3:.0f} ms) sparse {got['kld']:.9g} "
          f"({got['device_seconds'] * 1e3:.0f} ms, {got['supernodes']} supernodes, {got['front_bytes'] / 1e9:.2f} GB, {got['factor_flops'] / 1e9:.1f} GFLOP)")
    a, b = GraphWrapperHIP.from_dict(g, ctx=hip_ctx), GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    with solver(hip_ctx, abi.SOLVER_DENSE):
        r = a.optimize(10)
    with solver(hip_ctx, abi.SOLVER_SPARSE):
        s = b.optimize(10)
    assert (s["iterations"], s["trials"]) == (r["iterations"], r["trials"])
    assert s["chi2_final"] == pytest.approx(r["chi2_final"], rel=1e-9)
    assert _poses_close(6, b.vertices()[1], a.vertices()[1], 1e-9)   # the stored estimates are near the optimum: few, decisive trials
    print(f"sphere full optimize: {s['trials']} solves, sparse {s['device_seconds'] * 1e3:.0f} ms vs dense {r['device_seconds'] * 1e3:.0f} ms")


def test_marginalize_with_optimize_and_kld_at_20k(hip_ctx):
    """Beyond the dense capacity: GraphWrapperG2O::marginalize (= marginalizeNoOptimize + optimize, src/graph_wrapper_g2o.cpp:
    456-463) and kullbackLeibler on a 20 000-pose lattice (120 000 variables: dense would need 115 GB per matrix).
    Checked through properties: the solver reports SPARSE, chi2 does not increase, a graph against itself gives 0,
    the KLD of the sparsified graph is positive and finite."""
    g = g2o_io.synth_sphere(20000, 200)
    which = np.array([i for i in range(4, 20000) if i % 2], np.int32)
    base = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    sp = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    sp.marginalize(which, abi.make_options(6))
    st = sp.last_optimize_stats
    assert st["solver"] == abi.SOLVER_SPARSE and st["chi2_final"] <= st["chi2_initial"] * (1 + 1e-12)
    kself = base.kullbackLeibler(base)
    assert base.last_kld_terms["solver"] == abi.SOLVER_SPARSE and abs(kself) <= 1e-9 * base.last_kld_terms["n"]
    kld = base.kullbackLeibler(sp)
    t = base.last_kld_terms
    assert t["solver"] == abi.SOLVER_SPARSE and np.isfinite(kld) and kld > 0
    print(f"20k lattice: optimize {st['iterations']} it / {st['trials']} solves in {st['device_seconds']:.2f} s; "
          f"KLD {kld:.6g} in {t['device_seconds']:.2f} s ({t['front_bytes'] / 1e9:.1f} GB of fronts, {t['factor_flops'] / 1e12:.2f} TFLOP)")


def test_global_kld_at_headline_size(hip_ctx):
    """BASELINE config 5 (100 000 SE3 poses, 49 998 removals): baseline.kullbackLeibler(sparsified) — 300 006 kept
    variables, where the reference's dense marginal would take 720 GB — through the block-sparse path, checked through
    properties: a graph against itself gives 0 (300 k-variable selected inverse), the sparsified graph's KLD is finite
    and positive, estimates unchanged => no Mahalanobis term, the trace term stays near n (it is exactly n for a
    single tree blanket), and the per-term identity kld = (innerprod + mahalanobis - logdetx - logdety - n) / 2."""
    g = g2o_io.synth_sphere(100000, 400)
    which = np.array([i for i in range(4, 100000) if i % 2], np.int32)
    base = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    sp = GraphWrapperHIP.from_dict(g, ctx=hip_ctx)
    st = sp.marginalizeNoOptimize(which, abi.make_options(6))
    assert st["n_bad_status"] == 0 and st["n_removed"] == len(which)
    kld = base.kullbackLeibler(sp)
    t = dict(base.last_kld_terms)
    assert t["solver"] == abi.SOLVER_SPARSE and t["n"] == 6 * (100000 - len(which) - 1) and t["n_marginalized"] == 6 * len(which)
    assert np.isfinite(kld) and kld > 0 and abs(t["mahalanobis"]) <= 1e-12
    assert kld == pytest.approx(0.5 * (t["innerprod"] + t["mahalanobis"] - t["logdetx"] - t["logdety"] - t["n"]), rel=1e-12)
    assert abs(t["innerprod"] / t["n"] - 1) < 0.5
    kself = sp.kullbackLeibler(sp)
    ts = sp.last_kld_terms
    assert ts["solver"] == abi.SOLVER_SPARSE and abs(kself) <= 1e-9 * ts["n"] and abs(ts["innerprod"] - ts["n"]) <= 1e-9 * ts["n"]
    print(f"100k poses: global KLD {kld:.6g} (per-blanket sum {st['kld_sum']:.6g}) in {t['device_seconds']:.2f} s, {t['front_bytes'] / 1e9:.1f} GB of fronts, "
          f"{t['factor_flops'] / 1e12:.2f} TFLOP; self-KLD {kself:.2e} in {ts['device_seconds']:.2f} s")
