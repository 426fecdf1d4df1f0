"""Replay harness of the reference's evaluation loop (src/evaluate.cpp:32-195): feed a pose graph
vertex by vertex into an incremental and a baseline graph, decimate (online / cluster / global),
replace edges to already marginalised vertices by computeSubstituteEdge, optimise, marginalise,
and record the global KLD of the sparsified graph against the baseline every `kldPeriod` vertices.

The loop is written against the GraphWrapper interface (addVertex, addEdge, optimize, marginalize,
kullbackLeibler), so the same code drives the product (GraphWrapperHIP) and, in the tests, the oracle.
What the reference's harness does around it — job files, result directories, `.kld` / `.txt` files,
threads (src/main.cpp, src/evaluate.cpp:197-460) — is outside the scope of this repository.
"""
import numpy as np

from . import abi


class EvaluateInfo:
    """The fields of the reference's EvaluateInfo that the loop reads (src/evaluate.h:14-40)."""
    NFR, GLC, NoSparsification = "nfr", "glc", "none"

    def __init__(self, decimate, decimateOptions, sparsityOptions, algorithm="nfr", kldPeriod=10, useChi2=False):
        self.decimate = decimate
        self.decimateOptions = decimateOptions
        self.sparsityOptions = sparsityOptions
        self.algorithm = algorithm
        self.kldPeriod = int(kldPeriod)
        self.useChi2 = bool(useChi2)   # record baseline.chi2(incremental) - baseline.chi2() instead of the KLD


def _info_matrix(d, upper):
    M = np.zeros((d, d))
    M[np.triu_indices(d)] = upper
    return M + M.T - np.diag(np.diag(M))


def evaluate(g, info, make_graph, substitute_source):
    """g: graph dict (ids 0..last, poses, edge_ij, edge_data). make_graph(useGLC) -> empty GraphWrapper.
    substitute_source: a GraphWrapper holding the FULL graph g (computeSubstituteEdge walks it).
    Returns (kld_series [(i, kld)], incremental, baseline)."""
    d = g["pose_dim"]
    ps = abi.pose_stride(d)
    ids = [int(i) for i in g["ids"]]
    pose = {i: np.asarray(p, float) for i, p in zip(ids, g["poses"])}
    last = ids[-1]
    by_vertex = {i: [] for i in ids}
    for e, (a, b) in enumerate(g["edge_ij"]):
        by_vertex[int(a)].append(e)
        if int(b) != int(a):
            by_vertex[int(b)].append(e)
    use_glc = info.algorithm == EvaluateInfo.GLC
    incremental, baseline = make_graph(use_glc), make_graph(False)
    # clonePortion(3): vertices 0..3, the edges among them, optimised (src/graph_wrapper_g2o.cpp:329-349)
    for gw in (incremental, baseline):
        for i in ids:
            if i <= 3:
                gw.addVertex(i, pose[i])
        for (a, b), rec in zip(g["edge_ij"], g["edge_data"]):
            if int(a) <= 3 and int(b) <= 3:
                gw.addEdge(int(a), int(b), rec[:ps], _info_matrix(d, rec[ps:]))
        gw.optimize()
    marginalized = set()
    series = []
    for i in range(4, last + 1):
        for gw in (incremental, baseline):
            gw.addVertex(i, pose[i])
        for e in by_vertex[i]:
            frm, to = (int(x) for x in g["edge_ij"][e])
            if frm > i or to > i:
                continue
            linkto = to if frm == i else frm
            rec = g["edge_data"][e]
            if linkto in marginalized:
                frm, to, meas, up = substitute_source.computeSubstituteEdge(marginalized, i, frm, to)
                infom = _info_matrix(d, up)
            else:
                meas, infom = rec[:ps], _info_matrix(d, rec[ps:])
            incremental.addEdge(frm, to, meas, infom)
            baseline.addEdge(frm, to, meas, infom)
        which = info.decimate(i, last, info.decimateOptions)
        sparsify = info.algorithm != EvaluateInfo.NoSparsification
        if sparsify and (which or i % info.kldPeriod == 0 or i == last):
            incremental.optimize()
            baseline.optimize()
        if which and sparsify:
            incremental.marginalize(which, info.sparsityOptions)
        marginalized.update(which)
        if i % info.kldPeriod == 0 or i == last:
            if not sparsify:
                baseline.optimize()
                series.append((i, float(baseline.chi2()) if info.useChi2 else 0.0))
            elif info.useChi2:
                series.append((i, float(baseline.chi2(incremental) - baseline.chi2())))
            else:
                series.append((i, float(baseline.kullbackLeibler(incremental))))
    return series, incremental, baseline
