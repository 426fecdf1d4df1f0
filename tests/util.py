"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

from sparsifyposegraph_amd import abi

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# relative fp64 tolerance the north star states for recovered information matrices and KLD
RTOL = 1e-9
# Chow-Liu topology must be identical whenever consecutive popped weights differ by more than this
GAP_TOL = 1e-10


def golden_cases():
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith("digest_")]   # digests of full-size Dense runs: tests/golden/make_dense_digest.py


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {"pose_dim": int(z["pose_dim"]), "ids": z["ids"], "poses": z["poses"], "edge_ij": z["edge_ij"], "edge_data": z["edge_data"]}
    opts = abi.make_options(g["pose_dim"], int(z["algorithm"]), int(z["topology"]))
    out = {"kind": z["out_kind"], "vert_off": z["out_vert_off"], "vert_ids": z["out_vert_ids"], "data_off": z["out_data_off"], "data": z["out_data"]}
    bl = {k[3:]: z[k] for k in z.files if k.startswith("bl_")}
    return g, z["which"], opts, out, bl, z["out_vertex_ids"]


def edge_list(e):
    out = []
    for i in range(len(e["kind"])):
        ids = tuple(int(x) for x in e["vert_ids"][e["vert_off"][i]:e["vert_off"][i + 1]])
        out.append((int(e["kind"][i]), ids, np.array(e["data"][e["data_off"][i]:e["data_off"][i + 1]])))
    return out


def canonical(e):
    out = edge_list(e)
    out.sort(key=lambda t: (t[1], t[0], len(t[2]), tuple(np.round(t[2][:3], 6))))
    return out


def glc_gram(d, ids, data):
    """W^T W of a GLC record (W itself is only defined up to an orthogonal factor)."""
    n = d * len(ids)
    W = np.asarray(data[n:]).reshape(-1, n)
    return W.T @ W


def multi_parts(d, data):
    """SPG_EDGE_MULTI record (include/spg.h) -> (pairs [nm, 2], measurements [nm, ps], information W^T W)"""
    ps = abi.pose_stride(d)
    nm = int(data[0])
    r = d * nm
    pairs = np.asarray(data[1:1 + 2 * nm]).reshape(nm, 2).astype(int)
    meas = np.asarray(data[1 + 2 * nm:1 + 2 * nm + nm * ps]).reshape(nm, ps)
    W = np.asarray(data[1 + 2 * nm + nm * ps:1 + 2 * nm + nm * ps + r * r]).reshape(r, r)
    return pairs, meas, W.T @ W


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    den = max(np.abs(a).max(initial=0.0), np.abs(b).max(initial=0.0), 1e-300)
    return float(np.abs(a - b).max(initial=0.0) / den)


def compare_edge_sets(d, ea, eb, rtol=RTOL):
    """Asserts two edges() dicts describe the same graph: identical topology, measurements and
    information (binary) / W^T W (GLC) within rtol. Returns the worst relative error."""
    ca, cb = canonical(ea), canonical(eb)
    assert [(k, i) for k, i, _ in ca] == [(k, i) for k, i, _ in cb], "edge topology differs"
    ps = abi.pose_stride(d)
    worst = 0.0
    for (k, ids, xa), (_, _, xb) in zip(ca, cb):
        if k == abi.EDGE_BINARY:
            assert len(xa) == len(xb)
            worst = max(worst, rel_err(xa[:ps], xb[:ps]), rel_err(xa[ps:], xb[ps:]))
        elif k == abi.EDGE_MULTI:
            (pa, ma, oa), (pb, mb, ob) = multi_parts(d, xa), multi_parts(d, xb)
            assert np.array_equal(pa, pb), "measurement pairs of a correlated edge differ"
            worst = max(worst, rel_err(ma, mb), rel_err(oa, ob))
        else:
            n = d * len(ids)
            worst = max(worst, rel_err(xa[:n], xb[:n]))
            ga, gb = glc_gram(d, ids, xa), glc_gram(d, ids, xb)
            worst = max(worst, rel_err(ga, gb))
    assert worst <= rtol, f"edge payload mismatch: {worst:.3e} > {rtol}"
    return worst


def probe_vectors(n, seed=20240611):
    """Four deterministic probe vectors of length n (SplitMix64 -> uniform in [-1, 1))."""
    idx = np.arange(4 * n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (idx + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return ((x >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0).reshape(4, n)


def edge_digest(d, edges):
    """edges() dict -> [(kind, ids, meas, sketch[4, n])] in canonical order: the action of every edge's information on four
    fixed probe vectors — (W^T W) z for GLC edges (W is only defined up to an orthogonal factor), Omega z for pose-pose
    edges. What tests/golden/make_dense_digest.py stores for full-size Dense runs whose W blocks are too large to commit."""
    ps = abi.pose_stride(d)
    out = []
    for kind, ids, data in canonical(edges):
        if kind == abi.EDGE_BINARY:
            om = np.zeros((d, d))
            om[np.triu_indices(d)] = data[ps:]
            om = om + om.T - np.diag(np.diag(om))
            out.append((kind, ids, np.asarray(data[:ps]), probe_vectors(d) @ om))
        else:
            n = d * len(ids)
            W = np.asarray(data[n:]).reshape(-1, n)
            z = probe_vectors(n)
            out.append((kind, ids, np.asarray(data[:n]), (z @ W.T) @ W))
    return out


def load_digest(name):
    z = np.load(os.path.join(GOLDEN_DIR, "digest_" + name + ".npz"))
    out = []
    for e in range(len(z["kinds"])):
        ids = tuple(int(x) for x in z["ids"][z["id_off"][e]:z["id_off"][e + 1]])
        sk = z["sketch"][z["sketch_off"][e]:z["sketch_off"][e + 1]].reshape(4, -1)
        out.append((int(z["kinds"][e]), ids, z["meas"][z["meas_off"][e]:z["meas_off"][e + 1]], sk))
    return str(z["source"]), int(z["algorithm"]), int(z["topology"]), out, z


def compare_digests(da, db, rtol=RTOL):
    assert [(k, i) for k, i, *_ in da] == [(k, i) for k, i, *_ in db], "edge topology differs"
    worst = 0.0
    for (_, _, ma, sa), (_, _, mb, sb) in zip(da, db):
        worst = max(worst, rel_err(ma, mb), rel_err(sa, sb))
    assert worst <= rtol, f"edge payload mismatch: {worst:.3e} > {rtol}"
    return worst


def first_round_batch(g, which, opts, limit=None, k_range=None):
    """Gather, in pure Python, the blankets of the removal-list vertices that touch no other list
    vertex's blanket (a trivially independent set), as an spg_batch dict. Tree-type blankets only.
    k_range = (lo, hi): only blankets with lo <= kept vertices <= hi."""
    d = g["pose_dim"]
    ids = g["ids"]
    idx = {int(v): i for i, v in enumerate(ids)}
    adj = {int(v): [] for v in ids}
    for e, (a, b) in enumerate(g["edge_ij"]):
        adj[int(a)].append(e)
        if int(b) != int(a):
            adj[int(b)].append(e)
    used = set()
    vert_off, n_remove, vert_id, pose = [0], [], [], []
    edge_off, edge_kind, edge_vert_off, edge_vert, edge_data_off, edge_data = [0], [], [0], [], [0], []
    roots = []
    for v in which:
        v = int(v)
        nb = {v}
        for e in adj[v]:
            nb.update(int(x) for x in g["edge_ij"][e])
        if nb & used:
            continue
        if k_range is not None and not (k_range[0] <= len(nb) - 1 <= k_range[1]):
            continue
        used |= nb
        order = [v] + sorted(nb - {v})
        loc = {x: i for i, x in enumerate(order)}
        es = sorted({e for x in order for e in adj[x] if all(int(y) in loc for y in g["edge_ij"][e])})
        for x in order:
            vert_id.append(x)
            pose.append(g["poses"][idx[x]])
        vert_off.append(len(vert_id))
        n_remove.append(1)
        for e in es:
            a, b = (int(x) for x in g["edge_ij"][e])
            edge_kind.append(abi.EDGE_BINARY)
            edge_vert += [loc[a], loc[b]]
            edge_vert_off.append(len(edge_vert))
            edge_data.append(g["edge_data"][e])
            edge_data_off.append(edge_data_off[-1] + len(g["edge_data"][e]))
        edge_off.append(len(edge_kind))
        roots.append(v)
        if limit and len(roots) >= limit:
            break
    return {"vert_off": np.array(vert_off, np.int32), "n_remove": np.array(n_remove, np.int32),
            "vert_id": np.array(vert_id, np.int32), "pose": np.array(pose, np.float64).reshape(-1),
            "edge_off": np.array(edge_off, np.int32), "edge_kind": np.array(edge_kind, np.int32),
            "edge_vert_off": np.array(edge_vert_off, np.int32), "edge_vert": np.array(edge_vert, np.int32),
            "edge_data_off": np.array(edge_data_off, np.int64),
            "edge_data": np.concatenate(edge_data) if edge_data else np.zeros(0)}, roots


def prefix_graph(g, which, n):
    """Sub-graph over the first n vertices (by position in g["ids"]) and the removal ids inside it."""
    ids = np.asarray(g["ids"])[:n]
    keep = set(int(x) for x in ids)
    m = np.array([int(a) in keep and int(b) in keep for a, b in g["edge_ij"]], bool)
    sub = {"pose_dim": g["pose_dim"], "ids": ids, "poses": np.asarray(g["poses"])[:n],
           "edge_ij": np.asarray(g["edge_ij"])[m], "edge_data": np.asarray(g["edge_data"])[m]}
    w = np.array([int(v) for v in which if int(v) in keep and int(v) != int(ids[-1])], np.int32)
    return sub, w
