"""Host-side mirror of the reference's GraphWrapper / decimation interface over the C ABI.

Method names follow the reference (src/graph_wrapper.h:40-78, src/graph_wrapper_g2o.h:63-84,
src/decimation.h:18-22) so call sites read the same; the arithmetic lives in libspg_hip.so.
"""
import ctypes as C

import numpy as np

from . import abi
from .lib import Context, check, load

_f64p, _i32p, _i64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class SparsityOptions:
    """src/sparsity_options.h:11-30 (defaults identical) plus the algorithm selector."""
    Tree, Subgraph, CliqueySubgraph, Dense, CliqueyDense = range(5)
    Local, Global = 0, 1

    def __init__(self, topology=0, chordRatio=1.0, linPoint=0, includeIntraClique=True):
        self.topology = topology
        self.chordRatio = chordRatio
        self.linPoint = linPoint
        self.includeIntraClique = includeIntraClique

    def to_abi(self, pose_dim, use_glc, flags=0):
        return abi.make_options(pose_dim, abi.ALG_GLC if use_glc else abi.ALG_NFR, self.topology,
                                self.linPoint, flags, self.chordRatio, int(self.includeIntraClique))


class DecimateOptions:
    def __init__(self, sparsity=2, clusterSize=1):
        self.sparsity = sparsity
        self.clusterSize = clusterSize


def _decimate(fn, last, endvert, opts):
    L = load()
    cap = max(endvert + 8, 16)
    out = np.zeros(cap, np.int32)
    n = getattr(L, fn)(int(last), int(endvert), int(opts.sparsity), int(opts.clusterSize), _p(out, C.c_int32), cap)
    return [int(x) for x in out[:n]]


def globalDecimate(last, endvert, opts):
    """src/decimation.cpp:36-49"""
    return _decimate("spg_decimate_global", last, endvert, opts)


def onlineDecimate(last, endvert, opts):
    """src/decimation.cpp:27-34"""
    return _decimate("spg_decimate_online", last, endvert, opts)


def clusterDecimate(last, endvert, opts):
    """src/decimation.cpp:11-25"""
    return _decimate("spg_decimate_cluster", last, endvert, opts)


class GraphWrapperHIP:
    """GraphWrapperG2O(verbose, useGLC) (src/graph_wrapper_g2o.cpp:102) with the marginalisation
    path, optimize() and the global KLD on the MI355X."""

    def __init__(self, ctx=None, pose_dim=6, useGLC=False, _handle=None):
        self.L = load()
        self.ctx = ctx if ctx is not None else Context()
        self.useGLC = bool(useGLC)
        if _handle is None:
            h = C.c_void_p()
            check(self.L.spg_graph_create(self.ctx.h, int(pose_dim), C.byref(h)), self.ctx.h, "spg_graph_create")
            _handle = h
        self.h = _handle
        self.d = self.L.spg_graph_pose_dim(self.h)
        self.last_stats = None
        self.ctx._track(self)

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def load(cls, fname, ctx=None, useGLC=False):
        """GraphWrapperG2O(fname, optimize=false, useGLC) (src/graph_wrapper_g2o.cpp:107-154)"""
        ctx = ctx if ctx is not None else Context()
        h = C.c_void_p()
        check(load().spg_graph_load_g2o(ctx.h, fname.encode(), C.byref(h)), ctx.h, "spg_graph_load_g2o")
        return cls(ctx=ctx, useGLC=useGLC, _handle=h)

    @classmethod
    def from_dict(cls, g, ctx=None, useGLC=False):
        o = cls(ctx=ctx, pose_dim=g["pose_dim"], useGLC=useGLC)
        poses = np.ascontiguousarray(g["poses"], np.float64)
        ids = np.ascontiguousarray(g["ids"], np.int32)
        check(o.L.spg_graph_add_vertices(o.h, len(ids), _p(ids, C.c_int32), _p(poses, C.c_double)), o.ctx.h, "add_vertices")
        data = np.ascontiguousarray(g["edge_data"], np.float64)
        ij = np.ascontiguousarray(g["edge_ij"], np.int32)
        if len(ij):
            check(o.L.spg_graph_add_edges(o.h, len(ij), _p(ij, C.c_int32), _p(data, C.c_double)), o.ctx.h, "add_edges")
        return o

    def addVertex(self, id, init):
        init = np.ascontiguousarray(init, np.float64)
        check(self.L.spg_graph_add_vertex(self.h, int(id), _p(init, C.c_double)), self.ctx.h, "addVertex")

    def addEdge(self, frm, to, meas, info):
        """info: full d x d matrix or its row-wise upper triangle"""
        meas = np.ascontiguousarray(meas, np.float64)
        info = np.asarray(info, np.float64)
        if info.ndim == 2:
            info = info[np.triu_indices(self.d)]
        info = np.ascontiguousarray(info)
        check(self.L.spg_graph_add_edge(self.h, int(frm), int(to), _p(meas, C.c_double), _p(info, C.c_double)), self.ctx.h, "addEdge")

    def addGLCEdge(self, ids, meas, W):
        ids = np.ascontiguousarray(ids, np.int32)
        W = np.ascontiguousarray(W, np.float64)
        meas = np.ascontiguousarray(meas, np.float64)
        check(self.L.spg_graph_add_glc_edge(self.h, len(ids), _p(ids, C.c_int32), W.shape[0], _p(meas, C.c_double), _p(W, C.c_double)), self.ctx.h, "addGLCEdge")

    def addMultiEdge(self, ids, record):
        """MultiEdgeCorrelated over the vertices `ids` (src/multi_edge_correlated.hpp:28-63); `record` in the
        SPG_EDGE_MULTI layout of include/spg.h: nm | vertex pair of each measurement | measurements | W (information W^T W)."""
        ids = np.ascontiguousarray(ids, np.int32)
        record = np.ascontiguousarray(record, np.float64)
        check(self.L.spg_graph_add_multi_edge(self.h, len(ids), _p(ids, C.c_int32), _p(record, C.c_double), len(record)), self.ctx.h, "addMultiEdge")

    def setEstimate(self, vertexid, est):
        est = np.ascontiguousarray(est, np.float64)
        check(self.L.spg_graph_set_estimate(self.h, int(vertexid), _p(est, C.c_double)), self.ctx.h, "setEstimate")

    # ---- the hot path -------------------------------------------------------------------
    def marginalizeNoOptimize(self, which, options, flags=0):
        """src/graph_wrapper_g2o.cpp:398-453"""
        which = np.ascontiguousarray(which, np.int32)
        o = options.to_abi(self.d, self.useGLC, flags) if isinstance(options, SparsityOptions) else options
        st = abi.MargStats()
        rc = self.L.spg_graph_marginalize(self.h, _p(which, C.c_int32), len(which), C.byref(o), C.byref(st))
        self.last_stats = st.asdict()
        check(rc, self.ctx.h, "spg_graph_marginalize")
        return self.last_stats

    def marginalize_ranks(self, which, options, rank, nranks, flags=0):
        """spg_graph_marginalize_ranks with the library's built-in exchange (the context's RCCL communicator,
        Context.ranks): every rank calls this with identical arguments on an identical replica."""
        which = np.ascontiguousarray(which, np.int32)
        o = options.to_abi(self.d, self.useGLC, flags) if isinstance(options, SparsityOptions) else options
        st = abi.MargStats()
        rc = self.L.spg_graph_marginalize_ranks(self.h, _p(which, C.c_int32), len(which), C.byref(o), int(rank), int(nranks), None, None, C.byref(st))
        self.last_stats = st.asdict()
        check(rc, self.ctx.h, "spg_graph_marginalize_ranks")
        return self.last_stats

    def marginalize(self, which, options, flags=0):
        """src/graph_wrapper_g2o.cpp:455-463: marginalizeNoOptimize followed by optimize() (dense LM up to 12 k
        variables, the block-sparse multifrontal solver beyond: `last_optimize_stats["solver"]`)."""
        st = self.marginalizeNoOptimize(which, options, flags)
        self.optimize()
        return st

    def computeSubstituteEdge(self, marginalized, maxid, frm, to):
        """src/compute_substitute_edge.cpp:13-96 -> (from, to, meas, info_upper)"""
        marg = np.ascontiguousarray(sorted(marginalized), np.int32)
        f, t = C.c_int(int(frm)), C.c_int(int(to))
        meas = np.zeros(abi.pose_stride(self.d))
        info = np.zeros(self.d * (self.d + 1) // 2)
        check(self.L.spg_graph_substitute_edge(self.h, _p(marg, C.c_int32), len(marg), int(maxid), C.byref(f), C.byref(t),
                                               _p(meas, C.c_double), _p(info, C.c_double)), self.ctx.h, "computeSubstituteEdge")
        return f.value, t.value, meas, info

    def optimize(self, iterations=50, fixed_id=-1):
        """GraphWrapperG2O::optimize (src/graph_wrapper_g2o.cpp:250-269): g2o Levenberg-Marquardt with
        the first vertex fixed, on the device (dense or block-sparse factorisation, see
        Context.set_linear_solver). Returns the stats dict."""
        st = abi.OptimizeStats()
        check(self.L.spg_graph_optimize(self.h, int(iterations), int(fixed_id), C.byref(st)), self.ctx.h, "optimize")
        self.last_optimize_stats = st.asdict()
        return self.last_optimize_stats

    def chi2(self, other=None, iterations=50):
        """GraphWrapperG2O::chi2() / chi2(other) (src/graph_wrapper_g2o.cpp:501-529). Without `other`: the
        chi2 of the current estimates. With it: this graph's vertices that also exist in `other` are set
        to other's estimates and held fixed, the rest is optimised, chi2 is read and the estimates are
        restored (push / pop in the reference)."""
        if other is None:
            v = C.c_double()
            check(self.L.spg_graph_chi2(self.h, C.byref(v)), self.ctx.h, "chi2")
            return v.value
        ids, poses = self.vertices()
        mine = {int(i) for i in ids}
        oids, oposes = other.vertices()
        fixed = [int(i) for i in oids if int(i) in mine]
        for i, p in zip(oids, oposes):
            if int(i) in mine:
                self.setEstimate(int(i), p)
        fx = np.ascontiguousarray(sorted(set(fixed) | {int(ids[0])}), np.int32)
        st = abi.OptimizeStats()
        try:
            check(self.L.spg_graph_optimize_fixed(self.h, int(iterations), _p(fx, C.c_int32), len(fx), C.byref(st)), self.ctx.h, "chi2(other)")
        finally:
            for i, p in zip(ids, poses):
                self.setEstimate(int(i), p)
        return st.chi2_final

    def information(self, fixed_id=-1):
        """GraphWrapperG2O::information (src/graph_wrapper_g2o.cpp:351-358): dense Gauss-Newton
        information at the stored estimates, all vertices but the fixed one (default: smallest id)."""
        n = self.L.spg_graph_information(self.h, int(fixed_id), None, 0)
        check(min(int(n), 0), self.ctx.h, "information")
        out = np.zeros((int(n), int(n)))
        rc = self.L.spg_graph_information(self.h, int(fixed_id), _p(out, C.c_double), out.size)
        check(min(int(rc), 0), self.ctx.h, "information")
        return out

    def covariance(self, fixed_id=-1):
        """GraphWrapperG2O::covariance (src/graph_wrapper_g2o.cpp:368-373): inverse of information(), dense on
        the device (blocked fp64-MFMA Cholesky, triangular inverse, L^-T L^-1)."""
        n = self.L.spg_graph_covariance(self.h, int(fixed_id), None, 0)
        check(min(int(n), 0), self.ctx.h, "covariance")
        out = np.zeros((int(n), int(n)))
        rc = self.L.spg_graph_covariance(self.h, int(fixed_id), _p(out, C.c_double), out.size)
        check(min(int(rc), 0), self.ctx.h, "covariance")
        return out

    def clonePortion(self, maxid, optimize=True):
        """GraphWrapperG2O::clonePortion (src/graph_wrapper_g2o.cpp:334-356): vertices / edges up to maxid on the
        same context; the reference optimises the clone before returning it."""
        h = C.c_void_p()
        check(self.L.spg_graph_clone_portion(self.h, int(maxid), C.byref(h)), self.ctx.h, "clonePortion")
        gw = GraphWrapperHIP(ctx=self.ctx, useGLC=self.useGLC, _handle=h)
        if optimize and gw.numVertices() > 1 and gw.numEdges() > 0:
            gw.optimize()
        return gw

    def vertexEdges(self, vertexid):
        """GraphWrapper::Vertex::edges() (src/graph_wrapper.h:26): indices into edges() of the edges at the vertex"""
        n = self.L.spg_graph_vertex_edges(self.h, int(vertexid), None, 0)
        check(n, self.ctx.h, "vertexEdges")
        out = np.zeros(max(n, 1), np.int32)
        check(self.L.spg_graph_vertex_edges(self.h, int(vertexid), _p(out, C.c_int32), n), self.ctx.h, "vertexEdges")
        return out[:n]

    def writeString(self):
        """GraphWrapper::write(std::ofstream &) (src/graph_wrapper.h:62): the .g2o text"""
        txt, ln = C.c_void_p(), C.c_size_t()
        check(self.L.spg_graph_write_g2o_mem(self.h, C.byref(txt), C.byref(ln)), self.ctx.h, "write")
        try:
            return C.string_at(txt.value, ln.value).decode()
        finally:
            self.L.spg_free(txt)

    def kullbackLeibler(self, other, fixed_id=-1):
        """GraphWrapperG2O::kullbackLeibler(other) called on the baseline
        (src/graph_wrapper_g2o.cpp:531-548). Returns the KLD; the terms are in `last_kld_terms`."""
        t = abi.KldTerms()
        check(self.L.spg_graph_kullback_leibler(self.h, other.h, int(fixed_id), C.byref(t)), self.ctx.h, "kullbackLeibler")
        self.last_kld_terms = t.asdict()
        return t.kld

    # round-stepping form (multi-GPU driver in parallel.py)
    def begin(self, which, opts, rank, nranks):
        which = np.ascontiguousarray(which, np.int32)
        self._opts = opts
        check(self.L.spg_graph_marginalize_begin(self.h, _p(which, C.c_int32), len(which), C.byref(opts), rank, nranks), self.ctx.h, "marginalize_begin")

    def set_shard_threshold(self, min_blankets):
        check(self.L.spg_graph_set_shard_threshold(self.h, int(min_blankets)), self.ctx.h, "set_shard_threshold")

    def set_stream_emulation(self, seed):
        """-1: default drivers; >= 0: streaming driver on an injected backend, completion order from the seed (tests);
        -2: never use the streaming driver (A/B against the batch driver)."""
        check(self.L.spg_graph_set_stream_emulation(self.h, int(seed)), self.ctx.h, "set_stream_emulation")

    def round_prepare(self):
        info = abi.RoundInfo()
        rc = check(self.L.spg_graph_round_prepare(self.h, C.byref(info)), self.ctx.h, "round_prepare")
        return info if rc == 1 else None

    def round_compute(self):
        check(self.L.spg_graph_round_compute(self.h), self.ctx.h, "round_compute")

    def round_commit(self):
        check(self.L.spg_graph_round_commit(self.h), self.ctx.h, "round_commit")

    def end(self):
        st = abi.MargStats()
        rc = self.L.spg_graph_marginalize_end(self.h, C.byref(st))
        self.last_stats = st.asdict()
        check(rc, self.ctx.h, "marginalize_end")
        return self.last_stats

    def arena(self):
        cap = C.c_int64()
        p = self.L.spg_graph_arena(self.h, C.byref(cap))
        return p, cap.value

    def reserve(self, doubles):
        check(self.L.spg_graph_reserve(self.h, int(doubles)), self.ctx.h, "reserve")

    # ---- queries ------------------------------------------------------------------------
    def numVertices(self):
        return self.L.spg_graph_num_vertices(self.h)

    def numEdges(self):
        return self.L.spg_graph_num_edges(self.h)

    def vertices(self):
        n = self.numVertices()
        ids = np.zeros(n, np.int32)
        poses = np.zeros((n, abi.pose_stride(self.d)))
        check(self.L.spg_graph_get_vertices(self.h, _p(ids, C.c_int32), _p(poses, C.c_double)), self.ctx.h, "get_vertices")
        return ids, poses

    def edges(self):
        ne = self.numEdges()
        nd = self.L.spg_graph_edge_data_size(self.h)
        nv = self.L.spg_graph_edge_vert_size(self.h)
        kind = np.zeros(ne, np.int32)
        voff = np.zeros(ne + 1, np.int32)
        vids = np.zeros(max(nv, 1), np.int32)
        doff = np.zeros(ne + 1, np.int64)
        data = np.zeros(max(nd, 1))
        check(self.L.spg_graph_get_edges(self.h, _p(kind, C.c_int32), _p(voff, C.c_int32), _p(vids, C.c_int32), _p(doff, C.c_int64), _p(data, C.c_double)), self.ctx.h, "get_edges")
        return {"kind": kind, "vert_off": voff, "vert_ids": vids[:nv], "data_off": doff, "data": data[:nd]}

    def blankets(self):
        n = self.L.spg_graph_last_blanket_count(self.h)
        root, rnd, status, info = (np.zeros(n, np.int32) for _ in range(4))
        kld, gap = np.zeros(n), np.zeros(n)
        self.L.spg_graph_last_blankets(self.h, _p(root, C.c_int32), _p(rnd, C.c_int32), _p(status, C.c_int32), _p(info, C.c_int32), _p(kld, C.c_double), _p(gap, C.c_double))
        return {"root": root, "round": rnd, "status": status, "info": info, "kld": kld, "min_gap": gap}

    def write(self, fname):
        check(self.L.spg_graph_write_g2o(self.h, fname.encode()), self.ctx.h, "write")

    def printStats(self):
        """nodes / edges line of src/graph_wrapper_g2o.cpp:606-612 (fill-in needs the LM Hessian: omitted)"""
        return f"nodes = {self.numVertices() - 1}; edges = {self.numEdges()}"

    def close(self):
        if getattr(self, "h", None):
            # a graph's arena belongs to its context's backend: once the context is gone (Context.close() destroys the graphs
            # it still knows first, so this only happens when finalisers run in an arbitrary order inside a reference cycle)
            # the handle must not be touched any more
            if getattr(self.ctx, "h", None):
                self.L.spg_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
