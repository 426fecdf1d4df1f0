"""Loader for the CPU oracle (oracle/libspg_ref.so) — the CHECKER used by tests, smoke() and
bench.py's cpu_baseline leg. Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

from sparsifyposegraph_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, "libspg_ref.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ORACLE_DIR, "libspg_ref.so")
        srcs = [os.path.join(_ORACLE_DIR, f) for f in ("spg_ref.cpp", "ref_blanket.hpp", "ref_geom.hpp", "ref_la.hpp")]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            build_oracle()
        L = C.CDLL(path)
        f64p, i32p, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.spgref_graph_create.restype = C.c_void_p
        L.spgref_graph_create.argtypes = [C.c_int]
        L.spgref_graph_destroy.argtypes = [C.c_void_p]
        L.spgref_graph_add_vertex.argtypes = [C.c_void_p, C.c_int, f64p]
        L.spgref_graph_add_edge.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, f64p, C.c_int64]
        L.spgref_graph_marginalize.argtypes = [C.c_void_p, i32p, C.c_int, C.POINTER(abi.Options)]
        L.spgref_graph_last_seconds.restype = C.c_double
        L.spgref_graph_last_seconds.argtypes = [C.c_void_p]
        for n in ("spgref_graph_num_vertices", "spgref_graph_num_edges", "spgref_graph_last_blanket_count"):
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("spgref_graph_edge_data_size", "spgref_graph_edge_vert_size"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = C.c_int64
        L.spgref_graph_get_vertices.argtypes = [C.c_void_p, i32p, f64p]
        L.spgref_graph_get_edges.argtypes = [C.c_void_p, i32p, i32p, i32p, i64p, f64p]
        L.spgref_graph_last_blankets.argtypes = [C.c_void_p, i32p, i32p, i32p, f64p, f64p, i32p]
        L.spgref_marginalize_batch_mt.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.Batch), C.POINTER(abi.Result), C.c_int]
        L.spgref_graph_information.restype = C.c_int64
        L.spgref_graph_information.argtypes = [C.c_void_p, C.c_int32, f64p, C.c_int64]
        L.spgref_graph_kullback_leibler.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, f64p]
        L.spgref_graph_optimize.argtypes = [C.c_void_p, C.c_int, C.c_int32, f64p]
        L.spgref_graph_optimize_fixed.argtypes = [C.c_void_p, C.c_int, i32p, C.c_int, f64p]
        L.spgref_graph_chi2.restype = C.c_double
        L.spgref_graph_chi2.argtypes = [C.c_void_p, C.c_int32]
        L.spgref_graph_set_estimate.argtypes = [C.c_void_p, C.c_int, f64p]
        L.spgref_spd_logdet.restype = C.c_double
        L.spg_run_round.argtypes = [C.c_void_p, C.POINTER(abi.RoundDesc)]
        L.spg_run_round_mt.argtypes = [C.c_void_p, C.POINTER(abi.RoundDesc)]
        L.spgref_set_round_threads.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class OracleGraph:
    """Sequential, literal VertexRemover::remove loop of the oracle (spgref_graph_*)."""

    def __init__(self, pose_dim):
        self.L = lib()
        self.d = pose_dim
        self.h = self.L.spgref_graph_create(pose_dim)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.spgref_graph_destroy(self.h)
            self.h = None

    @classmethod
    def from_dict(cls, g):
        o = cls(g["pose_dim"])
        poses = np.ascontiguousarray(g["poses"], np.float64)
        for i, p in zip(g["ids"], poses):
            o.L.spgref_graph_add_vertex(o.h, int(i), _p(p, C.c_double))
        data = np.ascontiguousarray(g["edge_data"], np.float64)
        ij = np.ascontiguousarray(g["edge_ij"], np.int32)
        for e in range(len(ij)):
            rc = o.L.spgref_graph_add_edge(o.h, abi.EDGE_BINARY, 2, _p(ij[e], C.c_int32), _p(data[e], C.c_double), data.shape[1])
            assert rc == 0
        return o

    def add_edge(self, kind, ids, data):
        ids = np.ascontiguousarray(ids, np.int32)
        data = np.ascontiguousarray(data, np.float64)
        return self.L.spgref_graph_add_edge(self.h, kind, len(ids), _p(ids, C.c_int32), _p(data, C.c_double), len(data))

    def marginalize(self, which, opts):
        which = np.ascontiguousarray(which, np.int32)
        return self.L.spgref_graph_marginalize(self.h, _p(which, C.c_int32), len(which), C.byref(opts))

    def seconds(self):
        return self.L.spgref_graph_last_seconds(self.h)

    def optimize(self, iterations=50, fixed_id=0):
        """GraphWrapperG2O::optimize (src/graph_wrapper_g2o.cpp:250-269): g2o LM restated, dense."""
        st = np.zeros(5)
        rc = self.L.spgref_graph_optimize(self.h, int(iterations), int(fixed_id), _p(st, C.c_double))
        assert rc == 0, rc
        return dict(zip(("iterations", "trials", "chi2_initial", "chi2_final", "lambda_final"), st))

    def optimize_fixed(self, fixed_ids, iterations=50):
        fx = np.ascontiguousarray(fixed_ids, np.int32)
        st = np.zeros(5)
        assert self.L.spgref_graph_optimize_fixed(self.h, int(iterations), _p(fx, C.c_int32), len(fx), _p(st, C.c_double)) == 0
        return dict(zip(("iterations", "trials", "chi2_initial", "chi2_final", "lambda_final"), st))

    def chi2(self, fixed_id=0):
        return float(self.L.spgref_graph_chi2(self.h, int(fixed_id)))

    def set_estimate(self, vid, pose):
        pose = np.ascontiguousarray(pose, np.float64)
        return self.L.spgref_graph_set_estimate(self.h, int(vid), _p(pose, C.c_double))

    def information(self, fixed_id):
        """other->information() (src/graph_wrapper_g2o.cpp:351-358), fixed vertex dropped."""
        n = int(self.L.spgref_graph_information(self.h, int(fixed_id), None, 0))
        out = np.zeros((n, n))
        self.L.spgref_graph_information(self.h, int(fixed_id), _p(out, C.c_double), out.size)
        return out

    def kullback_leibler(self, other, fixed_id):
        """baseline.kullbackLeibler(other) (src/graph_wrapper_g2o.cpp:531-548) -> dict of terms"""
        t = np.zeros(6)
        rc = self.L.spgref_graph_kullback_leibler(self.h, other.h, int(fixed_id), _p(t, C.c_double))
        assert rc == 0, rc
        return dict(zip(("kld", "innerprod", "mahalanobis", "logdetx", "logdety", "n"), t))

    def vertices(self):
        n = self.L.spgref_graph_num_vertices(self.h)
        ids = np.zeros(n, np.int32)
        poses = np.zeros((n, abi.pose_stride(self.d)))
        self.L.spgref_graph_get_vertices(self.h, _p(ids, C.c_int32), _p(poses, C.c_double))
        return ids, poses

    def edges(self):
        ne = self.L.spgref_graph_num_edges(self.h)
        nd = self.L.spgref_graph_edge_data_size(self.h)
        nv = self.L.spgref_graph_edge_vert_size(self.h)
        kind = np.zeros(ne, np.int32)
        voff = np.zeros(ne + 1, np.int32)
        vids = np.zeros(max(nv, 1), np.int32)
        doff = np.zeros(ne + 1, np.int64)
        data = np.zeros(max(nd, 1))
        self.L.spgref_graph_get_edges(self.h, _p(kind, C.c_int32), _p(voff, C.c_int32), _p(vids, C.c_int32), _p(doff, C.c_int64), _p(data, C.c_double))
        return {"kind": kind, "vert_off": voff, "vert_ids": vids[:nv], "data_off": doff, "data": data[:nd]}

    def blankets(self):
        n = self.L.spgref_graph_last_blanket_count(self.h)
        root, status, info, k = (np.zeros(n, np.int32) for _ in range(4))
        kld, gap = np.zeros(n), np.zeros(n)
        self.L.spgref_graph_last_blankets(self.h, _p(root, C.c_int32), _p(status, C.c_int32), _p(info, C.c_int32), _p(kld, C.c_double), _p(gap, C.c_double), _p(k, C.c_int32))
        return {"root": root, "status": status, "info": info, "kld": kld, "min_gap": gap, "k": k}


def canonical_edges(e, d):
    """Sort edges of an edges() dict by (endpoints, kind) -> list of (kind, ids tuple, data array)."""
    out = []
    for i in range(len(e["kind"])):
        ids = tuple(int(x) for x in e["vert_ids"][e["vert_off"][i]:e["vert_off"][i + 1]])
        out.append((int(e["kind"][i]), ids, np.array(e["data"][e["data_off"][i]:e["data_off"][i + 1]])))
    out.sort(key=lambda t: (t[1], t[0], len(t[2]), tuple(np.round(t[2][:3], 6))))
    return out


# ---------------------------------------------------------------------------------------------
# Injected compute backend: host memory as "arena", the oracle's spg_run_round as the arithmetic.
# Lets the CPU suite drive the product's host scheduler / round protocol without a GPU.
class OracleBackend:
    """spg_backend whose arithmetic is the oracle's spg_run_round. With slots=True it also offers what
    the HIP backend offers the pipelined driver: per-slot host mailboxes (the out records of a launch
    are copied there, ready words included) and per-slot synchronisation — so the CPU suite drives the
    product's multi-batch-in-flight path (scheduling against uncommitted batches, batch splitting,
    mailbox polling, late KLD harvest) without a GPU."""

    NSLOT = 8

    def __init__(self, slots=False, threads=1):
        self.L = lib()
        self.L.spgref_set_round_threads(int(threads))
        run = self.L.spg_run_round_mt if threads > 1 else self.L.spg_run_round
        self.bufs = {}
        self.mail = [None] * self.NSLOT

        def _alloc(user, doubles):
            buf = (C.c_double * max(int(doubles), 1))()
            addr = C.addressof(buf)
            self.bufs[addr] = buf
            return addr

        def _release(user, p):
            self.bufs.pop(p, None)

        def _upload(user, dst, src, doubles):
            C.memmove(dst, src, int(doubles) * 8)
            return 0

        def _download(user, dst, src, doubles):
            C.memmove(dst, src, int(doubles) * 8)
            return 0

        def _run_round(user, arena, rd):
            rc = run(arena, rd)
            r = rd.contents
            if rc == 0 and slots and r.mail_len > 0:
                sl = r.slot % self.NSLOT
                if self.mail[sl] is None or len(self.mail[sl]) < r.mail_len:
                    self.mail[sl] = (C.c_double * int(max(r.mail_len, 2 * len(self.mail[sl] or []))))()
                C.memmove(self.mail[sl], arena + r.mail_base * 8, int(r.mail_len) * 8)
            return rc

        def _sync(user):
            return 0

        def _mailbox(user):
            return C.addressof(self.mail[0]) if self.mail[0] is not None else None

        def _sync_slot(user, slot):
            return 0

        def _mailbox_slot(user, slot):
            m = self.mail[slot % self.NSLOT]
            return C.addressof(m) if m is not None else None

        self.cb = (abi._ALLOC(_alloc), abi._RELEASE(_release), abi._UPLOAD(_upload), abi._DOWNLOAD(_download),
                   abi._RUN_ROUND(_run_round), abi._SYNC(_sync))
        if slots:
            self.cb += (abi._MAILBOX(_mailbox), abi._SYNC_SLOT(_sync_slot), abi._MAILBOX_SLOT(_mailbox_slot))
            self.struct = abi.Backend(None, *self.cb)
        else:
            self.struct = abi.Backend(None, *self.cb, abi._MAILBOX(), abi._SYNC_SLOT(), abi._MAILBOX_SLOT())


def injected_context(slots=False, threads=1):
    """Context of the PRODUCT library whose compute backend is the oracle (CPU tests and bench.py's
    all-cores CPU baseline only)."""
    from sparsifyposegraph_amd.lib import Context
    be = OracleBackend(slots, threads)
    return Context.injected(be.struct, keep=be)
