"""ctypes view of include/spg.h.

Pure plumbing: structure layouts and numpy marshalling for the C ABI. The same bindings drive the
product library (libspg_hip.so) and, from tests/ only, the CPU oracle (oracle/libspg_ref.so), which
exports `spg_marginalize_batch` / `spg_run_round` with identical signatures.
"""
import ctypes as C

import numpy as np

# enums (include/spg.h)
ALG_NFR, ALG_GLC = 0, 1
TOPO_TREE, TOPO_SUBGRAPH, TOPO_CLIQUEY_SUBGRAPH, TOPO_DENSE, TOPO_CLIQUEY_DENSE = range(5)
LIN_LOCAL, LIN_GLOBAL = 0, 1
EDGE_BINARY, EDGE_GLC, EDGE_MULTI = 0, 1, 2
ST_OK, ST_HMM_NOT_PD, ST_EIG_FAIL, ST_NONFINITE, ST_TIKHONOV_NOT_PD, ST_CLOSED_FORM_NOT_PD, \
    ST_KLD_NOT_PD, ST_NEEDS_INTERIOR_POINT, ST_MARGINAL_NOT_PD, ST_EMPTY_BLANKET, ST_UNSUPPORTED, \
    ST_NEEDS_LOCAL_OPTIMIZATION = range(12)
INFO_RANK_DEFICIENT, INFO_GLC_ROOT_EDGE, INFO_IP_HESSIAN_NOT_PD = 1, 2, 4
FLAG_FORCE_EIG = 2
EINVAL, ENODEV, ENOMEM, ECAPACITY, EHIP, EIO, ESTATE, EBLANKET, ENOTPD = -1, -2, -3, -4, -5, -6, -7, -8, -9
OUT_HDR = 6

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


class Options(C.Structure):
    _fields_ = [("pose_dim", C.c_int32), ("algorithm", C.c_int32), ("topology", C.c_int32),
                ("lin_point", C.c_int32), ("include_intra_clique", C.c_int32), ("flags", C.c_int32),
                ("chord_ratio", C.c_double)]


def make_options(pose_dim, algorithm=ALG_NFR, topology=TOPO_TREE, lin_point=LIN_GLOBAL, flags=0,
                 chord_ratio=1.0, include_intra_clique=1):
    return Options(pose_dim, algorithm, topology, lin_point, include_intra_clique, flags, chord_ratio)


class Batch(C.Structure):
    _fields_ = [("B", C.c_int32), ("vert_off", _i32p), ("n_remove", _i32p), ("vert_id", _i32p),
                ("pose", _f64p), ("edge_off", _i32p), ("edge_kind", _i32p), ("edge_vert_off", _i32p),
                ("edge_vert", _i32p), ("edge_data_off", _i64p), ("edge_data", _f64p)]


class Result(C.Structure):
    _fields_ = [("target_info", _f64p), ("target_info_off", _i64p), ("new_edge_off", _i32p),
                ("new_edge_kind", _i32p), ("new_edge_vert_off", _i32p), ("new_edge_vert", _i32p),
                ("new_edge_data_off", _i64p), ("new_edge_data", _f64p), ("new_edge_cap", C.c_int32),
                ("new_edge_vert_cap", C.c_int32), ("new_edge_data_cap", C.c_int64), ("kld", _f64p),
                ("min_gap", _f64p), ("status", _i32p), ("info", _i32p)]


class MargStats(C.Structure):
    _fields_ = [("n_removed", C.c_int32), ("n_rounds", C.c_int32), ("n_new_edges", C.c_int32),
                ("n_bad_status", C.c_int32), ("max_blanket", C.c_int32), ("n_launches", C.c_int32),
                ("kld_sum", C.c_double), ("host_seconds", C.c_double), ("device_seconds", C.c_double),
                ("schedule_seconds", C.c_double), ("commit_seconds", C.c_double), ("launch_seconds", C.c_double),
                ("n_batches", C.c_int32), ("n_exchanged", C.c_int32), ("exchange_seconds", C.c_double),
                ("exchanged_bytes", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class KldTerms(C.Structure):
    """spg_kld_terms (include/spg.h)"""
    _fields_ = [("kld", C.c_double), ("innerprod", C.c_double), ("mahalanobis", C.c_double), ("logdetx", C.c_double),
                ("logdety", C.c_double), ("n", C.c_int64), ("n_marginalized", C.c_int64), ("device_seconds", C.c_double),
                ("solver", C.c_int32), ("supernodes", C.c_int32), ("front_bytes", C.c_double), ("factor_flops", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class OptimizeStats(C.Structure):
    """spg_optimize_stats (include/spg.h)"""
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double), ("n", C.c_int64), ("device_seconds", C.c_double),
                ("solver", C.c_int32), ("supernodes", C.c_int32), ("front_bytes", C.c_double), ("factor_flops", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SparsePlanInfo(C.Structure):
    """spg_sparse_plan_info (include/spg.h)"""
    _fields_ = [("n_supernodes", C.c_int32), ("n_marg_supernodes", C.c_int32), ("n_levels", C.c_int32), ("pad_", C.c_int32),
                ("n_rows", C.c_int64), ("front_bytes", C.c_double), ("flops", C.c_double)]


SOLVER_AUTO, SOLVER_DENSE, SOLVER_SPARSE = 0, 1, 2


class RoundInfo(C.Structure):
    _fields_ = [("n_blankets", C.c_int32), ("my_first", C.c_int32), ("my_count", C.c_int32),
                ("region_off", C.c_int64), ("chunk_len", C.c_int64), ("exchange", C.c_int32), ("pad_", C.c_int32)]


class BlanketDesc(C.Structure):
    _fields_ = [("vert_begin", C.c_int32), ("n_vert", C.c_int32), ("n_remove", C.c_int32),
                ("edge_begin", C.c_int32), ("n_edge", C.c_int32), ("n_new_max", C.c_int32),
                ("n_new_vert_max", C.c_int32), ("pad_", C.c_int32), ("new_off", C.c_int64),
                ("new_len", C.c_int64), ("out_off", C.c_int64), ("tinfo_off", C.c_int64)]


class EdgeRef(C.Structure):
    _fields_ = [("off", C.c_int64), ("len", C.c_int32), ("kind", C.c_int32), ("vbegin", C.c_int32),
                ("nv", C.c_int32)]


class RoundDesc(C.Structure):
    _fields_ = [("opts", C.POINTER(Options)), ("n_blankets", C.c_int32), ("first", C.c_int32),
                ("count", C.c_int32), ("blankets", C.POINTER(BlanketDesc)), ("vert_pose_off", _i64p),
                ("edges", C.POINTER(EdgeRef)), ("edge_vert", _i32p), ("n_vert_total", C.c_int64),
                ("n_edge_total", C.c_int64), ("n_edge_vert_total", C.c_int64), ("mail_base", C.c_int64), ("mail_len", C.c_int64), ("slot", C.c_int32), ("tag", C.c_int32)]


_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int64)
_RELEASE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_UPLOAD = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, _f64p, C.c_int64)
_DOWNLOAD = C.CFUNCTYPE(C.c_int, C.c_void_p, _f64p, C.c_void_p, C.c_int64)
_RUN_ROUND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(RoundDesc))
_SYNC = C.CFUNCTYPE(C.c_int, C.c_void_p)
_MAILBOX = C.CFUNCTYPE(C.c_void_p, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int)
_SYNC_SLOT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
_MAILBOX_SLOT = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int)


class Backend(C.Structure):
    _fields_ = [("user", C.c_void_p), ("alloc", _ALLOC), ("release", _RELEASE), ("upload", _UPLOAD),
                ("download", _DOWNLOAD), ("run_round", _RUN_ROUND), ("synchronize", _SYNC), ("mailbox", _MAILBOX),
                ("synchronize_slot", _SYNC_SLOT), ("mailbox_slot", _MAILBOX_SLOT)]


def pose_stride(d):
    return 3 if d == 3 else 7


def binary_record_len(d):
    return pose_stride(d) + d * (d + 1) // 2


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None else typ()


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def pack_batch(b):
    """dict of numpy arrays (keys = spg_batch field names) -> (Batch, keepalive list)."""
    arrs = {
        "vert_off": _c(b["vert_off"], np.int32), "n_remove": _c(b["n_remove"], np.int32),
        "vert_id": _c(b["vert_id"], np.int32), "pose": _c(b["pose"], np.float64),
        "edge_off": _c(b["edge_off"], np.int32), "edge_kind": _c(b["edge_kind"], np.int32),
        "edge_vert_off": _c(b["edge_vert_off"], np.int32), "edge_vert": _c(b["edge_vert"], np.int32),
        "edge_data_off": _c(b["edge_data_off"], np.int64), "edge_data": _c(b["edge_data"], np.float64),
    }
    s = Batch()
    s.B = len(arrs["n_remove"])
    for k, a in arrs.items():
        typ = dict(Batch._fields_)[k]
        setattr(s, k, _p(a, typ))
    return s, arrs


def batch_capacities(b, opts):
    """Upper bounds for the result buffers of one batch (new edges / endpoints / doubles)."""
    d = opts.pose_dim
    vert_off = np.asarray(b["vert_off"])
    k = np.diff(vert_off) - np.asarray(b["n_remove"])
    k = np.maximum(k, 0).astype(np.int64)
    if opts.algorithm == ALG_NFR:
        ne = int(np.maximum(k - 1, 0).sum())
        if opts.topology in (TOPO_DENSE, TOPO_SUBGRAPH):
            ne = int((k * (k - 1) // 2).sum())
        nd = ne * binary_record_len(d)
        if opts.topology in (TOPO_CLIQUEY_SUBGRAPH, TOPO_CLIQUEY_DENSE):
            # correlated edges (SPG_EDGE_MULTI): at most k - 1 measurements in all, one record may hold all of them
            nm = np.maximum(k - 1, 0)
            nd += int((1 + 2 * nm + nm * pose_stride(d) + (d * nm) ** 2).sum())
        return ne, 2 * ne, nd
    if opts.topology == TOPO_DENSE:
        n = d * k
        return int((k > 0).sum()), int(k.sum()), int((n + n * n).sum())
    ne = int(k.sum())
    n2 = 2 * d
    return ne, int(np.maximum(2 * k - 1, 0).sum()), int((k * (n2 + n2 * n2)).sum())


def marginalize_batch(lib, ctx, opts, b, want_target=True):
    """Run spg_marginalize_batch of `lib` on batch dict `b`; returns a dict of numpy outputs."""
    sb, keep = pack_batch(b)
    B = sb.B
    d = opts.pose_dim
    ne_cap, nv_cap, nd_cap = batch_capacities(b, opts)
    k = (np.diff(keep["vert_off"]) - keep["n_remove"]).astype(np.int64)
    n = d * np.maximum(k, 0)
    toff = np.zeros(B + 1, np.int64)
    toff[1:] = np.cumsum(n * n)
    out = {
        "target_info": np.zeros(int(toff[-1]) if want_target else 0, np.float64),
        "target_info_off": toff,
        "new_edge_off": np.zeros(B + 1, np.int32),
        "new_edge_kind": np.zeros(max(ne_cap, 1), np.int32),
        "new_edge_vert_off": np.zeros(max(ne_cap, 1) + 1, np.int32),
        "new_edge_vert": np.zeros(max(nv_cap, 1), np.int32),
        "new_edge_data_off": np.zeros(max(ne_cap, 1) + 1, np.int64),
        "new_edge_data": np.zeros(max(nd_cap, 1), np.float64),
        "kld": np.full(B, np.nan), "min_gap": np.full(B, np.inf),
        "status": np.full(B, -1, np.int32), "info": np.zeros(B, np.int32),
    }
    r = Result()
    r.target_info = _p(out["target_info"], _f64p) if want_target and toff[-1] > 0 else _f64p()
    r.target_info_off = _p(out["target_info_off"], _i64p)
    for kf in ("new_edge_off", "new_edge_kind", "new_edge_vert_off", "new_edge_vert", "status", "info"):
        setattr(r, kf, _p(out[kf], _i32p))
    r.new_edge_data_off = _p(out["new_edge_data_off"], _i64p)
    r.new_edge_data = _p(out["new_edge_data"], _f64p)
    r.kld = _p(out["kld"], _f64p)
    r.min_gap = _p(out["min_gap"], _f64p)
    r.new_edge_cap, r.new_edge_vert_cap, r.new_edge_data_cap = max(ne_cap, 1), max(nv_cap, 1), max(nd_cap, 1)
    fn = lib.spg_marginalize_batch
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Batch), C.POINTER(Result)]
    rc = fn(ctx, C.byref(opts), C.byref(sb), C.byref(r))
    if rc != 0:
        raise RuntimeError(f"spg_marginalize_batch failed: rc={rc}")
    ne = int(out["new_edge_off"][B])
    out["new_edge_kind"] = out["new_edge_kind"][:ne]
    out["new_edge_vert_off"] = out["new_edge_vert_off"][:ne + 1]
    out["new_edge_data_off"] = out["new_edge_data_off"][:ne + 1]
    out["new_edge_vert"] = out["new_edge_vert"][:int(out["new_edge_vert_off"][ne])]
    out["new_edge_data"] = out["new_edge_data"][:int(out["new_edge_data_off"][ne])]
    out["n"] = n
    return out
