"""Loader for the product library sparsifyposegraph_amd/libspg_hip.so (C ABI of include/spg.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is present when a
context is created, this module raises.
"""
import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPG_LIB_PATH") or os.path.join(_HERE, "libspg_hip.so")   # (override: A/B of two builds)
_lib = None

_f64p, _i32p, _i64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)

# every symbol include/spg.h declares for the product library: name -> (restype, argtypes)
SYMBOLS = {
    "spg_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "spg_ctx_create_injected": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(abi.Backend)]),
    "spg_get_unique_id": (C.c_int, [C.c_void_p]),
    "spg_ctx_create_ranks": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "spg_ctx_rank": (C.c_int, [C.c_void_p]),
    "spg_ctx_nranks": (C.c_int, [C.c_void_p]),
    "spg_allgather_region": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "spg_free": (None, [C.c_void_p]),
    "spg_ctx_destroy": (None, [C.c_void_p]),
    "spg_last_error": (C.c_char_p, [C.c_void_p]),
    "spg_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "spg_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "spg_ctx_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "spg_ctx_profile_read": (C.c_int, [C.c_void_p, _f64p, _f64p, _i64p, _i64p]),
    "spg_ctx_profile_read_worker": (C.c_int, [C.c_void_p, _f64p, _f64p, _i64p, _i64p]),
    "spg_ctx_profile_read_big": (C.c_int, [C.c_void_p, _f64p, _f64p, _i64p, _i32p]),
    "spg_marginalize_batch": (C.c_int, [C.c_void_p, C.POINTER(abi.Options), C.POINTER(abi.Batch), C.POINTER(abi.Result)]),
    "spg_decimate_global": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int]),
    "spg_decimate_online": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int]),
    "spg_decimate_cluster": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int]),
    "spg_graph_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "spg_graph_destroy": (None, [C.c_void_p]),
    "spg_graph_load_g2o": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "spg_graph_write_g2o": (C.c_int, [C.c_void_p, C.c_char_p]),
    "spg_graph_write_g2o_mem": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "spg_graph_clone_portion": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "spg_graph_covariance": (C.c_int64, [C.c_void_p, C.c_int32, _f64p, C.c_int64]),
    "spg_graph_vertex_edges": (C.c_int, [C.c_void_p, C.c_int, _i32p, C.c_int]),
    "spg_graph_add_vertex": (C.c_int, [C.c_void_p, C.c_int, _f64p]),
    "spg_graph_add_edge": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _f64p, _f64p]),
    "spg_graph_add_vertices": (C.c_int, [C.c_void_p, C.c_int, _i32p, _f64p]),
    "spg_graph_add_edges": (C.c_int, [C.c_void_p, C.c_int, _i32p, _f64p]),
    "spg_graph_add_glc_edge": (C.c_int, [C.c_void_p, C.c_int, _i32p, C.c_int, _f64p, _f64p]),
    "spg_graph_add_multi_edge": (C.c_int, [C.c_void_p, C.c_int, _i32p, _f64p, C.c_int64]),
    "spg_graph_pose_dim": (C.c_int, [C.c_void_p]),
    "spg_graph_num_vertices": (C.c_int, [C.c_void_p]),
    "spg_graph_num_edges": (C.c_int, [C.c_void_p]),
    "spg_graph_edge_data_size": (C.c_int64, [C.c_void_p]),
    "spg_graph_edge_vert_size": (C.c_int64, [C.c_void_p]),
    "spg_graph_get_vertices": (C.c_int, [C.c_void_p, _i32p, _f64p]),
    "spg_graph_get_edges": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _i64p, _f64p]),
    "spg_graph_set_estimate": (C.c_int, [C.c_void_p, C.c_int, _f64p]),
    "spg_graph_marginalize": (C.c_int, [C.c_void_p, _i32p, C.c_int, C.POINTER(abi.Options), C.POINTER(abi.MargStats)]),
    "spg_graph_marginalize_ranks": (C.c_int, [C.c_void_p, _i32p, C.c_int, C.POINTER(abi.Options), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(abi.MargStats)]),
    "spg_graph_last_blanket_count": (C.c_int, [C.c_void_p]),
    "spg_graph_last_blankets": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _i32p, _f64p, _f64p]),
    "spg_graph_substitute_edge": (C.c_int, [C.c_void_p, _i32p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _f64p, _f64p]),
    "spg_graph_information": (C.c_int64, [C.c_void_p, C.c_int32, _f64p, C.c_int64]),
    "spg_graph_kullback_leibler": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(abi.KldTerms)]),
    "spg_ctx_set_linear_solver": (C.c_int, [C.c_void_p, C.c_int]),
    "spg_sparse_plan": (C.c_int, [C.c_int, _i32p, _i32p, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.POINTER(abi.SparsePlanInfo),
                                  _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, C.c_int64]),
    "spg_graph_optimize": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.POINTER(abi.OptimizeStats)]),
    "spg_graph_optimize_fixed": (C.c_int, [C.c_void_p, C.c_int, _i32p, C.c_int, C.POINTER(abi.OptimizeStats)]),
    "spg_graph_chi2": (C.c_int, [C.c_void_p, _f64p]),
    "spg_graph_marginalize_begin": (C.c_int, [C.c_void_p, _i32p, C.c_int, C.POINTER(abi.Options), C.c_int, C.c_int]),
    "spg_graph_set_shard_threshold": (C.c_int, [C.c_void_p, C.c_int]),
    "spg_graph_set_stream_emulation": (C.c_int, [C.c_void_p, C.c_int]),
    "spg_graph_round_prepare": (C.c_int, [C.c_void_p, C.POINTER(abi.RoundInfo)]),
    "spg_graph_round_compute": (C.c_int, [C.c_void_p]),
    "spg_graph_round_commit": (C.c_int, [C.c_void_p]),
    "spg_graph_marginalize_end": (C.c_int, [C.c_void_p, C.POINTER(abi.MargStats)]),
    "spg_graph_arena": (C.c_void_p, [C.c_void_p, _i64p]),
    "spg_graph_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
}


def build(force=False):
    """Compile libspg_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-C", csrc])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        # One HIP runtime per process: torch wheels bundle their own libamdhip64.so.7 (same SONAME as
        # /opt/rocm's). Whichever copy is loaded first serves both, and torch does not find its GPUs
        # on the system copy — so in a process that also uses torch (bench.py, parallel.py: RCCL and
        # device synchronisation) torch must be loaded before libspg_hip.so.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # raises AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class SpgError(RuntimeError):
    pass


def check(rc, ctx=None, what=""):
    if rc < 0:
        msg = ""
        if ctx:
            m = load().spg_last_error(ctx)
            msg = m.decode() if m else ""
        raise SpgError(f"{what} failed with code {rc}: {msg}")
    return rc


def get_unique_id():
    """spg_get_unique_id: the 128 bytes that identify an RCCL communicator (rank 0 calls this, every rank gets the bytes)."""
    buf = C.create_string_buffer(128)
    rc = load().spg_get_unique_id(buf)
    if rc != 0:
        raise SpgError(f"spg_get_unique_id failed with code {rc}")
    return bytes(buf.raw)


class Context:
    """spg_ctx: one per (host thread, device). Raises if no gfx950 device is available."""

    def __init__(self, device=0, _handle=None, _keep=None):
        self.L = load()
        self._keep = _keep
        if _handle is not None:
            self.h = _handle
            return
        h = C.c_void_p()
        rc = self.L.spg_ctx_create(C.byref(h), int(device))
        if rc != 0:
            raise SpgError(f"spg_ctx_create(device={device}) failed with code {rc}: no usable gfx950 device "
                           "(the product path has no CPU fallback)")
        self.h = h

    @classmethod
    def ranks(cls, device, rank, nranks, unique_id):
        """spg_ctx_create_ranks (include/spg.h): one process per GPU, RCCL communicator inside the library; `unique_id` =
        the bytes rank 0 got from get_unique_id(), handed to every rank out of band. Collective over the ranks."""
        L = load()
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        rc = L.spg_ctx_create_ranks(C.byref(h), int(device), int(rank), int(nranks), buf)
        if rc != 0:
            raise SpgError(f"spg_ctx_create_ranks(device={device}, rank={rank}/{nranks}) failed with code {rc}")
        return cls(_handle=h)

    def nranks(self):
        return int(self.L.spg_ctx_nranks(self.h))

    @classmethod
    def injected(cls, backend, keep=None):
        """Test seam: bind a caller-supplied compute backend (see include/spg.h `spg_backend`)."""
        L = load()
        h = C.c_void_p()
        rc = L.spg_ctx_create_injected(C.byref(h), C.byref(backend))
        if rc != 0:
            raise SpgError(f"spg_ctx_create_injected failed: {rc}")
        return cls(_handle=h, _keep=(backend, keep))

    def stream(self):
        return self.L.spg_ctx_stream(self.h)

    def set_linear_solver(self, solver):
        """abi.SOLVER_AUTO / SOLVER_DENSE / SOLVER_SPARSE for optimize() and the global KLD of this context's graphs."""
        check(self.L.spg_ctx_set_linear_solver(self.h, int(solver)), self.h, "spg_ctx_set_linear_solver")

    def synchronize(self):
        check(self.L.spg_ctx_synchronize(self.h), self.h, "spg_ctx_synchronize")

    def profile(self, enable=True):
        check(self.L.spg_ctx_profile(self.h, int(enable)), self.h, "spg_ctx_profile")

    def profile_read(self):
        ms, by = C.c_double(), C.c_double()
        nl, nb = C.c_int64(), C.c_int64()
        check(self.L.spg_ctx_profile_read(self.h, C.byref(ms), C.byref(by), C.byref(nl), C.byref(nb)), self.h, "spg_ctx_profile_read")
        return {"kernel_ms": ms.value, "alg_bytes": by.value, "launches": nl.value, "blankets": nb.value}

    def profile_read_worker(self):
        ms, by = C.c_double(), C.c_double()
        nl, nb = C.c_int64(), C.c_int64()
        check(self.L.spg_ctx_profile_read_worker(self.h, C.byref(ms), C.byref(by), C.byref(nl), C.byref(nb)), self.h, "spg_ctx_profile_read_worker")
        return {"kernel_ms": ms.value, "alg_bytes": by.value, "runs": nl.value, "blankets": nb.value}

    def profile_read_big(self):
        ms, fl = C.c_double(), C.c_double()
        nb, nm = C.c_int64(), C.c_int32()
        check(self.L.spg_ctx_profile_read_big(self.h, C.byref(ms), C.byref(fl), C.byref(nb), C.byref(nm)), self.h, "spg_ctx_profile_read_big")
        return {"kernel_ms": ms.value, "flops": fl.value, "blankets": nb.value, "n_max": nm.value}

    def marginalize_batch(self, opts, batch, want_target=True):
        return abi.marginalize_batch(self.L, self.h, opts, batch, want_target)

    def _track(self, graph):
        """Graphs created on this context (weak references): destroyed before the context is."""
        import weakref
        if not hasattr(self, "_graphs"):
            self._graphs = weakref.WeakSet()
        self._graphs.add(graph)

    def close(self):
        if getattr(self, "h", None):
            for g in list(getattr(self, "_graphs", ())):
                try:
                    g.close()
                except Exception:
                    pass
            self.L.spg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sparse_plan(adj_ptr, adj, pose_dim, is_marg=None, leaf=0):
    """spg_sparse_plan (include/spg.h): the symbolic phase of the block-sparse solver (host only, no GPU needed)."""
    import numpy as np
    L = load()
    adj_ptr = np.ascontiguousarray(adj_ptr, np.int32)
    adj = np.ascontiguousarray(adj, np.int32)
    n = len(adj_ptr) - 1
    im = None if is_marg is None else np.ascontiguousarray(is_marg, np.uint8)
    imp = None if im is None else im.ctypes.data_as(C.POINTER(C.c_uint8))
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
    info = abi.SparsePlanInfo()
    rc = L.spg_sparse_plan(n, i32(adj_ptr), i32(adj), pose_dim, imp, leaf, C.byref(info), None, None, None, None, None, None, None, 0)
    if rc:
        raise SpgError(f"spg_sparse_plan failed: {rc}")
    nsn = info.n_supernodes
    out = {"perm": np.zeros(n, np.int32), "first": np.zeros(nsn + 1, np.int32), "parent": np.zeros(nsn, np.int32),
           "level": np.zeros(nsn, np.int32), "rowptr": np.zeros(nsn + 1, np.int32),
           "rows": np.zeros(max(info.n_rows, 1), np.int32), "rel": np.zeros(max(info.n_rows, 1), np.int32)}
    rc = L.spg_sparse_plan(n, i32(adj_ptr), i32(adj), pose_dim, imp, leaf, C.byref(info), i32(out["perm"]), i32(out["first"]),
                           i32(out["parent"]), i32(out["level"]), i32(out["rowptr"]), i32(out["rows"]), i32(out["rel"]), max(info.n_rows, 1))
    if rc:
        raise SpgError(f"spg_sparse_plan failed: {rc}")
    out["rows"] = out["rows"][:info.n_rows]
    out["rel"] = out["rel"][:info.n_rows]
    out.update(n_marg_supernodes=info.n_marg_supernodes, n_levels=info.n_levels, front_bytes=info.front_bytes, flops=info.flops)
    return out
