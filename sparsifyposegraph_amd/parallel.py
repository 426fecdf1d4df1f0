"""Multi-GPU form of marginalizeNoOptimize: one process per GPU, graph replicated, each round's
independent blankets sharded over the ranks, ONE all-gather of the round's output region per round.

The reference has no counterpart (single process, no collective). All ranks run the
same deterministic host scheduler on identical replicas, so the only data that has to move is what
a rank computed for its slice: the recovered edge records and the per-blanket output records. They
live in a rank-chunked region of the arena (`spg_round_info`: region = nranks * chunk_len, rank r
owns chunk r), so the exchange is an in-place all-gather of equal-sized chunks — over RCCL/xGMI on
the GPU box (`torch.distributed` backend "nccl"), over gloo in the CPU tests.

Rounds that are too small to be worth an exchange (latency-bound: a few hundred blankets finish in
one kernel latency however they are split) are computed redundantly by every rank instead
(`spg_round_info.exchange == 0`); the kernels are bit-deterministic, so the replicas stay identical.
"""
import ctypes as C

import numpy as np


class _CudaArena:
    """Zero-copy view of the device arena for torch (CUDA array interface v2)."""

    def __init__(self, ptr, doubles):
        self.__cuda_array_interface__ = {"shape": (int(doubles),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def arena_tensor(graph, device=None):
    """torch tensor aliasing the graph's arena (device memory for the HIP backend, host memory for
    an injected backend)."""
    import torch
    ptr, cap = graph.arena()
    if device is not None and str(device).startswith("cuda"):
        return torch.as_tensor(_CudaArena(ptr, cap), device=device)
    buf = (C.c_double * cap).from_address(ptr)
    return torch.from_numpy(np.ctypeslib.as_array(buf))


def marginalize_sharded(graph, which, opts, device=None, group=None, stepwise=False):
    """Run graph.marginalizeNoOptimize(which) cooperatively on every rank of `group`.

    Every rank must call this with identical arguments on an identical replica. Returns the
    per-rank stats dict (identical on all ranks except timing fields).

    Default: the library's own (pipelined) driver, `spg_graph_marginalize_ranks`, which calls back
    into `exchange` only for batches wide enough to be sharded. `stepwise=True` drives the
    round-stepping ABI from Python instead (one batch at a time; used by tests).
    """
    import torch
    import torch.distributed as dist
    from . import abi
    from .lib import check
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    on_gpu = device is not None and str(device).startswith("cuda")
    state = {"view": None, "ptr": None}

    def all_gather_region(arena_ptr, region_off, chunk_len):
        if arena_ptr != state["ptr"]:  # the arena may be re-allocated while a batch is prepared
            state["view"], state["ptr"] = arena_tensor(graph, device), arena_ptr
        region = state["view"][region_off:region_off + ws * chunk_len]
        mine = region[rank * chunk_len:(rank + 1) * chunk_len]
        if on_gpu and dist.get_backend(group) == "gloo":
            # device arena, host-only collective (several ranks sharing one GPU, no RCCL): stage
            # through host memory. Same data movement, used by the single-GPU sharding test.
            out = torch.empty(ws * chunk_len, dtype=region.dtype)
            dist.all_gather_into_tensor(out, mine.cpu(), group=group)
            region.copy_(out)
        else:
            dist.all_gather_into_tensor(region, mine, group=group)
        if on_gpu:
            torch.cuda.synchronize()

    if not stepwise:
        def _cb(user, arena, region_off, chunk_len, nranks, r):
            try:
                all_gather_region(arena, region_off, chunk_len)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print(f"exchange failed: {e!r}")
                return abi.EHIP
        cb = abi.EXCHANGE_FN(_cb)
        which = np.ascontiguousarray(which, np.int32)
        st = abi.MargStats()
        rc = graph.L.spg_graph_marginalize_ranks(graph.h, which.ctypes.data_as(C.POINTER(C.c_int32)), len(which), C.byref(opts),
                                                 rank, ws, C.cast(cb, C.c_void_p) if ws > 1 else None, None, C.byref(st))
        graph.last_stats = st.asdict()
        check(rc, graph.ctx.h, "spg_graph_marginalize_ranks")
        return graph.last_stats

    graph.begin(which, opts, rank, ws)
    try:
        while True:
            info = graph.round_prepare()
            if info is None:
                break
            graph.round_compute()
            if ws > 1 and info.exchange:
                graph.ctx.synchronize()  # this rank's chunk is complete in memory
                ptr, _ = graph.arena()
                all_gather_region(ptr, info.region_off, info.chunk_len)
            graph.round_commit()
    finally:
        stats = graph.end()
    return stats
