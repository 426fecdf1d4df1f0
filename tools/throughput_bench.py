"""Diagnostic: the blanket kernel in throughput mode — B independent first-round blankets of the
synthetic SE3 graph in ONE launch (spg_marginalize_batch), HIP-event kernel time. Shows what the
kernel sustains when a launch fills the chip, as opposed to the latency-bound rounds of the
end-to-end workload. Not part of the product or the tests."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.lib import Context
from tests import util


def tile(batch, t):
    """t copies of every blanket of the batch (CSR offsets shifted)."""
    def rep_off(off, dtype):
        off = np.asarray(off)
        n = off[-1]
        return np.concatenate([off[:-1] + i * n for i in range(t)] + [[t * n]]).astype(dtype)
    out = dict(batch)
    out["vert_off"] = rep_off(batch["vert_off"], np.int32)
    out["edge_off"] = rep_off(batch["edge_off"], np.int32)
    out["edge_vert_off"] = rep_off(batch["edge_vert_off"], np.int32)
    out["edge_data_off"] = rep_off(batch["edge_data_off"], np.int64)
    for k in ("n_remove", "vert_id", "pose", "edge_kind", "edge_vert", "edge_data"):
        out[k] = np.tile(batch[k], t)
    return out


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [64, 512, 2048, 16384, 131072]
    g = g2o_io.synth_sphere(20000, 400)
    which = np.array([i for i in range(405, 20000 - 405) if i % 2], np.int32)
    base, _ = util.first_round_batch(g, which, None, limit=2048)
    nb0 = len(base["n_remove"])
    ctx = Context(0)
    opts = abi.make_options(6)
    rows = []
    for B in sizes:
        if B <= nb0:
            lim = {k: v for k, v in base.items()}
            # prefix of the base batch
            nv, ne = base["vert_off"][B], base["edge_off"][B]
            lim = {"vert_off": base["vert_off"][:B + 1], "n_remove": base["n_remove"][:B], "vert_id": base["vert_id"][:nv],
                   "pose": base["pose"][:nv * 7], "edge_off": base["edge_off"][:B + 1], "edge_kind": base["edge_kind"][:ne],
                   "edge_vert_off": base["edge_vert_off"][:ne + 1], "edge_vert": base["edge_vert"][:base["edge_vert_off"][ne]],
                   "edge_data_off": base["edge_data_off"][:ne + 1], "edge_data": base["edge_data"][:base["edge_data_off"][ne]]}
            batch = lim
        else:
            batch = tile(base, B // nb0)
        nb = len(batch["n_remove"])
        alg_bytes = 8 * (len(batch["pose"]) + len(batch["edge_data"]))  # inputs; outputs added below
        ctx.marginalize_batch(opts, batch, want_target=False)
        ctx.profile(True)
        reps = 5
        for _ in range(reps):
            res = ctx.marginalize_batch(opts, batch, want_target=False)
        ctx.synchronize()
        p = ctx.profile_read()
        ctx.profile(False)
        us = 1e3 * p["kernel_ms"] / reps
        row = {"blankets": nb, "kernel_us_per_batch": us, "launches_per_batch": p["launches"] / reps,
               "blankets_per_s": nb / (us * 1e-6), "alg_GBps": p["alg_bytes"] / reps / (us * 1e-6) / 1e9,
               "bad": int(np.count_nonzero(res["status"]))}
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
