"""Diagnostic: time the blanket kernel truncated after each phase (HIP events), on B first-round
blankets of the synthetic SE3 graph. Not part of the product or the tests."""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifyposegraph_amd import abi, g2o_io
from sparsifyposegraph_amd.lib import Context
from tests import util
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = g2o_io.synth_sphere(40000, 400)
which = np.array([i for i in range(405, 40000 - 405) if i % 2], np.int32)
batch, roots = util.first_round_batch(g, which, None, limit=B)
ctx = Context(0)
names = {1: 'assemble', 2: 'schur', 3: 'chowliu', 4: 'newedges', 5: 'eig', 6: 'sigma', 7: 'closedform', 0: 'full(kld)'}
prev = 0
for stop in [1, 2, 3, 4, 5, 6, 7, 0]:
    opts = abi.make_options(6, flags=(stop << 8))
    ctx.marginalize_batch(opts, batch, want_target=False)
    ctx.profile(True)
    for _ in range(5):
        ctx.marginalize_batch(opts, batch, want_target=False)
    ctx.synchronize()
    p = ctx.profile_read()
    us = 1e3 * p['kernel_ms'] / p['launches']
    print(f"stop={stop} {names[stop]:11s} launches={p['launches']} blankets/launch={p['blankets']/p['launches']:.0f} avg {us:9.1f} us  (+{us-prev:8.1f})")
    prev = us
