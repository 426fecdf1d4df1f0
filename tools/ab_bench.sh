#!/bin/bash
# tools/ab_bench.sh libA.so libB.so ... — the same bench.py run against several builds of the library on ONE box, interleaved,
# three rounds (boxes of the pool differ by +-10 %: only same-box comparisons mean anything). Prints ms per step / host seconds.
for round in 1 2 3; do
  for lib in "$@"; do
    SPG_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /tmp/ab_out.json 2>/dev/null
    python -c "import json,sys; d=json.load(open('/tmp/ab_out.json')); print('$lib', round(d['ms_per_step'],3), round(1e3*d['config']['host_seconds_per_step'],3), d['config']['batches'])"
  done
done
