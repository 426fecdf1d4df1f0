#!/bin/bash
# tools/ab_env.sh VAR v1 v2 ... — bench.py with environment variable VAR set to each value in turn, on one box, three rounds
var=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    env $var=$v timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /tmp/ab_out.json 2>/dev/null
    python -c "import json,sys; d=json.load(open('/tmp/ab_out.json')); print('$var=$v', round(d['ms_per_step'],3), round(1e3*d['config']['host_seconds_per_step'],3), round(1e3*d['config']['device_wait_seconds_per_step'],3), d['config']['batches'])"
  done
done
