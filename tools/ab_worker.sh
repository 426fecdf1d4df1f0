#!/bin/bash
# A/B of the pipelined driver's variants on the bench workload (diagnostic; prints ms/step and the trace line per variant)
run() { # name, env...
  name=$1; shift
  env "$@" SPG_TRACE=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/ab_{n}.json")); c=d["config"]
    print(f"{n:28s} {d['ms_per_step']:7.2f} ms/step host {1e3*c['host_seconds_per_step']:.1f} (sched {1e3*c['schedule_seconds_per_step']:.1f} commit {1e3*c['commit_seconds_per_step']:.1f}) wait {1e3*c['device_wait_seconds_per_step']:.1f} launch {1e3*c['launch_seconds_per_step']:.1f} batches {c.get('batches')} launches {c.get('kernel_launches')}")
except Exception as e:
    print(n, "FAILED", e)
PY
  grep -E "spg trace|sched " gpurun_out/ab_$name.err | tail -${TAILN:-1}
}
for v in "$@"; do
  name=$(echo "$v" | tr ' =' '__')
  run "$name" $v
done
