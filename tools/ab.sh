#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab.sh <base.so> [runs]   (diagnostic)
BASE=$1; N=${2:-3}
for i in $(seq $N); do for v in base new; do
  if [ $v = base ]; then export SPG_LIB_PATH=$BASE; else unset SPG_LIB_PATH; fi
  timeout -k 10 100 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('$v', round(d['value']), round(d['ms_per_step'],2), {k[:8]:round(c[k]*1e3,1) for k in c if 'seconds' in k})"
done; done
