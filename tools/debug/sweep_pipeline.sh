#!/bin/bash
# diagnostic: mean wall time of steps 1..7 of tools/debug/step_overhead.py for a few pipeline parameters
for cfg in ${SWEEP:-"48 12" "48 14" "48 16" "48 18" "48 20" "48 24" "40 16" "56 16" "44 14" "52 18"}; do
  set -- $cfg
  r=$(SPG_SPLIT_MIN=$1 SPG_PATIENCE=$2 python tools/debug/step_overhead.py 2>&1 | grep "^step [1-7]" | awk '{s+=$4; n++} END {printf "%.2f", s/n}')
  echo "split_min $1 patience $2: $r ms"
done
