#!/bin/bash
# tools/ab_safe.sh <log> <workload> <steps> <reps> "<ENV=.. ENV=..>" ...: tools/ab_cfg.sh with every bench run under its own timeout, so that an
# experimental kernel that never finishes costs two minutes and not the call (a killed run prints HUNG and the script stops)
log=$1; wl=$2; steps=$3; reps=$4; shift 4
for rep in $(seq 1 $reps); do
  for envs in "$@"; do
    out=$(env $envs timeout -k 10 150 python bench.py --workload $wl --steps $steps --warmup 1 --cpu-baseline 0 --count-pass 0 2>/dev/null)
    if [ -z "$out" ]; then echo "HUNG or failed: $envs" | tee -a $log; exit 1; fi
    echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%-64s %-10s frame %8.2f  trace %8.2f (b0 %6.2f mixed %7.2f any %5.2f)  shade %7.2f' % ('$envs', '$wl', d['ms_per_step'], k['trace'], k['trace_closest_bounce0'], k['trace_mixed'], k['trace_any_last_bounce'], k['shade']))" | tee -a $log
  done
done
