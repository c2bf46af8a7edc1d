#!/bin/bash
# tools/ab_cfg.sh <workload> <steps> <reps> "<ENV=.. ENV=..>" ["<env set 2>" ...]: kernel times of the bench under different
# environments (CRAY_LIB=exp/x.so selects an experimental build), interleaved <reps> times
wl=$1; steps=$2; reps=$3; shift 3
for rep in $(seq 1 $reps); do
  for envs in "$@"; do
    env $envs python bench.py --workload $wl --steps $steps --warmup 1 --cpu-baseline 0 --count-pass 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%-60s %-10s frame %8.2f  trace %8.2f (b0 %6.2f mixed %7.2f any %5.2f)  shade %7.2f' % ('$envs', '$wl', d['ms_per_step'], k['trace'], k['trace_closest_bounce0'], k['trace_mixed'], k['trace_any_last_bounce'], k['shade']))"
  done
done
