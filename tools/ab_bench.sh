#!/bin/bash
# tools/ab_bench.sh <workload> <steps> <so> [<so> ...]: the bench's kernel times for each experimental library, interleaved twice
wl=$1; steps=$2; shift 2
for rep in 1 2; do
  for so in "$@"; do
    CRAY_LIB=$so python bench.py --workload $wl --steps $steps --warmup 1 --cpu-baseline 0 --count-pass 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%-28s %-10s frame %8.2f  trace %8.2f (b0 %6.2f mixed %7.2f)  shade %7.2f  other %5.2f' % ('$so', '$wl', d['ms_per_step'], k['trace'], k['trace_closest_bounce0'], k['trace_mixed'], k['shade'], k['other']))"
  done
done
