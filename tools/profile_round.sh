#!/bin/bash
# tools/profile_round.sh <tag> [workload]: the rocprofv3 evidence of a round, on the GPU box.
#   1. --kernel-trace --stats over the default bench command (3 timed frames): per-kernel average duration
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, each in its OWN pass (TCC slots: 3 + 2 of 4) over one frame: HBM traffic
#   3. --pmc SQ pass (lane utilisation, waits), a TCC hit-rate pass and a TA pass over one frame
# Counter passes never carry trace options other than the implicit kernel dispatch records (gpurun refuses mixes).
# Results: gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_*.csv,bench.log,pmc_summary.json}.  Back in the build container:
#   tools/adopt_profile.sh <tag> [workload]   copies them to profiles/<tag>/ and rewrites profiles/hbm_traffic.json,
# keyed by the source hash of the library that was profiled (bench.py refuses the traffic figure for any other build).
set -e
tag=$1; wl=${2:-dragon}
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# The library times a scene's first two frames to choose the records its traversal launches read (f64 or certified f32 culling, per
# launch kind): a plain bench run first says what it chose; every profiled pass below is PINNED to that choice, so that the one frame
# of a counter pass and the timed frames of the stats pass run the steady-state kernels and nothing else.
python3 bench.py --workload $wl --steps 2 --warmup 2 --cpu-baseline 0 --count-pass 0 > $out/choice.log 2> $out/choice.err
pin=$(python3 - "$out/choice.log" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d['kernel_ms_per_step']['trace_records']
v = lambda n: '1' if n.startswith('certified') else '0'
print('CRAY_RECORDS_B0=%s CRAY_RECORDS_REST=%s' % (v(r['bounce0']), v(r['other_launches'])))
PY
)
echo "records pinned for the profiled passes: $pin" | tee $out/pin.txt
export $pin
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --workload $wl --steps 3 --warmup 1 --cpu-baseline 0 > $out/bench.log 2> $out/stats.log
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
for pass in "fetch_size:FETCH_SIZE" "write_size:WRITE_SIZE" "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" "tcc:TCC_HIT_sum TCC_MISS_sum" "ta:TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $out/pmc_$name -o run -- python3 bench.py --workload $wl --steps 1 --warmup 0 --cpu-baseline 0 --count-pass 0 > $out/pmc_$name.bench.log 2> $out/pmc_$name.log || echo "pass $name failed" >> $out/failed.txt
  f=$(find $out/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/pmc_$name.csv
  rm -rf $out/pmc_$name
done
rm -rf $out/stats
python3 tools/pmc_summary.py $out $out/pmc_summary.json > $out/pmc_summary.txt 2>&1 || true
ls -la $out
