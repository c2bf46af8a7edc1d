cd $GRAFT_REPO_ROOT
for envs in "X=0" "CRAY_REFILL_MIN_B0=48" "CRAY_REFILL_MIN_B0=32" "CRAY_REFILL_MIN_B0=16" "CRAY_TRACE_BLOCKS_PER_CU=2" "CRAY_TRACE_BLOCKS_PER_CU=8" "CRAY_REFILL_MIN=16 CRAY_REFILL_MIN_ANY=24" "CRAY_REFILL_MIN=40 CRAY_REFILL_MIN_ANY=48" "CRAY_SHADE_BLOCKS_PER_CU=2" "CRAY_SHADE_BLOCKS_PER_CU=8"; do
  echo "== $envs"
  env $envs python tools/shard_timing.py --worlds 8 --ranks 0,3 --reps 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        for r in d['ranks']: print('  rank %d: %.2f ms  closest %.2f mixed %.2f any %.2f shade %.2f other %.2f' % (r['rank'], r['ms'], r['closest_ms'], r['mixed_ms'], r['any_ms'], r['shade_ms'], r['other_ms']))"
done
