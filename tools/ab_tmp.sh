set -e
run() { CRAY_HYBRID=$3 CRAY_LIB=$1 python bench.py --workload $2 --steps $4 --warmup 1 --cpu-baseline 0 --count-pass 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%-14s hyb=%s %-10s frame %8.2f  trace %8.2f (b0 %6.2f mixed %7.2f)  shade %7.2f' % ('$1', '$3', '$2', d['ms_per_step'], k['trace'], k['trace_closest_bounce0'], k['trace_mixed'], k['shade']))"; }
for rep in 1 2; do
run exp/base.so dragon 0 3; run "" dragon 0 3; run "" dragon 1 3
done
run exp/base.so staircase 0 2; run "" staircase 0 2; run "" staircase 1 2
run exp/base.so cornell 0 5; run "" cornell 0 5; run "" cornell 1 5
run exp/base.so cornell 0 5; run "" cornell 0 5
