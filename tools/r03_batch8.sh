cd $GRAFT_REPO_ROOT
hipcc -std=c++17 -O2 -fPIC -shared -o /tmp/libmock_rccl.so tests/mock_rccl/mock_rccl.cpp -lrt 2>/dev/null
CRAY_RCCL_LIB=/tmp/libmock_rccl.so timeout -k 10 600 python tools/mock8_frame.py --world 8 > gpurun_out/r03_mock8.log 2>&1; echo "mock8 rc=$?"; tail -3 gpurun_out/r03_mock8.log
python tools/shard_timing.py --worlds 1,2,4,8 --reps 3 > gpurun_out/r03_shard_rehearsal.log 2>&1; python - <<'PY'
import json
for l in open('gpurun_out/r03_shard_rehearsal.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['world'], d['slowest_rank_ms'], d['mean_rank_ms'], d['speedup_vs_1'])
PY
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_default.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['achieved'], r['traffic'], r['avg_launch_ms'], r.get('measured_stream_read_GBs'), r['k_shade'], d['cpu_baseline']['value'])
PY
for wl in cornell staircase; do python bench.py --workload $wl --steps 3 --warmup 1 --cpu-baseline 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"; done
python -m pytest tests -m gpu -x -q > gpurun_out/r03_b8_tests.log 2>&1; tail -3 gpurun_out/r03_b8_tests.log
