cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03_v1 dragon > gpurun_out/r03_b6_profile.log 2>&1
tail -5 gpurun_out/r03_b6_profile.log
cat gpurun_out/prof_r03_v1/failed.txt 2>/dev/null
CRAY_LIB=exp/diag.so python3 tools/share_trace.py --world 1 --rank 0 --frames 1 2> gpurun_out/r03_diag.log; grep diag gpurun_out/r03_diag.log
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; tail -c 600 gpurun_out/r03_bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default.json')); print(d['value'], d['ms_per_step'], d['cpu_baseline'], d['kernel_ms_per_step'])"
