set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# 1. parity of the changed kernel
python -m pytest tests/test_gpu_parity.py tests/test_golden_films.py -x -q > gpurun_out/r03_b1_tests.log 2>&1 || { tail -30 gpurun_out/r03_b1_tests.log; exit 1; }
tail -3 gpurun_out/r03_b1_tests.log
# 2. A/B: encoded keys (product) vs plain keys
tools/ab_bench.sh dragon 3 craytracer_amd/csrc/libcray_hip.so exp/keyplain.so > gpurun_out/r03_b1_ab.log 2>&1
cat gpurun_out/r03_b1_ab.log
# 3. per-bounce queue lengths + per-dispatch durations of a whole frame
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_frame_trace -o run -- python3 tools/share_trace.py --world 1 --rank 0 --frames 2 > gpurun_out/r03_frame_trace.log 2>&1
python3 tools/share_trace.py --report gpurun_out/r03_frame_trace > gpurun_out/r03_frame_trace.txt 2>&1
CRAY_LOG_QUEUES=1 python3 tools/share_trace.py --world 1 --rank 0 --frames 1 2> gpurun_out/r03_queues_w1.log
# 4. the same for rank 0's 1/8 share
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_share_trace -o run -- python3 tools/share_trace.py --world 8 --rank 0 --frames 3 > gpurun_out/r03_share_trace.log 2>&1
python3 tools/share_trace.py --report gpurun_out/r03_share_trace > gpurun_out/r03_share_trace.txt 2>&1
CRAY_LOG_QUEUES=1 python3 tools/share_trace.py --world 8 --rank 0 --frames 1 2> gpurun_out/r03_queues_w8.log
rm -rf gpurun_out/r03_frame_trace gpurun_out/r03_share_trace
cat gpurun_out/r03_share_trace.txt
