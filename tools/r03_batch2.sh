set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_golden_films.py tests/test_alternatives.py tests/test_gpu_fast_mode.py tests/test_gpu_hybrid.py -x -q > gpurun_out/r03_b2_tests.log 2>&1 || { tail -40 gpurun_out/r03_b2_tests.log; exit 1; }
tail -3 gpurun_out/r03_b2_tests.log
tools/ab_bench.sh dragon 3 craytracer_amd/csrc/libcray_hip.so exp/prestate.so > gpurun_out/r03_b2_ab.log 2>&1
cat gpurun_out/r03_b2_ab.log
tools/ab_bench.sh staircase 2 craytracer_amd/csrc/libcray_hip.so exp/prestate.so > gpurun_out/r03_b2_ab_stair.log 2>&1
cat gpurun_out/r03_b2_ab_stair.log
tools/ab_bench.sh cornell 5 craytracer_amd/csrc/libcray_hip.so exp/prestate.so > gpurun_out/r03_b2_ab_cornell.log 2>&1
cat gpurun_out/r03_b2_ab_cornell.log
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_frame_trace2 -o run -- python3 tools/share_trace.py --world 1 --rank 0 --frames 2 > gpurun_out/r03_frame_trace2.log 2>&1
python3 tools/share_trace.py --report gpurun_out/r03_frame_trace2 > gpurun_out/r03_frame_trace2.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_share_trace2 -o run -- python3 tools/share_trace.py --world 8 --rank 0 --frames 3 > gpurun_out/r03_share_trace2.log 2>&1
python3 tools/share_trace.py --report gpurun_out/r03_share_trace2 > gpurun_out/r03_share_trace2.txt 2>&1
rm -rf gpurun_out/r03_frame_trace2 gpurun_out/r03_share_trace2
grep -v fillBuffer gpurun_out/r03_frame_trace2.txt | tail -24
grep -v fillBuffer gpurun_out/r03_share_trace2.txt | tail -24
