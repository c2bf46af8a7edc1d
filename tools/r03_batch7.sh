cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_golden_films.py tests/test_gpu_resident.py tests/test_gpu_fast_mode.py tests/test_gpu_hybrid.py -x -q > gpurun_out/r03_b7_tests.log 2>&1 || { tail -40 gpurun_out/r03_b7_tests.log; exit 1; }
tail -2 gpurun_out/r03_b7_tests.log
tools/ab_bench.sh dragon 3 craytracer_amd/csrc/libcray_hip.so exp/slot80.so
tools/ab_bench.sh staircase 2 craytracer_amd/csrc/libcray_hip.so exp/slot80.so
tools/ab_bench.sh cornell 5 craytracer_amd/csrc/libcray_hip.so exp/slot80.so
echo "== 5 waves per SIMD (exp/trace5.so, 80-B slots) vs exp/slot80.so"
CRAY_TRACE_BLOCKS_PER_CU=5 tools/ab_bench.sh dragon 2 exp/trace5.so
tools/ab_bench.sh dragon 2 exp/slot80.so
