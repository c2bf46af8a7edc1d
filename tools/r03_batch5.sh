cd $GRAFT_REPO_ROOT
CRAY_LOG_QUEUES=1 python3 tools/share_trace.py --world 1 --rank 0 --frames 1 --count 2 2> gpurun_out/r03_counts_w1.log
grep cray gpurun_out/r03_counts_w1.log
python -m pytest tests -m gpu -x -q > gpurun_out/r03_b5_tests.log 2>&1; tail -5 gpurun_out/r03_b5_tests.log
