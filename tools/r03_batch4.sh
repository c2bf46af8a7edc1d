cd $GRAFT_REPO_ROOT
timeout -k 10 200 ./exp/fetch_pattern 2>&1 | grep floor
tools/ab_bench.sh cornell 5 craytracer_amd/csrc/libcray_hip.so exp/shade3.so 2>&1
tools/ab_bench.sh dragon 3 craytracer_amd/csrc/libcray_hip.so exp/shade3.so 2>&1
tools/ab_bench.sh staircase 2 craytracer_amd/csrc/libcray_hip.so exp/shade3.so 2>&1
