set -e
python -m pytest tests/test_gpu_parity.py tests/test_golden_films.py tests/test_gpu_cry_scenes.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_bench.sh dragon 3 exp/base.so craytracer_amd/csrc/libcray_hip.so
bash tools/ab_bench.sh staircase 2 exp/base.so craytracer_amd/csrc/libcray_hip.so
