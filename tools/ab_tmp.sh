set -e
python -m pytest tests/test_gpu_parity.py tests/test_golden_films.py tests/test_gpu_cry_scenes.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_env.sh dragon 3 "CRAY_SHADE_LDS=0" "CRAY_SHADE_LDS=1"
bash tools/ab_env.sh staircase 2 "CRAY_SHADE_LDS=0" "CRAY_SHADE_LDS=1"
bash tools/ab_env.sh cornell 5 "CRAY_SHADE_LDS=0" "CRAY_SHADE_LDS=1"
