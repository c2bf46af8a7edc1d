set -e
python -m pytest tests -m gpu -x -q -k "bvh or trace or golden or cornell" 2>&1 | tail -2
bash tools/ab_env.sh dragon 3 "CRAY_LIB=exp/base.so CRAY_HYBRID=0" "CRAY_HYBRID=0" "CRAY_HYBRID=1"
bash tools/ab_env.sh staircase 2 "CRAY_LIB=exp/base.so CRAY_HYBRID=0" "CRAY_HYBRID=0"
bash tools/ab_env.sh cornell 5 "CRAY_LIB=exp/base.so CRAY_HYBRID=0" "CRAY_HYBRID=0"
