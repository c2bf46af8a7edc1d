#!/bin/bash
# tools/pmc_variants.sh <tag> "<ENV=.. ENV=..>" [workload]: the TCP / SQ / TA counter passes of one frame for an experimental build
# or environment (CRAY_LIB=exp/x.so CRAY_HYBRID=..), into gpurun_out/pmc_<tag>/ with a pmc_summary.json.  Counter passes only
# (no trace options), the program directly after `--`.
tag=$1; envs=$2; wl=${3:-dragon}
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
for pass in "tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" "ta:TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "tcc:TCC_HIT_sum TCC_MISS_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  env $envs rocprofv3 --pmc $ctrs --output-format csv -d $out/pmc_$name -o run -- python3 bench.py --workload $wl --steps 1 --warmup 0 --cpu-baseline 0 --count-pass 0 > $out/pmc_$name.bench.log 2> $out/pmc_$name.log || echo "pass $name failed" >> $out/failed.txt
  f=$(find $out/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" "$out/pmc_$name.csv" <<'PY'
import sys, pandas as pd
df = pd.read_csv(sys.argv[1])
keep = df['Kernel_Name'].str.contains(r'cray::k_(?:trace|shade|raygen|film|resolve)')
cols = [c for c in ('Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'VGPR_Count', 'SGPR_Count', 'Counter_Name', 'Counter_Value') if c in df.columns]
df[keep][cols].to_csv(sys.argv[2], index=False)
PY
  rm -rf $out/pmc_$name
done
python3 tools/pmc_summary.py $out $out/pmc_summary.json > $out/pmc_summary.txt 2>&1 || true
cat $out/pmc_summary.txt
