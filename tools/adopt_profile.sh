#!/bin/bash
# tools/adopt_profile.sh <tag> [workload]: gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) ->
# profiles/<tag>/ + profiles/hbm_traffic.json[workload], keyed by the source hash of the profiled library.
set -e
tag=$1; wl=${2:-dragon}
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
src=gpurun_out/prof_$tag
dst=profiles/$tag
mkdir -p $dst
# the per-dispatch counter files are tens of MB with the BVH builder's thousands of launches: keep the k_* render kernels only
for f in pmc_fetch_size pmc_write_size pmc_sq pmc_tcc pmc_ta pmc_tcp; do
  [ -f $src/$f.csv ] && python3 - "$src/$f.csv" "$dst/$f.csv" <<'PY'
import sys, pandas as pd
df = pd.read_csv(sys.argv[1])
keep = df['Kernel_Name'].str.contains(r'cray::k_(?:trace|shade|raygen|film|resolve)')
cols = [c for c in ('Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'Counter_Name', 'Counter_Value') if c in df.columns]
df[keep][cols].to_csv(sys.argv[2], index=False)
PY
done
cp $src/kernel_stats.csv $src/bench.log $dst/
python3 tools/pmc_summary.py $dst $dst/pmc_summary.json > /dev/null
python3 tools/traffic_from_pmc.py $wl $dst/pmc_fetch_size.csv $dst/pmc_write_size.csv $dst/pmc_summary.json $dst/bench.log > /dev/null
python3 - <<PY
import json
e = json.load(open('profiles/hbm_traffic.json'))['$wl']
print('$wl', 'source_hash', (e.get('source_hash') or '')[:16], 'trace GB/frame', e['trace_bytes_per_frame'] / 1e9, 'shade GB/frame', e['shade_bytes_per_frame'] / 1e9)
PY
