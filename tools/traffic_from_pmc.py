#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a rocprofv3 `--pmc FETCH_SIZE` (and optionally WRITE_SIZE) pass.

HBM bytes = FETCH_SIZE[KB] * 1024 * 2: on gfx950 FETCH_SIZE counts 128-B requests as 64 B
(MI355X_MICROARCH.md "HBM"); tools/calib_fetch.hip confirms the factor for THIS access pattern
(random whole 128-B records, 16 B per lane per load: 1.707 GB reported for 3.322 GB read; a
coalesced 16-B/lane stream: 1.611 GB reported for 3.221 GB read).  WRITE_SIZE is taken as is.
usage: traffic_from_pmc.py <workload> <fetch counter_collection.csv> [<write counter_collection.csv> [<pmc_summary.json> [<bench.log>]]]
The bench.log of the profiled run (its JSON line) names the source hash of the library that was profiled: bench.py only uses
the traffic figure when that hash equals the hash of the library it runs.
(the summary of tools/pmc_summary.py adds, per traversal / shading kernel, how busy the other units were: VALU issue, texture
addresser, share of wave-cycles spent waiting)"""
import json
import os
import sys

import pandas as pd

FETCH_CORRECTION = 2.0


def per_kernel(csv, counter):
    df = pd.read_csv(csv)
    df = df[df['Counter_Name'] == counter]
    df['k'] = df['Kernel_Name'].str.extract(r'(k_\w+(?:<[^>]*>)?)')
    g = df.groupby('k')['Counter_Value']
    return g.sum().to_dict(), g.count().to_dict()


if __name__ == '__main__':
    workload, fetch_csv = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'profiles', 'hbm_traffic.json')
    data = json.load(open(path)) if os.path.exists(path) else {}
    fs, fc = per_kernel(fetch_csv, 'FETCH_SIZE')
    import subprocess
    try:
        git = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=root, text=True).strip()
        dirty = bool(subprocess.check_output(['git', 'status', '--porcelain', '--', 'craytracer_amd/csrc'], cwd=root, text=True).strip())
        git += '+dirty' if dirty else ''
    except Exception:
        git = None
    prev = data.get(workload, {})
    if prev.get('source') == os.path.relpath(fetch_csv, root) and prev.get('git'):
        git = prev['git']   # the same counter file processed again: keep the hash of the build that was profiled
    entry = {'source': os.path.relpath(fetch_csv, root), 'git': git, 'fetch_correction': FETCH_CORRECTION, 'kernels': {}}
    for k in fs:
        entry['kernels'][k] = {'launches': int(fc[k]), 'fetch_bytes_per_launch': fs[k] * 1024 * FETCH_CORRECTION / fc[k]}
    if len(sys.argv) > 3:
        ws, wc = per_kernel(sys.argv[3], 'WRITE_SIZE')
        for k in ws:
            entry['kernels'].setdefault(k, {})['write_bytes_per_launch'] = ws[k] * 1024 / wc[k]
    # the traversal family (k_trace<closest>, k_trace_mixed, k_trace<any>): HBM bytes per launch over all its launches
    tot, launches = 0.0, 0
    for k, v in entry['kernels'].items():
        if k.startswith('k_trace'):
            tot += (v.get('fetch_bytes_per_launch', 0) + v.get('write_bytes_per_launch', 0)) * v['launches']
            launches += v['launches']
    entry['mixed_bytes_per_frame'] = round(sum((v.get('fetch_bytes_per_launch', 0) + v.get('write_bytes_per_launch', 0)) * v['launches']
                                               for k, v in entry['kernels'].items() if k.startswith('k_trace_mixed')))
    entry['trace_launches'] = launches                       # of the profiled run: ONE frame (bench.py --steps 1 --warmup 0)
    entry['trace_bytes_per_launch'] = round(tot / max(1, launches))
    entry['trace_bytes_per_frame'] = round(tot)              # what bench.py divides by ITS launches per frame
    sh = [v for k, v in entry['kernels'].items() if k.startswith('k_shade')]
    entry['shade_bytes_per_frame'] = round(sum((v.get('fetch_bytes_per_launch', 0) + v.get('write_bytes_per_launch', 0)) * v['launches'] for v in sh))
    if len(sys.argv) > 4:
        summ = json.load(open(sys.argv[4]))
        entry['units'] = {k: {f: v[f] for f in ('valu_issue_share_min', 'ta_busy', 'wait_any_share_of_wave_cycles', 'valu_lane_utilisation', 'tcc_hit_rate', 'l1_fill_bytes_per_clk_per_cu') if f in v}
                          for k, v in summ.items() if k.startswith('k_trace') or k.startswith('k_shade')}
        entry['units_source'] = os.path.relpath(sys.argv[4], root)
    if len(sys.argv) > 5:
        for line in open(sys.argv[5]):
            if line.startswith('{'):
                entry['source_hash'] = json.loads(line)['config']['build'].get('source_hash')
                entry['kernel_hash'] = json.loads(line)['config']['build'].get('kernel_hash')
                # the records the profiled passes were pinned to (tools/profile_round.sh): bench.py refuses the figure for frames that read others
                entry['records'] = json.loads(line).get('kernel_ms_per_step', {}).get('trace_records')
    data[workload] = entry
    json.dump(data, open(path, 'w'), indent=1)
    print(json.dumps(entry, indent=1))
