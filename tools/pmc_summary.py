#!/usr/bin/env python3
"""Summary of the counter passes of tools/profile_round.sh: per kernel family the sums of the SQ / TCC / TA counters and the
derived figures quoted in DESIGN.md (VALU lane utilisation, wait share, L2 hit rate, TA busy).
usage: pmc_summary.py <gpurun_out/prof_tag dir> <out.json>"""
import json, os, sys
import pandas as pd

d, out = sys.argv[1], sys.argv[2]
res = {}
for f in ('pmc_sq.csv', 'pmc_tcc.csv', 'pmc_ta.csv', 'pmc_tcp.csv'):
    p = os.path.join(d, f)
    if not os.path.exists(p):
        continue
    df = pd.read_csv(p)
    df['k'] = df['Kernel_Name'].str.extract(r'(k_\w+(?:<[^>]*>)?)')
    df = df[df['k'].notna() & ~df['Kernel_Name'].str.contains('bvhb')]
    piv = df.pivot_table(index='k', columns='Counter_Name', values='Counter_Value', aggfunc='sum')
    n = df.groupby('k')['Dispatch_Id'].nunique()
    for k, row in piv.iterrows():
        e = res.setdefault(k, {'dispatches': int(n[k])})
        e.update({c: float(v) for c, v in row.items()})
for k, e in res.items():
    if 'SQ_THREAD_CYCLES_VALU' in e and e.get('SQ_ACTIVE_INST_VALU'):
        e['valu_lane_utilisation'] = round(e['SQ_THREAD_CYCLES_VALU'] / (64.0 * e['SQ_ACTIVE_INST_VALU']), 3)
    if 'SQ_WAIT_ANY' in e and e.get('SQ_WAVE_CYCLES'):
        e['wait_any_share_of_wave_cycles'] = round(e['SQ_WAIT_ANY'] / e['SQ_WAVE_CYCLES'], 3)
    if e.get('SQ_INSTS_VALU'):
        e['salu_per_valu'] = round(e.get('SQ_INSTS_SALU', 0.0) / e['SQ_INSTS_VALU'], 3)
    if 'TCC_HIT_sum' in e:
        e['tcc_hit_rate'] = round(e['TCC_HIT_sum'] / (e['TCC_HIT_sum'] + e['TCC_MISS_sum']), 3)
    if 'TA_TA_BUSY_sum' in e and e.get('GRBM_GUI_ACTIVE'):
        # TA_TA_BUSY_sum adds the 256 TAs, GRBM_GUI_ACTIVE adds the 8 XCDs
        e['ta_busy'] = round((e['TA_TA_BUSY_sum'] / 256.0) / (e['GRBM_GUI_ACTIVE'] / 8.0), 3)
    if e.get('TCP_TCC_READ_REQ_sum') and e.get('GRBM_GUI_ACTIVE'):
        # read requests of the 256 L1s to L2 per clock and CU: what a CU fills its L1 with (MI355X_MICROARCH.md: an HBM-bound
        # global_load_dwordx4 stream gets ~10 B per clock and CU).  On gfx950 the counter tallies 128-B requests — like FETCH_SIZE,
        # which reports half the bytes: k_film, a pure stream of 3.2 GB, shows 2.51e7 requests = 3.2 GB at 128 B each
        # (profiles/r03_v4: FETCH_SIZE x 2 = TCC_MISS x 128 B = TCP_TCC_READ_REQ x 128 B).
        e['l1_fill_bytes_per_clk_per_cu'] = round(e['TCP_TCC_READ_REQ_sum'] * 128.0 / 256.0 / (e['GRBM_GUI_ACTIVE'] / 8.0), 2)
    if e.get('SQ_INSTS_VALU') and e.get('GRBM_GUI_ACTIVE'):
        # a wave64 VALU instruction occupies its SIMD for >= 4 cycles (more for the quarter-rate f64 ops): a LOWER bound of the
        # share of time the 1024 SIMDs spend issuing VALU work
        e['valu_issue_share_min'] = round(e['SQ_INSTS_VALU'] * 4.0 / (1024.0 * e['GRBM_GUI_ACTIVE'] / 8.0), 3)
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
for k, e in sorted(res.items()):
    print(k, {x: e[x] for x in ('dispatches', 'valu_lane_utilisation', 'wait_any_share_of_wave_cycles', 'salu_per_valu', 'tcc_hit_rate', 'ta_busy', 'SQ_INSTS_VALU') if x in e})
