"""Render the other BASELINE.json configs on the GPU (parity-test cases, not bench lines):
timing + a sampled check against the CPU oracle.  Usage: python tools/run_configs.py [cornell] [staircase] [simple]"""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from craytracer_amd import backend, scenes
from oracle import oracle_lib as ol

which = sys.argv[1:] or ['simple', 'cornell', 'staircase']
ctx = backend.Context(0)
for name in which:
    t0 = time.time()
    sc = {'simple': lambda: scenes.simple(256, 256, 16, 4),
          'cornell': lambda: scenes.cornell(512, 512, 64, 8),
          'staircase': lambda: scenes.staircase(1920, 1080, 256, 12)}[name]()
    t1 = time.time()
    host = backend.HostScene(sc); dev = ctx.upload(host)
    t2 = time.time()
    film, st = dev.render(seed=0)
    film, st = dev.render(seed=0)
    traced = st['closest_rays'] + st['shadow_rays'] - st['shadow_skipped']
    out = {'config': name, 'tris': int(len(sc.triangles)), 'gen_s': round(t1 - t0, 1), 'scene_new_s': round(t2 - t1, 1),
           'frame_s': round(st['seconds'], 4), 'mray_s': round(traced / st['seconds'] / 1e6, 1), 'paths': st['paths'],
           'closest_rays': st['closest_rays'], 'shadow_rays': st['shadow_rays'], 'shadow_skipped': st['shadow_skipped'],
           'nonfinite': st['nonfinite'], 'stack_overflow': st['stack_overflow'], 'mean': float(film.mean()),
           'kernel_ms': {k: round(st[k], 1) for k in ('trace_closest_ms', 'trace_mixed_ms', 'trace_any_ms', 'shade_ms', 'other_ms')}}
    # sampled parity: first sample of the frame, whole image, against the oracle
    orc = ol.OracleScene(sc)
    g, gst = dev.render(seed=0, sample_range=(0, 1), count_traversal=True)
    o, ost = orc.render(seed=0, sample_range=(0, 1))
    out['sample0_pixel_exact'] = bool(np.array_equal(g, o))
    out['sample0_rmse'] = float(np.sqrt(np.mean((g.astype(np.float64) - o) ** 2)))
    out['sample0_counters_equal'] = all(gst[k] == ost[k] for k in ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims'))
    out['oracle_sample0_s'] = round(ost['seconds'], 2)
    print(json.dumps(out), flush=True)
    dev.close()
