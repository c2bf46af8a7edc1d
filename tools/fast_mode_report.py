"""The f32 "fast" mode (cray_render_params.precision = CRAY_PRECISION_F32_TRAVERSAL) next to the exact f64 path: frame time,
Mray/s and the RMSE between the two films — per BASELINE.json config, and RMSE against the sample count on configs[2] at a
reduced film ("fp32 fast path reported separately ... report RMSE vs spp").
usage: python tools/fast_mode_report.py [out.json]"""
import json, sys
import numpy as np
sys.path.insert(0, '.')
from craytracer_amd import backend, scenes

ctx = backend.Context(0)
out = {}


def pair(dev, **kw):
    res = {}
    for prec in ('f64', 'f32'):
        dev.precision = prec
        dev.render(seed=0, **kw)
        film, st = dev.render(seed=0, **kw)
        rays = st['closest_rays'] + st['shadow_rays'] - st['shadow_skipped']
        res[prec] = (film.astype(np.float64), st, rays)
    a, b = res['f64'][0], res['f32'][0]
    rmse = float(np.sqrt(np.mean((a - b) ** 2)))
    return {'rmse': rmse, 'rel_rmse': rmse / float(a.mean()), 'mean_f64': float(a.mean()), 'mean_f32': float(b.mean()),
            'pixels_differing': float((a != b).any(axis=2).mean()),
            'f64': {'frame_ms': round(res['f64'][1]['seconds'] * 1e3, 2), 'trace_ms': round(sum(res['f64'][1][k] for k in ('trace_closest_ms', 'trace_mixed_ms', 'trace_any_ms')), 2),
                    'mray_s': round(res['f64'][2] / res['f64'][1]['seconds'] / 1e6, 1)},
            'f32': {'frame_ms': round(res['f32'][1]['seconds'] * 1e3, 2), 'trace_ms': round(sum(res['f32'][1][k] for k in ('trace_closest_ms', 'trace_mixed_ms', 'trace_any_ms')), 2),
                    'mray_s': round(res['f32'][2] / res['f32'][1]['seconds'] / 1e6, 1)}}


for name, make in (('cornell 512x512x64', lambda: scenes.cornell(512, 512, 64, 8)), ('dragon 1920x1080x64', lambda: scenes.dragon()),
                   ('staircase 1920x1080x256', lambda: scenes.staircase(1920, 1080, 256, 12))):
    dev = ctx.upload(backend.HostScene(make(), resident=True))
    out[name] = pair(dev)
    print(name, json.dumps(out[name]), flush=True)
    if name.startswith('dragon'):
        curve = {}
        for spp in (1, 4, 16, 64):
            r = pair(dev, sample_range=(0, spp))
            curve[spp] = {'rmse_of_the_spp_frame': r['rmse'] * 64 / spp, 'pixels_differing': r['pixels_differing']}
        out['dragon rmse vs spp (1920x1080, film rescaled to the rendered samples)'] = curve
        print(json.dumps(curve), flush=True)
    dev.close()
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
