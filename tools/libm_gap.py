#!/usr/bin/env python3
"""Per BASELINE.json config: RMSE between the oracle with correctly rounded sin/cos (what oracle and product use) and the
oracle with the platform libm (what the reference binary links against).  Writes profiles/libm_gap.json.
usage: python tools/libm_gap.py [--spp-cap N]   (host cores only; the large configs render a capped number of samples)"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from craytracer_amd import scenes
from oracle import oracle_lib as ol

ap = argparse.ArgumentParser()
ap.add_argument('--spp-cap', type=int, default=8)
ap.add_argument('--threads', type=int, default=0)
ap.add_argument('--out', default=None)
args = ap.parse_args()
cfgs = {'simple': (lambda: scenes.simple(256, 256, 16, 4), 16), 'cornell': (lambda: scenes.cornell(512, 512, 64, 8), 64),
        'dragon': (lambda: scenes.dragon(), args.spp_cap), 'staircase': (lambda: scenes.staircase(1920, 1080, 256, 12), args.spp_cap)}
out = {}
for name, (make, spp) in cfgs.items():
    sc = make()
    orc = ol.OracleScene(sc)
    t = time.time()
    ol.set_libm_mode(0); a, _ = orc.render(seed=0, threads=args.threads, sample_range=(0, spp))
    ol.set_libm_mode(1); b, _ = orc.render(seed=0, threads=args.threads, sample_range=(0, spp))
    ol.set_libm_mode(0)
    scale = sc.num_samples / spp        # the film is divided by num_samples: rescale to "a frame of `spp` samples"
    a64, b64 = a.astype(np.float64) * scale, b.astype(np.float64) * scale
    out[name] = {'film': list(sc.film_bounds()), 'spp_rendered': spp, 'spp_config': sc.num_samples,
                 'rmse': float(np.sqrt(np.mean((a64 - b64) ** 2))), 'pixels_differing': float((a != b).any(axis=2).mean()),
                 'mean_radiance': float(a64.mean()), 'seconds': round(time.time() - t, 1)}
    print(name, out[name], flush=True)
dst = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'libm_gap.json')
json.dump(out, open(dst, 'w'), indent=1)
