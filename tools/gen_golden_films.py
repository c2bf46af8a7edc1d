#!/usr/bin/env python3
"""Generates tests/golden/films/<scene>.npz — the CPU oracle's film (f32, seed 0) and query counters — and
tests/golden/hits/<scene>.npz — 256 rays with the oracle's closest-hit records and any-hit answers — for the
small parity scenes of tests/parity_util.py (self-generated fixtures: they come from the oracle, not from the reference).  The fixtures pin the oracle itself against regressions (CPU suite)
and give the GPU suite committed expected outputs besides the live oracle.

    python tools/gen_golden_films.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_lib as ol          # noqa: E402
from tests.parity_util import small_scenes, random_rays   # noqa: E402

KEYS = ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims')

if __name__ == '__main__':
    out_dir = os.path.join(ROOT, 'tests', 'golden', 'films')
    os.makedirs(out_dir, exist_ok=True)
    for name, sc in small_scenes():
        film, st = ol.OracleScene(sc).render(seed=0)
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), film=film.astype(np.float32),
                            counters=np.array([st[k] for k in KEYS], dtype=np.uint64))
        print(name, film.shape, float(film.mean()), [int(st[k]) for k in KEYS])
        orc = ol.OracleScene(sc)
        rays = random_rays(orc, 256, seed=11)
        closest, _ = orc.trace(rays)
        anyhit, _ = orc.trace(rays, any_hit=True)
        hits_dir = os.path.join(ROOT, 'tests', 'golden', 'hits')
        os.makedirs(hits_dir, exist_ok=True)
        np.savez_compressed(os.path.join(hits_dir, name + '.npz'), rays=rays, hit=closest['hit'], prim=closest['prim'], t=closest['t'],
                            location=closest['location'], normal=closest['normal'], uv=closest['uv'], any_hit=anyhit['hit'])
