#!/usr/bin/env python3
"""Generates tests/golden/films/<scene>.npz: the CPU oracle's film (f32, seed 0) and query counters for the five
small parity scenes of tests/parity_util.py.  The fixtures pin the oracle itself against regressions (CPU suite)
and give the GPU suite committed expected outputs besides the live oracle.

    python tools/gen_golden_films.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_lib as ol          # noqa: E402
from tests.parity_util import small_scenes   # noqa: E402

KEYS = ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims')

if __name__ == '__main__':
    out_dir = os.path.join(ROOT, 'tests', 'golden', 'films')
    os.makedirs(out_dir, exist_ok=True)
    for name, sc in small_scenes():
        film, st = ol.OracleScene(sc).render(seed=0)
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), film=film.astype(np.float32),
                            counters=np.array([st[k] for k in KEYS], dtype=np.uint64))
        print(name, film.shape, float(film.mean()), [int(st[k]) for k in KEYS])
