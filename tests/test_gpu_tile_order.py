"""Tiles rendered in the order of their cost (round 4): a frame that is big enough (>= 2 Mi paths, >= 64 tiles) probes
64 camera rays per tile once per scene and renders the most expensive tiles first, so that the bounce-0 launch does not end on
its slowest pixels.  The order of the pixel list enters no result: films with and without it are the oracle's, for the whole
film and for the shares of a tile shard (whose pack / gather order stays the canonical one).
Seam: the tile loop of the reference, src/bin/craytracer.rs:22-43, 271-291 (workers take tiles in whatever order they come)."""
import os

import numpy as np
import pytest

from craytracer_amd import backend, scenes
from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _ctx(order):
    old = os.environ.get('CRAY_TILE_ORDER')
    os.environ['CRAY_TILE_ORDER'] = order
    try:
        return backend.Context(0)
    finally:
        if old is None:
            del os.environ['CRAY_TILE_ORDER']
        else:
            os.environ['CRAY_TILE_ORDER'] = old


def test_cost_ordered_tiles_render_the_same_film():
    sc = scenes.dragon(640, 360, 20, 5, nu=160, nv=400)      # 4.6 M paths, 20 x 12 tiles of 32 x 32; a half share is still timed and ordered
    ref, ost = ol.OracleScene(sc).render(seed=4)
    ordered, plain = _ctx('1'), _ctx('0')
    host = backend.HostScene(sc, resident=True)
    for ctx in (ordered, plain):
        dev = ctx.upload(host)
        dev.tile = (32, 32)
        for _ in range(3):                                    # f64 records, f32 culling, the choice: all on the same tile order
            f, st = dev.render(seed=4)
            assert np.array_equal(f, ref)
        assert st['closest_rays'] == ost['closest_rays'] and st['shadow_rays'] == ost['shadow_rays']
        # two-way shard of the same frame: every rank's share in its own cost order, packed in the canonical order
        total = np.zeros_like(ref)
        for r in range(2):
            part, _ = dev.render(seed=4, rank=r, world_size=2)
            packed = ctx.film_pack(part, r, 2, tile=(32, 32))
            mine = backend.tile_pixels(640, 360, r, 2, tile=(32, 32))
            assert np.array_equal(packed, part.reshape(-1, 3)[mine])
            total += part
        assert np.array_equal(total, ref)
        dev.close()
    ordered.close(); plain.close()
