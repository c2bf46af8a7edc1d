"""The tile shard of a frame, restated in Python for the tests.

The product's multi-GPU path lives behind the C ABI (include/cray.h "multi-GPU": cray_comm_init, cray_scene_broadcast,
cray_render_gather; its shard map is cray_tile_pixels).  This module only keeps an INDEPENDENT restatement of that map,
written from the reference (src/bin/craytracer.rs:22-43, 232-233), which the tests hold the C ABI to.
"""
import numpy as np


def rank_pixels(width, height, rank, world_size, tile_w=64, tile_h=64):
    """Linear pixel indices (y*W + x) of the tiles owned by `rank`, tile by tile, row-major inside
    a tile; tiles numbered ty-major like generate_tiles (craytracer.rs:32-33)."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    out = []
    for t in range(rank, tiles_x * tiles_y, world_size):
        tx, ty = (t % tiles_x) * tile_w, (t // tiles_x) * tile_h
        x1, y1 = min(tx + tile_w, width), min(ty + tile_h, height)
        ys, xs = np.mgrid[ty:y1, tx:x1]
        out.append((ys * width + xs).reshape(-1))
    return np.concatenate(out).astype(np.int64) if out else np.zeros(0, dtype=np.int64)
