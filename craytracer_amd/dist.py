"""The tile shard of a frame, restated in Python for the tests.

The product's multi-GPU path lives behind the C ABI (include/cray.h "multi-GPU": cray_comm_init, cray_scene_broadcast,
cray_render_gather; its shard map is cray_tile_pixels).  This module only keeps an INDEPENDENT restatement of that map,
written from the reference (src/bin/craytracer.rs:22-43, 232-233), which the tests hold the C ABI to.
"""
import numpy as np


def shard_stride(world_size):
    """Row-to-row phase step of the shard: the smallest stride >= 2 coprime with the world size (so that a rank's tiles walk through
    every column residue), world - 1 when there is none, 1 for one or two ranks."""
    from math import gcd
    for s in range(2, world_size - 1):
        if gcd(s, world_size) == 1:
            return s
    return world_size - 1 if world_size > 2 else 1


def tile_owner(tx, ty, world_size):
    """The rank that renders tile (tx, ty) (tile coordinates, generate_tiles' grid, craytracer.rs:32-33)."""
    return (tx + shard_stride(world_size) * ty) % world_size


def rank_pixels(width, height, rank, world_size, tile_w=64, tile_h=64):
    """Linear pixel indices (y*W + x) of the tiles owned by `rank`, tile by tile in row-major tile order, row-major inside a tile."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    out = []
    for t in range(tiles_x * tiles_y):
        if tile_owner(t % tiles_x, t // tiles_x, world_size) != rank:
            continue
        tx, ty = (t % tiles_x) * tile_w, (t // tiles_x) * tile_h
        x1, y1 = min(tx + tile_w, width), min(ty + tile_h, height)
        ys, xs = np.mgrid[ty:y1, tx:x1]
        out.append((ys * width + xs).reshape(-1))
    return np.concatenate(out).astype(np.int64) if out else np.zeros(0, dtype=np.int64)
