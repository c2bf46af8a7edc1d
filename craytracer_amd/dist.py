"""Pixel-tile sharding of a frame over the GPUs of one node and the gather of Film tiles.

The reference merges tiles into one shared `Mutex<Vec<f32>>` (src/bin/craytracer.rs:245,
182-188); across processes that buffer is assembled with ONE collective: every rank packs the
pixels of the tiles it owns (tile_index % world == rank, the same 64x64 tiles as
craytracer.rs:232-233) and rank 0 gathers them over RCCL (`torch.distributed.gather` on the
"nccl" backend = ncclGather over xGMI).  Each rank renders ALL samples of its tiles, so the
per-pixel f32 accumulation order equals the single-GPU run and the assembled film is
bit-identical to it.  No other exchange happens during rendering.
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


def max_rank_pixels(width, height, world_size, tile_w=64, tile_h=64):
    return max(len(rank_pixels(width, height, r, world_size, tile_w, tile_h)) for r in range(world_size))


def gather_film(local_film, width, height, rank, world_size, group=None, tile_w=64, tile_h=64):
    """local_film: torch tensor [H, W, 3] float32 holding this rank's tiles (anything elsewhere).
    Returns the assembled film on rank 0 (None on other ranks).  One `gather` of packed tiles."""
    import torch
    import torch.distributed as dist

    if world_size == 1:
        return local_film
    dev = local_film.device
    flat = local_film.reshape(-1, 3)
    n_max = max_rank_pixels(width, height, world_size, tile_w, tile_h)
    mine = torch.from_numpy(rank_pixels(width, height, rank, world_size, tile_w, tile_h)).to(dev)
    packed = torch.zeros((n_max, 3), dtype=torch.float32, device=dev)
    packed[: len(mine)] = flat[mine]
    if rank == 0:
        parts = [torch.empty_like(packed) for _ in range(world_size)]
        dist.gather(packed, gather_list=parts, dst=0, group=group)
        out = torch.zeros_like(flat)
        for r in range(world_size):
            idx = torch.from_numpy(rank_pixels(width, height, r, world_size, tile_w, tile_h)).to(dev)
            out[idx] = parts[r][: len(idx)]
        return out.reshape(height, width, 3)
    dist.gather(packed, gather_list=None, dst=0, group=group)
    return None
