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


class FilmGather:
    """Per-(film, rank, world) plan of the tile gather: index tensors and buffers are built once, a
    frame then costs one index_select (pack), one `gather`, and on rank 0 one index_select (unpack)."""

    def __init__(self, width, height, rank, world_size, device, group=None, tile_w=64, tile_h=64):
        import torch

        self.width, self.height, self.rank, self.world, self.group = width, height, rank, world_size, group
        per_rank = [rank_pixels(width, height, r, world_size, tile_w, tile_h) for r in range(world_size)]
        self.n_max = max(len(p) for p in per_rank)
        # pack: row i of the packed buffer is pixel mine[i]; rows past len(mine) repeat pixel 0 (never read back)
        mine = np.zeros(self.n_max, dtype=np.int64)
        mine[: len(per_rank[rank])] = per_rank[rank]
        self.mine = torch.from_numpy(mine).to(device)
        self.packed = torch.empty((self.n_max, 3), dtype=torch.float32, device=device)
        self.parts = None
        if rank == 0:
            # unpack: pixel p lives at row where[p] of the concatenated parts
            where = np.zeros(width * height, dtype=np.int64)
            for r, p in enumerate(per_rank):
                where[p] = r * self.n_max + np.arange(len(p), dtype=np.int64)
            self.where = torch.from_numpy(where).to(device)
            self.parts = torch.empty((world_size, self.n_max, 3), dtype=torch.float32, device=device)
            self.out = torch.empty((width * height, 3), dtype=torch.float32, device=device)

    def gather(self, local_film):
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return local_film
        torch.index_select(local_film.reshape(-1, 3), 0, self.mine, out=self.packed)
        staged = self.packed.is_cuda and dist.get_backend(self.group) == 'gloo'  # gloo gathers host tensors only
        if self.rank == 0:
            if staged:
                parts = [torch.empty(self.packed.shape, dtype=torch.float32) for _ in range(self.world)]
                dist.gather(self.packed.cpu(), gather_list=parts, dst=0, group=self.group)
                self.parts.copy_(torch.stack(parts))
            else:
                dist.gather(self.packed, gather_list=list(self.parts.unbind(0)), dst=0, group=self.group)
            torch.index_select(self.parts.reshape(-1, 3), 0, self.where, out=self.out)
            return self.out.reshape(self.height, self.width, 3)
        dist.gather(self.packed.cpu() if staged else self.packed, gather_list=None, dst=0, group=self.group)
        return None


_plans = {}


def gather_film(local_film, width, height, rank, world_size, group=None, tile_w=64, tile_h=64):
    """local_film: torch tensor [H, W, 3] float32 holding this rank's tiles (anything elsewhere).
    Returns the assembled film on rank 0 (None on other ranks).  One `gather` of packed tiles;
    the plan (index tensors, buffers) is cached per film geometry."""
    if world_size == 1:
        return local_film
    key = (width, height, rank, world_size, str(local_film.device), id(group), tile_w, tile_h)
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = FilmGather(width, height, rank, world_size, local_film.device, group, tile_w, tile_h)
    return plan.gather(local_film)
