"""The N > 1 shard on CPU: a world-size-2 gloo run of the tile map the PRODUCT uses (cray_tile_pixels, the C ABI's own map —
the one cray_film_pack packs with and cray_render_gather sends with), gloo standing in for the RCCL transport.

Replaces the merge of the reference's worker threads into one `Mutex<Vec<f32>>` (src/bin/craytracer.rs:245, 271-291,
182-188): every rank holds the right values on its own tiles only, packs them in the ABI's order, rank 0 receives each
rank's block at its offset and un-permutes with the ABI's all-ranks map.  The assembled film must be the whole film.

What this CPU test covers of the product is the shard map (cray_tile_pixels) and, with it, the two conventions the device
code relies on: a rank's block in the gathered buffer starts at the sum of the lower ranks' pixel counts (gather_prepare's
rank_offset), and the all-ranks map is the concatenation of the per-rank maps in rank order (ensure_all_pix).  The pack /
receive / unpack steps themselves are restated in torch here; the product's k_pack_tiles / k_unpack_tiles / gather_tiles are
GPU code and are exercised by tests/test_gpu_comm.py (one GPU, pack + unpack against this same map; 2 and 3 processes through
the stand-in transport; a send that fails inside the exchange).
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from craytracer_amd import backend
from craytracer_amd import dist as cdist


def test_the_abi_s_tile_map_is_the_reference_s():
    """cray_tile_pixels against the independent Python restatement of generate_tiles (craytracer.rs:22-43)."""
    for (W, H) in [(1920, 1080), (100, 70), (64, 64), (65, 1), (130, 67)]:
        for world in (1, 2, 3, 8):
            parts = [backend.tile_pixels(W, H, r, world) for r in range(world)]
            for r, p in enumerate(parts):
                assert np.array_equal(p, cdist.rank_pixels(W, H, r, world)), (W, H, r, world)
            allp = np.concatenate(parts)
            assert len(allp) == W * H and len(np.unique(allp)) == W * H   # the shares partition the film
    p = backend.tile_pixels(130, 70, 1, 3)
    assert p[0] == 64 and p[1] == 65          # tile 1 = (tx = 64, ty = 0): ty outer, tx inner, 64x64 tiles
    assert np.array_equal(backend.tile_pixels(130, 67, 1, 2, tile=(32, 16)), cdist.rank_pixels(130, 67, 1, 2, 32, 16))
    # a tile edge near 2^32 must not wrap the tile arithmetic: one tile covers the film (round 4: 64-bit tile arithmetic)
    for big in (2 ** 32 - 1, 2 ** 32 - 64, 2 ** 31):
        assert np.array_equal(backend.tile_pixels(100, 70, 0, 1, tile=(big, big)), np.arange(7000, dtype=np.uint32)), big
        assert len(backend.tile_pixels(100, 70, 1, 2, tile=(big, 7))) == 0 + 100 * 7 * 5   # ten row bands of 7, rank 1 owns five
    # the size query (out = NULL) adds tile areas up without building the map: it must agree with the map
    import ctypes as C
    L = backend.lib()
    for (W, H, tw, th, r, n) in [(1920, 1080, 32, 32, 3, 8), (130, 67, 64, 64, 1, 2), (65, 1, 64, 64, 1, 3), (7, 9, 2, 5, 0, 4)]:
        cnt = C.c_uint64(0)
        assert L.cray_tile_pixels(W, H, tw, th, r, n, None, 0, C.byref(cnt)) == 0
        assert cnt.value == len(cdist.rank_pixels(W, H, r, n, tw, th)), (W, H, tw, th, r, n)
    for bad in ((0, 10, 0, 1), (10, 10, 2, 2), (10, 10, 0, 0)):
        try:
            backend.tile_pixels(*bad)
            raise AssertionError('accepted %r' % (bad,))
        except backend.CrayError:
            pass


def test_a_rank_s_tiles_walk_through_every_column_residue():
    """Round 5: tile (tx, ty) belongs to rank (tx + s ty) % world, s coprime with world — in every row a rank owns every world-th tile
    and over `world` consecutive rows its first tile starts at every residue once (with `t % world` and tiles_x = 4 mod 8 a rank of
    eight owned two column residues and none of the others: one rank carried 3 % more of the frame than the mean)."""
    for world in (2, 3, 4, 5, 6, 7, 8, 9, 16):
        s = cdist.shard_stride(world)
        assert np.gcd(s, world) == 1 and 1 <= s < max(world, 2)
        W, H, t = 32 * world * 2 + 5, 32 * world * 3, 32
        tiles_x = (W + t - 1) // t
        for rank in (0, world - 1):
            pix = backend.tile_pixels(W, H, rank, world, tile=(t, t))
            assert np.array_equal(pix, cdist.rank_pixels(W, H, rank, world, t, t))
            ty, tx = (pix // W) // t, (pix % W) // t
            first = {}
            for y, x in zip(ty.tolist(), tx.tolist()):
                first.setdefault(y, x)                      # pixels come tile by tile, rows of tiles top to bottom, left to right
                assert (x + s * y) % world == rank
            assert len(first) == H // t
            phases = [first[y] for y in range(world)]
            assert sorted(phases) == list(range(world)), (world, phases)      # every residue once in `world` rows
    sizes = [len(backend.tile_pixels(1920, 1080, r, 8, tile=(32, 32))) for r in range(8)]
    assert sum(sizes) == 1920 * 1080 and max(sizes) - min(sizes) <= 34 * 32 * 32   # at most one tile per row of tiles apart


def _worker(rank, world, port, W, H, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    # a per-rank film as a GPU of the shard would hold it: the right values on this rank's tiles, garbage elsewhere
    truth = torch.arange(W * H * 3, dtype=torch.float32).reshape(H, W, 3) * 0.25
    local = torch.full((H, W, 3), -7.0)
    mine = torch.from_numpy(backend.tile_pixels(W, H, rank, world).astype(np.int64))
    local.reshape(-1, 3)[mine] = truth.reshape(-1, 3)[mine]
    packed = local.reshape(-1, 3)[mine].contiguous()                 # cray_film_pack's order
    counts = [len(backend.tile_pixels(W, H, r, world)) for r in range(world)]
    if rank == 0:
        # grouped receive of exactly each rank's pixel count at its offset (gather_tiles in cray_hip.hip), then the un-permute
        gathered = torch.empty((W * H, 3), dtype=torch.float32)
        gathered[: counts[0]] = packed
        off = counts[0]
        for r in range(1, world):
            buf = torch.empty((counts[r], 3), dtype=torch.float32)
            dist.recv(buf, src=r)
            gathered[off: off + counts[r]] = buf
            off += counts[r]
        all_pix = torch.from_numpy(np.concatenate([backend.tile_pixels(W, H, r, world) for r in range(world)]).astype(np.int64))
        out = torch.empty((W * H, 3), dtype=torch.float32)
        out[all_pix] = gathered                                      # k_unpack_tiles
        assert torch.equal(out.reshape(H, W, 3), truth)
        open(out_path, 'w').write('ok')
    else:
        dist.send(packed, dst=0)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2_gloo(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'ok.txt')
    mp.spawn(_worker, args=(2, port, 200, 150, out), nprocs=2, join=True)
    assert open(out).read() == 'ok'
