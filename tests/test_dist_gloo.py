"""N>1 path on CPU: world_size-2 gloo run of the tile shard + Film-tile gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from craytracer_amd import dist as cdist


def test_rank_pixels_partition_the_film():
    for (W, H) in [(1920, 1080), (100, 70), (64, 64), (65, 1)]:
        for world in (1, 2, 3, 8):
            allp = np.concatenate([cdist.rank_pixels(W, H, r, world) for r in range(world)])
            assert len(allp) == W * H and len(np.unique(allp)) == W * H
    # tile order of generate_tiles (craytracer.rs:32-33): ty outer, tx inner; 64x64 tiles
    p = cdist.rank_pixels(130, 70, 1, 3)
    assert p[0] == 64 and p[1] == 65          # tile 1 = (tx=64, ty=0)


def _worker(rank, world, port, W, H, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    # a fake per-rank film: the right values on this rank's tiles, garbage elsewhere
    truth = torch.arange(W * H * 3, dtype=torch.float32).reshape(H, W, 3) * 0.25
    local = torch.full((H, W, 3), -7.0)
    mine = torch.from_numpy(cdist.rank_pixels(W, H, rank, world))
    local.reshape(-1, 3)[mine] = truth.reshape(-1, 3)[mine]
    out = cdist.gather_film(local, W, H, rank, world)
    if rank == 0:
        assert torch.equal(out, truth)
        open(out_path, 'w').write('ok')
    else:
        assert out is None
    dist.destroy_process_group()


def test_gather_film_world2_gloo(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'ok.txt')
    mp.spawn(_worker, args=(2, port, 200, 150, out), nprocs=2, join=True)
    assert open(out).read() == 'ok'
