"""One rank of tests/test_gpu_comm.py::test_several_ranks_on_one_gpu_through_the_mock_transport.
usage: worker.py <rank> <world> <comm id file> <out .npy (rank 0)>"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from craytracer_amd import backend, scenes  # noqa: E402

rank, world, id_path, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
ctx = backend.Context(0)                                   # every rank on the one GPU of the box
if rank == 0:
    cid = backend.Context.comm_unique_id()
    with open(id_path + '.tmp', 'wb') as f:
        f.write(cid)
    os.replace(id_path + '.tmp', id_path)
else:
    t0 = time.time()
    while not os.path.exists(id_path):
        assert time.time() - t0 < 120
        time.sleep(0.05)
    cid = open(id_path, 'rb').read()
ctx.comm_init(cid, rank, world)
assert ctx.comm_rank() == rank and ctx.comm_world_size() == world
ctx.barrier()
s = ctx.allreduce([float(rank + 1), float(rank), -float(rank)], 'sum')
assert list(s) == [world * (world + 1) / 2, world * (world - 1) / 2, -world * (world - 1) / 2], s
assert ctx.allreduce([float(rank)], 'max')[0] == world - 1 and ctx.allreduce([float(rank)], 'min')[0] == 0.0
dev = None
if rank == 0:                                              # the host side of the reference runs once, on rank 0
    dev = ctx.upload(backend.HostScene(scenes.dragon(200, 136, 8, 6, nu=60, nv=150), resident=True))
dev = ctx.broadcast_scene(dev, root=0)                     # HBM of rank 0 -> every rank
film, st = dev.render_gather(seed=5)
film2, st2 = dev.render_gather(seed=6)                     # buffers are reused across frames
if rank == 0:
    np.save(out_path, np.stack([film, film2]))
    print('rank0 paths', st['paths'], st2['paths'])
else:
    assert film is None
ctx.barrier()
dev.close()
ctx.close()
