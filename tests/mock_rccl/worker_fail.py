"""One rank of tests/test_gpu_comm.py::test_a_failing_rank_fails_every_rank_instead_of_hanging_them.
usage: worker_fail.py <rank> <world> <comm id file>
Rank 1 makes a call that fails locally (bad arguments: something only that rank can see).  Every rank must get a non-zero
code back from the collective call — the others must not be left waiting inside ncclRecv / ncclBroadcast — and the
communicator must still work afterwards."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from craytracer_amd import backend, scenes  # noqa: E402

rank, world, id_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
L = backend.lib()
ctx = backend.Context(0)
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

# 1. scene broadcast with a bad `root` on rank 1 only: every rank returns an error, nobody enters ncclBroadcast alone
dev = ctx.upload(backend.HostScene(scenes.cornell(96, 80, 8, 4), resident=True)) if rank == 0 else None
out = C.c_void_p()
rc = L.cray_scene_broadcast(ctx._h, dev._h if dev is not None else None, 99 if rank == 1 else 0, C.byref(out))
assert rc != 0, 'rank %d: broadcast with a failing peer returned 0' % rank
msg = L.cray_last_error().decode()
assert ('root 99' in msg) if rank == 1 else ('another rank failed' in msg), msg

# 2. the same call with good arguments works
scene = ctx.broadcast_scene(dev, root=0)

# 3. render + gather where rank 1's parameters are invalid (tile_width 0): every rank returns the error
p = scene.params(seed=3)
if rank == 1:
    p.tile_width = 0
film = np.zeros((scene.height, scene.width, 3), np.float32)
st = backend.Stats()
rc = L.cray_render_gather(ctx._h, scene._h, C.byref(p), C.c_void_p(film.ctypes.data) if rank == 0 else None, C.byref(st))
assert rc != 0, 'rank %d: gather with a failing peer returned 0' % rank
msg = L.cray_last_error().decode()
assert ('bad tile' in msg) if rank == 1 else ('another rank failed' in msg), msg

# 3b. ranks that pass different tile shapes would post sends and receives of different lengths: an error on every rank
p = scene.params(seed=3)
p.tile_width = 32 if rank == 1 else 64
rc = L.cray_render_gather(ctx._h, scene._h, C.byref(p), C.c_void_p(film.ctypes.data) if rank == 0 else None, C.byref(st))
assert rc != 0 and 'disagree' in L.cray_last_error().decode(), (rank, rc, L.cray_last_error().decode())

# 4. and the next frame is fine: the film of the gather equals the unsharded film
got, _ = scene.render_gather(seed=3)
if rank == 0:
    want, _ = scene.render(seed=3)
    assert np.array_equal(got, want)
ctx.barrier()
scene.close()
ctx.close()
print('rank', rank, 'ok')
