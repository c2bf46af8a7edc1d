"""One rank of tests/test_gpu_comm.py::test_a_send_that_fails_inside_the_exchange_strands_nobody.
usage: worker_sendfail.py <rank> <world> <comm id file>
The stand-in transport makes rank 1's SECOND ncclSend return an error (MOCK_RCCL_FAIL_SEND=1:2): a failure INSIDE the exchange,
after the ranks agreed to start it.  Rank 1 must report it; rank 0, whose receive from rank 1 can then never complete, must come
back too (the stand-in gives up after MOCK_RCCL_STUCK_MS; real RCCL would need the host to abort the communicator), with its
ncclGroupStart closed — the stand-in exits with code 5 from ncclCommDestroy when a group was left open — and with every receive
posted, so that the ranks whose sends did succeed are not left waiting.  The communicator must still deliver the next frame."""
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
info = ctx.comm_describe()
assert info['world'] == world and info['ranks_seen'] == world and info['rank'] == rank and info['transport'] == 'stand-in', info

dev = ctx.upload(backend.HostScene(scenes.cornell(96, 80, 8, 4), resident=True)) if rank == 0 else None
scene = ctx.broadcast_scene(dev, root=0)
want = None
if rank == 0:
    want, _ = scene.render(seed=3)

# frame 1: every send succeeds
got, _ = scene.render_gather(seed=3)
if rank == 0:
    assert np.array_equal(got, want)

# frame 2: rank 1's ncclSend fails inside the exchange
p = scene.params(seed=3)
film = np.zeros((scene.height, scene.width, 3), np.float32)
st = backend.Stats()
t0 = time.time()
rc = L.cray_render_gather(ctx._h, scene._h, C.byref(p), C.c_void_p(film.ctypes.data) if rank == 0 else None, C.byref(st))
msg = L.cray_last_error().decode()
if rank == 1:
    assert rc != 0 and 'Send' in msg and 'rank 1' in msg, (rc, msg)
elif rank == 0:
    assert rc != 0 and 'Recv' in msg, (rc, msg)          # the receive from rank 1 gave up; the group was closed all the same
    assert time.time() - t0 < 60
else:
    assert rc == 0, (rc, msg)                             # a rank whose own send went through has nothing to report

# frame 3: the same communicator, the same buffers.  (Rank 0 sat in its receive for MOCK_RCCL_STUCK_MS = 3 s; the stand-in applies
# that patience to every wait, so the ranks that came back at once wait here rather than inside the barrier.)
time.sleep(max(0.0, t0 + 5.0 - time.time()))
ctx.barrier()
got, _ = scene.render_gather(seed=3)
if rank == 0:
    assert np.array_equal(got, want)
ctx.barrier()
scene.close()
ctx.close()   # ncclCommDestroy: the stand-in exits 5 here if a group was left open
print('rank', rank, 'ok')
