"""The multi-GPU seam of the C ABI (include/cray.h "multi-GPU") as far as ONE GPU can exercise it.

The reference merges worker tiles into one `Mutex<Vec<f32>>` (src/bin/craytracer.rs:245, 271-291, 182-188); the
replacement is pack -> RCCL gather -> unpack.  Here:
  * pack / unpack through the ABI against craytracer_amd.dist.rank_pixels for ragged films and world in {1, 2, 3, 8};
  * a world-size-1 communicator (ncclCommInitRank with one rank): barrier, all-reduce, scene broadcast and
    cray_render_gather run through librccl and give the cray_render film;
  * the share films of cray_render(rank, world) packed per rank, concatenated and unpacked equal the unsharded film.
The N > 1 transport itself (grouped ncclSend / ncclRecv) needs N GPUs: bench.py --gpus N and examples/multi_gpu.c.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from craytracer_amd import backend, dist, scenes
from oracle import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    c = backend.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize('w,h,tile', [(200, 150, (64, 64)), (64, 64, (64, 64)), (65, 1, (64, 64)), (130, 67, (32, 16)), (300, 200, (64, 64))])
@pytest.mark.parametrize('world', [1, 2, 3, 8])
def test_pack_and_unpack_follow_rank_pixels(ctx, w, h, tile, world):
    rng = np.random.default_rng(w * 1000 + h + world)
    film = rng.standard_normal((h, w, 3)).astype(np.float32)
    parts = []
    for r in range(world):
        packed = ctx.film_pack(film, r, world, tile)
        mine = dist.rank_pixels(w, h, r, world, tile[0], tile[1])
        assert len(packed) == len(mine)
        assert np.array_equal(packed, film.reshape(-1, 3)[mine])      # tile by tile, row-major inside a tile
        parts.append(packed)
    gathered = np.concatenate(parts) if parts else np.zeros((0, 3), np.float32)
    assert len(gathered) == w * h                                      # the shares partition the film
    assert np.array_equal(ctx.film_unpack(gathered, w, h, world, tile), film)


def test_world1_communicator_runs_through_rccl():
    c = backend.Context(0)
    assert c.comm_rank() == 0 and c.comm_world_size() == 1
    info = c.comm_describe()                                        # no communicator yet: one rank, nothing loaded
    assert (info['world'], info['rank'], info['ranks_seen'], info['transport']) == (1, 0, 1, 'none')
    c.comm_init(backend.Context.comm_unique_id(), 0, 1)
    assert c.comm_rank() == 0 and c.comm_world_size() == 1
    info = c.comm_describe()                                        # the real collective library, counted by its own all-reduce
    assert info['world'] == 1 and info['ranks_seen'] == 1 and info['transport'] == 'rccl' and info['rccl_version'] > 0, info
    c.barrier()
    assert np.array_equal(c.allreduce([1.5, -2.0, 7.0], 'sum'), [1.5, -2.0, 7.0])
    assert np.array_equal(c.allreduce([3.0], 'max'), [3.0])
    sc = scenes.cornell(96, 80, 8, 6)
    dev = c.upload(backend.HostScene(sc))
    same = c.broadcast_scene(dev, root=0)
    assert same is dev
    a, _ = dev.render(seed=3)
    b, st = dev.render_gather(seed=3)
    assert np.array_equal(a, b) and st['paths'] == 96 * 80 * 8
    with pytest.raises(backend.CrayError):
        c.comm_init(backend.Context.comm_unique_id(), 0, 1)           # one communicator per context
    dev.close()
    c.close()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_rank_shares_packed_and_unpacked_equal_the_unsharded_film(ctx, world):
    sc = scenes.dragon(200, 150, 8, 6, nu=60, nv=150)                  # 4 x 3 tiles, ragged right and bottom edge
    dev = ctx.upload(backend.HostScene(sc))
    whole, _ = dev.render(seed=1)
    parts = []
    for r in range(world):
        share, _ = dev.render(seed=1, rank=r, world_size=world)        # what rank r's GPU holds before the gather
        parts.append(ctx.film_pack(share, r, world))
    film = ctx.film_unpack(np.concatenate(parts), 200, 150, world)
    assert np.array_equal(film, whole)
    o, _ = ol.OracleScene(sc).render(seed=1)
    assert np.array_equal(film, o)
    dev.close()


def test_render_to_host_memory_needs_no_allocation_per_frame(ctx):
    """cray_render with out_is_device = 0 stages through a buffer the context keeps (it used to hipMalloc / hipFree the
    whole film on every call): the film is the same and repeated frames are identical."""
    sc = scenes.simple(128, 96, 8, 4)
    dev = ctx.upload(backend.HostScene(sc))
    a, _ = dev.render(seed=0)
    b, _ = dev.render(seed=0)
    o, _ = ol.OracleScene(sc).render(seed=0)
    assert np.array_equal(a, b) and np.array_equal(a, o)
    dev.close()


def test_c_host_multi_gpu_example_with_one_rank(tmp_path):
    """examples/multi_gpu.c (plain C: fork one process per GPU, id over a pipe, scene broadcast, render + gather)
    with one rank on the one GPU of this box writes the film examples/minimal.c writes."""
    backend.lib()
    csrc = os.path.join(ROOT, 'craytracer_amd', 'csrc')
    exes = {}
    for name in ('minimal', 'multi_gpu'):
        exes[name] = str(tmp_path / name)
        subprocess.check_call(['gcc', '-std=c11', '-Wall', '-Werror', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', name + '.c'),
                               '-L' + csrc, '-lcray_hip', '-Wl,-rpath,' + csrc, '-lm', '-o', exes[name]])
    out1, out2 = str(tmp_path / 'a.exr'), str(tmp_path / 'b.exr')
    r = subprocess.run([exes['minimal'], out1], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exes['multi_gpu'], '1', out2], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(backend.read_exr(out1), backend.read_exr(out2))
    # ... and with three ranks on the one GPU, through the shared-memory stand-in for the collective library
    so = str(tmp_path / 'libmock_rccl.so')
    subprocess.check_call(['hipcc', '-std=c++17', '-O2', '-fPIC', '-shared', '-o', so,
                           os.path.join(ROOT, 'tests', 'mock_rccl', 'mock_rccl.cpp'), '-lrt'], stderr=subprocess.DEVNULL)
    out3 = str(tmp_path / 'c.exr')
    r = subprocess.run([exes['multi_gpu'], '3', out3], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, CRAY_RCCL_LIB=so, CRAY_ONE_DEVICE='1'))
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(backend.read_exr(out1), backend.read_exr(out3))


@pytest.mark.parametrize('world', [2, 3])
def test_several_ranks_on_one_gpu_through_the_mock_transport(tmp_path, world):
    """The N > 1 code path of the C ABI with N processes on the ONE GPU of this box: real RCCL refuses two ranks on a device,
    so the collective library is replaced (CRAY_RCCL_LIB) by tests/mock_rccl — shared-memory mailboxes behind the ten nccl*
    entry points the library binds.  Everything but the transport is the product's: communicator set-up, barrier and
    all-reduce, the scene built on rank 0 and replicated array by array, every rank rendering its tiles, the packed tiles
    received at their offsets and unpacked on rank 0.  The gathered film must be the unsharded film, bit for bit."""
    backend.lib()
    so = str(tmp_path / 'libmock_rccl.so')
    subprocess.check_call(['hipcc', '-std=c++17', '-O2', '-fPIC', '-shared', '-o', so,
                           os.path.join(ROOT, 'tests', 'mock_rccl', 'mock_rccl.cpp'), '-lrt'], stderr=subprocess.DEVNULL)
    env = dict(os.environ, CRAY_RCCL_LIB=so)
    worker = os.path.join(ROOT, 'tests', 'mock_rccl', 'worker.py')
    id_path, out_path = str(tmp_path / 'comm.id'), str(tmp_path / 'film.npy')
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), id_path, out_path], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    films = np.load(out_path)
    c = backend.Context(0)
    dev = c.upload(backend.HostScene(scenes.dragon(200, 136, 8, 6, nu=60, nv=150), resident=True))
    for k, seed in enumerate((5, 6)):
        want, _ = dev.render(seed=seed)
        assert np.array_equal(films[k], want)
    dev.close()
    c.close()


@pytest.mark.parametrize('world', [2, 3])
def test_a_failing_rank_fails_every_rank_instead_of_hanging_them(tmp_path, world):
    """Real RCCL has no timeout: a rank that returns early from cray_render_gather / cray_scene_broadcast (bad arguments, out of
    memory, a HIP error in its render) would leave rank 0 waiting in ncclRecv for ever.  The ranks therefore agree on their
    status (one-word all-reduce) before any transfer; here rank 1 is made to fail locally and EVERY rank must come back with
    an error, within the timeout, and the communicator must still render a correct frame afterwards."""
    backend.lib()
    so = str(tmp_path / 'libmock_rccl.so')
    subprocess.check_call(['hipcc', '-std=c++17', '-O2', '-fPIC', '-shared', '-o', so,
                           os.path.join(ROOT, 'tests', 'mock_rccl', 'mock_rccl.cpp'), '-lrt'], stderr=subprocess.DEVNULL)
    env = dict(os.environ, CRAY_RCCL_LIB=so)
    worker = os.path.join(ROOT, 'tests', 'mock_rccl', 'worker_fail.py')
    id_path = str(tmp_path / 'comm.id')
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), id_path], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError('a rank was left waiting for a peer that had already failed')
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)


@pytest.mark.parametrize('world', [2, 3])
def test_a_send_that_fails_inside_the_exchange_strands_nobody(tmp_path, world):
    """The status agreement covers what a rank does BEFORE the exchange; this is a failure inside it: rank 1's ncclSend returns an
    error (injected by the stand-in transport).  gather_tiles must still post every receive and close its ncclGroupStart on rank 0
    (round 3 returned from inside the group), rank 1 must name the failed call, and the communicator must carry the next frame.
    Seam: the merge of the reference's workers, src/bin/craytracer.rs:245, 182-188."""
    backend.lib()
    so = str(tmp_path / 'libmock_rccl.so')
    subprocess.check_call(['hipcc', '-std=c++17', '-O2', '-fPIC', '-shared', '-o', so,
                           os.path.join(ROOT, 'tests', 'mock_rccl', 'mock_rccl.cpp'), '-lrt'], stderr=subprocess.DEVNULL)
    env = dict(os.environ, CRAY_RCCL_LIB=so, MOCK_RCCL_FAIL_SEND='1:2', MOCK_RCCL_STUCK_MS='3000')
    worker = os.path.join(ROOT, 'tests', 'mock_rccl', 'worker_sendfail.py')
    id_path = str(tmp_path / 'comm.id')
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), id_path], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError('a rank was left waiting after a send failed inside the exchange')
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)


def test_bench_with_two_ranks_as_the_driver_launches_it(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` — the driver's command for N = 2 — on the one GPU
    of this box, the collective library replaced by the shared-memory stand-in: the id exchange over the rendezvous store,
    communicator, scene broadcast, render + gather and the max-over-ranks timing run as they will on two GPUs, and rank 0
    prints exactly one JSON line."""
    import json
    so = str(tmp_path / 'libmock_rccl.so')
    subprocess.check_call(['hipcc', '-std=c++17', '-O2', '-fPIC', '-shared', '-o', so,
                           os.path.join(ROOT, 'tests', 'mock_rccl', 'mock_rccl.cpp'), '-lrt'], stderr=subprocess.DEVNULL)
    env = dict(os.environ, CRAY_RCCL_LIB=so)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29547', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--workload', 'cornell', '--cpu-baseline', '0'], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['value'] > 0 and d['scaling'] == 'strong'
    assert 'cray_render_gather' in d['config']['parallelism']
    # the line proves what it ran on: ranks counted by an all-reduce, the collective library named, every rank's kernel times
    assert d['comm']['world'] == 2 and d['comm']['ranks_seen'] == 2 and d['comm']['transport'] == 'stand-in' and d['comm']['library'].endswith('libmock_rccl.so')
    assert [r['rank'] for r in d['kernel_ms_per_rank']] == [0, 1] and all(r['trace_ms'] > 0 and r['paths_per_step'] > 0 for r in d['kernel_ms_per_rank'])
    assert sum(r['paths_per_step'] for r in d['kernel_ms_per_rank']) == 512 * 512 * 64
    assert d['roofline'] is None or d['roofline'].get('frac') is None

