#!/usr/bin/env python3
"""The whole N-rank frame through the C ABI on ONE GPU: N host threads of one process, one cray_ctx each (own stream), the
collective library replaced by tests/mock_rccl (CRAY_RCCL_LIB) — communicator, scene broadcast, every rank's tiles, pack,
gather, unpack and the copy of the film to the host all run as `bench.py --gpus N` runs them.  The box allows six GPU
processes, so eight ranks are threads here.  This is a FUNCTIONAL rehearsal with the pieces timed — the ranks share one GPU and
the stand-in moves bytes through host memory, so neither the frame time nor the gather time is a scaling number.

    CRAY_RCCL_LIB=<libmock_rccl.so> python tools/mock8_frame.py [--world 8] [--workload dragon]
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--workload', default='dragon')
    ap.add_argument('--frames', type=int, default=2)
    args = ap.parse_args()
    assert os.environ.get('CRAY_RCCL_LIB'), 'set CRAY_RCCL_LIB to the stand-in library (real RCCL refuses several ranks on one device)'
    import numpy as np
    import torch
    import bench
    from craytracer_amd import backend, scenes
    torch.zeros(1, device='cuda')
    scene = bench.make_scene(scenes, args.workload)
    W, H = scene.film_bounds()
    world = args.world
    cid = backend.Context.comm_unique_id()
    out = {}
    errors = []
    bar = threading.Barrier(world)

    def rank_main(rank):
        try:
            ctx = backend.Context(0)                      # its own stream
            ctx.comm_init(cid, rank, world)
            dev = ctx.upload(backend.HostScene(scene, resident=True)) if rank == 0 else None
            t0 = time.perf_counter()
            dev = ctx.broadcast_scene(dev, root=0)
            t_bcast = time.perf_counter() - t0
            film = np.zeros((H, W, 3), np.float32) if rank == 0 else None
            frames = []
            for _ in range(args.frames):
                bar.wait()
                t0 = time.perf_counter()
                _, st = dev.render_gather(seed=0, out=film)
                frames.append((time.perf_counter() - t0, st))
            # the gather alone: every rank's tiles already rendered into a device film, then only pack + transport + unpack + D2H
            local = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
            dev.render(seed=0, rank=rank, world_size=world, out_device_ptr=local.data_ptr())
            torch.cuda.synchronize()
            bar.wait()
            t0 = time.perf_counter()
            ctx.film_gather(local.data_ptr(), W, H, out=film)
            t_gather = time.perf_counter() - t0
            out[rank] = dict(bcast=t_bcast, frames=frames, gather=t_gather, film=film)
            ctx.barrier()
            dev.close()
            ctx.close()
        except Exception as e:   # a dead rank would leave the others in the barrier
            errors.append((rank, repr(e)))
            bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # the gathered film equals the unsharded one
    ctx = backend.Context(0)
    dev = ctx.upload(backend.HostScene(scene, resident=True))
    whole, wst = dev.render(seed=0)
    same = bool(np.array_equal(out[0]['film'], whole))
    last = [out[r]['frames'][-1] for r in range(world)]
    kern = [sum(st[k] for k in ('trace_closest_ms', 'trace_mixed_ms', 'trace_any_ms', 'shade_ms', 'other_ms')) for _, st in last]
    print(json.dumps({'world': world, 'workload': args.workload, 'film_equals_unsharded': same,
                      'frame_wall_ms_rank0': round(last[0][0] * 1e3, 2), 'slowest_rank_wall_ms': round(max(t for t, _ in last) * 1e3, 2),
                      'kernel_ms_per_rank': [round(k, 2) for k in kern], 'sum_kernel_ms': round(sum(kern), 2),
                      'gather_only_ms_rank0': round(out[0]['gather'] * 1e3, 2),
                      'scene_broadcast_s': round(max(out[r]['bcast'] for r in range(world)), 2),
                      'unsharded_frame_ms': round(wst['seconds'] * 1e3, 2),
                      'note': 'ranks are threads sharing ONE GPU and the transport is a host-memory stand-in: functional rehearsal, not a scaling figure'}))
    assert same


if __name__ == '__main__':
    main()
