#!/usr/bin/env python3
"""How much would two path pools on two streams buy?  Probe without touching the product: two contexts (own stream,
own pools) render the two halves of the tile set (rank 0/2 and 1/2) of the bench frame from two host threads at
the same time, against the same two halves one after the other.

    python tools/overlap_probe.py [--workload dragon] [--parts 2]
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='dragon')
    ap.add_argument('--parts', type=int, default=2)
    ap.add_argument('--of', type=int, default=0, help='render parts [0, parts) of a shard into `of` pieces (0 = parts)')
    args = ap.parse_args()
    import torch
    import bench
    from craytracer_amd import backend, scenes
    scene = bench.make_scene(scenes, args.workload)
    W, H = scene.film_bounds()
    world = args.of or args.parts
    ctxs = [backend.Context(0) for _ in range(args.parts)]
    host = backend.HostScene(scene, bvh_ctx=ctxs[0])
    devs = [c.upload(host) for c in ctxs]
    films = [torch.zeros((H, W, 3), dtype=torch.float32, device='cuda') for _ in range(args.parts)]

    def run(i):
        devs[i].render(seed=0, rank=i, world_size=world, out_device_ptr=films[i].data_ptr())

    for i in range(args.parts):
        run(i)
    for mode in ('sequential', 'concurrent', 'sequential', 'concurrent'):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == 'sequential':
            for i in range(args.parts):
                run(i)
        else:
            ts = [threading.Thread(target=run, args=(i,)) for i in range(args.parts)]
            [t.start() for t in ts]
            [t.join() for t in ts]
        torch.cuda.synchronize()
        print('%-10s %d of %d shards: %.2f ms' % (mode, args.parts, world, (time.perf_counter() - t0) * 1e3), flush=True)


if __name__ == '__main__':
    main()
