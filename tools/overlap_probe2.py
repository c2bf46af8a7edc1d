#!/usr/bin/env python3
"""Do two DESYNCHRONISED sub-passes hide each other's launch tails?  (tools/overlap_probe.py ran two equal halves in lockstep:
both reach their tails together.)  Rank `rank` of `world` is rendered (a) in one piece, (b) as two parts by sample range,
concurrently on two contexts (own streams and pools), with the split point and a start stagger as parameters.

    python tools/overlap_probe2.py [--world 8] [--rank 0]
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
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--rank', type=int, default=0)
    args = ap.parse_args()
    import torch
    import bench
    from craytracer_amd import backend, scenes
    scene = bench.make_scene(scenes, args.workload)
    spp = bench.WORKLOADS[args.workload]['spp']
    W, H = scene.film_bounds()
    ctxs = [backend.Context(0) for _ in range(2)]
    host = backend.HostScene(scene, resident=True)
    devs = [c.upload(host) for c in ctxs]
    films = [torch.zeros((H, W, 3), dtype=torch.float32, device='cuda') for _ in range(2)]

    def run(i, rng, delay=0.0):
        if delay:
            time.sleep(delay)
        devs[i].render(seed=0, rank=args.rank, world_size=args.world, sample_range=rng, out_device_ptr=films[i].data_ptr())

    def timed(fn, reps=4):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    run(0, (0, spp)); run(1, (0, spp))
    print('one piece                          : %.2f ms' % timed(lambda: run(0, (0, spp))), flush=True)
    for split in (32, 40, 48, 24):
        for stagger_ms in (0.0, 1.0, 2.0):
            def both():
                ts = [threading.Thread(target=run, args=(0, (0, split))), threading.Thread(target=run, args=(1, (split, spp), stagger_ms * 1e-3))]
                [t.start() for t in ts]
                [t.join() for t in ts]
            print('samples [0,%d) + [%d,%d), stagger %.1f ms: %.2f ms concurrent, %.2f ms one after the other'
                  % (split, split, spp, stagger_ms, timed(both), timed(lambda: (run(0, (0, split)), run(1, (split, spp))))), flush=True)


if __name__ == '__main__':
    main()
