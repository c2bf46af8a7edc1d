#!/usr/bin/env python3
"""configs[4] (3840x2160, 1024 spp: the 8-GPU configuration of BASELINE.json) rehearsed on ONE GPU: the whole frame once and rank
`--rank`'s eighth of it (32x32 shard tiles, (tx + 3 ty) % 8 == rank), both timed after a warm-up that absorbs the path pool's allocation.
One-GPU rehearsal: no gather, no RCCL.   python tools/eighth_4k.py [--rank 3] [--whole 1]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rank', type=int, default=3)
    ap.add_argument('--whole', type=int, default=1)
    args = ap.parse_args()
    import torch
    import bench
    from craytracer_amd import backend, scenes
    scene = bench.make_scene(scenes, 'dragon_4k')
    W, H = scene.film_bounds()
    ctx = backend.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    dev = ctx.upload(backend.HostScene(scene, resident=True))
    dev.tile = (32, 32)
    film = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')

    def run(rank, world, sample_range=None):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, st = dev.render(seed=0, rank=rank, world_size=world, out_device_ptr=film.data_ptr(), sample_range=sample_range)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, st

    for _ in range(2):                     # the path pool's allocation (190 GB: seconds, once) + the two frames that choose the records
        run(args.rank, 8)
    out = {}
    dt, st = run(args.rank, 8)
    out['eighth'] = {'rank': args.rank, 'seconds': round(dt, 4), 'paths': st['paths'], 'rays': st['closest_rays'] + st['shadow_rays'] - st['shadow_skipped'],
                     'trace_ms': round(st['trace_closest_ms'] + st['trace_mixed_ms'] + st['trace_any_ms'], 1), 'shade_ms': round(st['shade_ms'], 1),
                     'launches': st['trace_closest_launches'] + st['trace_mixed_launches'] + st['trace_any_launches']}
    print(json.dumps(out['eighth']), flush=True)
    if args.whole:
        dt1, st1 = run(0, 1)
        out['whole'] = {'seconds': round(dt1, 4), 'paths': st1['paths'], 'rays': st1['closest_rays'] + st1['shadow_rays'] - st1['shadow_skipped'],
                        'trace_ms': round(st1['trace_closest_ms'] + st1['trace_mixed_ms'] + st1['trace_any_ms'], 1), 'shade_ms': round(st1['shade_ms'], 1),
                        'mray_s': round((st1['closest_rays'] + st1['shadow_rays'] - st1['shadow_skipped']) / dt1 / 1e6, 1)}
        out['whole_over_eighth'] = round(dt1 / dt, 3)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
