#!/usr/bin/env python3
"""Rehearse the N-GPU tile shard on ONE GPU: render every rank's share of the bench frame in turn and
report the slowest share (what the N-GPU frame time would be without the gather).

    python tools/shard_timing.py [--workload dragon] [--worlds 1,2,4,8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='dragon')
    ap.add_argument('--worlds', default='1,2,4,8')
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--ranks', default='', help='comma list: time only these ranks of each world')
    ap.add_argument('--tile', type=int, default=64, help='edge of the square shard tiles (bench.py --gpus N uses 32)')
    args = ap.parse_args()
    import torch
    import bench
    from craytracer_amd import backend, scenes
    wl = bench.WORKLOADS[args.workload]
    scene = bench.make_scene(scenes, args.workload)
    W, H = scene.film_bounds()
    host = backend.HostScene(scene)
    ctx = backend.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    dev = ctx.upload(host)
    dev.tile = (args.tile, args.tile)
    film = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
    w0 = int(args.worlds.split(',')[0])
    dev.render(seed=0, rank=0, world_size=w0, out_device_ptr=film.data_ptr(), sample_range=(0, min(8, wl['spp'])))  # warm-up
    base = None
    for world in [int(w) for w in args.worlds.split(',')]:
        per_rank = []
        for rank in ([int(r) for r in args.ranks.split(',') if int(r) < world] if args.ranks else range(world)):
            best = None
            for _ in range(args.reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, st = dev.render(seed=0, rank=rank, world_size=world, out_device_ptr=film.data_ptr())
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, st)
            dt, st = best
            per_rank.append({'rank': rank, 'ms': round(dt * 1e3, 2), 'rays': st['closest_rays'] + st['shadow_rays'] - st['shadow_skipped'],
                             'closest_ms': round(st['trace_closest_ms'], 2), 'mixed_ms': round(st['trace_mixed_ms'], 2), 'any_ms': round(st['trace_any_ms'], 2),
                             'shade_ms': round(st['shade_ms'], 2), 'other_ms': round(st['other_ms'], 2), 'tail_split': [st.get('tail_split', 0) & 0xffffff, st.get('tail_split', 0) >> 24]})
        if not per_rank:
            continue
        worst = max(r['ms'] for r in per_rank)
        if base is None:
            base = worst
        print(json.dumps({'world': world, 'slowest_rank_ms': worst, 'mean_rank_ms': round(sum(r['ms'] for r in per_rank) / len(per_rank), 2),
                          'speedup_vs_1': round(base / worst, 2), 'ranks': per_rank}), flush=True)


if __name__ == '__main__':
    main()
