#!/usr/bin/env python3
"""Per-kernel timeline of ONE rank's share of the bench frame (diagnostics for the N-GPU efficiency).

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/share_trace -o run -- python3 tools/share_trace.py --world 8 --rank 0
  python3 tools/share_trace.py --report gpurun_out/share_trace      # after the run: durations and gaps of the last frame

Without --report it renders the share `--frames` times (the profiler records every dispatch); with --report it reads the
kernel trace and prints, for the last frame, every launch with its start offset, duration and the idle gap before it.
"""
import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def report(d):
    f = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        k = name.split('(')[0].split('::')[-1]
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k))
    rows.sort()
    # the last frame starts at the last k_raygen
    last = max(i for i, r in enumerate(rows) if r[2].startswith('k_raygen'))
    t0 = rows[last][0]
    prev_end = rows[last - 1][1] if last else t0
    tot = {}
    print('%-34s %10s %10s %10s' % ('kernel', 'start us', 'dur us', 'gap us'))
    for s, e, k in rows[last:]:
        print('%-34s %10.1f %10.1f %10.1f' % (k[:34], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        tot[k.split('<')[0]] = tot.get(k.split('<')[0], 0.0) + (e - s) / 1e6
        prev_end = e
    print('frame: %.2f ms first start -> last end; kernel ms by family: %s' % ((prev_end - t0) / 1e6, {k: round(v, 2) for k, v in tot.items()}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='dragon')
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--frames', type=int, default=3)
    ap.add_argument('--report', default='')
    ap.add_argument('--count', type=int, default=0, help='count_traversal mode of the renders (with CRAY_LOG_QUEUES=1: per-bounce counters)')
    args = ap.parse_args()
    if args.report:
        return report(args.report)
    import torch
    import bench
    from craytracer_amd import backend, scenes
    scene = bench.make_scene(scenes, args.workload)
    W, H = scene.film_bounds()
    ctx = backend.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    dev = ctx.upload(backend.HostScene(scene, resident=True))
    film = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
    for _ in range(args.frames):
        _, st = dev.render(seed=0, rank=args.rank, world_size=args.world, out_device_ptr=film.data_ptr(), count_traversal=args.count)
        torch.cuda.synchronize()
    print({k: round(v, 2) if isinstance(v, float) else v for k, v in st.items() if k.endswith('_ms') or k in ('seconds', 'paths')}, file=sys.stderr)


if __name__ == '__main__':
    main()
