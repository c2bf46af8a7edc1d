#!/usr/bin/env python3
"""How does the shape of the shard's tiles change an eighth of the frame?  (The film does not depend on it.)
    python tools/tile_shape_probe.py [--world 8] [--ranks 0,1,6]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='dragon')
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--ranks', default='0,1,6')
    args = ap.parse_args()
    import torch
    import bench
    from craytracer_amd import backend, scenes
    scene = bench.make_scene(scenes, args.workload)
    W, H = scene.film_bounds()
    ctx = backend.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    dev = ctx.upload(backend.HostScene(scene, resident=True))
    film = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
    L = backend.lib()
    for tw, th in ((64, 64), (32, 32), (16, 16), (128, 64), (128, 128), (256, 64), (1920, 8), (1920, 16)):
        rows = []
        for rank in [int(r) for r in args.ranks.split(',')]:
            best = None
            for _ in range(3):
                p = dev.params(seed=0, rank=rank, world_size=args.world)
                p.tile_width, p.tile_height, p.out_is_device = tw, th, 1
                st = backend.Stats()
                rc = L.cray_render(ctx._h, dev._h, C.byref(p), C.c_void_p(film.data_ptr()), C.byref(st))
                assert rc == 0, L.cray_last_error()
                d = st.as_dict()
                if best is None or d['seconds'] < best['seconds']:
                    best = d
            rows.append({'rank': rank, 'ms': round(best['seconds'] * 1e3, 2), 'closest': round(best['trace_closest_ms'], 2), 'mixed': round(best['trace_mixed_ms'], 2),
                         'shade': round(best['shade_ms'], 2), 'paths_M': round(best['paths'] / 1e6, 2)})
        print(json.dumps({'tile': [tw, th], 'ranks': rows}), flush=True)


if __name__ == '__main__':
    main()
