#!/usr/bin/env python3
"""Which side binds the traversal: the compute units (VALU issue, texture addresser, latency x occupancy — all per CU) or the
memory system they share (L2 miss path, Infinity Fabric, HBM)?  Render the bench frame on streams restricted to a subset of the
CUs (hipExtStreamCreateWithCUMask) and compare kernel times: a per-CU bound doubles with half the CUs, a shared bound does not.

    python tools/cu_mask_probe.py [--workload dragon]
"""
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
    ap.add_argument('--reps', type=int, default=2)
    args = ap.parse_args()
    import torch
    import bench
    from craytracer_amd import backend, scenes
    assert torch.cuda.is_available()
    torch.cuda.init()
    torch.zeros(1, device='cuda')
    hip = C.CDLL('libamdhip64.so')
    scene = bench.make_scene(scenes, args.workload)
    W, H = scene.film_bounds()
    host = backend.HostScene(scene, resident=True)
    masks = {
        'all 256 CUs': [0xffffffff] * 8,
        'every other CU (128)': [0x55555555] * 8,
        'first half of the mask words (128)': [0xffffffff] * 4 + [0] * 4,
        'a quarter (64): every fourth CU': [0x11111111] * 8,
    }
    film = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
    for name, words in masks.items():
        stream = C.c_void_p()
        arr = (C.c_uint32 * 8)(*words)
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(stream), 8, arr)
        assert rc == 0, rc
        ctx = backend.Context(0, stream=stream.value)
        dev = ctx.upload(host)
        best = None
        for _ in range(args.reps + 1):
            _, st = dev.render(seed=0, out_device_ptr=film.data_ptr())
            if best is None or st['seconds'] < best['seconds']:
                best = st
        print(json.dumps({'cus': name, 'frame_ms': round(best['seconds'] * 1e3, 2), 'trace_closest_ms': round(best['trace_closest_ms'], 2),
                          'trace_mixed_ms': round(best['trace_mixed_ms'], 2), 'shade_ms': round(best['shade_ms'], 2), 'other_ms': round(best['other_ms'], 2)}), flush=True)
        dev.close()
        ctx.close()
        hip.hipStreamDestroy(stream)


if __name__ == '__main__':
    main()
