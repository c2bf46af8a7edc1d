#!/usr/bin/env python3
"""What would sorting the rays of a bounce buy the traversal?  Bounce-1-like rays of the configs[2] scene (start on the surfaces the
camera sees, random directions on the incoming side), traced by the timed closest-hit instantiation in pixel order, shuffled, and
sorted by (direction octant, Morton code of the origin) / by Morton code alone."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from craytracer_amd import backend, scenes

scene = bench.make_scene(scenes, 'dragon')
host = backend.HostScene(scene)
ctx = backend.Context(0, stream=torch.cuda.current_stream().cuda_stream)
dev = ctx.upload(host)
W, H, SPP = 1920, 1080, 2
eye = np.array([150.0, 70.0, 150.0]); at = np.array([30.0, -50.0, 0.0]); up = np.array([0.0, 1.0, 0.0])
f = at - eye; f /= np.linalg.norm(f); r = np.cross(f, up); r /= np.linalg.norm(r); u = np.cross(r, f)
rng = np.random.default_rng(1)
ys, xs = np.mgrid[0:H, 0:W]
xs = np.repeat(xs.reshape(-1), SPP); ys = np.repeat(ys.reshape(-1), SPP)
jx = rng.random(xs.size); jy = rng.random(xs.size)
t = np.tan(np.radians(60.0) / 2)
sx = ((xs + jx) / W * 2 - 1) * t * W / H; sy = (1 - (ys + jy) / H * 2) * t
d = f[None, :] + sx[:, None] * r[None, :] + sy[:, None] * u[None, :]
d /= np.linalg.norm(d, axis=1)[:, None]
rays = np.concatenate([np.broadcast_to(eye, d.shape), d, np.full((len(d), 1), np.inf)], axis=1)
hits, st = dev.trace(rays, timed=True)
print(json.dumps({'camera_rays': len(rays), 'closest_ms': st['trace_closest_ms'], 'hit_fraction': float((hits['hit'] != 0).mean())}), flush=True)
m = hits['hit'] != 0
n = hits['normal'][m].copy()
n[(n * rays[m, 3:6]).sum(axis=1) > 0] *= -1
o = hits['location'][m] + n * 1e-4
nd = rng.normal(size=o.shape); nd /= np.linalg.norm(nd, axis=1)[:, None]
nd[(nd * n).sum(axis=1) < 0] *= -1
b1 = np.concatenate([o, nd, np.full((len(o), 1), np.inf)], axis=1)
def morton(o):
    lo = o.min(axis=0); hi = o.max(axis=0)
    q = np.clip(((o - lo) / (hi - lo + 1e-12) * 1023).astype(np.uint64), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
mc = morton(o)
octant = ((nd[:, 0] < 0).astype(np.uint64) | ((nd[:, 1] < 0).astype(np.uint64) << 1) | ((nd[:, 2] < 0).astype(np.uint64) << 2))
orders = {
    'pixel order': np.arange(len(b1)),
    'shuffled': rng.permutation(len(b1)),
    'morton(origin)': np.argsort(mc, kind='stable'),
    'octant, morton(origin)': np.argsort((octant << 30) | mc, kind='stable'),
    'morton(origin >> 9 bits), octant, morton': np.argsort(((mc >> 9) << 33) | (octant << 30) | mc, kind='stable'),
}
for name, idx in orders.items():
    best = None
    for rep in range(3):
        h, st = dev.trace(np.ascontiguousarray(b1[idx]), timed=True)
        best = st['trace_closest_ms'] if best is None else min(best, st['trace_closest_ms'])
    print(json.dumps({'order': name, 'rays': len(b1), 'closest_ms': round(best, 3), 'Mray_s': round(len(b1) / best / 1e3, 1)}), flush=True)
