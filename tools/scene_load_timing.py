"""Scene::new + upload at configs[2] (7.2 M triangles): host BVH + upload, GPU BVH through the host + upload, resident build.
usage: python tools/scene_load_timing.py"""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from craytracer_amd import backend, scenes
import torch
sc = scenes.dragon()
ctx = backend.Context(0)
ctx.upload(backend.HostScene(scenes.simple(8, 8, 1, 1))).close()   # warm the context
out = {}
films = {}
for mode in ('resident', 'gpu_bvh_via_host', 'host_bvh', 'resident'):
    t0 = time.time()
    host = backend.HostScene(sc, resident=True) if mode == 'resident' else backend.HostScene(sc, bvh_ctx=ctx if mode == 'gpu_bvh_via_host' else None)
    t1 = time.time()
    dev = ctx.upload(host)
    torch.cuda.synchronize()
    t2 = time.time()
    film, st = dev.render(seed=0, sample_range=(0, 2), count_traversal=True)
    films[mode] = (film, tuple(st[k] for k in ('closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims')))
    out[mode] = {'scene_new_s': round(t1 - t0, 3), 'upload_s': round(t2 - t1, 3), 'total_s': round(t2 - t0, 3),
                 'bvh_kernels_s': round(dev.build_stats['device_seconds'] if mode == 'resident' else host.gpu_build['device_seconds'], 4)}
    print(mode, out[mode], flush=True)
    dev.close(); host.close()
same = all(np.array_equal(films['host_bvh'][0], f[0]) and films['host_bvh'][1] == f[1] for f in films.values())
print(json.dumps({'identical_films_and_counters': same, **out}))
