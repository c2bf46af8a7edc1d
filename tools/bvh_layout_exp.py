"""Storage-order experiment for the BVH records (CRAY_BVH_LAYOUT, cray_scene_upload): DFS pre-order (default) against
breadth-first tops + treelet clustering.  Same scene, same host BVH; per layout: kernel ms of two timed frames, film and
traversal counters compared with the default layout's (must be identical: traversal order is defined by the child refs)."""
import os, sys, json
import numpy as np
sys.path.insert(0, '.')
from craytracer_amd import backend, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else 'dragon'
layouts = sys.argv[2:] or ['0', '7', '15', '31', '63', '15:4096', '31:65536', '255:65536', '0']
sc = {'dragon': lambda: scenes.dragon(), 'staircase': lambda: scenes.staircase(1920, 1080, 256, 12)}[wl]()
ctx = backend.Context(0)
host = backend.HostScene(sc, bvh_ctx=ctx)
ref_film = ref_cnt = None
for lay in layouts:
    if lay == '0':
        os.environ.pop('CRAY_BVH_LAYOUT', None)
    else:
        os.environ['CRAY_BVH_LAYOUT'] = lay
    dev = ctx.upload(host)
    dev.render(seed=0)
    ms = []
    for _ in range(2):
        film, st = dev.render(seed=0)
        ms.append((st['trace_closest_ms'], st['trace_mixed_ms'], st['shade_ms'], st['seconds'] * 1e3))
    _, cst = dev.render(seed=0, sample_range=(0, 2), count_traversal=True)
    cnt = tuple(cst[k] for k in ('closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims', 'closest_rays', 'shadow_rays'))
    if ref_film is None:
        ref_film, ref_cnt = film, cnt
    same = bool(np.array_equal(film, ref_film)) and cnt == ref_cnt
    print(json.dumps({'layout': lay, 'b0_ms': [round(m[0], 2) for m in ms], 'mixed_ms': [round(m[1], 2) for m in ms], 'shade_ms': [round(m[2], 2) for m in ms],
                      'frame_ms': [round(m[3], 2) for m in ms], 'identical_to_default': same}), flush=True)
    dev.close()
