#!/usr/bin/env python3
"""Time Bvh::new for the bench scene: host (one thread, the reference's recursion) vs cray_bvh_build_sah on the GPU,
and check that both give the same tree.   python tools/bvh_build_timing.py [nu nv]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from craytracer_amd import backend, scenes

nu, nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1200, 3000)
sc = scenes.dragon(width=64, height=36, spp=1, max_depth=2, nu=nu, nv=nv)
ctx = backend.Context(0)
t0 = time.time(); g = backend.HostScene(sc, bvh_ctx=ctx); t1 = time.time()
g2 = backend.HostScene(sc, bvh_ctx=ctx); t2 = time.time()
h = backend.HostScene(sc); t3 = time.time()
gn, gr = g2.bvh(); hn, hr = h.bvh()
same = len(gn) == len(hn) and all(np.array_equal(gn[f], hn[f]) for f in gn.dtype.names) and np.array_equal(gr, hr)
print(json.dumps({'triangles': len(sc.triangles), 'nodes': int(len(hn)), 'same_tree': bool(same),
                  'host_bvh_s': round(h.bvh_seconds, 3), 'gpu_bvh_s_first_call': round(g.bvh_seconds, 3),
                  'gpu_bvh_s': round(g2.bvh_seconds, 3), 'gpu_build': g2.gpu_build,
                  'scene_new_s': {'host': round(t3 - t2, 2), 'gpu_builder': round(t2 - t1, 2)}}))
