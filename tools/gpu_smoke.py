import numpy as np, time, sys
sys.path.insert(0, '.')
from craytracer_amd import backend, scenes
from oracle import oracle_lib as ol
from tests.parity_util import small_scenes, random_rays, rmse
ctx = backend.Context(0)
for name, sc in small_scenes():
    host = backend.HostScene(sc); dev = ctx.upload(host); orc = ol.OracleScene(sc)
    rays = random_rays(orc, 2000, 1)
    g, gst = dev.trace(rays); o, ost = orc.trace(rays)
    print(name, 'hit eq', np.array_equal(g['hit'],o['hit']), 'prim eq', np.array_equal(g['prim'],o['prim']), 't eq', np.array_equal(g['t'],o['t']),
          'loc eq', np.array_equal(g['location'],o['location']), 'nrm eq', np.array_equal(g['normal'],o['normal']), 'uv maxdiff', np.abs(g['uv']-o['uv']).max(),
          'nodes', gst['closest_nodes'], ost['closest_nodes'], 'prims', gst['closest_prims'], ost['closest_prims'], flush=True)
    g2,_ = dev.trace(rays, any_hit=True); o2,_ = orc.trace(rays, any_hit=True)
    print('   any eq', np.array_equal(g2['hit'], o2['hit']))
    t=time.time(); gi, st = dev.render(seed=0, count_traversal=True); tg=time.time()-t
    oi, ost = orc.render(seed=0)
    print('   render rmse %.3e maxabs %.3e'%(rmse(gi,oi), np.abs(gi-oi).max()), 'exact frac %.4f'%np.mean(gi==oi), 'gpu %.3fs'%st['seconds'], 'rays', st['closest_rays'], ost['closest_rays'], st['shadow_rays'], ost['shadow_rays'], 'nonfinite', st['nonfinite'], flush=True)
