import numpy as np, sys
sys.path.insert(0, '.')
from craytracer_amd import backend, scenes
from oracle import oracle_lib as ol
ctx = backend.Context(0)
W=H=48
for depth in [1,2,3,4,8]:
    sc = scenes.cornell(W,H,8,depth)
    host = backend.HostScene(sc); dev = ctx.upload(host); orc = ol.OracleScene(sc)
    L = dev.render_samples((0,8), seed=0)
    ref = np.zeros_like(L)
    for y in range(H):
        for x in range(W):
            for s in range(8):
                ref[y,x,s] = orc.render_pixel(x,y,s,seed=0)
    neq = np.any(L != ref, axis=-1)
    rel = np.abs(L-ref)/np.maximum(np.abs(ref),1e-6)
    print('depth', depth, 'paths', neq.size, 'not bit-equal', int(neq.sum()), 'max rel', rel.max(), 'n rel>1e-9', int((rel.max(axis=-1)>1e-9).sum()))
    idx = np.argwhere(rel.max(axis=-1)>1e-9)
    for (y,x,s) in idx[:5]:
        print('   path', x,y,s, 'gpu', L[y,x,s], 'ref', ref[y,x,s])
