"""Reference scene files (scenes/{simple,test,materials}.cry, fixtures under tests/golden) through
the whole product path: .cry reader -> Scene::new mirror -> GPU, against the oracle fed with the
same parsed description.  materials.cry covers Oren-Nayar at several sigmas, four metals, four
glasses, four plastics, a spherical area light and an Infinite light."""
import os

import numpy as np
import pytest

from craytracer_amd import backend, cry
from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'scenes')


@pytest.mark.parametrize('name,w,h,spp', [('simple', 70, 40, 8), ('test', 50, 50, 8), ('materials', 64, 42, 8)])
def test_parsed_scene_renders_pixel_exact(name, w, h, spp):
    sc = cry.load_scene_file(os.path.join(GOLDEN, name + '.cry'), width=w, height=h, num_samples=spp)
    ctx = backend.Context(0)
    dev = ctx.upload(backend.HostScene(sc))
    orc = ol.OracleScene(sc)
    g, gst = dev.render(seed=7, count_traversal=True)
    o, ost = orc.render(seed=7)
    for k in ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims'):
        assert gst[k] == ost[k], k
    rmse = float(np.sqrt(np.mean((g.astype(np.float64) - o) ** 2)))
    assert rmse < 1e-4, rmse          # north-star tolerance
    assert np.array_equal(g, o)       # and in fact every f32 is equal
    assert gst['nonfinite'] == 0
    dev.close()
    ctx.close()
