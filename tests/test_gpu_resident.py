"""Resident build (cray_host_scene_new_resident + cray_flat_scene.build_on_device): Bvh::new runs on the GPU inside
cray_scene_upload and the tree never visits the host (src/scene.rs:25-53, src/bvh.rs:38-56, 234-336).  It must give the scene
the ordinary path gives: the same film bit for bit, the same traversal counters (= the same tree and leaf order), per-ray
records equal, on every parity scene and on a 300 k-triangle mesh; a scene with Distant / Infinite lights (world radius from
the host's union of boxes) included."""
import numpy as np
import pytest

from craytracer_amd import backend, scenes
from oracle import oracle_lib as ol
from tests.parity_util import random_rays, small_scenes

pytestmark = pytest.mark.gpu
COUNTERS = ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims')


@pytest.fixture(scope='module')
def ctx():
    c = backend.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize('name', [n for n, _ in small_scenes()])
def test_resident_scene_equals_the_uploaded_one(ctx, name):
    sc = dict(small_scenes())[name]
    ref = ctx.upload(backend.HostScene(sc))                 # host Bvh::new, tree uploaded
    host = backend.HostScene(sc, resident=True)
    assert host.flat.n_nodes == 0 and host.flat.build_on_device == 1
    dev = ctx.upload(host)                                  # Bvh::new + layout on the GPU
    a, ast = ref.render(seed=5, count_traversal=True)
    b, bst = dev.render(seed=5, count_traversal=True)
    for k in COUNTERS:
        assert ast[k] == bst[k], k
    assert np.array_equal(a, b)
    o, ost = ol.OracleScene(sc).render(seed=5)
    assert np.array_equal(b, o) and all(bst[k] == ost[k] for k in COUNTERS)
    orc = ol.OracleScene(sc)
    rays = random_rays(orc, 400, 3)
    ha, _ = ref.trace(rays)
    hb, _ = dev.trace(rays)
    for field in ('hit', 'prim', 't', 'location', 'normal'):
        assert np.array_equal(ha[field], hb[field]), field
    assert dev.device_bytes == ref.device_bytes
    ref.close(); dev.close()


def test_resident_build_of_a_large_mesh(ctx):
    sc = scenes.dragon(160, 90, 4, 8, nu=250, nv=600)       # 300 k triangles + sphere + disk
    ref = ctx.upload(backend.HostScene(sc, bvh_ctx=ctx))
    dev = ctx.upload(backend.HostScene(sc, resident=True))
    assert dev.build_stats['leaves'] > 100_000 and dev.build_stats['device_seconds'] > 0
    a, ast = ref.render(seed=0, count_traversal=True)
    b, bst = dev.render(seed=0, count_traversal=True)
    assert np.array_equal(a, b) and all(ast[k] == bst[k] for k in COUNTERS)
    ref.close(); dev.close()


def test_resident_build_reports_the_reference_s_build_panics(ctx):
    from craytracer_amd import scene as S
    cam = S.Camera.perspective(S.Film(8, 8), (0, 0, -5), (0, 0, 0), (0, 1, 0), 40)
    m = S.Material.new_matte(S.Color(1, 1, 1), 0.0)
    # more than four primitives with one and the same centroid: the SAH split has nowhere to cut (bvh.rs:304 / 327-328)
    tri = [S.Primitive.new(S.Shape.new_triangle((0, 0, 0), (1, 0, 0), (0, 1, 0)), m) for _ in range(6)]
    sc = S.Scene(3, 4, cam, [S.Light.Point((0, 3, 0), S.Color(1, 1, 1))], tri)
    with pytest.raises(backend.CrayError):
        backend.HostScene(sc)                               # the host builder reports it ...
    host = backend.HostScene(sc, resident=True)
    with pytest.raises(backend.CrayError) as e:
        ctx.upload(host)                                    # ... and so does the build inside the upload
    assert 'panic' in str(e.value)
