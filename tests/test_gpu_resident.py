"""Resident build (cray_host_scene_new_resident + cray_flat_scene.build_on_device): Bvh::new runs on the GPU inside
cray_scene_upload and the tree never visits the host (src/scene.rs:25-53, src/bvh.rs:38-56, 234-336).  It must give the scene
the ordinary path gives: the same film bit for bit, the same traversal counters (= the same tree and leaf order), per-ray
records equal, on every parity scene and on a 300 k-triangle mesh; a scene with Distant / Infinite lights (world radius from
the host's union of boxes) included."""
import ctypes as C
import types

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


PRIM_BOUND_DT = np.dtype([('prim', '<u4'), ('pad_', '<u4'), ('bmin', '<f8', 3), ('bmax', '<f8', 3)])


def _tampered(host, edit):
    """A copy of a resident flat scene with its primitive table / other_bounds edited (the arrays the C ABI takes as they come)."""
    from craytracer_amd import scene as S
    flat = backend.FlatScene.from_buffer_copy(host.flat)
    prims = np.ctypeslib.as_array(C.cast(flat.prims, C.POINTER(C.c_uint8)), shape=(flat.n_prims * 16,)).view(S.PRIM_DT).copy()
    other = np.ctypeslib.as_array(C.cast(flat.other_bounds, C.POINTER(C.c_uint8)), shape=(flat.n_other_bounds * 56,)).view(PRIM_BOUND_DT).copy()
    prims, other = edit(prims, other)
    flat.prims, flat.n_prims = prims.ctypes.data, len(prims)
    flat.other_bounds, flat.n_other_bounds = other.ctypes.data, len(other)
    return types.SimpleNamespace(flat=flat, keep=(prims, other, host))


@pytest.mark.parametrize('case,needle', [
    ('kind', 'bad shape'), ('sphere_index', 'bad shape'), ('disk_index', 'bad shape'), ('triangle_index', 'bad triangle index'),
    ('missing_box', 'needs its box'), ('duplicate_box', 'already has a box'), ('nan_box', 'non-finite box'),
    ('box_of_triangle', 'is a triangle'), ('box_prim_range', 'out of range')])
def test_resident_upload_rejects_malformed_primitive_tables(ctx, case, needle):
    """cray_scene_upload with build_on_device = 1 makes the checks the tree-upload path makes in fill_slot (shape kind, shape index
    per kind) plus one finite box per sphere / disk: a bad table is CRAY_ERR_INVALID, not an out-of-bounds read in k_trace."""
    host = backend.HostScene(scenes.dragon(64, 48, 4, 4, nu=20, nv=40), resident=True)   # triangles + a sphere + a disk light
    kinds = np.ctypeslib.as_array(C.cast(host.flat.prims, C.POINTER(C.c_int32)), shape=(host.flat.n_prims * 4,))[0::4]
    i_sph, i_dsk, i_tri = int(np.argmax(kinds == 0)), int(np.argmax(kinds == 2)), int(np.argmax(kinds == 1))
    assert kinds[i_sph] == 0 and kinds[i_dsk] == 2 and kinds[i_tri] == 1

    def edit(prims, other):
        if case == 'kind':
            prims['shape_kind'][i_tri] = 7
        elif case == 'sphere_index':
            prims['shape'][i_sph] = host.flat.n_spheres
        elif case == 'disk_index':
            prims['shape'][i_dsk] = host.flat.n_disks + 5
        elif case == 'triangle_index':
            prims['shape'][i_tri] = host.flat.n_triangles
        elif case == 'missing_box':
            other = other[other['prim'] != i_sph]
        elif case == 'duplicate_box':
            other = np.concatenate([other, other[:1]])
        elif case == 'nan_box':
            other['bmax'][0, 1] = np.nan
        elif case == 'box_of_triangle':
            other['prim'][0] = i_tri
        elif case == 'box_prim_range':
            other['prim'][0] = host.flat.n_prims
        return prims, other

    bad = _tampered(host, edit)
    with pytest.raises(backend.CrayError) as e:
        backend.DeviceScene(ctx, bad)
    assert needle in str(e.value), str(e.value)
    ok = _tampered(host, lambda p, o: (p, o))                  # the untouched copy uploads and renders like the original
    a, _ = backend.DeviceScene(ctx, ok).render(seed=0)
    b, _ = ctx.upload(host).render(seed=0)
    assert np.array_equal(a, b)
