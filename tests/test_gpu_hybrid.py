"""The certified f32 culling of the exact traversal (CRAY_HYBRID=1): 64-B f32 node records and f32 copies of the
triangles decide what they can certify, everything else is retaken from the f64 records.  It must change nothing: hits,
distances and the node / primitive counters against the oracle, films bit for bit against the f64-records context."""
import numpy as np
import pytest

from craytracer_amd import backend
from oracle import oracle_lib as ol
from tests.parity_util import small_scenes, random_rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['1'])
def ctxs(request):
    import os
    old = os.environ.get('CRAY_HYBRID')
    os.environ['CRAY_HYBRID'] = request.param          # read when the context is created
    hyb = backend.Context(0)
    os.environ['CRAY_HYBRID'] = '0'
    ref = backend.Context(0)
    if old is None:
        del os.environ['CRAY_HYBRID']
    else:
        os.environ['CRAY_HYBRID'] = old
    yield hyb, ref
    hyb.close()
    ref.close()


@pytest.mark.parametrize('name', [n for n, _ in small_scenes()])
def test_hybrid_traversal_is_the_reference_traversal(ctxs, name):
    hyb, _ = ctxs
    sc = dict(small_scenes())[name]
    dev = hyb.upload(backend.HostScene(sc))
    orc = ol.OracleScene(sc)
    rays = random_rays(orc, 4000, seed=21)
    # origins exactly on bounding planes, axis-parallel directions: what the f32 side must hand to the exact path
    nodes, _ = orc.bvh()
    rng = np.random.default_rng(3)
    pick = nodes[rng.integers(0, len(nodes), 500)]
    pts = np.where(rng.random((500, 3)) < 0.5, pick['bmin'], pick['bmax'])
    pts = pts[np.all(np.abs(pts) < 1e6, axis=1)]
    extra = np.zeros((len(pts), 7))
    extra[:, :3] = pts
    d = rng.normal(size=(len(pts), 3))
    d[rng.random(len(pts)) < 0.3, 0] = 0.0
    extra[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    extra[:, 6] = np.where(rng.random(len(pts)) < 0.5, np.inf, rng.uniform(1e-3, 10.0, len(pts)))
    rays = np.concatenate([rays, extra])
    g, gst = dev.trace(rays)
    o, ost = orc.trace(rays)
    assert np.array_equal(g['hit'], o['hit'])
    h = o['hit'] != 0
    assert np.array_equal(g['prim'][h], o['prim'][h])
    assert np.array_equal(g['t'][h], o['t'][h])
    assert gst['closest_nodes'] == ost['closest_nodes'] and gst['closest_prims'] == ost['closest_prims']
    ga, gast = dev.trace(rays, any_hit=True)
    oa, oast = orc.trace(rays, any_hit=True)
    assert np.array_equal(ga['hit'], oa['hit'])
    assert gast['shadow_nodes'] == oast['shadow_nodes'] and gast['shadow_prims'] == oast['shadow_prims']
    # the timed kernels (no counting) find the same hits
    gt, gtst = dev.trace(rays, timed=True)
    assert np.array_equal(gt['prim'], g['prim']) and np.array_equal(gt['t'], g['t'])
    gat, gatst = dev.trace(rays, any_hit=True, timed=True)
    assert np.array_equal(gat['hit'], oa['hit'])
    # the per-ray hook reports what its one launch took (cray.h, cray_trace)
    assert gtst['trace_closest_ms'] > 0 and gtst['trace_closest_launches'] == 1 and gtst['trace_any_ms'] == 0
    assert gatst['trace_any_ms'] > 0 and gatst['trace_any_launches'] == 1 and gatst['trace_closest_ms'] == 0
    dev.close()


@pytest.mark.parametrize('name', ['cornell', 'dragon', 'staircase'])
def test_hybrid_film_is_the_default_film(ctxs, name):
    hyb, ref = ctxs
    sc = dict(small_scenes())[name]
    host = backend.HostScene(sc)
    dh, dr = hyb.upload(host), ref.upload(host)
    fh, sh = dh.render(seed=4)
    fr, sr = dr.render(seed=4)
    assert np.array_equal(fh, fr)
    assert sh['closest_rays'] == sr['closest_rays'] and sh['shadow_rays'] == sr['shadow_rays']
    fh2, sh2 = dh.render(seed=4, count_traversal=True)
    fr2, sr2 = dr.render(seed=4, count_traversal=True)
    assert np.array_equal(fh2, fr) and np.array_equal(fr2, fr)
    for k in ('closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims'):
        assert sh2[k] == sr2[k], k
    dh.close(); dr.close()


def _default_context():
    import os
    old = {k: os.environ.pop(k, None) for k in ('CRAY_HYBRID', 'CRAY_RECORDS_B0', 'CRAY_RECORDS_REST')}
    try:
        return backend.Context(0)
    finally:
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v


def test_a_context_that_chooses_its_records_per_scene_renders_the_same_film():
    """Default contexts (CRAY_HYBRID unset) choose a scene's records BEFORE its first frame that is big enough to time, from a few
    small probe passes on both kinds (cray_hip.hip probe_trace_records), per launch kind; cray_stats.trace_records says which
    records a call read.  The choice must never show in the film — every frame equals the oracle's, bit for bit —, it is made once
    (every later frame reads the same records, also after a counting frame, which always reads f64 records), and frames too small
    to time stay on f64 records and choose nothing."""
    ctx = _default_context()
    from craytracer_amd import scenes
    sc = scenes.dragon(512, 288, 16, 6, nu=200, nv=500)          # 2.36 M paths: big enough to be timed
    dev = ctx.upload(backend.HostScene(sc, resident=True))
    ref, _ = ol.OracleScene(sc).render(seed=2)
    seen = []
    for _ in range(3):
        f, st = dev.render(seed=2)
        assert np.array_equal(f, ref)
        seen.append(st['trace_records'])
    assert (seen[0] & 15) in (0, 1) and (seen[0] >> 4) in (0, 1), seen      # the first frame already reads the chosen records
    assert seen[1] == seen[0] and seen[2] == seen[0], seen                   # ... and so does every later one
    info = dev.records_info()                                                # cray_scene_records_info: the choice and what it cost
    assert info['chosen'] == (seen[0] & 15, seen[0] >> 4) and 0.0 < info['probe_ms'] < 2000.0
    assert all(v > 0.0 for k in info['probe_kernel_ms'].values() for v in k.values())
    pool_bytes, pool_paths = ctx.pool_info()                                 # cray_ctx_pool_info: the pool the first frame allocated
    assert pool_paths >= 512 * 288 * 16 and pool_bytes == pool_paths * 340
    fc, stc = dev.render(seed=2, count_traversal=True)                       # counting frames always read f64 records
    assert stc['trace_records'] == 0 and np.array_equal(fc, ref)
    f, st = dev.render(seed=2)                                               # ... and leave the scene's choice alone
    assert st['trace_records'] == seen[0] and np.array_equal(f, ref)
    dev.close()
    small = ctx.upload(backend.HostScene(dict(small_scenes())['cornell']))
    for _ in range(3):
        assert small.render(seed=1)[1]['trace_records'] == 0
    assert small.records_info()['chosen'] == (-1, -1) and small.records_info()['probe_ms'] == 0.0
    small.close()
    ctx.close()


def test_pinned_records_are_what_a_call_reads():
    """CRAY_RECORDS_B0 / CRAY_RECORDS_REST pin the records per launch kind (hosts that want run-to-run identical kernels,
    INTEGRATION.md; the profiling passes): no probe, every frame reads what was pinned, same film."""
    import os
    from craytracer_amd import scenes
    sc = scenes.dragon(512, 288, 16, 6, nu=200, nv=500)
    ref, _ = ol.OracleScene(sc).render(seed=2)
    for b0, rest in (('1', '0'), ('0', '1')):
        old = {k: os.environ.get(k) for k in ('CRAY_HYBRID', 'CRAY_RECORDS_B0', 'CRAY_RECORDS_REST')}
        os.environ.pop('CRAY_HYBRID', None)
        os.environ['CRAY_RECORDS_B0'] = b0
        os.environ['CRAY_RECORDS_REST'] = rest
        try:
            ctx = backend.Context(0)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        dev = ctx.upload(backend.HostScene(sc, resident=True))
        for _ in range(2):
            f, st = dev.render(seed=2)
            assert st['trace_records'] == (int(b0) | (int(rest) << 4)) and np.array_equal(f, ref)
        dev.close()
        ctx.close()


@pytest.mark.parametrize('records', ['0', '1'])
def test_shape_records_in_lds_or_in_global_memory_same_film(records):
    """The traversal blocks stage a scene's sphere / disk records in LDS when there are at most eight of them (four in the
    five-waves instantiations of the certified-f32 records; their own instantiations); a scene with more, or
    CRAY_LDS_SHAPES=0, reads them from global memory.  Same hits either way:
    films of both contexts equal the oracle's, on a scene that fits (two disks, one sphere), on one between the two capacities
    (two disks, four spheres) and on one that does not fit either (twelve spheres and a disk light), with f64 records and with
    certified f32 culling."""
    import os
    from craytracer_amd import scene as S, scenes
    old = {k: os.environ.get(k) for k in ('CRAY_LDS_SHAPES', 'CRAY_HYBRID')}
    os.environ['CRAY_HYBRID'] = records
    try:
        os.environ['CRAY_LDS_SHAPES'] = '1'
        staged = backend.Context(0)
        os.environ['CRAY_LDS_SHAPES'] = '0'
        plain = backend.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    cam = S.Camera.perspective(S.Film(96, 64), (0, 3, -9), (0, 1, 4), (0, 1, 0), 40)
    matte = S.Material.new_matte(S.Color(0.7, 0.6, 0.5), 0.0)
    glass = S.Material.new_glass(S.Color(1, 1, 1), S.Color(0.8, 0.8, 0.8), 1.5)
    light = S.Shape.new_disk((0, 8, 4), 90, 0, 2.5, 0)
    prims = [S.Primitive.new(S.Shape.new_disk((0, 0, 4), 90, 0, 30, 0), matte),
             S.Primitive.new_area_light(light, S.Light.Area(light, S.Color(12, 11, 9)))]
    for i in range(12):
        prims.append(S.Primitive.new(S.Shape.new_sphere((-5.5 + i, 0.45 + 0.3 * (i % 3), 2.0 + 0.5 * (i % 4)), 0.45), glass if i % 2 else matte))
    many = S.Scene(5, 8, cam, [S.Light.Infinite(S.Color(0.05, 0.07, 0.1))], prims)
    # six records: staged by the f64-record instantiations (room for eight), read from global memory by the five-waves ones (room for four)
    some = S.Scene(5, 8, cam, [S.Light.Infinite(S.Color(0.05, 0.07, 0.1))], prims[:6])
    for sc in (scenes.simple(64, 48, 8, 4), some, many):
        ref, _ = ol.OracleScene(sc).render(seed=6)
        for ctx in (staged, plain):
            dev = ctx.upload(backend.HostScene(sc))
            f, st = dev.render(seed=6)
            assert st['trace_records'] == int(records) * 0x11
            assert np.array_equal(f, ref)
            dev.close()
    staged.close(); plain.close()


@pytest.mark.parametrize('name', ['test', 'staircase', 'dragon'])
def test_shading_tables_in_lds_or_in_global_memory_same_film(name):
    """k_shade has two instantiations: the small shading tables staged in LDS (when they fit) or read from global memory
    (CRAY_SHADE_LDS=0 at upload forces the latter, which is what a scene with thousands of materials gets).  Same film, bit for bit."""
    import os
    sc = dict(small_scenes())[name]
    ctx = backend.Context(0)
    host = backend.HostScene(sc)
    a = ctx.upload(host)
    old = os.environ.get('CRAY_SHADE_LDS')
    os.environ['CRAY_SHADE_LDS'] = '0'
    try:
        b = ctx.upload(host)
    finally:
        if old is None:
            del os.environ['CRAY_SHADE_LDS']
        else:
            os.environ['CRAY_SHADE_LDS'] = old
    fa, sa = a.render(seed=9)
    fb, sb = b.render(seed=9)
    assert np.array_equal(fa, fb)
    assert sa['closest_rays'] == sb['closest_rays'] and sa['shadow_rays'] == sb['shadow_rays']
    ref, _ = ol.OracleScene(sc).render(seed=9)
    assert np.array_equal(fa, ref)
    a.close(); b.close(); ctx.close()
