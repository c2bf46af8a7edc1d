"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle.

Bars (north_star): BVH queries bit-exact (hit, primitive, distance, location, normal);
uv of spheres/disks within 1e-12 (atan2/acos come from OCML vs glibc); rendered pixels
RMSE < 1e-4 against the oracle on identical seeds.
"""
import numpy as np
import pytest

from craytracer_amd import backend
from oracle import oracle_lib as ol
from tests.parity_util import small_scenes, random_rays, rmse

pytestmark = pytest.mark.gpu

PIXEL_RMSE_TOL = 1e-4   # BASELINE.json north_star: pixel RMSE < 1e-4 vs the CPU reference arithmetic
UV_TOL = 1e-12          # sphere/disk uv go through atan2/acos (libm vs OCML)


@pytest.fixture(scope='module')
def ctx():
    c = backend.Context(0)
    yield c
    c.close()


@pytest.fixture(scope='module', params=[n for n, _ in small_scenes()])
def setup(request, ctx):
    sc = dict(small_scenes())[request.param]
    host = backend.HostScene(sc)
    dev = ctx.upload(host)
    orc = ol.OracleScene(sc)
    yield request.param, sc, host, dev, orc
    dev.close()


def test_closest_hit_bit_exact(setup):
    name, sc, host, dev, orc = setup
    rays = random_rays(orc, 4000, seed=11)
    g, gst = dev.trace(rays)
    o, ost = orc.trace(rays)
    assert np.array_equal(g['hit'], o['hit'])
    assert np.array_equal(g['prim'], o['prim'])
    assert np.array_equal(g['t'], o['t'])                      # f64 distances, bit for bit
    assert np.array_equal(g['location'], o['location'])
    assert np.array_equal(g['normal'], o['normal'])
    assert np.max(np.abs(g['uv'] - o['uv'])) <= UV_TOL
    # the traversal visits exactly the nodes / primitives the reference visits
    assert gst['closest_nodes'] == ost['closest_nodes']
    assert gst['closest_prims'] == ost['closest_prims']
    assert gst['stack_overflow'] == 0
    assert g['hit'].sum() > 100


def test_any_hit_exact(setup):
    name, sc, host, dev, orc = setup
    rays = random_rays(orc, 4000, seed=12)
    rng = np.random.default_rng(5)
    rays[::3, 6] = rng.uniform(0.01, 50.0, len(rays[::3]))      # finite shadow-ray lengths too
    g, gst = dev.trace(rays, any_hit=True)
    o, ost = orc.trace(rays, any_hit=True)
    assert np.array_equal(g['hit'], o['hit'])
    assert gst['shadow_nodes'] == ost['shadow_nodes']
    assert gst['shadow_prims'] == ost['shadow_prims']


def test_per_path_radiance(setup):
    name, sc, host, dev, orc = setup
    L = dev.render_samples((0, 4), seed=3)
    H, W = L.shape[:2]
    rng = np.random.default_rng(1)
    worst = 0.0
    n_exact = 0
    for _ in range(300):
        x, y, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 4))
        ref = orc.render_pixel(x, y, s, seed=3)
        got = L[y, x, s]
        denom = np.maximum(np.abs(ref), 1e-3)
        worst = max(worst, float(np.max(np.abs(got - ref) / denom)))
        n_exact += int(np.array_equal(got, ref))
    assert worst < 1e-9, (name, worst)
    assert n_exact > 150, (name, n_exact)


def test_rendered_pixels(setup):
    name, sc, host, dev, orc = setup
    g, gst = dev.render(seed=0, count_traversal=True)
    o, ost = orc.render(seed=0)
    assert gst['closest_rays'] == ost['closest_rays']
    assert gst['shadow_rays'] == ost['shadow_rays']
    assert gst['closest_nodes'] == ost['closest_nodes']
    assert gst['closest_prims'] == ost['closest_prims']
    assert gst['shadow_nodes'] == ost['shadow_nodes']
    assert gst['shadow_prims'] == ost['shadow_prims']
    assert gst['nonfinite'] == 0 and gst['stack_overflow'] == 0
    assert rmse(g, o) < PIXEL_RMSE_TOL, (name, rmse(g, o))
    assert float(np.mean(g)) > 1e-3


def test_tile_sharding_reassembles(setup):
    name, sc, host, dev, orc = setup
    full, _ = dev.render(seed=1)
    acc = np.zeros_like(full)
    for r in range(3):
        part, _ = dev.render(seed=1, rank=r, world_size=3)
        acc += part
    assert np.array_equal(acc, full)


def test_small_path_pool_matches(setup):
    name, sc, host, dev, orc = setup
    a, _ = dev.render(seed=2)
    b, _ = dev.render(seed=2, max_paths_in_flight=1000)
    assert np.array_equal(a, b)


def test_fast_path_film_is_the_counting_path_film_and_the_oracle_film(setup):
    """The timed configuration (zero-term shadow rays skipped, no light sampling for materials without a diffuse
    lobe, shadow rays of bounce b traced together with the segments of bounce b+1) must produce the very film of
    the instrumented configuration (every reference query traced, one launch per query kind) — and that film is
    the oracle's, pixel for pixel."""
    name, sc, host, dev, orc = setup
    fast, fst = dev.render(seed=0)
    slow, sst = dev.render(seed=0, count_traversal=True)
    part, pst = dev.render(seed=0, count_traversal=2)
    o, ost = orc.render(seed=0)
    assert np.array_equal(fast, slow) and np.array_equal(fast, part)
    assert np.array_equal(fast, o.astype(np.float32))
    assert fst['trace_mixed_launches'] > 0 and sst['trace_mixed_launches'] == 0
    # the reference's query counts are reported in every mode; skipped ones are a subset of the shadow rays
    for st in (fst, sst, pst):
        assert st['closest_rays'] == ost['closest_rays'] and st['shadow_rays'] == ost['shadow_rays']
    assert sst['shadow_skipped'] == 0 and fst['shadow_skipped'] == pst['shadow_skipped'] <= fst['shadow_rays']
    # count_traversal=2 counts only what the fast path traverses: same closest work, no more shadow work
    assert pst['closest_nodes'] == sst['closest_nodes'] and pst['closest_prims'] == sst['closest_prims']
    assert pst['shadow_nodes'] <= sst['shadow_nodes'] and pst['shadow_prims'] <= sst['shadow_prims']
    if pst['shadow_skipped'] == 0:
        assert pst['shadow_nodes'] == sst['shadow_nodes']


def test_bvh_deeper_than_the_fast_traversal_stack_still_matches_the_oracle():
    """220 nested triangles at x = 4^i (each large enough to fill the camera's cone) make the reference's SAH tree
    a 120-level chain; a camera ray keeps one far child pending per level, more than the 96 entries of LDS + scratch.
    The runtime then adds a third stack level in HBM and renders again: the film must be the oracle's."""
    from craytracer_amd import scene as S
    tris = np.array([[(x, -x - 1, -x - 1), (x, 3 * x + 3, -x - 1), (x, -x - 1, 3 * x + 3)] for x in (4.0 ** i for i in range(220))], dtype=float)
    white = S.Material.new_matte(S.Color(1, 1, 1), 0.0)
    cam = S.Camera.perspective(S.Film(8, 8), (-5, 0.2, 0.2), (1, 0.2, 0.2), (0, 1, 0), 20)
    sc = S.Scene(3, 2, cam, [S.Light.Point((0, 5, 0), S.Color.WHITE)], [S.Mesh(S.triangles_flat(tris), material=white)])
    ctx = backend.Context(0)
    dev = ctx.upload(backend.HostScene(sc))
    orc = ol.OracleScene(sc)
    g, gst = dev.render(seed=0, count_traversal=True)
    o, ost = orc.render(seed=0)
    assert gst['stack_overflow'] == 0
    assert np.array_equal(g, o.astype(np.float32))
    for k in ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims'):
        assert gst[k] == ost[k], k
    rays = random_rays(orc, 512, seed=3, scale=50.0)
    gh, _ = dev.trace(rays)
    oh, _ = orc.trace(rays)
    assert np.array_equal(gh['hit'], oh['hit']) and np.array_equal(gh['t'], oh['t'])
    dev.close()
    ctx.close()


@pytest.mark.parametrize('n_prims', [1, 2])
def test_tiny_scenes_with_a_leaf_root_or_one_split(n_prims):
    """A BVH that is a single leaf (one primitive) or one interior node over two leaves: the degenerate ends of
    the traversal (root reference is a leaf; empty stack after the first step)."""
    from craytracer_amd import scene as S
    s_light = S.Shape.new_disk((0, 3, 0), 90, 0, 1.5, 0)
    prims = [S.Primitive.new_area_light(s_light, S.Light.Area(s_light, S.Color(4, 4, 4)))]
    if n_prims == 2:
        prims.append(S.Primitive.new(S.Shape.new_sphere((0, 0, 0), 1.0), S.Material.new_matte(S.Color(0.8, 0.6, 0.4), 20.0)))
    cam = S.Camera.perspective(S.Film(24, 16), (0, 1, -6), (0, 0.5, 0), (0, 1, 0), 50)
    sc = S.Scene(4, 4, cam, [], prims)
    ctx = backend.Context(0)
    host = backend.HostScene(sc, bvh_ctx=ctx)
    assert len(host.bvh()[0]) == (1 if n_prims == 1 else 3)
    dev = ctx.upload(host)
    g, gst = dev.render(seed=2, count_traversal=True)
    o, ost = ol.OracleScene(sc).render(seed=2)
    assert np.array_equal(g, o.astype(np.float32))
    for k in ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims'):
        assert gst[k] == ost[k], k
    assert np.array_equal(dev.render(seed=2)[0], g)
    dev.close()
    ctx.close()


@pytest.mark.parametrize('name', ['cornell', 'test', 'dragon'])
def test_adversarial_rays_bit_exact(name):
    """Rays the slab test's corner cases are made of: axis-parallel directions (zero and negative-zero components:
    0/0 = NaN inside the reference's min/max, and the kernel's plain-division path since the exact-FMA guard rejects
    d = 0), origins exactly on bounding planes / vertices / inside boxes, tiny and huge max distances, denormal
    direction components.  Hits, distances, locations, normals and the node / primitive counters must equal the oracle's."""
    sc = dict(small_scenes())[name]
    ctx = backend.Context(0)
    dev = ctx.upload(backend.HostScene(sc))
    orc = ol.OracleScene(sc)
    nodes, _ = orc.bvh()
    rng = np.random.default_rng(5)
    # points of interest: corners / face centres of random BVH boxes and of the root
    pick = nodes[rng.integers(0, len(nodes), 400)]
    pts = [np.where(rng.random((len(pick), 3)) < 0.5, pick['bmin'], pick['bmax']),
           (pick['bmin'] + pick['bmax']) * 0.5,
           np.where(rng.random((len(pick), 3)) < 0.5, pick['bmin'], (pick['bmin'] + pick['bmax']) * 0.5)]
    pts = np.concatenate(pts)
    pts = pts[np.all(np.abs(pts) < 1e6, axis=1)]
    dirs = []
    for ax in range(3):
        for sgn in (1.0, -1.0):
            d = np.zeros(3); d[ax] = sgn; dirs.append(d)
            d = np.zeros(3); d[ax] = sgn; d[(ax + 1) % 3] = -0.0; dirs.append(d)          # negative zero component
            d = np.zeros(3); d[ax] = sgn; d[(ax + 2) % 3] = 5e-324; dirs.append(d)        # denormal component
            d = np.zeros(3); d[ax] = sgn * np.sqrt(0.5); d[(ax + 1) % 3] = np.sqrt(0.5); dirs.append(d)  # in a coordinate plane
    dirs = np.array(dirs)
    n = len(pts)
    rays = np.zeros((n * 3, 7))
    for k, tmax in enumerate((np.inf, 1e-6, 3.0)):
        rays[k * n:(k + 1) * n, :3] = pts
        rays[k * n:(k + 1) * n, 3:6] = dirs[rng.integers(0, len(dirs), n)]
        rays[k * n:(k + 1) * n, 6] = tmax
    # origins pushed off the plane by one ulp in both directions
    extra = rays[:n].copy()
    extra[:, :3] = np.nextafter(extra[:, :3], np.where(rng.random((n, 3)) < 0.5, -np.inf, np.inf))
    rays = np.concatenate([rays, extra])
    g, gst = dev.trace(rays)
    o, ost = orc.trace(rays)
    assert np.array_equal(g['hit'], o['hit'])
    h = o['hit'] != 0
    assert np.array_equal(g['prim'][h], o['prim'][h])
    for f in ('t', 'location', 'normal'):
        assert np.array_equal(g[f][h], o[f][h]), f
    assert gst['closest_nodes'] == ost['closest_nodes'] and gst['closest_prims'] == ost['closest_prims']
    ga, gast = dev.trace(rays, any_hit=True)
    oa, oast = orc.trace(rays, any_hit=True)
    assert np.array_equal(ga['hit'], oa['hit'])
    assert gast['shadow_nodes'] == oast['shadow_nodes'] and gast['shadow_prims'] == oast['shadow_prims']
    assert int(h.sum()) > 20
    dev.close()
    ctx.close()


def _needle_rays(rng, v0, e1, e2, n, far):
    """rays that cross the triangle's plane near v0 + a e1 + b e2 with small a, b — from `far` away along a direction that is not
    in the plane — and as many that miss on either side"""
    nrm = np.cross(e1 / np.linalg.norm(e1), e2 / np.linalg.norm(e2))
    nrm /= np.linalg.norm(nrm)
    rays = []
    for _ in range(n):
        a, b = rng.uniform(-0.2, 0.6, 2)
        target = v0 + a * e1 + b * e2
        d = -(nrm + 0.3 * rng.normal(size=3))
        d /= np.linalg.norm(d)
        rays.append(np.concatenate([target - d * far, d, [np.inf]]))
    return np.array(rays)


@pytest.mark.parametrize('case', ['ordinary', 'denominator_near_the_top', 'denominator_just_above_epsilon_seen_from_1e280', 'large'])
def test_triangle_quotients_at_the_ends_of_the_exponent_range(case):
    """The triangle test's three quotients share one refined reciprocal of their denominator (tri_test_shared, cray_shading.h)
    when v_div_scale leaves the denominator alone, and fall back to a plain division when it does not.  One-triangle scenes
    (the tree is a single leaf: every ray runs the test) whose operands reach both: a denominator near 2^1020 (edges of
    2^620 and 2^400, rays that cross next to the corner), a denominator just above EPSILON under numerators near 1e280
    (exponent gap > 768), and ordinary ones.  Hits, distances and barycentric-dependent locations must equal the oracle's,
    whose divisions are the CPU's."""
    from craytracer_amd import scene as S
    rng = np.random.default_rng(17)
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    b = np.cross(a, rng.normal(size=3)); b /= np.linalg.norm(b)
    if case == 'ordinary':
        v0, e1, e2, far = rng.normal(size=3), 3.0 * a + 0.5 * b, 2.0 * b, 7.0
    elif case == 'large':
        v0, e1, e2, far = rng.normal(size=3) * 2.0 ** 300, 2.0 ** 300 * (a + 0.2 * b), 2.0 ** 299 * b, 2.0 ** 302
    elif case == 'denominator_near_the_top':
        v0, e1, e2, far = rng.normal(size=3), 2.0 ** 620 * a, 2.0 ** 400 * b, 3.0
    else:
        v0, e1, e2, far = rng.normal(size=3), 3e-4 * a, 2e-4 * b, 1e280
    if case == 'denominator_near_the_top':
        # cross the plane within a few units of the corner: u ~ 2^-620, v ~ 2^-400
        rays = []
        nrm = np.cross(a, b)
        for _ in range(256):
            target = v0 + rng.uniform(-0.5, 2.0) * a + rng.uniform(-0.5, 2.0) * b
            d = -(nrm + 0.3 * rng.normal(size=3)); d /= np.linalg.norm(d)
            rays.append(np.concatenate([target - d * far, d, [np.inf]]))
        rays = np.array(rays)
    else:
        rays = _needle_rays(rng, v0, e1, e2, 256, far)
    tris = np.array([[v0, v0 + e1, v0 + e2]])
    white = S.Material.new_matte(S.Color(1, 1, 1), 0.0)
    cam = S.Camera.perspective(S.Film(8, 8), (-5, 0.2, 0.2), (1, 0.2, 0.2), (0, 1, 0), 20)
    sc = S.Scene(2, 1, cam, [S.Light.Point((0, 5, 0), S.Color.WHITE)], [S.Mesh(S.triangles_flat(tris), material=white)])
    ctx = backend.Context(0)
    dev = ctx.upload(backend.HostScene(sc))
    orc = ol.OracleScene(sc)
    g, gst = dev.trace(rays)
    o, ost = orc.trace(rays)
    assert np.array_equal(g['hit'], o['hit'])
    h = o['hit'] != 0
    assert 20 < int(h.sum()) < len(rays) - 20, int(h.sum())
    assert np.array_equal(g['t'][h], o['t'][h])
    assert np.array_equal(g['location'][h], o['location'][h])
    assert gst['closest_prims'] == ost['closest_prims']
    ga, _ = dev.trace(rays, any_hit=True)
    oa, _ = orc.trace(rays, any_hit=True)
    assert np.array_equal(ga['hit'], oa['hit'])
    dev.close()
    ctx.close()


def test_ragged_film_and_sample_counts():
    """Film sizes that are no multiple of the 64x64 tile, a sample count that is no multiple of the 8-sample batch,
    and a path pool that cuts both: the film must still be the oracle's, pixel for pixel."""
    from craytracer_amd import scenes
    sc = scenes.test_scene(37, 23, 13, 5, with_infinite=True, with_point=True)
    ctx = backend.Context(0)
    dev = ctx.upload(backend.HostScene(sc))
    o, _ = ol.OracleScene(sc).render(seed=4)
    for pool in (0, 777, 8 * 37):
        g, st = dev.render(seed=4, max_paths_in_flight=pool)
        assert np.array_equal(g, o.astype(np.float32)), pool
        assert st['paths'] == 37 * 23 * 13
    # three ranks, ragged tiles: the shares add up to the film
    acc = np.zeros_like(o, dtype=np.float32)
    for r in range(3):
        part, _ = dev.render(seed=4, rank=r, world_size=3, max_paths_in_flight=500)
        acc += part
    assert np.array_equal(acc, o.astype(np.float32))
    dev.close()
    ctx.close()


def test_limits_are_reported_not_truncated():
    """sobol_burley indexes 2^16 samples and 256 dimensions (max_depth <= 31): beyond that the reference's crate asserts;
    here the calls fail with CRAY_ERR_UNSUPPORTED instead of rendering something else."""
    from craytracer_amd import scenes
    ctx = backend.Context(0)
    with pytest.raises(backend.CrayError, match='dimension'):
        ctx.upload(backend.HostScene(scenes.test_scene(8, 8, 1, 32)))
    dev = ctx.upload(backend.HostScene(scenes.test_scene(4, 4, 70000, 2)))
    with pytest.raises(backend.CrayError, match='2\\^16'):
        dev.render(seed=0)
    dev.close()
    ctx.close()


@pytest.mark.parametrize('kind', ['orthographic', 'orthographic_lens', 'perspective_lens'])
def test_other_cameras(kind):
    """Camera::orthographic and the square thin lens (camera.rs:100-162) on a non-square film (the reference's
    `film_height = film.width` quirk shifts the principal point): film and camera matrices equal the oracle's."""
    from craytracer_amd import scene as S
    s_light = S.Shape.new_sphere((3, 4, -2), 1.0)
    tex = S.Texture.checkerboard(S.Color(0.9, 0.9, 0.9), S.Color(0.2, 0.3, 0.8), 3.0)
    prims = [S.Primitive.new_area_light(s_light, S.Light.Area(s_light, S.Color(6, 6, 6))),
             S.Primitive.new(S.Shape.new_disk((0, 0, 0), 90, 0, 6, 0), S.Material.new_matte(tex, 0.0)),
             S.Primitive.new(S.Shape.new_sphere((-1, 1, 0), 1.0), S.Material.new_plastic(S.Color(0.8, 0.3, 0.2), S.Color(1, 1, 1), 20.0)),
             S.Primitive.new(S.Shape.new_triangle((1, 0, 1), (2.5, 0, 1), (1, 2, 1)), S.Material.new_metal(S.Color(0.2, 0.9, 1.1), S.Color(3.9, 2.4, 2.2)))]
    film = S.Film(40, 24)
    if kind == 'orthographic':
        cam = S.Camera.orthographic(film, (0, 6, -6), (0, 0.5, 0), (0, 1, 0))
    elif kind == 'orthographic_lens':
        cam = S.Camera.orthographic(film, (0, 6, -6), (0, 0.5, 0), (0, 1, 0), lens_radius=0.05, focal_distance=8.0)
    else:
        cam = S.Camera.perspective(film, (0, 3, -7), (0, 0.5, 0), (0, 1, 0), 40, lens_radius=0.08, focal_distance=7.0)
    sc = S.Scene(5, 8, cam, [S.Light.Distant((0.3, -1, 0.2), S.Color(0.5, 0.5, 0.5))], prims)
    ctx = backend.Context(0)
    host = backend.HostScene(sc, bvh_ctx=ctx)
    orc = ol.OracleScene(sc)
    for a, b in zip(host.camera_matrices(), orc.camera_matrices()):
        assert np.array_equal(a, b)
    dev = ctx.upload(host)
    g, _ = dev.render(seed=6)
    o, _ = orc.render(seed=6)
    assert np.array_equal(g, o.astype(np.float32))
    assert float(g.mean()) > 1e-3
    dev.close()
    ctx.close()


def test_leaves_of_four_primitives_of_mixed_kinds():
    """Six leaves of four co-centred primitives each — sphere, triangle, ring disk, triangle, in the reference's leaf order —
    (the SAH cannot split equal centroids): the traversal steps through a leaf one slot per iteration, across shape kinds,
    with any-hit rays leaving in the middle of a leaf.  Hits, distances and the node / primitive counters of the oracle; film
    pixel-exact through metal, glass and matte on every kind."""
    from tests.parity_util import mixed_leaf_scene
    sc = mixed_leaf_scene()
    orc = ol.OracleScene(sc)
    nodes, _ = orc.bvh()
    assert int((nodes[nodes['leaf'] != 0]['count'] == 4).sum()) == 6
    ctx = backend.Context(0)
    for resident in (False, True):
        dev = ctx.upload(backend.HostScene(sc, resident=resident))
        g, gst = dev.render(seed=1, count_traversal=True)
        o, ost = orc.render(seed=1)
        assert np.array_equal(g, o.astype(np.float32))
        for k in ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims'):
            assert gst[k] == ost[k], k
        assert np.array_equal(dev.render(seed=1)[0], g)                 # the timed kernels (mixed launches)
        rays = random_rays(orc, 6000, seed=8, scale=6.0)
        rays[::2, 6] = np.random.default_rng(2).uniform(0.5, 8.0, len(rays[::2]))
        gh, ghs = dev.trace(rays)
        oh, ohs = orc.trace(rays)
        assert np.array_equal(gh['hit'], oh['hit']) and np.array_equal(gh['prim'], oh['prim']) and np.array_equal(gh['t'], oh['t'])
        assert ghs['closest_nodes'] == ohs['closest_nodes'] and ghs['closest_prims'] == ohs['closest_prims']
        ga, gas = dev.trace(rays, any_hit=True)
        oa, oas = orc.trace(rays, any_hit=True)
        assert np.array_equal(ga['hit'], oa['hit'])
        assert gas['shadow_nodes'] == oas['shadow_nodes'] and gas['shadow_prims'] == oas['shadow_prims']
        assert int(oh['hit'].sum()) > 500
        dev.close()
    ctx.close()
