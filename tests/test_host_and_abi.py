"""CPU-only tests of the product's host layer and of the C-ABI library (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from craytracer_amd import backend, scenes
from craytracer_amd import scene as S
from oracle import oracle_lib as ol
from tests.parity_util import small_scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in sorted(f for f in os.listdir(os.path.join(ROOT, 'include')) if f.endswith('.h')):
        text = open(os.path.join(ROOT, 'include', h)).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        names |= set(re.findall(r'\b(cray_[a-z_0-9]+)\s*\(', text))
    return names


def test_library_exports_every_declared_symbol():
    lib = backend.lib()
    decl = declared_symbols()
    assert decl == set(backend.ABI_SYMBOLS), decl ^ set(backend.ABI_SYMBOLS)
    for name in decl:
        assert getattr(lib, name) is not None


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    with pytest.raises(backend.CrayError) as e:
        backend.Context(0)
    assert 'no CPU fallback' in str(e.value) or 'HIP' in str(e.value)


@pytest.mark.parametrize('name', [n for n, _ in small_scenes()])
def test_host_scene_new_equals_oracle_bitwise(name):
    """Scene::new mirror: BVH topology, bounds, leaf order, light CDF and camera matrices are
    bit-identical to the oracle's independent restatement of bvh.rs / light.rs / camera.rs."""
    sc = dict(small_scenes())[name]
    host, orc = backend.HostScene(sc), ol.OracleScene(sc)
    hn, hr = host.bvh()
    on, orf = orc.bvh()
    assert len(hn) == len(on)
    for a, b in [('bmin', 'bmin'), ('bmax', 'bmax'), ('left', 'left'), ('right', 'right'), ('first', 'first'),
                 ('count', 'count'), ('axis', 'axis'), ('is_leaf', 'leaf')]:
        assert np.array_equal(hn[a], on[b]), a
    assert np.array_equal(hr, orf)
    assert np.array_equal(host.light_cdf(), orc.light_cdf())
    for a, b in zip(host.camera_matrices(), orc.camera_matrices()):
        assert np.array_equal(a, b)
    leaf = hn['is_leaf'] == 1
    assert hn['count'][leaf].max() <= 4      # MAX_LEAF_PRIMITIVES, bvh.rs:237
    assert sorted(hr.tolist()) == list(range(len(sc.prims)))


def test_median_split_small_is_single_leaf():
    # tests/test_bvh.rs builds with SplitMethod::Median over 2 spheres: <= 4 primitives -> one leaf
    white = S.Material.new_matte(S.Color.WHITE, 0.0)
    prims = [S.Primitive.new(S.Shape.new_sphere((0.5, 0.5, 0.5), 0.5), white),
             S.Primitive.new(S.Shape.new_sphere((1.5, 0.5, 0.5), 0.5), white)]
    cam = S.Camera.perspective(S.Film(4, 4), (0, 0, -5), (0, 0, 0), (0, 1, 0), 60)
    sc = S.Scene(8, 1, cam, [S.Light.Point((0, 5, 0), S.Color.WHITE)], prims)
    nodes, refs = backend.HostScene(sc, split_method=backend.HostScene.MEDIAN).bvh()
    assert len(nodes) == 1 and nodes[0]['is_leaf'] == 1 and nodes[0]['count'] == 2
    assert tuple(nodes[0]['bmin']) == (0, 0, 0) and tuple(nodes[0]['bmax']) == (2, 1, 1)


def test_first_equal_light_is_position_of_first_equal_value():
    tri = S.triangles_flat(np.array([[(0, 0, 0), (0, 0, -1), (0, 1, 0)]] * 3, dtype=float))
    tri2 = S.triangles_flat(np.array([[(5, 0, 0), (5, 0, -1), (5, 1, 0)]], dtype=float))
    cam = S.Camera.perspective(S.Film(4, 4), (0, 0, -5), (0, 0, 0), (0, 1, 0), 60)
    sc = S.Scene(4, 1, cam, [], [S.Mesh(tri, emittance=S.Color(1, 1, 1)), S.Mesh(tri2, emittance=S.Color(1, 1, 1)),
                                 S.Mesh(tri2, emittance=S.Color(2, 1, 1))])
    assert backend.HostScene(sc).first_equal_light().tolist() == [0, 0, 0, 3, 4]


def test_scene_without_lights_is_rejected():
    cam = S.Camera.perspective(S.Film(4, 4), (0, 0, -5), (0, 0, 0), (0, 1, 0), 60)
    white = S.Material.new_matte(S.Color.WHITE, 0.0)
    with pytest.raises(ValueError, match='No lights in the scene'):   # scene_parser.rs:1104-1109
        S.Scene(8, 1, cam, [], [S.Primitive.new(S.Shape.new_sphere((0, 0, 0), 1), white)])


def test_build_panics_become_error_codes():
    # all centroids coincide -> the reference's partition leaves one side empty and panics (bvh.rs:327)
    tri = S.triangles_flat(np.array([[(0, 0, 0), (1, 0, 0), (0, 1, 0)]] * 6, dtype=float))
    cam = S.Camera.perspective(S.Film(4, 4), (0, 0, -5), (0, 0, 0), (0, 1, 0), 60)
    white = S.Material.new_matte(S.Color.WHITE, 0.0)
    sc = S.Scene(4, 1, cam, [S.Light.Point((0, 5, 0), S.Color.WHITE)], [S.Mesh(tri, material=white)])
    with pytest.raises(backend.CrayError, match='panic'):
        backend.HostScene(sc)
    assert ol.OracleScene(sc).build_error != 0


def test_material_constructors_follow_the_reference():
    assert S.Material.new_matte(S.Color.WHITE, 0.0).bxdfs[0].kind == S.BXDF_LAMBERTIAN        # material.rs:21-22
    assert S.Material.new_matte(S.Color.WHITE, 20.0).bxdfs[0].kind == S.BXDF_OREN_NAYAR
    m = S.Material.new_plastic(S.Color(0.5, 0.5, 0.5), S.Color.BLACK, 0.0)                     # material.rs:39-63
    assert m.is_bsdf and [b.kind for b in m.bxdfs] == [S.BXDF_LAMBERTIAN]
    m = S.Material.new_plastic(S.Color(0.5, 0.5, 0.5), S.Color(1, 1, 1), 10.0)
    assert [b.kind for b in m.bxdfs] == [S.BXDF_OREN_NAYAR, S.BXDF_SPECULAR_BRDF]
    assert (m.bxdfs[1].eta_i, m.bxdfs[1].eta_t) == (1.0, 1.5)
    assert S.Material.new_plastic(S.Color.BLACK, S.Color.BLACK, 0.0).bxdfs == []
    g = S.Material.new_glass(S.Color.WHITE, S.Color.WHITE, 1.75)
    assert not g.is_bsdf and g.bxdfs[0].kind == S.BXDF_FRESNEL_SPECULAR and g.bxdfs[0].eta_t == 1.75
    assert S.Material.new_metal(S.Color.WHITE, S.Color.WHITE).is_bsdf
    assert S.Shape.new_triangle((0, 0, 0), (1, 0, 0), (2, 0, 0)) is None                        # degenerate, shape.rs:78-81
    t = S.Shape.new_triangle((1, 0, 0), (1, 1, 0), (2, 0, 0)).tris[0]
    assert tuple(t['n0']) == (0, 0, 1) and tuple(t['uv02']) == (1, 1)


def test_sincos_cr_equals_binary128_rounding():
    """The product's sampling sin/cos (double-double) against the oracle's (libquadmath): both
    must be THE correctly rounded value, on the argument distributions of sample_disk/sphere."""
    lib, L = backend.lib(), ol.lib()
    rng = np.random.default_rng(0)
    n = 60000
    u = 2.0 * rng.integers(1, 1 << 23, n).astype(np.float32).astype(np.float64) / (1 << 23) - 1.0
    v = 2.0 * rng.integers(1, 1 << 23, n).astype(np.float32).astype(np.float64) / (1 << 23) - 1.0
    ok = (u != 0) & (v != 0)
    u, v = u[ok], v[ok]
    big = np.abs(u) > np.abs(v)
    theta = np.where(big, 0.7853981633974483 * v / u, 1.5707963267948966 - 0.7853981633974483 * u / v)
    phi = 2.0 * np.pi * rng.integers(0, 1 << 23, n).astype(np.float64) / (1 << 23)
    xs = np.concatenate([theta, phi, rng.uniform(-7, 7, n)])
    s, c = C.c_double(), C.c_double()
    bad = 0
    for x in xs:
        lib.cray_host_sincos(float(x), C.byref(s), C.byref(c))
        bad += (s.value != L.orc_sample_sin(float(x))) + (c.value != L.orc_sample_cos(float(x)))
    assert bad == 0


def test_sincos_short_evaluation_never_differs_from_the_double_double_one():
    """sincos_cr first tries a short evaluation (leading terms exact, the rest in f64) with a rounding test of radius 2^-64
    relative (Ziv's strategy) and falls back to the double-double evaluation when the test is inconclusive.  On 2 x 10^7
    arguments of the sampling routines' distributions, near multiples of 1/64 and pi/2 and tiny: not one differing bit, the
    candidate never further than a third of the radius from the double-double value, fallbacks well under 1 %."""
    L = backend.lib()
    rng = np.random.default_rng(3)
    n = 4_000_000
    u, v = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    with np.errstate(all='ignore'):
        theta = np.where(np.abs(u) > np.abs(v), 0.7853981633974483 * v / u, 1.5707963267948966 - 0.7853981633974483 * u / v)
    sets = [theta, 2.0 * np.pi * rng.random(n), rng.uniform(-0.8, 6.4, n),
            rng.integers(-100, 420, n) / 64.0 + rng.uniform(-1, 1, n) * 2.0 ** rng.uniform(-60, -6, n),
            rng.integers(-8, 9, n) * (np.pi / 2) + rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-18, -2, n)]
    for xs in sets:
        xs = np.ascontiguousarray(xs[np.isfinite(xs)], dtype=np.float64)
        st = np.zeros(3)
        bad = L.cray_host_sincos_fast_check(xs.ctypes.data, len(xs), st.ctypes.data)
        assert bad == 0
        assert st[1] < 0.34, st        # observed: 0.26 of the radius on 10^9 arguments
        assert st[0] < 0.01 * len(xs), st


def test_div_fast_is_the_correctly_rounded_quotient():
    """cray_math.h div_fast (two FMA corrections on a*RN(1/d)) must equal a/d bit for bit: it replaces
    the reference's divisions in the slab test.  Random, adversarial (quotients next to products,
    divisor mantissas with long runs of ones) and scene-like operands."""
    L = backend.lib()
    rng = np.random.default_rng(0)
    n = 2_000_000
    cases = []
    cases.append((rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n), rng.standard_normal(n) * 10.0 ** rng.integers(-3, 3, n)))
    q, d = rng.uniform(1, 2, n), rng.uniform(1, 2, n) * rng.choice([-1, 1], n)
    cases.append((np.nextafter(q * d, rng.choice([-np.inf, np.inf], n)), d))
    m = rng.integers(0, 2 ** 52, n, dtype=np.uint64) | rng.choice(
        np.array([0, (1 << 52) - 1, (1 << 26) - 1, ((1 << 26) - 1) << 26], dtype=np.uint64), n)
    cases.append((rng.uniform(-1e3, 1e3, n), (m | np.uint64(0x3ff0000000000000)).view(np.float64)))
    cases.append((rng.uniform(-200, 200, n) - rng.uniform(-200, 200, n), np.cos(rng.uniform(0, np.pi, n))))
    cases.append((np.zeros(n), rng.standard_normal(n)))
    for a, d in cases:
        a, d = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(d, dtype=np.float64)
        assert L.cray_host_div_fast_mismatches(a.ctypes.data, d.ctypes.data, n) == 0


def test_fast_slab_test_equals_the_literal_one():
    """child_key_fast (min/max of exact quotients, contains() from the signs, three comparisons) must give the
    key of the literal Bounds::intersects / contains restatement bit for bit: random boxes and rays, origins
    exactly on faces / edges / corners and inside, flat (zero-width) boxes, tiny and axis-dominated directions,
    boxes behind the origin, huge coordinates (the dragon scene's 1e5 ground sphere)."""
    L = backend.lib()
    rng = np.random.default_rng(1)
    n = 1_000_000

    def run(lo, hi, o, d):
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (lo, hi, o, d)]
        checked = C.c_uint64(0)
        bad = L.cray_host_child_key_mismatches(*[a.ctypes.data for a in arrs], len(arrs[0]), C.byref(checked))
        return bad, checked.value

    c = rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-2, 5, (n, 1))
    half = np.abs(rng.normal(size=(n, 3))) * 10.0 ** rng.integers(-3, 3, (n, 1))
    half[rng.random((n, 3)) < 0.05] = 0.0                       # flat boxes
    lo, hi = c - half, c + half
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[rng.random(n) < 0.1] *= np.array([1.0, 1e-12, 1e-150])    # nearly axis-parallel
    o = c + rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-3, 4, (n, 1))
    # origins exactly on a bound of some axes, or exactly inside
    pick = rng.random((n, 3))
    o = np.where(pick < 0.15, lo, np.where(pick < 0.3, hi, o))
    inside = rng.random(n) < 0.1
    o[inside] = (lo + (hi - lo) * rng.random((n, 3)))[inside]
    bad, checked = run(lo, hi, o, d)
    assert checked > 0.9 * n and bad == 0
    # hits at distances around EPSILON = 1e-9: origin a hair in front of / behind a face
    eps_off = 10.0 ** rng.uniform(-12, -6, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
    o2 = np.where(pick < 0.5, lo + eps_off, o)
    bad, checked = run(lo, hi, o2, d)
    assert checked > 0.9 * n and bad == 0
    # inputs outside the guard (zero direction components) are skipped by the kernel too (plain division path)
    d0 = d.copy(); d0[:, 0] = 0.0
    assert run(lo, hi, o, d0)[1] == 0
