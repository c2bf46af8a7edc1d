"""Certified f32 culling (cray_math.h hyb_key / hyb_status): a decision the f32 side certifies must never
contradict the literal f64 slab test (bounds.rs:46-88 as restated by child_key).  Runs the very functions the kernel uses,
compiled for the host."""
import ctypes as C

import numpy as np

from craytracer_amd import backend


def _run(lo, hi, o, d, tmax):
    L = backend.lib()
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (lo, hi, o, d, tmax)]
    counts = np.zeros(4, dtype=np.uint64)
    bad = L.cray_host_hyb_key_violations(*[a.ctypes.data for a in arrs], len(arrs[0]), counts.ctypes.data)
    return bad, counts


def _exact_key(lo, hi, o, d):
    """tmin of the slab test in f64 (numpy), to place ray.tmax right at the decision boundary"""
    with np.errstate(all='ignore'):
        t0, t1 = (lo - o) / d, (hi - o) / d
        return np.max(np.minimum(t0, t1), axis=1), np.min(np.maximum(t0, t1), axis=1)


def _boxes(rng, n, scale_lo=-2, scale_hi=5):
    c = rng.normal(size=(n, 3)) * 10.0 ** rng.integers(scale_lo, scale_hi, (n, 1))
    half = np.abs(rng.normal(size=(n, 3))) * 10.0 ** rng.integers(-3, 3, (n, 1))
    half[rng.random((n, 3)) < 0.1] = 0.0   # flat boxes (axis-aligned triangles)
    return c - half, c + half


def _dirs(rng, n):
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return d


def test_certified_decisions_agree_with_the_exact_key_on_random_boxes():
    rng = np.random.default_rng(11)
    n = 1_000_000
    lo, hi = _boxes(rng, n)
    d = _dirs(rng, n)
    o = (lo + hi) / 2 + rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-3, 4, (n, 1))
    tmax = 10.0 ** rng.uniform(-3, 4, n)
    tmax[rng.random(n) < 0.3] = np.inf
    bad, counts = _run(lo, hi, o, d, tmax)
    assert bad == 0
    # the f32 side must decide most of these (flat boxes hit head-on have tmin == tmax and stay with the exact path)
    assert counts[0] < 0.1 * n and counts[3] == 0, counts


def test_ray_tmax_at_the_key():
    """ray.tmax within a few ulps (f64 and f32) of tmin / tmax of the box: the certified side must step back, never guess"""
    rng = np.random.default_rng(12)
    n = 1_000_000
    lo, hi = _boxes(rng, n, -1, 3)
    d = _dirs(rng, n)
    o = (lo + hi) / 2 - d * 10.0 ** rng.uniform(-1, 3, (n, 1))   # mostly in front of the box, looking at it
    tmin, tmx = _exact_key(lo, hi, o, d)
    base = np.where(rng.random(n) < 0.7, tmin, tmx)
    rel = rng.choice([0.0, 1e-16, -1e-16, 3e-16, -3e-16, 1e-12, -1e-12, 6e-8, -6e-8, 2e-7, -2e-7, 1e-6, -1e-6, 1e-5, -1e-5], n)
    tmax = base * (1.0 + rel)
    bad, counts = _run(lo, hi, o, d, tmax)
    assert bad == 0
    assert counts[0] > 0   # some of these really are undecidable in f32


def test_origins_on_faces_inside_and_a_hair_away():
    rng = np.random.default_rng(13)
    n = 1_000_000
    lo, hi = _boxes(rng, n)
    d = _dirs(rng, n)
    o = (lo + hi) / 2 + rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-3, 3, (n, 1))
    pick = rng.random((n, 3))
    o = np.where(pick < 0.2, lo, np.where(pick < 0.4, hi, o))
    inside = rng.random(n) < 0.2
    o[inside] = (lo + (hi - lo) * rng.random((n, 3)))[inside]
    # a hair in front of / behind a face: distances around EPSILON = 1e-9 and around the f32 resolution of the coordinates
    off = 10.0 ** rng.uniform(-13, -4, (n, 3)) * rng.choice([-1.0, 1.0], (n, 3))
    o = np.where(rng.random((n, 3)) < 0.3, o + off, o)
    tmax = 10.0 ** rng.uniform(-10, 4, n)
    tmax[rng.random(n) < 0.3] = np.inf
    bad, counts = _run(lo, hi, o, d, tmax)
    assert bad == 0


def test_axis_dominated_far_and_degenerate_rays():
    rng = np.random.default_rng(14)
    n = 500_000
    lo, hi = _boxes(rng, n)
    d = _dirs(rng, n)
    d[: n // 2] *= 10.0 ** rng.uniform(-12, 0, (n // 2, 3))          # tiny components (some below the certified range)
    d[rng.random(n) < 0.05, 0] = 0.0                                   # exactly axis-parallel: the exact path decides
    o = (lo + hi) / 2 + rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-3, 7, (n, 1))   # origins far from the box and from 0
    tmax = 10.0 ** rng.uniform(-3, 9, n)
    tmax[rng.random(n) < 0.1] = 0.0
    tmax[rng.random(n) < 0.05] = -1.0
    tmax[rng.random(n) < 0.05] = np.nan
    bad, counts = _run(lo, hi, o, d, tmax)
    assert bad == 0
    assert counts[3] > 0   # the out-of-range rays were seen (and left to the exact path)


def test_huge_coordinates_next_to_small_geometry():
    """the dragon scene: a 1e5 ground sphere in the same tree as millimetre triangles"""
    rng = np.random.default_rng(15)
    n = 500_000
    lo, hi = _boxes(rng, n, -3, 1)
    big = rng.random(n) < 0.3
    lo[big] = -1e5 * (1 + rng.random((big.sum(), 3)))
    hi[big] = 1e5 * (1 + rng.random((big.sum(), 3)))
    d = _dirs(rng, n)
    o = rng.normal(size=(n, 3)) * 3.0
    tmin, tmx = _exact_key(lo, hi, o, d)
    tmax = np.where(rng.random(n) < 0.5, np.abs(tmin) * (1 + rng.choice([0, 1e-15, -1e-15, 1e-7, -1e-7], n)), np.inf)
    bad, counts = _run(lo, hi, o, d, tmax)
    assert bad == 0
