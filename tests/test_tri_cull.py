"""Certified f32 culling of the triangle test (cray_math.h tri_cull32): an answer the f32 side certifies —
"the reference's Moller-Trumbore returns false" (every lane), "it returns true" (any-hit lanes) — must never contradict the
literal f64 test (shape.rs:216-262).  Runs the very function the kernels use, compiled for the host."""
import numpy as np

from craytracer_amd import backend


def _run(v0, e1, e2, o, d, tmax):
    L = backend.lib()
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (v0, e1, e2, o, d, tmax)]
    counts = np.zeros(4, dtype=np.uint64)
    bad = L.cray_host_tri_cull_violations(*[a.ctypes.data for a in arrs], len(arrs[0]), counts.ctypes.data)
    return bad, counts


def _dirs(rng, n):
    d = rng.normal(size=(n, 3))
    return d / np.linalg.norm(d, axis=1, keepdims=True)


def _tris(rng, n, pos_lo=-2, pos_hi=3, size_lo=-4, size_hi=1):
    v0 = rng.normal(size=(n, 3)) * 10.0 ** rng.integers(pos_lo, pos_hi, (n, 1))
    s = 10.0 ** rng.uniform(size_lo, size_hi, (n, 1))
    return v0, rng.normal(size=(n, 3)) * s, rng.normal(size=(n, 3)) * s


def _aim(rng, v0, e1, e2, bu, bv, dist):
    """rays through the point v0 + bu e1 + bv e2 of each triangle, from `dist` away"""
    n = len(v0)
    target = v0 + bu[:, None] * e1 + bv[:, None] * e2
    d = _dirs(rng, n)
    return target - d * dist[:, None], d


def test_random_rays_against_random_triangles():
    rng = np.random.default_rng(21)
    n = 1_000_000
    v0, e1, e2 = _tris(rng, n)
    bu, bv = rng.uniform(-1.5, 2.5, n), rng.uniform(-1.5, 2.5, n)
    o, d = _aim(rng, v0, e1, e2, bu, bv, 10.0 ** rng.uniform(-3, 3, n))
    tmax = 10.0 ** rng.uniform(-3, 4, n)
    tmax[rng.random(n) < 0.4] = np.inf
    bad, counts = _run(v0, e1, e2, o, d, tmax)
    assert bad == 0
    # the f32 side must decide nearly all of them (that is the point), and both kinds of certificate occur
    assert counts[0] < 0.05 * n and counts[1] > 0.3 * n and counts[2] > 0.01 * n, counts


def test_rays_through_edges_and_vertices():
    """barycentrics within a few f64 / f32 ulps of 0, 1 and of u + v = 1: the certified side must step back, never guess"""
    rng = np.random.default_rng(22)
    n = 1_000_000
    v0, e1, e2 = _tris(rng, n, -1, 2, -3, 0)
    eps = rng.choice([0.0, 1e-16, -1e-16, 1e-13, -1e-13, 1e-9, -1e-9, 6e-8, -6e-8, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4], n)
    kind = rng.integers(0, 5, n)
    bu = rng.uniform(0.0, 1.0, n)
    bv = rng.uniform(0.0, 1.0, n)
    bu = np.where(kind == 0, eps, np.where(kind == 1, 1.0 + eps, bu))
    bv = np.where(kind == 2, eps, np.where(kind == 3, 1.0 - bu + eps, np.where(kind == 4, rng.choice([0.0, 1.0], n) + eps, bv)))
    o, d = _aim(rng, v0, e1, e2, bu, bv, 10.0 ** rng.uniform(-2, 2, n))
    tmax = np.full(n, np.inf)
    bad, counts = _run(v0, e1, e2, o, d, tmax)
    assert bad == 0
    assert counts[0] > 0   # some of these really are undecidable in f32


def test_tmax_at_the_hit_distance_and_origins_on_the_triangle():
    rng = np.random.default_rng(23)
    n = 1_000_000
    v0, e1, e2 = _tris(rng, n, -1, 2, -3, 0)
    bu = rng.uniform(0.05, 0.45, n)
    bv = rng.uniform(0.05, 0.45, n)
    dist = 10.0 ** rng.uniform(-3, 2, n)
    # a third of the rays START on the triangle (secondary rays: t ~ 0 +- rounding), some a hair in front of / behind it
    dist = np.where(rng.random(n) < 0.33, rng.choice([0.0, 1e-12, -1e-12, 1e-9, -1e-9, 2e-9, 1e-8, -1e-8, 1e-6], n), dist)
    o, d = _aim(rng, v0, e1, e2, bu, bv, dist)
    rel = rng.choice([0.0, 1e-16, -1e-16, 3e-16, 1e-12, -1e-12, 6e-8, -6e-8, 2e-7, -2e-7, 1e-6, -1e-6, 1e-5, -1e-5, 0.5, -0.5], n)
    tmax = np.abs(dist) * (1.0 + rel)
    tmax[rng.random(n) < 0.2] = np.inf
    bad, counts = _run(v0, e1, e2, o, d, tmax)
    assert bad == 0


def test_grazing_degenerate_and_tiny_triangles():
    """|denom| around EPSILON = 1e-9 (the reference answers `false` there), zero-area and needle triangles, 1e-7-sized triangles
    next to coordinates of 1e3, rays in the triangle's plane"""
    rng = np.random.default_rng(24)
    n = 1_000_000
    v0, e1, e2 = _tris(rng, n, -1, 4, -8, 0)
    k = rng.integers(0, 6, n)
    e2 = np.where((k == 0)[:, None], e1 * rng.uniform(-2, 2, (n, 1)), e2)                   # zero area
    e2 = np.where((k == 1)[:, None], e1 * rng.uniform(-2, 2, (n, 1)) + rng.normal(size=(n, 3)) * 1e-12, e2)   # needles
    e1 = np.where((k == 2)[:, None], 0.0, e1)                                                # a zero edge
    bu, bv = rng.uniform(-0.5, 1.5, n), rng.uniform(-0.5, 1.5, n)
    o, d = _aim(rng, v0, e1, e2, bu, bv, 10.0 ** rng.uniform(-3, 3, n))
    # rays (almost) in the plane of the triangle
    nrm = np.cross(e1, e2)
    inplane = (k == 3) & (np.linalg.norm(nrm, axis=1) > 0)
    with np.errstate(all='ignore'):
        nn = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    d2 = d - nn * np.sum(d * nn, axis=1, keepdims=True) * (1.0 - 10.0 ** rng.uniform(-12, -1, (n, 1)))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    d = np.where(inplane[:, None], d2, d)
    tmax = 10.0 ** rng.uniform(-3, 5, n)
    tmax[rng.random(n) < 0.3] = np.inf
    bad, counts = _run(v0, e1, e2, o, d, tmax)
    assert bad == 0


def test_rays_outside_the_certified_range_and_odd_tmax():
    rng = np.random.default_rng(25)
    n = 500_000
    v0, e1, e2 = _tris(rng, n)
    bu, bv = rng.uniform(-0.5, 1.5, n), rng.uniform(-0.5, 1.5, n)
    o, d = _aim(rng, v0, e1, e2, bu, bv, 10.0 ** rng.uniform(-3, 3, n))
    d[: n // 3] *= 10.0 ** rng.uniform(-14, 0, (n // 3, 3))       # tiny components, some below the certified range
    d[rng.random(n) < 0.05, 1] = 0.0                               # exactly axis-parallel
    d[n // 3: n // 2] *= 10.0 ** rng.uniform(0, 12, (n // 2 - n // 3, 1))   # unnormalised, some above the range
    tmax = 10.0 ** rng.uniform(-3, 9, n)
    tmax[rng.random(n) < 0.1] = 0.0
    tmax[rng.random(n) < 0.05] = -1.0
    tmax[rng.random(n) < 0.05] = np.nan
    bad, counts = _run(v0, e1, e2, o, d, tmax)
    assert bad == 0
    assert counts[3] > 0


def test_millimetre_triangles_far_from_the_origin():
    """the dragon scene's proportions: coordinates of order 1, edges of order 1e-3, origins on neighbouring triangles"""
    rng = np.random.default_rng(26)
    n = 1_000_000
    v0 = rng.normal(size=(n, 3))
    e1 = rng.normal(size=(n, 3)) * 1e-3
    e2 = rng.normal(size=(n, 3)) * 1e-3
    bu, bv = rng.uniform(-3, 4, n), rng.uniform(-3, 4, n)
    o, d = _aim(rng, v0, e1, e2, bu, bv, 10.0 ** rng.uniform(-4, 0.5, n))
    tmax = np.where(rng.random(n) < 0.5, np.inf, 10.0 ** rng.uniform(-4, 1, n))
    bad, counts = _run(v0, e1, e2, o, d, tmax)
    assert bad == 0
    assert counts[0] < 0.03 * n, counts   # what stays with the exact test


def test_the_check_fails_when_the_error_constant_is_too_small(tmp_path):
    """Mutation check: the same host code compiled with K = 0.5 u instead of 47 u (the f32 side then trusts its numerators far more
    than their rounding allows) must be caught by these very inputs — certified answers that contradict the f64 test."""
    import ctypes as C
    import os
    import shutil
    import subprocess
    clang = shutil.which('clang++') or '/opt/rocm/lib/llvm/bin/clang++'
    if not os.path.exists(clang):
        import pytest
        pytest.skip('no clang++ to build the mutated check with')
    src = tmp_path / 'mut.cpp'
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'craytracer_amd', 'csrc')
    src.write_text('#include "cray_cull_check.h"\nextern "C" uint64_t check(const double* a, const double* b, const double* c, const double* d, const double* e, '
                   'const double* f, uint64_t n, uint64_t* k) { return cray::tri_cull_violations(a, b, c, d, e, f, n, k); }\n')
    so = str(tmp_path / 'mut.so')
    subprocess.check_call([clang, '-O2', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-DCRAY_CULL_K_UNITS=0.5f', '-I', csrc, '-o', so, str(src)])
    L = C.CDLL(so)
    L.check.restype = C.c_uint64
    L.check.argtypes = [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p]
    rng = np.random.default_rng(22)
    n = 200_000
    v0, e1, e2 = _tris(rng, n, -1, 2, -3, 0)
    eps = rng.choice([1e-9, -1e-9, 6e-8, -6e-8, 1e-6, -1e-6], n)
    bu = np.where(rng.random(n) < 0.5, eps, 1.0 + eps)
    bv = rng.uniform(0.0, 0.5, n)
    o, d = _aim(rng, v0, e1, e2, bu, bv, 10.0 ** rng.uniform(-2, 2, n))
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (v0, e1, e2, o, d, np.full(n, np.inf))]
    assert L.check(*[a.ctypes.data for a in arrs], n, None) > 0                      # the mutant is caught
    assert _run(*arrs)[0] == 0                                                         # the real constant is not
