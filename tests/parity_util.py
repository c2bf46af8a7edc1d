"""Shared helpers for the GPU-vs-oracle parity tests."""
import numpy as np

from craytracer_amd import scenes


def small_scenes():
    """(name, Scene) pairs that together touch every shape, lobe, texture and light kind."""
    return [
        ('simple', scenes.simple(48, 48, 8, 4)),
        ('cornell', scenes.cornell(48, 48, 8, 8)),
        ('test', scenes.test_scene(40, 40, 8, 6, with_infinite=True, with_point=True)),
        ('dragon', scenes.dragon(64, 36, 8, 8, nu=100, nv=250)),
        ('staircase', scenes.staircase(48, 27, 8, 12, detail=0.2, texture_scale=0.1)),
    ]


def random_rays(oracle_scene, n, seed, scale=None):
    """Half camera rays, half rays between random points of the scene's bounding region."""
    rng = np.random.default_rng(seed)
    W, H = oracle_scene.width, oracle_scene.height
    rays = np.zeros((n, 7))
    for i in range(n // 2):
        rays[i] = oracle_scene.camera_ray(int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 8)))
    nodes, _ = oracle_scene.bvh()
    lo, hi = nodes[0]['bmin'], nodes[0]['bmax']
    ext = np.minimum(hi - lo, 1e3 if scale is None else scale)
    ctr = np.clip((lo + hi) / 2, -1e3, 1e3)
    for i in range(n // 2, n):
        a = ctr + (rng.uniform(-0.6, 0.6, 3)) * ext
        b = ctr + (rng.uniform(-0.6, 0.6, 3)) * ext
        d = b - a
        d /= np.sqrt(d @ d)
        rays[i, :3], rays[i, 3:6], rays[i, 6] = a, d, np.inf
    return rays


def rmse(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))
