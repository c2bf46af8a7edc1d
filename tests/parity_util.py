"""Shared helpers for the GPU-vs-oracle parity tests."""
import numpy as np

from craytracer_amd import scenes


def mixed_leaf_scene(width=64, height=48, spp=8, max_depth=6):
    """Six leaves of four co-centred primitives each — sphere, triangle, ring disk, triangle (the SAH cannot split equal
    centroids) — over a ground sphere, under a disk emitter and a point light; metal, glass and two mattes rotate over the kinds."""
    from craytracer_amd import scene as S

    def cluster(c, mats):
        cx, cy, cz = c
        return [S.Primitive.new(S.Shape.new_sphere(c, 0.25), mats[0]),
                S.Primitive.new(S.Shape.new_triangle((cx - 0.375, cy - 0.375, cz + 0.125), (cx + 0.375, cy - 0.375, cz - 0.125), (cx, cy + 0.375, cz)), mats[1]),
                S.Primitive.new(S.Shape.new_disk(c, 0, 0, 0.375, 0.125), mats[2]),
                S.Primitive.new(S.Shape.new_triangle((cx - 0.25, cy + 0.25, cz - 0.0625), (cx + 0.25, cy + 0.25, cz + 0.0625), (cx, cy - 0.25, cz)), mats[3])]

    white = S.Material.new_matte(S.Color(1, 1, 1), 0.0)
    red = S.Material.new_matte(S.Color(1, 0.2, 0.2), 20.0)
    metal = S.Material.new_metal(S.Color(0.9, 0.8, 0.4), S.Color(4, 3, 2))
    glass = S.Material.new_glass(S.Color(1, 1, 1), S.Color(0.9, 0.9, 0.9), 1.5)
    prims = []
    for i, c in enumerate([(0, 0, 0), (1.5, 0, 0), (0, 1.5, 0), (1.5, 1.5, 0), (0.75, 0.75, 1.0), (3, 0, 0.5)]):
        m = [white, red, metal, glass]
        prims += cluster(tuple(float(v) for v in c), m[i % 4:] + m[:i % 4])
    prims.append(S.Primitive.new(S.Shape.new_sphere((0.75, -100.5, 0), 100.0), white))
    sl = S.Shape.new_disk((0.75, 3.5, -1.0), 90, 0, 0.75, 0)
    prims.append(S.Primitive.new_area_light(sl, S.Light.Area(sl, S.Color(8, 8, 8))))
    cam = S.Camera.perspective(S.Film(width, height), (0.75, 1.0, -5.0), (0.75, 0.75, 0), (0, 1, 0), 50)
    return S.Scene(max_depth, spp, cam, [S.Light.Point((0.75, 3, -3), S.Color(4, 4, 4))], prims)


def small_scenes():
    """(name, Scene) pairs that together touch every shape, lobe, texture and light kind."""
    return [
        ('leaves4', mixed_leaf_scene()),
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
