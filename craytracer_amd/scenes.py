"""Synthetic, seeded stand-ins for the scenes BASELINE.json names.

The reference's large meshes are absent (.MISSING_LARGE_BLOBS), so every config is
restated from its `.cry` file with procedural geometry of the same class and size.
All generators are deterministic in their `seed`.
"""
import math

import numpy as np

from . import scene as S


# ---------------------------------------------------------------------------
# config 1: scenes/simple.cry:1-51, verbatim except film/spp/depth overrides
# ---------------------------------------------------------------------------
def simple(width=256, height=256, spp=16, max_depth=4):
    cam = S.Camera.perspective(S.Film(width, height), (-7.5, 6, -2), (-2.5, -1, 12), (0, 1, 0), 35)
    ground = S.Material.new_matte(S.Color(0.8, 0.8, 0.8), 0.0)
    glass = S.Material.new_glass(S.Color(1, 1, 1), S.Color(0.6, 0.6, 0.6), 1.75)
    s_ground = S.Shape.new_disk((0, 0, 10), 90, 0, 40, 0)
    s_glass = S.Shape.new_sphere((0, 1.5, 12.5), 1.5)
    s_light = S.Shape.new_disk((5, 5, 15), 90, -40, 2, 0)
    prims = [S.Primitive.new(s_ground, ground), S.Primitive.new(s_glass, glass),
             S.Primitive.new_area_light(s_light, S.Light.Area(s_light, S.Color(10, 7, 1.2)))]
    return S.Scene(max_depth, spp, cam, [S.Light.Infinite(S.Color(0.02, 0.08, 0.6))], prims)


# ---------------------------------------------------------------------------
# config 2: scenes/cornell.cry:4-13 camera + a procedural Cornell box (the OBJ is absent)
# ---------------------------------------------------------------------------
def _quad(a, b, c, d):
    """two triangles a-b-c, a-c-d"""
    return [(a, b, c), (a, c, d)]


def _box(lo, hi, rot_y_deg, centre):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    c = [np.array([x, y, z]) for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])]
    a = math.radians(rot_y_deg)
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    c = [R @ p + np.asarray(centre, float) for p in c]
    idx = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for q in idx:
        tris += _quad(c[q[0]], c[q[1]], c[q[2]], c[q[3]])
    return tris


def _plastic(kd, ks=(0, 0, 0), ns=0.0):
    """MTL -> Material mapping of src/obj.rs:82-100 for a non-emissive, opaque, illum-2 material."""
    roughness = 180.0 * (1.0 - math.pow(math.e, -ns / 100.0))
    return S.Material.new_plastic(S.Color(*kd), S.Color(*ks), roughness)


def cornell(width=512, height=512, spp=64, max_depth=8):
    cam = S.Camera.perspective(S.Film(width, height), (0, 1, -2.8), (0, 1, 0), (0, 1, 0), 60)
    white, red, green = _plastic((0.725, 0.71, 0.68)), _plastic((0.63, 0.065, 0.05)), _plastic((0.14, 0.45, 0.091))
    P = lambda *a: np.array(a, float)
    groups = [
        (white, _quad(P(-1, 0, -1), P(-1, 0, 1), P(1, 0, 1), P(1, 0, -1))),      # floor
        (white, _quad(P(-1, 2, -1), P(1, 2, -1), P(1, 2, 1), P(-1, 2, 1))),      # ceiling
        (white, _quad(P(-1, 0, 1), P(-1, 2, 1), P(1, 2, 1), P(1, 0, 1))),        # back wall
        (red, _quad(P(-1, 0, -1), P(-1, 2, -1), P(-1, 2, 1), P(-1, 0, 1))),      # left
        (green, _quad(P(1, 0, -1), P(1, 0, 1), P(1, 2, 1), P(1, 2, -1))),        # right
        (white, _box((-0.3, 0, -0.3), (0.3, 0.6, 0.3), -18, (0.33, 0, -0.25))),  # short box
        (white, _box((-0.3, 0, -0.3), (0.3, 1.2, 0.3), 17, (-0.35, 0, 0.3))),    # tall box
    ]
    prims = []
    for mat, tris in groups:
        prims.append(S.Mesh(S.triangles_flat(np.array(tris)), material=mat))
    light = _quad(P(-0.24, 1.98, -0.22), P(0.23, 1.98, -0.22), P(0.23, 1.98, 0.16), P(-0.24, 1.98, 0.16))
    prims.append(S.Mesh(S.triangles_flat(np.array(light)), emittance=S.Color(17, 12, 4)))  # obj.rs:184-192
    return S.Scene(max_depth, spp, cam, [], prims)


# ---------------------------------------------------------------------------
# config 3/5: scenes/dragon.cry:3-39 with a procedural dragon-class mesh
# ---------------------------------------------------------------------------
def torus_knot_mesh(nu, nv, seed=0, bounds=((-105, 105), (-40, 55), (-50, 50))):
    """Closed (2,3)-torus-knot tube with seeded multi-octave displacement.
    Returns (vertices[nu*nv,3], indices[2*nu*nv,3]); both directions wrap."""
    rng = np.random.default_rng(seed)
    u = np.arange(nu) * (2 * math.pi / nu)
    v = np.arange(nv) * (2 * math.pi / nv)
    p, q = 2, 3

    def curve(t):
        r = 2.0 + np.cos(q * t)
        return np.stack([r * np.cos(p * t), np.sin(q * t), r * np.sin(p * t)], axis=-1)

    h = 1e-4
    c = curve(v)
    tan = curve(v + h) - curve(v - h)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    acc = curve(v + h) - 2 * c + curve(v - h)
    nrm = acc - tan * np.sum(acc * tan, axis=1, keepdims=True)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    bin_ = np.cross(tan, nrm)
    U, V = np.meshgrid(u, v, indexing='ij')  # [nu, nv]
    disp = np.zeros_like(U)
    amp = 0.22
    for octave in range(6):
        fu = int(rng.integers(1, 4)) * (2 ** octave)
        fv = int(rng.integers(2, 9)) * (2 ** octave)
        ph = rng.uniform(0, 2 * math.pi, 3)
        disp += amp * np.sin(fu * U + ph[0]) * np.sin(fv * V + ph[1] + 0.5 * np.sin(fu * U + ph[2]))
        amp *= 0.55
    radius = 0.55 * (1.0 + disp)
    pos = (c[None] + radius[..., None] * (np.cos(U)[..., None] * nrm[None] + np.sin(U)[..., None] * bin_[None]))
    pos = pos.reshape(-1, 3)
    for ax in range(3):
        lo, hi = pos[:, ax].min(), pos[:, ax].max()
        pos[:, ax] = (pos[:, ax] - lo) / (hi - lo) * (bounds[ax][1] - bounds[ax][0]) + bounds[ax][0]
    iu = np.arange(nu)[:, None]
    iv = np.arange(nv)[None, :]
    a = (iu * nv + iv).reshape(-1)
    b = (((iu + 1) % nu) * nv + iv).reshape(-1)
    c2 = (((iu + 1) % nu) * nv + (iv + 1) % nv).reshape(-1)
    d = (iu * nv + (iv + 1) % nv).reshape(-1)
    idx = np.concatenate([np.stack([a, b, c2], axis=1), np.stack([a, c2, d], axis=1)])
    return pos, idx


def dragon(width=1920, height=1080, spp=64, max_depth=8, nu=1200, nv=3000, seed=0):
    """nu*nv*2 triangles (default 7.2 M, the size of xyzrgb_dragon.obj)."""
    cam = S.Camera.perspective(S.Film(width, height), (150, 70, 150), (30, -50, 0), (0, 1, 0), 60)
    ground = S.Material.new_matte(S.Color(1, 1, 1), 0.0)
    metal = S.Material.new_metal(S.Color(0.18299, 0.42108, 1.37340), S.Color(3.42420, 2.34590, 1.77040))
    s_light = S.Shape.new_disk((0, 80, 0), 90, 0, 50, 0)
    s_ground = S.Shape.new_sphere((0, -100040, 10), 100000)
    verts, idx = torus_knot_mesh(nu, nv, seed)
    prims = [S.Primitive.new(s_ground, ground),
             S.Primitive.new_area_light(s_light, S.Light.Area(s_light, S.Color(1, 1, 1))),
             S.Mesh.from_indexed(verts, idx, material=metal)]
    return S.Scene(max_depth, spp, cam, [], prims)


# ---------------------------------------------------------------------------
# config 4: scenes/staircase.cry:3-26 with a procedural textured interior
# ---------------------------------------------------------------------------
def _procedural_texture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3), dtype=np.float64)
    base = rng.uniform(60, 200, 3)
    for k in range(3):
        fx, fy = rng.uniform(2, 40, 2)
        ph = rng.uniform(0, 6.28, 2)
        img[..., k] = base[k] + 40 * np.sin(fx * x / w * 6.28 + ph[0]) * np.cos(fy * y / h * 6.28 + ph[1])
    img += rng.integers(-8, 9, size=(h, w, 1))
    return np.clip(img, 0, 255).astype(np.uint8)


def _grid_patch(origin, du, dv, nu, nv, bump=0.0, seed=0):
    """Tessellated parallelogram with smooth normals + UVs: returns verts, normals, uvs, idx."""
    origin, du, dv = (np.asarray(a, float) for a in (origin, du, dv))
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing='ij')
    n = np.cross(dv, du)
    n = n / np.linalg.norm(n)
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 6.28, 4)
    hgt = bump * (np.sin(9 * s * 6.28 + ph[0]) * np.sin(7 * t * 6.28 + ph[1]) + 0.5 * np.sin(23 * s * 6.28 + ph[2]) * np.sin(19 * t * 6.28 + ph[3]))
    pos = origin + s[..., None] * du + t[..., None] * dv + hgt[..., None] * n
    # analytic-ish normals via finite differences of the height field
    gs = np.gradient(hgt, axis=0) * nu / np.linalg.norm(du)
    gt = np.gradient(hgt, axis=1) * nv / np.linalg.norm(dv)
    nn = n[None, None] - gs[..., None] * (du / np.linalg.norm(du)) - gt[..., None] * (dv / np.linalg.norm(dv))
    nn /= np.linalg.norm(nn, axis=-1, keepdims=True)
    uv = np.stack([s * 3.0, t * 3.0], axis=-1)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * (nv + 1) + j).reshape(-1); b = ((i + 1) * (nv + 1) + j).reshape(-1)
    c = ((i + 1) * (nv + 1) + j + 1).reshape(-1); d = (i * (nv + 1) + j + 1).reshape(-1)
    idx = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)])
    return pos.reshape(-1, 3), nn.reshape(-1, 3), uv.reshape(-1, 2), idx


STAIRCASE_TEXTURE_SIZES = [(512, 512), (1024, 1024), (2048, 1365), (1600, 1200), (3500, 2625), (512, 512),
                           (1024, 683), (2500, 2500), (800, 800), (2048, 2048)]


def staircase(width=1920, height=1080, spp=256, max_depth=12, detail=1.0, seed=0, texture_scale=1.0):
    """Procedural interior: stairs, walls, floor, props; >= 1 M triangles at detail=1, 24 materials
    of the kinds `objs/staircase/staircase.mtl` maps to (src/obj.rs:88-102), 10 RGB8 textures."""
    cam = S.Camera.perspective(S.Film(width, height), (0, 2, -4.92), (0, 2.5, 0), (0, 1, 0), 35,
                               lens_radius=0.001, focal_distance=3)
    texs = []
    for i, (w, h) in enumerate(STAIRCASE_TEXTURE_SIZES):
        w2, h2 = max(8, int(w * texture_scale)), max(8, int(h * texture_scale))
        texs.append(S.Texture.image(_procedural_texture(w2, h2, seed * 100 + i)))

    def rough(ns):
        return 180.0 * (1.0 - math.pow(math.e, -ns / 100.0))

    mats = []
    rng = np.random.default_rng(seed + 1)
    for i in range(24):
        kd = S.Color(*rng.uniform(0.2, 0.9, 3))
        kind = i % 6
        tex = texs[i % len(texs)] if i % 2 == 0 else kd
        if kind == 0:    # Plastic, Ns=250, no specular (e.g. "Black")
            mats.append(S.Material.new_plastic(tex, S.Color.BLACK, rough(250.0)))
        elif kind == 1:  # Plastic with specular, Ns=1000 (e.g. "Wood_*")
            mats.append(S.Material.new_plastic(tex, S.Color(0.5, 0.5, 0.5), rough(1000.0)))
        elif kind == 2:  # illum 4 -> Metal(eta = Kd, k = Ks), obj.rs:99
            mats.append(S.Material.new_metal(tex, S.Color(4.0, 3.0, 2.0)))
        elif kind == 3:  # Ns = 0 -> Lambert lobe
            mats.append(S.Material.new_plastic(tex, S.Color.BLACK, rough(0.0)))
        elif kind == 4:  # d < 1 -> Glass(Kd, Kd, Ni), obj.rs:91-94
            mats.append(S.Material.new_glass(kd, kd, 1.1))
        else:            # Plastic + specular, Lambert diffuse
            mats.append(S.Material.new_plastic(tex, S.Color(0.04, 0.04, 0.04), rough(0.0)))

    n = max(2, int(224 * detail))
    meshes = []

    def add(origin, du, dv, mat, bump=0.0, res=n):
        v, nn, uv, idx = _grid_patch(origin, du, dv, res, res, bump, seed=len(meshes) + seed)
        meshes.append(S.Mesh.from_indexed(v, idx, material=mat, normals=nn, uvs=uv))

    add((-2, 0, -5.5), (4, 0, 0), (0, 0, 9), mats[1], bump=0.004)       # floor (wood)
    add((-2, 6, -5.5), (0, 0, 9), (4, 0, 0), mats[3])                    # ceiling
    add((-2, 0, 3.5), (4, 0, 0), (0, 6, 0), mats[0], bump=0.002)         # back wall (wallpaper)
    add((-2, 0, -5.5), (0, 0, 9), (0, 6, 0), mats[5], bump=0.002)        # left wall
    add((2, 0, -5.5), (0, 6, 0), (0, 0, 9), mats[6], bump=0.002)         # right wall
    steps = 14
    for i in range(steps):
        y0, z0 = 0.2 * i, -1.0 + 0.3 * i
        add((-1.2, y0 + 0.2, z0), (2.4, 0, 0), (0, 0, 0.3), mats[7 + i % 3], bump=0.001, res=max(2, n // 3))  # tread
        add((-1.2, y0, z0), (2.4, 0, 0), (0, 0.2, 0), mats[10 + i % 3], res=max(2, n // 3))                   # riser
    for i in range(8):  # props: framed "paintings", glass panes, metal rails
        x = -1.9 + 0.5 * i
        add((x, 1.2 + 0.1 * (i % 3), 3.45), (0.4, 0, 0), (0, 0.6, 0), mats[12 + i], res=max(2, n // 4))
    add((-1.25, 0.9, -1.0), (0, 0, 4.2), (0, 0.05, 2.8), mats[2], res=max(2, n // 2))   # rail (metal)
    add((1.25, 0.9, -1.0), (0, 0.05, 2.8), (0, 0, 4.2), mats[8], res=max(2, n // 2))    # rail (metal)
    add((-0.8, 0.0, -3.0), (1.6, 0, 0), (0, 1.4, 0.1), mats[4], res=max(2, n // 2))     # glass pane
    add((-1.0, 0.01, -4.0), (2.0, 0, 0), (0, 0, 2.0), mats[20], bump=0.01, res=n)       # rug

    s_light = S.Shape.new_disk((1, 5.5, 2.5), 60, 0, 2, 0)
    prims = [S.Primitive.new_area_light(s_light, S.Light.Area(s_light, S.Color(5, 5, 5)))] + meshes
    lights = [S.Light.Point((0, 2.25, -4.5), S.Color(0.3, 0.3, 0.3))]
    return S.Scene(max_depth, spp, cam, lights, prims)


# ---------------------------------------------------------------------------
# small mixed scene that touches every shape / lobe / light / texture kind (scenes/test.cry restated)
# ---------------------------------------------------------------------------
def test_scene(width=96, height=96, spp=8, max_depth=6, with_infinite=False, with_point=False):
    cam = S.Camera.perspective(S.Film(width, height), (1.5, 1.5, -4), (1.5, 1, 0), (0, 1, 0), 60)
    white = S.Material.new_matte(S.Color(1, 1, 1), 100.0)
    red = S.Material.new_matte(S.Color(1, 0, 0), 0.0)
    green = S.Material.new_matte(S.Color(0, 1, 0), 0.0)
    blue = S.Material.new_matte(S.Texture.checkerboard(S.Color(0, 0, 1), S.Color(1, 1, 1), 4.0), 0.0)
    mirror = S.Material.new_metal(S.Color(0.9, 0.8, 0.4), S.Color(4.0, 3.0, 2.0))
    glass = S.Material.new_glass(S.Color(1, 1, 1), S.Color(0.9, 0.9, 0.9), 1.5)
    plastic = S.Material.new_plastic(S.Color(0.9, 0.1, 0.1), S.Color(1, 1, 1), 10.0)
    tex = S.Texture.image(_procedural_texture(64, 48, 5))
    textured = S.Material.new_plastic(tex, S.Color(0.2, 0.2, 0.2), 0.0)
    s_light = S.Shape.new_triangle((0, 0, 0), (0, 0, -1), (0, 1, 0))
    prims = [
        S.Primitive.new_area_light(s_light, S.Light.Area(s_light, S.Color(1, 1, 1))),
        S.Primitive.new(S.Shape.new_sphere((0, -100, 0), 100), white),
        S.Primitive.new(S.Shape.new_triangle((0, 0, 0), (1, 0, 0), (0, 1, 0)), red),
        S.Primitive.new(S.Shape.new_triangle((2, 0, 0), (2, 1, 0), (3, 0, 0)), green),
        S.Primitive.new(S.Shape.new_disk((0.5, 2, 0), 180, 0, 0.5, 0), red),
        S.Primitive.new(S.Shape.new_disk((2.5, 2, 0), 0, 0, 0.5, 0.2), textured),
        S.Primitive.new(S.Shape.new_sphere((0.5, 0.25, -1), 0.25), glass),
        S.Primitive.new(S.Shape.new_sphere((2.5, 0.25, -1), 0.25), mirror),
        S.Primitive.new(S.Shape.new_sphere((1.5, 0.25, -1), 0.25), blue),
        S.Primitive.new(S.Shape.new_sphere((1.5, 0.9, -0.5), 0.3), plastic),
    ]
    lights = [S.Light.Distant((0, 0, -1), S.Color(1, 1, 1))]
    if with_infinite:
        lights.append(S.Light.Infinite(S.Color(0.1, 0.1, 0.1)))
    if with_point:
        lights.append(S.Light.Point((0, 3, -2), S.Color(1, 1, 1)))
    return S.Scene(max_depth, spp, cam, lights, prims)
