#!/usr/bin/env python3
"""Writes the OBJ + MTL + PPM fixture of tests/golden/meshes/ (original data, written for this repository) and the scene
file tests/golden/scenes/mesh_room.cry that loads it through `Mesh { file_name: .. }` (scene_parser.rs:1042-1076, obj.rs).

What the fixture exercises of src/obj.rs: smooth per-vertex normals (vn) and UVs (vt) with the RH -> LH flips (:131, :140,
:149); faces without vn / vt (flat normal :159, default uvs :166-168); quads (fan triangulation); negative indices;
a `map_Kd` texture (PPM, decoded by cray_load_image); `Ke` emissive faces -> one area light per triangle (:184-192);
`d < 1` -> Glass(Kd, Kd, Ni) (:91-94); `illum 4` -> Metal(eta = Kd, k = Ks) (:98-99); Plastic with and without Ks and
Ns 0 / 250 / 1000 -> roughness 180 (1 - e^(-Ns/100)) (:84, :100); `usemtl` of an unknown name -> fallback material;
several `o` groups."""
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden', 'meshes')
os.makedirs(OUT, exist_ok=True)

lines = ['# fixture written by tools/gen_mesh_fixture.py', 'mtllib room.mtl']
nv = nt = nn = 0


def fmt(x):
    return ('%.6f' % x).rstrip('0').rstrip('.') if abs(x) > 5e-7 else '0'


def grid(name, mtl, origin, du, dv, n, uv_scale, bump):
    """n x n quads, per-vertex normals from the bump height field, uvs"""
    global nv, nt, nn
    lines.append('o ' + name)
    o, du, dv = (np.array(a, float) for a in (origin, du, dv))
    nrm = np.cross(du, dv)
    nrm /= np.linalg.norm(nrm)
    for i in range(n + 1):
        for j in range(n + 1):
            s, t = i / n, j / n
            hgt = bump * math.sin(5 * s) * math.cos(4 * t)
            p = o + s * du + t * dv + hgt * nrm
            lines.append('v %s %s %s' % tuple(fmt(x) for x in p))
            g = nrm - bump * 5 * math.cos(5 * s) * math.cos(4 * t) * du / np.dot(du, du) + bump * 4 * math.sin(5 * s) * math.sin(4 * t) * dv / np.dot(dv, dv)
            g /= np.linalg.norm(g)
            lines.append('vn %s %s %s' % tuple(fmt(x) for x in g))
            lines.append('vt %s %s' % (fmt(s * uv_scale), fmt(t * uv_scale)))
    lines.append('usemtl ' + mtl)
    for i in range(n):
        for j in range(n):
            a = nv + i * (n + 1) + j + 1
            b, c, d = a + (n + 1), a + (n + 1) + 1, a + 1
            lines.append('f %d/%d/%d %d/%d/%d %d/%d/%d %d/%d/%d' % (a, a - nv + nt, a - nv + nn, b, b - nv + nt, b - nv + nn, c, c - nv + nt, c - nv + nn, d, d - nv + nt, d - nv + nn))
    k = (n + 1) ** 2
    nv += k; nt += k; nn += k


def ball(name, mtl, centre, radius, rings, segs):
    """UV sphere with smooth normals, no uvs: f v//vn"""
    global nv, nn
    lines.append('o ' + name)
    idx = {}
    for r in range(rings + 1):
        th = math.pi * r / rings
        for s in range(segs):
            ph = 2 * math.pi * s / segs
            d = np.array([math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)])
            p = np.array(centre, float) + radius * d
            lines.append('v %s %s %s' % tuple(fmt(x) for x in p))
            lines.append('vn %s %s %s' % tuple(fmt(x) for x in d))
            idx[(r, s)] = len(idx) + 1
    lines.append('usemtl ' + mtl)
    for r in range(rings):
        for s in range(segs):
            a, b = idx[(r, s)], idx[(r, (s + 1) % segs)]
            c, d = idx[(r + 1, (s + 1) % segs)], idx[(r + 1, s)]
            if r == 0:
                lines.append('f %d//%d %d//%d %d//%d' % (a + nv, a + nn, c + nv, c + nn, d + nv, d + nn))
            elif r == rings - 1:
                lines.append('f %d//%d %d//%d %d//%d' % (a + nv, a + nn, b + nv, b + nn, d + nv, d + nn))
            else:
                lines.append('f %d//%d %d//%d %d//%d %d//%d' % (a + nv, a + nn, b + nv, b + nn, c + nv, c + nn, d + nv, d + nn))
    nv += len(idx); nn += len(idx)


def box(name, mtl, lo, hi, relative=False):
    """six quads, positions only (flat normals, default uvs); `relative`: negative indices"""
    global nv
    lines.append('o ' + name)
    c = [(x, y, z) for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])]
    for p in c:
        lines.append('v %s %s %s' % tuple(fmt(x) for x in p))
    lines.append('usemtl ' + mtl)
    for q in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]:
        if relative:
            lines.append('f ' + ' '.join(str(i - 8) for i in q))
        else:
            lines.append('f ' + ' '.join(str(nv + i + 1) for i in q))
    nv += 8


grid('floor', 'tiles', (-2, 0, -2), (4, 0, 0), (0, 0, 4), 10, 2.5, 0.02)
grid('back', 'wallpaper', (-2, 0, -2), (0, 2.5, 0), (4, 0, 0), 6, 1.0, 0.0)
ball('ball', 'brass', (-0.8, 0.5, 0.2), 0.5, 8, 12)
box('crate', 'varnish', (0.3, 0.0, -0.9), (1.1, 0.8, -0.1))
box('pane', 'glass', (-0.2, 0.0, 0.9), (1.4, 1.2, 0.95), relative=True)
box('lamp', 'lamp', (-0.5, 2.3, -0.5), (0.5, 2.35, 0.5))
box('plinth', 'no_such_material', (-1.6, 0.0, -1.6), (-1.1, 0.3, -1.1))
open(os.path.join(OUT, 'room.obj'), 'w').write('\n'.join(lines) + '\n')

open(os.path.join(OUT, 'room.mtl'), 'w').write('''# fixture written by tools/gen_mesh_fixture.py
newmtl tiles
Ns 0
Kd 0.8 0.8 0.8
Ks 0 0 0
map_Kd tiles.ppm
illum 2

newmtl wallpaper
Ns 250
Kd 0.55 0.6 0.7
Ks 0.04 0.04 0.04
illum 2

newmtl brass
Ns 250
Kd 0.9 0.8 0.5
Ks 3.0 2.0 1.0
illum 4

newmtl varnish
Ns 1000
Kd 0.5 0.25 0.1
Ks 0.5 0.5 0.5
illum 2

newmtl glass
Kd 0.9 0.95 1.0
Ni 1.1
d 0.1
illum 2

newmtl lamp
Kd 0 0 0
Ke 9 8 6
illum 2
''')

rng = np.random.default_rng(12)
h, w = 24, 32
y, x = np.mgrid[0:h, 0:w]
img = np.zeros((h, w, 3), np.uint8)
img[..., 0] = np.where(((x // 4) + (y // 4)) % 2 == 0, 210, 60)
img[..., 1] = (x * 7 + 20) % 256
img[..., 2] = (y * 9 + 40) % 256
img = np.clip(img.astype(int) + rng.integers(-6, 7, size=img.shape), 0, 255).astype(np.uint8)
with open(os.path.join(OUT, 'tiles.ppm'), 'wb') as f:
    f.write(b'P6\n# tiles\n%d %d\n255\n' % (w, h))
    f.write(img.tobytes())

open(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'mesh_room.cry'), 'w').write('''// written by tools/gen_mesh_fixture.py: an OBJ/MTL mesh with a PPM texture, emissive faces, glass and metal
{
    max_depth: 7,
    num_samples: 8,
    camera: Perspective {
        origin: Point(0.4, 1.4, -4.6),
        target: Point(0, 0.7, 0),
        up: Vector(0, 1, 0),
        fov: 48,
        film: { width: 96, height: 64 },
    },
    lights: [
        Point { origin: Point(1.5, 2, -2.5), intensity: Color(0.6, 0.6, 0.7) },
    ],
    materials: {
        fallback: Matte { reflectance: Color(0.7, 0.3, 0.3), sigma: 20 },
    },
    shapes: {},
    primitives: [
        Mesh { file_name: 'meshes/room.obj', fallback_material: 'fallback' },
    ]
}
''')
print('wrote', OUT)
