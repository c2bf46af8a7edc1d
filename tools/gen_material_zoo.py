#!/usr/bin/env python3
"""Writes tests/golden/scenes/material_zoo.cry: one sphere per material (Oren-Nayar at five sigmas, four conductors,
four dielectrics, plastics with and without a specular / diffuse lobe) on a rough floor, a spherical area light and
a grey sky.  An original test scene of this repository, not a file of the reference."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mats = []
for i, (c, s) in enumerate([((0.3, 0.8, 0.5), 0), ((0.55, 0.5, 0.9), 8), ((0.85, 0.25, 0.45), 25), ((0.6, 0.65, 0.8), 70), ((0.9, 0.9, 0.9), 140)]):
    mats.append(('chalk%d' % i, 'Matte { reflectance: Color(%g, %g, %g), sigma: %g }' % (*c, s)))
for n, eta, k in [('alu', (1.35, 0.95, 0.6), (7.4, 6.3, 5.2)), ('bronze', (0.4, 0.55, 1.0), (3.3, 2.6, 2.0)),
                  ('iron', (2.9, 2.9, 2.6), (3.0, 2.9, 2.8)), ('silver', (0.15, 0.14, 0.13), (3.9, 2.9, 2.2))]:
    mats.append((n, 'Metal { eta: Color(%g, %g, %g), k: Color(%g, %g, %g) }' % (*eta, *k)))
for n, r, t, e in [('crown', (1, 1, 1), (0.9, 0.9, 0.9), 1.52), ('ice', (0.8, 0.95, 1), (0.75, 0.85, 0.9), 1.31),
                   ('zircon', (0.6, 0.6, 0.6), (0.85, 0.8, 0.7), 1.95), ('amber', (0.3, 0.2, 0.05), (0.9, 0.6, 0.2), 1.55)]:
    mats.append((n, 'Glass { reflectance: Color(%g, %g, %g), transmittance: Color(%g, %g, %g), eta: %g }' % (*r, *t, e)))
for n, d, s, ro in [('toy0', (0.95, 0.85, 0.1), (1, 1, 1), 0), ('toy1', (0.1, 0.3, 0.9), (0.6, 0.6, 0.9), 2),
                    ('toy2', (0.85, 0.15, 0.1), (0, 0, 0), 15), ('toy3', (0.3, 0.75, 0.85), (1, 1, 1), 75), ('lacquer', (0, 0, 0), (0.9, 0.9, 0.9), 0)]:
    mats.append((n, 'Plastic { diffuse: Color(%g, %g, %g), specular: Color(%g, %g, %g), roughness: %g }' % (*d, *s, ro)))
lines = ["// Fixture written for this repository's tests by tools/gen_material_zoo.py (not a file of the reference): one sphere",
         '// per material on a rough floor, a spherical area light and a grey sky.',
         '{', '    num_samples: 128,', '    camera: Perspective {',
         '        origin: Point(-12, 13, 42), target: Point(0.5, -1, 2), up: Vector(0.03, 1, 0), fov: 14,',
         '        film: { width: 480, height: 320 }', '    },', '    lights: [ Infinite { intensity: Color(0.35, 0.38, 0.45) } ],', '    materials: {',
         '        floor: Matte { reflectance: Color(0.9, 0.88, 0.84), sigma: 40 },']
lines += ['        %s: %s,' % m for m in mats]
lines += ['    },', '    shapes: {', '        sun: Sphere { origin: Point(12, 11, 8), radius: 4 },', '        floor: Sphere { origin: Point(0, -50000, 5), radius: 50000 },']
for i, (n, _) in enumerate(mats):
    row, col = divmod(i, 5)
    lines.append('        s_%s: Sphere { origin: Point(%g, 0.6, %g), radius: 0.6 },' % (n, -4 + 2 * col, -3 + 2.2 * row))
lines += ['    },', '    primitives: [', "        Shape { shape: 'sun', emittance: Color(7, 7, 6.5) },", "        Shape { shape: 'floor', material: 'floor' },"]
lines += ["        Shape { shape: 's_%s', material: '%s' }," % (n, n) for n, _ in mats]
lines += ['    ]', '}']
open(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'material_zoo.cry'), 'w').write('\n'.join(lines) + '\n')
print(len(lines), 'lines')
