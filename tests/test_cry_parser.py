"""The reference's tests/test_parser.rs re-expressed against the C++ `.cry` reader
(include/cray_cry.h), plus OBJ/MTL ingest checks.  tests/golden/triangle.obj is the five-line data file the reference's
parser test reads (test_parser.rs:587-640); the scene files under tests/golden/scenes/ were written for this repository."""
import os

import numpy as np
import pytest

from craytracer_amd import backend, cry, scenes
from craytracer_amd import scene as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def values(text):
    return [(k, v) for k, v, _ in cry.tokenize(text)]


def tokenize_error(text):
    with pytest.raises(cry.ParserError) as e:
        cry.tokenize(text)
    return e.value.message, e.value.location


# ---- mod tokenizer (test_parser.rs:32-397) ---------------------------------------------------
def test_tokenizer_simple():
    assert values('') == [('Eof', None)]
    assert values(' \r\t\n') == [('Eof', None)]
    assert values('// foo') == [('Eof', None)]
    assert values('{}') == [('LeftBrace', None), ('RightBrace', None), ('Eof', None)]
    assert values('[]') == [('LeftBracket', None), ('RightBracket', None), ('Eof', None)]
    assert values('()') == [('LeftParen', None), ('RightParen', None), ('Eof', None)]
    assert values('1') == [('Number', 1.0), ('Eof', None)]
    assert values("'hello'") == [('String', 'hello'), ('Eof', None)]


def test_tokenizer_comments():
    assert values("// foo = 'hello'") == [('Eof', None)]
    assert values('//\n') == [('Eof', None)]
    assert values('//\n1') == [('Number', 1.0), ('Eof', None)]
    assert values('1 // one') == [('Number', 1.0), ('Eof', None)]
    assert tokenize_error('/') == ("Expected a second '/' to start a comment", (1, 2))
    assert tokenize_error('/ /') == ("Expected a second '/' to start a comment", (1, 2))


def test_tokenizer_numbers():
    for text, v in [('1', 1.0), ('1.0', 1.0), ('1.0000', 1.0), ('001.0000', 1.0), ('-1', -1.0), ('+1', 1.0), ('-1.1', -1.1), ('+1.1', 1.1)]:
        assert values(text) == [('Number', v), ('Eof', None)]
    assert values('2+3') == [('Number', 2.0), ('Number', 3.0), ('Eof', None)]
    assert values('4-5') == [('Number', 4.0), ('Number', -5.0), ('Eof', None)]
    assert tokenize_error('9.8.7') == ("Unexpected character: '.'", (1, 4))


def test_tokenizer_strings():
    for s in ['""', "''", "'a'", '"Once upon a midnight dreary"', "'\"'", '"\'"']:
        assert values(s) == [('String', s[1:-1]), ('Eof', None)]


def test_tokenizer_identifiers():
    for s in ['simple', 'snake_case', 'camelCase', 'SCREAMING_CASE', 'agent007']:
        assert values(s) == [('Identifier', s), ('Eof', None)]


def test_tokenizer_locations():
    assert cry.tokenize('{}') == [('LeftBrace', None, (1, 1)), ('RightBrace', None, (1, 2)), ('Eof', None, (1, 3))]
    text = "\n{\n    x: 1,\n    y: ['foo', 3.14],\n}"
    assert cry.tokenize(text) == [
        ('LeftBrace', None, (2, 1)), ('Identifier', 'x', (3, 5)), ('Colon', None, (3, 6)), ('Number', 1.0, (3, 8)),
        ('Comma', None, (3, 9)), ('Identifier', 'y', (4, 5)), ('Colon', None, (4, 6)), ('LeftBracket', None, (4, 8)),
        ('String', 'foo', (4, 9)), ('Comma', None, (4, 14)), ('Number', 3.14, (4, 16)), ('RightBracket', None, (4, 20)),
        ('Comma', None, (4, 21)), ('RightBrace', None, (5, 1)), ('Eof', None, (5, 2))]


def test_tokenizer_full():
    text = """
{
    camera: ProjectionCamera {
        origin: Point(0, 8, -10),
        up: Vector(0, 1, 0),
        fov: 5,
    },
    materials: {
        sky: Emissive {
            emittance: Color(0, 10, 60)
        }
    },
    shapes: {
        sky: Sphere {
            center: Point(0, 0, 0),
            radius: 1000
        }
    },
    primitives: [
        Shape {
            shape: 'sky',
            material: 'sky'
        }
    ]
}
"""
    I, N, St = (lambda s: ('Identifier', s)), (lambda v: ('Number', float(v))), (lambda s: ('String', s))
    LB, RB, LK, RK, LP, RP, CM, CL = (('LeftBrace', None), ('RightBrace', None), ('LeftBracket', None), ('RightBracket', None),
                                      ('LeftParen', None), ('RightParen', None), ('Comma', None), ('Colon', None))
    expected = [LB, I('camera'), CL, I('ProjectionCamera'), LB, I('origin'), CL, I('Point'), LP, N(0), CM, N(8), CM, N(-10), RP, CM,
                I('up'), CL, I('Vector'), LP, N(0), CM, N(1), CM, N(0), RP, CM, I('fov'), CL, N(5), CM, RB, CM,
                I('materials'), CL, LB, I('sky'), CL, I('Emissive'), LB, I('emittance'), CL, I('Color'), LP, N(0), CM, N(10), CM, N(60), RP, RB, RB, CM,
                I('shapes'), CL, LB, I('sky'), CL, I('Sphere'), LB, I('center'), CL, I('Point'), LP, N(0), CM, N(0), CM, N(0), RP, CM,
                I('radius'), CL, N(1000), RB, RB, CM, I('primitives'), CL, LK, I('Shape'), LB, I('shape'), CL, St('sky'), CM,
                I('material'), CL, St('sky'), RB, RK, RB, ('Eof', None)]
    assert values(text) == expected


# ---- mod parser (test_parser.rs:399-640) -------------------------------------------------------
def parse_error(text):
    with pytest.raises(cry.ParserError) as e:
        cry.parse_value(text)
    return e.value.message, e.value.location


def test_raw_value():
    assert cry.parse_value('1.23') == 'Number(1.23)'
    assert cry.parse_value("'hello'") == 'String("hello")'
    assert cry.parse_value('Vector(1, -2, 3.1)') == 'Vector(1,-2,3.1)'
    assert cry.parse_value('Color(0, 0.5, 1)') == 'Color(0,0.5,1)'
    assert cry.parse_value('{}') == 'Map@1:1{}'
    assert cry.parse_value("{ x: 1, y: 'z' }") == 'Map@1:1{x:Number(1),y:String("z")}'
    assert cry.parse_value('Sphere { center: Point(0, 0, 0), radius: 1000 }') == 'Typed:Sphere@1:8{center:Point(0,0,0),radius:Number(1000)}'
    assert cry.parse_value('[]') == 'Array[]'
    assert cry.parse_value("[1, 'foo', {}]") == 'Array[Number(1),String("foo"),Map@1:12{}]'
    assert parse_error('x') == ("Expected '(' or '{', got EOF", (1, 2))
    assert parse_error(',') == ("Expected a raw value. Got ','", (1, 1))


def test_raw_value_map():
    assert cry.parse_value('{}') == 'Map@1:1{}'
    assert cry.parse_value("{ hello: 'world' }") == 'Map@1:1{hello:String("world")}'
    assert cry.parse_value("{ hello: 'world', }") == 'Map@1:1{hello:String("world")}'   # trailing comma
    assert cry.parse_value("{ x: 1, y: 'z', v: Vector(1,2,3), c: Color(1,0,0) }") == \
        'Map@1:1{c:Color(1,0,0),v:Vector(1,2,3),x:Number(1),y:String("z")}'
    assert parse_error('{ x: 1, x: 2 }') == ('Duplicate key x', (1, 1))
    assert parse_error('{ x: 1') == ("Expected '}', got EOF", (1, 7))
    assert parse_error('{ x 1 }') == ("Expected ':', got '1'", (1, 5))
    assert parse_error('{ 1: x }') == ("Expected '}', got '1'", (1, 3))
    assert parse_error('{ x: 1 y: 2 }') == ("Expected '}', got 'y'", (1, 8))


def test_raw_value_array():
    assert cry.parse_value('[]') == 'Array[]'
    assert cry.parse_value('[1, 2, 3]') == 'Array[Number(1),Number(2),Number(3)]'
    assert cry.parse_value('[1, 2, 3,]') == 'Array[Number(1),Number(2),Number(3)]'
    assert cry.parse_value("[1, 'foo', {}]") == 'Array[Number(1),String("foo"),Map@1:12{}]'
    assert parse_error('[') == ('Expected a raw value. Got EOF', (1, 2))
    assert parse_error('[,]') == ("Expected a raw value. Got ','", (1, 2))
    assert parse_error('[1 2]') == ("Expected ']', got '2'", (1, 4))


SCENE_TEXT = """
{
    // Comment
    max_depth: 3,
    num_samples: 1,
    camera: Perspective {
        origin: Point(0, 0, 0),
        target: Point(0, 0, 1),
        up: Vector(0, 1, 0),
        fov: 60,
        lens_radius: 1,
        focal_distance: 100,
        film: {
            width: 400,
            height: 300
        },
    },
    lights: [
        Point {
            origin: Point(0, 0, 0),
            intensity: Color(1, 1, 1)
        }
    ],
    materials: {
        matte: Matte {
            reflectance: Color(1, 1, 1),
            sigma: 0
        },
        // Textures
        checks: Matte {
            reflectance: Checkerboard { a: Color(1, 1, 1), b: Color(0, 0, 0), scale: 2.5 },
            sigma: Checkerboard { a: 0, b: 1 }
        },
    },
    shapes: {
        ball: Sphere {
            origin: Point(0, 0, 2),
            radius: 1
        }
    },
    primitives: [
       Shape { shape: 'ball', material: 'matte' },
       Mesh { file_name: 'triangle.obj', fallback_material: 'checks' },
    ]
}
"""


def test_parse_scene_smoke():
    """test_parser.rs:587-640 (`objs/triangle.obj` -> the same file under tests/golden)."""
    sc = cry.parse_scene(SCENE_TEXT, base_dir=GOLDEN)
    d = sc.desc()
    assert (d.max_depth, d.num_samples) == (3, 1)
    assert sc.film_bounds() == (400, 300)
    assert (d.camera.lens_radius, d.camera.focal_distance, d.camera.fov) == (1.0, 100.0, 60.0)
    assert (d.n_prims, d.n_spheres, d.n_triangles, d.n_lights) == (2, 1, 1, 1)
    tri = np.ctypeslib.as_array(backend.C.cast(d.triangles, backend.C.POINTER(backend.C.c_double)), shape=(24,))
    # v (1,0,0) (0,1,0) (0,0,1) with z flipped (obj.rs:131): v0 = (1,0,0), e1 = (-1,1,0), e2 = (-1,0,-1)
    assert tri[:9].tolist() == [1, 0, 0, -1, 1, 0, -1, 0, -1]
    assert sc.warnings == 0
    host = backend.HostScene(sc)          # Scene::new accepts what the reader produced
    assert host.flat.n_nodes >= 1


# ---- scene-level errors and defaults (scene_parser.rs:796-798, 1025-1109) ------------------------
def scene_error(text, **kw):
    with pytest.raises(cry.ParserError) as e:
        cry.parse_scene(text, base_dir=GOLDEN, **kw)
    return e.value.message, e.value.location


MINI = """{ camera: Perspective { origin: Point(0,0,-5), target: Point(0,0,0), up: Vector(0,1,0), fov: 40, film: { width: 8, height: 6 } },
  lights: [%s], materials: { m: Matte { reflectance: Color(1,1,1), sigma: 0 } },
  shapes: { s: Sphere { origin: Point(0,0,0), radius: 1 } }, primitives: [%s] }"""


def test_scene_defaults_and_errors():
    sc = cry.parse_scene(MINI % ("Infinite { intensity: Color(1,1,1) }", "Shape { shape: 's', material: 'm' }"))
    d = sc.desc()
    assert (d.max_depth, d.num_samples, d.camera.focal_distance, d.camera.lens_radius) == (8, 4, 1e6, 0.0)
    assert scene_error(MINI % ("", "Shape { shape: 's', material: 'm' }")) == ('No lights in the scene.', (0, 0))
    msg, loc = scene_error(MINI % ("Infinite { intensity: Color(1,1,1) }", "Shape { shape: 'nope', material: 'm' }"))
    assert msg == "Cannot find shape named 'nope'" and loc is not None
    msg, _ = scene_error(MINI % ("Infinite { intensity: Color(1,1,1) }", "Shape { shape: 's', material: 'nope' }"))
    assert msg == "Cannot find material named 'nope'"
    msg, _ = scene_error(MINI % ("Infinite { intensity: Color(1,1,1) }", "Blob { shape: 's' }"))
    assert msg == 'Unknown primitive type: Blob'
    msg, _ = scene_error(MINI % ("Spot { intensity: Color(1,1,1) }", "Shape { shape: 's', material: 'm' }"))
    assert msg == "Error converting map value for 'lights' to expected type: Unknown light type: Spot"
    msg, _ = scene_error('{ lights: [] }')
    assert msg == 'camera not found in map'
    # an emissive shape is an area light: no explicit lights needed; unused keys only warn (:571-586)
    sc = cry.parse_scene(MINI % ("", "Shape { shape: 's', emittance: Color(1,2,3), bogus: 1 }"))
    assert sc.desc().n_lights == 1 and sc.warnings == 1
    # the CLI overrides of this build
    sc = cry.parse_scene(MINI % ("Infinite { intensity: Color(1,1,1) }", "Shape { shape: 's', material: 'm' }"), width=32, height=16, num_samples=9, max_depth=2)
    assert sc.film_bounds() == (32, 16) and (sc.num_samples, sc.max_depth) == (9, 2)


def desc_arrays(sc):
    d = sc.desc()
    def arr(ptr, n, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(backend.C.cast(ptr, backend.C.POINTER(backend.C.c_uint8)), shape=(n * dt.itemsize,)).view(dt).copy()
    return {'spheres': arr(d.spheres, d.n_spheres, S.SPHERE_DT), 'disks': arr(d.disks, d.n_disks, S.DISK_DT),
            'triangles': arr(d.triangles, d.n_triangles, S.TRIANGLE_DT), 'prims': arr(d.prims, d.n_prims, S.PRIM_DT),
            'lights': arr(d.lights, d.n_lights, S.LIGHT_DT)}


def test_scene_file_equals_programmatic_scene():
    """tests/golden/scenes/glass_lamp.cry through the reader == the same scene assembled with the reference's constructor
    names (Scene::new arguments bit for bit, then the same BVH, light CDF and camera matrices)."""
    parsed = cry.load_scene_file(os.path.join(GOLDEN, 'scenes', 'glass_lamp.cry'))
    floor = S.Material.new_matte(S.Color(0.7, 0.72, 0.68), 0.0)
    ball = S.Material.new_glass(S.Color(0.95, 1, 0.95), S.Color(0.7, 0.55, 0.6), 1.6)
    bearing = S.Material.new_metal(S.Color(0.2, 0.9, 1.1), S.Color(3.9, 2.4, 2.2))
    lamp = S.Shape.new_disk((-4, 6, 7), 90, 35, 1.5, 0.25)
    prims = [S.Primitive.new(S.Shape.new_disk((0, 0, 4), 90, 0, 25, 0), floor),
             S.Primitive.new(S.Shape.new_sphere((1, 1.25, 4), 1.25), ball),
             S.Primitive.new(S.Shape.new_sphere((-2, 0.5, 2.5), 0.5), bearing),
             S.Primitive.new_area_light(lamp, S.Light.Area(lamp, S.Color(9, 6.5, 2)))]
    cam = S.Camera.perspective(S.Film(320, 200), (6, 4.5, -9), (0.5, 1, 3), (0, 1, 0), 38)
    built = S.Scene(5, 32, cam, [S.Light.Infinite(S.Color(0.03, 0.06, 0.4))], prims)
    a, b = desc_arrays(parsed), desc_arrays(built)
    for k in ('spheres', 'disks', 'lights'):
        assert a[k].tobytes() == b[k].tobytes(), k
    for col in ('shape_kind', 'shape', 'light'):   # material ids are numbered differently by the two front ends
        assert a['prims'][col].tolist() == b['prims'][col].tolist(), col
    assert (parsed.desc().max_depth, parsed.desc().num_samples) == (5, 32)
    ha, hb = backend.HostScene(parsed), backend.HostScene(built)
    na, ra = ha.bvh()
    nb, rb = hb.bvh()
    assert na.tobytes() == nb.tobytes() and np.array_equal(ra, rb)
    assert np.array_equal(ha.light_cdf(), hb.light_cdf())
    for x, y in zip(ha.camera_matrices(), hb.camera_matrices()):
        assert np.array_equal(x, y)


@pytest.mark.parametrize('name', ['shapes_and_lights', 'material_zoo'])
def test_scene_files_parse(name):
    sc = cry.load_scene_file(os.path.join(GOLDEN, 'scenes', name + '.cry'))
    assert sc.n_prims > 5 and sc.warnings == 0
    assert backend.HostScene(sc).flat.n_nodes > 1


# ---- OBJ / MTL ingest (src/obj.rs) ------------------------------------------------------------------
OBJ = """# quad with per-corner normals and uvs, then a bare triangle, then an emissive and a glass face
mtllib mesh.mtl
o quad
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vn 0 0 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
usemtl shiny
f 1/1/1 2/2/1 3/3/1 4/4/1
o bare
v 0 0 1
v 1 0 1
v 0 1 1
usemtl lamp
f 5 6 7
usemtl pane
f -3 -2 -1
usemtl wood
f 5 6 6
"""
MTL = """newmtl shiny
Kd 0.9 0.8 0.5
Ks 3.0 2.0 1.0
Ns 250
illum 4
newmtl lamp
Kd 0 0 0
Ke 4 5 6
newmtl pane
Kd 0.2 0.3 0.4
d 0.1
Ni 1.1
newmtl wood
Ns 1000
Ks 0.5 0.5 0.5
map_Kd wood.png
"""


def test_obj_mtl_ingest(tmp_path):
    from PIL import Image
    (tmp_path / 'mesh.obj').write_text(OBJ)
    (tmp_path / 'mesh.mtl').write_text(MTL)
    Image.fromarray(np.arange(4 * 5 * 3, dtype=np.uint8).reshape(4, 5, 3)).save(tmp_path / 'wood.png')
    text = MINI % ("", "Mesh { file_name: 'mesh.obj', fallback_material: 'm' }")
    sc = cry.parse_scene(text, base_dir=str(tmp_path))
    d = sc.desc()
    a = desc_arrays(sc)
    # quad -> fan (0,1,2), (0,2,3); bare triangle emissive; pane glass; degenerate wood face skipped
    assert d.n_triangles == 4 and d.n_prims == 4
    t0 = a['triangles'][0]
    assert tuple(t0['v0']) == (0, 0, 0) and tuple(t0['e1']) == (1, 0, 0) and tuple(t0['e2']) == (1, 1, 0)
    assert tuple(t0['n0']) == (0, 0, -1) and tuple(t0['n01']) == (0, 0, 0)          # vn z flipped (obj.rs:140)
    assert t0['uv0'].tolist() == [0, 1] and t0['uv01'].tolist() == [1, 0] and t0['uv02'].tolist() == [1, -1]   # v -> 1 - v
    t2 = a['triangles'][2]
    assert tuple(t2['v0']) == (0, 0, -1)                                            # z flip (obj.rs:131)
    # flat normal (vk - vi) x (vj - vi) (obj.rs:159) and default uvs
    assert tuple(t2['n0']) == (0, 0, -1) and t2['uv01'].tolist() == [1, 0] and t2['uv02'].tolist() == [1, 1]
    # materials: illum 4 -> Metal(eta = Kd, k = Ks); Ke -> area light per triangle; d < 1 -> Glass(Kd, Kd, Ni)
    mats = np.ctypeslib.as_array(backend.C.cast(d.materials, backend.C.POINTER(backend.C.c_uint8)), shape=(d.n_materials * 16,)).view(S.MATERIAL_DT)
    bx = np.ctypeslib.as_array(backend.C.cast(d.bxdfs, backend.C.POINTER(backend.C.c_uint8)), shape=(d.n_bxdfs * 104,)).view(S.BXDF_DT)
    p = a['prims']
    m_quad = mats[p['material'][0]]
    assert m_quad['is_bsdf'] == 1 and bx[m_quad['first_bxdf']]['kind'] == S.BXDF_FRESNEL_CONDUCTOR
    assert p['material'][2] == -1 and p['light'][2] == 0 and tuple(a['lights'][0]['c']) == (4, 5, 6)
    m_pane = mats[p['material'][3]]
    assert m_pane['is_bsdf'] == 0 and bx[m_pane['first_bxdf']]['kind'] == S.BXDF_FRESNEL_SPECULAR and bx[m_pane['first_bxdf']]['eta_t'] == 1.1
    # the wood material decoded its texture through the image loader: 5x4 RGB8
    assert d.n_images == 1
    img = np.ctypeslib.as_array(backend.C.cast(d.images, backend.C.POINTER(backend.C.c_uint8)), shape=(16,)).view(S.IMAGE_DT)[0]
    assert (img['width'], img['height']) == (5, 4) and d.image_pool_bytes == 60
    assert backend.HostScene(sc).flat.n_nodes >= 1
