"""The reference's own scene files and meshes, read IN PLACE through the product's reader (build container only: the
reference tree never travels to the GPU box, and nothing of it is copied into this repository).

  scenes/{simple,test,materials,anthropic,rounding-error}.cry  parse + Scene::new mirror, tree == the oracle's
  objs/anthropic.obj          20 060 faces -> 20 060 triangles (none degenerate), z flipped (obj.rs:131), flat normals
  objs/staircase/staircase.mtl  every material through the MTL -> Material rules of src/obj.rs:61-105, with the ten
                              `map_Kd` JPEGs decoded by the library itself (cray_load_image)"""
import math
import os
import re

import numpy as np
import pytest

from craytracer_amd import backend, cry
from craytracer_amd import scene as S
from oracle import oracle_lib as ol

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'scenes')), reason='the reference tree is only present in the build container')


def _arrays(sc):
    d = sc.desc()
    u8 = backend.C.POINTER(backend.C.c_uint8)

    def view(ptr, n, dt):
        return np.ctypeslib.as_array(backend.C.cast(ptr, u8), shape=(n * dt.itemsize,)).view(dt) if n else np.zeros(0, dt)
    return d, {'materials': view(d.materials, d.n_materials, S.MATERIAL_DT), 'bxdfs': view(d.bxdfs, d.n_bxdfs, S.BXDF_DT),
               'textures': view(d.textures, d.n_textures, S.TEXTURE_DT), 'images': view(d.images, d.n_images, S.IMAGE_DT),
               'triangles': view(d.triangles, d.n_triangles, S.TRIANGLE_DT), 'prims': view(d.prims, d.n_prims, S.PRIM_DT)}


@pytest.mark.parametrize('name,film,spp,n_prims,n_lights', [
    ('simple', (700, 400), 256, 3, 2), ('test', None, None, None, None), ('materials', None, 1000, None, None),
    ('rounding-error', None, None, None, None)])
def test_reference_scene_files_parse_and_build(name, film, spp, n_prims, n_lights):
    sc = cry.load_scene_file(os.path.join(REF, 'scenes', name + '.cry'), base_dir=REF, image_loader=None)
    d = sc.desc()
    assert sc.warnings == 0                       # no unused keys: the reference's own files use only known keys
    if film:
        assert sc.film_bounds() == film
    if spp:
        assert sc.num_samples == spp
    if n_prims:
        assert d.n_prims == n_prims and d.n_lights == n_lights
    host, orc = backend.HostScene(sc), ol.OracleScene(sc)
    hn, hr = host.bvh()
    on, orf = orc.bvh()
    assert len(hn) == len(on) and np.array_equal(hn['bmin'], on['bmin']) and np.array_equal(hn['bmax'], on['bmax']) and np.array_equal(hr, orf)
    assert np.array_equal(host.light_cdf(), orc.light_cdf())


def test_anthropic_scene_and_mesh():
    sc = cry.load_scene_file(os.path.join(REF, 'scenes', 'anthropic.cry'), base_dir=REF, image_loader=None)
    d, a = _arrays(sc)
    assert sc.film_bounds() == (800, 600) and sc.num_samples == 1024 and sc.warnings == 0
    # ground disk + 20 060 mesh triangles + the emissive disk; lights = Infinite, then the area light
    assert d.n_triangles == 20060 and d.n_prims == 20062 and d.n_lights == 2
    text = open(os.path.join(REF, 'objs', 'anthropic.obj')).read()
    V = np.array([[float(x) for x in m.groups()] for m in re.finditer(r'^v (\S+) (\S+) (\S+)', text, re.M)])
    faces = [[int(c.split('/')[0]) for c in m.group(1).split()] for m in re.finditer(r'^f (.+)$', text, re.M)]
    assert len(V) == 10032 and len(faces) == 20060 and all(len(f) == 3 for f in faces)
    V[:, 2] = -V[:, 2]                                              # RH -> LH (obj.rs:131)
    F = np.array(faces) - 1
    t = a['triangles']
    assert np.array_equal(_xyz(t['v0']), V[F[:, 0]])
    assert np.array_equal(_xyz(t['e1']), V[F[:, 1]] - V[F[:, 0]]) and np.array_equal(_xyz(t['e2']), V[F[:, 2]] - V[F[:, 0]])
    lo, hi = V[F.reshape(-1)].min(0), V[F.reshape(-1)].max(0)
    # every face names a normal (`f v//vn`): the corners get the file's vn with z flipped (obj.rs:140), not the flat normal
    VN = np.array([[float(x) for x in m.groups()] for m in re.finditer(r'^vn (\S+) (\S+) (\S+)', text, re.M)])
    VN[:, 2] = -VN[:, 2]
    FN = np.array([[int(c.split('/')[2]) for c in m.group(1).split()] for m in re.finditer(r'^f (.+)$', text, re.M)]) - 1
    assert len(VN) == 153
    assert np.array_equal(_xyz(t['n0']), VN[FN[:, 0]])
    assert np.array_equal(_xyz(t['n01']), VN[FN[:, 1]] - VN[FN[:, 0]]) and np.array_equal(_xyz(t['n02']), VN[FN[:, 2]] - VN[FN[:, 0]])
    assert (t['uv0'] == [0.0, 0.0]).all() and (t['uv01'] == [1.0, 0.0]).all() and (t['uv02'] == [1.0, 1.0]).all()   # no vt: obj.rs:166-168
    mesh_prims = a['prims'][1:-1]
    assert (mesh_prims['shape_kind'] == S.SHAPE_TRIANGLE).all() and (mesh_prims['material'] == mesh_prims['material'][0]).all()   # fallback 'text'
    m = a['materials'][mesh_prims['material'][0]]
    assert m['is_bsdf'] == 1 and m['n_bxdfs'] == 2                  # Plastic(diffuse, specular, roughness 120): Oren-Nayar + SpecularBRDF
    kinds = a['bxdfs']['kind'][m['first_bxdf']: m['first_bxdf'] + 2].tolist()
    assert kinds == [S.BXDF_OREN_NAYAR, S.BXDF_SPECULAR_BRDF]
    host = backend.HostScene(sc)
    nodes, _ = host.bvh()
    on, _ = ol.OracleScene(sc).bvh()
    assert len(nodes) == len(on) and np.array_equal(nodes['bmin'], on['bmin'])
    assert (nodes[0]['bmin'] <= lo + 1e-9).all() and (nodes[0]['bmax'] >= hi - 1e-9).all()   # the root holds the mesh (and the r = 20 ground disk)


def _xyz(a):
    return np.stack([a['x'], a['y'], a['z']], axis=-1)


def _parse_mtl(text):
    mats, cur = [], None
    for line in text.splitlines():
        w = line.split()
        if not w or w[0].startswith('#'):
            continue
        if w[0] == 'newmtl':
            cur = {'name': line.split(None, 1)[1].strip()}
            mats.append(cur)
        elif cur is not None:
            cur[w[0]] = line.split(None, 1)[1].strip() if len(w) > 1 else ''
    return mats


def test_staircase_mtl_through_the_material_mapping(tmp_path):
    """objs/staircase/staircase.mtl (the OBJ itself is a missing large blob): a generated one-triangle-per-material OBJ in
    a temp directory names the reference's MTL and textures through symlinks; every material must come out as
    obj.rs:61-105 prescribes."""
    src = os.path.join(REF, 'objs', 'staircase')
    os.symlink(os.path.join(src, 'staircase.mtl'), tmp_path / 'staircase.mtl')
    os.symlink(os.path.join(src, 'textures'), tmp_path / 'textures')
    mtl = _parse_mtl(open(os.path.join(src, 'staircase.mtl')).read())
    assert len(mtl) == 26 and sum('map_Kd' in m for m in mtl) == 10
    lines = ['mtllib staircase.mtl']
    for i, m in enumerate(mtl):
        lines += ['v %d 0 0' % i, 'v %d 1 0' % i, 'v %d 0 1' % i, 'usemtl ' + m['name'], 'f %d %d %d' % (3 * i + 1, 3 * i + 2, 3 * i + 3)]
    (tmp_path / 'one_each.obj').write_text('\n'.join(lines) + '\n')
    text = '''{ camera: Perspective { origin: Point(0,0,-5), target: Point(0,0,0), up: Vector(0,1,0), fov: 40, film: { width: 8, height: 8 } },
        lights: [ Point { origin: Point(0,5,0), intensity: Color(1,1,1) } ],
        materials: { fb: Matte { reflectance: Color(1,1,1), sigma: 0 } }, shapes: {},
        primitives: [ Mesh { file_name: 'one_each.obj', fallback_material: 'fb' } ] }'''
    sc = cry.parse_scene(text, base_dir=str(tmp_path), image_loader=None)      # textures decoded by cray_load_image
    d, a = _arrays(sc)
    assert d.n_triangles == 26 and d.n_images == 10
    from PIL import Image
    sizes = sorted((im['width'], im['height']) for im in a['images'])
    assert sizes == sorted(Image.open(os.path.join(src, m['map_Kd'])).size for m in mtl if 'map_Kd' in m)
    n_metal = n_glass = n_plastic = 0
    for i, m in enumerate(mtl):
        prim = a['prims'][i]
        ke = [float(x) for x in m.get('Ke', '0 0 0').split()]
        assert ke == [0, 0, 0] and prim['light'] == -1             # no emissive material in this file
        mat = a['materials'][prim['material']]
        bx = a['bxdfs'][mat['first_bxdf']: mat['first_bxdf'] + max(1, mat['n_bxdfs'])]
        kd_tex = a['textures'][bx[0]['tex_a']]
        if 'map_Kd' in m:
            assert kd_tex['kind'] == S.TEX_IMAGE
        else:
            assert kd_tex['kind'] == S.TEX_CONSTANT and np.allclose([kd_tex['a'][k] for k in 'rgb'], [float(x) for x in m['Kd'].split()])
        ns, dissolve, illum = float(m.get('Ns', 0)), float(m.get('d', 1)), int(m.get('illum', 2))
        ks = [float(x) for x in m.get('Ks', '0 0 0').split()]
        if dissolve < 1.0:                                          # Glass(Kd, Kd, Ni) (obj.rs:91-94)
            n_glass += 1
            assert mat['is_bsdf'] == 0 and bx[0]['kind'] == S.BXDF_FRESNEL_SPECULAR and bx[0]['eta_t'] == float(m.get('Ni', 1)) and bx[0]['tex_b'] == bx[0]['tex_a'] or \
                a['textures'][bx[0]['tex_b']]['kind'] == kd_tex['kind']
        elif 3 <= illum <= 9:                                       # Metal(eta = Kd, k = Ks) (obj.rs:98-99)
            n_metal += 1
            assert mat['is_bsdf'] == 1 and mat['n_bxdfs'] == 1 and bx[0]['kind'] == S.BXDF_FRESNEL_CONDUCTOR
            k_tex = a['textures'][bx[0]['tex_b']]
            assert np.allclose([k_tex['a'][k] for k in 'rgb'], ks)
        else:                                                       # Plastic(Kd, Ks, 180 (1 - e^(-Ns/100))) (obj.rs:84, 100)
            n_plastic += 1
            rough = 180.0 * (1.0 - math.pow(math.e, -ns / 100.0))
            assert mat['is_bsdf'] == 1 and mat['n_bxdfs'] == (2 if any(ks) else 1)
            assert bx[0]['kind'] == (S.BXDF_OREN_NAYAR if rough != 0.0 else S.BXDF_LAMBERTIAN)
            if rough != 0.0:
                assert a['textures'][bx[0]['tex_b']]['a']['r'] == rough
            if any(ks):
                assert bx[1]['kind'] == S.BXDF_SPECULAR_BRDF and bx[1]['eta_i'] == 1.0 and bx[1]['eta_t'] == 1.5
    assert n_glass >= 1 and n_metal >= 3 and n_plastic >= 15 and n_glass + n_metal + n_plastic == 26
