"""Host-side mirror of craytracer's scene-construction API.

Same names and argument meaning as the reference's Rust constructors, so scene
set-up code (and the parity tests) read like the reference's:

    Color, Texture.constant/checkerboard/image      src/color.rs, src/texture.rs:49-59
    Material.new_matte/new_glass/new_plastic/new_metal   src/material.rs:19-70
    Shape.new_sphere/new_triangle/new_disk/...      src/shape.rs:55-153
    Light.Point/Distant/Infinite/Area               src/light.rs:25-43
    Primitive.new / Primitive.new_area_light        src/primitive.rs:28-48
    Film, Camera.perspective / Camera.orthographic  src/film.rs, src/camera.rs:78-129
    Scene(max_depth, num_samples, camera, lights, primitives)   src/scene.rs:25-31

The output is the POD `cray_scene_desc` of include/cray_scene_desc.h.  No
rendering arithmetic happens here: only the constructors' own arithmetic (edge
vectors, flat normals, MTL heuristics) in f64 via numpy, which evaluates the same
IEEE operations in the same order as the Rust code.
"""
import ctypes as C
import math

import numpy as np

# ---------------------------------------------------------------------------
# POD layouts (must match include/cray_scene_desc.h)
# ---------------------------------------------------------------------------
VEC3 = [('x', '<f8'), ('y', '<f8'), ('z', '<f8')]
COLOR = [('r', '<f8'), ('g', '<f8'), ('b', '<f8')]

TEXTURE_DT = np.dtype([('kind', '<i4'), ('image', '<i4'), ('a', COLOR), ('b', COLOR), ('scale', '<f8')], align=True)
IMAGE_DT = np.dtype([('width', '<u4'), ('height', '<u4'), ('offset', '<u8')], align=True)
BXDF_DT = np.dtype([('kind', '<i4'), ('tex_a', '<i4'), ('tex_b', '<i4'), ('fresnel_kind', '<i4'),
                    ('eta_i', '<f8'), ('eta_t', '<f8'), ('c_eta_i', COLOR), ('c_eta_t', COLOR), ('c_k', COLOR)],
                   align=True)
MATERIAL_DT = np.dtype([('is_bsdf', '<i4'), ('n_bxdfs', '<i4'), ('first_bxdf', '<i4'), ('pad_', '<i4')], align=True)
SPHERE_DT = np.dtype([('origin', VEC3), ('radius', '<f8')], align=True)
DISK_DT = np.dtype([('origin', VEC3), ('rotate_x', '<f8'), ('rotate_y', '<f8'), ('radius', '<f8'),
                    ('inner_radius', '<f8')], align=True)
TRIANGLE_DT = np.dtype([('v0', VEC3), ('e1', VEC3), ('e2', VEC3), ('n0', VEC3), ('n01', VEC3), ('n02', VEC3),
                        ('uv0', '<f8', 2), ('uv01', '<f8', 2), ('uv02', '<f8', 2)], align=True)
PRIM_DT = np.dtype([('shape_kind', '<i4'), ('shape', '<u4'), ('material', '<i4'), ('light', '<i4')], align=True)
LIGHT_DT = np.dtype([('kind', '<i4'), ('prim', '<i4'), ('v', VEC3), ('c', COLOR)], align=True)

assert TEXTURE_DT.itemsize == 64 and IMAGE_DT.itemsize == 16 and BXDF_DT.itemsize == 104
assert MATERIAL_DT.itemsize == 16 and SPHERE_DT.itemsize == 32 and DISK_DT.itemsize == 56
assert TRIANGLE_DT.itemsize == 192 and PRIM_DT.itemsize == 16 and LIGHT_DT.itemsize == 56

TEX_CONSTANT, TEX_CHECKERBOARD, TEX_IMAGE = 0, 1, 2
BXDF_LAMBERTIAN, BXDF_OREN_NAYAR, BXDF_FRESNEL_CONDUCTOR, BXDF_SPECULAR_BRDF, BXDF_SPECULAR_BTDF, \
    BXDF_FRESNEL_SPECULAR = range(6)
FRESNEL_DIELECTRIC, FRESNEL_CONDUCTOR = 0, 1
SHAPE_SPHERE, SHAPE_TRIANGLE, SHAPE_DISK = 0, 1, 2
LIGHT_POINT, LIGHT_DISTANT, LIGHT_INFINITE, LIGHT_AREA = 0, 1, 2, 3
CAMERA_PERSPECTIVE, CAMERA_ORTHOGRAPHIC = 0, 1

DEFAULT_MAX_DEPTH = 8          # scene_parser.rs:796
DEFAULT_NUM_SAMPLES = 4        # scene_parser.rs:797
DEFAULT_FOCAL_DISTANCE = 1e6   # scene_parser.rs:798


class CVec3(C.Structure):
    _fields_ = [('x', C.c_double), ('y', C.c_double), ('z', C.c_double)]


class CCameraDesc(C.Structure):
    _fields_ = [('type', C.c_int32), ('film_width', C.c_uint32), ('film_height', C.c_uint32), ('pad_', C.c_int32),
                ('origin', CVec3), ('target', CVec3), ('up', CVec3), ('fov', C.c_double),
                ('lens_radius', C.c_double), ('focal_distance', C.c_double)]


class CSceneDesc(C.Structure):
    _fields_ = [('max_depth', C.c_uint32), ('num_samples', C.c_uint32), ('camera', CCameraDesc),
                ('n_spheres', C.c_uint32), ('spheres', C.c_void_p),
                ('n_disks', C.c_uint32), ('disks', C.c_void_p),
                ('n_triangles', C.c_uint32), ('triangles', C.c_void_p),
                ('n_prims', C.c_uint32), ('prims', C.c_void_p),
                ('n_lights', C.c_uint32), ('lights', C.c_void_p),
                ('n_materials', C.c_uint32), ('materials', C.c_void_p),
                ('n_bxdfs', C.c_uint32), ('bxdfs', C.c_void_p),
                ('n_textures', C.c_uint32), ('textures', C.c_void_p),
                ('n_images', C.c_uint32), ('images', C.c_void_p),
                ('image_pool_bytes', C.c_uint64), ('image_pool', C.c_void_p)]


# ---------------------------------------------------------------------------
# Value types
# ---------------------------------------------------------------------------
class Color:
    """src/color.rs:6-11"""
    __slots__ = ('r', 'g', 'b')

    def __init__(self, r, g, b):
        self.r, self.g, self.b = float(r), float(g), float(b)

    def is_black(self):
        return self.r == 0.0 and self.g == 0.0 and self.b == 0.0

    def __iter__(self):
        return iter((self.r, self.g, self.b))

    def __eq__(self, o):
        return isinstance(o, Color) and tuple(self) == tuple(o)

    def __repr__(self):
        return 'Color(%r, %r, %r)' % (self.r, self.g, self.b)


Color.BLACK = Color(0, 0, 0)
Color.WHITE = Color(1, 1, 1)


def _as_color(c):
    return c if isinstance(c, Color) else Color(*c)


class Texture:
    """Texture<T> with T = Color or f64 (src/texture.rs:7-12)."""

    def __init__(self, kind, a=None, b=None, scale=1.0, image=None):
        self.kind, self.a, self.b, self.scale, self.img = kind, a, b, float(scale), image

    @staticmethod
    def constant(t):
        return Texture(TEX_CONSTANT, a=t)

    @staticmethod
    def checkerboard(a, b, scale):
        return Texture(TEX_CHECKERBOARD, a=a, b=b, scale=scale)

    @staticmethod
    def image(rgb8):
        """`image.to_rgb8()` result: uint8 array [height, width, 3] (src/texture.rs:57-58)."""
        arr = np.ascontiguousarray(rgb8, dtype=np.uint8)
        assert arr.ndim == 3 and arr.shape[2] == 3
        return Texture(TEX_IMAGE, image=arr)

    # src/texture.rs:83-101
    def is_black(self):
        if self.kind == TEX_CONSTANT:
            return self.a.is_black()
        if self.kind == TEX_CHECKERBOARD:
            return self.a.is_black() and self.b.is_black()
        return False

    def is_zero(self):
        if self.kind == TEX_CONSTANT:
            return self.a == 0.0
        if self.kind == TEX_CHECKERBOARD:
            return self.a == 0.0 and self.b == 0.0
        return False


def _tex(t, scalar):
    """Shorthand used by the .cry parser (scene_parser.rs:911-921): a bare value is a constant texture."""
    if isinstance(t, Texture):
        return t
    return Texture.constant(float(t) if scalar else _as_color(t))


class BxDF:
    def __init__(self, kind, tex_a=None, tex_b=None, fresnel_kind=0, eta_i=0.0, eta_t=0.0,
                 c_eta_i=Color.BLACK, c_eta_t=Color.BLACK, c_k=Color.BLACK):
        self.kind, self.tex_a, self.tex_b = kind, tex_a, tex_b
        self.fresnel_kind, self.eta_i, self.eta_t = fresnel_kind, float(eta_i), float(eta_t)
        self.c_eta_i, self.c_eta_t, self.c_k = c_eta_i, c_eta_t, c_k


class Material:
    """src/material.rs:13-70"""

    def __init__(self, is_bsdf, bxdfs):
        self.is_bsdf, self.bxdfs = is_bsdf, list(bxdfs)

    @staticmethod
    def new_matte(reflectance, sigma):
        reflectance, sigma = _tex(reflectance, False), _tex(sigma, True)
        if sigma.is_zero():
            return Material(False, [BxDF(BXDF_LAMBERTIAN, reflectance)])
        return Material(False, [BxDF(BXDF_OREN_NAYAR, reflectance, sigma)])

    @staticmethod
    def new_glass(reflectance, transmittance, eta):
        return Material(False, [BxDF(BXDF_FRESNEL_SPECULAR, _tex(reflectance, False), _tex(transmittance, False),
                                     eta_i=1.0, eta_t=eta)])

    @staticmethod
    def new_plastic(diffuse, specular, roughness):
        diffuse, specular, roughness = _tex(diffuse, False), _tex(specular, False), _tex(roughness, True)
        bxdfs = []
        if not diffuse.is_black():
            if not roughness.is_zero():
                bxdfs.append(BxDF(BXDF_OREN_NAYAR, diffuse, roughness))
            else:
                bxdfs.append(BxDF(BXDF_LAMBERTIAN, diffuse))
        if not specular.is_black():
            bxdfs.append(BxDF(BXDF_SPECULAR_BRDF, specular, fresnel_kind=FRESNEL_DIELECTRIC, eta_i=1.0, eta_t=1.5))
        return Material(True, bxdfs)

    @staticmethod
    def new_metal(eta, k):
        return Material(True, [BxDF(BXDF_FRESNEL_CONDUCTOR, _tex(eta, False), _tex(k, False))])


def _v(p):
    return np.asarray(p, dtype=np.float64).reshape(3)


class Shape:
    """src/shape.rs:24-153. A Shape is one of: sphere, disk, or a *batch* of triangles
    (a batch of one for Shape.new_triangle)."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    @staticmethod
    def new_sphere(origin, radius):
        return Shape(SHAPE_SPHERE, origin=_v(origin), radius=float(radius))

    @staticmethod
    def new_disk(origin, rotate_x, rotate_y, radius, inner_radius):
        return Shape(SHAPE_DISK, origin=_v(origin), rotate_x=float(rotate_x), rotate_y=float(rotate_y),
                     radius=float(radius), inner_radius=float(inner_radius))

    @staticmethod
    def new_triangle(v0, v1, v2):
        """src/shape.rs:70-95; None for a degenerate triangle."""
        tris = triangles_flat(np.stack([_v(v0), _v(v1), _v(v2)])[None])
        if len(tris) == 0:
            return None
        return Shape(SHAPE_TRIANGLE, tris=tris)

    @staticmethod
    def new_triangle_with_normals_and_texture_coordinates(v0, v1, v2, n0, n1, n2, uv0, uv1, uv2):
        """src/shape.rs:96-132"""
        tris = triangles_full(np.stack([_v(v0), _v(v1), _v(v2)])[None], np.stack([_v(n0), _v(n1), _v(n2)])[None],
                              np.asarray([uv0, uv1, uv2], dtype=np.float64)[None])
        if len(tris) == 0:
            return None
        return Shape(SHAPE_TRIANGLE, tris=tris)


def _cross(a, b):
    # src/geometry.rs:58-64, component order and operation order preserved
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                     a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                     a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)


def _dot(a, b):
    return a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1] + a[:, 2] * b[:, 2]


def _set3(arr, field, v):
    arr[field]['x'], arr[field]['y'], arr[field]['z'] = v[:, 0], v[:, 1], v[:, 2]


def triangles_flat(verts):
    """Vectorised Shape::new_triangle (src/shape.rs:70-95) over verts[n, 3, 3]; drops degenerate ones."""
    verts = np.asarray(verts, dtype=np.float64)
    v0, v1, v2 = verts[:, 0], verts[:, 1], verts[:, 2]
    e1, e2 = v1 - v0, v2 - v0
    n0 = _cross(e2, e1)
    mag = np.sqrt(_dot(n0, n0))
    keep = mag != 0.0
    v0, e1, e2, n0, mag = v0[keep], e1[keep], e2[keep], n0[keep], mag[keep]
    n0 = n0 / mag[:, None]
    out = np.zeros(len(v0), dtype=TRIANGLE_DT)
    _set3(out, 'v0', v0); _set3(out, 'e1', e1); _set3(out, 'e2', e2); _set3(out, 'n0', n0)
    out['uv0'] = (0.0, 0.0); out['uv01'] = (1.0, 0.0); out['uv02'] = (1.0, 1.0)
    return out


def triangles_full(verts, normals, uvs):
    """Vectorised new_triangle_with_normals_and_texture_coordinates (src/shape.rs:96-132)."""
    verts = np.asarray(verts, dtype=np.float64)
    normals = np.asarray(normals, dtype=np.float64)
    uvs = np.asarray(uvs, dtype=np.float64)
    v0, v1, v2 = verts[:, 0], verts[:, 1], verts[:, 2]
    n0, n1, n2 = normals[:, 0], normals[:, 1], normals[:, 2]
    e1, e2 = v1 - v0, v2 - v0
    c = _cross(e2, e1)
    keep = (_dot(c, c) != 0.0) & (_dot(n0, n0) != 0.0) & (_dot(n1, n1) != 0.0) & (_dot(n2, n2) != 0.0)
    out = np.zeros(int(keep.sum()), dtype=TRIANGLE_DT)
    _set3(out, 'v0', v0[keep]); _set3(out, 'e1', e1[keep]); _set3(out, 'e2', e2[keep])
    _set3(out, 'n0', n0[keep]); _set3(out, 'n01', (n1 - n0)[keep]); _set3(out, 'n02', (n2 - n0)[keep])
    out['uv0'] = uvs[keep, 0]; out['uv01'] = (uvs[:, 1] - uvs[:, 0])[keep]; out['uv02'] = (uvs[:, 2] - uvs[:, 0])[keep]
    return out


class Light:
    """src/light.rs:25-43"""

    def __init__(self, kind, v=(0, 0, 0), c=Color.BLACK, shape=None):
        self.kind, self.v, self.c, self.shape = kind, _v(v), _as_color(c), shape

    @staticmethod
    def Point(origin, intensity):
        return Light(LIGHT_POINT, origin, intensity)

    @staticmethod
    def Distant(direction, intensity):
        """The .cry parser normalises the direction (scene_parser.rs:888); so do we."""
        d = _v(direction)
        mag = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        return Light(LIGHT_DISTANT, d / mag, intensity)

    @staticmethod
    def Infinite(intensity):
        return Light(LIGHT_INFINITE, c=intensity)

    @staticmethod
    def Area(shape, emittance):
        return Light(LIGHT_AREA, c=emittance, shape=shape)


class Primitive:
    """src/primitive.rs:14-48"""

    def __init__(self, shape, material=None, area_light=None):
        self.shape, self.material, self.area_light = shape, material, area_light

    @staticmethod
    def new(shape, material):
        return Primitive(shape, material=material)

    @staticmethod
    def new_area_light(shape, area_light):
        assert area_light.kind == LIGHT_AREA, 'Non area light provided as area light for shape'
        return Primitive(shape, area_light=area_light)

    def get_area_light(self):
        return self.area_light


class Mesh:
    """What `obj::load_obj` returns for one model (src/obj.rs:108-199): a run of triangle
    primitives sharing one material, or — when `emittance` is set — one area light per triangle."""

    def __init__(self, tris, material=None, emittance=None):
        assert tris.dtype == TRIANGLE_DT
        self.tris, self.material = tris, material
        self.emittance = _as_color(emittance) if emittance is not None else None

    @staticmethod
    def from_indexed(vertices, indices, material=None, normals=None, uvs=None, emittance=None):
        """vertices[n,3] (already left-handed), indices[m,3]; mirrors src/obj.rs:153-199."""
        vertices = np.asarray(vertices, dtype=np.float64)
        indices = np.asarray(indices, dtype=np.int64)
        verts = vertices[indices]                      # [m,3,3]
        vi, vj, vk = verts[:, 0], verts[:, 1], verts[:, 2]
        if normals is None:
            c = _cross(vk - vi, vj - vi)               # obj.rs:159
            mag = np.sqrt(_dot(c, c))
            with np.errstate(invalid='ignore', divide='ignore'):
                flat = c / mag[:, None]
            nrm = np.repeat(flat[:, None, :], 3, axis=1)
        else:
            nrm = np.asarray(normals, dtype=np.float64)[indices]
        if uvs is None:
            uv = np.broadcast_to(np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0]]), (len(indices), 3, 2))  # obj.rs:169-171
        else:
            uv = np.asarray(uvs, dtype=np.float64)[indices]
        return Mesh(triangles_full(verts, nrm, uv), material=material, emittance=emittance)


class Film:
    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)


class Camera:
    """src/camera.rs:56-129"""

    def __init__(self, camera_type, film, origin, target, up, fov, lens_radius, focal_distance):
        self.camera_type, self.film = camera_type, film
        self.origin, self.target, self.up = _v(origin), _v(target), _v(up)
        self.fov, self.lens_radius, self.focal_distance = float(fov), float(lens_radius), float(focal_distance)

    @staticmethod
    def perspective(film, origin, target, up, fov, lens_radius=0.0, focal_distance=DEFAULT_FOCAL_DISTANCE):
        return Camera(CAMERA_PERSPECTIVE, film, origin, target, up, fov, lens_radius, focal_distance)

    @staticmethod
    def orthographic(film, origin, target, up, lens_radius=0.0, focal_distance=DEFAULT_FOCAL_DISTANCE):
        return Camera(CAMERA_ORTHOGRAPHIC, film, origin, target, up, 0.0, lens_radius, focal_distance)


# ---------------------------------------------------------------------------
# Scene::new arguments -> cray_scene_desc
# ---------------------------------------------------------------------------
class Scene:
    """`Scene::new(max_depth, num_samples, camera, lights, primitives)` (src/scene.rs:25-31).

    `lights` holds the explicit lights only; area lights of the primitives are appended in
    primitive order exactly as `parse_scene` does (scene_parser.rs:1093-1101).
    `primitives` may mix Primitive and Mesh entries.
    """

    def __init__(self, max_depth, num_samples, camera, lights, primitives):
        self.max_depth, self.num_samples, self.camera = int(max_depth), int(num_samples), camera
        self._tex_ids, self._textures, self._images, self._pool = {}, [], [], []
        self._pool_bytes = 0
        self._mat_ids, self._materials, self._bxdfs = {}, [], []
        spheres, disks, tri_chunks, prim_chunks = [], [], [], []
        light_rows = [(l.kind, -1, l.v, l.c) for l in lights]
        n_tris = 0
        n_prims = 0
        for p in primitives:
            if isinstance(p, Mesh):
                m = len(p.tris)
                if m == 0:
                    continue
                chunk = np.zeros(m, dtype=PRIM_DT)
                chunk['shape_kind'] = SHAPE_TRIANGLE
                chunk['shape'] = np.arange(n_tris, n_tris + m, dtype=np.uint32)
                if p.emittance is not None:
                    chunk['material'] = -1
                    chunk['light'] = np.arange(len(light_rows), len(light_rows) + m, dtype=np.int32)
                    light_rows.extend((LIGHT_AREA, n_prims + i, np.zeros(3), p.emittance) for i in range(m))
                else:
                    chunk['material'] = self._material_id(p.material)
                    chunk['light'] = -1
                tri_chunks.append(p.tris)
                prim_chunks.append(chunk)
                n_tris += m
                n_prims += m
                continue
            sh = p.shape
            chunk = np.zeros(1, dtype=PRIM_DT)
            chunk['shape_kind'] = sh.kind
            if sh.kind == SHAPE_SPHERE:
                chunk['shape'] = len(spheres)
                spheres.append((tuple(sh.origin), sh.radius))
            elif sh.kind == SHAPE_DISK:
                chunk['shape'] = len(disks)
                disks.append((tuple(sh.origin), sh.rotate_x, sh.rotate_y, sh.radius, sh.inner_radius))
            else:
                assert len(sh.tris) == 1
                chunk['shape'] = n_tris
                tri_chunks.append(sh.tris)
                n_tris += 1
            if p.area_light is not None:
                chunk['material'] = -1
                chunk['light'] = len(light_rows)
                light_rows.append((LIGHT_AREA, n_prims, np.zeros(3), p.area_light.c))
            else:
                chunk['material'] = self._material_id(p.material)
                chunk['light'] = -1
            prim_chunks.append(chunk)
            n_prims += 1

        if len(light_rows) == 0:
            raise ValueError('No lights in the scene.')  # scene_parser.rs:1104-1109

        self.spheres = np.array(spheres, dtype=SPHERE_DT) if spheres else np.zeros(0, dtype=SPHERE_DT)
        self.disks = np.array(disks, dtype=DISK_DT) if disks else np.zeros(0, dtype=DISK_DT)
        self.triangles = np.concatenate(tri_chunks) if tri_chunks else np.zeros(0, dtype=TRIANGLE_DT)
        self.prims = np.concatenate(prim_chunks) if prim_chunks else np.zeros(0, dtype=PRIM_DT)
        self.lights = np.zeros(len(light_rows), dtype=LIGHT_DT)
        for i, (kind, prim, v, c) in enumerate(light_rows):
            self.lights[i] = (kind, prim, tuple(v), tuple(c))
        self.materials = np.array(self._materials, dtype=MATERIAL_DT) if self._materials else np.zeros(0, MATERIAL_DT)
        self.bxdfs = np.zeros(len(self._bxdfs), dtype=BXDF_DT)
        for i, row in enumerate(self._bxdfs):
            self.bxdfs[i] = row
        self.textures = np.zeros(len(self._textures), dtype=TEXTURE_DT)
        for i, row in enumerate(self._textures):
            self.textures[i] = row
        self.images = np.array(self._images, dtype=IMAGE_DT) if self._images else np.zeros(0, dtype=IMAGE_DT)
        self.image_pool = (np.concatenate([a.reshape(-1) for a in self._pool]) if self._pool
                           else np.zeros(0, dtype=np.uint8))
        self._desc = None

    # -- interning -----------------------------------------------------------
    def _texture_id(self, t, scalar):
        if t is None:
            return -1
        key = id(t)
        if key in self._tex_ids:
            return self._tex_ids[key][0]
        img = -1
        a = b = (0.0, 0.0, 0.0)
        if t.kind == TEX_IMAGE:
            img = len(self._images)
            self._images.append((t.img.shape[1], t.img.shape[0], self._pool_bytes))
            self._pool.append(t.img)
            self._pool_bytes += t.img.size
        elif scalar:
            a = (float(t.a), 0.0, 0.0)
            b = (float(t.b), 0.0, 0.0) if t.kind == TEX_CHECKERBOARD else b
        else:
            a = tuple(_as_color(t.a))
            b = tuple(_as_color(t.b)) if t.kind == TEX_CHECKERBOARD else b
        idx = len(self._textures)
        self._textures.append((t.kind, img, a, b, t.scale))
        self._tex_ids[key] = (idx, t)  # keep t alive so id() stays unique
        return idx

    def _material_id(self, m):
        key = id(m)
        if key in self._mat_ids:
            return self._mat_ids[key][0]
        first = len(self._bxdfs)
        for bx in m.bxdfs:
            scalar_b = bx.kind == BXDF_OREN_NAYAR
            self._bxdfs.append((bx.kind, self._texture_id(bx.tex_a, False), self._texture_id(bx.tex_b, scalar_b),
                                bx.fresnel_kind, bx.eta_i, bx.eta_t, tuple(bx.c_eta_i), tuple(bx.c_eta_t),
                                tuple(bx.c_k)))
        idx = len(self._materials)
        self._materials.append((1 if m.is_bsdf else 0, len(m.bxdfs), first, 0))
        self._mat_ids[key] = (idx, m)
        return idx

    # -- POD view --------------------------------------------------------------
    def film_bounds(self):
        return self.camera.film.width, self.camera.film.height

    def desc(self):
        """ctypes `cray_scene_desc` whose pointers stay valid while this Scene is alive."""
        if self._desc is not None:
            return self._desc
        d = CSceneDesc()
        d.max_depth, d.num_samples = self.max_depth, self.num_samples
        cam = self.camera
        d.camera.type = cam.camera_type
        d.camera.film_width, d.camera.film_height = cam.film.width, cam.film.height
        for name in ('origin', 'target', 'up'):
            v = getattr(cam, name)
            setattr(d.camera, name, CVec3(v[0], v[1], v[2]))
        d.camera.fov, d.camera.lens_radius, d.camera.focal_distance = cam.fov, cam.lens_radius, cam.focal_distance

        def put(count_field, ptr_field, arr):
            arr = np.ascontiguousarray(arr)
            setattr(self, '_keep_' + ptr_field, arr)
            setattr(d, count_field, len(arr))
            setattr(d, ptr_field, arr.ctypes.data if len(arr) else None)

        put('n_spheres', 'spheres', self.spheres)
        put('n_disks', 'disks', self.disks)
        put('n_triangles', 'triangles', self.triangles)
        put('n_prims', 'prims', self.prims)
        put('n_lights', 'lights', self.lights)
        put('n_materials', 'materials', self.materials)
        put('n_bxdfs', 'bxdfs', self.bxdfs)
        put('n_textures', 'textures', self.textures)
        put('n_images', 'images', self.images)
        pool = np.ascontiguousarray(self.image_pool)
        self._keep_pool = pool
        d.image_pool_bytes = pool.size
        d.image_pool = pool.ctypes.data if pool.size else None
        self._desc = d
        return d
