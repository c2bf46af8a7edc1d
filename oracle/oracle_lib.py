"""ctypes binding of the CPU oracle (oracle/libcray_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libcray_oracle.so')


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims',
                 'closest_hits', 'closest_tri_tests', 'shadow_tri_tests', 'paths', 'nonfinite', 'assert_fail')] + \
               [('seconds', C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


ORC_HIT_DT = np.dtype([('hit', '<i4'), ('prim', '<i4'), ('t', '<f8'), ('location', '<f8', 3), ('normal', '<f8', 3),
                       ('uv', '<f8', 2)], align=True)
ORC_NODE_DT = np.dtype([('bmin', '<f8', 3), ('bmax', '<f8', 3), ('left', '<u4'), ('right', '<u4'), ('first', '<u4'),
                        ('count', '<u4'), ('axis', '<i4'), ('leaf', '<i4')], align=True)
assert ORC_HIT_DT.itemsize == 80 and ORC_NODE_DT.itemsize == 72


def build(force=False):
    """Compile the oracle with g++ (see oracle/Makefile)."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ('cray_oracle.cpp', 'orc_math.h', 'sobol_rev_vectors.h')):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_build_error.argtypes = [C.c_void_p]
        L.orc_bvh_num_nodes.restype = C.c_uint32
        L.orc_bvh_num_nodes.argtypes = [C.c_void_p]
        L.orc_bvh_num_prim_refs.restype = C.c_uint32
        L.orc_bvh_num_prim_refs.argtypes = [C.c_void_p]
        L.orc_bvh_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_scene_light_cdf.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_render_pixel.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_path_log.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
        L.orc_camera_ray.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_siphash.restype = C.c_uint64
        L.orc_siphash.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        L.orc_pixel_hash.restype = C.c_uint32
        L.orc_pixel_hash.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_sobol_sample.restype = C.c_float
        L.orc_sobol_sample.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_bounds_intersects.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_shape_intersect.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_shape_bounds.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_reflect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_fresnel_dielectric.restype = C.c_double
        L.orc_fresnel_dielectric.argtypes = [C.c_double, C.c_double, C.c_double]
        L.orc_fresnel_conductor.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_mat_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mat_inverse.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_transformation.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_transform.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_color_from_rgb.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p]
        L.orc_color_to_rgb.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_vec_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_col_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_bounds_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_frame.argtypes = [C.c_void_p] * 5
        L.orc_partition_by.restype = C.c_uint64
        L.orc_partition_by.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int64, C.c_int64]
        L.orc_set_libm_mode.argtypes = [C.c_int]
        L.orc_set_sobol_vectors.argtypes = [C.c_void_p]
        L.orc_set_mode.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64]
        L.orc_sample_sin.restype = C.c_double
        L.orc_sample_sin.argtypes = [C.c_double]
        L.orc_sample_cos.restype = C.c_double
        L.orc_sample_cos.argtypes = [C.c_double]
        L.orc_sampling_fn.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def set_tile(tile):
    """Pixels per edge of the square tile one job of OracleScene.render covers (default 64, the reference's; 0 restores it).
    The film does not depend on it; bench.py's CPU baseline uses small tiles so that a bounded sample keeps every thread busy."""
    lib().orc_set_tile(int(tile))


def set_libm_mode(mode):
    """0: correctly rounded sin/cos in the sampling functions (default); 1: platform libm."""
    lib().orc_set_libm_mode(mode)


def set_mode(integrator='path', uniform_sampler=None, independent_sampler=False):
    """Which of the reference's integrators / samplers the oracle runs: 'path' (main's) or 'simple'
    (src/simple_integrator.rs); uniform_sampler = (nx, ny) selects UniformSampler (sampling.rs:154-194), independent_sampler
    IndependentSampler (:102-146, restated from rand 0.8.5's published algorithms: unpinned), neither the Sobol sampler."""
    nx, ny = uniform_sampler if uniform_sampler is not None else (0, 0)
    kind = 1 if uniform_sampler is not None else (2 if independent_sampler else 0)
    lib().orc_set_mode({'path': 0, 'simple': 1}[integrator], kind, nx, ny)


def set_sobol_vectors(table):
    """uint16 [64, 16, 4] bit-reversed direction vectors (sobol_burley REV_VECTORS layout), or None for the built-in table."""
    if table is None:
        lib().orc_set_sobol_vectors(None)
        return
    t = np.ascontiguousarray(table, dtype=np.uint16)
    assert t.shape == (64, 16, 4)
    lib().orc_set_sobol_vectors(t.ctypes.data)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleScene:
    """`Scene::new` + `render` + `Scene::intersect(s)` of the reference, on the CPU."""

    SAH, MEDIAN = 1, 0

    def __init__(self, scene, split_method=1):
        self._scene = scene  # keeps the desc arrays alive
        self._h = lib().orc_scene_create(C.addressof(scene.desc()), split_method)
        self.width, self.height = scene.film_bounds()
        self.build_error = lib().orc_scene_build_error(self._h)

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bvh(self):
        n, m = lib().orc_bvh_num_nodes(self._h), lib().orc_bvh_num_prim_refs(self._h)
        nodes, order = np.zeros(n, dtype=ORC_NODE_DT), np.zeros(m, dtype=np.uint32)
        lib().orc_bvh_export(self._h, nodes.ctypes.data, order.ctypes.data)
        return nodes, order

    def light_cdf(self):
        out = np.zeros(len(self._scene.lights), dtype=np.float64)
        lib().orc_scene_light_cdf(self._h, out.ctypes.data)
        return out

    def camera_matrices(self):
        a, b = np.zeros((4, 4)), np.zeros((4, 4))
        lib().orc_scene_camera(self._h, a.ctypes.data, b.ctypes.data)
        return a, b

    def trace(self, rays, any_hit=False):
        rays = _f64(rays).reshape(-1, 7)
        hits = np.zeros(len(rays), dtype=ORC_HIT_DT)
        st = OrcStats()
        lib().orc_trace(self._h, rays.ctypes.data, len(rays), 1 if any_hit else 0, hits.ctypes.data, C.byref(st))
        return hits, st.as_dict()

    def render(self, seed=0, threads=0, sample_range=None):
        s0, s1 = sample_range if sample_range is not None else (0, self._scene.num_samples)
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        st = OrcStats()
        lib().orc_render(self._h, seed, threads, s0, s1, out.ctypes.data, C.byref(st))
        return out, st.as_dict()

    def render_pixel(self, x, y, sample, seed=0):
        L = np.zeros(3)
        lib().orc_render_pixel(self._h, seed, x, y, sample, L.ctypes.data)
        return L

    def path_log(self, x, y, sample, seed=0, cap=40):
        rec = np.zeros((cap, 20))
        n = lib().orc_path_log(self._h, seed, x, y, sample, rec.ctypes.data, cap)
        return rec[:n]

    def camera_ray(self, x, y, sample, seed=0):
        r = np.zeros(7)
        lib().orc_camera_ray(self._h, seed, x, y, sample, r.ctypes.data)
        return r
