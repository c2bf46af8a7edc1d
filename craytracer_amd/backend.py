"""ctypes binding of the C ABI in include/cray.h and include/cray_host.h.

The product path: scene description -> host `Scene::new` mirror (BVH, light CDF, camera)
-> cray_scene_upload -> cray_render / cray_trace on the GPU.  There is no CPU fallback: if the
HIP library is missing or no GPU is visible, these calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))


class CrayError(RuntimeError):
    pass


class RenderParams(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('tile_width', C.c_uint32), ('tile_height', C.c_uint32),
                ('sample_batch', C.c_uint32), ('rank', C.c_uint32), ('world_size', C.c_uint32),
                ('sample_begin', C.c_uint32), ('sample_end', C.c_uint32), ('out_is_device', C.c_uint32),
                ('count_traversal', C.c_uint32), ('max_paths_in_flight', C.c_uint64),
                ('integrator', C.c_uint32), ('sampler', C.c_uint32), ('uniform_nx', C.c_uint32), ('uniform_ny', C.c_uint32),
                ('precision', C.c_uint32), ('pad_', C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ('paths', 'closest_rays', 'shadow_rays', 'shadow_skipped', 'closest_nodes', 'closest_prims', 'shadow_nodes',
                 'shadow_prims', 'closest_tri_tests', 'shadow_tri_tests', 'nonfinite', 'stack_overflow')] + \
               [(n, C.c_double) for n in ('seconds', 'trace_closest_ms', 'trace_any_ms', 'shade_ms', 'other_ms')] + \
               [(n, C.c_uint32) for n in ('trace_closest_launches', 'trace_any_launches', 'shade_launches', 'trace_records')] + \
               [('trace_mixed_ms', C.c_double), ('trace_mixed_launches', C.c_uint32), ('tail_split', C.c_uint32),
                ('closest_hits', C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n not in ('pad_', 'pad2_')}


RAY_DT = np.dtype([('o', '<f8', 3), ('d', '<f8', 3), ('tmax', '<f8')], align=True)
HIT_DT = np.dtype([('hit', '<i4'), ('prim', '<i4'), ('t', '<f8'), ('location', '<f8', 3), ('normal', '<f8', 3),
                   ('uv', '<f8', 2)], align=True)
BVH_NODE_DT = np.dtype([('bmin', '<f8', 3), ('bmax', '<f8', 3), ('left', '<u4'), ('right', '<u4'), ('first', '<u4'),
                        ('count', '<u4'), ('axis', '<i4'), ('is_leaf', '<i4')], align=True)
assert RAY_DT.itemsize == 56 and HIT_DT.itemsize == 80 and BVH_NODE_DT.itemsize == 72


class FlatScene(C.Structure):
    _fields_ = [('abi_version', C.c_uint32), ('max_depth', C.c_uint32), ('num_samples', C.c_uint32),
                ('camera_type', C.c_int32), ('film_width', C.c_uint32), ('film_height', C.c_uint32),
                ('camera_from_raster', C.c_double * 16), ('world_from_camera', C.c_double * 16),
                ('lens_radius', C.c_double), ('focal_distance', C.c_double),
                ('n_nodes', C.c_uint32), ('nodes', C.c_void_p),
                ('n_prim_refs', C.c_uint32), ('prim_refs', C.c_void_p),
                ('n_prims', C.c_uint32), ('prims', C.c_void_p),
                ('n_triangles', C.c_uint32), ('triangles', C.c_void_p),
                ('n_spheres', C.c_uint32), ('spheres', C.c_void_p),
                ('n_disks', C.c_uint32), ('disks', C.c_void_p),
                ('n_materials', C.c_uint32), ('materials', C.c_void_p),
                ('n_bxdfs', C.c_uint32), ('bxdfs', C.c_void_p),
                ('n_textures', C.c_uint32), ('textures', C.c_void_p),
                ('n_images', C.c_uint32), ('images', C.c_void_p),
                ('image_pool_bytes', C.c_uint64), ('image_pool', C.c_void_p),
                ('n_lights', C.c_uint32), ('lights', C.c_void_p),
                ('light_cdf', C.c_void_p), ('first_equal_light', C.c_void_p),
                ('build_on_device', C.c_uint32), ('n_other_bounds', C.c_uint32), ('other_bounds', C.c_void_p)]


#: every symbol include/cray.h and include/cray_host.h declare
class CommInfo(C.Structure):
    _fields_ = [('world_size', C.c_int32), ('rank', C.c_int32), ('ranks_seen', C.c_int32), ('rccl_version', C.c_int32), ('library', C.c_char * 256)]


class BvhBuildStats(C.Structure):
    """cray_bvh_build_stats (include/cray.h)."""
    _fields_ = [('device_seconds', C.c_double), ('total_seconds', C.c_double), ('levels', C.c_uint32),
                ('top_nodes', C.c_uint32), ('small_subtrees', C.c_uint32), ('leaves', C.c_uint32)]


ABI_SYMBOLS = ['cray_ctx_create', 'cray_ctx_destroy', 'cray_scene_upload', 'cray_scene_free',
               'cray_scene_device_bytes', 'cray_render', 'cray_render_params_default', 'cray_render_samples',
               'cray_trace', 'cray_last_error', 'cray_host_scene_new', 'cray_host_scene_flat',
               'cray_host_scene_build_seconds', 'cray_host_scene_free', 'cray_host_scene_new_on',
               'cray_host_scene_bvh_seconds', 'cray_bvh_build_sah', 'cray_write_exr', 'cray_read_exr', 'cray_host_sincos', 'cray_host_div_fast_mismatches', 'cray_host_child_key_mismatches', 'cray_host_hyb_key_violations', 'cray_host_tri_cull_violations', 'cray_host_sincos_fast_check', 'cray_host_chacha_block', 'cray_host_independent_draws',
               'cray_cry_tokenize', 'cray_cry_free_tokens', 'cray_cry_parse_value', 'cray_cry_free_string',
               'cray_cry_parse_scene', 'cray_owned_scene_desc', 'cray_owned_scene_warnings', 'cray_owned_scene_free',
               'cray_scene_info', 'cray_comm_unique_id', 'cray_comm_init', 'cray_comm_rank', 'cray_comm_world_size',
               'cray_comm_barrier', 'cray_comm_allreduce_f64', 'cray_comm_describe', 'cray_scene_broadcast', 'cray_render_gather',
               'cray_film_gather', 'cray_film_pack', 'cray_film_unpack', 'cray_measure_stream_read', 'cray_ctx_pool_info', 'cray_scene_records_info', 'cray_load_image', 'cray_free_image', 'cray_default_image_loader', 'cray_set_sobol_vectors',
               'cray_host_scene_new_resident', 'cray_scene_build_stats', 'cray_tile_pixels', 'cray_preview_checkerboard', 'cray_preview_pixels']

_lib = None
#: how the loaded library came to be: 'shipped' (the .so in the tree was current), 'rebuilt' (sources were newer, hipcc ran),
#: 'STALE' (sources are newer than the .so and the rebuild failed: the kernels do not match the sources)
BUILD_MODE = None


def lib():
    """Load libcray_hip.so (building it if the sources are newer). Raises if it cannot be loaded.
    In a process that also uses torch, import torch BEFORE the first call: torch brings its own HIP runtime, and a library
    loaded first binds the system one — two runtimes in one process, the second of which sees no device."""
    global _lib, BUILD_MODE
    if _lib is not None:
        return _lib
    so = _build.SO
    BUILD_MODE = 'shipped'
    if os.environ.get('CRAY_LIB'):   # A/B experiments: load this build of the library instead (never rebuilt here)
        so = os.environ['CRAY_LIB']
        BUILD_MODE = 'override:' + so
    elif _build.stale():
        try:
            _build.build()
            BUILD_MODE = 'rebuilt'
        except Exception as e:  # no hipcc on this box: use the shipped .so if there is one — loudly
            if not os.path.exists(so):
                raise CrayError('libcray_hip.so is missing and could not be built: %s' % e)
            if os.environ.get('CRAY_ALLOW_STALE', '0') != '1':
                raise CrayError('libcray_hip.so is older than its sources and the rebuild failed (%s); '
                                'set CRAY_ALLOW_STALE=1 to run the stale library anyway' % e)
            BUILD_MODE = 'STALE'
            import sys
            print('WARNING: libcray_hip.so is OLDER than its sources and could not be rebuilt (%s): running stale kernels' % e,
                  file=sys.stderr, flush=True)
    try:
        L = C.CDLL(so)
    except OSError as e:
        raise CrayError('cannot load %s: %s (the HIP backend has no CPU fallback)' % (so, e))
    L.cray_last_error.restype = C.c_char_p
    L.cray_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.cray_ctx_destroy.argtypes = [C.c_void_p]
    L.cray_scene_upload.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.cray_scene_free.argtypes = [C.c_void_p]
    L.cray_scene_device_bytes.restype = C.c_uint64
    L.cray_scene_device_bytes.argtypes = [C.c_void_p]
    L.cray_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
    L.cray_render_params_default.argtypes = [C.POINTER(RenderParams)]
    L.cray_render_samples.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RenderParams), C.c_void_p]
    L.cray_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(Stats)]
    L.cray_host_scene_new.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.cray_host_scene_new_on.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.cray_host_scene_new_resident.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.cray_scene_build_stats.restype = None
    L.cray_scene_build_stats.argtypes = [C.c_void_p, C.POINTER(BvhBuildStats)]
    L.cray_host_scene_bvh_seconds.restype = C.c_double
    L.cray_host_scene_bvh_seconds.argtypes = [C.c_void_p, C.POINTER(BvhBuildStats)]
    L.cray_bvh_build_sah.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                     C.c_void_p, C.POINTER(BvhBuildStats)]
    L.cray_write_exr.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.cray_read_exr.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64]
    L.cray_scene_info.restype = None
    L.cray_scene_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4
    L.cray_comm_unique_id.argtypes = [C.c_void_p]
    L.cray_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.cray_comm_rank.argtypes = [C.c_void_p]
    L.cray_comm_world_size.argtypes = [C.c_void_p]
    L.cray_comm_barrier.argtypes = [C.c_void_p]
    L.cray_comm_allreduce_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.cray_scene_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.cray_render_gather.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
    L.cray_film_gather.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    L.cray_film_pack.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.POINTER(C.c_uint64)]
    L.cray_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]
    L.cray_free_image.restype = None
    L.cray_free_image.argtypes = [C.c_void_p]
    L.cray_set_sobol_vectors.argtypes = [C.c_void_p]
    L.cray_preview_checkerboard.restype = None
    L.cray_preview_checkerboard.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.cray_preview_pixels.restype = None
    L.cray_preview_pixels.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_void_p]
    L.cray_measure_stream_read.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    L.cray_film_unpack.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.cray_host_scene_flat.restype = C.POINTER(FlatScene)
    L.cray_host_scene_flat.argtypes = [C.c_void_p]
    L.cray_host_scene_build_seconds.restype = C.c_double
    L.cray_host_scene_build_seconds.argtypes = [C.c_void_p]
    L.cray_host_scene_free.argtypes = [C.c_void_p]
    L.cray_host_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.cray_host_div_fast_mismatches.restype = C.c_uint64
    L.cray_host_div_fast_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.cray_host_child_key_mismatches.restype = C.c_uint64
    L.cray_host_child_key_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    if hasattr(L, 'cray_host_chacha_block'):
        L.cray_host_chacha_block.restype = None
        L.cray_host_chacha_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.cray_host_independent_draws.restype = None
        L.cray_host_independent_draws.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
    if hasattr(L, 'cray_host_sincos_fast_check'):
        L.cray_host_sincos_fast_check.restype = C.c_uint64
        L.cray_host_sincos_fast_check.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    if hasattr(L, 'cray_host_hyb_key_violations'):   # an older experimental build (CRAY_LIB) may lack it
        L.cray_host_hyb_key_violations.restype = C.c_uint64
        L.cray_host_hyb_key_violations.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    if hasattr(L, 'cray_host_tri_cull_violations'):
        L.cray_host_tri_cull_violations.restype = C.c_uint64
        L.cray_host_tri_cull_violations.argtypes = [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p]
    _lib = L
    return L


def _check(code, what):
    if code != 0:
        raise CrayError('%s failed (%d): %s' % (what, code, lib().cray_last_error().decode()))


def tile_pixels(width, height, rank, world_size, tile=(64, 64)):
    """cray_tile_pixels: the pixels (y*W + x) rank `rank` of `world_size` renders, in pack / send order.  Host only: no GPU."""
    n = C.c_uint64(0)
    L = lib()
    L.cray_tile_pixels.argtypes = [C.c_uint32] * 6 + [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    _check(L.cray_tile_pixels(width, height, tile[0], tile[1], rank, world_size, None, 0, C.byref(n)), 'cray_tile_pixels')
    out = np.zeros(n.value, dtype=np.uint32)
    _check(L.cray_tile_pixels(width, height, tile[0], tile[1], rank, world_size, out.ctypes.data, len(out), C.byref(n)), 'cray_tile_pixels')
    return out


class HostScene:
    """`Scene::new` on the host (src/scene.rs:25-53): BVH (SAH), LightSampler, Camera matrices."""

    SAH, MEDIAN = 1, 0

    def __init__(self, scene, split_method=1, bvh_ctx=None, resident=False):
        """bvh_ctx: a Context whose GPU runs Bvh::new (cray_bvh_build_sah, same tree); None = host build.
        resident=True: no tree here at all — Context.upload builds it on the GPU and keeps it there (the fastest path from a
        description to a renderable scene; bvh() is then unavailable)."""
        self.scene = scene  # keeps the description arrays alive (the flat view borrows them)
        h = C.c_void_p()
        if resident:
            _check(lib().cray_host_scene_new_resident(C.addressof(scene.desc()), C.byref(h)), 'cray_host_scene_new_resident')
        elif bvh_ctx is None:
            _check(lib().cray_host_scene_new(C.addressof(scene.desc()), split_method, C.byref(h)), 'cray_host_scene_new')
        else:
            _check(lib().cray_host_scene_new_on(C.addressof(scene.desc()), split_method, bvh_ctx._h, C.byref(h)),
                   'cray_host_scene_new_on')
        self._h = h
        self.flat = lib().cray_host_scene_flat(h).contents
        self.build_seconds = lib().cray_host_scene_build_seconds(h)
        g = BvhBuildStats()
        self.bvh_seconds = lib().cray_host_scene_bvh_seconds(h, C.byref(g))
        self.gpu_build = {k: getattr(g, k) for k, _ in BvhBuildStats._fields_}

    def bvh(self):
        n, m = self.flat.n_nodes, self.flat.n_prim_refs
        nodes = np.ctypeslib.as_array(C.cast(self.flat.nodes, C.POINTER(C.c_uint8)), shape=(n * 72,)).view(BVH_NODE_DT)
        refs = np.ctypeslib.as_array(C.cast(self.flat.prim_refs, C.POINTER(C.c_uint32)), shape=(m,))
        return nodes.copy(), refs.copy()

    def light_cdf(self):
        return np.ctypeslib.as_array(C.cast(self.flat.light_cdf, C.POINTER(C.c_double)),
                                     shape=(self.flat.n_lights,)).copy()

    def first_equal_light(self):
        return np.ctypeslib.as_array(C.cast(self.flat.first_equal_light, C.POINTER(C.c_int32)),
                                     shape=(self.flat.n_lights,)).copy()

    def camera_matrices(self):
        return (np.array(self.flat.camera_from_raster[:]).reshape(4, 4),
                np.array(self.flat.world_from_camera[:]).reshape(4, 4))

    def close(self):
        if self._h:
            lib().cray_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU (one process per GPU)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        _check(lib().cray_ctx_create(device, stream, C.byref(h)), 'cray_ctx_create')
        self._h = h
        self.device = device

    def upload(self, host_scene):
        return DeviceScene(self, host_scene)

    def build_bvh(self, prim_bounds):
        """Bvh::new(.., SplitMethod::SAH) on this GPU: prim_bounds [n, 6] f64 -> (nodes, prim_refs, stats)."""
        pb = np.ascontiguousarray(prim_bounds, dtype=np.float64).reshape(-1, 6)
        n = len(pb)
        nodes = np.zeros(max(1, 2 * n - 1), dtype=BVH_NODE_DT)
        refs = np.zeros(n, dtype=np.uint32)
        n_nodes = C.c_uint32(0)
        st = BvhBuildStats()
        _check(lib().cray_bvh_build_sah(self._h, pb.ctypes.data, n, nodes.ctypes.data, len(nodes), C.byref(n_nodes),
                                        refs.ctypes.data, C.byref(st)), 'cray_bvh_build_sah')
        return nodes[:n_nodes.value], refs, {k: getattr(st, k) for k, _ in BvhBuildStats._fields_}

    # ---- multi-GPU (include/cray.h "multi-GPU"): one Context per rank, RCCL underneath
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId as 128 bytes: create on one rank, hand to every rank by any host channel."""
        buf = C.create_string_buffer(128)
        _check(lib().cray_comm_unique_id(buf), 'cray_comm_unique_id')
        return buf.raw

    def comm_init(self, comm_id, rank, world_size):
        assert len(comm_id) == 128
        _check(lib().cray_comm_init(self._h, C.create_string_buffer(comm_id, 128), rank, world_size), 'cray_comm_init')
        self.rank, self.world_size = rank, world_size

    def comm_rank(self):
        return lib().cray_comm_rank(self._h)

    def comm_world_size(self):
        return lib().cray_comm_world_size(self._h)

    def barrier(self):
        _check(lib().cray_comm_barrier(self._h), 'cray_comm_barrier')

    def allreduce(self, values, op='sum'):
        """In-place all-reduce of up to 64 host doubles; returns the reduced list."""
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        _check(lib().cray_comm_allreduce_f64(self._h, v.ctypes.data, len(v), {'sum': 0, 'max': 1, 'min': 2}[op]),
               'cray_comm_allreduce_f64')
        return v

    def comm_describe(self):
        """cray_comm_describe: world size and rank of the communicator, the ranks an all-reduce of 1.0 actually counted, the
        collective library's version and the file it was loaded from.  Collective when there is a communicator."""
        info = CommInfo()
        _check(lib().cray_comm_describe(self._h, C.byref(info)), 'cray_comm_describe')
        path = info.library.decode(errors='replace')
        return {'world': info.world_size, 'rank': info.rank, 'ranks_seen': info.ranks_seen, 'rccl_version': info.rccl_version,
                'library': path, 'transport': 'rccl' if os.path.basename(path).startswith('librccl') else ('none' if not path else 'stand-in')}

    def broadcast_scene(self, device_scene=None, root=0):
        """cray_scene_broadcast: pass the uploaded DeviceScene on `root`, None elsewhere; returns a DeviceScene."""
        out = C.c_void_p()
        _check(lib().cray_scene_broadcast(self._h, device_scene._h if device_scene is not None else None, root, C.byref(out)),
               'cray_scene_broadcast')
        if device_scene is not None:
            return device_scene
        return DeviceScene(self, None, handle=out)

    def pool_info(self):
        """(bytes of the path-state pool this context holds, paths it has room for): cray_ctx_pool_info"""
        b, n = C.c_uint64(0), C.c_uint64(0)
        L = lib()
        L.cray_ctx_pool_info.restype = None
        L.cray_ctx_pool_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.cray_ctx_pool_info(self._h, C.byref(b), C.byref(n))
        return b.value, n.value

    def measure_stream_read(self, nbytes=4 << 30, repeats=5):
        """GB/s of a plain 16-B/lane streaming read on this GPU (cray_measure_stream_read)."""
        g = C.c_double(0.0)
        _check(lib().cray_measure_stream_read(self._h, nbytes, repeats, C.byref(g)), 'cray_measure_stream_read')
        return g.value

    def film_pack(self, film, rank, world_size, tile=(64, 64)):
        """Test hook: the pixels of `rank`'s tiles of a host film [H, W, 3], packed [n, 3]."""
        f = np.ascontiguousarray(film, dtype=np.float32)
        h, w = f.shape[:2]
        packed = np.zeros((h * w, 3), dtype=np.float32)
        n = C.c_uint64(0)
        _check(lib().cray_film_pack(self._h, w, h, tile[0], tile[1], rank, world_size, f.ctypes.data, packed.ctypes.data, C.byref(n)),
               'cray_film_pack')
        return packed[:n.value].copy()

    def film_unpack(self, gathered, width, height, world_size, tile=(64, 64)):
        """Test hook: the rank-ordered concatenation of packed tiles [W*H, 3] -> film [H, W, 3]."""
        g = np.ascontiguousarray(gathered, dtype=np.float32).reshape(-1, 3)
        assert len(g) == width * height
        out = np.zeros((height, width, 3), dtype=np.float32)
        _check(lib().cray_film_unpack(self._h, width, height, tile[0], tile[1], world_size, g.ctypes.data, out.ctypes.data),
               'cray_film_unpack')
        return out

    def film_gather(self, local_film_ptr, width, height, out=None, out_device_ptr=None, tile=(64, 64)):
        """cray_film_gather: local_film_ptr = this rank's W*H*3 device film; on rank 0 fills `out` (numpy) or out_device_ptr."""
        dst, is_dev = None, 0
        if out_device_ptr is not None:
            dst, is_dev = C.c_void_p(out_device_ptr), 1
        elif out is not None:
            dst = C.c_void_p(out.ctypes.data)
        _check(lib().cray_film_gather(self._h, width, height, tile[0], tile[1], C.c_void_p(local_film_ptr), dst, is_dev), 'cray_film_gather')
        return out

    def close(self):
        if self._h:
            lib().cray_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceScene:
    """A scene resident in HBM; `render` replaces craytracer.rs:224 `render`, `trace` replaces
    `Scene::intersect` / `Scene::intersects`."""

    def __init__(self, ctx, host_scene, handle=None):
        self.ctx, self.host = ctx, host_scene
        if handle is None:
            handle = C.c_void_p()
            _check(lib().cray_scene_upload(ctx._h, C.addressof(host_scene.flat), C.byref(handle)), 'cray_scene_upload')
        self._h = handle  # else: received by cray_scene_broadcast
        w, h, ns, depth = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().cray_scene_info(self._h, C.byref(w), C.byref(h), C.byref(ns), C.byref(depth))
        self.width, self.height, self.num_samples, self.max_depth = w.value, h.value, ns.value, depth.value
        self.device_bytes = lib().cray_scene_device_bytes(self._h)
        g = BvhBuildStats()
        lib().cray_scene_build_stats(self._h, C.byref(g))
        self.build_stats = {k: getattr(g, k) for k, _ in BvhBuildStats._fields_}   # resident build: Bvh::new inside the upload

    #: selectable alternatives of the reference, applied to every later render of this DeviceScene:
    #: integrator 'path' | 'simple' (src/simple_integrator.rs), uniform_sampler None | (nx, ny) (sampling.rs:154-194)
    integrator = 'path'
    uniform_sampler = None
    #: True: IndependentSampler (sampling.rs:102-146; restated from rand 0.8.5's published algorithms, not pinned against the crate)
    independent_sampler = False
    #: 'f64' (the reference's arithmetic) or 'f32' (fast mode: f32 traversal, NOT bit-exact; cray_render_params.precision)
    precision = 'f64'
    #: (tile_width, tile_height) of the shard and of the pixel order inside a pass; None = the reference's 64 x 64 (craytracer.rs:232-233).
    #: The film does not depend on it; 32 x 32 balances the ranks of a multi-GPU frame better.
    tile = None

    def records_info(self):
        """cray_scene_records_info: {'chosen': (bounce 0, others) with -1 = nothing chosen, 'probe_ms', 'probe_kernel_ms': {...}}"""
        L = lib()
        L.cray_scene_records_info.restype = None
        L.cray_scene_records_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        ch, ms, k = (C.c_int32 * 2)(), C.c_double(0.0), (C.c_double * 4)()
        L.cray_scene_records_info(self._h, ch, C.byref(ms), k)
        return {'chosen': (ch[0], ch[1]), 'probe_ms': ms.value,
                'probe_kernel_ms': {'f64': {'bounce0': k[0], 'other_launches': k[1]}, 'f32_culling': {'bounce0': k[2], 'other_launches': k[3]}}}

    def params(self, seed=0, rank=0, world_size=1, sample_range=None, count_traversal=False, max_paths_in_flight=0):
        p = RenderParams()
        lib().cray_render_params_default(C.byref(p))
        p.integrator = {'path': 0, 'simple': 1}[self.integrator]
        p.precision = {'f64': 0, 'f32': 1}[self.precision]
        if self.uniform_sampler is not None:
            p.sampler, (p.uniform_nx, p.uniform_ny) = 1, self.uniform_sampler
        elif self.independent_sampler:
            p.sampler = 2
        p.seed, p.rank, p.world_size = seed, rank, world_size
        if self.tile is not None:
            p.tile_width, p.tile_height = self.tile
        if sample_range is not None:
            p.sample_begin, p.sample_end = sample_range
        p.count_traversal = int(count_traversal)  # True/1: every reference query; 2: only the traversed ones
        p.max_paths_in_flight = max_paths_in_flight
        return p

    def render(self, seed=0, rank=0, world_size=1, sample_range=None, count_traversal=False, max_paths_in_flight=0,
               out_device_ptr=None, out=None):
        """Returns (film[h, w, 3] float32 or None when writing to out_device_ptr, stats dict); `out`: a C-contiguous
        float32 [h, w, 3] host array to fill (e.g. pinned memory) instead of a new one."""
        p = self.params(seed, rank, world_size, sample_range, count_traversal, max_paths_in_flight)
        st = Stats()
        if out_device_ptr is not None:
            p.out_is_device = 1
            _check(lib().cray_render(self.ctx._h, self._h, C.byref(p), C.c_void_p(out_device_ptr), C.byref(st)), 'cray_render')
            return None, st.as_dict()
        if out is None:
            out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        assert out.dtype == np.float32 and out.shape == (self.height, self.width, 3) and out.flags['C_CONTIGUOUS']
        _check(lib().cray_render(self.ctx._h, self._h, C.byref(p), out.ctypes.data, C.byref(st)), 'cray_render')
        return out, st.as_dict()

    def render_gather(self, seed=0, sample_range=None, max_paths_in_flight=0, out=None, out_device_ptr=None):
        """cray_render_gather: this rank's tiles (rank / world of the context's communicator), then the RCCL gather.
        Rank 0 gets the film (numpy [H, W, 3], or written to out_device_ptr); other ranks get None. Returns (film, stats)."""
        p = self.params(seed, 0, 1, sample_range, False, max_paths_in_flight)
        st = Stats()
        root = self.ctx.comm_rank() == 0
        dst = None
        if out_device_ptr is not None:
            p.out_is_device = 1
            dst = C.c_void_p(out_device_ptr) if root else None
        elif root:
            if out is None:
                out = np.zeros((self.height, self.width, 3), dtype=np.float32)
            dst = C.c_void_p(out.ctypes.data)
        _check(lib().cray_render_gather(self.ctx._h, self._h, C.byref(p), dst, C.byref(st)), 'cray_render_gather')
        return (out if root and out_device_ptr is None else None), st.as_dict()

    def render_samples(self, sample_range, seed=0):
        """Per-path radiance L[h, w, n, 3] (f64) of samples [a, b)."""
        p = self.params(seed, sample_range=sample_range)
        n = sample_range[1] - sample_range[0]
        out = np.zeros((self.height, self.width, n, 3), dtype=np.float64)
        _check(lib().cray_render_samples(self.ctx._h, self._h, C.byref(p), out.ctypes.data), 'cray_render_samples')
        return out

    def trace(self, rays, any_hit=False, timed=False):
        """Scene::intersect / Scene::intersects for a batch of rays [n, 7] (o, d, tmax).  timed=True runs the instantiation
        the frame loop times (no traversal counters) instead of the instrumented one."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        hits = np.zeros(len(rays), dtype=HIT_DT)
        st = Stats()
        _check(lib().cray_trace(self.ctx._h, self._h, rays.ctypes.data, len(rays), hits.ctypes.data,
                                (1 if any_hit else 0) + (2 if timed else 0), C.byref(st)), 'cray_trace')
        return hits, st.as_dict()

    def trace_mixed(self, rays):
        """k_trace_mixed as the frame loop launches it: every ray as a shadow ray (its tmax) and as a path segment (tmax = inf)
        in ONE launch -> (any-hit answers [n], closest-hit records [n])."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        hits = np.zeros(2 * len(rays), dtype=HIT_DT)
        _check(lib().cray_trace(self.ctx._h, self._h, rays.ctypes.data, len(rays), hits.ctypes.data, 4, None), 'cray_trace')
        return hits[:len(rays)], hits[len(rays):]

    def close(self):
        if self._h:
            lib().cray_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_exr(path, film):
    """Film [H, W, 3] f32 -> OpenEXR (cray_write_exr; the reference's `image_buffer.save`, craytracer.rs:366-370)."""
    img = np.ascontiguousarray(film, dtype=np.float32)
    assert img.ndim == 3 and img.shape[2] == 3
    _check(lib().cray_write_exr(os.fsencode(path), img.shape[1], img.shape[0], img.ctypes.data), 'cray_write_exr')


def set_sobol_vectors(table):
    """cray_set_sobol_vectors: uint16 [64, 16, 4] (sobol_burley REV_VECTORS layout) for scenes uploaded afterwards; None = built-in."""
    if table is None:
        _check(lib().cray_set_sobol_vectors(None), 'cray_set_sobol_vectors')
        return
    t = np.ascontiguousarray(table, dtype=np.uint16)
    assert t.shape == (64, 16, 4)
    _check(lib().cray_set_sobol_vectors(t.ctypes.data), 'cray_set_sobol_vectors')


def preview_checkerboard(width, height, tile=(64, 64)):
    out = np.zeros((height, width), dtype=np.uint32)
    lib().cray_preview_checkerboard(width, height, tile[0], tile[1], out.ctypes.data)
    return out


def preview_pixels(film, divisor=1.0):
    """Film [H, W, 3] f32 -> the preview's 0x00RRGGBB pixels (craytracer.rs:190-205, Color::to_rgb)."""
    f = np.ascontiguousarray(film, dtype=np.float32)
    out = np.zeros(f.shape[:2], dtype=np.uint32)
    lib().cray_preview_pixels(f.ctypes.data, out.size, divisor, out.ctypes.data)
    return out


def load_image(path):
    """cray_load_image: PNM / JPEG -> uint8 [H, W, 3] (the library's own decoder, no Pillow)."""
    w, h, px = C.c_uint32(0), C.c_uint32(0), C.c_void_p()
    _check(lib().cray_load_image(os.fsencode(path), C.byref(w), C.byref(h), C.byref(px)), 'cray_load_image')
    try:
        return np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 3)).copy()
    finally:
        lib().cray_free_image(px)


def read_exr(path):
    w, h = C.c_uint32(0), C.c_uint32(0)
    _check(lib().cray_read_exr(os.fsencode(path), C.byref(w), C.byref(h), None, 0), 'cray_read_exr')
    img = np.zeros((h.value, w.value, 3), dtype=np.float32)
    _check(lib().cray_read_exr(os.fsencode(path), C.byref(w), C.byref(h), img.ctypes.data, img.size), 'cray_read_exr')
    return img
