// cray_sys.rs — Rust declarations of the C ABI of libcray_hip.so, ABI version 2.
//
// Field for field what include/cray_scene_desc.h, include/cray.h, include/cray_cry.h and include/cray_io.h declare.
// A maintainer of craytracer copies this file to src/cray_sys.rs; the seam it serves is the body of `render`
// (reference src/bin/craytracer.rs:224-319), see INTEGRATION.md.
//
// This file is checked, not only shipped: tests/test_abi_layout.py parses every `#[repr(C)] pub struct` below, lays it
// out by the repr(C) rules, and compares size and every field offset with what gcc reports for the C struct named in the
// `// C: <name>` line above it (and with the ctypes / numpy mirrors of craytracer_amd/).  The struct grammar the parser
// understands is deliberately plain: one field per `pub name: type,` line; types are u8 u32 i32 u64 f32 f64 usize,
// `[T; N]`, `*const T` / `*mut T`, `Option<extern "C" fn ...>` and the names of structs defined earlier in the file.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const CRAY_ABI_VERSION: u32 = 2;

pub const CRAY_OK: c_int = 0;
pub const CRAY_ERR_INVALID: c_int = -1;
pub const CRAY_ERR_HIP: c_int = -2;
pub const CRAY_ERR_UNSUPPORTED: c_int = -3;
pub const CRAY_ERR_NO_DEVICE: c_int = -4;
pub const CRAY_ERR_BUILD: c_int = -5;

// ---------------------------------------------------------------------------------------------
// include/cray_scene_desc.h — what Scene::new receives (src/scene.rs:25-31), as plain arrays
// ---------------------------------------------------------------------------------------------

// C: cray_vec3
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayVec3 {
    pub x: f64,
    pub y: f64,
    pub z: f64,
}

// C: cray_color
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayColor {
    pub r: f64,
    pub g: f64,
    pub b: f64,
}

pub const CRAY_TEX_CONSTANT: i32 = 0;
pub const CRAY_TEX_CHECKERBOARD: i32 = 1;
pub const CRAY_TEX_IMAGE: i32 = 2;

// C: cray_texture
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayTexture {
    pub kind: i32,
    pub image: i32,
    pub a: CrayColor,
    pub b: CrayColor,
    pub scale: f64,
}

// C: cray_image
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayImage {
    pub width: u32,
    pub height: u32,
    pub offset: u64,
}

pub const CRAY_BXDF_LAMBERTIAN: i32 = 0;
pub const CRAY_BXDF_OREN_NAYAR: i32 = 1;
pub const CRAY_BXDF_FRESNEL_CONDUCTOR: i32 = 2;
pub const CRAY_BXDF_SPECULAR_BRDF: i32 = 3;
pub const CRAY_BXDF_SPECULAR_BTDF: i32 = 4;
pub const CRAY_BXDF_FRESNEL_SPECULAR: i32 = 5;
pub const CRAY_FRESNEL_DIELECTRIC: i32 = 0;
pub const CRAY_FRESNEL_CONDUCTOR: i32 = 1;

// C: cray_bxdf
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayBxdf {
    pub kind: i32,
    pub tex_a: i32,
    pub tex_b: i32,
    pub fresnel_kind: i32,
    pub eta_i: f64,
    pub eta_t: f64,
    pub c_eta_i: CrayColor,
    pub c_eta_t: CrayColor,
    pub c_k: CrayColor,
}

// C: cray_material
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayMaterial {
    pub is_bsdf: i32,
    pub n_bxdfs: i32,
    pub first_bxdf: i32,
    pub pad_: i32,
}

pub const CRAY_SHAPE_SPHERE: i32 = 0;
pub const CRAY_SHAPE_TRIANGLE: i32 = 1;
pub const CRAY_SHAPE_DISK: i32 = 2;

// C: cray_sphere_desc
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CraySphereDesc {
    pub origin: CrayVec3,
    pub radius: f64,
}

// C: cray_disk_desc
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayDiskDesc {
    pub origin: CrayVec3,
    pub rotate_x: f64,
    pub rotate_y: f64,
    pub radius: f64,
    pub inner_radius: f64,
}

// C: cray_triangle
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayTriangle {
    pub v0: CrayVec3,
    pub e1: CrayVec3,
    pub e2: CrayVec3,
    pub n0: CrayVec3,
    pub n01: CrayVec3,
    pub n02: CrayVec3,
    pub uv0: [f64; 2],
    pub uv01: [f64; 2],
    pub uv02: [f64; 2],
}

// C: cray_prim
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayPrim {
    pub shape_kind: i32,
    pub shape: u32,
    pub material: i32,
    pub light: i32,
}

pub const CRAY_LIGHT_POINT: i32 = 0;
pub const CRAY_LIGHT_DISTANT: i32 = 1;
pub const CRAY_LIGHT_INFINITE: i32 = 2;
pub const CRAY_LIGHT_AREA: i32 = 3;

// C: cray_light
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayLight {
    pub kind: i32,
    pub prim: i32,
    pub v: CrayVec3,
    pub c: CrayColor,
}

pub const CRAY_CAMERA_PERSPECTIVE: i32 = 0;
pub const CRAY_CAMERA_ORTHOGRAPHIC: i32 = 1;

// C: cray_camera_desc
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayCameraDesc {
    pub type_: i32, // C: type
    pub film_width: u32,
    pub film_height: u32,
    pub pad_: i32,
    pub origin: CrayVec3,
    pub target: CrayVec3,
    pub up: CrayVec3,
    pub fov: f64,
    pub lens_radius: f64,
    pub focal_distance: f64,
}

// C: cray_scene_desc
#[repr(C)]
pub struct CraySceneDesc {
    pub max_depth: u32,
    pub num_samples: u32,
    pub camera: CrayCameraDesc,
    pub n_spheres: u32,
    pub spheres: *const CraySphereDesc,
    pub n_disks: u32,
    pub disks: *const CrayDiskDesc,
    pub n_triangles: u32,
    pub triangles: *const CrayTriangle,
    pub n_prims: u32,
    pub prims: *const CrayPrim,
    pub n_lights: u32,
    pub lights: *const CrayLight,
    pub n_materials: u32,
    pub materials: *const CrayMaterial,
    pub n_bxdfs: u32,
    pub bxdfs: *const CrayBxdf,
    pub n_textures: u32,
    pub textures: *const CrayTexture,
    pub n_images: u32,
    pub images: *const CrayImage,
    pub image_pool_bytes: u64,
    pub image_pool: *const u8,
}

// ---------------------------------------------------------------------------------------------
// include/cray.h — the flattened Scene, render parameters, statistics, test hooks, multi-GPU
// ---------------------------------------------------------------------------------------------

// C: cray_bvh_node
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayBvhNode {
    pub bmin: [f64; 3],
    pub bmax: [f64; 3],
    pub left: u32,
    pub right: u32,
    pub first: u32,
    pub count: u32,
    pub axis: i32,
    pub is_leaf: i32,
}

// C: cray_xf_shape
#[repr(C)]
#[derive(Clone, Copy)]
pub struct CrayXfShape {
    pub m: [f64; 16],
    pub inv: [f64; 16],
    pub radius: f64,
    pub inner_radius: f64,
}

// C: cray_prim_bound
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayPrimBound {
    pub prim: u32,
    pub pad_: u32,
    pub bmin: [f64; 3],
    pub bmax: [f64; 3],
}

// C: cray_flat_scene
#[repr(C)]
pub struct CrayFlatScene {
    pub abi_version: u32,
    pub max_depth: u32,
    pub num_samples: u32,
    pub camera_type: i32,
    pub film_width: u32,
    pub film_height: u32,
    pub camera_from_raster: [f64; 16],
    pub world_from_camera: [f64; 16],
    pub lens_radius: f64,
    pub focal_distance: f64,
    pub n_nodes: u32,
    pub nodes: *const CrayBvhNode,
    pub n_prim_refs: u32,
    pub prim_refs: *const u32,
    pub n_prims: u32,
    pub prims: *const CrayPrim,
    pub n_triangles: u32,
    pub triangles: *const CrayTriangle,
    pub n_spheres: u32,
    pub spheres: *const CrayXfShape,
    pub n_disks: u32,
    pub disks: *const CrayXfShape,
    pub n_materials: u32,
    pub materials: *const CrayMaterial,
    pub n_bxdfs: u32,
    pub bxdfs: *const CrayBxdf,
    pub n_textures: u32,
    pub textures: *const CrayTexture,
    pub n_images: u32,
    pub images: *const CrayImage,
    pub image_pool_bytes: u64,
    pub image_pool: *const u8,
    pub n_lights: u32,
    pub lights: *const CrayLight,
    pub light_cdf: *const f64,
    pub first_equal_light: *const i32,
    // ABI 2: resident build (Bvh::new inside cray_scene_upload; n_nodes = n_prim_refs = 0)
    pub build_on_device: u32,
    pub n_other_bounds: u32,
    pub other_bounds: *const CrayPrimBound,
}

pub const CRAY_PRECISION_F64: u32 = 0;
pub const CRAY_PRECISION_F32_TRAVERSAL: u32 = 1;
pub const CRAY_INTEGRATOR_PATH: u32 = 0;
pub const CRAY_INTEGRATOR_SIMPLE: u32 = 1;
pub const CRAY_SAMPLER_SOBOL: u32 = 0;
pub const CRAY_SAMPLER_UNIFORM: u32 = 1;
pub const CRAY_SAMPLER_INDEPENDENT: u32 = 2;

// C: cray_render_params
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayRenderParams {
    pub seed: u64,
    pub tile_width: u32,
    pub tile_height: u32,
    pub sample_batch: u32,
    pub rank: u32,
    pub world_size: u32,
    pub sample_begin: u32,
    pub sample_end: u32,
    pub out_is_device: u32,
    pub count_traversal: u32,
    pub max_paths_in_flight: u64,
    // ABI 2
    pub integrator: u32,
    pub sampler: u32,
    pub uniform_nx: u32,
    pub uniform_ny: u32,
    pub precision: u32,
    pub pad_: u32,
}

// C: cray_stats
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayStats {
    pub paths: u64,
    pub closest_rays: u64,
    pub shadow_rays: u64,
    pub shadow_skipped: u64,
    pub closest_nodes: u64,
    pub closest_prims: u64,
    pub shadow_nodes: u64,
    pub shadow_prims: u64,
    pub closest_tri_tests: u64,
    pub shadow_tri_tests: u64,
    pub nonfinite: u64,
    pub stack_overflow: u64,
    pub seconds: f64,
    pub trace_closest_ms: f64,
    pub trace_any_ms: f64,
    pub shade_ms: f64,
    pub other_ms: f64,
    pub trace_closest_launches: u32,
    pub trace_any_launches: u32,
    pub shade_launches: u32,
    pub trace_records: u32,
    pub trace_mixed_ms: f64,
    pub trace_mixed_launches: u32,
    pub tail_split: u32,
    pub closest_hits: u64,
}

// C: cray_ray
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayRay {
    pub o: [f64; 3],
    pub d: [f64; 3],
    pub tmax: f64,
}

// C: cray_hit
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayHit {
    pub hit: i32,
    pub prim: i32,
    pub t: f64,
    pub location: [f64; 3],
    pub normal: [f64; 3],
    pub uv: [f64; 2],
}

pub const CRAY_TRACE_CLOSEST: c_int = 0;
pub const CRAY_TRACE_ANY: c_int = 1;
pub const CRAY_TRACE_CLOSEST_TIMED: c_int = 2;
pub const CRAY_TRACE_ANY_TIMED: c_int = 3;
pub const CRAY_TRACE_MIXED_TIMED: c_int = 4;

// C: cray_bvh_build_stats
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CrayBvhBuildStats {
    pub device_seconds: f64,
    pub total_seconds: f64,
    pub levels: u32,
    pub top_nodes: u32,
    pub small_subtrees: u32,
    pub leaves: u32,
}

pub const CRAY_COMM_ID_BYTES: usize = 128;

// C: cray_comm_id
#[repr(C)]
#[derive(Clone, Copy)]
pub struct CrayCommId {
    pub bytes: [u8; 128],
}

// C: cray_comm_info
#[repr(C)]
#[derive(Clone, Copy)]
pub struct CrayCommInfo {
    pub world_size: i32,
    pub rank: i32,
    pub ranks_seen: i32,
    pub rccl_version: i32,
    pub library: [u8; 256],
}

pub const CRAY_REDUCE_SUM: c_int = 0;
pub const CRAY_REDUCE_MAX: c_int = 1;
pub const CRAY_REDUCE_MIN: c_int = 2;

pub enum CrayCtx {}
pub enum CrayScene {}
pub enum CrayHostScene {}
pub enum CrayOwnedScene {}

// ---------------------------------------------------------------------------------------------
// include/cray_cry.h
// ---------------------------------------------------------------------------------------------

// C: cray_parser_error
#[repr(C)]
#[derive(Clone, Copy)]
pub struct CrayParserError {
    pub has_location: i32,
    pub line: u32,
    pub column: u32,
    pub message: [u8; 512],
}

// C: cray_token
#[repr(C)]
#[derive(Clone, Copy)]
pub struct CrayToken {
    pub kind: i32,
    pub line: u32,
    pub column: u32,
    pub number: f64,
    pub text: *const u8,
}

// C: cray_scene_overrides
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CraySceneOverrides {
    pub width: u32,
    pub height: u32,
    pub num_samples: u32,
    pub max_depth: u32,
}

pub type CrayImageLoader =
    Option<unsafe extern "C" fn(path: *const c_char, user: *mut c_void, width: *mut u32, height: *mut u32, rgb8: *mut *mut u8) -> c_int>;

#[link(name = "cray_hip")]
extern "C" {
    // ---- include/cray.h ----
    pub fn cray_ctx_create(device_id: c_int, stream: *mut c_void, out: *mut *mut CrayCtx) -> c_int;
    pub fn cray_ctx_destroy(ctx: *mut CrayCtx);
    pub fn cray_scene_upload(ctx: *mut CrayCtx, scene: *const CrayFlatScene, out: *mut *mut CrayScene) -> c_int;
    pub fn cray_scene_free(scene: *mut CrayScene);
    pub fn cray_scene_device_bytes(scene: *const CrayScene) -> u64;
    pub fn cray_scene_info(scene: *const CrayScene, film_width: *mut u32, film_height: *mut u32, num_samples: *mut u32, max_depth: *mut u32);
    pub fn cray_set_sobol_vectors(rev_vectors: *const u16) -> c_int;
    pub fn cray_render(ctx: *mut CrayCtx, scene: *mut CrayScene, params: *const CrayRenderParams, out_rgb: *mut f32, stats: *mut CrayStats) -> c_int;
    pub fn cray_render_params_default(params: *mut CrayRenderParams);
    pub fn cray_render_samples(ctx: *mut CrayCtx, scene: *mut CrayScene, params: *const CrayRenderParams, out_l: *mut f64) -> c_int;
    pub fn cray_trace(ctx: *mut CrayCtx, scene: *mut CrayScene, rays: *const CrayRay, n: usize, hits: *mut CrayHit, mode: c_int, stats: *mut CrayStats) -> c_int;
    pub fn cray_bvh_build_sah(ctx: *mut CrayCtx, prim_bounds: *const f64, n: u32, out_nodes: *mut CrayBvhNode, node_capacity: u32,
                              out_n_nodes: *mut u32, out_prim_refs: *mut u32, stats: *mut CrayBvhBuildStats) -> c_int;
    pub fn cray_scene_build_stats(scene: *const CrayScene, out: *mut CrayBvhBuildStats);
    pub fn cray_comm_unique_id(out: *mut CrayCommId) -> c_int;
    pub fn cray_comm_init(ctx: *mut CrayCtx, id: *const CrayCommId, rank: c_int, world_size: c_int) -> c_int;
    pub fn cray_comm_rank(ctx: *const CrayCtx) -> c_int;
    pub fn cray_comm_world_size(ctx: *const CrayCtx) -> c_int;
    pub fn cray_comm_barrier(ctx: *mut CrayCtx) -> c_int;
    pub fn cray_comm_allreduce_f64(ctx: *mut CrayCtx, values: *mut f64, n: c_int, op: c_int) -> c_int;
    pub fn cray_comm_describe(ctx: *mut CrayCtx, out: *mut CrayCommInfo) -> c_int;
    pub fn cray_scene_broadcast(ctx: *mut CrayCtx, scene_on_root: *mut CrayScene, root: c_int, out: *mut *mut CrayScene) -> c_int;
    pub fn cray_render_gather(ctx: *mut CrayCtx, scene: *mut CrayScene, params: *const CrayRenderParams, out_rgb: *mut f32, stats: *mut CrayStats) -> c_int;
    pub fn cray_film_gather(ctx: *mut CrayCtx, width: u32, height: u32, tile_width: u32, tile_height: u32,
                            local_film_device: *const f32, out_rgb: *mut f32, out_is_device: c_int) -> c_int;
    pub fn cray_film_pack(ctx: *mut CrayCtx, width: u32, height: u32, tile_width: u32, tile_height: u32, rank: u32, world_size: u32,
                          film: *const f32, packed: *mut f32, n_pixels: *mut u64) -> c_int;
    pub fn cray_film_unpack(ctx: *mut CrayCtx, width: u32, height: u32, tile_width: u32, tile_height: u32, world_size: u32,
                            gathered: *const f32, out: *mut f32) -> c_int;
    pub fn cray_tile_pixels(width: u32, height: u32, tile_width: u32, tile_height: u32, rank: u32, world_size: u32,
                            out: *mut u32, capacity: u64, n_pixels: *mut u64) -> c_int;
    pub fn cray_measure_stream_read(ctx: *mut CrayCtx, bytes: u64, repeats: c_int, gb_per_s: *mut f64) -> c_int;
    pub fn cray_ctx_pool_info(ctx: *const CrayCtx, pool_bytes: *mut u64, paths: *mut u64);
    pub fn cray_scene_records_info(scene: *const CrayScene, chosen: *mut i32, probe_ms: *mut f64, probe_kernel_ms: *mut f64);
    pub fn cray_last_error() -> *const c_char;

    // ---- include/cray_host.h: Scene::new in C++ for hosts that are not craytracer itself ----
    pub fn cray_host_scene_new(desc: *const CraySceneDesc, split_method: c_int, out: *mut *mut CrayHostScene) -> c_int;
    pub fn cray_host_scene_new_on(desc: *const CraySceneDesc, split_method: c_int, bvh_ctx: *mut CrayCtx, out: *mut *mut CrayHostScene) -> c_int;
    pub fn cray_host_scene_new_resident(desc: *const CraySceneDesc, out: *mut *mut CrayHostScene) -> c_int;
    pub fn cray_host_scene_flat(scene: *const CrayHostScene) -> *const CrayFlatScene;
    pub fn cray_host_scene_free(scene: *mut CrayHostScene);

    // ---- include/cray_cry.h ----
    pub fn cray_cry_tokenize(input: *const c_char, tokens: *mut *mut CrayToken, n_tokens: *mut usize, err: *mut CrayParserError) -> c_int;
    pub fn cray_cry_free_tokens(tokens: *mut CrayToken, n_tokens: usize);
    pub fn cray_cry_parse_scene(input: *const c_char, base_dir: *const c_char, loader: CrayImageLoader, loader_user: *mut c_void,
                                overrides: *const CraySceneOverrides, out: *mut *mut CrayOwnedScene, err: *mut CrayParserError) -> c_int;
    pub fn cray_owned_scene_desc(scene: *const CrayOwnedScene) -> *const CraySceneDesc;
    pub fn cray_owned_scene_warnings(scene: *const CrayOwnedScene) -> u32;
    pub fn cray_owned_scene_free(scene: *mut CrayOwnedScene);

    // ---- include/cray_io.h ----
    pub fn cray_write_exr(path: *const c_char, width: u32, height: u32, rgb: *const f32) -> c_int;
    pub fn cray_load_image(path: *const c_char, width: *mut u32, height: *mut u32, rgb8: *mut *mut u8) -> c_int;
    pub fn cray_free_image(rgb8: *mut u8);
}
