/*
 * cray.h — C ABI of the MI355X render backend for craytracer's hot path.
 *
 * The reference has no FFI; the seam this library replaces is the body of
 *     fn render(scene: &Scene, sampler: SobolSampler, ..) -> Vec<f32>
 * (reference src/bin/craytracer.rs:224-319, output contract :253-259: W*H*3 f32,
 * row-major, y down, RGB interleaved, already divided by num_samples), plus the
 * finer seams Scene::intersect / Scene::intersects (src/scene.rs:55-61) as a
 * test hook.  INTEGRATION.md shows the Rust `extern "C"` block that binds it.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every pointer it passes;
 *    the library copies what it needs during cray_scene_upload.
 *  - every call returns 0 on success or a negative CRAY_ERR_* code;
 *    cray_last_error() gives a thread-local message.  Nothing aborts or throws
 *    across the ABI: the reference's hot-path panics become
 *    counters in cray_stats.
 *  - one host thread per cray_ctx at a time; one process per GPU (multi-GPU =
 *    one ctx per rank, tiles sharded by (rank, world) in cray_render_params).
 */
#ifndef CRAY_H
#define CRAY_H

#include <stddef.h>
#include <stdint.h>

#include "cray_scene_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CRAY_ABI_VERSION 2

enum {
    CRAY_OK = 0,
    CRAY_ERR_INVALID = -1,      /* bad argument / inconsistent scene */
    CRAY_ERR_HIP = -2,          /* HIP runtime error (message has the detail) */
    CRAY_ERR_UNSUPPORTED = -3,  /* e.g. BVH leaf with more than 8 primitives, depth > 31 */
    CRAY_ERR_NO_DEVICE = -4,
    CRAY_ERR_BUILD = -5         /* Bvh::new would have panicked (bvh.rs:245,304,327-328) */
};

/* ---- flattened Scene: what a host-side `Scene::flatten()` hands over ------ */

/* BvhNode (src/bvh.rs:13-24) in DFS pre-order; node 0 is the root. */
typedef struct {
    double bmin[3], bmax[3];    /* Bounds */
    uint32_t left, right;       /* InteriorNode children (indices into nodes[]) */
    uint32_t first, count;      /* LeafNode: prim_refs[first .. first+count) in the leaf's own order */
    int32_t axis;               /* InteriorNode split_axis: 0 X, 1 Y, 2 Z */
    int32_t is_leaf;
} cray_bvh_node;

/* Sphere / Disk with their Transformation pair resolved (src/shape.rs:55-69,133-153):
 * m = object_to_world.matrix, inv = object_to_world.inverse (== world_to_object.matrix). */
typedef struct {
    double m[16], inv[16];
    double radius, inner_radius;
} cray_xf_shape;

typedef struct {
    uint32_t abi_version;       /* CRAY_ABI_VERSION */
    /* Scene fields (src/scene.rs:15-23) */
    uint32_t max_depth, num_samples;
    /* Camera (src/camera.rs:16-23) */
    int32_t camera_type;
    uint32_t film_width, film_height;
    double camera_from_raster[16];  /* row-major Transformation.matrix */
    double world_from_camera[16];
    double lens_radius, focal_distance;
    /* Bvh (src/bvh.rs:31-35) */
    uint32_t n_nodes;     const cray_bvh_node* nodes;
    uint32_t n_prim_refs; const uint32_t* prim_refs;
    /* primitives / shapes */
    uint32_t n_prims;     const cray_prim* prims;
    uint32_t n_triangles; const cray_triangle* triangles;
    uint32_t n_spheres;   const cray_xf_shape* spheres;
    uint32_t n_disks;     const cray_xf_shape* disks;
    /* materials */
    uint32_t n_materials; const cray_material* materials;
    uint32_t n_bxdfs;     const cray_bxdf* bxdfs;
    uint32_t n_textures;  const cray_texture* textures;
    uint32_t n_images;    const cray_image* images;
    uint64_t image_pool_bytes; const uint8_t* image_pool;
    /* lights + LightSampler (src/light.rs:182-220) */
    uint32_t n_lights;    const cray_light* lights;
    const double* light_cdf;            /* LightSampler.cdfs, n_lights entries */
    const int32_t* first_equal_light;   /* lights.iter().position(|l| l == light), path_integrator.rs:116 */
    /* ABI 2 — resident build.  With build_on_device != 0 the scene comes WITHOUT a tree (n_nodes = n_prim_refs = 0):
     * cray_scene_upload computes the primitive bounds (shape.rs:402-438) and runs Bvh::new(.., SAH) (bvh.rs:38-56, 234-336)
     * on the GPU — the same tree cray_bvh_build_sah returns — and derives the traversal layout from it in place; the
     * 0.9 GB of nodes never visit the host.  Triangle bounds are computed on the device from `triangles`; the bounds of
     * the other primitives (spheres, disks: they need the host's transformation code) come in other_bounds. */
    uint32_t build_on_device;
    uint32_t n_other_bounds;
    const struct cray_prim_bound* other_bounds;
} cray_flat_scene;
typedef struct cray_prim_bound { uint32_t prim, pad_; double bmin[3], bmax[3]; } cray_prim_bound;

/* ---- render ----------------------------------------------------------------- */
typedef struct {
    uint64_t seed;              /* SobolSampler::new(seed, ..), craytracer.rs:361; Cli.seed :332-333 */
    uint32_t tile_width;        /* 64  (craytracer.rs:232) */
    uint32_t tile_height;       /* 64  (craytracer.rs:233) */
    uint32_t sample_batch;      /* 8   (craytracer.rs:234): f64 sum of a batch -> f32 add into the film */
    uint32_t rank, world_size;  /* this ctx renders the tiles (tx, ty) with (tx + s ty) % world_size == rank, s = the smallest stride >= 2 coprime with world_size (cray_tile_pixels);
                                   pixels of other tiles are left 0 in out_rgb. (0,1) = whole film */
    uint32_t sample_begin, sample_end; /* render samples [begin,end); (0,0) = all. Always divides by num_samples */
    uint32_t out_is_device;     /* 0: out_rgb is host memory; 1: device memory on the ctx's GPU */
    uint32_t count_traversal;   /* 1: also count BVH nodes / primitive tests (cray_stats) of EVERY query the reference
                                   makes; 2: count only the queries actually traversed (zero-term shadow rays skipped) */
    uint64_t max_paths_in_flight; /* 0 = default */
    /* The reference's selectable alternatives (ABI 2).  `main` hard-wires the path integrator with the Sobol sampler
     * (craytracer.rs:159-160, 361); simple_integrator::estimate_Li (src/simple_integrator.rs:36-143: no MIS, no roulette)
     * and UniformSampler (src/sampling.rs:154-194: slot centres, uniform_nx * uniform_ny must equal Scene.num_samples) are
     * what a maintainer gets by editing those lines; IndependentSampler (src/sampling.rs:102-146, rand's ChaCha12 StdRng) is
     * provided too, restated from the crates' published algorithms and NOT pinned against them. */
    uint32_t integrator;        /* CRAY_INTEGRATOR_PATH (default) | CRAY_INTEGRATOR_SIMPLE */
    uint32_t sampler;           /* CRAY_SAMPLER_SOBOL (default) | CRAY_SAMPLER_UNIFORM | CRAY_SAMPLER_INDEPENDENT */
    uint32_t uniform_nx, uniform_ny;
    /* CRAY_PRECISION_F64 (default): the reference's arithmetic, results identical to it.
     * CRAY_PRECISION_F32_TRAVERSAL: the "fast" mode of SURVEY.md §8(b) — the same tree traversed with f32 node / triangle
     * records and f32 slab / triangle tests (shading stays f64).  NOT bit-exact: films differ from the f64 path by a small
     * RMSE that shrinks with the sample count; reported separately, never the headline. */
    uint32_t precision, pad_;
} cray_render_params;
enum { CRAY_PRECISION_F64 = 0, CRAY_PRECISION_F32_TRAVERSAL = 1 };
enum { CRAY_INTEGRATOR_PATH = 0, CRAY_INTEGRATOR_SIMPLE = 1 };
enum { CRAY_SAMPLER_SOBOL = 0, CRAY_SAMPLER_UNIFORM = 1,
       CRAY_SAMPLER_INDEPENDENT = 2 /* sampling.rs:102-146: per pixel sample a ChaCha12 StdRng seeded from the SipHash of (seed, x, y,
                                       sample); restated from the published algorithms of rand 0.8.5 — not pinned against the crate */ };

typedef struct {
    uint64_t paths;                         /* W*H*spp rendered by this call */
    uint64_t closest_rays, shadow_rays;     /* Scene::intersect / Scene::intersects calls the reference makes */
    uint64_t shadow_skipped;                /* of shadow_rays: not traversed because the NEE term they gate is
                                               exactly (0,0,0) (e.g. any hit on a purely specular material);
                                               always 0 when count_traversal is set */
    uint64_t closest_nodes, closest_prims;  /* nodes popped / primitives tested (count_traversal) */
    uint64_t shadow_nodes, shadow_prims;
    uint64_t closest_tri_tests, shadow_tri_tests;
    uint64_t nonfinite;                     /* paths where the reference's assert!(is_finite) would fire */
    uint64_t stack_overflow;                /* lanes whose traversal stack overflowed: 0 in every successful call (the runtime
                                               adds a deeper stack level and repeats the work; beyond ~2100 pending nodes per
                                               ray the call fails with CRAY_ERR_UNSUPPORTED) */
    double seconds;                         /* first launch -> film complete, host clock around a device sync */
    double trace_closest_ms, trace_any_ms, shade_ms, other_ms; /* HIP-event time per kernel family */
    uint32_t trace_closest_launches, trace_any_launches, shade_launches;
    uint32_t trace_records;                 /* which records the traversal launches of this call read: low nibble the bounce-0
                                               launch, next nibble the others; 0 f64, 1 certified f32 culling.  The hits are the
                                               reference's either way; by default the library picks per scene
                                               whichever a few probe passes before the scene's first frame show to be faster */
    double trace_mixed_ms;                  /* launches that trace the shadow rays of bounce b together with the
                                               path segments of bounce b+1 (not used with count_traversal) */
    uint32_t trace_mixed_launches;
    uint32_t tail_split;                    /* small launches: low 24 bits = parts of closest-hit rays that idle lanes
                                               walked as helpers (saturating); high 8 bits = rays walked again alone because a helper's
                                               winning hit could not be certified (saturating) */
    uint64_t closest_hits;                  /* of closest_rays: segments that hit a primitive (the paths k_shade shades fully) */
} cray_stats;

/* test hook: batched Scene::intersect / Scene::intersects */
typedef struct { double o[3], d[3], tmax; } cray_ray;
typedef struct {
    int32_t hit;        /* 0/1 */
    int32_t prim;       /* index into prims[], -1 on miss (closest only) */
    double t;           /* PrimitiveIntersection.distance */
    double location[3], normal[3], uv[2];
} cray_hit;

typedef struct cray_ctx cray_ctx;
typedef struct cray_scene cray_scene;

/* Create a context on HIP device `device_id`. `stream` is a hipStream_t to launch on
 * (e.g. torch's current stream) or NULL for a stream owned by the context. */
int cray_ctx_create(int device_id, void* stream, cray_ctx** out);
void cray_ctx_destroy(cray_ctx* ctx);

/* Copy a flattened scene into HBM in the device layout (DESIGN.md "Data layout"). */
int cray_scene_upload(cray_ctx* ctx, const cray_flat_scene* scene, cray_scene** out);
void cray_scene_free(cray_scene* scene);
uint64_t cray_scene_device_bytes(const cray_scene* scene);
/* Film size, Scene.num_samples and Scene.max_depth of an uploaded (or broadcast-received) scene; any pointer may be NULL. */
void cray_scene_info(const cray_scene* scene, uint32_t* film_width, uint32_t* film_height, uint32_t* num_samples, uint32_t* max_depth);

/* The sampler's direction numbers (src/sampling.rs:196-247 -> sobol_burley 0.5.0, Cargo.lock:1019; the crate's source is
 * not part of the reference tree).  The built-in table is the bit-reversed 16-bit form of Joe & Kuo's
 * new-joe-kuo-6.21201 direction numbers for the first 256 dimensions; whether the crate embeds exactly this set cannot be
 * verified offline.  A host that has the crate can hand over its `REV_VECTORS` ([64 sets][16 bits][4 lanes] u16): scenes
 * uploaded afterwards sample with it.  NULL restores the built-in table.  Process-wide; call before cray_scene_upload. */
int cray_set_sobol_vectors(const uint16_t* rev_vectors /* 64 * 16 * 4 */);

/* Replaces `render` (craytracer.rs:224): fills out_rgb[W*H*3]. */
int cray_render(cray_ctx* ctx, cray_scene* scene, const cray_render_params* params, float* out_rgb,
                cray_stats* stats);
void cray_render_params_default(cray_render_params* params);

/* Per-path radiance of one sample batch, for per-sample parity checks:
 * out_L[(y*W + x)*n + (s - sample_begin)][3] f64, host memory. */
int cray_render_samples(cray_ctx* ctx, cray_scene* scene, const cray_render_params* params, double* out_L);

/* Replaces Scene::intersect / Scene::intersects (scene.rs:55-61) for a batch of rays: the per-ray parity hook.
 *   CRAY_TRACE_CLOSEST / CRAY_TRACE_ANY        the instrumented instantiations: stats carry the node / primitive counters
 *   CRAY_TRACE_CLOSEST_TIMED / _ANY_TIMED      the very kernels the frame loop times (k_trace<*, false>), no counters
 *   CRAY_TRACE_MIXED_TIMED                     k_trace_mixed as the frame loop launches it: every ray is traced as a shadow
 *                                              ray (with its tmax) AND as a path segment (tmax = +inf, like Ray::new) in ONE
 *                                              launch; hits[0..n) = the any-hit answers, hits[n..2n) = the closest-hit records
 * Closest-hit modes honour rays[i].tmax except the mixed one. hits: n entries (2n for the mixed mode).  stats (may be NULL): the
 * traversal counters of the instrumented modes, and the launch's time in trace_closest_ms / trace_any_ms / trace_mixed_ms. */
enum { CRAY_TRACE_CLOSEST = 0, CRAY_TRACE_ANY = 1, CRAY_TRACE_CLOSEST_TIMED = 2, CRAY_TRACE_ANY_TIMED = 3, CRAY_TRACE_MIXED_TIMED = 4 };
int cray_trace(cray_ctx* ctx, cray_scene* scene, const cray_ray* rays, size_t n, cray_hit* hits, int mode,
               cray_stats* stats);

/* Replaces Bvh::new(primitives, SplitMethod::SAH) (src/bvh.rs:38-56, 234-336) — the scene-load step that
 * takes seconds on the host for millions of triangles — with a build on the GPU that yields the SAME tree:
 * node bounds, split axes, topology and the order of primitives inside the leaves (util::partition_by,
 * src/util.rs:4-26) are those of the reference's sequential recursion.
 *   prim_bounds: n x 6 doubles (Bounds.min xyz, Bounds.max xyz of primitive i, shape.rs:402-438)
 *   out_nodes:   DFS pre-order, at most 2n-1 (node_capacity entries provided by the caller)
 *   out_prim_refs: n primitive indices in leaf order (cray_flat_scene.prim_refs) */
typedef struct {
    double device_seconds;  /* first kernel -> last kernel (HIP events) */
    double total_seconds;   /* including allocation and the copies in and out */
    uint32_t levels, top_nodes, small_subtrees, leaves;
} cray_bvh_build_stats;
int cray_bvh_build_sah(cray_ctx* ctx, const double* prim_bounds, uint32_t n, cray_bvh_node* out_nodes, uint32_t node_capacity,
                       uint32_t* out_n_nodes, uint32_t* out_prim_refs, cray_bvh_build_stats* stats);
/* Figures of the Bvh::new that ran inside cray_scene_upload for a resident build (cray_flat_scene.build_on_device);
 * zeros for a scene uploaded with its tree. */
void cray_scene_build_stats(const cray_scene* scene, cray_bvh_build_stats* out);

/* ---- multi-GPU: pixel-tile shard + gather of Film tiles over RCCL / xGMI -----------------------------
 * Replaces the reference's merge point, the shared `Mutex<Vec<f32>>` every worker thread adds its tile into
 * (src/bin/craytracer.rs:245, workers :271-291, merge :182-188), for workers that are GPUs of one node:
 * one process (or host thread) per GPU, one cray_ctx each.  Rank r renders the 64x64 tiles with
 * (tx + s ty) % world == r (tiles (tx, ty) of generate_tiles' grid, :22-43; s as in cray_render_params.rank) for ALL samples, so every pixel is
 * accumulated on one GPU in the single-GPU order and the assembled film is bit-identical to a 1-GPU render.
 * The only exchange is ONE gather of packed tiles to rank 0 after rendering (grouped ncclSend / ncclRecv:
 * W*H*12/world bytes per rank over the direct xGMI link to rank 0); nothing is communicated while rendering.
 *
 *   rank 0:  cray_comm_unique_id(&id);  -> ship the 128 bytes to the other ranks (pipe, file, MPI, a TCP store)
 *   all:     cray_ctx_create(local_gpu, NULL, &ctx); cray_comm_init(ctx, &id, rank, world);
 *   rank 0:  parse + Scene::new + cray_scene_upload;  all: cray_scene_broadcast(ctx, scene_or_NULL, 0, &scene)
 *            (or every rank uploads the scene itself)
 *   all:     cray_render_gather(ctx, scene, &params, rank == 0 ? film : NULL, &stats);
 *
 * librccl.so.1 is loaded on the first cray_comm_* call (dlopen), so single-GPU hosts do not need it. */
#define CRAY_COMM_ID_BYTES 128
typedef struct { char bytes[CRAY_COMM_ID_BYTES]; } cray_comm_id;   /* = ncclUniqueId */

/* ncclGetUniqueId: call on one rank, hand the bytes to all ranks by any host channel. */
int cray_comm_unique_id(cray_comm_id* out);
/* ncclCommInitRank on the ctx's GPU (collective: every rank calls it with the same id and world_size). */
int cray_comm_init(cray_ctx* ctx, const cray_comm_id* id, int rank, int world_size);
int cray_comm_rank(const cray_ctx* ctx);        /* 0 without a communicator */
int cray_comm_world_size(const cray_ctx* ctx);  /* 1 without a communicator */
/* All ranks wait for each other (one-word all-reduce on the ctx's stream + stream sync). */
int cray_comm_barrier(cray_ctx* ctx);
enum { CRAY_REDUCE_SUM = 0, CRAY_REDUCE_MAX = 1, CRAY_REDUCE_MIN = 2 };
/* In-place all-reduce of n host doubles (timings, ray counts): every rank gets the result. n <= 64. */
int cray_comm_allreduce_f64(cray_ctx* ctx, double* values, int n, int op);

/* What the communicator is made of, so that a multi-GPU timing can prove what it ran on (bench.py prints it next to every N > 1
 * figure): `ranks_seen` is an all-reduce (sum) of 1.0 over the communicator — the number of processes that actually took part,
 * as the transport counts them, not as the launcher's environment claims; `rccl_version` is ncclGetVersion() of the loaded
 * collective library (0 if it exports none); `library` the path it was loaded from (dladdr of its ncclSend), which tells
 * librccl.so from a stand-in.  Collective (one all-reduce) when the ctx has a communicator; without one: world 1, ranks_seen 1. */
typedef struct {
    int32_t world_size, rank;
    int32_t ranks_seen;
    int32_t rccl_version;
    char library[256];
} cray_comm_info;
int cray_comm_describe(cray_ctx* ctx, cray_comm_info* out);

/* C1: replicate a scene that is resident on `root`'s GPU into every other rank's HBM with ncclBroadcast over xGMI
 * (instead of parsing / building / uploading it once per rank).  On `root` pass the uploaded scene, *out == scene;
 * elsewhere pass NULL and receive a new scene (free it with cray_scene_free). Collective. */
int cray_scene_broadcast(cray_ctx* ctx, cray_scene* scene_on_root, int root, cray_scene** out);

/* cray_render of this rank's tiles followed by the gather: params->rank / world_size are taken from the
 * communicator.  out_rgb (W*H*3, host or device per params->out_is_device) is written on rank 0 only and may be
 * NULL elsewhere.  stats (optional) are this rank's; stats->seconds includes the gather.  Collective.
 * Without a communicator (or world_size 1) this is cray_render. */
int cray_render_gather(cray_ctx* ctx, cray_scene* scene, const cray_render_params* params, float* out_rgb,
                       cray_stats* stats);

/* The gather alone, for a host that rendered with cray_render(out_is_device = 1, rank, world_size): local_film is the
 * W*H*3 device film of this rank (pixels of other ranks' tiles are ignored). Collective. */
int cray_film_gather(cray_ctx* ctx, uint32_t width, uint32_t height, uint32_t tile_width, uint32_t tile_height,
                     const float* local_film_device, float* out_rgb, int out_is_device);

/* The two halves of the gather on ONE GPU, for tests and for hosts with their own transport:
 * pack: the pixels of `rank`'s tiles, tile by tile (row-major inside a tile), 3 floats each -> packed (host);
 *       *n_pixels = how many.  film and packed are host arrays.
 * unpack: `gathered` = the world_size packed arrays concatenated in rank order (W*H*3 floats) -> out (W*H*3). */
int cray_film_pack(cray_ctx* ctx, uint32_t width, uint32_t height, uint32_t tile_width, uint32_t tile_height,
                   uint32_t rank, uint32_t world_size, const float* film, float* packed, uint64_t* n_pixels);
int cray_film_unpack(cray_ctx* ctx, uint32_t width, uint32_t height, uint32_t tile_width, uint32_t tile_height,
                     uint32_t world_size, const float* gathered, float* out);

/* The shard map on its own (host only, no context, no GPU): the linear pixel indices y*W + x of the tiles `rank` owns, tile by
 * tile (the tiles (tx, ty) of generate_tiles' grid, craytracer.rs:22-43, with (tx + s ty) % world_size == rank, row by row; s = the smallest stride >= 2 coprime with world_size, world_size - 1 when there is none, 1 for one or two ranks), row-major inside a tile —
 * the order of cray_film_pack's output and of the buffer cray_render_gather sends.  out may be NULL to query *n_pixels. */
int cray_tile_pixels(uint32_t width, uint32_t height, uint32_t tile_width, uint32_t tile_height, uint32_t rank, uint32_t world_size,
                     uint32_t* out, uint64_t capacity, uint64_t* n_pixels);

/* Diagnostics for bench lines and cold-start accounting (the reference renders one frame per process, craytracer.rs:336-372):
 *   cray_ctx_pool_info       bytes of the path-state pool the context holds (allocated by its first cray_render, then kept) and the
 *                            paths it has room for; any pointer may be NULL
 *   cray_scene_records_info  what the scene's traversal launches read: chosen[2] = records of the bounce-0 launch / of the others
 *                            (0 f64, 1 certified f32 culling, -1 nothing chosen yet: pinned, or no frame big enough to time so far),
 *                            probe_ms = wall time of the probe passes that chose them (once per scene), and probe_kernel_ms[4] = best
 *                            time of those passes' bounce-0 launch / other traversal launches on f64 records, then on f32 culling */
void cray_ctx_pool_info(const cray_ctx* ctx, uint64_t* pool_bytes, uint64_t* paths);
void cray_scene_records_info(const cray_scene* scene, int32_t chosen[2], double* probe_ms, double probe_kernel_ms[4]);

/* Measurement aid for bench.py: GB/s of a plain 16-B-per-lane streaming read of `bytes` of HBM on this GPU (HIP events,
 * `repeats` launches after one warm-up) — the denominator "what a read-only kernel gets on this very box". */
int cray_measure_stream_read(cray_ctx* ctx, uint64_t bytes, int repeats, double* gb_per_s);

const char* cray_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* CRAY_H */
