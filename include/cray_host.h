/*
 * cray_host.h — C ABI of the host-side mirror of `Scene::new`.
 *
 * In a Rust deployment the host keeps craytracer's own Scene/Camera/Bvh code and
 * only adds a `flatten()`.  Rust is not available in this build
 * environment, so the same host logic exists here in C++:
 *   Scene::new                src/scene.rs:25-53
 *   Bvh::new (SAH / median)   src/bvh.rs:38-56, 191-336, util::partition_by src/util.rs:4-26
 *   Shape constructors/bounds src/shape.rs:55-69, 133-153, 402-438
 *   Camera::new               src/camera.rs:25-76
 *   LightSampler::new         src/light.rs:187-200, Light::power :170-177
 * and produces the `cray_flat_scene` that cray_scene_upload (cray.h) consumes.
 */
#ifndef CRAY_HOST_H
#define CRAY_HOST_H

#include "cray.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { CRAY_SPLIT_MEDIAN = 0, CRAY_SPLIT_SAH = 1 }; /* bvh.rs:26-29; Scene::new uses SAH (scene.rs:38) */

typedef struct cray_host_scene cray_host_scene;

/* Returns CRAY_ERR_BUILD where the reference would panic while building (zero surface
 * area, empty side after partition, no primitives); CRAY_ERR_INVALID for "No lights in
 * the scene." (scene_parser.rs:1104-1109) and bad indices. */
int cray_host_scene_new(const cray_scene_desc* desc, int split_method, cray_host_scene** out);
/* Same, with Bvh::new running on the GPU of `bvh_ctx` (cray_bvh_build_sah, cray.h; SAH only). The tree is
 * the one cray_host_scene_new builds; NULL = build on the host. */
int cray_host_scene_new_on(const cray_scene_desc* desc, int split_method, cray_ctx* bvh_ctx, cray_host_scene** out);
/* Scene::new for a RESIDENT build: everything but Bvh::new, which cray_scene_upload then runs on the GPU (triangle bounds,
 * SAH tree and traversal layout all stay in HBM: cray_flat_scene.build_on_device in cray.h).  The flat scene carries no
 * nodes; it is the fastest way from a cray_scene_desc to a renderable cray_scene (7.2 M triangles: DESIGN.md §10). */
int cray_host_scene_new_resident(const cray_scene_desc* desc, cray_host_scene** out);
/* The flat view stays valid until cray_host_scene_free; it borrows the desc's
 * material/texture/image/triangle arrays, which must outlive it too. */
const cray_flat_scene* cray_host_scene_flat(const cray_host_scene* scene);
double cray_host_scene_build_seconds(const cray_host_scene* scene);
/* Time spent in Bvh::new alone; `gpu` (optional) receives the GPU builder's figures (zeros for a host build). */
double cray_host_scene_bvh_seconds(const cray_host_scene* scene, cray_bvh_build_stats* gpu);
void cray_host_scene_free(cray_host_scene* scene);

/* The correctly rounded sin/cos the kernels use in sample_disk / sample_sphere
 * (src/sampling.rs:17-39), evaluated on the host by the very same code (cray_math.h). */
void cray_host_sincos(double x, double* sin_out, double* cos_out);

/* IndependentSampler's generator as the kernels run it (cray_math.h): one ChaCha block with words 12..15 of the state given
 * explicitly (double_rounds = 6 is ChaCha12, 10 is the RFC 8439 ChaCha20 whose section 2.3.2 vector pins the block function),
 * and draws first .. first + 7 of pixel sample (seed, x, y, sample_index). */
void cray_host_chacha_block(const uint32_t* key, const uint32_t* w12_15, int double_rounds, uint32_t* out);
void cray_host_independent_draws(uint64_t seed, uint64_t x, uint64_t y, uint64_t sample_index, uint32_t first, double* out);

/* The short evaluation + rounding test of the sampling sin/cos (cray_math.h sincos_fast_core) against the double-double evaluation
 * it falls back to: returns the number of arguments on which sincos_cr differs from the double-double result in any bit (must be 0);
 * stats[0] = calls that took the fallback, stats[1] = largest deviation of the short evaluation's candidate from the double-double
 * value, relative, in units of the test radius 2^-64 (must stay well below 1), stats[2] = n. */
uint64_t cray_host_sincos_fast_check(const double* x, uint64_t n, double* stats);

/* The exact FMA-based division the traversal kernel uses for (bound - origin) / direction
 * (cray_math.h div_fast): returns the number of i in [0,n) with div_fast(a[i], d[i], 1/d[i]) != a[i]/d[i]
 * bitwise, among the pairs that pass the range guard. */
uint64_t cray_host_div_fast_mismatches(const double* a, const double* d, uint64_t n);

/* The traversal kernel's fast slab test (cray_math.h child_key_fast) against the literal restatement of
 * Bounds::intersects / Bounds::contains (child_key) on n boxes (lo, hi: n x 3) and rays (o, d: n x 3): returns the
 * number of bitwise-different keys among the inputs inside the guarded range (their count in *n_checked). */
uint64_t cray_host_child_key_mismatches(const double* lo, const double* hi, const double* o, const double* d, uint64_t n,
                                        uint64_t* n_checked);

/* Certified f32 culling of the triangle test (cray_math.h tri_cull32) against the literal f64 Moller-Trumbore
 * (reference: src/shape.rs:216-262) on n triangles (v0, e1, e2), rays (o, d) and ray.tmax values: returns the number of certified
 * answers that contradict the exact test.  counts[4] (optional): unknown, certified misses, certified hits, rays outside the range. */
uint64_t cray_host_tri_cull_violations(const double* v0, const double* e1, const double* e2, const double* o, const double* d, const double* tmax,
                                       uint64_t n, uint64_t* counts);
/* Certified f32 culling (cray_math.h hyb_key / hyb_status) against the literal slab test on n boxes, rays and
 * ray.tmax values: returns how many times the f32 side certified a decision (visit: key < tmax, cull: key >= tmax) that the
 * exact f64 key contradicts — must be 0.  counts[4] = decisions left to the exact path, certified visits, certified culls,
 * rays outside the certified range. */
uint64_t cray_host_hyb_key_violations(const double* lo, const double* hi, const double* o, const double* d, const double* tmax,
                                      uint64_t n, uint64_t* counts);

#ifdef __cplusplus
}
#endif
#endif
