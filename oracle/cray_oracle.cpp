/*
 * cray_oracle.cpp — TEST INFRASTRUCTURE. CPU oracle for the craytracer hot path.
 *
 * A plain, scalar, f64 restatement of the reference algorithm
 *   render -> render_pixel -> path_integrator::estimate_Li -> Bvh::intersect/intersects
 *   -> Shape::intersect -> Material/BxDF/Light
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (craytracer_amd/) never does.
 *
 * PARITY STATUS
 *  - pinned by the reference's own known-answer tests: BVH 4 hits,
 *    sphere grids incl. 1e9 offsets, triangle vertex/behind/parallel, slab hits &
 *    misses, reflect/refract, transforms, matrix inverse, from_rgb
 *    (tests/test_oracle_reference_vectors.py).
 *  - UNPINNED: sobol_burley 0.5.0 (Cargo.lock:1019) is not in /root/reference; the
 *    sampler below restates Burley's Owen-scrambled Sobol from the published
 *    algorithm with scipy's Joe-Kuo direction numbers.  End-to-end images are
 *    therefore "oracle == product", conditional on that table for "== real binary".
 *  - UNPINNED: IndependentSampler's generator (rand 0.8.5, rand_chacha 0.3.1, rand_core 0.6.4: Cargo.lock, not in
 *    /root/reference) is restated from the crates' published algorithms (StdRngRestated below); `main` never selects that
 *    sampler.  Pinned only as far as RFC 8439's ChaCha vector and agreement of two implementations go (tests/test_alternatives.py).
 *  - The reference binary itself cannot be built here (no rustc/cargo).
 */
#include <quadmath.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../include/cray_scene_desc.h"
#include "orc_math.h"
#include "sobol_rev_vectors.h"

namespace orc {

/* ======================================================================== */
/* Sampler: SipHash-1-3 + Burley Owen-scrambled Sobol                         */
/* ======================================================================== */

static inline uint64_t rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }

/* Rust std DefaultHasher = SipHasher13 with k0 = k1 = 0 (src/sampling.rs:224-228:
 * seed.hash, x.hash, y.hash -> three LE u64 words = 24 message bytes). */
static uint64_t siphash13(const uint64_t* words, int n_words, uint64_t k0, uint64_t k1) {
    uint64_t v0 = k0 ^ 0x736f6d6570736575ULL, v1 = k1 ^ 0x646f72616e646f6dULL;
    uint64_t v2 = k0 ^ 0x6c7967656e657261ULL, v3 = k1 ^ 0x7465646279746573ULL;
#define ORC_SIPROUND                                                     \
    do {                                                                 \
        v0 += v1; v1 = rotl64(v1, 13); v1 ^= v0; v0 = rotl64(v0, 32);    \
        v2 += v3; v3 = rotl64(v3, 16); v3 ^= v2;                         \
        v0 += v3; v3 = rotl64(v3, 21); v3 ^= v0;                         \
        v2 += v1; v1 = rotl64(v1, 17); v1 ^= v2; v2 = rotl64(v2, 32);    \
    } while (0)
    for (int i = 0; i < n_words; i++) {
        uint64_t m = words[i];
        v3 ^= m;
        ORC_SIPROUND; /* c = 1 */
        v0 ^= m;
    }
    uint64_t b = ((uint64_t)(n_words * 8)) << 56; /* length byte, no tail bytes */
    v3 ^= b;
    ORC_SIPROUND;
    v0 ^= b;
    v2 ^= 0xff;
    ORC_SIPROUND; ORC_SIPROUND; ORC_SIPROUND; /* d = 3 */
#undef ORC_SIPROUND
    return v0 ^ v1 ^ v2 ^ v3;
}

/* generic SipHash c-d over bytes, used only to validate the round function with
 * the published SipHash-2-4 vector. */
static uint64_t siphash_cd(const uint8_t* msg, size_t len, uint64_t k0, uint64_t k1, int c, int d) {
    uint64_t v0 = k0 ^ 0x736f6d6570736575ULL, v1 = k1 ^ 0x646f72616e646f6dULL;
    uint64_t v2 = k0 ^ 0x6c7967656e657261ULL, v3 = k1 ^ 0x7465646279746573ULL;
    auto round = [&]() {
        v0 += v1; v1 = rotl64(v1, 13); v1 ^= v0; v0 = rotl64(v0, 32);
        v2 += v3; v3 = rotl64(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl64(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl64(v1, 17); v1 ^= v2; v2 = rotl64(v2, 32);
    };
    size_t nblocks = len / 8;
    for (size_t i = 0; i < nblocks; i++) {
        uint64_t m = 0;
        for (int j = 0; j < 8; j++) m |= (uint64_t)msg[i * 8 + j] << (8 * j);
        v3 ^= m;
        for (int r = 0; r < c; r++) round();
        v0 ^= m;
    }
    uint64_t b = (uint64_t)len << 56;
    for (size_t j = 0; j < (len & 7); j++) b |= (uint64_t)msg[nblocks * 8 + j] << (8 * j);
    v3 ^= b;
    for (int r = 0; r < c; r++) round();
    v0 ^= b;
    v2 ^= 0xff;
    for (int r = 0; r < d; r++) round();
    return v0 ^ v1 ^ v2 ^ v3;
}

static inline uint32_t pixel_hash(uint64_t seed, uint64_t x, uint64_t y) {
    uint64_t w[3] = {seed, x, y};
    return (uint32_t)siphash13(w, 3, 0, 0); /* `hasher.finish() as u32`, sampling.rs:228 */
}

/* sobol_burley 0.5.0 as published (Burley 2020 JCGT 9(4); Vegdahl's LK hash).
 * SURVEY Appendix C; constants recalled from the crate, table from scipy. */
static inline uint32_t reverse_bits32(uint32_t x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
}
static inline uint32_t sb_hash(uint32_t n) {
    uint32_t h = n ^ 0x79c68e4au;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
static inline uint32_t sb_hash_lane(uint32_t n, int lane) {
    static const uint32_t K[4] = {0x912f69bau, 0x174f18abu, 0x691e72cau, 0xb40cc1b8u};
    uint32_t h = n ^ K[lane];
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
static inline uint32_t sb_scramble_core(uint32_t n, uint32_t s) {
    n ^= n * 0x3d20adeau;
    n += s;
    n *= (s >> 16) | 1u;
    n ^= n * 0x05526c56u;
    n ^= n * 0x53a22864u;
    return n;
}
static inline float u32_to_f32_norm(uint32_t n) {
    uint32_t bits = (n >> 9) | 0x3f800000u;
    float f;
    memcpy(&f, &bits, 4);
    return f - 1.0f;
}
/* orc_set_sobol_vectors: a holder of the sobol_burley crate can load its own REV_VECTORS (the built-in table is scipy's
 * new-joe-kuo-6.21201, bit-reversed; whether the crate ships the same Joe-Kuo set cannot be checked offline). */
static uint16_t g_sobol_override[CRAY_SOBOL_SETS][CRAY_SOBOL_BITS][4];
static const uint16_t (*g_sobol_table)[CRAY_SOBOL_BITS][4] = CRAY_SOBOL_REV_VECTORS;
static void sobol_sample_4d(uint32_t sample_index, uint32_t dimension_set, uint32_t seed, float out[4]) {
    const uint16_t(*vecs)[4] = g_sobol_table[dimension_set];
    uint32_t shuffled_rev_index = sb_scramble_core(reverse_bits32(sample_index), sb_hash(seed));
    uint32_t sob[4] = {0, 0, 0, 0};
    uint32_t index = shuffled_rev_index & 0xffff0000u; /* top 16 bits only */
    for (int bit = 0; bit < 16; bit++) {
        if (index & (0x80000000u >> bit))
            for (int l = 0; l < 4; l++) sob[l] ^= vecs[bit][l];
    }
    for (int l = 0; l < 4; l++) {
        uint32_t s = sb_hash_lane(dimension_set ^ seed, l);
        out[l] = u32_to_f32_norm(reverse_bits32(sb_scramble_core(sob[l], s)));
    }
}
static inline float sobol_sample(uint32_t sample_index, uint32_t dimension, uint32_t seed) {
    float v[4];
    sobol_sample_4d(sample_index, dimension >> 2, seed, v);
    return v[dimension & 3];
}

/* Which of the reference's samplers / integrators the oracle runs (orc_set_mode): `main` uses SobolSampler + the path
 * integrator (craytracer.rs:159-160, 361); UniformSampler (sampling.rs:154-194) and simple_integrator::estimate_Li
 * (simple_integrator.rs:36-143) are its selectable alternatives. */
static int g_integrator = 0;                 /* 0 path_integrator, 1 simple_integrator */
static int g_sampler_kind = 0;               /* 0 Sobol, 1 Uniform, 2 Independent */
static uint64_t g_uniform_nx = 1, g_uniform_ny = 1;

/* IndependentSampler's generator (sampling.rs:102-146): `StdRng::seed_from_u64(hash)` and `rng.sample(Uniform::new(0.0, 1.0))`.
 * rand 0.8.5 / rand_chacha 0.3.1 / rand_core 0.6.4 are NOT in the container: this restates their published algorithms (PCG32
 * XSH-RR seed expansion, ChaCha with 12 rounds, 64-bit block counter and stream id 0, words consumed in order, 52 random mantissa
 * bits) — PARITY UNPINNED against the crates.  Written as a byte-oriented stream generator, independently of the kernels' version. */
struct StdRngRestated {
    uint8_t seed[32];
    uint64_t block;      /* next block counter */
    uint8_t buf[64];     /* current block, little-endian words */
    int used;            /* bytes of buf consumed */
    static uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
    void seed_from_u64(uint64_t state) {
        for (int i = 0; i < 8; i++) {
            state = state * 6364136223846793005ULL + 11634580027462260723ULL;
            uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
            uint32_t v = rot ? ((xs >> rot) | (xs << (32 - rot))) : xs;
            for (int b = 0; b < 4; b++) seed[4 * i + b] = (uint8_t)(v >> (8 * b));
        }
        block = 0; used = 64;
    }
    void refill() {
        uint32_t st[16], x[16];
        static const char sigma[17] = "expand 32-byte k";
        for (int i = 0; i < 4; i++) st[i] = (uint32_t)(uint8_t)sigma[4 * i] | (uint32_t)(uint8_t)sigma[4 * i + 1] << 8 | (uint32_t)(uint8_t)sigma[4 * i + 2] << 16 | (uint32_t)(uint8_t)sigma[4 * i + 3] << 24;
        for (int i = 0; i < 8; i++) st[4 + i] = (uint32_t)seed[4 * i] | (uint32_t)seed[4 * i + 1] << 8 | (uint32_t)seed[4 * i + 2] << 16 | (uint32_t)seed[4 * i + 3] << 24;
        st[12] = (uint32_t)block; st[13] = (uint32_t)(block >> 32); st[14] = 0; st[15] = 0;
        memcpy(x, st, sizeof(x));
        static const int idx[8][4] = {{0, 4, 8, 12}, {1, 5, 9, 13}, {2, 6, 10, 14}, {3, 7, 11, 15}, {0, 5, 10, 15}, {1, 6, 11, 12}, {2, 7, 8, 13}, {3, 4, 9, 14}};
        for (int round = 0; round < 12; round += 2)
            for (int q = 0; q < 8; q++) {
                uint32_t &a = x[idx[q][0]], &b = x[idx[q][1]], &c = x[idx[q][2]], &d = x[idx[q][3]];
                a += b; d = rotl(d ^ a, 16);
                c += d; b = rotl(b ^ c, 12);
                a += b; d = rotl(d ^ a, 8);
                c += d; b = rotl(b ^ c, 7);
            }
        for (int i = 0; i < 16; i++) {
            uint32_t v = x[i] + st[i];
            for (int b = 0; b < 4; b++) buf[4 * i + b] = (uint8_t)(v >> (8 * b));
        }
        block += 1; used = 0;
    }
    uint64_t next_u64() {   /* two consecutive u32 words, low word first (BlockRng::next_u64) */
        if (used == 64) refill();
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v |= (uint64_t)buf[used + b] << (8 * b);
        used += 8;
        return v;
    }
    double uniform01() {    /* UniformFloat<f64>::sample with low = 0, scale = 1 */
        uint64_t bits = (next_u64() >> 12) | 0x3ff0000000000000ULL;
        double v;
        memcpy(&v, &bits, 8);
        return v - 1.0;
    }
};

/* SobolSampler, src/sampling.rs:196-247; UniformSampler, :154-194; IndependentSampler, :102-146 */
struct Sampler {
    uint64_t seed;
    uint32_t hash, sample_index, dimension;
    StdRngRestated rng;
    void start_pixel(uint64_t x, uint64_t y, uint64_t s) {
        hash = pixel_hash(seed, x, y);
        sample_index = (uint32_t)s;
        dimension = 0;
        if (g_sampler_kind == 2) { /* :125-137 */
            uint64_t w[4] = {seed, x, y, s};
            rng.seed_from_u64(siphash13(w, 4, 0, 0));
        }
    }
    double sample_1d() { /* :234-238 */
        if (g_sampler_kind == 2) return rng.uniform01(); /* :139-141 */
        if (g_sampler_kind == 1) /* :180-183 */
            return ((double)sample_index + 0.5) / (double)(g_uniform_nx * g_uniform_ny);
        float s = sobol_sample(sample_index, dimension, hash);
        dimension += 1;
        return (double)s;
    }
    void sample_2d(double* a, double* b) { /* :240-246 */
        if (g_sampler_kind == 2) { *a = rng.uniform01(); *b = rng.uniform01(); return; } /* :143-145 */
        if (g_sampler_kind == 1) { /* :185-192 */
            uint64_t x = sample_index % g_uniform_nx, y = sample_index / g_uniform_nx;
            *a = ((double)x + 0.5) / (double)g_uniform_nx;
            *b = ((double)y + 0.5) / (double)g_uniform_ny;
            return;
        }
        float sx = sobol_sample(sample_index, dimension, hash);
        dimension += 1;
        float sy = sobol_sample(sample_index, dimension, hash);
        dimension += 1;
        *a = (double)sx;
        *b = (double)sy;
    }
};

/* ======================================================================== */
/* sin / cos of the sampling functions                                        */
/* ======================================================================== */
/* Rust documents f64::sin/cos as platform-precision ("The precision of this function is
 * non-deterministic", std docs): on Linux they are glibc's, which round ~0.15 % of calls the
 * "wrong" way and differ in the last bit from any other libm.  The reference's shadow-ray
 * leak (scenes/rounding-error.cry) amplifies such a 1-ulp difference into a different path
 * about once per 3000 paths, so "identical results" needs ONE definite sin/cos.  The oracle
 * uses the canonical one: the correctly rounded value, obtained through binary128
 * (libquadmath).  g_libm_mode = 1 switches to the platform libm to measure the difference
 * (tests/test_oracle_libm.py). */
static int g_libm_mode = 0;
static inline double sample_sin(double x) { return g_libm_mode ? std::sin(x) : (double)sinq((__float128)x); }
static inline double sample_cos(double x) { return g_libm_mode ? std::cos(x) : (double)cosq((__float128)x); }

/* ======================================================================== */
/* sampling_fns, src/sampling.rs:1-66                                         */
/* ======================================================================== */
static inline double power_heuristic(double pdf_f, double pdf_g) { /* :11-15 with n_f = n_g = 1 */
    double f = 1.0 * pdf_f, g = 1.0 * pdf_g;
    return (f * f) / (f * f + g * g);
}
static inline void sample_disk(double u, double v, double* x, double* y) { /* :17-29 */
    if (u == 0.0 || v == 0.0) { *x = 0.0; *y = 0.0; return; }
    u = 2.0 * u - 1.0; v = 2.0 * v - 1.0;
    double r, theta;
    if (std::fabs(u) > std::fabs(v)) { r = u; theta = FRAC_PI_4 * v / u; }
    else { r = v; theta = FRAC_PI_2 - FRAC_PI_4 * u / v; }
    *x = sample_cos(theta) * r;
    *y = sample_sin(theta) * r;
}
static inline V3 sample_sphere(double u, double v) { /* :31-39 */
    double z = 1.0 - 2.0 * u;
    double r = std::sqrt(rmax(1.0 - sq(z), 0.0));
    double phi = 2.0 * PI * v;
    return v3(r * sample_cos(phi), r * sample_sin(phi), z);
}
static inline V3 sample_hemisphere(double u, double v, V3 normal) { /* :41-48 */
    V3 r = sample_sphere(u, v);
    return dot(r, normal) > 0.0 ? r : neg(r);
}
static inline void sample_triangle(double u, double v, double* b1, double* b2) { /* :51-55 */
    double su = std::sqrt(u);
    *b1 = 1.0 - su; *b2 = v * su;
}
static inline V3 cosine_sample_hemisphere(double u, double v, V3 normal, int* assert_fail) { /* :57-65 */
    V3 t, b;
    generate_tangents(normal, &t, &b);
    double x, y;
    sample_disk(u, v, &x, &y);
    double z = std::sqrt(rmax(1.0 - x * x - y * y, 0.0));
    V3 a = t * x + b * y + normal * z;
    if (!(dot(a, normal) >= 0.0)) *assert_fail += 1; /* reference: assert! -> panic */
    return normalized(a);
}

/* ======================================================================== */
/* Scene data                                                                 */
/* ======================================================================== */
struct XfPair { Xf o2w, w2o; };
/* A Shape as stored per primitive.  Sphere/Disk transformations live behind a pointer so that a
 * 7.2 M-triangle scene does not carry four unused matrices per triangle. */
struct Shape {
    int kind;
    const XfPair* xf; /* sphere / disk */
    double radius, inner_radius;
    cray_triangle tri;
};

struct Hit { /* ShapeIntersection + PrimitiveIntersection, src/intersection.rs */
    double distance;
    V3 location, normal;
    double u, v;
    int prim;
};

struct Light { int kind; int prim; V3 v; Col c; };

struct Node { /* BvhNode, src/bvh.rs:13-24, flattened in DFS pre-order */
    Bounds bounds;
    uint32_t left, right; /* interior */
    uint32_t first, count; /* leaf: range in prim_order */
    int axis;
    bool leaf;
};

struct Stats {
    uint64_t closest_rays, shadow_rays;
    uint64_t closest_nodes, closest_prims, shadow_nodes, shadow_prims;
    uint64_t closest_hits, closest_tri_tests, shadow_tri_tests;
    uint64_t paths, nonfinite, assert_fail;
    void add(const Stats& o) {
        closest_rays += o.closest_rays; shadow_rays += o.shadow_rays;
        closest_nodes += o.closest_nodes; closest_prims += o.closest_prims;
        shadow_nodes += o.shadow_nodes; shadow_prims += o.shadow_prims;
        closest_hits += o.closest_hits; closest_tri_tests += o.closest_tri_tests;
        shadow_tri_tests += o.shadow_tri_tests;
        paths += o.paths; nonfinite += o.nonfinite; assert_fail += o.assert_fail;
    }
};

struct Scene {
    cray_scene_desc d; /* shallow copy; arrays below are owned deep copies */
    std::vector<cray_prim> prims;
    std::vector<Shape> shapes; /* per primitive */
    std::vector<cray_material> materials;
    std::vector<cray_bxdf> bxdfs;
    std::vector<cray_texture> textures;
    std::vector<cray_image> images;
    std::vector<uint8_t> pool;
    std::vector<Light> lights;
    std::vector<double> cdfs;
    std::vector<int> first_equal_light;
    /* camera */
    int cam_type;
    Xf camera_from_raster, world_from_camera;
    double lens_radius, focal_distance;
    uint32_t W, H, max_depth, num_samples;
    /* bvh */
    std::vector<Node> nodes;
    std::vector<uint32_t> prim_order;
    Bounds bvh_bounds;
    int build_error;
};

/* Shape constructors, src/shape.rs:55-69, 133-153 */
static Shape make_sphere(V3 o, double radius) {
    Shape s; memset(&s, 0, sizeof(s));
    s.kind = CRAY_SHAPE_SPHERE;
    s.radius = radius;
    XfPair* x = new XfPair(); /* lives as long as the process: a handful per scene */
    x->o2w = xf_translate(o.x, o.y, o.z);
    x->w2o = xf_translate(-o.x, -o.y, -o.z);
    s.xf = x;
    return s;
}
static Shape make_disk(V3 o, double rx, double ry, double radius, double inner) {
    Shape s; memset(&s, 0, sizeof(s));
    s.kind = CRAY_SHAPE_DISK;
    s.radius = radius; s.inner_radius = inner;
    XfPair* x = new XfPair();
    x->o2w = xf_mul(xf_mul(xf_translate(o.x, o.y, o.z), xf_rotate_x(to_radians(rx))), xf_rotate_y(to_radians(ry)));
    x->w2o = xf_inverse(x->o2w);
    s.xf = x;
    return s;
}
static inline V3 cv(const cray_vec3& a) { return v3(a.x, a.y, a.z); }
static inline Col cc(const cray_color& a) { return col(a.r, a.g, a.b); }

/* Shape::intersect, src/shape.rs:157-311. Mutates ray.tmax on acceptance. */
static bool shape_intersect(const Shape& s, Ray& ray, Hit* h) {
    if (s.kind == CRAY_SHAPE_SPHERE) { /* :159-215 */
        Ray obj = xf_ray(s.xf->w2o, ray);
        V3 oc = obj.o;
        double a = magnitude_squared(obj.d);
        double b = 2.0 * dot(oc, obj.d);
        double c = magnitude_squared(oc) - sq(s.radius);
        double disc = b * b - 4.0 * a * c;
        if (disc < 0.0) return false;
        double disc_sqrt = std::sqrt(disc);
        double inv_2_a = 1.0 / (2.0 * a);
        double roots[2] = {(-b - disc_sqrt) * inv_2_a, (-b + disc_sqrt) * inv_2_a};
        for (int k = 0; k < 2; k++) {
            double distance = roots[k];
            if (update_max_distance(obj, distance)) {
                V3 loc = ray_at(obj, distance);
                update_max_distance(ray, distance);
                double phi = std::atan2(loc.y, loc.x);
                if (phi < 0.0) phi += PI * 2.0;
                double u = phi / (PI * 2.0);
                double theta = std::acos(loc.z / s.radius);
                double v = theta * FRAC_1_PI;
                h->location = xf_point(s.xf->o2w, loc);
                h->normal = xf_normal(s.xf->o2w, loc / s.radius);
                h->u = u; h->v = v;
                return true;
            }
        }
        return false;
    } else if (s.kind == CRAY_SHAPE_TRIANGLE) { /* :216-262 */
        const cray_triangle& t = s.tri;
        V3 e1 = cv(t.e1), e2 = cv(t.e2);
        V3 P = cross(ray.d, e2);
        double denom = dot(P, e1);
        if (denom > -EPSILON && denom < EPSILON) return false;
        V3 T = ray.o - cv(t.v0);
        double u = dot(P, T) / denom;
        if (u < 0.0 || u > 1.0) return false;
        V3 Q = cross(T, e1);
        double v = dot(Q, ray.d) / denom;
        if (v < 0.0 || u + v > 1.0) return false;
        double distance = dot(cross(T, e1), e2) / denom;
        if (update_max_distance(ray, distance)) {
            h->location = ray_at(ray, distance);
            h->normal = normalized(cv(t.n0) + cv(t.n01) * u + cv(t.n02) * v);
            h->u = t.uv0[0] + t.uv01[0] * u + t.uv02[0] * v;
            h->v = t.uv0[1] + t.uv01[1] * u + t.uv02[1] * v;
            return true;
        }
        return false;
    } else { /* Disk, :263-309 */
        Ray obj = xf_ray(s.xf->w2o, ray);
        if (obj.d.z == 0.0) return false;
        double t = -obj.o.z / obj.d.z;
        if (!contains_distance(obj, t)) return false;
        V3 loc = v3(obj.o.x + obj.d.x * t, obj.o.y + obj.d.y * t, 0.0);
        double d2 = sq(loc.x) + sq(loc.y);
        if (d2 < sq(s.inner_radius) || d2 > sq(s.radius)) return false;
        double theta = std::atan2(loc.y, loc.x);
        if (theta < 0.0) theta += PI * 2.0;
        double u = theta / (PI * 2.0);
        double v = std::sqrt(d2) / s.radius;
        if (update_max_distance(ray, t)) {
            h->location = xf_point(s.xf->o2w, loc);
            h->normal = xf_normal(s.xf->o2w, v3(0.0, 0.0, 1.0));
            h->u = u; h->v = v;
            return true;
        }
        return false;
    }
}

/* Shape::intersects, src/shape.rs:314-400 */
static bool shape_intersects(const Shape& s, const Ray& ray) {
    if (s.kind == CRAY_SHAPE_SPHERE) {
        Ray obj = xf_ray(s.xf->w2o, ray);
        V3 oc = obj.o;
        double a = magnitude_squared(obj.d);
        double b = 2.0 * dot(oc, obj.d);
        double c = magnitude_squared(oc) - sq(s.radius);
        double disc = b * b - 4.0 * a * c;
        if (disc < 0.0) return false;
        double disc_sqrt = std::sqrt(disc);
        double inv_2_a = 1.0 / (2.0 * a);
        double distance = (-b - disc_sqrt) * inv_2_a;
        if (contains_distance(obj, distance)) return true;
        distance = (-b + disc_sqrt) * inv_2_a;
        return contains_distance(obj, distance);
    } else if (s.kind == CRAY_SHAPE_TRIANGLE) {
        const cray_triangle& t = s.tri;
        V3 e1 = cv(t.e1), e2 = cv(t.e2);
        V3 P = cross(ray.d, e2);
        double denom = dot(P, e1);
        if (denom > -EPSILON && denom < EPSILON) return false;
        V3 T = ray.o - cv(t.v0);
        double u = dot(P, T) / denom;
        if (u < 0.0 || u > 1.0) return false;
        V3 Q = cross(T, e1);
        double v = dot(Q, ray.d) / denom;
        if (v < 0.0 || u + v > 1.0) return false;
        double distance = dot(cross(T, e1), e2) / denom;
        return contains_distance(ray, distance);
    } else {
        Ray obj = xf_ray(s.xf->w2o, ray);
        if (obj.d.z == 0.0) return false;
        double t = -obj.o.z / obj.d.z;
        if (!contains_distance(obj, t)) return false;
        V3 loc = v3(obj.o.x + obj.d.x * t, obj.o.y + obj.d.y * t, 0.0);
        double d2 = sq(loc.x) + sq(loc.y);
        if (d2 < sq(s.inner_radius) || d2 > sq(s.radius)) return false;
        return contains_distance(ray, t);
    }
}

/* Shape::bounds, src/shape.rs:402-438 */
static Bounds shape_bounds(const Shape& s) {
    if (s.kind == CRAY_SHAPE_SPHERE) {
        double r = s.radius;
        return xf_bounds(s.xf->o2w, bounds_new(v3(-r, -r, -r), v3(r, r, r)));
    } else if (s.kind == CRAY_SHAPE_TRIANGLE) {
        V3 v0 = cv(s.tri.v0);
        V3 v1 = v0 + cv(s.tri.e1), v2 = v0 + cv(s.tri.e2);
        return bounds_new(v3(rmin(v1.x, rmin(v2.x, v0.x)), rmin(v1.y, rmin(v2.y, v0.y)), rmin(v1.z, rmin(v2.z, v0.z))),
                          v3(rmax(v1.x, rmax(v2.x, v0.x)), rmax(v1.y, rmax(v2.y, v0.y)), rmax(v1.z, rmax(v2.z, v0.z))));
    } else {
        double r = s.radius;
        return xf_bounds(s.xf->o2w, bounds_new(v3(-r, -r, 0.0), v3(r, r, 0.0)));
    }
}

/* Shape::area, src/shape.rs:504-514 (sphere is PI r^2, sic) */
static double shape_area(const Shape& s) {
    if (s.kind == CRAY_SHAPE_SPHERE) return PI * sq(s.radius);
    if (s.kind == CRAY_SHAPE_TRIANGLE) return magnitude(cross(cv(s.tri.e1), cv(s.tri.e2))) / 2.0;
    return PI * (sq(s.radius) - sq(s.inner_radius));
}

/* Shape::sample, src/shape.rs:445-470 */
static V3 shape_sample(const Shape& s, double u, double v) {
    if (s.kind == CRAY_SHAPE_SPHERE) {
        V3 p = v3(0, 0, 0) + sample_sphere(u, v) * s.radius;
        return xf_point(s.xf->o2w, p);
    } else if (s.kind == CRAY_SHAPE_TRIANGLE) {
        double b1, b2;
        sample_triangle(u, v, &b1, &b2);
        return cv(s.tri.v0) + cv(s.tri.e1) * b1 + cv(s.tri.e2) * b2;
    } else {
        double x, y;
        sample_disk(u, v, &x, &y);
        return xf_point(s.xf->o2w, v3(x * s.radius, y * s.radius, 0.0));
    }
}

/* Shape::pdf_from, src/shape.rs:487-502. Returns NonDelta(pdf). */
static double shape_pdf_from(const Shape& s, V3 isect_location, V3 isect_normal, V3 w_i) {
    Ray ray = ray_new(isect_location, w_i);
    Hit h;
    if (shape_intersect(s, ray, &h)) {
        double distance_squared = magnitude_squared(h.location - isect_location);
        double cos_theta = std::fabs(dot(w_i, isect_normal));
        return distance_squared / (cos_theta * shape_area(s));
    }
    return 0.0;
}

/* ======================================================================== */
/* Textures, src/texture.rs:19-47, 103-113                                    */
/* ======================================================================== */
static inline double fract(double x) { return x - std::trunc(x); } /* f64::fract */

static Col tex_eval_color(const Scene& sc, int tex, double u, double v) {
    const cray_texture& t = sc.textures[tex];
    if (t.kind == CRAY_TEX_CONSTANT) return cc(t.a);
    if (t.kind == CRAY_TEX_CHECKERBOARD) {
        uint64_t uu = sat_u64(u * t.scale * 2.0), vv = sat_u64(v * t.scale * 2.0);
        return ((uu & 1) ^ (vv & 1)) == 0 ? cc(t.a) : cc(t.b);
    }
    const cray_image& im = sc.images[t.image];
    double fu = fract(u); if (fu < 0.0) fu += 1.0;
    double fv = fract(v); if (fv < 0.0) fv += 1.0;
    uint32_t x = sat_u32((double)(im.width - 1) * fu);
    uint32_t y = sat_u32((double)(im.height - 1) * fv);
    const uint8_t* p = &sc.pool[im.offset + 3ull * ((uint64_t)y * im.width + x)];
    return from_rgb(p[0], p[1], p[2]);
}
/* Texture<f64>; image -> Rgb::to_luma (image 0.24: (2126 r + 7152 g + 722 b) / 10000) / 255 */
static double tex_eval_f64(const Scene& sc, int tex, double u, double v) {
    const cray_texture& t = sc.textures[tex];
    if (t.kind == CRAY_TEX_CONSTANT) return t.a.r;
    if (t.kind == CRAY_TEX_CHECKERBOARD) {
        uint64_t uu = sat_u64(u * t.scale * 2.0), vv = sat_u64(v * t.scale * 2.0);
        return ((uu & 1) ^ (vv & 1)) == 0 ? t.a.r : t.b.r;
    }
    const cray_image& im = sc.images[t.image];
    double fu = fract(u); if (fu < 0.0) fu += 1.0;
    double fv = fract(v); if (fv < 0.0) fv += 1.0;
    uint32_t x = sat_u32((double)(im.width - 1) * fu);
    uint32_t y = sat_u32((double)(im.height - 1) * fv);
    const uint8_t* p = &sc.pool[im.offset + 3ull * ((uint64_t)y * im.width + x)];
    uint32_t luma = (2126u * p[0] + 7152u * p[1] + 722u * p[2]) / 10000u;
    return (double)(uint8_t)luma / 255.0;
}

/* ======================================================================== */
/* BxDF / BSDF / Material, src/bxdf.rs, src/bsdf.rs, src/material.rs          */
/* ======================================================================== */
static inline V3 reflect(V3 direction, V3 normal) { /* bxdf.rs:287-290 */
    return normal * (dot(normal, direction) * 2.0) - direction;
}
static bool refract(V3 direction, V3 normal, double cos_theta_i, double eta_i, double eta_t, V3* out) { /* :292-314 */
    double eta_relative, cos_theta;
    if (std::signbit(cos_theta_i)) { normal = neg(normal); eta_relative = eta_i / eta_t; cos_theta = -cos_theta_i; }
    else { eta_relative = eta_t / eta_i; cos_theta = cos_theta_i; }
    double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
    if (sin_theta > eta_relative) return false;
    V3 r_perp = (normal * cos_theta - direction) / eta_relative;
    V3 r_par = normal * -std::sqrt(1.0 - dot(r_perp, r_perp));
    *out = r_perp + r_par;
    return true;
}
static double fresnel_dielectric(double eta_i, double eta_t, double cos_theta_i) { /* :338-357 */
    if (std::signbit(cos_theta_i)) { cos_theta_i = -cos_theta_i; double t = eta_i; eta_i = eta_t; eta_t = t; }
    double sin_theta_i = std::sqrt(1.0 - cos_theta_i * cos_theta_i);
    double sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0) return 1.0;
    double cos_theta_t = std::sqrt(1.0 - sin_theta_t * sin_theta_t);
    double r_par = (eta_t * cos_theta_i - eta_i * cos_theta_t) / (eta_t * cos_theta_i + eta_i * cos_theta_t);
    double r_perp = (eta_i * cos_theta_i - eta_t * cos_theta_t) / (eta_i * cos_theta_i + eta_t * cos_theta_t);
    return (r_par * r_par + r_perp * r_perp) * 0.5;
}
static Col fresnel_conductor(Col eta_i, Col eta_t, Col k, double cos_theta_i, int* assert_fail) { /* :359-382 */
    if (!(cos_theta_i >= 0.0)) *assert_fail += 1;
    Col eta_rel = eta_t / eta_i;
    Col eta_rel_2 = eta_rel * eta_rel;
    Col k_rel = k / eta_i;
    Col k_rel_2 = k_rel * k_rel;
    double cos_theta_2 = cos_theta_i * cos_theta_i;
    double sin_theta_2 = 1.0 - cos_theta_2;
    Col t0 = eta_rel_2 - k_rel_2 - WHITE * sin_theta_2;
    Col a2_plus_b2 = col_pow_half(t0 * t0 + eta_rel_2 * k_rel_2 * 4.0);
    Col a = col_pow_half((a2_plus_b2 + t0) * 0.5);
    Col t1 = a2_plus_b2 + WHITE * cos_theta_2;
    Col t2 = a * cos_theta_i * 2.0;
    Col r_perp = (t1 - t2) / (t1 + t2);
    Col t3 = a2_plus_b2 * cos_theta_2 + WHITE * sin_theta_2 * sin_theta_2;
    Col t4 = a * cos_theta_i * sin_theta_2 * 2.0;
    Col r_par = r_perp * (t3 - t4) / (t3 + t4);
    return (r_par * r_par + r_perp * r_perp) * 0.5;
}

struct SurfaceSample { V3 w_i; Col f; bool delta; double pdf; bool is_specular; };

static inline bool bxdf_has_reflection(int kind) { return kind != CRAY_BXDF_SPECULAR_BTDF; }      /* bxdf.rs:57-66 */
static inline bool bxdf_has_transmission(int kind) {                                               /* :68-77 */
    return kind == CRAY_BXDF_SPECULAR_BTDF || kind == CRAY_BXDF_FRESNEL_SPECULAR;
}

/* BxDF::f, bxdf.rs:214-265 */
static Col bxdf_f(const Scene& sc, const cray_bxdf& bx, V3 w_o, V3 w_i, V3 normal, double u, double v) {
    if (bx.kind == CRAY_BXDF_LAMBERTIAN) {
        if (same_hemisphere(normal, w_o, w_i)) return tex_eval_color(sc, bx.tex_a, u, v) * FRAC_1_PI;
        return BLACK;
    }
    if (bx.kind == CRAY_BXDF_OREN_NAYAR) {
        if (!same_hemisphere(normal, w_o, w_i)) return BLACK;
        double cos_theta_i = std::fabs(dot(w_i, normal));
        double cos_theta_o = std::fabs(dot(w_o, normal));
        double sin_theta_i = std::sqrt(rmax(1.0 - cos_theta_i * cos_theta_i, 0.0));
        double sin_theta_o = std::sqrt(rmax(1.0 - cos_theta_o * cos_theta_o, 0.0));
        double max_cos = 0.0;
        if (sin_theta_i > 1e-4 && sin_theta_o > 1e-4) {
            V3 tangent, bt;
            generate_tangents(normal, &tangent, &bt);
            double cos_phi_i = std::fabs(dot(w_i, tangent));
            double cos_phi_o = std::fabs(dot(w_o, tangent));
            double sin_phi_i = std::sqrt(1.0 - cos_phi_i * cos_phi_i);
            double sin_phi_o = std::sqrt(1.0 - cos_phi_o * cos_phi_o);
            max_cos = rmax(cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o, 0.0);
        }
        double sin_alpha, tan_beta;
        if (cos_theta_i > cos_theta_o) { sin_alpha = sin_theta_o; tan_beta = sin_theta_i / cos_theta_i; }
        else { sin_alpha = sin_theta_i; tan_beta = sin_theta_o / cos_theta_o; }
        double sigma = to_radians(tex_eval_f64(sc, bx.tex_b, u, v));
        double sigma_2 = sigma * sigma;
        double A = 1.0 - sigma_2 / (2.0 * (sigma_2 + 0.33));
        double B = 0.45 * sigma_2 / (sigma_2 + 0.09);
        return tex_eval_color(sc, bx.tex_a, u, v) * (A + B * max_cos * sin_alpha * tan_beta) * FRAC_1_PI;
    }
    return BLACK;
}
/* BxDF::pdf, bxdf.rs:269-284; returns false for Pdf::Delta */
static bool bxdf_pdf(const cray_bxdf& bx, V3 w_i, V3 normal, double* pdf) {
    if (bx.kind == CRAY_BXDF_LAMBERTIAN || bx.kind == CRAY_BXDF_OREN_NAYAR) {
        double cos_theta = std::fabs(dot(w_i, normal));
        *pdf = FRAC_1_PI * cos_theta;
        return true;
    }
    return false;
}
/* BxDF::sample, bxdf.rs:83-209 */
static bool bxdf_sample(const Scene& sc, const cray_bxdf& bx, double s0, double s1, V3 w_o, V3 normal, double u,
                        double v, SurfaceSample* out, int* assert_fail) {
    switch (bx.kind) {
    case CRAY_BXDF_LAMBERTIAN:
    case CRAY_BXDF_OREN_NAYAR: {
        V3 w_i = cosine_sample_hemisphere(s0, s1, normal, assert_fail);
        if (dot(normal, w_o) < 0.0) w_i = neg(w_i);
        out->w_i = w_i;
        out->f = bxdf_f(sc, bx, w_o, w_i, normal, u, v);
        out->delta = !bxdf_pdf(bx, w_i, normal, &out->pdf);
        out->is_specular = false;
        return true;
    }
    case CRAY_BXDF_FRESNEL_CONDUCTOR: {
        V3 w_i = reflect(w_o, normal);
        if (!(std::fabs(magnitude(w_i) - 1.0) <= EPSILON)) *assert_fail += 1;
        double cos_theta_i = std::fabs(dot(w_o, normal));
        Col fr = fresnel_conductor(WHITE, tex_eval_color(sc, bx.tex_a, u, v), tex_eval_color(sc, bx.tex_b, u, v),
                                   cos_theta_i, assert_fail);
        out->w_i = w_i; out->f = fr / cos_theta_i; out->delta = true; out->pdf = 0.0; out->is_specular = true;
        return true;
    }
    case CRAY_BXDF_SPECULAR_BRDF: {
        V3 w_i = reflect(w_o, normal);
        if (!(std::fabs(magnitude(w_i) - 1.0) <= EPSILON)) *assert_fail += 1;
        double cos_theta_i = std::fabs(dot(w_o, normal));
        Col fr;
        if (bx.fresnel_kind == CRAY_FRESNEL_DIELECTRIC) fr = WHITE * fresnel_dielectric(bx.eta_i, bx.eta_t, cos_theta_i);
        else fr = fresnel_conductor(cc(bx.c_eta_i), cc(bx.c_eta_t), cc(bx.c_k), cos_theta_i, assert_fail);
        out->w_i = w_i;
        out->f = tex_eval_color(sc, bx.tex_a, u, v) * fr / std::fabs(cos_theta_i);
        out->delta = true; out->pdf = 0.0; out->is_specular = true;
        return true;
    }
    case CRAY_BXDF_SPECULAR_BTDF: {
        double cos_theta_i = std::fabs(dot(w_o, normal));
        V3 w_i;
        if (!refract(w_o, normal, cos_theta_i, bx.eta_i, bx.eta_t, &w_i)) return false;
        if (!(std::fabs(magnitude(w_i) - 1.0) <= EPSILON)) *assert_fail += 1;
        double fr = fresnel_dielectric(bx.eta_i, bx.eta_t, cos_theta_i);
        out->w_i = w_i;
        out->f = tex_eval_color(sc, bx.tex_a, u, v) * (1.0 - fr) / cos_theta_i;
        out->delta = true; out->pdf = 0.0; out->is_specular = true;
        return true;
    }
    case CRAY_BXDF_FRESNEL_SPECULAR: {
        double cos_theta_i = dot(w_o, normal);
        double F = fresnel_dielectric(bx.eta_i, bx.eta_t, cos_theta_i);
        if (s0 < F) {
            out->w_i = reflect(w_o, normal);
            out->f = tex_eval_color(sc, bx.tex_a, u, v) * F / std::fabs(cos_theta_i);
            out->delta = false; out->pdf = F; out->is_specular = true;
            return true;
        }
        V3 w_i;
        if (!refract(w_o, normal, cos_theta_i, bx.eta_i, bx.eta_t, &w_i)) return false;
        out->w_i = w_i;
        out->f = tex_eval_color(sc, bx.tex_b, u, v) * (1.0 - F) / std::fabs(cos_theta_i);
        out->delta = false; out->pdf = 1.0 - F; out->is_specular = true;
        return true;
    }
    }
    return false;
}

/* The black matte of an AreaLightPrimitive (primitive.rs:40-46, material.rs:19-25):
 * Lambertian with reflectance Constant(BLACK) -> f == BLACK, pdf = |cos|/pi. */
static const int MATERIAL_AREA_LIGHT = -1;

/* Material::f, material.rs:84-89; BSDF::f, bsdf.rs:73-79 */
static Col material_f(const Scene& sc, int mat, V3 w_o, V3 w_i, V3 normal, double u, double v) {
    if (mat == MATERIAL_AREA_LIGHT) {
        return same_hemisphere(normal, w_o, w_i) ? BLACK * FRAC_1_PI : BLACK;
    }
    const cray_material& m = sc.materials[mat];
    if (!m.is_bsdf) return bxdf_f(sc, sc.bxdfs[m.first_bxdf], w_o, w_i, normal, u, v);
    bool is_reflecting = dot(w_o, normal) * dot(w_i, normal) > 0.0; /* bsdf.rs:60 */
    Col f = BLACK;
    for (int i = 0; i < m.n_bxdfs; i++) {
        const cray_bxdf& bx = sc.bxdfs[m.first_bxdf + i];
        bool relevant = is_reflecting ? bxdf_has_reflection(bx.kind) : bxdf_has_transmission(bx.kind);
        if (relevant) f = f + bxdf_f(sc, bx, w_o, w_i, normal, u, v);
    }
    return f;
}
/* Material::pdf, material.rs:90-95; BSDF::pdf, bsdf.rs:81-98. false == Pdf::Delta */
static bool material_pdf(const Scene& sc, int mat, V3 w_o, V3 w_i, V3 normal, double* pdf) {
    if (mat == MATERIAL_AREA_LIGHT) { *pdf = FRAC_1_PI * std::fabs(dot(w_i, normal)); return true; }
    const cray_material& m = sc.materials[mat];
    if (!m.is_bsdf) return bxdf_pdf(sc.bxdfs[m.first_bxdf], w_i, normal, pdf);
    bool is_reflecting = dot(w_o, normal) * dot(w_i, normal) > 0.0;
    double acc = 0.0;
    int n_match = 0;
    for (int i = 0; i < m.n_bxdfs; i++) {
        const cray_bxdf& bx = sc.bxdfs[m.first_bxdf + i];
        bool relevant = is_reflecting ? bxdf_has_reflection(bx.kind) : bxdf_has_transmission(bx.kind);
        if (!relevant) continue;
        double p;
        if (bxdf_pdf(bx, w_i, normal, &p)) { acc += p; n_match += 1; }
    }
    if (n_match > 0) { *pdf = acc / (double)n_match; return true; }
    return false;
}
/* Material::sample, material.rs:72-83; BSDF::sample, bsdf.rs:15-55 */
static bool material_sample(const Scene& sc, int mat, double s1d, double s0, double s1, V3 w_o, V3 normal, double u,
                            double v, SurfaceSample* out, int* assert_fail) {
    if (mat == MATERIAL_AREA_LIGHT) {
        V3 w_i = cosine_sample_hemisphere(s0, s1, normal, assert_fail);
        if (dot(normal, w_o) < 0.0) w_i = neg(w_i);
        out->w_i = w_i;
        out->f = same_hemisphere(normal, w_o, w_i) ? BLACK * FRAC_1_PI : BLACK;
        out->delta = false; out->pdf = FRAC_1_PI * std::fabs(dot(w_i, normal)); out->is_specular = false;
        return true;
    }
    const cray_material& m = sc.materials[mat];
    if (!m.is_bsdf) return bxdf_sample(sc, sc.bxdfs[m.first_bxdf], s0, s1, w_o, normal, u, v, out, assert_fail);
    if (m.n_bxdfs == 0) return false;
    int sample_index = (int)sat_u64(s1d * (double)m.n_bxdfs);
    const cray_bxdf& bx = sc.bxdfs[m.first_bxdf + sample_index];
    SurfaceSample s;
    if (!bxdf_sample(sc, bx, s0, s1, w_o, normal, u, v, &s, assert_fail)) return false;
    if (!s.delta) {
        double pdf = s.pdf;
        Col f = s.f;
        bool is_reflecting = dot(w_o, normal) * dot(s.w_i, normal) > 0.0;
        for (int i = 0; i < m.n_bxdfs; i++) {
            const cray_bxdf& other = sc.bxdfs[m.first_bxdf + i];
            bool relevant = is_reflecting ? bxdf_has_reflection(other.kind) : bxdf_has_transmission(other.kind);
            if (!relevant || i == sample_index) continue;
            f = f + bxdf_f(sc, other, w_o, s.w_i, normal, u, v);
            double op;
            if (bxdf_pdf(other, s.w_i, normal, &op)) pdf += op;
        }
        out->w_i = s.w_i; out->f = f; out->delta = false; out->pdf = pdf / (double)m.n_bxdfs;
        out->is_specular = s.is_specular;
        return true;
    }
    *out = s;
    return true;
}

/* ======================================================================== */
/* Lights + LightSampler, src/light.rs                                        */
/* ======================================================================== */
static Col light_Le(const Light& l) { return l.kind == CRAY_LIGHT_INFINITE ? l.c : BLACK; } /* :161-168 */

/* Light::pdf_Li, :136-143. false == Delta */
static bool light_pdf_Li(const Scene& sc, const Light& l, V3 isect_location, V3 isect_normal, V3 w_i, double* pdf) {
    switch (l.kind) {
    case CRAY_LIGHT_POINT:
    case CRAY_LIGHT_DISTANT: return false;
    case CRAY_LIGHT_INFINITE: *pdf = FRAC_1_PI / 4.0; return true;
    default: *pdf = shape_pdf_from(sc.shapes[l.prim], isect_location, isect_normal, w_i); return true;
    }
}
static Col light_power(const Scene& sc, const Light& l, double world_radius) { /* :170-177 */
    switch (l.kind) {
    case CRAY_LIGHT_POINT: return l.c * 4.0 * PI;
    case CRAY_LIGHT_DISTANT:
    case CRAY_LIGHT_INFINITE: return l.c * PI * world_radius * world_radius;
    default: return l.c * PI * shape_area(sc.shapes[l.prim]);
    }
}
static void light_sampler_new(Scene& sc, double world_radius) { /* :187-200 */
    double total_power = 0.0;
    sc.cdfs.clear();
    for (size_t i = 0; i < sc.lights.size(); i++) {
        Col power = light_power(sc, sc.lights[i], world_radius);
        double power_avg = (power.r + power.g + power.b) / 3.0;
        total_power += power_avg;
        sc.cdfs.push_back(total_power);
    }
    for (size_t i = 0; i < sc.cdfs.size(); i++) sc.cdfs[i] = sc.cdfs[i] / total_power;
}
static double light_sampler_pdf(const Scene& sc, size_t idx) { /* :213-219 */
    return idx > 0 ? sc.cdfs[idx] - sc.cdfs[idx - 1] : sc.cdfs[idx];
}
/* f64::total_cmp */
static inline int total_cmp(double a, double b) {
    int64_t x, y;
    memcpy(&x, &a, 8); memcpy(&y, &b, 8);
    x ^= (int64_t)(((uint64_t)(x >> 63)) >> 1);
    y ^= (int64_t)(((uint64_t)(y >> 63)) >> 1);
    return x < y ? -1 : (x > y ? 1 : 0);
}
/* LightSampler::sample, :203-211 — Rust slice::binary_search_by (std 1.5x-1.7x
 * implementation: size halves, `mid = left + size/2`).  With a strictly
 * increasing CDF every implementation returns the same index: exact match ->
 * its index, else the insertion point. */
static size_t light_sampler_sample(const Scene& sc, double u, double* pdf) {
    size_t lo = 0, hi = sc.cdfs.size();
    size_t idx = hi;
    bool found = false;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        int c = total_cmp(sc.cdfs[mid], u);
        if (c == 0) { idx = mid; found = true; break; }
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    if (!found) idx = lo;
    *pdf = light_sampler_pdf(sc, idx);
    return idx;
}

struct LightSample { Col Li; V3 w_i; bool delta; double pdf; Ray shadow_ray; };

/* Light::sample_Li, :59-133 */
static LightSample light_sample_Li(const Scene& sc, const Light& l, double s1d, double s0, double s1, V3 loc,
                                   V3 normal) {
    LightSample r;
    switch (l.kind) {
    case CRAY_LIGHT_POINT: {
        V3 op = l.v - loc;
        double dist_squared = magnitude_squared(op);
        double dist = std::sqrt(dist_squared);
        V3 w_i = op / dist;
        r.shadow_ray = ray_new(loc, w_i);
        update_max_distance(r.shadow_ray, dist);
        r.Li = l.c / dist_squared; r.w_i = w_i; r.delta = true; r.pdf = 0.0;
        return r;
    }
    case CRAY_LIGHT_DISTANT: {
        r.shadow_ray = ray_new(loc, l.v);
        r.Li = l.c; r.w_i = l.v; r.delta = true; r.pdf = 0.0;
        return r;
    }
    case CRAY_LIGHT_INFINITE: {
        V3 n = s1d < 0.5 ? v3(1, 0, 0) : v3(-1, 0, 0);
        V3 w_i = sample_hemisphere(s0, s1, n);
        r.shadow_ray = ray_new(loc, w_i);
        r.Li = l.c; r.w_i = w_i; r.delta = false; r.pdf = FRAC_1_PI / 4.0;
        return r;
    }
    default: {
        const Shape& s = sc.shapes[l.prim];
        /* Shape::sample_from, shape.rs:472-484 */
        V3 point = shape_sample(s, s0, s1);
        V3 w_i = normalized(point - loc);
        double pdf = shape_pdf_from(s, loc, normal, w_i);
        double distance = magnitude(point - loc);
        r.shadow_ray = ray_new(loc, w_i);
        update_max_distance(r.shadow_ray, distance - EPSILON);
        r.Li = l.c; r.w_i = w_i; r.delta = false; r.pdf = pdf;
        return r;
    }
    }
}

/* ======================================================================== */
/* BVH, src/bvh.rs                                                            */
/* ======================================================================== */
struct PrimInfo { uint32_t prim; Bounds bounds; V3 centroid; };

/* util::partition_by, src/util.rs:4-26 */
template <class T, class F>
static size_t partition_by(T* slice, size_t len, F pred) {
    if (len == 0) return 0;
    size_t left = 0, right = len - 1;
    while (left != right) {
        while (left < right && pred(slice[left])) left += 1;
        while (right > left && !pred(slice[right])) right -= 1;
        std::swap(slice[left], slice[right]);
    }
    return pred(slice[left]) ? left + 1 : left;
}

static uint32_t emit_leaf(Scene& sc, const PrimInfo* pi, size_t n, Bounds bounds) { /* bvh.rs:181-189 */
    Node nd; memset(&nd, 0, sizeof(nd));
    nd.bounds = bounds; nd.leaf = true; nd.first = (uint32_t)sc.prim_order.size(); nd.count = (uint32_t)n;
    for (size_t i = 0; i < n; i++) sc.prim_order.push_back(pi[i].prim);
    sc.nodes.push_back(nd);
    return (uint32_t)sc.nodes.size() - 1;
}

/* BvhNode::from_sah_splitting, bvh.rs:234-336 */
static uint32_t build_sah(Scene& sc, PrimInfo* pi, size_t n) {
    const size_t NUM_BUCKETS = 12;
    const double TRAVERSAL_TO_INTERSECTION_COST_RATIO = 1.0 / 8.0;
    const size_t MAX_LEAF_PRIMITIVES = 4;

    Bounds bounds = pi[0].bounds;
    for (size_t i = 1; i < n; i++) bounds = bounds_union(bounds, pi[i].bounds);
    if (n <= 1) return emit_leaf(sc, pi, n, bounds);

    double total_surface_area = bounds_surface_area(bounds);
    if (!(total_surface_area > 0.0)) { sc.build_error = 1; return emit_leaf(sc, pi, n, bounds); }

    Bounds cb = bounds_new(pi[0].centroid, pi[0].centroid);
    for (size_t i = 1; i < n; i++) cb = bounds_union(cb, bounds_new(pi[i].centroid, pi[i].centroid));
    int split_axis = bounds_maximum_extent(cb);

    auto bucket_idx = [&](const PrimInfo& p) -> size_t {
        double off = bounds_offset(cb, p.centroid)[split_axis];
        size_t idx = (size_t)sat_u64((double)NUM_BUCKETS * off);
        return idx < NUM_BUCKETS - 1 ? idx : NUM_BUCKETS - 1;
    };

    bool have[NUM_BUCKETS]; Bounds bb[NUM_BUCKETS]; size_t cnt[NUM_BUCKETS];
    for (size_t b = 0; b < NUM_BUCKETS; b++) { have[b] = false; cnt[b] = 0; }
    for (size_t i = 0; i < n; i++) {
        size_t b = bucket_idx(pi[i]);
        if (have[b]) { bb[b] = bounds_union(bb[b], pi[i].bounds); cnt[b] += 1; }
        else { have[b] = true; bb[b] = pi[i].bounds; cnt[b] = 1; }
    }
    double costs[NUM_BUCKETS - 1];
    for (size_t i = 0; i < NUM_BUCKETS - 1; i++) {
        double cost = TRAVERSAL_TO_INTERSECTION_COST_RATIO;
        for (int part = 0; part < 2; part++) {
            size_t lo = part == 0 ? 0 : i + 1, hi = part == 0 ? i + 1 : NUM_BUCKETS;
            bool m_have = false; Bounds m_b = bb[0]; size_t m_c = 0;
            for (size_t b = lo; b < hi; b++) {
                if (!have[b]) continue;
                if (m_have) { m_b = bounds_union(m_b, bb[b]); m_c += cnt[b]; }
                else { m_have = true; m_b = bb[b]; m_c = cnt[b]; }
            }
            if (m_have) cost += (double)m_c * bounds_surface_area(m_b) / total_surface_area;
        }
        if (!std::isfinite(cost)) sc.build_error = 2;
        costs[i] = cost;
    }
    size_t min_idx = 0;
    for (size_t i = 0; i < NUM_BUCKETS - 1; i++)
        if (costs[i] < costs[min_idx]) min_idx = i;

    double leaf_cost = (double)n;
    if (leaf_cost <= costs[min_idx] && n <= MAX_LEAF_PRIMITIVES) return emit_leaf(sc, pi, n, bounds);

    size_t split = partition_by(pi, n, [&](const PrimInfo& p) { return bucket_idx(p) <= min_idx; });
    if (split == 0 || split == n) { /* reference: assert!(left.len() > 0 && right.len() > 0) -> panic */
        sc.build_error = 3;
        return emit_leaf(sc, pi, n, bounds);
    }
    Node nd; memset(&nd, 0, sizeof(nd));
    nd.bounds = bounds; nd.leaf = false; nd.axis = split_axis;
    uint32_t me = (uint32_t)sc.nodes.size();
    sc.nodes.push_back(nd);
    uint32_t l = build_sah(sc, pi, split);
    uint32_t r = build_sah(sc, pi + split, n - split);
    sc.nodes[me].left = l; sc.nodes[me].right = r;
    return me;
}

/* BvhNode::from_median_splitting, bvh.rs:191-230.  Rust's select_nth_unstable_by
 * leaves an implementation-defined permutation; only the n <= 4 (single leaf) case
 * used by tests/test_bvh.rs is pinned, larger inputs use std::nth_element. */
static uint32_t build_median(Scene& sc, PrimInfo* pi, size_t n) {
    Bounds bounds = pi[0].bounds;
    for (size_t i = 1; i < n; i++) bounds = bounds_union(bounds, pi[i].bounds);
    if (n <= 4) return emit_leaf(sc, pi, n, bounds);
    Bounds cb = bounds_new(pi[0].centroid, pi[0].centroid);
    for (size_t i = 1; i < n; i++) cb = bounds_union(cb, bounds_new(pi[i].centroid, pi[i].centroid));
    int axis = bounds_maximum_extent(cb);
    if (cb.mn[axis] == cb.mx[axis]) return emit_leaf(sc, pi, n, bounds);
    size_t mid = (n - 1) / 2;
    std::nth_element(pi, pi + mid, pi + n,
                     [&](const PrimInfo& a, const PrimInfo& b) { return total_cmp(a.centroid[axis], b.centroid[axis]) < 0; });
    Node nd; memset(&nd, 0, sizeof(nd));
    nd.bounds = bounds; nd.leaf = false; nd.axis = axis;
    uint32_t me = (uint32_t)sc.nodes.size();
    sc.nodes.push_back(nd);
    if (mid == 0) mid = 1; /* split_at_mut(mid) with mid == 0 would recurse on an empty slice and panic */
    uint32_t l = build_median(sc, pi, mid);
    uint32_t r = build_median(sc, pi + mid, n - mid);
    sc.nodes[me].left = l; sc.nodes[me].right = r;
    return me;
}

/* Bvh::new, bvh.rs:38-56 */
static void bvh_new(Scene& sc, int split_method) {
    std::vector<PrimInfo> infos(sc.prims.size());
    for (size_t i = 0; i < sc.prims.size(); i++) {
        infos[i].prim = (uint32_t)i;
        infos[i].bounds = shape_bounds(sc.shapes[i]);
        infos[i].centroid = bounds_centroid(shape_bounds(sc.shapes[i]));
    }
    sc.nodes.clear(); sc.prim_order.clear(); sc.build_error = 0;
    if (infos.empty()) { sc.build_error = 4; return; }
    sc.nodes.reserve(infos.size() * 2);
    sc.prim_order.reserve(infos.size());
    if (split_method == 0) build_median(sc, infos.data(), infos.size());
    else build_sah(sc, infos.data(), infos.size());
    Bounds b = infos[0].bounds;
    for (size_t i = 1; i < infos.size(); i++) b = bounds_union(b, infos[i].bounds);
    sc.bvh_bounds = b;
}

/* Primitive::intersect, primitive.rs:50-73 */
static inline bool primitive_intersect(const Scene& sc, uint32_t prim, Ray& ray, Hit* h) {
    if (!shape_intersect(sc.shapes[prim], ray, h)) return false;
    h->distance = ray.tmax;
    h->prim = (int)prim;
    return true;
}

/* Bvh::intersect, bvh.rs:58-104 */
static bool bvh_intersect(const Scene& sc, Ray& ray, Hit* out, Stats* st) {
    std::vector<uint32_t> q;
    q.reserve(64);
    q.push_back(0);
    bool have = false;
    Hit current; memset(&current, 0, sizeof(current));
    st->closest_rays += 1;
    while (!q.empty()) {
        uint32_t ni = q.back(); q.pop_back();
        const Node& node = sc.nodes[ni];
        st->closest_nodes += 1;
        if (!bounds_intersects(node.bounds, ray) && !bounds_contains(node.bounds, ray.o)) continue;
        if (node.leaf) {
            for (uint32_t k = 0; k < node.count; k++) {
                uint32_t p = sc.prim_order[node.first + k];
                st->closest_prims += 1;
                if (sc.shapes[p].kind == CRAY_SHAPE_TRIANGLE) st->closest_tri_tests += 1;
                Hit h;
                if (primitive_intersect(sc, p, ray, &h)) {
                    if (!have || h.distance < current.distance) { current = h; have = true; }
                }
            }
        } else {
            if (ray.d[node.axis] < 0.0) { q.push_back(node.left); q.push_back(node.right); }
            else { q.push_back(node.right); q.push_back(node.left); }
        }
    }
    if (have) { *out = current; st->closest_hits += 1; }
    return have;
}

/* Bvh::intersects, bvh.rs:106-147 */
static bool bvh_intersects(const Scene& sc, const Ray& ray, Stats* st) {
    std::vector<uint32_t> q;
    q.reserve(64);
    q.push_back(0);
    st->shadow_rays += 1;
    while (!q.empty()) {
        uint32_t ni = q.back(); q.pop_back();
        const Node& node = sc.nodes[ni];
        st->shadow_nodes += 1;
        if (!bounds_intersects(node.bounds, ray) && !bounds_contains(node.bounds, ray.o)) continue;
        if (node.leaf) {
            for (uint32_t k = 0; k < node.count; k++) {
                uint32_t p = sc.prim_order[node.first + k];
                st->shadow_prims += 1;
                if (sc.shapes[p].kind == CRAY_SHAPE_TRIANGLE) st->shadow_tri_tests += 1;
                if (shape_intersects(sc.shapes[p], ray)) return true;
            }
        } else {
            if (ray.d[node.axis] < 0.0) { q.push_back(node.left); q.push_back(node.right); }
            else { q.push_back(node.right); q.push_back(node.left); }
        }
    }
    return false;
}

/* ======================================================================== */
/* Camera, src/camera.rs                                                      */
/* ======================================================================== */
static void camera_new(Scene& sc, const cray_camera_desc& c) {
    sc.cam_type = c.type;
    sc.W = c.film_width; sc.H = c.film_height;
    sc.lens_radius = c.lens_radius; sc.focal_distance = c.focal_distance;
    Xf screen_from_camera = c.type == CRAY_CAMERA_PERSPECTIVE ? xf_perspective(c.fov, 1e-2, 1000.0) /* :87-92 */
                                                              : xf_orthographic(0.0, 1.0);          /* :114-117 */
    sc.world_from_camera = xf_look_at(cv(c.origin), cv(c.target), cv(c.up));                          /* :66 */
    /* get_camera_from_raster_transformation, :25-53 (film_height = film.width, sic :30) */
    double film_width = (double)c.film_width;
    double film_height = (double)c.film_width;
    double screen_width, screen_height;
    if (film_width > film_height) { screen_width = film_width / film_height; screen_height = 1.0; }
    else { screen_width = 1.0; screen_height = film_height / film_width; }
    Xf screen_from_raster = xf_mul(xf_scale(2.0 * screen_width / film_width, -2.0 * screen_height / film_height, 1.0),
                                   xf_translate(-film_width / 2.0, -film_height / 2.0, 0.0));
    sc.camera_from_raster = xf_mul(xf_inverse(screen_from_camera), screen_from_raster);
}
/* Camera::sample + generate_ray, :131-162 */
static Ray camera_sample(const Scene& sc, double fx, double fy, double lx, double ly, uint64_t rx, uint64_t ry) {
    double dx = 2.0 * fx - 1.0, dy = 2.0 * fy - 1.0;
    V3 p_raster = v3((double)rx + dx, (double)ry + dy, 0.0);
    V3 p_camera = xf_point(sc.camera_from_raster, p_raster);
    Ray ray = sc.cam_type == CRAY_CAMERA_PERSPECTIVE ? ray_new(p_camera, normalized(p_camera - v3(0, 0, 0)))
                                                     : ray_new(p_camera, v3(0, 0, 1));
    if (sc.lens_radius != 0.0) {
        double lens_x = 2.0 * lx - 1.0, lens_y = 2.0 * ly - 1.0;
        V3 p_lens = v3(lens_x * sc.lens_radius, lens_y * sc.lens_radius, 0.0);
        V3 p_focal = ray_at(ray, sc.focal_distance / ray.d.z);
        ray = ray_new(p_lens, normalized(p_focal - p_lens));
    }
    return xf_ray(sc.world_from_camera, ray);
}

/* ======================================================================== */
/* path_integrator::estimate_Li, src/path_integrator.rs:41-215                */
/* ======================================================================== */
/* `log` (optional, tests only): 20 doubles per loop iteration that reaches a hit:
 * prim, t, location[3], normal[3], shadow_occluded, L[3] after NEE, w_i[3], beta[3], continued, nee_light_pdf */
static Col estimate_Li(const Scene& sc, Sampler& sampler, Ray ray, Stats* st, double* log = nullptr, int log_cap = 0, int* log_n = nullptr) {
    Col L = BLACK, beta = WHITE;
    uint32_t bounces = 0;
    bool is_specular_bounce = true;
    double prev_bsdf_pdf = 0.0;
    V3 prev_loc = v3(0, 0, 0), prev_normal = v3(0, 0, 0);
    int af = 0;

    while (bounces < sc.max_depth && !is_black(beta)) {
        V3 w_o = neg(ray.d);
        Hit isect;
        if (!bvh_intersect(sc, ray, &isect, st)) {
            if (is_specular_bounce) { /* :64-67 */
                for (size_t i = 0; i < sc.lights.size(); i++) L = L + beta * light_Le(sc.lights[i]);
            } else { /* :68-88 */
                for (size_t i = 0; i < sc.lights.size(); i++) {
                    Col Le = light_Le(sc.lights[i]);
                    if (!is_black(Le)) {
                        double lp = 0.0;
                        light_pdf_Li(sc, sc.lights[i], prev_loc, prev_normal, w_o, &lp);
                        double light_pdf = lp * light_sampler_pdf(sc, i);
                        double weight = power_heuristic(light_pdf, prev_bsdf_pdf);
                        L = L + beta * Le * weight;
                    }
                }
            }
            break;
        }
        V3 normal = isect.normal, location = isect.location;
        double tu = isect.u, tv = isect.v;
        const cray_prim& prim = sc.prims[isect.prim];
        int material = prim.light >= 0 ? MATERIAL_AREA_LIGHT : prim.material;

        /* PathSegmentSamples::from, :26-36 (struct-literal field order) */
        double m1 = sampler.sample_1d();
        double m2a, m2b; sampler.sample_2d(&m2a, &m2b);
        double li_idx = sampler.sample_1d();
        double l1 = sampler.sample_1d();
        double l2a, l2b; sampler.sample_2d(&l2a, &l2b);
        double rr = sampler.sample_1d();

        /* emission on hit, :106-126 */
        if (prim.light >= 0) {
            const Light& light = sc.lights[prim.light];
            Col Le = light.c; /* Light::L for Area, light.rs:147-157 */
            if (!is_black(Le)) {
                if (is_specular_bounce) {
                    L = L + beta * Le;
                } else {
                    int light_idx = sc.first_equal_light[prim.light]; /* position(|l| l == light), :116 */
                    double lp = 0.0;
                    light_pdf_Li(sc, light, location, normal, w_o, &lp);
                    double light_pdf = lp * light_sampler_pdf(sc, (size_t)light_idx);
                    double weight = power_heuristic(light_pdf, prev_bsdf_pdf);
                    L = L + beta * Le * weight;
                }
            }
        }

        double* rec = (log && log_n && *log_n < log_cap) ? log + 20 * (*log_n) : nullptr;
        if (rec) {
            for (int k = 0; k < 20; k++) rec[k] = 0.0;
            rec[0] = (double)isect.prim; rec[1] = isect.distance;
            rec[2] = location.x; rec[3] = location.y; rec[4] = location.z;
            rec[5] = normal.x; rec[6] = normal.y; rec[7] = normal.z;
            *log_n += 1;
        }
        /* next-event estimation, :129-164 */
        {
            double light_sampler_pdf_v;
            size_t light_index = light_sampler_sample(sc, li_idx, &light_sampler_pdf_v);
            const Light& light = sc.lights[light_index];
            LightSample ls = light_sample_Li(sc, light, l1, l2a, l2b, location, normal);
            bool occluded = bvh_intersects(sc, ls.shadow_ray, st);
            if (rec) { rec[8] = occluded ? 1.0 : 0.0; rec[19] = ls.pdf; }
            if (!occluded) {
                Col f = material_f(sc, material, w_o, ls.w_i, normal, tu, tv);
                double cos_theta = std::fabs(dot(ls.w_i, normal));
                if (!ls.delta) {
                    if (ls.pdf > 0.0) {
                        double light_pdf = ls.pdf * light_sampler_pdf_v;
                        double bsdf_pdf = 0.0;
                        if (!material_pdf(sc, material, w_o, ls.w_i, normal, &bsdf_pdf)) bsdf_pdf = 0.0;
                        double weight = power_heuristic(light_pdf, bsdf_pdf);
                        L = L + beta * ls.Li * f * cos_theta * weight / light_pdf;
                    }
                } else {
                    double light_pdf = light_sampler_pdf_v;
                    L = L + beta * ls.Li * f * cos_theta / light_pdf;
                }
            }
        }

        if (rec) { rec[9] = L.r; rec[10] = L.g; rec[11] = L.b; }
        /* BSDF sampling, :167-195 */
        {
            SurfaceSample ss;
            if (!material_sample(sc, material, m1, m2a, m2b, w_o, normal, tu, tv, &ss, &af)) break;
            if (is_black(ss.f)) break;
            double cos_theta = std::fabs(dot(ss.w_i, normal));
            double bsdf_pdf = ss.delta ? 1.0 : ss.pdf;
            if (bsdf_pdf == 0.0) break;
            beta = beta * ss.f * cos_theta / bsdf_pdf;
            ray = ray_new(location, ss.w_i);
            is_specular_bounce = ss.is_specular;
            prev_bsdf_pdf = bsdf_pdf;
            prev_loc = location; prev_normal = normal;
            if (rec) { rec[12] = ss.w_i.x; rec[13] = ss.w_i.y; rec[14] = ss.w_i.z; rec[15] = beta.r; rec[16] = beta.g; rec[17] = beta.b; rec[18] = 1.0; }
        }

        /* Russian roulette, :197-206 */
        if (bounces > 0) {
            double max_beta = rmax(beta.r, rmax(beta.g, beta.b));
            if (max_beta < 1.0) {
                double q = 1.0 - max_beta;
                if (rr < q) break;
                beta = beta / (1.0 - q);
            }
        }
        if (!is_finite(L) || !is_finite(beta)) st->nonfinite += 1; /* reference: assert! -> panic, :208-209 */
        bounces += 1;
    }
    st->assert_fail += (uint64_t)af;
    st->paths += 1;
    return L;
}

/* simple_integrator::estimate_Li, src/simple_integrator.rs:36-143: direct lighting by one light sample per segment, no
 * multiple importance sampling, no Russian roulette; seven samples per segment (:26-32, struct-literal field order). */
static Col estimate_Li_simple(const Scene& sc, Sampler& sampler, Ray ray, Stats* st) {
    Col L = BLACK, beta = WHITE;
    uint32_t bounces = 0;
    bool is_specular_bounce = true;
    int af = 0;
    while (bounces < sc.max_depth && !is_black(beta)) {
        V3 w_o = neg(ray.d);
        Hit isect;
        if (!bvh_intersect(sc, ray, &isect, st)) {
            if (is_specular_bounce) /* :57-61 */
                for (size_t i = 0; i < sc.lights.size(); i++) L = L + beta * light_Le(sc.lights[i]);
            break;
        }
        V3 normal = isect.normal, location = isect.location;
        double tu = isect.u, tv = isect.v;
        const cray_prim& prim = sc.prims[isect.prim];
        int material = prim.light >= 0 ? MATERIAL_AREA_LIGHT : prim.material;

        double m1 = sampler.sample_1d();
        double m2a, m2b; sampler.sample_2d(&m2a, &m2b);
        double li_idx = sampler.sample_1d();
        double l1 = sampler.sample_1d();
        double l2a, l2b; sampler.sample_2d(&l2a, &l2b);

        if (is_specular_bounce && prim.light >= 0) L = L + beta * sc.lights[prim.light].c; /* :84-86, PrimitiveIntersection::Le */

        { /* :89-112 */
            double light_sampler_pdf_v;
            size_t light_index = light_sampler_sample(sc, li_idx, &light_sampler_pdf_v);
            const Light& light = sc.lights[light_index];
            LightSample ls = light_sample_Li(sc, light, l1, l2a, l2b, location, normal);
            double light_pdf = ls.delta ? 1.0 : ls.pdf;
            if (light_pdf > 0.0 && !bvh_intersects(sc, ls.shadow_ray, st)) {
                Col f = material_f(sc, material, w_o, ls.w_i, normal, tu, tv);
                double cos_theta = std::fabs(dot(ls.w_i, normal));
                L = L + beta * ls.Li * f * cos_theta / light_sampler_pdf_v / light_pdf;
            }
        }
        { /* :115-137 */
            SurfaceSample ss;
            if (!material_sample(sc, material, m1, m2a, m2b, w_o, normal, tu, tv, &ss, &af)) break;
            if (is_black(ss.f)) break;
            double cos_theta = std::fabs(dot(ss.w_i, normal));
            double bsdf_pdf = ss.delta ? 1.0 : ss.pdf;
            if (bsdf_pdf == 0.0) break;
            beta = beta * ss.f * cos_theta / bsdf_pdf;
            ray = ray_new(location, ss.w_i);
            is_specular_bounce = ss.is_specular;
        }
        if (!is_finite(L) || !is_finite(beta)) st->nonfinite += 1; /* :139-140 */
        bounces += 1;
    }
    st->assert_fail += (uint64_t)af;
    st->paths += 1;
    return L;
}

/* render_pixel, src/bin/craytracer.rs:148-162 */
static Col render_pixel(const Scene& sc, Sampler& sampler, uint64_t x, uint64_t y, uint64_t s, Stats* st) {
    sampler.start_pixel(x, y, s);
    double fx, fy, lx, ly;
    sampler.sample_2d(&fx, &fy);
    sampler.sample_2d(&lx, &ly);
    Ray ray = camera_sample(sc, fx, fy, lx, ly, x, y);
    return g_integrator == 1 ? estimate_Li_simple(sc, sampler, ray, st) : estimate_Li(sc, sampler, ray, st);
}

/* ======================================================================== */
/* Scene construction (Scene::new, src/scene.rs:25-53)                        */
/* ======================================================================== */
static Scene* scene_create(const cray_scene_desc* d, int split_method) {
    Scene* sc = new Scene();
    sc->d = *d;
    sc->prims.assign(d->prims, d->prims + d->n_prims);
    sc->materials.assign(d->materials, d->materials + d->n_materials);
    sc->bxdfs.assign(d->bxdfs, d->bxdfs + d->n_bxdfs);
    sc->textures.assign(d->textures, d->textures + d->n_textures);
    sc->images.assign(d->images, d->images + d->n_images);
    sc->pool.assign(d->image_pool, d->image_pool + d->image_pool_bytes);
    sc->max_depth = d->max_depth; sc->num_samples = d->num_samples;
    sc->shapes.resize(d->n_prims);
    for (uint32_t i = 0; i < d->n_prims; i++) {
        const cray_prim& p = d->prims[i];
        if (p.shape_kind == CRAY_SHAPE_SPHERE) {
            const cray_sphere_desc& s = d->spheres[p.shape];
            sc->shapes[i] = make_sphere(cv(s.origin), s.radius);
        } else if (p.shape_kind == CRAY_SHAPE_DISK) {
            const cray_disk_desc& s = d->disks[p.shape];
            sc->shapes[i] = make_disk(cv(s.origin), s.rotate_x, s.rotate_y, s.radius, s.inner_radius);
        } else {
            Shape s; memset(&s, 0, sizeof(s));
            s.kind = CRAY_SHAPE_TRIANGLE; s.tri = d->triangles[p.shape];
            sc->shapes[i] = s;
        }
    }
    sc->lights.resize(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; i++) {
        Light l = {d->lights[i].kind, d->lights[i].prim, cv(d->lights[i].v), cc(d->lights[i].c)};
        sc->lights[i] = l;
    }
    camera_new(*sc, d->camera);
    bvh_new(*sc, split_method);
    double world_radius = magnitude(bounds_diagonal(sc->bvh_bounds)) * 0.5; /* scene.rs:42 */
    light_sampler_new(*sc, world_radius);
    /* `scene.lights.iter().position(|l| l == light)` (path_integrator.rs:116): index of the
     * first light equal *by value*.  Keyed on the value bytes (-0.0 folded onto +0.0 so that
     * byte equality == f64 `==`; NaNs are not expected) to stay O(n log n) for emissive meshes. */
    sc->first_equal_light.resize(sc->lights.size());
    {
        std::map<std::string, int> seen;
        for (size_t i = 0; i < sc->lights.size(); i++) {
            const Light& l = sc->lights[i];
            std::vector<double> key;
            key.push_back((double)l.kind);
            key.push_back(l.c.r); key.push_back(l.c.g); key.push_back(l.c.b);
            if (l.kind == CRAY_LIGHT_POINT || l.kind == CRAY_LIGHT_DISTANT) {
                key.push_back(l.v.x); key.push_back(l.v.y); key.push_back(l.v.z);
            } else if (l.kind == CRAY_LIGHT_AREA) {
                const Shape& sh = sc->shapes[l.prim];
                key.push_back((double)sh.kind);
                if (sh.kind == CRAY_SHAPE_TRIANGLE) {
                    const double* t = (const double*)&sh.tri;
                    for (int k = 0; k < 24; k++) key.push_back(t[k]);
                } else {
                    key.push_back(sh.radius);
                    if (sh.kind == CRAY_SHAPE_DISK) key.push_back(sh.inner_radius);
                    const Mat* ms[4] = {&sh.xf->o2w.matrix, &sh.xf->o2w.inverse, &sh.xf->w2o.matrix, &sh.xf->w2o.inverse};
                    for (int k = 0; k < 4; k++) for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) key.push_back(ms[k]->m[a][b]);
                }
            }
            for (auto& x : key) if (x == 0.0) x = 0.0;
            std::string sk((const char*)key.data(), key.size() * sizeof(double));
            auto it = seen.find(sk);
            if (it == seen.end()) { seen.emplace(sk, (int)i); sc->first_equal_light[i] = (int)i; }
            else sc->first_equal_light[i] = it->second;
        }
    }
    return sc;
}

/* render, src/bin/craytracer.rs:224-319 with generate_tiles :22-43, render_tile :164-206.
 * Deterministic variant: per pixel the sample batches are added in ascending
 * order (the reference's thread completion order is unspecified). */
/* Edge of the square pixel tiles one job covers: 64 like the reference (craytracer.rs:232-233).  The film does not depend on it
 * (per pixel the batches are accumulated in ascending order either way); bench.py's CPU baseline lowers it so that a bounded
 * sample of the frame still gives every host thread many jobs (the reference gets them from tiles x sample batches). */
static uint32_t g_tile = 64;
static void render(const Scene& sc, uint64_t seed, int n_threads, uint32_t s_begin, uint32_t s_end, float* out,
                   Stats* total, double* seconds) {
    const uint32_t W = sc.W, H = sc.H, TILE = g_tile, BATCH = 8;
    std::vector<float> pixels((size_t)W * H * 3, 0.0f);
    uint32_t tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
    std::atomic<uint32_t> next(0);
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    std::vector<Stats> stats((size_t)n_threads);
    for (auto& s : stats) memset(&s, 0, sizeof(Stats));
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        Sampler sampler; sampler.seed = seed; sampler.hash = 0; sampler.sample_index = 0; sampler.dimension = 0;
        for (;;) {
            uint32_t t = next.fetch_add(1);
            if (t >= tiles_x * tiles_y) break;
            uint32_t tx = (t % tiles_x) * TILE, ty = (t / tiles_x) * TILE;
            uint32_t x1 = std::min(tx + TILE, W), y1 = std::min(ty + TILE, H);
            /* batches are aligned to multiples of 8 from 0, as generate_tiles does */
            for (uint32_t si = (s_begin / BATCH) * BATCH; si < s_end; si += BATCH) {
                uint32_t b0 = std::max(si, s_begin), b1 = std::min(std::min(si + BATCH, s_end), sc.num_samples);
                for (uint32_t y = ty; y < y1; y++)
                    for (uint32_t x = tx; x < x1; x++) {
                        Col color = BLACK;
                        for (uint32_t s = b0; s < b1; s++) color = color + render_pixel(sc, sampler, x, y, s, &stats[tid]);
                        size_t off = (size_t)x + (size_t)y * W;
                        pixels[3 * off] += (float)color.r;
                        pixels[3 * off + 1] += (float)color.g;
                        pixels[3 * off + 2] += (float)color.b;
                    }
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < n_threads; i++) th.emplace_back(worker, i);
    for (auto& t : th) t.join();
    /* on_finish, :253-259 */
    for (size_t i = 0; i < pixels.size(); i++) out[i] = pixels[i] / (float)sc.num_samples;
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    Stats tot; memset(&tot, 0, sizeof(tot));
    for (auto& s : stats) tot.add(s);
    if (total) *total = tot;
}

} /* namespace orc */

/* ========================================================================== */
/* C API (ctypes)                                                               */
/* ========================================================================== */
using namespace orc;

extern "C" {

typedef struct {
    uint64_t closest_rays, shadow_rays, closest_nodes, closest_prims, shadow_nodes, shadow_prims;
    uint64_t closest_hits, closest_tri_tests, shadow_tri_tests, paths, nonfinite, assert_fail;
    double seconds;
} orc_stats;

typedef struct {
    int32_t hit, prim;
    double t;
    double location[3], normal[3], uv[2];
} orc_hit;

typedef struct {
    double bmin[3], bmax[3];
    uint32_t left, right, first, count;
    int32_t axis, leaf;
} orc_node;

static void copy_stats(const Stats& s, double seconds, orc_stats* o) {
    if (!o) return;
    o->closest_rays = s.closest_rays; o->shadow_rays = s.shadow_rays;
    o->closest_nodes = s.closest_nodes; o->closest_prims = s.closest_prims;
    o->shadow_nodes = s.shadow_nodes; o->shadow_prims = s.shadow_prims;
    o->closest_hits = s.closest_hits; o->closest_tri_tests = s.closest_tri_tests;
    o->shadow_tri_tests = s.shadow_tri_tests; o->paths = s.paths; o->nonfinite = s.nonfinite;
    o->assert_fail = s.assert_fail; o->seconds = seconds;
}

/* split_method: 0 = Median, 1 = SAH (Scene::new always uses SAH, scene.rs:38) */
void* orc_scene_create(const cray_scene_desc* d, int split_method) { return scene_create(d, split_method); }
void orc_scene_destroy(void* s) { delete (Scene*)s; }
int orc_scene_build_error(void* s) { return ((Scene*)s)->build_error; }

uint32_t orc_bvh_num_nodes(void* s) { return (uint32_t)((Scene*)s)->nodes.size(); }
uint32_t orc_bvh_num_prim_refs(void* s) { return (uint32_t)((Scene*)s)->prim_order.size(); }
void orc_bvh_export(void* s, orc_node* nodes, uint32_t* prim_order) {
    Scene* sc = (Scene*)s;
    for (size_t i = 0; i < sc->nodes.size(); i++) {
        const Node& n = sc->nodes[i];
        orc_node& o = nodes[i];
        o.bmin[0] = n.bounds.mn.x; o.bmin[1] = n.bounds.mn.y; o.bmin[2] = n.bounds.mn.z;
        o.bmax[0] = n.bounds.mx.x; o.bmax[1] = n.bounds.mx.y; o.bmax[2] = n.bounds.mx.z;
        o.left = n.left; o.right = n.right; o.first = n.first; o.count = n.count; o.axis = n.axis; o.leaf = n.leaf;
    }
    memcpy(prim_order, sc->prim_order.data(), sc->prim_order.size() * 4);
}
void orc_scene_light_cdf(void* s, double* cdf) {
    Scene* sc = (Scene*)s;
    memcpy(cdf, sc->cdfs.data(), sc->cdfs.size() * 8);
}
void orc_scene_camera(void* s, double* camera_from_raster16, double* world_from_camera16) {
    Scene* sc = (Scene*)s;
    memcpy(camera_from_raster16, sc->camera_from_raster.matrix.m, 128);
    memcpy(world_from_camera16, sc->world_from_camera.matrix.m, 128);
}

/* Scene::intersect / Scene::intersects over a batch. rays: n x {o[3], d[3], tmax} */
void orc_trace(void* s, const double* rays, uint64_t n, int any_hit, orc_hit* hits, orc_stats* stats) {
    Scene* sc = (Scene*)s;
    Stats st; memset(&st, 0, sizeof(st));
    for (uint64_t i = 0; i < n; i++) {
        const double* r = rays + 7 * i;
        Ray ray = {v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), r[6]};
        orc_hit& o = hits[i];
        memset(&o, 0, sizeof(o));
        o.prim = -1;
        if (any_hit) {
            o.hit = bvh_intersects(*sc, ray, &st) ? 1 : 0;
        } else {
            Hit h;
            if (bvh_intersect(*sc, ray, &h, &st)) {
                o.hit = 1; o.prim = h.prim; o.t = h.distance;
                o.location[0] = h.location.x; o.location[1] = h.location.y; o.location[2] = h.location.z;
                o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
                o.uv[0] = h.u; o.uv[1] = h.v;
            }
        }
    }
    copy_stats(st, 0.0, stats);
}

void orc_render(void* s, uint64_t seed, int n_threads, uint32_t s_begin, uint32_t s_end, float* out_rgb,
                orc_stats* stats) {
    Scene* sc = (Scene*)s;
    Stats st; double sec = 0.0;
    render(*sc, seed, n_threads, s_begin, s_end, out_rgb, &st, &sec);
    copy_stats(st, sec, stats);
}

/* One path: radiance of (x, y, sample_index) -> L[3] (f64), for per-path parity checks */
void orc_render_pixel(void* s, uint64_t seed, uint32_t x, uint32_t y, uint32_t sample, double* L) {
    Scene* sc = (Scene*)s;
    Sampler sampler; sampler.seed = seed;
    Stats st; memset(&st, 0, sizeof(st));
    Col c = render_pixel(*sc, sampler, x, y, sample, &st);
    L[0] = c.r; L[1] = c.g; L[2] = c.b;
}
/* test-only: per-bounce log of one path (see estimate_Li) */
int orc_path_log(void* s, uint64_t seed, uint32_t x, uint32_t y, uint32_t sample, double* rec, int cap) {
    Scene* sc = (Scene*)s;
    Sampler sampler; sampler.seed = seed;
    Stats st; memset(&st, 0, sizeof(st));
    sampler.start_pixel(x, y, sample);
    double fx, fy, lx, ly;
    sampler.sample_2d(&fx, &fy); sampler.sample_2d(&lx, &ly);
    Ray ray = camera_sample(*sc, fx, fy, lx, ly, x, y);
    int n = 0;
    estimate_Li(*sc, sampler, ray, &st, rec, cap, &n);
    return n;
}
/* Camera ray of (x, y, sample): {o[3], d[3], tmax} */
void orc_camera_ray(void* s, uint64_t seed, uint32_t x, uint32_t y, uint32_t sample, double* ray7) {
    Scene* sc = (Scene*)s;
    Sampler sampler; sampler.seed = seed;
    sampler.start_pixel(x, y, sample);
    double fx, fy, lx, ly;
    sampler.sample_2d(&fx, &fy); sampler.sample_2d(&lx, &ly);
    Ray r = camera_sample(*sc, fx, fy, lx, ly, x, y);
    ray7[0] = r.o.x; ray7[1] = r.o.y; ray7[2] = r.o.z; ray7[3] = r.d.x; ray7[4] = r.d.y; ray7[5] = r.d.z; ray7[6] = r.tmax;
}

/* job granularity of orc_render (pixels per tile edge; 0 restores the reference's 64) */
void orc_set_tile(uint32_t tile) { g_tile = tile ? tile : 64; }
/* 0: correctly rounded sin/cos in the sampling functions (default); 1: platform libm */
void orc_set_libm_mode(int mode) { g_libm_mode = mode; }
/* the first n draws of IndependentSampler for pixel sample (seed, x, y, sample_index) */
void orc_independent_draws(uint64_t seed, uint64_t x, uint64_t y, uint64_t sample_index, int n, double* out) {
    StdRngRestated r;
    uint64_t w[4] = {seed, x, y, sample_index};
    r.seed_from_u64(siphash13(w, 4, 0, 0));
    for (int i = 0; i < n; i++) out[i] = r.uniform01();
}

/* integrator: 0 path, 1 simple; sampler: 0 Sobol, 1 Uniform(nx, ny), 2 Independent */
void orc_set_mode(int integrator, int sampler, uint64_t nx, uint64_t ny) {
    g_integrator = integrator; g_sampler_kind = sampler;
    g_uniform_nx = nx ? nx : 1; g_uniform_ny = ny ? ny : 1;
}
/* 64 x 16 x 4 bit-reversed direction vectors (sobol_burley's REV_VECTORS layout); NULL restores the built-in table */
void orc_set_sobol_vectors(const uint16_t* v) {
    if (!v) { g_sobol_table = CRAY_SOBOL_REV_VECTORS; return; }
    memcpy(g_sobol_override, v, sizeof(g_sobol_override));
    g_sobol_table = g_sobol_override;
}
double orc_sample_sin(double x) { return sample_sin(x); }
double orc_sample_cos(double x) { return sample_cos(x); }

/* ---- unit hooks for the reference's known-answer tests ------------------- */
uint64_t orc_siphash(const uint8_t* msg, uint64_t len, uint64_t k0, uint64_t k1, int c, int d) {
    return siphash_cd(msg, (size_t)len, k0, k1, c, d);
}
uint32_t orc_pixel_hash(uint64_t seed, uint64_t x, uint64_t y) { return pixel_hash(seed, x, y); }
float orc_sobol_sample(uint32_t index, uint32_t dim, uint32_t seed) { return sobol_sample(index, dim, seed); }

int orc_bounds_intersects(const double* bmin, const double* bmax, const double* ray7) {
    Bounds b = {v3(bmin[0], bmin[1], bmin[2]), v3(bmax[0], bmax[1], bmax[2])};
    Ray r = {v3(ray7[0], ray7[1], ray7[2]), v3(ray7[3], ray7[4], ray7[5]), ray7[6]};
    return bounds_intersects(b, r) ? 1 : 0;
}
/* kind: 0 sphere {origin, radius}; 1 triangle from 3 vertices (Shape::new_triangle); 2 disk */
int orc_shape_intersect(int kind, const double* params, double* ray7_inout, orc_hit* out) {
    Shape s;
    if (kind == 0) s = make_sphere(v3(params[0], params[1], params[2]), params[3]);
    else if (kind == 2) s = make_disk(v3(params[0], params[1], params[2]), params[3], params[4], params[5], params[6]);
    else {
        /* Shape::new_triangle, shape.rs:70-95 */
        V3 v0 = v3(params[0], params[1], params[2]), v1 = v3(params[3], params[4], params[5]), v2 = v3(params[6], params[7], params[8]);
        V3 e1 = v1 - v0, e2 = v2 - v0;
        V3 n0 = cross(e2, e1);
        double mag = magnitude(n0);
        if (mag == 0.0) return -1;
        n0 = n0 / mag;
        memset(&s, 0, sizeof(s));
        s.kind = CRAY_SHAPE_TRIANGLE;
        cray_triangle t; memset(&t, 0, sizeof(t));
        t.v0 = {v0.x, v0.y, v0.z}; t.e1 = {e1.x, e1.y, e1.z}; t.e2 = {e2.x, e2.y, e2.z}; t.n0 = {n0.x, n0.y, n0.z};
        t.uv0[0] = 0; t.uv0[1] = 0; t.uv01[0] = 1; t.uv01[1] = 0; t.uv02[0] = 1; t.uv02[1] = 1;
        s.tri = t;
    }
    Ray r = {v3(ray7_inout[0], ray7_inout[1], ray7_inout[2]), v3(ray7_inout[3], ray7_inout[4], ray7_inout[5]), ray7_inout[6]};
    Hit h; memset(&h, 0, sizeof(h));
    bool hit = shape_intersect(s, r, &h);
    ray7_inout[6] = r.tmax;
    memset(out, 0, sizeof(*out));
    out->hit = hit; out->t = r.tmax;
    out->location[0] = h.location.x; out->location[1] = h.location.y; out->location[2] = h.location.z;
    out->normal[0] = h.normal.x; out->normal[1] = h.normal.y; out->normal[2] = h.normal.z;
    out->uv[0] = h.u; out->uv[1] = h.v;
    return hit ? 1 : 0;
}
void orc_shape_bounds(int kind, const double* params, double* bmin, double* bmax) {
    Shape s;
    if (kind == 0) s = make_sphere(v3(params[0], params[1], params[2]), params[3]);
    else if (kind == 2) s = make_disk(v3(params[0], params[1], params[2]), params[3], params[4], params[5], params[6]);
    else {
        V3 v0 = v3(params[0], params[1], params[2]), v1 = v3(params[3], params[4], params[5]), v2 = v3(params[6], params[7], params[8]);
        memset(&s, 0, sizeof(s)); s.kind = CRAY_SHAPE_TRIANGLE;
        V3 e1 = v1 - v0, e2 = v2 - v0;
        s.tri.v0 = {v0.x, v0.y, v0.z}; s.tri.e1 = {e1.x, e1.y, e1.z}; s.tri.e2 = {e2.x, e2.y, e2.z};
    }
    Bounds b = shape_bounds(s);
    bmin[0] = b.mn.x; bmin[1] = b.mn.y; bmin[2] = b.mn.z; bmax[0] = b.mx.x; bmax[1] = b.mx.y; bmax[2] = b.mx.z;
}
void orc_reflect(const double* d, const double* n, double* out) {
    V3 r = reflect(v3(d[0], d[1], d[2]), v3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int orc_refract(const double* d, const double* n, double cos_theta_i, double eta_i, double eta_t, double* out) {
    V3 r;
    if (!refract(v3(d[0], d[1], d[2]), v3(n[0], n[1], n[2]), cos_theta_i, eta_i, eta_t, &r)) return 0;
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return 1;
}
double orc_fresnel_dielectric(double eta_i, double eta_t, double cos_theta_i) { return fresnel_dielectric(eta_i, eta_t, cos_theta_i); }
void orc_fresnel_conductor(const double* eta_i, const double* eta_t, const double* k, double cos_theta_i, double* out) {
    int af = 0;
    Col c = fresnel_conductor(col(eta_i[0], eta_i[1], eta_i[2]), col(eta_t[0], eta_t[1], eta_t[2]), col(k[0], k[1], k[2]), cos_theta_i, &af);
    out[0] = c.r; out[1] = c.g; out[2] = c.b;
}
void orc_mat_mul(const double* a, const double* b, double* out) {
    Mat A, B; memcpy(A.m, a, 128); memcpy(B.m, b, 128);
    Mat C = mat_mul(A, B); memcpy(out, C.m, 128);
}
int orc_mat_inverse(const double* a, double* out) {
    Mat A, R; memcpy(A.m, a, 128);
    if (!mat_inverse(A, &R)) return 0;
    memcpy(out, R.m, 128);
    return 1;
}
/* kind: 0 translate(a,b,c) 1 scale(a,b,c) 2 rotate_x(a) 3 rotate_y(a) 4 rotate_z(a)
 *       5 look_at(origin p[0..3], target p[3..6], up p[6..9]) 6 perspective(fov,near,far) 7 orthographic(near,far)
 * out: matrix[16] then inverse[16] */
void orc_transformation(int kind, const double* p, double* out32) {
    Xf t;
    switch (kind) {
    case 0: t = xf_translate(p[0], p[1], p[2]); break;
    case 1: t = xf_scale(p[0], p[1], p[2]); break;
    case 2: t = xf_rotate_x(p[0]); break;
    case 3: t = xf_rotate_y(p[0]); break;
    case 4: t = xf_rotate_z(p[0]); break;
    case 5: t = xf_look_at(v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5]), v3(p[6], p[7], p[8])); break;
    case 6: t = xf_perspective(p[0], p[1], p[2]); break;
    default: t = xf_orthographic(p[0], p[1]); break;
    }
    memcpy(out32, t.matrix.m, 128); memcpy(out32 + 16, t.inverse.m, 128);
}
/* what: 0 point 1 vector 2 normal 3 ray(7) 4 bounds(6) */
void orc_transform(const double* xf32, int what, const double* in, double* out) {
    Xf t; memcpy(t.matrix.m, xf32, 128); memcpy(t.inverse.m, xf32 + 16, 128);
    if (what == 0) { V3 r = xf_point(t, v3(in[0], in[1], in[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    else if (what == 1) { V3 r = xf_vector(t, v3(in[0], in[1], in[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    else if (what == 2) { V3 r = xf_normal(t, v3(in[0], in[1], in[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    else if (what == 3) {
        Ray r = {v3(in[0], in[1], in[2]), v3(in[3], in[4], in[5]), in[6]};
        Ray o = xf_ray(t, r);
        out[0] = o.o.x; out[1] = o.o.y; out[2] = o.o.z; out[3] = o.d.x; out[4] = o.d.y; out[5] = o.d.z; out[6] = o.tmax;
    } else {
        Bounds b = {v3(in[0], in[1], in[2]), v3(in[3], in[4], in[5])};
        Bounds o = xf_bounds(t, b);
        out[0] = o.mn.x; out[1] = o.mn.y; out[2] = o.mn.z; out[3] = o.mx.x; out[4] = o.mx.y; out[5] = o.mx.z;
    }
}
void orc_color_from_rgb(uint8_t r, uint8_t g, uint8_t b, double* out) {
    Col c = from_rgb(r, g, b); out[0] = c.r; out[1] = c.g; out[2] = c.b;
}
void orc_color_to_rgb(const double* c, uint8_t* out) { to_rgb(col(c[0], c[1], c[2]), out); }
/* probes of the arithmetic types the path is built from (geometry/vector.rs, point.rs, color.rs, bounds.rs) */
void orc_vec_op(int op, const double* a, const double* b, double s, double* out) {
    V3 x = v3(a[0], a[1], a[2]), y = b ? v3(b[0], b[1], b[2]) : v3(0, 0, 0), r = v3(0, 0, 0);
    switch (op) {
        case 0: r = normalized(x); break;
        case 1: out[0] = magnitude(x); return;
        case 2: out[0] = dot(x, y); return;
        case 3: r = cross(x, y); break;
        case 4: r = x + y; break;
        case 5: r = x - y; break;
        case 6: r = x * s; break;
        case 7: r = x / s; break;
        default: break;
    }
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_col_op(int op, const double* a, const double* b, double s, double* out) {
    Col x = col(a[0], a[1], a[2]), y = b ? col(b[0], b[1], b[2]) : col(0, 0, 0), r = x;
    switch (op) {
        case 0: r = x + y; break;
        case 1: r = x * s; break;
        case 2: r = x / s; break;
        case 3: r = x * y; break;
        default: break;
    }
    out[0] = r.r; out[1] = r.g; out[2] = r.b;
}
void orc_bounds_sum(const double* a6, const double* b6, double* out6) {
    Bounds r = bounds_union(bounds_new(v3(a6[0], a6[1], a6[2]), v3(a6[3], a6[4], a6[5])), bounds_new(v3(b6[0], b6[1], b6[2]), v3(b6[3], b6[4], b6[5])));
    out6[0] = r.mn.x; out6[1] = r.mn.y; out6[2] = r.mn.z; out6[3] = r.mx.x; out6[4] = r.mx.y; out6[5] = r.mx.z;
}
/* partition_by over i64 with predicate (x % mod == rem) or (x > thr) */
uint64_t orc_partition_by(int64_t* data, uint64_t n, int mode, int64_t a, int64_t b) {
    return partition_by(data, (size_t)n, [&](const int64_t& x) { return mode == 0 ? (x > a) : (((x % a) + a) % a == b); });
}
/* Frame (src/transformation.rs:494-535): three orthonormal vectors; from_xy normalises x, y and takes z = normalized(x cross y)
 * (:502-509); from_local = x t.x + y t.y + z t.z (:518-520, the Normal impl :528-530 is the same arithmetic), to_local =
 * (v . x, v . y, v . z) (:522-524, :532-534).  No code of the reference's hot path uses it; restated for its tests only
 * (tests/test_transformation.rs:186-225). */
void orc_frame(const double* fx, const double* fy, const double* v, double* from_local, double* to_local) {
    const V3 x = normalized(v3(fx[0], fx[1], fx[2])), y = normalized(v3(fy[0], fy[1], fy[2]));
    const V3 z = normalized(cross(v3(fx[0], fx[1], fx[2]), v3(fy[0], fy[1], fy[2])));
    const V3 t = v3(v[0], v[1], v[2]);
    const V3 f = x * t.x + y * t.y + z * t.z;
    from_local[0] = f.x; from_local[1] = f.y; from_local[2] = f.z;
    to_local[0] = dot(t, x); to_local[1] = dot(t, y); to_local[2] = dot(t, z);
}
void orc_sampling_fn(int which, double u, double v, const double* n, double* out) {
    int af = 0;
    if (which == 0) { sample_disk(u, v, &out[0], &out[1]); }
    else if (which == 1) { V3 r = sample_sphere(u, v); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    else if (which == 2) { V3 r = sample_hemisphere(u, v, v3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    else if (which == 3) { sample_triangle(u, v, &out[0], &out[1]); }
    else { V3 r = cosine_sample_hemisphere(u, v, v3(n[0], n[1], n[2]), &af); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
}

} /* extern "C" */
