// cray_shading.h — device functions for everything the integrator calls besides the BVH:
// sampler, camera, shape re-evaluation at the hit, textures, BxDF/BSDF/Material, lights.
// Each function names the reference code whose arithmetic it reproduces (operand order
// included); all of it is f64 with FMA contraction off.
#pragma once

#include "cray_device.h"

namespace cray {

// =============================================================================
// Scene feature mask: what the uploaded scene can make k_shade do.  k_shade is instantiated for a handful of masks
// (kShadeVariants, cray_kernels.h) and cray_scene_upload picks the leanest instantiation that covers the scene, so a
// scene of constant-textured Lambert + conductor surfaces under one disk light does not carry the registers of the
// image-texture, Oren-Nayar, glass, point/distant/infinite-light and sphere-emitter code.  A feature bit only removes
// code the scene cannot reach: the arithmetic of what remains is untouched.
// =============================================================================
enum : uint32_t {
    SF_TEX_CHECKER = 1u << 0, SF_TEX_IMAGE = 1u << 1,
    SF_OREN_NAYAR = 1u << 2, SF_CONDUCTOR = 1u << 3, SF_SPEC_BRDF = 1u << 4, SF_SPEC_BTDF = 1u << 5, SF_FRESNEL_SPEC = 1u << 6,
    SF_MULTI_LOBE = 1u << 7,        // some Material::BSDF holds a number of lobes other than one
    SF_LIGHT_POINT = 1u << 8, SF_LIGHT_DISTANT = 1u << 9, SF_LIGHT_INFINITE = 1u << 10,
    SF_AREA_TRI = 1u << 11, SF_AREA_SPHERE = 1u << 12, SF_AREA_DISK = 1u << 13,   // shapes of the area lights
    SF_MANY_LIGHTS = 1u << 14,      // more than one light: LightSampler::sample has something to search
    SF_HIT_TRI = 1u << 15, SF_HIT_SPHERE = 1u << 16, SF_HIT_DISK = 1u << 17,      // shapes a path can hit
    SF_ALL = (1u << 18) - 1u
};
#define CRAY_HAS(F, bits) (((F) & (bits)) != 0u)

// =============================================================================
// Sampler: SipHash-1-3 pixel seed + Burley's Owen-scrambled Sobol
// (src/sampling.rs:196-247 -> std DefaultHasher, sobol_burley 0.5.0)
// =============================================================================
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }

#define CRAY_SIPROUND(v0, v1, v2, v3)                                        \
    do {                                                                     \
        v0 += v1; v1 = rotl64(v1, 13); v1 ^= v0; v0 = rotl64(v0, 32);        \
        v2 += v3; v3 = rotl64(v3, 16); v3 ^= v2;                             \
        v0 += v3; v3 = rotl64(v3, 21); v3 ^= v0;                             \
        v2 += v1; v1 = rotl64(v1, 17); v1 ^= v2; v2 = rotl64(v2, 32);        \
    } while (0)

// DefaultHasher::new() (zero keys); seed.hash, x.hash, y.hash = three LE u64 words;
// `finish() as u32` (sampling.rs:224-228)
__device__ __forceinline__ uint32_t pixel_seed(uint64_t seed, uint64_t x, uint64_t y) {
    uint64_t v0 = 0x736f6d6570736575ULL, v1 = 0x646f72616e646f6dULL;
    uint64_t v2 = 0x6c7967656e657261ULL, v3 = 0x7465646279746573ULL;
    uint64_t m[3] = {seed, x, y};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        v3 ^= m[i];
        CRAY_SIPROUND(v0, v1, v2, v3);
        v0 ^= m[i];
    }
    const uint64_t b = 24ULL << 56;
    v3 ^= b;
    CRAY_SIPROUND(v0, v1, v2, v3);
    v0 ^= b;
    v2 ^= 0xff;
    CRAY_SIPROUND(v0, v1, v2, v3);
    CRAY_SIPROUND(v0, v1, v2, v3);
    CRAY_SIPROUND(v0, v1, v2, v3);
    return (uint32_t)(v0 ^ v1 ^ v2 ^ v3);
}

// (IndependentSampler's generator — SipHash of the pixel sample, PCG32 seed expansion, ChaCha12 — lives in cray_math.h: host + device)

__device__ __forceinline__ uint32_t lk_seed_hash(uint32_t n, uint32_t k) {
    uint32_t h = n ^ k;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t lk_scramble(uint32_t n, uint32_t s) {  // Laine-Karras style hash on reversed bits
    n ^= n * 0x3d20adeau;
    n += s;
    n *= (s >> 16) | 1u;
    n ^= n * 0x05526c56u;
    n ^= n * 0x53a22864u;
    return n;
}
__device__ __forceinline__ float unit_float(uint32_t n) { return __uint_as_float((n >> 9) | 0x3f800000u) - 1.0f; }

// sobol_burley::sample_4d(index, set, seed): the four dimensions 4*set .. 4*set+3.
// `set` is uniform over a launch (every path of a bounce consumes the same dimensions),
// so the table reads are scalar loads.
__device__ __forceinline__ void sobol4(const uint16_t* __restrict__ table, uint32_t index, uint32_t set, uint32_t seed,
                                       double out[4]) {
    const uint2* v = reinterpret_cast<const uint2*>(table + (size_t)set * 64);
    uint32_t idx = lk_scramble(__brev(index), lk_seed_hash(seed, 0x79c68e4au)) & 0xffff0000u;
    uint32_t a = 0, b = 0;  // a = lanes 0|1, b = lanes 2|3 (u16 each)
#pragma unroll
    for (int bit = 0; bit < 16; bit++) {
        uint2 w = v[bit];
        uint32_t m = 0u - ((idx >> (31 - bit)) & 1u);
        a ^= w.x & m;
        b ^= w.y & m;
    }
    const uint32_t s = set ^ seed;
    out[0] = (double)unit_float(__brev(lk_scramble(a & 0xffffu, lk_seed_hash(s, 0x912f69bau))));
    out[1] = (double)unit_float(__brev(lk_scramble(a >> 16, lk_seed_hash(s, 0x174f18abu))));
    out[2] = (double)unit_float(__brev(lk_scramble(b & 0xffffu, lk_seed_hash(s, 0x691e72cau))));
    out[3] = (double)unit_float(__brev(lk_scramble(b >> 16, lk_seed_hash(s, 0xb40cc1b8u))));
}

// sample_4d through nibble tables (round 3).  Which direction vectors are XORed depends on the top 16 bits of the scrambled
// index; `sobol4` walks them bit by bit (16 x {extract, 2 and, 2 xor} = 80 VALU instructions per call, twice per path and
// bounce: the largest single block of k_shade's ~1 600).  A kernel that uses a set for a whole launch can instead fold the
// set's 16 vectors into four tables of 16 XOR combinations (nibble j of the index -> the XOR of its set bits' vectors, both
// lane pairs: 64 x 8 B) once per block in LDS and XOR four entries: the same word, by associativity of XOR.
//   T[16 j + v] = XOR over t in 0..3 with bit (3 - t) of v set of  vectors[4 j + t]      (bit 0 of the loop = index bit 31)
__device__ __forceinline__ void sobol_fill_lut(const uint16_t* __restrict__ table, uint32_t set, uint32_t entry, uint2* __restrict__ lut) {
    const uint2* v = reinterpret_cast<const uint2*>(table + (size_t)set * 64);
    const uint32_t j = entry >> 4, nv = entry & 15u;
    uint2 acc = make_uint2(0u, 0u);
#pragma unroll
    for (uint32_t t = 0; t < 4; t++)
        if ((nv >> (3u - t)) & 1u) { const uint2 w = v[4 * j + t]; acc.x ^= w.x; acc.y ^= w.y; }
    lut[entry] = acc;
}
__device__ __forceinline__ void sobol4_lut(const uint2* __restrict__ lut, uint32_t index, uint32_t set, uint32_t seed, double out[4]) {
    const uint32_t idx = lk_scramble(__brev(index), lk_seed_hash(seed, 0x79c68e4au));   // only the top 16 bits select vectors
    uint32_t a = 0, b = 0;  // a = lanes 0|1, b = lanes 2|3 (u16 each)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint2 w = lut[16 * j + ((idx >> (28 - 4 * j)) & 15u)];
        a ^= w.x;
        b ^= w.y;
    }
    const uint32_t s = set ^ seed;
    out[0] = (double)unit_float(__brev(lk_scramble(a & 0xffffu, lk_seed_hash(s, 0x912f69bau))));
    out[1] = (double)unit_float(__brev(lk_scramble(a >> 16, lk_seed_hash(s, 0x174f18abu))));
    out[2] = (double)unit_float(__brev(lk_scramble(b & 0xffffu, lk_seed_hash(s, 0x691e72cau))));
    out[3] = (double)unit_float(__brev(lk_scramble(b >> 16, lk_seed_hash(s, 0xb40cc1b8u))));
}

// =============================================================================
// sampling_fns (src/sampling.rs:11-65)
// =============================================================================
__device__ __forceinline__ double power_heuristic(double pf, double pg) {
    double f = 1.0 * pf, g = 1.0 * pg;
    return (f * f) / (f * f + g * g);
}
__device__ __forceinline__ void sample_disk(double u, double v, double& x, double& y) {
    if (u == 0.0 || v == 0.0) { x = 0.0; y = 0.0; return; }
    u = 2.0 * u - 1.0;
    v = 2.0 * v - 1.0;
    double r, theta;
    if (fabs(u) > fabs(v)) { r = u; theta = kQuarterPi * v / u; }
    else { r = v; theta = kHalfPi - kQuarterPi * u / v; }
    double st, ct;
#ifdef CRAY_EXPERIMENT_OCML_TRIG
    st = sin(theta); ct = cos(theta);
#else
    sincos_cr(theta, st, ct);  // correctly rounded, see cray_math.h
#endif
    x = ct * r;
    y = st * r;
}
__device__ __forceinline__ vec3 sample_sphere(double u, double v) {
    double z = 1.0 - 2.0 * u;
    double r = sqrt(max_nn(1.0 - square(z), 0.0));
    double phi = 2.0 * kPi * v;
    double sp, cp;
    sincos_cr(phi, sp, cp);
    return mk(r * cp, r * sp, z);
}
__device__ __forceinline__ vec3 cosine_hemisphere(double u, double v, vec3 n) {
    vec3 t, b;
    tangents(n, t, b);
    double x, y;
    sample_disk(u, v, x, y);
    double z = sqrt(max_nn(1.0 - x * x - y * y, 0.0));
    return unit(t * x + b * y + n * z);
}

// =============================================================================
// Camera::sample + generate_ray (src/camera.rs:131-162)
// =============================================================================
__device__ __forceinline__ ray_t camera_ray(const DevScene& sc, double fx, double fy, double lx, double ly, uint32_t px, uint32_t py) {
    double ddx = 2.0 * fx - 1.0, ddy = 2.0 * fy - 1.0;
    vec3 p_raster = mk((double)px + ddx, (double)py + ddy, 0.0);
    vec3 p_cam = xf_point(sc.camera_from_raster, p_raster);
    ray_t r = sc.camera_type == CRAY_CAMERA_PERSPECTIVE ? mkray(p_cam, unit(p_cam - mk(0, 0, 0))) : mkray(p_cam, mk(0, 0, 1));
    if (sc.lens_radius != 0.0) {
        double lens_x = 2.0 * lx - 1.0, lens_y = 2.0 * ly - 1.0;
        vec3 p_lens = mk(lens_x * sc.lens_radius, lens_y * sc.lens_radius, 0.0);
        vec3 p_focal = at(r, sc.focal_distance / r.d.z);
        r = mkray(p_lens, unit(p_focal - p_lens));
    }
    return xf_ray(sc.world_from_camera, r);
}

// =============================================================================
// Sphere / Disk intersection (src/shape.rs:159-215, 263-309, 316-342, 367-398).
// Triangles live in the traversal kernel.  `full` also produces location/normal/uv.
// =============================================================================
struct SurfPoint {
    vec3 location, normal;
    double u, v;
};

__device__ inline bool sphere_hit(const cray_xf_shape& s, ray_t& ray, bool any_only, SurfPoint* sp) {
    ray_t obj = xf_ray(s.inv, ray);  // world_to_object.matrix == object_to_world.inverse
    vec3 oc = obj.o;
    double a = len2(obj.d);
    double b = 2.0 * dot(oc, obj.d);
    double c = len2(oc) - square(s.radius);
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return false;
    double root = sqrt(disc);
    double inv_2a = 1.0 / (2.0 * a);
    double t0 = (-b - root) * inv_2a, t1 = (-b + root) * inv_2a;
    if (any_only) return in_range(obj, t0) || in_range(obj, t1);
    for (int k = 0; k < 2; k++) {
        double t = k == 0 ? t0 : t1;
        if (shrink(obj, t)) {
            vec3 loc = at(obj, t);
            shrink(ray, t);
            if (sp) {
                double phi = atan2(loc.y, loc.x);
                if (phi < 0.0) phi += kPi * 2.0;
                sp->u = phi / (kPi * 2.0);
                sp->v = acos(loc.z / s.radius) * kInvPi;
                sp->location = xf_point(s.m, loc);
                sp->normal = xf_normal(s.inv, loc / s.radius);
            }
            return true;
        }
    }
    return false;
}

__device__ inline bool disk_hit(const cray_xf_shape& s, ray_t& ray, bool any_only, SurfPoint* sp) {
    ray_t obj = xf_ray(s.inv, ray);
    if (obj.d.z == 0.0) return false;
    double t = -obj.o.z / obj.d.z;
    if (!in_range(obj, t)) return false;
    vec3 loc = mk(obj.o.x + obj.d.x * t, obj.o.y + obj.d.y * t, 0.0);
    double d2 = square(loc.x) + square(loc.y);
    if (d2 < square(s.inner_radius) || d2 > square(s.radius)) return false;
    if (any_only) return in_range(ray, t);
    if (!shrink(ray, t)) return false;
    if (sp) {
        double theta = atan2(loc.y, loc.x);
        if (theta < 0.0) theta += kPi * 2.0;
        sp->u = theta / (kPi * 2.0);
        sp->v = sqrt(d2) / s.radius;
        sp->location = xf_point(s.m, loc);
        sp->normal = xf_normal(s.inv, mk(0.0, 0.0, 1.0));
    }
    return true;
}

// Location / normal / uv of a closest hit found by the traversal kernel.  For sphere and
// disk the shape code is re-run on the recorded distance: same inputs, same operations,
// hence the same bits the reference computed eagerly inside Shape::intersect.
// need_uv = false leaves (u, v) of a sphere / disk hit at 0: for materials whose textures are all constant the
// reference computes them (atan2, acos) and never looks at them.
template <uint32_t F = SF_ALL>
__device__ inline SurfPoint surface_at(const DevScene& sc, const cray_prim& pr, const ray_t& ray_in, double t, double bu, double bv, bool need_uv) {
    SurfPoint sp;
    sp.u = 0.0; sp.v = 0.0;
    if (CRAY_HAS(F, SF_HIT_TRI) && (!CRAY_HAS(F, SF_HIT_SPHERE | SF_HIT_DISK) || pr.shape_kind == CRAY_SHAPE_TRIANGLE)) {
        const TriShade& ts = sc.tri_shade[pr.shape];
        sp.location = at(ray_in, t);
        sp.normal = unit(mk(ts.n0[0], ts.n0[1], ts.n0[2]) + mk(ts.n01[0], ts.n01[1], ts.n01[2]) * bu + mk(ts.n02[0], ts.n02[1], ts.n02[2]) * bv);
        sp.u = ts.uv0[0] + ts.uv01[0] * bu + ts.uv02[0] * bv;
        sp.v = ts.uv0[1] + ts.uv01[1] * bu + ts.uv02[1] * bv;
        return sp;
    }
    const bool is_sphere = CRAY_HAS(F, SF_HIT_SPHERE) && (!CRAY_HAS(F, SF_HIT_DISK) || pr.shape_kind == CRAY_SHAPE_SPHERE);
    const cray_xf_shape& s = is_sphere ? sc.spheres[pr.shape] : sc.disks[pr.shape];
    vec3 oo = xf_point(s.inv, ray_in.o), od = xf_vector(s.inv, ray_in.d);
    if (is_sphere) {
        vec3 loc = oo + od * t;
        if (need_uv) {
            double phi = atan2(loc.y, loc.x);
            if (phi < 0.0) phi += kPi * 2.0;
            sp.u = phi / (kPi * 2.0);
            sp.v = acos(loc.z / s.radius) * kInvPi;
        }
        sp.location = xf_point(s.m, loc);
        sp.normal = xf_normal(s.inv, loc / s.radius);
    } else {
        vec3 loc = mk(oo.x + od.x * t, oo.y + od.y * t, 0.0);
        if (need_uv) {
            double d2 = square(loc.x) + square(loc.y);
            double theta = atan2(loc.y, loc.x);
            if (theta < 0.0) theta += kPi * 2.0;
            sp.u = theta / (kPi * 2.0);
            sp.v = sqrt(d2) / s.radius;
        }
        sp.location = xf_point(s.m, loc);
        sp.normal = xf_normal(s.inv, mk(0.0, 0.0, 1.0));
    }
    return sp;
}

// Shape::Triangle intersection (Moller-Trumbore as written in shape.rs:216-262 / 343-366).
// Returns true when the candidate distance lies in (EPSILON, ray.tmax); the caller shrinks tmax.
__device__ __forceinline__ bool tri_test(vec3 v0, vec3 e1, vec3 e2, const ray_t& ray, double& t, double& u, double& v) {
    vec3 P = cross(ray.d, e2);
    double denom = dot(P, e1);
    if (denom > -kEps && denom < kEps) return false;
    vec3 T = ray.o - v0;
    u = dot(P, T) / denom;
    if (u < 0.0 || u > 1.0) return false;
    vec3 Q = cross(T, e1);
    v = dot(Q, ray.d) / denom;
    if (v < 0.0 || u + v > 1.0) return false;
    t = dot(cross(T, e1), e2) / denom;
    return in_range(ray, t);
}

// The same test with the reciprocal of the denominator formed once for its (up to) three quotients.  A true f64 division is
// v_div_scale (denominator), v_rcp, two Newton steps (4 fma), v_div_scale (numerator), q = n' y, r = n' - d' q, v_div_fmas,
// v_div_fixup; the first six depend on the numerator only through the scaling v_div_scale may apply to the denominator for
// extreme operand pairs.  The scaled denominator is formed for each numerator (one instruction) and when it is the one the
// reciprocal was refined for — always, unless an exponent is extreme — the quotient takes five more instructions instead of ten,
// bit for bit the compiler's.  A lane where it differs divides plainly.  (Without the early-outs — three quotients side by
// side for every lane — the mixed launches were 0.9 ms slower: whole waves do fail the first test together.)
__device__ __forceinline__ bool tri_test_shared(vec3 v0, vec3 e1, vec3 e2, const ray_t& ray, double& t, double& u, double& v) {
    const vec3 P = cross(ray.d, e2);
    const double denom = dot(P, e1);
    if (denom > -kEps && denom < kEps) return false;
    const vec3 T = ray.o - v0;
    const double nu = dot(P, T);
    bool f;
    const double du = __builtin_amdgcn_div_scale(nu, denom, false, &f);
    double y = __builtin_amdgcn_rcp(du);
    y = __builtin_fma(y, __builtin_fma(-du, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-du, y, 1.0), y);
    auto quot = [&](double n) {
        bool fd, fn;
        const double dn = __builtin_amdgcn_div_scale(n, denom, false, &fd);
        if (__builtin_bit_cast(unsigned long long, dn) != __builtin_bit_cast(unsigned long long, du)) return n / denom;
        const double ns = __builtin_amdgcn_div_scale(n, denom, true, &fn);
        const double q = ns * y;
        const double r = __builtin_fma(-du, q, ns);
        return __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(r, y, q, fn), denom, n);
    };
    u = quot(nu);
    if (u < 0.0 || u > 1.0) return false;
    const vec3 Q = cross(T, e1);
    v = quot(dot(Q, ray.d));
    if (v < 0.0 || u + v > 1.0) return false;
    t = quot(dot(Q, e2));
    return in_range(ray, t);
}

// Shape::pdf_from (shape.rs:487-502): re-intersect the light's OWN shape from the shading
// point; pdf = d^2 / (|w_i . n_x| * area) with n_x the *shading point's* normal (sic).
template <uint32_t F = SF_ALL>
__device__ inline double light_shape_pdf_from(const DevScene& sc, const DevLight& l, vec3 x, vec3 n_x, vec3 w_i) {
    ray_t ray = mkray(x, w_i);
    vec3 hit_location;
    if (CRAY_HAS(F, SF_AREA_TRI) && (!CRAY_HAS(F, SF_AREA_SPHERE | SF_AREA_DISK) || l.shape_kind == CRAY_SHAPE_TRIANGLE)) {
        double t, u, v;
        if (!tri_test(mk(l.v0[0], l.v0[1], l.v0[2]), mk(l.e1[0], l.e1[1], l.e1[2]), mk(l.e2[0], l.e2[1], l.e2[2]), ray, t, u, v)) return 0.0;
        hit_location = at(ray, t);
    } else {
        SurfPoint sp;
        const bool is_sphere = CRAY_HAS(F, SF_AREA_SPHERE) && (!CRAY_HAS(F, SF_AREA_DISK) || l.shape_kind == CRAY_SHAPE_SPHERE);
        bool hit = is_sphere ? sphere_hit(sc.spheres[l.shape], ray, false, &sp) : disk_hit(sc.disks[l.shape], ray, false, &sp);
        if (!hit) return 0.0;
        hit_location = sp.location;
    }
    double d2 = len2(hit_location - x);
    double cos_theta = fabs(dot(w_i, n_x));
    return d2 / (cos_theta * l.area);
}

// Shape::sample (shape.rs:445-470)
template <uint32_t F = SF_ALL>
__device__ inline vec3 light_shape_sample(const DevScene& sc, const DevLight& l, double u, double v) {
    if (CRAY_HAS(F, SF_AREA_SPHERE) && (!CRAY_HAS(F, SF_AREA_TRI | SF_AREA_DISK) || l.shape_kind == CRAY_SHAPE_SPHERE)) {
        const cray_xf_shape& s = sc.spheres[l.shape];
        return xf_point(s.m, mk(0, 0, 0) + sample_sphere(u, v) * s.radius);
    }
    if (CRAY_HAS(F, SF_AREA_TRI) && (!CRAY_HAS(F, SF_AREA_DISK) || l.shape_kind == CRAY_SHAPE_TRIANGLE)) {
        double su = sqrt(u);  // sample_triangle, sampling.rs:51-55
        double b1 = 1.0 - su, b2 = v * su;
        return mk(l.v0[0], l.v0[1], l.v0[2]) + mk(l.e1[0], l.e1[1], l.e1[2]) * b1 + mk(l.e2[0], l.e2[1], l.e2[2]) * b2;
    }
    const cray_xf_shape& s = sc.disks[l.shape];
    double x, y;
    sample_disk(u, v, x, y);  // ignores inner_radius, as the reference does
    return xf_point(s.m, mk(x * s.radius, y * s.radius, 0.0));
}

// =============================================================================
// Texture<T>::eval (src/texture.rs:19-47, 103-113)
// =============================================================================
__device__ __forceinline__ double fract_keep_sign(double x) { return x - trunc(x); }
__device__ inline const uint8_t* image_texel(const DevScene& sc, const cray_texture& t, double u, double v) {
    const cray_image& im = sc.images[t.image];
    double fu = fract_keep_sign(u);
    if (fu < 0.0) fu += 1.0;
    double fv = fract_keep_sign(v);
    if (fv < 0.0) fv += 1.0;
    uint32_t x = to_u32_sat((double)(im.width - 1) * fu);
    uint32_t y = to_u32_sat((double)(im.height - 1) * fv);
    return sc.pool + im.offset + 3ull * ((uint64_t)y * im.width + x);
}
template <uint32_t F = SF_ALL>
__device__ inline rgb tex_color(const DevScene& sc, int32_t id, double u, double v) {
    const cray_texture& t = sc.textures[id];
    if (!CRAY_HAS(F, SF_TEX_CHECKER | SF_TEX_IMAGE) || t.kind == CRAY_TEX_CONSTANT) return mkc(t.a.r, t.a.g, t.a.b);
    if (CRAY_HAS(F, SF_TEX_CHECKER) && (!CRAY_HAS(F, SF_TEX_IMAGE) || t.kind == CRAY_TEX_CHECKERBOARD)) {
        uint64_t uu = to_u64_sat(u * t.scale * 2.0), vv = to_u64_sat(v * t.scale * 2.0);
        return ((uu & 1) ^ (vv & 1)) == 0 ? mkc(t.a.r, t.a.g, t.a.b) : mkc(t.b.r, t.b.g, t.b.b);
    }
    const uint8_t* p = image_texel(sc, t, u, v);
    return mkc(sc.gamma_lut[p[0]], sc.gamma_lut[p[1]], sc.gamma_lut[p[2]]);  // == (c/255).powf(2.2) per lookup
}
template <uint32_t F = SF_ALL>
__device__ inline double tex_scalar(const DevScene& sc, int32_t id, double u, double v) {
    const cray_texture& t = sc.textures[id];
    if (!CRAY_HAS(F, SF_TEX_CHECKER | SF_TEX_IMAGE) || t.kind == CRAY_TEX_CONSTANT) return t.a.r;
    if (CRAY_HAS(F, SF_TEX_CHECKER) && (!CRAY_HAS(F, SF_TEX_IMAGE) || t.kind == CRAY_TEX_CHECKERBOARD)) {
        uint64_t uu = to_u64_sat(u * t.scale * 2.0), vv = to_u64_sat(v * t.scale * 2.0);
        return ((uu & 1) ^ (vv & 1)) == 0 ? t.a.r : t.b.r;
    }
    const uint8_t* p = image_texel(sc, t, u, v);
    uint32_t luma = (2126u * p[0] + 7152u * p[1] + 722u * p[2]) / 10000u;  // image 0.24 Rgb::to_luma
    return (double)(luma & 0xffu) / 255.0;
}

// =============================================================================
// BxDF / BSDF / Material (src/bxdf.rs, src/bsdf.rs, src/material.rs)
// =============================================================================
__device__ __forceinline__ vec3 reflect(vec3 d, vec3 n) { return n * (dot(n, d) * 2.0) - d; }  // bxdf.rs:287-290
__device__ inline bool refract(vec3 d, vec3 n, double cos_i, double eta_i, double eta_t, vec3& out) {  // :292-314
    double eta_rel, c;
    if (sign_neg(cos_i)) { n = flip(n); eta_rel = eta_i / eta_t; c = -cos_i; }
    else { eta_rel = eta_t / eta_i; c = cos_i; }
    double s = sqrt(1.0 - c * c);
    if (s > eta_rel) return false;
    vec3 perp = (n * c - d) / eta_rel;
    vec3 par = n * -sqrt(1.0 - dot(perp, perp));
    out = perp + par;
    return true;
}
__device__ inline double fresnel_dielectric(double eta_i, double eta_t, double cos_i) {  // :338-357
    if (sign_neg(cos_i)) { cos_i = -cos_i; double t = eta_i; eta_i = eta_t; eta_t = t; }
    double sin_i = sqrt(1.0 - cos_i * cos_i);
    double sin_t = eta_i / eta_t * sin_i;
    if (sin_t >= 1.0) return 1.0;
    double cos_t = sqrt(1.0 - sin_t * sin_t);
    double r_par = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
    double r_perp = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
    return (r_par * r_par + r_perp * r_perp) * 0.5;
}
__device__ inline rgb fresnel_conductor(rgb eta_i, rgb eta_t, rgb k, double cos_i) {  // :359-382
    const rgb one = mkc(1.0, 1.0, 1.0);
    rgb eta = eta_t / eta_i;
    rgb eta2 = eta * eta;
    rgb kk = k / eta_i;
    rgb k2 = kk * kk;
    double cos2 = cos_i * cos_i;
    double sin2 = 1.0 - cos2;
    rgb t0 = eta2 - k2 - one * sin2;
    rgb a2b2 = pow_half3(t0 * t0 + eta2 * k2 * 4.0);
    rgb a = pow_half3((a2b2 + t0) * 0.5);
    rgb t1 = a2b2 + one * cos2;
    rgb t2 = a * cos_i * 2.0;
    rgb r_perp = (t1 - t2) / (t1 + t2);
    rgb t3 = a2b2 * cos2 + one * sin2 * sin2;
    rgb t4 = a * cos_i * sin2 * 2.0;
    rgb r_par = r_perp * (t3 - t4) / (t3 + t4);
    return (r_par * r_par + r_perp * r_perp) * 0.5;
}

struct LobeSample {
    vec3 w_i;
    rgb f;
    double pdf;
    bool delta, specular;
};

__device__ __forceinline__ bool lobe_reflects(int kind) { return kind != CRAY_BXDF_SPECULAR_BTDF; }
__device__ __forceinline__ bool lobe_transmits(int kind) { return kind == CRAY_BXDF_SPECULAR_BTDF || kind == CRAY_BXDF_FRESNEL_SPECULAR; }

// BxDF::f (bxdf.rs:214-265)
template <uint32_t F = SF_ALL>
__device__ inline rgb lobe_f(const DevScene& sc, const cray_bxdf& bx, vec3 w_o, vec3 w_i, vec3 n, double u, double v) {
    const rgb zero = mkc(0, 0, 0);
    if (bx.kind == CRAY_BXDF_LAMBERTIAN) return same_side(n, w_o, w_i) ? tex_color<F>(sc, bx.tex_a, u, v) * kInvPi : zero;
    if (!CRAY_HAS(F, SF_OREN_NAYAR) || bx.kind != CRAY_BXDF_OREN_NAYAR) return zero;
    if (!same_side(n, w_o, w_i)) return zero;
    double cos_i = fabs(dot(w_i, n)), cos_o = fabs(dot(w_o, n));
    double sin_i = sqrt(max_nn(1.0 - cos_i * cos_i, 0.0)), sin_o = sqrt(max_nn(1.0 - cos_o * cos_o, 0.0));
    double max_cos = 0.0;
    if (sin_i > 1e-4 && sin_o > 1e-4) {
        vec3 tg, bt;
        tangents(n, tg, bt);
        double cpi = fabs(dot(w_i, tg)), cpo = fabs(dot(w_o, tg));
        double spi = sqrt(1.0 - cpi * cpi), spo = sqrt(1.0 - cpo * cpo);
        max_cos = max_nn(cpi * cpo + spi * spo, 0.0);
    }
    double sin_alpha, tan_beta;
    if (cos_i > cos_o) { sin_alpha = sin_o; tan_beta = sin_i / cos_i; }
    else { sin_alpha = sin_i; tan_beta = sin_o / cos_o; }
    double sigma = deg2rad(tex_scalar<F>(sc, bx.tex_b, u, v));
    double s2 = sigma * sigma;
    double A = 1.0 - s2 / (2.0 * (s2 + 0.33));
    double B = 0.45 * s2 / (s2 + 0.09);
    return tex_color<F>(sc, bx.tex_a, u, v) * (A + B * max_cos * sin_alpha * tan_beta) * kInvPi;
}
// BxDF::pdf (bxdf.rs:269-284); false = Pdf::Delta
__device__ __forceinline__ bool lobe_pdf(const cray_bxdf& bx, vec3 w_i, vec3 n, double& pdf) {
    if (bx.kind == CRAY_BXDF_LAMBERTIAN || bx.kind == CRAY_BXDF_OREN_NAYAR) {
        pdf = kInvPi * fabs(dot(w_i, n));
        return true;
    }
    return false;
}
// BxDF::sample (bxdf.rs:83-209); false = None
template <uint32_t F = SF_ALL>
__device__ inline bool lobe_sample(const DevScene& sc, const cray_bxdf& bx, double s0, double s1, vec3 w_o, vec3 n, double u, double v, LobeSample& out) {
    const rgb one = mkc(1, 1, 1);
    constexpr uint32_t kSpecular = SF_CONDUCTOR | SF_SPEC_BRDF | SF_SPEC_BTDF | SF_FRESNEL_SPEC;
    if (!CRAY_HAS(F, kSpecular) || bx.kind == CRAY_BXDF_LAMBERTIAN || bx.kind == CRAY_BXDF_OREN_NAYAR) {
        vec3 w_i = cosine_hemisphere(s0, s1, n);
        if (dot(n, w_o) < 0.0) w_i = flip(w_i);
        out.w_i = w_i;
        out.f = lobe_f<F>(sc, bx, w_o, w_i, n, u, v);
        out.delta = !lobe_pdf(bx, w_i, n, out.pdf);
        out.specular = false;
        return true;
    }
    if (CRAY_HAS(F, SF_CONDUCTOR) && (!CRAY_HAS(F, kSpecular & ~SF_CONDUCTOR) || bx.kind == CRAY_BXDF_FRESNEL_CONDUCTOR)) {
        double c = fabs(dot(w_o, n));
        out.w_i = reflect(w_o, n);
        out.f = fresnel_conductor(one, tex_color<F>(sc, bx.tex_a, u, v), tex_color<F>(sc, bx.tex_b, u, v), c) / c;
        out.delta = true; out.pdf = 0.0; out.specular = true;
        return true;
    }
    if (CRAY_HAS(F, SF_SPEC_BRDF) && (!CRAY_HAS(F, SF_SPEC_BTDF | SF_FRESNEL_SPEC) || bx.kind == CRAY_BXDF_SPECULAR_BRDF)) {
        double c = fabs(dot(w_o, n));
        rgb fr = bx.fresnel_kind == CRAY_FRESNEL_DIELECTRIC
                     ? one * fresnel_dielectric(bx.eta_i, bx.eta_t, c)
                     : fresnel_conductor(mkc(bx.c_eta_i.r, bx.c_eta_i.g, bx.c_eta_i.b), mkc(bx.c_eta_t.r, bx.c_eta_t.g, bx.c_eta_t.b),
                                         mkc(bx.c_k.r, bx.c_k.g, bx.c_k.b), c);
        out.w_i = reflect(w_o, n);
        out.f = tex_color<F>(sc, bx.tex_a, u, v) * fr / fabs(c);
        out.delta = true; out.pdf = 0.0; out.specular = true;
        return true;
    }
    if (CRAY_HAS(F, SF_SPEC_BTDF) && (!CRAY_HAS(F, SF_FRESNEL_SPEC) || bx.kind == CRAY_BXDF_SPECULAR_BTDF)) {
        double c = fabs(dot(w_o, n));
        vec3 w_i;
        if (!refract(w_o, n, c, bx.eta_i, bx.eta_t, w_i)) return false;
        double fr = fresnel_dielectric(bx.eta_i, bx.eta_t, c);
        out.w_i = w_i;
        out.f = tex_color<F>(sc, bx.tex_a, u, v) * (1.0 - fr) / c;
        out.delta = true; out.pdf = 0.0; out.specular = true;
        return true;
    }
    if (CRAY_HAS(F, SF_FRESNEL_SPEC)) {  // FresnelSpecularBxDF
        double c = dot(w_o, n);
        double Fr = fresnel_dielectric(bx.eta_i, bx.eta_t, c);
        if (s0 < Fr) {
            out.w_i = reflect(w_o, n);
            out.f = tex_color<F>(sc, bx.tex_a, u, v) * Fr / fabs(c);
            out.delta = false; out.pdf = Fr; out.specular = true;
            return true;
        }
        vec3 w_i;
        if (!refract(w_o, n, c, bx.eta_i, bx.eta_t, w_i)) return false;
        out.w_i = w_i;
        out.f = tex_color<F>(sc, bx.tex_b, u, v) * (1.0 - Fr) / fabs(c);
        out.delta = false; out.pdf = 1.0 - Fr; out.specular = true;
        return true;
    }
    return false;  // unreachable for a scene inside the mask
}

// material < 0 is the black matte of an AreaLightPrimitive (primitive.rs:40-46): a Lambertian lobe
// with reflectance Constant(BLACK); kept implicit so that no table entry is needed.
// Only Lambertian and Oren-Nayar lobes have a non-zero f() (bxdf.rs: every specular lobe returns BLACK from f and
// 0 from pdf); the implicit material of an area light's own surface is a black Lambertian (primitive.rs:40-46).
// For a material without such a lobe the NEE term  beta * Li * f * cos * w / pdf  is exactly (0,0,0) whenever its
// other factors are finite.
__device__ inline bool material_has_diffuse_lobe(const DevScene& sc, int32_t mat) {
    if (mat < 0) return false;
    const cray_material& m = sc.materials[mat];
    const int nb = m.is_bsdf ? m.n_bxdfs : 1;
    for (int i = 0; i < nb; i++)
        if (sc.bxdfs[m.first_bxdf + i].kind <= CRAY_BXDF_OREN_NAYAR) return true;
    return false;
}

template <uint32_t F = SF_ALL>
__device__ inline rgb material_f(const DevScene& sc, int32_t mat, vec3 w_o, vec3 w_i, vec3 n, double u, double v) {
    const rgb zero = mkc(0, 0, 0);
    if (mat < 0) return same_side(n, w_o, w_i) ? zero * kInvPi : zero;
    const cray_material& m = sc.materials[mat];
    if (!m.is_bsdf) return lobe_f<F>(sc, sc.bxdfs[m.first_bxdf], w_o, w_i, n, u, v);
    bool reflecting = dot(w_o, n) * dot(w_i, n) > 0.0;  // bsdf.rs:60
    rgb f = zero;
    const int nb = CRAY_HAS(F, SF_MULTI_LOBE) ? m.n_bxdfs : 1;
    for (int i = 0; i < nb; i++) {
        const cray_bxdf& bx = sc.bxdfs[m.first_bxdf + i];
        if (reflecting ? lobe_reflects(bx.kind) : lobe_transmits(bx.kind)) f = f + lobe_f<F>(sc, bx, w_o, w_i, n, u, v);
    }
    return f;
}
template <uint32_t F = SF_ALL>
__device__ inline bool material_pdf(const DevScene& sc, int32_t mat, vec3 w_o, vec3 w_i, vec3 n, double& pdf) {
    if (mat < 0) { pdf = kInvPi * fabs(dot(w_i, n)); return true; }
    const cray_material& m = sc.materials[mat];
    if (!m.is_bsdf) return lobe_pdf(sc.bxdfs[m.first_bxdf], w_i, n, pdf);
    bool reflecting = dot(w_o, n) * dot(w_i, n) > 0.0;
    double acc = 0.0;
    int matching = 0;
    const int nb = CRAY_HAS(F, SF_MULTI_LOBE) ? m.n_bxdfs : 1;
    for (int i = 0; i < nb; i++) {
        const cray_bxdf& bx = sc.bxdfs[m.first_bxdf + i];
        if (!(reflecting ? lobe_reflects(bx.kind) : lobe_transmits(bx.kind))) continue;
        double p;
        if (lobe_pdf(bx, w_i, n, p)) { acc += p; matching++; }
    }
    if (matching > 0) { pdf = acc / (double)matching; return true; }
    return false;
}
// Material::sample / BSDF::sample (material.rs:72-83, bsdf.rs:15-55)
template <uint32_t F = SF_ALL>
__device__ inline bool material_sample(const DevScene& sc, int32_t mat, double s1d, double s0, double s1, vec3 w_o, vec3 n, double u, double v, LobeSample& out) {
    if (mat < 0) {
        vec3 w_i = cosine_hemisphere(s0, s1, n);
        if (dot(n, w_o) < 0.0) w_i = flip(w_i);
        out.w_i = w_i;
        out.f = same_side(n, w_o, w_i) ? mkc(0, 0, 0) * kInvPi : mkc(0, 0, 0);
        out.delta = false; out.pdf = kInvPi * fabs(dot(w_i, n)); out.specular = false;
        return true;
    }
    const cray_material& m = sc.materials[mat];
    if (!m.is_bsdf) return lobe_sample<F>(sc, sc.bxdfs[m.first_bxdf], s0, s1, w_o, n, u, v, out);
    if (!CRAY_HAS(F, SF_MULTI_LOBE)) {
        // every BSDF of the scene holds exactly one lobe: pick = floor(u * 1) = 0 (u < 1), nothing to add, pdf / 1.0
        if (!lobe_sample<F>(sc, sc.bxdfs[m.first_bxdf], s0, s1, w_o, n, u, v, out)) return false;
        if (!out.delta) out.pdf = out.pdf / 1.0;
        return true;
    }
    if (m.n_bxdfs == 0) return false;
    int pick = (int)to_u64_sat(s1d * (double)m.n_bxdfs);
    LobeSample s;
    if (!lobe_sample<F>(sc, sc.bxdfs[m.first_bxdf + pick], s0, s1, w_o, n, u, v, s)) return false;
    if (s.delta) { out = s; return true; }  // Delta samples are returned unscaled (bsdf.rs:51-53)
    double pdf = s.pdf;
    rgb f = s.f;
    bool reflecting = dot(w_o, n) * dot(s.w_i, n) > 0.0;
    for (int i = 0; i < m.n_bxdfs; i++) {
        const cray_bxdf& other = sc.bxdfs[m.first_bxdf + i];
        if (i == pick || !(reflecting ? lobe_reflects(other.kind) : lobe_transmits(other.kind))) continue;
        f = f + lobe_f<F>(sc, other, w_o, s.w_i, n, u, v);
        double op;
        if (lobe_pdf(other, s.w_i, n, op)) pdf += op;
    }
    out.w_i = s.w_i; out.f = f; out.delta = false; out.pdf = pdf / (double)m.n_bxdfs; out.specular = s.specular;
    return true;
}

// =============================================================================
// Lights (src/light.rs)
// =============================================================================
__device__ __forceinline__ double light_select_pdf(const DevScene& sc, uint32_t i) {  // LightSampler::pdf :213-219
    return i > 0 ? sc.light_cdf[i] - sc.light_cdf[i - 1] : sc.light_cdf[i];
}
__device__ __forceinline__ int total_order(double a, double b) {  // f64::total_cmp
    long long x = __double_as_longlong(a), y = __double_as_longlong(b);
    x ^= (long long)(((unsigned long long)(x >> 63)) >> 1);
    y ^= (long long)(((unsigned long long)(y >> 63)) >> 1);
    return x < y ? -1 : (x > y ? 1 : 0);
}
// LightSampler::sample (:203-211): binary search, exact match -> that index, else insertion point
template <uint32_t F = SF_ALL>
__device__ inline uint32_t light_select(const DevScene& sc, double u, double& pdf) {
    if (!CRAY_HAS(F, SF_MANY_LIGHTS)) { pdf = light_select_pdf(sc, 0); return 0; }  // one light: the search ends at index 0 for every u
    uint32_t lo = 0, hi = sc.n_lights, idx = 0;
    bool found = false;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        int c = total_order(sc.light_cdf[mid], u);
        if (c == 0) { idx = mid; found = true; break; }
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    if (!found) idx = lo;
    pdf = light_select_pdf(sc, idx);
    return idx;
}

}  // namespace cray
