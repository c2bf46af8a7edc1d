// cray_math.h — f64 math shared by the host scene builder and the gfx950 kernels.
//
// Everything on craytracer's hot path is f64 with one IEEE operation per source
// operator (rustc never contracts to FMA), so this header is compiled with
// -ffp-contract=off on both sides and spells every expression in the reference's
// operand order.  f64 + - * / sqrt are correctly rounded on gfx950, which is what
// makes the traversal and intersection results bit-identical to the CPU's.
//
// Reference: src/geometry.rs (Vector/Point/Normal), src/transformation.rs
// (Matrix/Transformation/Transformable), src/bounds.rs, src/ray.rs, src/color.rs.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CRAY_HD __host__ __device__ __forceinline__
#else
#define CRAY_HD inline
#endif

namespace cray {

constexpr double kEps = 1e-9;  // src/constants.rs:1
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kInvPi = 0.318309886183790671537767526745028724;
constexpr double kHalfPi = 1.57079632679489661923132169163975144;
constexpr double kQuarterPi = 0.785398163397448309615660845819875721;

CRAY_HD double inf64() { return __builtin_huge_val(); }

// Rust f64::min / f64::max semantics (NaN-ignoring) == IEEE minNum / maxNum
CRAY_HD double min_nn(double a, double b) { return fmin(a, b); }
CRAY_HD double max_nn(double a, double b) { return fmax(a, b); }
CRAY_HD bool sign_neg(double x) { return __builtin_signbit(x); }          // f64::is_sign_negative
CRAY_HD double signum1(double x) { return x != x ? x : copysign(1.0, x); } // f64::signum
CRAY_HD double square(double x) { return x * x; }                          // powf(2.0) after LLVM folding
// powf(0.5) as LLVM lowers the pow intrinsic without fast-math: -inf -> +inf, else |sqrt(x)|
CRAY_HD double pow_half(double x) { return x == -inf64() ? inf64() : fabs(sqrt(x)); }
CRAY_HD double deg2rad(double d) { return d * (kPi / 180.0); }             // f64::to_radians
// Rust `as usize` / `as u32`: saturating, NaN -> 0
CRAY_HD uint64_t to_u64_sat(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551616.0) return ~0ull;
    return (uint64_t)x;
}
CRAY_HD uint32_t to_u32_sat(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 4294967296.0) return ~0u;
    return (uint32_t)x;
}

struct vec3 {
    double x, y, z;
};
CRAY_HD vec3 mk(double x, double y, double z) { return vec3{x, y, z}; }
CRAY_HD vec3 operator+(vec3 a, vec3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
CRAY_HD vec3 operator-(vec3 a, vec3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
CRAY_HD vec3 operator*(vec3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
CRAY_HD vec3 operator/(vec3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
CRAY_HD vec3 flip(vec3 a) { return a * -1.0; }  // Neg is `self * -1.0` (geometry.rs:150-156)
CRAY_HD double dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
CRAY_HD double len2(vec3 a) { return dot(a, a); }
CRAY_HD double len(vec3 a) { return sqrt(len2(a)); }
CRAY_HD vec3 unit(vec3 a) {  // normalized(): three divisions (geometry.rs:54-57)
    double m = len(a);
    return mk(a.x / m, a.y / m, a.z / m);
}
CRAY_HD vec3 cross(vec3 a, vec3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
CRAY_HD double comp(vec3 a, int axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }
CRAY_HD bool same_side(vec3 n, vec3 a, vec3 b) { return dot(n, a) * dot(n, b) > 0.0; }  // geometry.rs:403-405
// Normal::generate_tangents (geometry.rs:406-417)
CRAY_HD void tangents(vec3 n, vec3& t, vec3& b) {
    vec3 v = unit(n);
    double sign = signum1(v.z);
    double a = -1.0 / (sign + v.z);
    double bb = v.x * v.y * a;
    t = mk(1.0 + sign * v.x * v.x * a, sign * bb, -sign * v.x);
    b = mk(bb, sign + v.y * v.y * a, -v.y);
}

// ---- Color ------------------------------------------------------------------
struct rgb {
    double r, g, b;
};
CRAY_HD rgb mkc(double r, double g, double b) { return rgb{r, g, b}; }
CRAY_HD rgb operator+(rgb a, rgb b) { return mkc(a.r + b.r, a.g + b.g, a.b + b.b); }
CRAY_HD rgb operator-(rgb a, rgb b) { return mkc(a.r - b.r, a.g - b.g, a.b - b.b); }
CRAY_HD rgb operator*(rgb a, rgb b) { return mkc(a.r * b.r, a.g * b.g, a.b * b.b); }
CRAY_HD rgb operator/(rgb a, rgb b) { return mkc(a.r / b.r, a.g / b.g, a.b / b.b); }
CRAY_HD rgb operator*(rgb a, double s) { return mkc(a.r * s, a.g * s, a.b * s); }
CRAY_HD rgb operator/(rgb a, double s) { return mkc(a.r / s, a.g / s, a.b / s); }
CRAY_HD bool black(rgb c) { return c.r == 0.0 && c.g == 0.0 && c.b == 0.0; }
CRAY_HD bool finite3(rgb c) { return isfinite(c.r) && isfinite(c.g) && isfinite(c.b); }
CRAY_HD rgb pow_half3(rgb c) { return mkc(pow_half(c.r), pow_half(c.g), pow_half(c.b)); }

// ---- Ray ----------------------------------------------------------------------
struct ray_t {
    vec3 o, d;
    double tmax;
};
CRAY_HD ray_t mkray(vec3 o, vec3 d) { return ray_t{o, d, inf64()}; }
CRAY_HD vec3 at(const ray_t& r, double t) { return r.o + r.d * t; }
CRAY_HD bool in_range(const ray_t& r, double t) { return t > kEps && t < r.tmax; }  // ray.rs:26-28
CRAY_HD bool shrink(ray_t& r, double t) {                                            // ray.rs:30-37
    if (in_range(r, t)) {
        r.tmax = t;
        return true;
    }
    return false;
}

// ---- Matrix / Transformation -------------------------------------------------------
struct mat4 {
    double m[4][4];
};
struct xform {
    mat4 fwd, inv;
};
// Transformable<Point>: homogeneous divide always performed (transformation.rs:418-428)
CRAY_HD vec3 xf_point(const double* m /*16, row-major*/, vec3 p) {
    vec3 r = mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
                m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
    // Transformable<Point> divides by w (transformation.rs:420-429).  Every transformation the path applies to a
    // point is affine (translate / rotate compositions): w is exactly 1.0 and x / 1.0 == x bit for bit, so the three
    // divisions are skipped then; anything else (NaN, a projective matrix) takes the division.
    const double w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (w == 1.0) return r;
    return r / w;
}
CRAY_HD vec3 xf_vector(const double* m, vec3 v) {  // :431-440
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
CRAY_HD vec3 xf_normal(const double* inv, vec3 n) {  // inverse-transpose, :443-453
    return mk(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z, inv[1] * n.x + inv[5] * n.y + inv[9] * n.z,
              inv[2] * n.x + inv[6] * n.y + inv[10] * n.z);
}
CRAY_HD ray_t xf_ray(const double* m, const ray_t& r) {  // :456-462
    ray_t out = mkray(xf_point(m, r.o), xf_vector(m, r.d));
    shrink(out, r.tmax);
    return out;
}

}  // namespace cray
// -----------------------------------------------------------------------------------------
// Exact division by a per-ray constant (used by the slab tests of the traversal kernel).
//
// The reference divides: (bound - origin) / direction, six times per box.  A correctly rounded
// f64 division costs ~11 dependent VALU instructions on gfx950 (v_div_scale x2, v_rcp, 5 fma,
// v_div_fmas, v_div_fixup).  With y = RN(1/d) computed ONCE per ray by a true division, the
// quotient is recovered exactly by two FMA-based correction steps (Markstein 1990; Muller et al.,
// Handbook of Floating-Point Arithmetic, Newton-Raphson division with FMA):
//     q0 = RN(a*y)                         (within 2 ulp of a/d)
//     q1 = RN(q0 + (a - q0*d) * y)         (faithful: within 1 ulp; the residual is exact in an FMA)
//     q2 = RN(q1 + (a - q1*d) * y)         (= RN(a/d) by Markstein's theorem: y correctly rounded, q1 faithful)
// valid when nothing over/underflows: the caller guarantees that the bounds and the ray origin are zero or
// within [2^-500, 2^500] in magnitude (div_range_ok, so a = bound - origin is 0 or >= 2^-553) and that
// 2^-500 <= |d| <= 2^100 (div_fast_ok): every quotient is then a normal number with the sign of a, or an
// exact zero for a == 0.  Otherwise the kernel divides plainly.
// tests/test_host_and_abi.py checks q2 == a/d bit for bit on random and adversarial operands.
// -----------------------------------------------------------------------------------------
namespace cray {
CRAY_HD bool div_range_ok(double x) {  // 0 or 2^-500 <= |x| <= 2^500
    const double ax = fabs(x);
    return x == 0.0 || (ax >= 0x1p-500 && ax <= 0x1p500);
}
CRAY_HD bool div_fast_ok(double d) { const double ad = fabs(d); return ad >= 0x1p-500 && ad <= 0x1p100; }
CRAY_HD double div_fast(double a, double d, double y) {
    const double q0 = a * y;
    const double q1 = fma(fma(-q0, d, a), y, q0);
    return fma(fma(-q1, d, a), y, q1);
}
}  // namespace cray


namespace cray {
// ---------------------------------------------------------------------------------
// Slab test of one child box, split into its ray.tmax-independent part.
// Bounds::intersects (bounds.rs:62-88) returns
//     ok && (in(tmin) || in(tmax)),  in(t) = t > EPS && t < ray.tmax
// where ok = no early-out fired.  tmin/tmax/ok do not depend on ray.tmax, so a child is
// summarised by one key:  accepted  <=>  key < ray.tmax
//     key = -inf                      if Bounds::contains(origin)        (bvh.rs:70)
//         = +inf                      if an early-out fired
//         = min(tmin if > EPS else +inf, tmax if > EPS else +inf)        otherwise.
// The far child is re-checked against the *shrunken* ray.tmax when it is popped, exactly
// when the reference tests it.  The sequential early-outs equal one final test because
// tmax only decreases and tmin only increases over the three axes.
// ---------------------------------------------------------------------------------
CRAY_HD double child_key(const double* __restrict__ lo, const double* __restrict__ hi, vec3 o, vec3 d) {
    double tmin = -inf64(), tmax = inf64();
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
        double d_i = ax == 0 ? d.x : (ax == 1 ? d.y : d.z);
        double o_i = ax == 0 ? o.x : (ax == 1 ? o.y : o.z);
        double mn = lo[ax], mx = hi[ax];
        if (sign_neg(d_i)) { double t = mn; mn = mx; mx = t; }
        tmax = min_nn(tmax, (mx - o_i) / d_i);
        tmin = max_nn(tmin, (mn - o_i) / d_i);
    }
    bool ok = !(tmax < kEps) && !(tmin > tmax);
    bool inside = lo[0] <= o.x && lo[1] <= o.y && lo[2] <= o.z && hi[0] >= o.x && hi[1] >= o.y && hi[2] >= o.z;
    double a = tmin > kEps ? tmin : inf64();
    double b = tmax > kEps ? tmax : inf64();
    double key = ok ? min_nn(a, b) : inf64();
    return inside ? -inf64() : key;
}

// child_key for rays and scenes inside div_fast's guarded range (cray_math.h): identical result, ~60 instead of
// ~110 VALU instructions.
//  * the six divisions are the exact FMA sequence div_fast (rd = 1/d per axis, once per ray);
//  * in that range every quotient is finite, non-NaN and has the sign of its numerator (zero only for a zero
//    numerator: nothing underflows), so
//      - swapping (min,max) by the sign of d and then dividing equals min/max of the two quotients
//        (division by a fixed d is monotone under correct rounding);
//      - Bounds::contains(origin) (lo <= o <= hi on every axis) is  tmin <= 0 <= tmax;
//      - the early-outs / in-range selection collapse to the three comparisons below.
CRAY_HD double child_key_fast(const double* __restrict__ lo, const double* __restrict__ hi, vec3 o, vec3 d, vec3 rd) {
    const double ax0 = div_fast(lo[0] - o.x, d.x, rd.x), bx0 = div_fast(hi[0] - o.x, d.x, rd.x);
    const double ax1 = div_fast(lo[1] - o.y, d.y, rd.y), bx1 = div_fast(hi[1] - o.y, d.y, rd.y);
    const double ax2 = div_fast(lo[2] - o.z, d.z, rd.z), bx2 = div_fast(hi[2] - o.z, d.z, rd.z);
    const double tmin = fmax(fmax(fmin(ax0, bx0), fmin(ax1, bx1)), fmin(ax2, bx2));
    const double tmax = fmin(fmin(fmax(ax0, bx0), fmax(ax1, bx1)), fmax(ax2, bx2));
    const bool inside = tmin <= 0.0 && tmax >= 0.0;
    double key = tmax > kEps ? tmax : inf64();
    key = tmin > kEps ? tmin : key;
    key = tmin > tmax ? inf64() : key;
    return inside ? -inf64() : key;
}

// child_key_fast with the two special values ENCODED instead of materialised (round 3): the traversal only ever asks
// `key < ray.tmax` (now, or when the far child is popped), with ray.tmax > 0.  So "contains(origin)" may be ANY value below
// every ray.tmax and "rejected" ANY value for which the comparison is false:
//     contains(origin)  ->  high word 0xffe00000: -(2^1023 .. 2^1024), finite, below every ray.tmax > -1.7e308
//     rejected          ->  high word 0x7ff80000: a NaN, `key < x` is false for every x
// which takes one 32-bit select each instead of a 64-bit select against +-inf (13 -> 9 VALU instructions per child box).
// canonical_key() maps a code back to child_key's value; cray_host_child_key_mismatches holds the two together.
CRAY_HD double child_key_code(const double* __restrict__ lo, const double* __restrict__ hi, vec3 o, vec3 d, vec3 rd) {
    const double ax0 = div_fast(lo[0] - o.x, d.x, rd.x), bx0 = div_fast(hi[0] - o.x, d.x, rd.x);
    const double ax1 = div_fast(lo[1] - o.y, d.y, rd.y), bx1 = div_fast(hi[1] - o.y, d.y, rd.y);
    const double ax2 = div_fast(lo[2] - o.z, d.z, rd.z), bx2 = div_fast(hi[2] - o.z, d.z, rd.z);
    const double tmin = fmax(fmax(fmin(ax0, bx0), fmin(ax1, bx1)), fmin(ax2, bx2));
    const double tmax = fmin(fmin(fmax(ax0, bx0), fmax(ax1, bx1)), fmax(ax2, bx2));
    const bool inside = tmin <= 0.0 && tmax >= 0.0;
    const bool reject = tmin > tmax || !(tmax > kEps);
    const double k2 = tmin > kEps ? tmin : tmax;
    unsigned long long bits;
    __builtin_memcpy(&bits, &k2, 8);
    uint32_t hi32 = (uint32_t)(bits >> 32);
    hi32 = reject ? 0x7ff80000u : hi32;
    hi32 = inside ? 0xffe00000u : hi32;
    bits = ((unsigned long long)hi32 << 32) | (bits & 0xffffffffull);
    double out;
    __builtin_memcpy(&out, &bits, 8);
    return out;
}
CRAY_HD double canonical_key(double code) { return code != code ? inf64() : (code < -0x1p1022 ? -inf64() : code); }


// ---------------------------------------------------------------------------------
// Certified f32 culling ("hybrid" records).
//
// The traversal decides  key < ray.tmax  for every child box, with key = child_key (exact f64).  Most of those decisions are
// nowhere near a tie, so they can be taken from a 64-B record of f32 bounds — HALF the bytes of the f64 record — provided the
// f32 arithmetic carries a rigorous enclosure of the f64 values and every decision it cannot certify is retaken exactly from
// the f64 record.  The result of the traversal (hits, node and primitive counters) is then bit for bit the f64 result.
//
// Notation: u = 2^-24.  For a ray inside hyb_ray_ok's range and a scene whose bounds are within 2^40:
//   b32 = the f64 bound b rounded outward to f32            |b32 - b| <= 2u|b|        (k_make_inner32)
//   r32 = RN32(RN64(1/d_i)),  p32 = RN32(RN64(o_i * RN64(1/d_i)))      |r32 d - 1| <= u(1 + 2^-28),  |p32 d / o - 1| <= u(1 + 2^-27)
//   qt  = RN32(b32 * r32 - p32)                             the f32 slab quotient: ONE fma (round 4; rounds 2-3: (b32 - o32) * r32)
//   q   = RN64(RN64(b - o) / d)                             the reference's quotient (= div_fast, exactly)
// With Q = (b - o)/d:  b32 r32 - p32 = Q + (b/d)(beta + rho + beta rho) - (o/d) pi  with |beta| <= 2u, |rho|, |pi| <= u(1 + 2^-27), and
// |b/d| <= |Q| + |o/d|; the fma rounds once more (1 + delta, |delta| <= u) and q = Q(1 + 2^-52), hence
//        |qt - q| <= 4.01u |qt| + 4.01u |o_i| / |d_i|.
// f+-(x) = x +- (a + c|x|) with c = 8u, a = 5u max_i |o_i| |1/d_i| (>= 2^-100: absorbs f32 underflow of qt and p32) are increasing,
// so min / max over the axes commute with them:
//   tmin in [f-(tmin32), f+(tmin32)],  tmax in [f-(tmax32), f+(tmax32)],   tmin32 / tmax32 = the plain f32 slab results.
// The spare 3.99u|x| and 0.99u|o|/|d| cover the roundings of evaluating f+- themselves in f32 (and of a).
//
// hyb_key classifies a box from the two enclosures (tl,th), (xl,xh) and returns an ENCODED key estimate kc:
//   +inf    certainly no intersection (tmin > tmax, or tmax < 0)                      key = +inf
//   -1e-30  the origin is certainly inside (tmin <= 0 <= tmax)                          key = -inf
//   x > 0   certainly tmin > EPS and tmin <= tmax:                                      key = tmin in [f-(x), f+(x)]
//   x < 0   certainly tmax > EPS and tmin <= tmax, tmin near 0 or EPS:                  key in {-inf, tmin, tmax} <= f+(|x|)
//   NaN     nothing certain (grazing boxes, out-of-range rays: a = NaN)
// hyb_status turns kc into a decision against ray.tmax in [T_lo, T_hi]:  V(isit)  key < tmax certainly,  C(ull)  key >= tmax
// certainly,  R(esolve)  neither — the kernel then fetches the f64 bounds of that child and compares child_key exactly.
// tests/test_hybrid_key.py checks on random and adversarial boxes (origins on faces, flat boxes, tmax equal to the key) that a
// V or C is never wrong, on the host build of these same functions.
// ---------------------------------------------------------------------------------
// outward rounding of a bound to f32 (the records k_make_inner32 writes)
CRAY_HD float f32_down(double x) {
#ifdef __HIP_DEVICE_COMPILE__
    return __double2float_rd(x);
#else
    const float f = (float)x;
    return (double)f > x ? nextafterf(f, -__builtin_huge_valf()) : f;
#endif
}
CRAY_HD float f32_up(double x) {
#ifdef __HIP_DEVICE_COMPILE__
    return __double2float_ru(x);
#else
    const float f = (float)x;
    return (double)f < x ? nextafterf(f, __builtin_huge_valf()) : f;
#endif
}
struct HybRay {   // (scalars, not arrays: a struct of arrays in a kernel's lane state gets promoted to LDS by the compiler)
    float px, py, pz, rx, ry, rz;   // o / d and 1 / d per axis, each rounded once from f64: a slab quotient is fma(bound, r, -p)
    float a;   // NaN: this ray is outside the certified range, every decision is resolved exactly
};
constexpr float kHybC = 0x1p-21f;     // 8u
constexpr float kHybEpsUp = 1.1e-9f;  // > EPS (1e-9) as a f32 threshold
enum { kHybResolve = 0, kHybVisit = 1, kHybCull = 2 };

CRAY_HD bool hyb_scene_ok(const double* root_lo, const double* root_hi) {  // every node's bounds lie inside the root's
    bool ok = true;
    for (int k = 0; k < 3; k++) ok = ok && fabs(root_lo[k]) <= 0x1p40 && fabs(root_hi[k]) <= 0x1p40;   // false for NaN
    return ok;
}
CRAY_HD HybRay hyb_ray(vec3 o, vec3 d, vec3 rd, bool fast_div) {
    HybRay h;
    h.px = (float)(o.x * rd.x); h.py = (float)(o.y * rd.y); h.pz = (float)(o.z * rd.z);
    h.rx = (float)rd.x; h.ry = (float)rd.y; h.rz = (float)rd.z;
    const double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
    const bool ok = fast_div && ax >= 0x1p-30 && ax <= 0x1p30 && ay >= 0x1p-30 && ay <= 0x1p30 && az >= 0x1p-30 && az <= 0x1p30 &&
                    fabs(o.x) <= 0x1p40 && fabs(o.y) <= 0x1p40 && fabs(o.z) <= 0x1p40;
    const double m = fmax(fmax(fabs(o.x) * fabs(rd.x), fabs(o.y) * fabs(rd.y)), fabs(o.z) * fabs(rd.z));
    const float a = (float)(m * 0x1.4p-22);   // 5u
    h.a = ok ? fmaxf(a, 0x1p-100f) : __builtin_nanf("");
    return h;
}
// [T_lo, T_hi] around ray.tmax; everything is resolved exactly for a tmax that f32 cannot bracket this way
CRAY_HD void hyb_tmax(double t, float& t_lo, float& t_hi) {
    const float x = (float)t;
    const bool ok = t >= 1e-30;   // false for NaN
    t_lo = ok ? x * (1.0f - 0x1p-22f) : -__builtin_huge_valf();
    t_hi = ok ? x * (1.0f + 0x1p-22f) : __builtin_huge_valf();
}
CRAY_HD float hyb_key(const float* lo, const float* hi, const HybRay& h) {
    const float ax = fmaf(lo[0], h.rx, -h.px), bx = fmaf(hi[0], h.rx, -h.px);
    const float ay = fmaf(lo[1], h.ry, -h.py), by = fmaf(hi[1], h.ry, -h.py);
    const float az = fmaf(lo[2], h.rz, -h.pz), bz = fmaf(hi[2], h.rz, -h.pz);
    const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float e0 = fmaf(kHybC, fabsf(tmin), h.a), e1 = fmaf(kHybC, fabsf(tmax), h.a);
    const float tl = tmin - e0, th = tmin + e0, xl = tmax - e1, xh = tmax + e1;
    float kc = __builtin_nanf("");
    if (th <= xl && xl > kHybEpsUp) kc = tl > kHybEpsUp ? tmin : -tmax;
    if (th <= 0.0f && xl >= 0.0f) kc = -1e-30f;
    if (tl > xh || xh < 0.0f) kc = __builtin_huge_valf();
    return kc;
}
CRAY_HD int hyb_status(float kc, float a, float t_lo, float t_hi) {   // a popped entry against the current ray.tmax; no branches
    const float m = fabsf(kc), e = fmaf(kHybC, m, a);
    const bool visit = m + e < t_lo;
    // (kc = +inf, "certainly no intersection": m - e is inf - inf = NaN, so `not below t_hi` holds without a comparison of its own;
    // t_hi is never NaN)
    const bool cull = kc > 0.0f && !(m - e < t_hi);
    return cull ? kHybCull : (visit ? kHybVisit : kHybResolve);
}

// Both children of a node at once, as the kernel evaluates them: the record holds the two children's bounds interleaved
// (InnerNodeH: lo[axis][child], hi[axis][child]), so the slab arithmetic is one packed instruction (v_pk_fma_f32) per bound for
// both boxes.  Same classification as hyb_key + hyb_status, fused: the enclosure ends are the very values hyb_status would
// recompute from kc.  The children are put in the ray's order (bvh.rs:92-98) BEFORE they are classified, so the step gets what
// it branches on — may the near child be entered, is it certain; the same for the far child and the key estimate it is deferred
// with — as conditions, not as codes to select and compare again, and the near child's key estimate is never formed.
typedef float hyb_f2 __attribute__((ext_vector_type(2)));
struct HybPair {
    bool visit_n, cull_n;   // near child: key < tmax certainly / key >= tmax certainly (neither: RESOLVE)
    bool visit_f, cull_f;   // far child
    float kc_f;             // the far child's encoded key estimate (what it carries on the stack)
};
CRAY_HD hyb_f2 hyb_slab(hyb_f2 b, float r, float p) {
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_elementwise_fma(b, hyb_f2{r, r}, hyb_f2{-p, -p});
#else
    return hyb_f2{fmaf(b[0], r, -p), fmaf(b[1], r, -p)};
#endif
}
// (the ray's f32 view as seven scalars: the kernel keeps them in registers, an aggregate argument makes the compiler build it in LDS)
CRAY_HD HybPair hyb_pair(hyb_f2 lox, hyb_f2 loy, hyb_f2 loz, hyb_f2 hix, hyb_f2 hiy, hyb_f2 hiz, float h_px, float h_py, float h_pz,
                         float h_rx, float h_ry, float h_rz, float h_a, float t_lo, float t_hi, bool right_first) {
    const hyb_f2 ax = hyb_slab(lox, h_rx, h_px), bx = hyb_slab(hix, h_rx, h_px);
    const hyb_f2 ay = hyb_slab(loy, h_ry, h_py), by = hyb_slab(hiy, h_ry, h_py);
    const hyb_f2 az = hyb_slab(loz, h_rz, h_pz), bz = hyb_slab(hiz, h_rz, h_pz);
    float tmn[2], tmx[2];
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
    for (int c = 0; c < 2; c++) {
        tmn[c] = fmaxf(fmaxf(fminf(ax[c], bx[c]), fminf(ay[c], by[c])), fminf(az[c], bz[c]));
        tmx[c] = fminf(fminf(fmaxf(ax[c], bx[c]), fmaxf(ay[c], by[c])), fmaxf(az[c], bz[c]));
    }
    HybPair out;
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
    for (int k = 0; k < 2; k++) {   // 0: the near child, 1: the far one
        const bool second = right_first ? k == 0 : k == 1;
        const float tmin = second ? tmn[1] : tmn[0], tmax = second ? tmx[1] : tmx[0];
        const float e0 = fmaf(kHybC, fabsf(tmin), h_a), e1 = fmaf(kHybC, fabsf(tmax), h_a);
        const float tl = tmin - e0, th = tmin + e0, xl = tmax - e1, xh = tmax + e1;
        const bool ordered = th <= xl && xl > kHybEpsUp;           // certainly tmin <= tmax and tmax > EPS
        const bool front = tl > kHybEpsUp;                          // certainly tmin > EPS
        const bool inside = th <= 0.0f && xl >= 0.0f;               // certainly tmin <= 0 <= tmax
        const bool miss = tl > xh || xh < 0.0f;                     // certainly tmin > tmax, or tmax < 0
        const float upper = inside ? -3e38f : (front ? th : xh);    // key <= upper when `inside` or `ordered`
        const bool visit = (inside || ordered) && upper < t_lo;     // (t_lo = -inf for a ray.tmax that is NaN or not positive)
        const bool cull = miss || (front && tl >= t_hi);
        if (k == 0) { out.visit_n = visit; out.cull_n = cull; }   // (cull is looked at first by every user)
        else {
            out.visit_f = visit; out.cull_f = cull;
            float kc = ordered ? (front ? tmin : -tmax) : __builtin_nanf("");
            kc = inside ? -1e-30f : kc;
            out.kc_f = miss ? __builtin_huge_valf() : kc;
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------
// Certified f32 culling of the TRIANGLE test (round 5) — the leaf analogue of hyb_pair.
//
// Shape::Triangle::intersect (shape.rs:216-262) answers `false` for four out of five triangles a ray is tested against, and the
// exact f64 test costs ~90 instructions, three of them divisions.  The numerators of Moller-Trumbore decide that answer without
// any division:  with s = sign(den),
//     u < 0  <=>  nu s < 0      u > 1  <=>  nu s > |den|      v < 0  <=>  nv s < 0      u + v > 1  <=>  (nu + nv) s > |den|
//     t <= 0 <=>  nt s <= 0     t >= tmax  <=>  nt s >= tmax |den|
// tri_cull32 evaluates den, nu, nv, nt in f32 from f32 copies of d, e1, e2 and from T = RN32(RN64(o - v0)) — the subtraction in
// f64 from the f64 vertex, so that T carries a RELATIVE error only, however close the origin is to the triangle — together with
// a rigorous bound of their distance from the f64 values the reference computes, and says MISS only when one of the six
// conditions holds for every value inside the bounds.  Everything else (hits, near-edge cases, degenerate triangles, rays outside
// the certified range) is `unknown` and takes the exact f64 test.  What is certified is the reference's ANSWER ("returns false"),
// not each comparison: a condition that holds for the real quotients makes the reference return false at that comparison or at
// an earlier one.
//
// Bound (u = 2^-24; * = real arithmetic on the f64 inputs; Dm, E1m, E2m, Tm = max-norms of d, e1, e2, T):
//   inputs        d32, e1_32, e2_32: relative error u;  T32: u(1 + 2^-29)
//   P32 = fl(fma(d_a, e2_b, -fl(d_b e2_a)))    |P32_i - P*_i| <= 2u|d_a e2_b| + 3u|d_b e2_a| + u|P32_i| <= 7.01u Dm E2m,  |P*_i| <= 2 Dm E2m
//   a term P32_i x32_i of a dot product           |P32_i x32_i - P*_i x_i| <= 7.01u Dm E2m Xm (1 + u) + 2 Dm E2m u Xm = 9.02u Dm E2m Xm
//   three terms + three roundings of partial sums (each <= 6.01 Dm E2m Xm)      27.06u + 18.03u = 45.1u Dm E2m Xm
// and the same with (T, e1) in place of (d, e2) for Q = T x e1.  Hence, with K = 47u (4 % spare: the f32 evaluation of the bounds
// themselves, the max-norms taken from rounded values, second-order terms, and the f64 roundings of the reference, ~60 x 2^-53):
//   |den32 - den64| <= K Dm E1m E2m     |nu32 - nu64| <= K Dm E2m Tm     |nv32 - nv64| <= K Dm E1m Tm     |nt32 - nt64| <= K E1m E2m Tm
// each plus 2^-100 for f32 underflow (the caller guarantees |coordinates| <= 2^40 and 2^-30 <= |d_i| <= 2^30: nothing overflows).
// Quotients: u64 = RN(nu64 / den64) > 1 needs nu64 s > |den64| (1 + 2^-52); the comparisons below leave a relative 8u (kCullRel)
// on the larger side, which also covers their own f32 roundings.  t_hi >= ray.tmax comes from hyb_tmax (+inf for a tmax f32
// cannot bracket: then `t >= tmax` is never certified).
// tests/test_tri_cull.py: a certified answer never contradicts the literal f64 test (host build of this function).
// ---------------------------------------------------------------------------------
#ifndef CRAY_CULL_K_UNITS
#define CRAY_CULL_K_UNITS 47.0f
#endif
constexpr float kCullK = CRAY_CULL_K_UNITS * 0x1p-24f;   // (the macro exists for the mutation check of tests/test_tri_cull.py)
constexpr float kCullFloor = 0x1p-100f;
constexpr float kCullRel = 1.0f + 0x1p-21f;
enum { kCullUnknown = 0, kCullMiss = 1, kCullHit = 2 };
struct TriCull {
    bool miss;   // the reference's test certainly returns false
    bool hit;    // (WANT_HIT only) it certainly returns true: every comparison certainly passes
};
CRAY_HD float cull_flip(float x, uint32_t sign_bit) {
    uint32_t b;
    __builtin_memcpy(&b, &x, 4);
    b ^= sign_bit;
    float r;
    __builtin_memcpy(&r, &b, 4);
    return r;
}
// e1m / e2m: max |e1_i| / max |e2_i| of the f64 edges, rounded UP to f32.  ray_ok: the ray is inside the certified range (hyb_ray).
template <bool WANT_HIT>
CRAY_HD TriCull tri_cull32(double v0x, double v0y, double v0z, float e1x, float e1y, float e1z, float e2x, float e2y, float e2z, float e1m, float e2m,
                           double ox, double oy, double oz, float dx, float dy, float dz, float t_lo, float t_hi, bool ray_ok) {
    const float Tx = (float)(ox - v0x), Ty = (float)(oy - v0y), Tz = (float)(oz - v0z);
    const float Px = fmaf(dy, e2z, -(dz * e2y)), Py = fmaf(dz, e2x, -(dx * e2z)), Pz = fmaf(dx, e2y, -(dy * e2x));
    const float den = fmaf(Pz, e1z, fmaf(Py, e1y, Px * e1x));
    const float nu = fmaf(Pz, Tz, fmaf(Py, Ty, Px * Tx));
    const float Qx = fmaf(Ty, e1z, -(Tz * e1y)), Qy = fmaf(Tz, e1x, -(Tx * e1z)), Qz = fmaf(Tx, e1y, -(Ty * e1x));
    const float nv = fmaf(Qz, dz, fmaf(Qy, dy, Qx * dx));
    const float nt = fmaf(Qz, e2z, fmaf(Qy, e2y, Qx * e2x));
    // the four bounds
    const float x = kCullK * fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
    const float y = e1m * e2m;
    const float c = fmaxf(fmaxf(fabsf(Tx), fabsf(Ty)), fabsf(Tz));
    const float xc = x * c;
    const float e_den = fmaf(x, y, kCullFloor), e_nu = fmaf(xc, e2m, kCullFloor), e_nv = fmaf(xc, e1m, kCullFloor), e_nt = fmaf(kCullK * c, y, kCullFloor);
    // numerators with the sign of den folded in; bounds of |den|
    uint32_t sb;
    __builtin_memcpy(&sb, &den, 4);
    sb &= 0x80000000u;
    const float aden = fabsf(den), nus = cull_flip(nu, sb), nvs = cull_flip(nv, sb), nts = cull_flip(nt, sb);
    const bool den_ok = aden > e_den;   // the sign of den64 is the sign of den32 (false for NaN bounds)
    const float d_hi = aden + e_den;
    const float nlu = nus - e_nu, nlv = nvs - e_nv, nlt = nts - e_nt;   // lower ends
    bool miss = nus < -e_nu;                                        // u < 0
    miss = miss || fmaf(-kCullRel, d_hi, nlu) > 0.0f;              // u > 1
    miss = miss || nvs < -e_nv;                                     // v < 0
    miss = miss || fmaf(-kCullRel, d_hi, nlu + nlv) > 0.0f;        // u + v > 1
    miss = miss || nts < -e_nt;                                     // t < 0 (<= EPSILON)
    miss = miss || fmaf(-kCullRel, t_hi * d_hi, nlt) > 0.0f;       // t >= ray.tmax
    TriCull out;
    out.miss = miss && den_ok && ray_ok;
    out.hit = false;
    if (WANT_HIT) {
        // every comparison of the reference certainly passes: |den| >= EPSILON, 0 <= u, 0 <= v, u + v <= 1 (hence u <= 1),
        // EPSILON < t < ray.tmax.  (t_lo <= ray.tmax; kHybEpsUp = 1.1e-9 leaves 10 % on EPSILON)
        const float d_lo = aden - e_den;
        const float nhu = nus + e_nu, nhv = nvs + e_nv, nht = nts + e_nt;   // upper ends
        bool hit = d_lo >= kHybEpsUp;
        hit = hit && nlu >= 0.0f && nlv >= 0.0f;
        hit = hit && fmaf(-kCullRel, nhu + nhv, d_lo) > 0.0f;
        hit = hit && nlt > kHybEpsUp * d_hi;
        hit = hit && fmaf(-kCullRel, nht, t_lo * d_lo) > 0.0f;
        out.hit = hit && ray_ok;
    }
    return out;
}

}  // namespace cray

// -----------------------------------------------------------------------------------------
// Correctly rounded sin / cos for the sampling routines (sample_disk, sample_sphere).
//
// Rust documents f64::sin/cos as platform-precision ("non-deterministic"): on Linux they are
// glibc's (which misrounds ~0.15 % of calls and has an FMA ifunc variant), different in the last
// bit from every other libm.  The reference's shadow-ray leak (scenes/rounding-error.cry)
// amplifies such a 1-ulp difference into a visibly different path about once per 3000 paths.
// To make "same inputs -> same pixels" well defined, the test oracle rounds sin/cos correctly
// (through binary128) and so does this routine: double-double evaluation (~2^-100 relative),
// one final rounding.  Only fma() and + - * are used, so host and gfx950 agree bit for bit.
// Valid for |x| <= 16; the callers stay within [-pi/4, 2 pi].
// Constants generated exactly (tools: see DESIGN.md "libm").
// -----------------------------------------------------------------------------------------
namespace cray {
struct dd {
    double hi, lo;
};
CRAY_HD dd dd_two_sum(double a, double b) {
    double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
CRAY_HD dd dd_quick_two_sum(double a, double b) {
    double s = a + b;
    return dd{s, b - (s - a)};
}
CRAY_HD dd dd_two_prod(double a, double b) {
    double p = a * b;
    return dd{p, fma(a, b, -p)};
}
CRAY_HD dd dd_add(dd a, dd b) {
    dd s = dd_two_sum(a.hi, b.hi), t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick_two_sum(s.hi, s.lo);
}
CRAY_HD dd dd_mul(dd a, dd b) {
    dd p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick_two_sum(p.hi, p.lo);
}

CRAY_HD dd dd_mul_d(dd a, double b) {
    dd p = dd_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return dd_quick_two_sum(p.hi, p.lo);
}
CRAY_HD dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }

// Argument reduction and table lookup shared by the two evaluations below: x = kq pi/2 +- (k/64 + l), |l| <= 2^-7 as a double-double,
// S = sin(k/64), C = cos(k/64) as double-doubles.
struct SinCosArg {
    dd l, S, C;
    bool neg;
    int q;
};
CRAY_HD SinCosArg sincos_reduce(double x) {
    // pi/2 = P1 + P2 + P3 + P4; P1..P3 carry 33 significant bits, so k * Pi is exact for |k| < 2^20
    const double P1 = 0x1.921fb54400000p+0, P2 = 0x1.0b4611a600000p-34, P3 = 0x1.3198a2e000000p-69, P4 = 0x1.b839a252049c1p-104;
    const double kq = rint(x * 0x1.45f306dc9c883p-1);  // nearest multiple of pi/2
    const double t = x - kq * P1;                       // exact: Sterbenz for kq != 0
    dd r = dd_two_sum(t, -(kq * P2));
    r = dd_add(r, dd{-(kq * P3), 0.0});
    r = dd_add(r, dd{-(kq * P4), 0.0});                 // |r| <= pi/4 (+ rounding), ~2^-120 accurate
    // r = +-(h + l), h = k/64 from a table of sin(h), cos(h) as double-doubles, |l| <= 2^-7
    const bool neg = r.hi < 0.0;
    if (neg) r = dd_neg(r);
    const double kf = rint(r.hi * 64.0);
    const int k = (int)kf;
    const dd l = dd_add(r, dd{-(kf * 0.015625), 0.0});
    const double kTab[52][4] = {
        {0x0.0p+0, 0x0.0p+0, 0x1.0000000000000p+0, 0x0.0p+0},
        {0x1.fffaaaaeeeed5p-7, -0x1.2ab639a9f0776p-63, 0x1.fff000155549fp-1, 0x1.28a28a03a5ef3p-55},
        {0x1.ffeaaaeeee86fp-6, -0x1.cd406fb224ae2p-60, 0x1.ffc00155527d3p-1, -0x1.3b54492d89b5bp-55},
        {0x1.7fdc01032fba9p-5, -0x1.599bdf46e997ap-59, 0x1.ff7006bfdf99fp-1, -0x1.8b3b560648d5fp-56},
        {0x1.ffaaaeeed4edbp-5, -0x1.2d16d32684b69p-59, 0x1.ff0015549f4d3p-1, 0x1.328387b99426fp-55},
        {0x1.3facb12d1755bp-4, -0x1.921915299468bp-58, 0x1.fe7034129ef6fp-1, -0x1.cbf4337c96f97p-57},
        {0x1.7f701032550e4p-4, 0x1.afc2d1800501ap-60, 0x1.fdc06bf7e6b9bp-1, 0x1.31902b535f8dbp-55},
        {0x1.bf1b78568391dp-4, 0x1.e91841dea4cc8p-58, 0x1.fcf0c800e99b1p-1, 0x1.ea3d786d186acp-57},
        {0x1.feaaeee86ee36p-4, -0x1.afcb2bcc6f03bp-59, 0x1.fc015527d5bd3p-1, 0x1.b68f35094efb8p-55},
        {0x1.1f0d3d7afceafp-3, -0x1.6ef95099769a5p-57, 0x1.faf22263c4bd3p-1, -0x1.52ace133a2769p-58},
        {0x1.3eb312c5d66cbp-3, 0x1.47d666b66cb91p-57, 0x1.f9c340a7cc428p-1, 0x1.c5b6b063b7462p-55},
        {0x1.5e44fcfa126f3p-3, -0x1.6f443063f89b6p-57, 0x1.f874c2e1eecf6p-1, -0x1.c6514e1332b16p-55},
        {0x1.7dc102fbaf2b5p-3, 0x1.5ab50e23c97c3p-59, 0x1.f706bdf9ece1cp-1, -0x1.698c80c36dcb4p-55},
        {0x1.9d252d0cec312p-3, 0x1.9c43d80b1137dp-58, 0x1.f57948cff6797p-1, 0x1.e3a0d3e03b1d4p-57},
        {0x1.bc6f84edc6199p-3, 0x1.9c1a56a7b0cabp-57, 0x1.f3cc7c3b3d16ep-1, -0x1.21a3ad28a3494p-57},
        {0x1.db9e15fb5a5d0p-3, -0x1.32e20d6cc6fc2p-57, 0x1.f20073086649fp-1, 0x1.b940416c1984bp-56},
        {0x1.faaeed4f31577p-3, -0x1.15d88508e32b8p-57, 0x1.f01549f7deea1p-1, 0x1.d3c1e99e5cafdp-55},
        {0x1.0cd00cef36436p-2, -0x1.9fb0a0c93e2b4p-56, 0x1.ee0b1fbc0f11cp-1, -0x1.bfd2380bbc3b1p-59},
        {0x1.1c37d64c6b876p-2, 0x1.46076fe0dcff4p-56, 0x1.ebe214f76efa8p-1, -0x1.02f9f12ba543ep-55},
        {0x1.2b8ddc43eb49fp-2, 0x1.1553899f2d807p-57, 0x1.e99a4c3a7cd83p-1, -0x1.2264b1bc53ce8p-55},
        {0x1.3ad129769d3d8p-2, 0x1.03d550487839ap-63, 0x1.e733ea0193d40p-1, -0x1.6428b3546ce13p-55},
        {0x1.4a00c9b0f3d20p-2, 0x1.823ba6bb08eadp-56, 0x1.e4af14b2a449cp-1, -0x1.68ca02e8a6833p-55},
        {0x1.591bc9fa2f597p-2, 0x1.7c74bac3fe0cbp-57, 0x1.e20bf49acd6c1p-1, -0x1.660aec7ef636bp-58},
        {0x1.682138a38d7f7p-2, -0x1.d889202444aadp-56, 0x1.df4ab3ebd875ep-1, -0x1.e2d8a7e6736c4p-55},
        {0x1.7710255764214p-2, -0x1.6ead7314bb6cep-57, 0x1.dc6b7eb995912p-1, 0x1.4b364776dcd35p-58},
        {0x1.85e7a12826949p-2, 0x1.8a40e9b5face0p-56, 0x1.d96e82f71a9dcp-1, 0x1.ff61bd5d2039dp-55},
        {0x1.94a6be9f546c5p-2, -0x1.69ce13e683f58p-56, 0x1.d653f073e4040p-1, -0x1.76236434bec37p-55},
        {0x1.a34c91cc50ccap-2, -0x1.a310e3b50cecdp-58, 0x1.d31bf8d8d7c06p-1, 0x1.e60dd3089cbddp-56},
        {0x1.b1d8305321617p-2, -0x1.ae242cb99f519p-56, 0x1.cfc6cfa52ad9fp-1, 0x1.8b5b5508f2a0dp-55},
        {0x1.c048b17b140a3p-2, 0x1.19fe6757e9fa7p-57, 0x1.cc54aa2b2972ep-1, 0x1.4ee162ba83a98p-57},
        {0x1.ce9d2e3d4a51fp-2, -0x1.2fc8a12dae298p-57, 0x1.c8c5bf8ce1a84p-1, 0x1.ab3d1a1590123p-56},
        {0x1.dcd4c15329c9ap-2, 0x1.0d4c6e171fd9ap-56, 0x1.c51a48b8b175ep-1, -0x1.1bbb43b9aa880p-57},
        {0x1.eaee8744b05f0p-2, -0x1.789b43c9b027dp-58, 0x1.c1528065b7d50p-1, -0x1.892111312e828p-55},
        {0x1.f8e99e76abc97p-2, 0x1.9d950af2d00a3p-58, 0x1.bd6ea310294f5p-1, 0x1.31bbcc88c109dp-56},
        {0x1.0362939c69955p-1, -0x1.2d8cd78397b01p-55, 0x1.b96eeef58840ep-1, 0x1.45a3cc78fade0p-58},
        {0x1.0a4021e9e1001p-1, -0x1.6f643a13914f6p-55, 0x1.b553a410c104ep-1, 0x1.8ff7947027a15p-58},
        {0x1.110d0c4b69c3bp-1, 0x1.d918998809981p-55, 0x1.b11d04162a4c6p-1, 0x1.1dd561efbc0c2p-56},
        {0x1.17c8e5f2eedb0p-1, 0x1.35e57102e2488p-57, 0x1.accb526f69de5p-1, 0x1.8fb6a8dd6b6ccp-55},
        {0x1.1e7343236574cp-1, 0x1.22a3fa4f41d5ap-56, 0x1.a85ed4373e02dp-1, 0x1.9be06385ec792p-57},
        {0x1.250bb93788bbbp-1, 0x1.ea3d02457bccep-56, 0x1.a3d7d0352bdcfp-1, -0x1.68dbaeca19669p-55},
        {0x1.2b91dea88421ep-1, -0x1.fa371db216ab0p-55, 0x1.9f368ed912f85p-1, -0x1.1d200c5791606p-55},
        {0x1.32054b148bc4fp-1, 0x1.f6b42095a135bp-55, 0x1.9a7b5a36a6514p-1, 0x1.722cfcc9fa7a9p-55},
        {0x1.386597456282bp-1, -0x1.10fada93b07a8p-56, 0x1.95a67e00cb1fdp-1, -0x1.0befda21f862dp-55},
        {0x1.3eb25d36cd53ap-1, -0x1.be570e1570fc0p-58, 0x1.90b84784ddaf7p-1, -0x1.0feb10ab93b87p-56},
        {0x1.44eb381cf386bp-1, -0x1.3ed6c1e6a5505p-55, 0x1.8bb105a5dc900p-1, 0x1.863e03e9474c1p-55},
        {0x1.4b0fc46aab761p-1, 0x1.0da05738cc59cp-61, 0x1.869108d77a6c6p-1, 0x1.338ffe2bfe9ddp-56},
        {0x1.511f9fd7b351cp-1, -0x1.5c0e861c48831p-55, 0x1.8158a31916d5dp-1, -0x1.de8b90b8228dep-57},
        {0x1.571a6966d59b3p-1, 0x1.c843b4d0fb197p-58, 0x1.7c0827f09e54fp-1, -0x1.c73d6d72aee68p-57},
        {0x1.5cffc16bf8f0dp-1, 0x1.96cb370eb578ap-55, 0x1.769fec655211fp-1, -0x1.827d5cf8c68c5p-57},
        {0x1.62cf49921ac79p-1, -0x1.edd9855b6241ap-55, 0x1.712046fa77678p-1, 0x1.425b0a5029c81p-55},
        {0x1.6888a4e134b2fp-1, -0x1.6b7d37644d5e6p-55, 0x1.6b898fa9efb5dp-1, 0x1.15ac786ccf4b2p-56},
        {0x1.6e2b77c40bde1p-1, -0x1.0e729857fad53p-56, 0x1.65dc1fdeb8cbap-1, -0x1.97c1b47337c77p-58}};
    SinCosArg out;
    out.l = l; out.neg = neg; out.q = ((int)kq) & 3;
    out.S = dd{kTab[k][0], kTab[k][1]}; out.C = dd{kTab[k][2], kTab[k][3]};
    return out;
}

// sin(k/64 + l), cos(k/64 + l) in double-double (~2^-100), one final rounding each: the reference evaluation.
CRAY_HD void sincos_dd_core(const SinCosArg& A, double& sv, double& cv) {
    const dd l = A.l, S = A.S, C = A.C;
    const dd l2 = dd_mul(l, l);
    // sin(l)/l and cos(l): the two leading correction terms in double-double, the (tiny) tails in double
    const double ts = -0x1.a01a01a01a01ap-13 + l2.hi * (0x1.71de3a556c734p-19 + l2.hi * -0x1.ae64567f544e4p-26);       // -1/7! + l2/9! - l2^2/11!
    const double tc = -0x1.6c16c16c16c17p-10 + l2.hi * (0x1.a01a01a01a01ap-16 + l2.hi * (-0x1.27e4fb7789f5cp-22 + l2.hi * 0x1.1eed8eff8d898p-29));  // -1/6! + l2/8! - l2^2/10! + l2^3/12!
    dd ps = dd_add(dd{0x1.1111111111111p-7, 0x1.1111111111111p-63}, dd_mul_d(l2, ts));       // 1/5! + ...
    ps = dd_add(dd{-0x1.5555555555555p-3, -(0x1.5555555555555p-57)}, dd_mul(l2, ps));         // -1/3! + ...
    ps = dd_add(dd{1.0, 0.0}, dd_mul(l2, ps));
    const dd sl = dd_mul(l, ps);
    dd pc = dd_add(dd{0x1.5555555555555p-5, 0x1.5555555555555p-59}, dd_mul_d(l2, tc));       // 1/4! + ...
    pc = dd_add(dd{-0.5, 0.0}, dd_mul(l2, pc));
    const dd cl = dd_add(dd{1.0, 0.0}, dd_mul(l2, pc));
    dd sr = dd_add(dd_mul(S, cl), dd_mul(C, sl));        // sin(h + l)
    const dd cr = dd_add(dd_mul(C, cl), dd_neg(dd_mul(S, sl)));
    sv = sr.hi + sr.lo;
    cv = cr.hi + cr.lo;
}

// The same two values from a SHORT evaluation with a rounding test (Ziv's strategy): the leading terms exactly (two_prod / two_sum),
// the small ones in plain f64.  With z = l^2, f = cos(l) - 1, g = sin(l)/l - 1:
//     sin(h + l) = S.hi + C.hi l.hi + { e(C.hi l.hi) + S.lo + C.hi l.lo + C.lo l.hi + (C.hi l.hi) g + S.hi f },
//     cos(h + l) = C.hi - S.hi l.hi + { ... - (S.hi l.hi) g + C.hi f }.
// Error of the braces: three roundings at the magnitude of the last term (<= 2^-15 |S.hi|, |C.hi|) = 3 x 2^-68 relative to S.hi / C.hi,
// the truncated series 2^-90, the neglected S.lo f, C.lo l.lo ... 2^-69; the result is >= |S.hi| / 2 (k >= 1; for k = 0 every term
// scales with l).  So the candidate hi + lo is within 2^-65 |result| of the true value, and kSinCosEps = 2^-64 |hi| is a safe radius:
// if RN(hi + (lo - eps)) == RN(hi + (lo + eps)) that IS the correctly rounded value, otherwise (2^-10 of the calls) the caller takes the
// double-double evaluation.  tests/test_host_and_abi.py compares the two on 2 x 10^7 arguments and records the largest deviation seen
// (10^9 arguments off-line: no difference, largest deviation 0.26 of the radius).
constexpr double kSinCosEps = 0x1p-64;
CRAY_HD void sincos_fast_parts(const SinCosArg& A, double& s_hi, double& s_lo, double& c_hi, double& c_lo) {
    const double lh = A.l.hi, ll = A.l.lo;
    const dd zz = dd_two_prod(lh, lh);
    const double z = zz.hi, zl = fma(2.0 * lh, ll, zz.lo);   // l^2 = z + zl
    // f = -z/2 + z^2/24 - z^3/720 + z^4/40320 (+ the low part of the leading term), g = -z/6 + z^2/120 - z^3/5040
    const double hf = -0.5 + z * (0x1.5555555555555p-5 + z * (-0x1.6c16c16c16c17p-10 + z * 0x1.a01a01a01a01ap-16));
    const double f = fma(z, hf, -0.5 * zl);
    const double g = z * (-0x1.5555555555555p-3 + z * (0x1.1111111111111p-7 + z * -0x1.a01a01a01a01ap-13));
    const double Sh = A.S.hi, Sl = A.S.lo, Ch = A.C.hi, Cl = A.C.lo;
    const dd ps = dd_two_prod(Ch, lh);
    double ts = ps.lo + Sl;
    ts += Ch * ll + Cl * lh;
    ts = fma(ps.hi, g, ts);
    ts = fma(Sh, f, ts);
    const dd s1 = dd_two_sum(ps.hi, ts), s2 = dd_two_sum(Sh, s1.hi);
    s_hi = s2.hi; s_lo = s2.lo + s1.lo;
    const dd pc = dd_two_prod(-Sh, lh);
    double tc = pc.lo + Cl;
    tc -= Sh * ll + Sl * lh;
    tc = fma(pc.hi, g, tc);
    tc = fma(Ch, f, tc);
    const dd c1 = dd_two_sum(pc.hi, tc), c2 = dd_two_sum(Ch, c1.hi);
    c_hi = c2.hi; c_lo = c2.lo + c1.lo;
}
CRAY_HD bool sincos_fast_core(const SinCosArg& A, double& sv, double& cv) {
    double s_hi, s_lo, c_hi, c_lo;
    sincos_fast_parts(A, s_hi, s_lo, c_hi, c_lo);
    const double se = kSinCosEps * fabs(s_hi), ce = kSinCosEps * fabs(c_hi);
    const double sa = s_hi + (s_lo - se), sb = s_hi + (s_lo + se);
    const double ca = c_hi + (c_lo - ce), cb = c_hi + (c_lo + ce);
    sv = sa; cv = ca;
    return sa == sb && ca == cb;
}

CRAY_HD void sincos_cr(double x, double& s_out, double& c_out) {
    const SinCosArg A = sincos_reduce(x);
    double sv, cv;
    if (!sincos_fast_core(A, sv, cv)) sincos_dd_core(A, sv, cv);
    if (A.neg) sv = -sv;
    const int q = A.q;
    s_out = q == 0 ? sv : (q == 1 ? cv : (q == 2 ? -sv : -cv));
    c_out = q == 0 ? cv : (q == 1 ? -sv : (q == 2 ? -cv : sv));
}
// the double-double evaluation alone (what sincos_cr was before the short path; kept callable for the tests)
CRAY_HD void sincos_cr_dd(double x, double& s_out, double& c_out) {
    const SinCosArg A = sincos_reduce(x);
    double sv, cv;
    sincos_dd_core(A, sv, cv);
    if (A.neg) sv = -sv;
    const int q = A.q;
    s_out = q == 0 ? sv : (q == 1 ? cv : (q == 2 ? -sv : -cv));
    c_out = q == 0 ? cv : (q == 1 ? -sv : (q == 2 ? -cv : sv));
}
}  // namespace cray

namespace cray {
CRAY_HD uint64_t rotl64_hd(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
#define CRAY_SIPROUND_HD(v0, v1, v2, v3)                                              \
    do {                                                                              \
        v0 += v1; v1 = rotl64_hd(v1, 13); v1 ^= v0; v0 = rotl64_hd(v0, 32);           \
        v2 += v3; v3 = rotl64_hd(v3, 16); v3 ^= v2;                                   \
        v0 += v3; v3 = rotl64_hd(v3, 21); v3 ^= v0;                                   \
        v2 += v1; v1 = rotl64_hd(v1, 17); v1 ^= v2; v2 = rotl64_hd(v2, 32);           \
    } while (0)
// =============================================================================
// IndependentSampler (src/sampling.rs:102-146): start_pixel hashes (seed, x, y, sample_index) with DefaultHasher and reseeds a
// StdRng from the 64-bit hash; every draw is rng.sample(Uniform::new(0.0, 1.0)).  The crates behind it are NOT in the container
// (rand 0.8.5, rand_chacha 0.3.1, rand_core 0.6.4), so this restates their PUBLISHED algorithms — PARITY UNPINNED against the crates
// themselves:
//   * DefaultHasher = SipHash-1-3 with zero keys over the four usize values as little-endian u64 (pinned like pixel_seed);
//   * SeedableRng::seed_from_u64: eight steps of a PCG32 (multiplier 6364136223846793005, increment 11634580027462260723, output
//     XSH-RR) fill the 32-byte seed, little-endian;
//   * StdRng = ChaCha12: state = "expand 32-byte k", the key, a 64-bit block counter from 0, a 64-bit stream id 0 (djb layout);
//     12 rounds; blocks are consumed word by word in order (the block function is pinned with rounds = 20 by RFC 8439 2.3.2);
//   * next_u64 = word[i] | word[i + 1] << 32;  Uniform<f64>::sample = (u64 >> 12 | exponent of 1.0) as f64 - 1.0 (scale 1, low 0).
// A pixel sample draws film 2-D, lens 2-D (words 0..7), then 8 draws per path segment (7 for simple_integrator), in order.
// =============================================================================
CRAY_HD uint64_t indep_pixel_hash(uint64_t seed, uint64_t x, uint64_t y, uint64_t sample_index) {
    uint64_t v0 = 0x736f6d6570736575ULL, v1 = 0x646f72616e646f6dULL;
    uint64_t v2 = 0x6c7967656e657261ULL, v3 = 0x7465646279746573ULL;
    uint64_t m[4] = {seed, x, y, sample_index};
    for (int i = 0; i < 4; i++) {
        v3 ^= m[i];
        CRAY_SIPROUND_HD(v0, v1, v2, v3);
        v0 ^= m[i];
    }
    const uint64_t b = 32ULL << 56;   // 32 bytes hashed, no tail bytes
    v3 ^= b;
    CRAY_SIPROUND_HD(v0, v1, v2, v3);
    v0 ^= b;
    v2 ^= 0xff;
    CRAY_SIPROUND_HD(v0, v1, v2, v3);
    CRAY_SIPROUND_HD(v0, v1, v2, v3);
    CRAY_SIPROUND_HD(v0, v1, v2, v3);
    return v0 ^ v1 ^ v2 ^ v3;
}
CRAY_HD uint32_t rotl32(uint32_t x, int b) { return (x << b) | (x >> (32 - b)); }
CRAY_HD void indep_key(uint64_t state, uint32_t key[8]) {   // rand_core seed_from_u64
    for (int i = 0; i < 8; i++) {
        state = state * 6364136223846793005ULL + 11634580027462260723ULL;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        key[i] = (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
}
#define CRAY_CHACHA_QR(a, b, c, d)                         \
    do {                                                   \
        a += b; d ^= a; d = rotl32(d, 16);                 \
        c += d; b ^= c; b = rotl32(b, 12);                 \
        a += b; d ^= a; d = rotl32(d, 8);                  \
        c += d; b ^= c; b = rotl32(b, 7);                  \
    } while (0)
// one ChaCha block: words 12..15 of the state are passed explicitly (64-bit counter + 64-bit stream id for the generator; counter +
// nonce for the RFC's test vector)
CRAY_HD void chacha_block(const uint32_t key[8], uint32_t w12, uint32_t w13, uint32_t w14, uint32_t w15, int double_rounds,
                                    uint32_t out[16]) {
    uint32_t x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u;
    uint32_t x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3], x8 = key[4], x9 = key[5], x10 = key[6], x11 = key[7];
    uint32_t x12 = w12, x13 = w13, x14 = w14, x15 = w15;
    for (int r = 0; r < double_rounds; r++) {
        CRAY_CHACHA_QR(x0, x4, x8, x12); CRAY_CHACHA_QR(x1, x5, x9, x13); CRAY_CHACHA_QR(x2, x6, x10, x14); CRAY_CHACHA_QR(x3, x7, x11, x15);
        CRAY_CHACHA_QR(x0, x5, x10, x15); CRAY_CHACHA_QR(x1, x6, x11, x12); CRAY_CHACHA_QR(x2, x7, x8, x13); CRAY_CHACHA_QR(x3, x4, x9, x14);
    }
    out[0] = x0 + 0x61707865u; out[1] = x1 + 0x3320646eu; out[2] = x2 + 0x79622d32u; out[3] = x3 + 0x6b206574u;
    out[4] = x4 + key[0]; out[5] = x5 + key[1]; out[6] = x6 + key[2]; out[7] = x7 + key[3];
    out[8] = x8 + key[4]; out[9] = x9 + key[5]; out[10] = x10 + key[6]; out[11] = x11 + key[7];
    out[12] = x12 + w12; out[13] = x13 + w13; out[14] = x14 + w14; out[15] = x15 + w15;
}
CRAY_HD double indep_unit(uint32_t lo, uint32_t hi) {   // Uniform::new(0.0, 1.0) on one next_u64
    const uint64_t v = (((uint64_t)hi << 32) | lo) >> 12;
    return __builtin_bit_cast(double, 0x3ff0000000000000ULL | v) - 1.0;
}
// draws first .. first + 7 of the pixel sample whose generator key is `key` (draw j = words 2j, 2j + 1 of the stream)
CRAY_HD void indep_draws(const uint32_t key[8], uint32_t first, double out[8]) {
    const uint32_t word0 = 2u * first, blk = word0 >> 4, off = word0 & 15u;
    uint32_t w[32];
    chacha_block(key, blk, 0u, 0u, 0u, 6, w);
    chacha_block(key, blk + 1u, 0u, 0u, 0u, 6, w + 16);
    for (int j = 0; j < 8; j++) out[j] = indep_unit(w[off + 2 * j], w[off + 2 * j + 1]);
}

}  // namespace cray
