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
    return r / (m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15]);
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
// valid when nothing over/underflows: the caller guarantees that a and d are zero or within
// [2^-500, 2^500] in magnitude and d != 0 (div_fast_ok / scene check); otherwise it divides plainly.
// tests/test_host_and_abi.py checks q2 == a/d bit for bit on random and adversarial operands.
// -----------------------------------------------------------------------------------------
namespace cray {
CRAY_HD bool div_range_ok(double x) {  // 0 or 2^-500 <= |x| <= 2^500
    const double ax = fabs(x);
    return x == 0.0 || (ax >= 0x1p-500 && ax <= 0x1p500);
}
CRAY_HD bool div_fast_ok(double d) { return d != 0.0 && div_range_ok(d); }
CRAY_HD double div_fast(double a, double d, double y) {
    const double q0 = a * y;
    const double q1 = fma(fma(-q0, d, a), y, q0);
    return fma(fma(-q1, d, a), y, q1);
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

CRAY_HD void sincos_cr(double x, double& s_out, double& c_out) {
    // pi/2 = P1 + P2 + P3 + P4; P1..P3 carry 33 significant bits, so k * Pi is exact for |k| < 2^20
    const double P1 = 0x1.921fb54400000p+0, P2 = 0x1.0b4611a600000p-34, P3 = 0x1.3198a2e000000p-69, P4 = 0x1.b839a252049c1p-104;
    const double k = rint(x * 0x1.45f306dc9c883p-1);  // nearest multiple of pi/2
    const double t = x - k * P1;    // exact: Sterbenz for k != 0
    dd r = dd_two_sum(t, -(k * P2));
    r = dd_add(r, dd{-(k * P3), 0.0});
    r = dd_add(r, dd{-(k * P4), 0.0});
    const dd r2 = dd_mul(r, r);
    // (-1)^j / (2j+1)!  and  (-1)^j / (2j)!  as double-doubles
    const double S[14][2] = {
        {0x1.0000000000000p+0, 0x0.0p+0},
        {-0x1.5555555555555p-3, -0x1.5555555555555p-57},
        {0x1.1111111111111p-7, 0x1.1111111111111p-63},
        {-0x1.a01a01a01a01ap-13, -0x1.a01a01a01a01ap-73},
        {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73},
        {-0x1.ae64567f544e4p-26, 0x1.c062e06d1f209p-80},
        {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
        {-0x1.ae7f3e733b81fp-41, -0x1.1d8656b0ee8cbp-97},
        {0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103},
        {-0x1.2f49b46814157p-57, -0x1.2650f61dbdcb4p-112},
        {0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120},
        {-0x1.761b41316381ap-75, 0x1.3423c7d91404fp-130},
        {0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139},
        {-0x1.d1ab1c2dccea3p-94, -0x1.054d0c78aea14p-149}};
    const double C[15][2] = {
        {0x1.0000000000000p+0, 0x0.0p+0},
        {-0x1.0000000000000p-1, 0x0.0p+0},
        {0x1.5555555555555p-5, 0x1.5555555555555p-59},
        {-0x1.6c16c16c16c17p-10, 0x1.f49f49f49f49fp-65},
        {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76},
        {-0x1.27e4fb7789f5cp-22, -0x1.cbbc05b4fa99ap-76},
        {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
        {-0x1.93974a8c07c9dp-37, -0x1.05d6f8a2efd1fp-92},
        {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101},
        {-0x1.6827863b97d97p-53, -0x1.eec01221a8b0bp-107},
        {0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120},
        {-0x1.0ce396db7f853p-70, 0x1.aebcdbd20331cp-124},
        {0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135},
        {-0x1.88e85fc6a4e5ap-89, 0x1.71c37ebd16540p-143},
        {0x1.0a18a2635085dp-98, 0x1.b9e2e28e1aa54p-153}};
    dd ps = dd{S[13][0], S[13][1]};
    for (int j = 12; j >= 0; j--) ps = dd_add(dd_mul(ps, r2), dd{S[j][0], S[j][1]});
    dd pc = dd{C[14][0], C[14][1]};
    for (int j = 13; j >= 0; j--) pc = dd_add(dd_mul(pc, r2), dd{C[j][0], C[j][1]});
    const dd sr = dd_mul(ps, r);
    const double sv = sr.hi + sr.lo, cv = pc.hi + pc.lo;
    const int q = ((int)k) & 3;
    s_out = q == 0 ? sv : (q == 1 ? cv : (q == 2 ? -sv : -cv));
    c_out = q == 0 ? cv : (q == 1 ? -sv : (q == 2 ? -cv : sv));
}
}  // namespace cray
