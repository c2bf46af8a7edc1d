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
