/*
 * orc_math.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * f64 restatement of craytracer's math layer.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference).  Built
 * with -ffp-contract=off: rustc does not contract a*b+c into FMA, so neither
 * may we.
 *
 * Parity status: pinned by the reference's own known-answer tests
 * tests/test_transformation.rs, tests/test_bounds.rs, tests/test_color.rs
 * (re-expressed in tests/test_oracle_reference_vectors.py).
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <cmath>
#include <cstdint>
#include <limits>

namespace orc {

static const double EPSILON = 1e-9;                 /* src/constants.rs:1 */
static const double PI = 3.14159265358979323846264338327950288;        /* std::f64::consts::PI */
static const double FRAC_1_PI = 0.318309886183790671537767526745028724; /* consts::FRAC_1_PI */
static const double FRAC_PI_2 = 1.57079632679489661923132169163975144;
static const double FRAC_PI_4 = 0.785398163397448309615660845819875721;
static const double INF = std::numeric_limits<double>::infinity();

/* Rust f64::min/max return the non-NaN operand == C fmin/fmax (Appendix A.3) */
static inline double rmin(double a, double b) { return std::fmin(a, b); }
static inline double rmax(double a, double b) { return std::fmax(a, b); }
/* x.powf(2.0): LLVM folds pow(x,2.0) to x*x; restated as x*x (DESIGN.md "libm") */
static inline double sq(double x) { return x * x; }
/* x.powf(0.5): LLVM's replacePowWithSqrt turns the llvm.pow intrinsic into
 * (x == -inf ? +inf : fabs(sqrt(x))) without fast-math; restated that way. */
static inline double pow_half(double x) { return x == -INF ? INF : std::fabs(std::sqrt(x)); }
/* f64::to_radians: self * (PI / 180.0) */
static inline double to_radians(double deg) { return deg * (PI / 180.0); }
/* f64::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN */
static inline double signum(double x) { return std::isnan(x) ? x : std::copysign(1.0, x); }
/* Rust `as usize` / `as u32` on f64: saturating, NaN -> 0 */
static inline uint64_t sat_u64(double x) {
    if (!(x > 0.0)) return 0;                      /* NaN, negatives, zero */
    if (x >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)x;
}
static inline uint32_t sat_u32(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 4294967296.0) return UINT32_MAX;
    return (uint32_t)x;
}

/* ---- Vector / Point / Normal (src/geometry.rs:19-589) ------------------- */
struct V3 {
    double x, y, z;
    double operator[](int a) const { return a == 0 ? x : (a == 1 ? y : z); }
    double& at(int a) { return a == 0 ? x : (a == 1 ? y : z); }
};
static inline V3 v3(double x, double y, double z) { V3 r = {x, y, z}; return r; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   /* :88-94 */
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :104-110 */
static inline V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }     /* :120-125 */
static inline V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }     /* :135-140 */
static inline V3 neg(V3 a) { return a * -1.0; }                                          /* :150-156 */
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       /* :566-570 */
static inline double magnitude_squared(V3 a) { return dot(a, a); }                       /* :48-50 */
static inline double magnitude(V3 a) { return std::sqrt(magnitude_squared(a)); }         /* :51-53 */
static inline V3 normalized(V3 a) {                                                      /* :54-57 */
    double mag = magnitude(a);
    return v3(a.x / mag, a.y / mag, a.z / mag);
}
static inline V3 cross(V3 a, V3 b) {                                                     /* :58-64 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* Normal::same_hemisphere, src/geometry.rs:403-405 */
static inline bool same_hemisphere(V3 n, V3 v1, V3 v2) { return dot(n, v1) * dot(n, v2) > 0.0; }
/* Normal::generate_tangents, src/geometry.rs:406-417 */
static inline void generate_tangents(V3 n, V3* t, V3* b) {
    V3 v = normalized(n);
    double sign = signum(v.z);
    double a = -1.0 / (sign + v.z);
    double bb = v.x * v.y * a;
    *t = v3(1.0 + sign * v.x * v.x * a, sign * bb, -sign * v.x);
    *b = v3(bb, sign + v.y * v.y * a, -v.y);
}

/* ---- Color (src/color.rs) ------------------------------------------------ */
struct Col { double r, g, b; };
static inline Col col(double r, double g, double b) { Col c = {r, g, b}; return c; }
static const Col BLACK = {0.0, 0.0, 0.0};
static const Col WHITE = {1.0, 1.0, 1.0};
static inline Col operator+(Col a, Col b) { return col(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline Col operator-(Col a, Col b) { return col(a.r - b.r, a.g - b.g, a.b - b.b); }
static inline Col operator*(Col a, Col b) { return col(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline Col operator/(Col a, Col b) { return col(a.r / b.r, a.g / b.g, a.b / b.b); }
static inline Col operator*(Col a, double s) { return col(a.r * s, a.g * s, a.b * s); }
static inline Col operator/(Col a, double s) { return col(a.r / s, a.g / s, a.b / s); }
static inline bool is_black(Col c) { return c.r == 0.0 && c.g == 0.0 && c.b == 0.0; }   /* :55-57 */
static inline bool is_finite(Col c) { return std::isfinite(c.r) && std::isfinite(c.g) && std::isfinite(c.b); }
static inline Col col_pow_half(Col c) { return col(pow_half(c.r), pow_half(c.g), pow_half(c.b)); }
static const double GAMMA = 2.2;                                                        /* :13 */
/* Color::from_rgb, src/color.rs:39-46 — genuine pow() on glibc, like Rust's powf */
static inline Col from_rgb(uint8_t r, uint8_t g, uint8_t b) {
    return col(std::pow((double)r / 255.0, GAMMA), std::pow((double)g / 255.0, GAMMA),
               std::pow((double)b / 255.0, GAMMA));
}
/* Color::to_rgb, src/color.rs:47-54 (`as u8` saturates) */
static inline void to_rgb(Col c, uint8_t out[3]) {
    double v[3] = {std::pow(c.r, 1.0 / GAMMA), std::pow(c.g, 1.0 / GAMMA), std::pow(c.b, 1.0 / GAMMA)};
    for (int i = 0; i < 3; i++) {
        double x = v[i];
        /* f64::clamp(0,1): NaN stays NaN; then `as u8` -> 0 */
        if (x < 0.0) x = 0.0;
        if (x > 1.0) x = 1.0;
        double y = x * 255.0;
        out[i] = (uint8_t)(!(y > 0.0) ? 0 : (y >= 255.0 ? 255 : (int)y));
    }
}

/* ---- Ray (src/ray.rs) ------------------------------------------------------ */
struct Ray {
    V3 o, d;
    double tmax;
};
static inline Ray ray_new(V3 o, V3 d) { Ray r = {o, d, INF}; return r; }                /* :14-20 */
static inline V3 ray_at(const Ray& r, double t) { return r.o + r.d * t; }               /* :22-24 */
static inline bool contains_distance(const Ray& r, double t) { return t > EPSILON && t < r.tmax; } /* :26-28 */
static inline bool update_max_distance(Ray& r, double t) {                              /* :30-37 */
    if (contains_distance(r, t)) { r.tmax = t; return true; }
    return false;
}

/* ---- Bounds (src/bounds.rs) ------------------------------------------------ */
struct Bounds { V3 mn, mx; };
static inline Bounds bounds_new(V3 a, V3 b) {                                            /* :16-21 */
    Bounds r = {v3(rmin(a.x, b.x), rmin(a.y, b.y), rmin(a.z, b.z)),
                v3(rmax(a.x, b.x), rmax(a.y, b.y), rmax(a.z, b.z))};
    return r;
}
static inline Bounds bounds_union(Bounds a, Bounds b) {                                  /* :91-108 */
    Bounds r = {v3(rmin(a.mn.x, b.mn.x), rmin(a.mn.y, b.mn.y), rmin(a.mn.z, b.mn.z)),
                v3(rmax(a.mx.x, b.mx.x), rmax(a.mx.y, b.mx.y), rmax(a.mx.z, b.mx.z))};
    return r;
}
static inline V3 bounds_centroid(Bounds b) {                                             /* :22-28 */
    return v3((b.mn.x + b.mx.x) * 0.5, (b.mn.y + b.mx.y) * 0.5, (b.mn.z + b.mx.z) * 0.5);
}
static inline V3 bounds_diagonal(Bounds b) { return b.mx - b.mn; }                       /* :33-35 */
static inline double bounds_surface_area(Bounds b) {                                     /* :29-32 */
    V3 d = bounds_diagonal(b);
    return 2.0 * (d.x * d.y + d.y * d.z + d.z * d.x);
}
static inline int bounds_maximum_extent(Bounds b) {                                      /* :36-45 */
    V3 d = bounds_diagonal(b);
    if (d.x > d.y && d.x > d.z) return 0;
    if (d.y > d.z) return 1;
    return 2;
}
static inline bool bounds_contains(Bounds b, V3 p) {                                     /* :46-53 */
    return b.mn.x <= p.x && b.mn.y <= p.y && b.mn.z <= p.z && b.mx.x >= p.x && b.mx.y >= p.y && b.mx.z >= p.z;
}
static inline V3 bounds_offset(Bounds b, V3 p) {                                         /* :55-61 */
    return v3((p.x - b.mn.x) / (b.mx.x - b.mn.x), (p.y - b.mn.y) / (b.mx.y - b.mn.y),
              (p.z - b.mn.z) / (b.mx.z - b.mn.z));
}
/* Bounds::intersects, src/bounds.rs:62-88 */
static inline bool bounds_intersects(Bounds b, const Ray& ray) {
    double min_distance = -INF;
    double max_distance = INF;
    for (int axis = 0; axis < 3; axis++) {
        double d_i = ray.d[axis];
        double o_i = ray.o[axis];
        double min_i = b.mn[axis];
        double max_i = b.mx[axis];
        if (std::signbit(d_i)) { double t = min_i; min_i = max_i; max_i = t; }
        max_distance = rmin(max_distance, (max_i - o_i) / d_i);
        if (max_distance < EPSILON) return false;
        min_distance = rmax(min_distance, (min_i - o_i) / d_i);
        if (min_distance > max_distance) return false;
    }
    return contains_distance(ray, min_distance) || contains_distance(ray, max_distance);
}

/* ---- Matrix / Transformation (src/transformation.rs) ----------------------- */
struct Mat { double m[4][4]; };
static inline Mat mat_identity() {
    Mat r = {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
    return r;
}
static inline Mat mat_transpose(const Mat& a) {                                           /* :58-68 */
    Mat r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i];
    return r;
}
static inline Mat mat_mul(const Mat& a, const Mat& b) {                                   /* :202-218 */
    Mat r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
/* Matrix::inverse (adjugate / Cramer), src/transformation.rs:71-195.  The term
 * order of every cofactor is kept because the camera matrices must come out
 * bit-identical.  t3(a,b,c) multiplies left to right. */
static inline bool mat_inverse(const Mat& a, Mat* out) {
    const double (*m)[4] = a.m;
#define T3(a_, b_, c_) ((a_) * (b_) * (c_))
    double inv[4][4];
    inv[0][0] = T3(m[1][1], m[2][2], m[3][3]) - T3(m[1][1], m[2][3], m[3][2]) - T3(m[2][1], m[1][2], m[3][3]) +
                T3(m[2][1], m[1][3], m[3][2]) + T3(m[3][1], m[1][2], m[2][3]) - T3(m[3][1], m[1][3], m[2][2]);
    inv[0][1] = T3(-m[0][1], m[2][2], m[3][3]) + T3(m[0][1], m[2][3], m[3][2]) + T3(m[2][1], m[0][2], m[3][3]) -
                T3(m[2][1], m[0][3], m[3][2]) - T3(m[3][1], m[0][2], m[2][3]) + T3(m[3][1], m[0][3], m[2][2]);
    inv[0][2] = T3(m[0][1], m[1][2], m[3][3]) - T3(m[0][1], m[1][3], m[3][2]) - T3(m[1][1], m[0][2], m[3][3]) +
                T3(m[1][1], m[0][3], m[3][2]) + T3(m[3][1], m[0][2], m[1][3]) - T3(m[3][1], m[0][3], m[1][2]);
    inv[0][3] = T3(-m[0][1], m[1][2], m[2][3]) + T3(m[0][1], m[1][3], m[2][2]) + T3(m[1][1], m[0][2], m[2][3]) -
                T3(m[1][1], m[0][3], m[2][2]) - T3(m[2][1], m[0][2], m[1][3]) + T3(m[2][1], m[0][3], m[1][2]);
    inv[1][0] = T3(-m[1][0], m[2][2], m[3][3]) + T3(m[1][0], m[2][3], m[3][2]) + T3(m[2][0], m[1][2], m[3][3]) -
                T3(m[2][0], m[1][3], m[3][2]) - T3(m[3][0], m[1][2], m[2][3]) + T3(m[3][0], m[1][3], m[2][2]);
    inv[1][1] = T3(m[0][0], m[2][2], m[3][3]) - T3(m[0][0], m[2][3], m[3][2]) - T3(m[2][0], m[0][2], m[3][3]) +
                T3(m[2][0], m[0][3], m[3][2]) + T3(m[3][0], m[0][2], m[2][3]) - T3(m[3][0], m[0][3], m[2][2]);
    inv[1][2] = T3(-m[0][0], m[1][2], m[3][3]) + T3(m[0][0], m[1][3], m[3][2]) + T3(m[1][0], m[0][2], m[3][3]) -
                T3(m[1][0], m[0][3], m[3][2]) - T3(m[3][0], m[0][2], m[1][3]) + T3(m[3][0], m[0][3], m[1][2]);
    inv[1][3] = T3(m[0][0], m[1][2], m[2][3]) - T3(m[0][0], m[1][3], m[2][2]) - T3(m[1][0], m[0][2], m[2][3]) +
                T3(m[1][0], m[0][3], m[2][2]) + T3(m[2][0], m[0][2], m[1][3]) - T3(m[2][0], m[0][3], m[1][2]);
    inv[2][0] = T3(m[1][0], m[2][1], m[3][3]) - T3(m[1][0], m[2][3], m[3][1]) - T3(m[2][0], m[1][1], m[3][3]) +
                T3(m[2][0], m[1][3], m[3][1]) + T3(m[3][0], m[1][1], m[2][3]) - T3(m[3][0], m[1][3], m[2][1]);
    inv[2][1] = T3(-m[0][0], m[2][1], m[3][3]) + T3(m[0][0], m[2][3], m[3][1]) + T3(m[2][0], m[0][1], m[3][3]) -
                T3(m[2][0], m[0][3], m[3][1]) - T3(m[3][0], m[0][1], m[2][3]) + T3(m[3][0], m[0][3], m[2][1]);
    inv[2][2] = T3(m[0][0], m[1][1], m[3][3]) - T3(m[0][0], m[1][3], m[3][1]) - T3(m[1][0], m[0][1], m[3][3]) +
                T3(m[1][0], m[0][3], m[3][1]) + T3(m[3][0], m[0][1], m[1][3]) - T3(m[3][0], m[0][3], m[1][1]);
    inv[2][3] = T3(-m[0][0], m[1][1], m[2][3]) + T3(m[0][0], m[1][3], m[2][1]) + T3(m[1][0], m[0][1], m[2][3]) -
                T3(m[1][0], m[0][3], m[2][1]) - T3(m[2][0], m[0][1], m[1][3]) + T3(m[2][0], m[0][3], m[1][1]);
    inv[3][0] = T3(-m[1][0], m[2][1], m[3][2]) + T3(m[1][0], m[2][2], m[3][1]) + T3(m[2][0], m[1][1], m[3][2]) -
                T3(m[2][0], m[1][2], m[3][1]) - T3(m[3][0], m[1][1], m[2][2]) + T3(m[3][0], m[1][2], m[2][1]);
    inv[3][1] = T3(m[0][0], m[2][1], m[3][2]) - T3(m[0][0], m[2][2], m[3][1]) - T3(m[2][0], m[0][1], m[3][2]) +
                T3(m[2][0], m[0][2], m[3][1]) + T3(m[3][0], m[0][1], m[2][2]) - T3(m[3][0], m[0][2], m[2][1]);
    inv[3][2] = T3(-m[0][0], m[1][1], m[3][2]) + T3(m[0][0], m[1][2], m[3][1]) + T3(m[1][0], m[0][1], m[3][2]) -
                T3(m[1][0], m[0][2], m[3][1]) - T3(m[3][0], m[0][1], m[1][2]) + T3(m[3][0], m[0][2], m[1][1]);
    inv[3][3] = T3(m[0][0], m[1][1], m[2][2]) - T3(m[0][0], m[1][2], m[2][1]) - T3(m[1][0], m[0][1], m[2][2]) +
                T3(m[1][0], m[0][2], m[2][1]) + T3(m[2][0], m[0][1], m[1][2]) - T3(m[2][0], m[0][2], m[1][1]);
#undef T3
    double det = m[0][0] * inv[0][0] + m[0][1] * inv[1][0] + m[0][2] * inv[2][0] + m[0][3] * inv[3][0];
    if (det != 0.0) {
        double inv_det = 1.0 / det;
        for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) out->m[i][j] = inv[i][j] * inv_det;
        return true;
    }
    return false;
}

struct Xf { Mat matrix, inverse; };
static inline Xf xf_inverse(const Xf& t) { Xf r = {t.inverse, t.matrix}; return r; }     /* :262-267 */
static inline Xf xf_mul(const Xf& a, const Xf& b) {                                       /* :392-412 */
    Xf r = {mat_mul(a.matrix, b.matrix), mat_mul(b.inverse, a.inverse)};
    return r;
}
static inline Xf xf_translate(double dx, double dy, double dz) {                          /* :269-288 */
    Xf r = {mat_identity(), mat_identity()};
    r.matrix.m[0][3] = dx; r.matrix.m[1][3] = dy; r.matrix.m[2][3] = dz;
    r.inverse.m[0][3] = -dx; r.inverse.m[1][3] = -dy; r.inverse.m[2][3] = -dz;
    return r;
}
static inline Xf xf_scale(double x, double y, double z) {                                 /* :290-309 */
    Xf r = {mat_identity(), mat_identity()};
    r.matrix.m[0][0] = x; r.matrix.m[1][1] = y; r.matrix.m[2][2] = z;
    r.inverse.m[0][0] = 1.0 / x; r.inverse.m[1][1] = 1.0 / y; r.inverse.m[2][2] = 1.0 / z;
    return r;
}
static inline Xf xf_rotate_x(double radians) {                                            /* :311-324 */
    double s = std::sin(radians), c = std::cos(radians);
    Xf r; r.matrix = mat_identity();
    r.matrix.m[1][1] = c; r.matrix.m[1][2] = -s; r.matrix.m[2][1] = s; r.matrix.m[2][2] = c;
    r.inverse = mat_transpose(r.matrix);
    return r;
}
static inline Xf xf_rotate_y(double radians) {                                            /* :326-339 */
    double s = std::sin(radians), c = std::cos(radians);
    Xf r; r.matrix = mat_identity();
    r.matrix.m[0][0] = c; r.matrix.m[0][2] = s; r.matrix.m[2][0] = -s; r.matrix.m[2][2] = c;
    r.inverse = mat_transpose(r.matrix);
    return r;
}
static inline Xf xf_rotate_z(double radians) {                                            /* :341-354 */
    double s = std::sin(radians), c = std::cos(radians);
    Xf r; r.matrix = mat_identity();
    r.matrix.m[0][0] = c; r.matrix.m[0][1] = -s; r.matrix.m[1][0] = s; r.matrix.m[1][1] = c;
    r.inverse = mat_transpose(r.matrix);
    return r;
}
static inline Xf xf_look_at(V3 origin, V3 target, V3 up) {                                /* :356-370 */
    V3 z = normalized(target - origin);
    V3 x = normalized(cross(normalized(up), z));
    V3 y = normalized(cross(z, x));
    Xf r;
    Mat mm = {{{x.x, y.x, z.x, origin.x}, {x.y, y.y, z.y, origin.y}, {x.z, y.z, z.z, origin.z}, {0, 0, 0, 1}}};
    r.matrix = mm;
    mat_inverse(r.matrix, &r.inverse);
    return r;
}
static inline Xf xf_perspective(double fov, double near, double far) {                    /* :372-385 */
    Xf persp;
    Mat mm = {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, far / (far - near), -far * near / (far - near)}, {0, 0, 1, 0}}};
    persp.matrix = mm;
    mat_inverse(persp.matrix, &persp.inverse);
    double inv_tan_ang = 1.0 / std::tan(to_radians(fov) * 0.5);
    return xf_mul(persp, xf_scale(inv_tan_ang, inv_tan_ang, 1.0));
}
static inline Xf xf_orthographic(double near, double far) {                               /* :387-389 */
    return xf_mul(xf_scale(1.0, 1.0, 1.0 / (far - near)), xf_translate(0.0, 0.0, -near));
}
static inline V3 xf_point(const Xf& t, V3 p) {                                            /* :418-428 */
    const double (*m)[4] = t.matrix.m;
    V3 r = v3(m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3],
              m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3],
              m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3]);
    return r / (m[3][0] * p.x + m[3][1] * p.y + m[3][2] * p.z + m[3][3]);
}
static inline V3 xf_vector(const Xf& t, V3 v) {                                           /* :431-440 */
    const double (*m)[4] = t.matrix.m;
    return v3(m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
              m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z);
}
static inline V3 xf_normal(const Xf& t, V3 n) {                                           /* :443-453 */
    const double (*inv)[4] = t.inverse.m;
    return v3(inv[0][0] * n.x + inv[1][0] * n.y + inv[2][0] * n.z, inv[0][1] * n.x + inv[1][1] * n.y + inv[2][1] * n.z,
              inv[0][2] * n.x + inv[1][2] * n.y + inv[2][2] * n.z);
}
static inline Ray xf_ray(const Xf& t, const Ray& r) {                                     /* :456-462 */
    Ray out = ray_new(xf_point(t, r.o), xf_vector(t, r.d));
    update_max_distance(out, r.tmax);
    return out;
}
static inline Bounds xf_bounds(const Xf& t, Bounds b) {                                   /* :465-481 */
    V3 c[8] = {v3(b.mn.x, b.mn.y, b.mn.z), v3(b.mn.x, b.mn.y, b.mx.z), v3(b.mn.x, b.mx.y, b.mn.z),
               v3(b.mn.x, b.mx.y, b.mx.z), v3(b.mx.x, b.mn.y, b.mn.z), v3(b.mx.x, b.mn.y, b.mx.z),
               v3(b.mx.x, b.mx.y, b.mn.z), v3(b.mx.x, b.mx.y, b.mx.z)};
    V3 p0 = xf_point(t, c[0]);
    Bounds acc = bounds_new(p0, p0);
    for (int i = 1; i < 8; i++) {
        V3 p = xf_point(t, c[i]);
        acc = bounds_union(acc, bounds_new(p, p));
    }
    return acc;
}

} /* namespace orc */
#endif
