// cray_cull_check.h — the host-side check of tri_cull32 (cray_math.h) against the literal f64 Moller-Trumbore, as a header so that
// tests/test_tri_cull.py can also compile it with a deliberately wrong error constant (-DCRAY_CULL_K_UNITS=0.5f) and see the
// check FAIL: a test that cannot fail proves nothing.  cray_host.cpp wraps it as cray_host_tri_cull_violations.
#pragma once
#include <cstdint>

#include "cray_math.h"

namespace cray {
// Certified f32 culling of the triangle test (cray_math.h tri_cull32) against the literal f64 test (shape.rs:216-262 as the
// kernels spell it: cray_shading.h tri_test) on n triangles / rays / tmax: a certified MISS where the reference hits, or a
// certified HIT where it misses, is a violation.  counts[4]: unknown, certified misses, certified hits, rays outside the range.
inline uint64_t tri_cull_violations(const double* v0, const double* e1, const double* e2, const double* o, const double* d, const double* tmax,
                                                  uint64_t n, uint64_t* counts) {
    uint64_t bad = 0, cnt[4] = {0, 0, 0, 0};
    for (uint64_t i = 0; i < n; i++) {
        const double *pv = v0 + 3 * i, *p1 = e1 + 3 * i, *p2 = e2 + 3 * i, *po = o + 3 * i, *pd = d + 3 * i;
        bool scene_ok = true, fast = true;
        for (int k = 0; k < 3; k++) {
            // what hyb_scene_ok guarantees for every vertex of an uploaded scene (edges are differences of two vertices)
            scene_ok = scene_ok && fabs(pv[k]) <= 0x1p40 && fabs(pv[k] + p1[k]) <= 0x1p40 && fabs(pv[k] + p2[k]) <= 0x1p40;
            fast = fast && div_fast_ok(pd[k]) && div_range_ok(po[k]);
        }
        if (!scene_ok) continue;
        const vec3 ov = mk(po[0], po[1], po[2]), dv = mk(pd[0], pd[1], pd[2]);
        const vec3 rd = mk(1.0 / pd[0], 1.0 / pd[1], 1.0 / pd[2]);
        const HybRay hr = hyb_ray(ov, dv, rd, fast);
        const bool ray_ok = hr.a == hr.a;
        if (!ray_ok) cnt[3]++;
        float t_lo, t_hi;
        hyb_tmax(tmax[i], t_lo, t_hi);
        const float e1m = f32_up(fmax(fmax(fabs(p1[0]), fabs(p1[1])), fabs(p1[2]))), e2m = f32_up(fmax(fmax(fabs(p2[0]), fabs(p2[1])), fabs(p2[2])));
        const TriCull tc = tri_cull32<true>(pv[0], pv[1], pv[2], (float)p1[0], (float)p1[1], (float)p1[2], (float)p2[0], (float)p2[1], (float)p2[2], e1m, e2m,
                                            po[0], po[1], po[2], (float)pd[0], (float)pd[1], (float)pd[2], t_lo, t_hi, ray_ok);
        // the literal test
        ray_t ray = mkray(ov, dv);
        ray.tmax = tmax[i];
        const vec3 V0 = mk(pv[0], pv[1], pv[2]), E1 = mk(p1[0], p1[1], p1[2]), E2 = mk(p2[0], p2[1], p2[2]);
        bool hit = false;
        {
            const vec3 P = cross(ray.d, E2);
            const double denom = dot(P, E1);
            if (!(denom > -kEps && denom < kEps)) {
                const vec3 T = ray.o - V0;
                const double u = dot(P, T) / denom;
                if (!(u < 0.0 || u > 1.0)) {
                    const vec3 Q = cross(T, E1);
                    const double v = dot(Q, ray.d) / denom;
                    if (!(v < 0.0 || u + v > 1.0)) {
                        const double t = dot(cross(T, E1), E2) / denom;
                        hit = in_range(ray, t);
                    }
                }
            }
        }
        if (tc.miss && tc.hit) bad++;
        if (tc.miss && hit) bad++;
        if (tc.hit && !hit) bad++;
        cnt[tc.miss ? 1 : (tc.hit ? 2 : 0)]++;
    }
    if (counts) for (int k = 0; k < 4; k++) counts[k] = cnt[k];
    return bad;
}

}  // namespace cray
