// cray_host.cpp — host-side mirror of craytracer's `Scene::new` (include/cray_host.h).
//
// Builds, on the CPU and untimed by the render metric:
//   * Sphere/Disk transformation pairs and all primitive bounds   (src/shape.rs:55-69,133-153,402-438)
//   * the reference-topology BVH, so that traversal order, leaf order and equal-t
//     tie-breaks on the GPU are those of the reference              (src/bvh.rs:38-56,191-336)
//   * camera matrices                                               (src/camera.rs:25-76)
//   * the power-proportional light CDF                              (src/light.rs:170-200)
//   * first-equal light indices                                     (src/path_integrator.rs:116)
// and exposes them as a cray_flat_scene for cray_scene_upload.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/cray_host.h"
#include "cray_math.h"
#include "cray_cull_check.h"

namespace cray {
void set_last_error(const char* fmt, ...);  // cray_hip.hip
}

namespace {
using namespace cray;

// ---------------------------------------------------------------- matrices
mat4 identity() {
    mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = i == j ? 1.0 : 0.0;
    return r;
}
mat4 mul(const mat4& a, const mat4& b) {  // transformation.rs:202-218: acc starts at 0.0, k ascending
    mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += a.m[i][k] * b.m[k][j];
            r.m[i][j] = acc;
        }
    return r;
}
mat4 transpose(const mat4& a) {
    mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i];
    return r;
}
// Matrix::inverse (transformation.rs:71-195): adjugate over determinant.  Entry (i,j) is the
// signed 3x3 minor with row j and column i removed, expanded along the first remaining column,
// each product evaluated left to right and the six products summed in expansion order —
// which is the order the reference writes them in.
bool inverse(const mat4& a, mat4& out) {
    double adj[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            int R[3], K[3], nr = 0, nk = 0;
            for (int t = 0; t < 4; t++) {
                if (t != j) R[nr++] = t;
                if (t != i) K[nk++] = t;
            }
            double sgn = ((i + j) & 1) ? -1.0 : 1.0;
            double acc = 0.0;
            bool first = true;
            for (int p = 0; p < 3; p++) {
                int r = R[p], ra = R[p == 0 ? 1 : 0], rb = R[p == 2 ? 1 : 2];
                double s = (p & 1) ? -sgn : sgn;
                double head_pos = a.m[r][K[0]], head_neg = -a.m[r][K[0]];
                // + head * m[ra][K1] * m[rb][K2]
                double t1 = (s > 0 ? head_pos : head_neg) * a.m[ra][K[1]] * a.m[rb][K[2]];
                // - head * m[ra][K2] * m[rb][K1]
                double t2 = (s > 0 ? head_pos : head_neg) * a.m[ra][K[2]] * a.m[rb][K[1]];
                if (first) { acc = t1; first = false; } else acc = acc + t1;
                acc = acc - t2;
            }
            adj[i][j] = acc;
        }
    double det = a.m[0][0] * adj[0][0] + a.m[0][1] * adj[1][0] + a.m[0][2] * adj[2][0] + a.m[0][3] * adj[3][0];
    if (det == 0.0) return false;
    double inv_det = 1.0 / det;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out.m[i][j] = adj[i][j] * inv_det;
    return true;
}

xform compose(const xform& a, const xform& b) { return xform{mul(a.fwd, b.fwd), mul(b.inv, a.inv)}; }  // :392-412
xform swapped(const xform& t) { return xform{t.inv, t.fwd}; }                                           // :262-267
xform translate(double dx, double dy, double dz) {                                                      // :269-288
    xform t{identity(), identity()};
    t.fwd.m[0][3] = dx; t.fwd.m[1][3] = dy; t.fwd.m[2][3] = dz;
    t.inv.m[0][3] = -dx; t.inv.m[1][3] = -dy; t.inv.m[2][3] = -dz;
    return t;
}
xform scale(double x, double y, double z) {                                                             // :290-309
    xform t{identity(), identity()};
    t.fwd.m[0][0] = x; t.fwd.m[1][1] = y; t.fwd.m[2][2] = z;
    t.inv.m[0][0] = 1.0 / x; t.inv.m[1][1] = 1.0 / y; t.inv.m[2][2] = 1.0 / z;
    return t;
}
xform rotate_x(double rad) {                                                                             // :311-324
    double s = sin(rad), c = cos(rad);
    xform t; t.fwd = identity();
    t.fwd.m[1][1] = c; t.fwd.m[1][2] = -s; t.fwd.m[2][1] = s; t.fwd.m[2][2] = c;
    t.inv = transpose(t.fwd);
    return t;
}
xform rotate_y(double rad) {                                                                             // :326-339
    double s = sin(rad), c = cos(rad);
    xform t; t.fwd = identity();
    t.fwd.m[0][0] = c; t.fwd.m[0][2] = s; t.fwd.m[2][0] = -s; t.fwd.m[2][2] = c;
    t.inv = transpose(t.fwd);
    return t;
}
bool look_at(vec3 origin, vec3 target, vec3 up, xform& t) {                                              // :356-370
    vec3 z = unit(target - origin);
    vec3 x = unit(cross(unit(up), z));
    vec3 y = unit(cross(z, x));
    double rows[4][4] = {{x.x, y.x, z.x, origin.x}, {x.y, y.y, z.y, origin.y}, {x.z, y.z, z.z, origin.z}, {0, 0, 0, 1}};
    memcpy(t.fwd.m, rows, sizeof(rows));
    return inverse(t.fwd, t.inv);
}
bool perspective(double fov, double near, double far, xform& out) {                                      // :372-385
    xform p;
    double rows[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, far / (far - near), -far * near / (far - near)}, {0, 0, 1, 0}};
    memcpy(p.fwd.m, rows, sizeof(rows));
    if (!inverse(p.fwd, p.inv)) return false;
    double inv_tan = 1.0 / tan(deg2rad(fov) * 0.5);
    out = compose(p, scale(inv_tan, inv_tan, 1.0));
    return true;
}
xform orthographic(double near, double far) {                                                            // :387-389
    return compose(scale(1.0, 1.0, 1.0 / (far - near)), translate(0.0, 0.0, -near));
}

// ---------------------------------------------------------------- bounds
struct box3 {
    vec3 lo, hi;
};
box3 box_of(vec3 a, vec3 b) {  // Bounds::new
    return box3{mk(min_nn(a.x, b.x), min_nn(a.y, b.y), min_nn(a.z, b.z)), mk(max_nn(a.x, b.x), max_nn(a.y, b.y), max_nn(a.z, b.z))};
}
box3 join(const box3& a, const box3& b) {  // Add for Bounds
    return box3{mk(min_nn(a.lo.x, b.lo.x), min_nn(a.lo.y, b.lo.y), min_nn(a.lo.z, b.lo.z)),
                mk(max_nn(a.hi.x, b.hi.x), max_nn(a.hi.y, b.hi.y), max_nn(a.hi.z, b.hi.z))};
}
double area(const box3& b) {  // surface_area, bounds.rs:29-32
    vec3 d = b.hi - b.lo;
    return 2.0 * (d.x * d.y + d.y * d.z + d.z * d.x);
}
int widest(const box3& b) {  // maximum_extent, bounds.rs:36-45
    vec3 d = b.hi - b.lo;
    if (d.x > d.y && d.x > d.z) return 0;
    return d.y > d.z ? 1 : 2;
}
box3 xf_box(const mat4& m, const box3& b) {  // Transformable<Bounds>, transformation.rs:465-481
    const double* mm = &m.m[0][0];
    box3 acc{};
    int n = 0;
    for (int ix = 0; ix < 2; ix++)
        for (int iy = 0; iy < 2; iy++)
            for (int iz = 0; iz < 2; iz++) {
                vec3 p = xf_point(mm, mk(ix ? b.hi.x : b.lo.x, iy ? b.hi.y : b.lo.y, iz ? b.hi.z : b.lo.z));
                acc = n++ == 0 ? box_of(p, p) : join(acc, box_of(p, p));
            }
    return acc;
}

// ---------------------------------------------------------------- BVH build
struct Item {
    uint32_t prim;
    box3 box;
    vec3 centroid;
};

struct Builder {
    std::vector<cray_bvh_node> nodes;
    std::vector<uint32_t> refs;
    int error = 0;

    uint32_t leaf(const Item* it, size_t n, const box3& b) {
        cray_bvh_node nd;
        memset(&nd, 0, sizeof(nd));
        nd.bmin[0] = b.lo.x; nd.bmin[1] = b.lo.y; nd.bmin[2] = b.lo.z;
        nd.bmax[0] = b.hi.x; nd.bmax[1] = b.hi.y; nd.bmax[2] = b.hi.z;
        nd.first = (uint32_t)refs.size();
        nd.count = (uint32_t)n;
        nd.is_leaf = 1;
        for (size_t i = 0; i < n; i++) refs.push_back(it[i].prim);
        nodes.push_back(nd);
        return (uint32_t)nodes.size() - 1;
    }
    uint32_t interior(const box3& b, int axis) {
        cray_bvh_node nd;
        memset(&nd, 0, sizeof(nd));
        nd.bmin[0] = b.lo.x; nd.bmin[1] = b.lo.y; nd.bmin[2] = b.lo.z;
        nd.bmax[0] = b.hi.x; nd.bmax[1] = b.hi.y; nd.bmax[2] = b.hi.z;
        nd.axis = axis;
        nodes.push_back(nd);
        return (uint32_t)nodes.size() - 1;
    }

    // BvhNode::from_sah_splitting (bvh.rs:234-336)
    uint32_t sah(Item* it, size_t n) {
        constexpr int kBuckets = 12;
        constexpr double kTraversalCost = 1.0 / 8.0;
        constexpr size_t kMaxLeaf = 4;
        box3 all = it[0].box;
        for (size_t i = 1; i < n; i++) all = join(all, it[i].box);
        if (n <= 1) return leaf(it, n, all);
        double total_area = area(all);
        if (!(total_area > 0.0)) { error = 1; return leaf(it, n, all); }  // assert!, bvh.rs:245

        box3 cb = box_of(it[0].centroid, it[0].centroid);
        for (size_t i = 1; i < n; i++) cb = join(cb, box_of(it[i].centroid, it[i].centroid));
        const int axis = widest(cb);
        const double c_lo = comp(cb.lo, axis), c_hi = comp(cb.hi, axis);
        auto bucket_of = [&](const Item& p) -> int {
            double off = (comp(p.centroid, axis) - c_lo) / (c_hi - c_lo);  // Bounds::offset, bounds.rs:55-61
            uint64_t idx = to_u64_sat((double)kBuckets * off);
            return (int)(idx < (uint64_t)(kBuckets - 1) ? idx : (uint64_t)(kBuckets - 1));
        };
        bool used[kBuckets] = {};
        box3 bbox[kBuckets];
        size_t bcnt[kBuckets] = {};
        for (size_t i = 0; i < n; i++) {
            int b = bucket_of(it[i]);
            if (used[b]) { bbox[b] = join(bbox[b], it[i].box); bcnt[b]++; }
            else { used[b] = true; bbox[b] = it[i].box; bcnt[b] = 1; }
        }
        double cost[kBuckets - 1];
        for (int s = 0; s < kBuckets - 1; s++) {
            double c = kTraversalCost;
            for (int side = 0; side < 2; side++) {
                int b0 = side ? s + 1 : 0, b1 = side ? kBuckets : s + 1;
                bool any = false;
                box3 mb{};
                size_t mc = 0;
                for (int b = b0; b < b1; b++) {
                    if (!used[b]) continue;
                    if (any) { mb = join(mb, bbox[b]); mc += bcnt[b]; }
                    else { any = true; mb = bbox[b]; mc = bcnt[b]; }
                }
                if (any) c += (double)mc * area(mb) / total_area;
            }
            if (!isfinite(c)) error = 2;  // assert!(cost.is_finite()), bvh.rs:304
            cost[s] = c;
        }
        int best = 0;
        for (int s = 0; s < kBuckets - 1; s++)
            if (cost[s] < cost[best]) best = s;
        if ((double)n <= cost[best] && n <= kMaxLeaf) return leaf(it, n, all);

        // util::partition_by (util.rs:4-26) with pred = bucket <= best
        size_t l = 0, r = n - 1;
        auto pred = [&](const Item& p) { return bucket_of(p) <= best; };
        while (l != r) {
            while (l < r && pred(it[l])) l++;
            while (r > l && !pred(it[r])) r--;
            std::swap(it[l], it[r]);
        }
        size_t split = pred(it[l]) ? l + 1 : l;
        if (split == 0 || split == n) { error = 3; return leaf(it, n, all); }  // assert!, bvh.rs:327-328

        uint32_t me = interior(all, axis);
        uint32_t a = sah(it, split);
        uint32_t b = sah(it + split, n - split);
        nodes[me].left = a;
        nodes[me].right = b;
        return me;
    }

    // BvhNode::from_median_splitting (bvh.rs:191-230). Only the n <= 4 -> single leaf case is
    // pinned by the reference tests; the selection permutation of select_nth_unstable_by is
    // implementation-defined, std::nth_element is used for larger inputs.
    uint32_t median(Item* it, size_t n) {
        box3 all = it[0].box;
        for (size_t i = 1; i < n; i++) all = join(all, it[i].box);
        if (n <= 4) return leaf(it, n, all);
        box3 cb = box_of(it[0].centroid, it[0].centroid);
        for (size_t i = 1; i < n; i++) cb = join(cb, box_of(it[i].centroid, it[i].centroid));
        int axis = widest(cb);
        if (comp(cb.lo, axis) == comp(cb.hi, axis)) return leaf(it, n, all);
        size_t mid = (n - 1) / 2;
        std::nth_element(it, it + mid, it + n, [&](const Item& a, const Item& b) { return comp(a.centroid, axis) < comp(b.centroid, axis); });
        if (mid == 0) mid = 1;
        uint32_t me = interior(all, axis);
        uint32_t a = median(it, mid);
        uint32_t b = median(it + mid, n - mid);
        nodes[me].left = a;
        nodes[me].right = b;
        return me;
    }
};

void store16(const mat4& m, double* out) { memcpy(out, m.m, sizeof(double) * 16); }

}  // namespace

struct cray_host_scene {
    cray_flat_scene flat;
    std::vector<cray_bvh_node> nodes;
    std::vector<uint32_t> refs;
    std::vector<cray_xf_shape> spheres, disks;
    std::vector<double> cdf;
    std::vector<int32_t> first_equal;
    std::vector<cray_prim_bound> other_bounds;   // resident build: boxes of the non-triangle primitives
    double build_seconds;
    double bvh_seconds = 0.0;
    cray_bvh_build_stats gpu_build{};
};

extern "C" int cray_host_scene_new(const cray_scene_desc* d, int split_method, cray_host_scene** out) {
    return cray_host_scene_new_on(d, split_method, nullptr, out);
}

static int scene_new_impl(const cray_scene_desc* d, int split_method, cray_ctx* bvh_ctx, bool resident, cray_host_scene** out);

extern "C" int cray_host_scene_new_on(const cray_scene_desc* d, int split_method, cray_ctx* bvh_ctx, cray_host_scene** out) {
    return scene_new_impl(d, split_method, bvh_ctx, false, out);
}

extern "C" int cray_host_scene_new_resident(const cray_scene_desc* d, cray_host_scene** out) {
    return scene_new_impl(d, CRAY_SPLIT_SAH, nullptr, true, out);
}

static int scene_new_impl(const cray_scene_desc* d, int split_method, cray_ctx* bvh_ctx, bool resident, cray_host_scene** out) {
    using namespace cray;
    if (!d || !out) { set_last_error("cray_host_scene_new: null argument"); return CRAY_ERR_INVALID; }
    if (bvh_ctx && split_method != CRAY_SPLIT_SAH) { set_last_error("the GPU builder implements SplitMethod::SAH only"); return CRAY_ERR_UNSUPPORTED; }
    *out = nullptr;
    if (d->n_lights == 0) { set_last_error("No lights in the scene."); return CRAY_ERR_INVALID; }
    if (d->n_prims == 0) { set_last_error("Bvh::new: no primitives"); return CRAY_ERR_BUILD; }
    auto t0 = std::chrono::steady_clock::now();
    cray_host_scene* hs = new cray_host_scene();

    // --- Sphere / Disk transformation pairs (shape.rs:55-69, 133-153)
    hs->spheres.resize(d->n_spheres);
    for (uint32_t i = 0; i < d->n_spheres; i++) {
        const cray_sphere_desc& s = d->spheres[i];
        xform t = translate(s.origin.x, s.origin.y, s.origin.z);
        store16(t.fwd, hs->spheres[i].m);
        store16(t.inv, hs->spheres[i].inv);
        hs->spheres[i].radius = s.radius;
        hs->spheres[i].inner_radius = 0.0;
    }
    hs->disks.resize(d->n_disks);
    for (uint32_t i = 0; i < d->n_disks; i++) {
        const cray_disk_desc& s = d->disks[i];
        xform t = compose(compose(translate(s.origin.x, s.origin.y, s.origin.z), rotate_x(deg2rad(s.rotate_x))),
                          rotate_y(deg2rad(s.rotate_y)));
        store16(t.fwd, hs->disks[i].m);
        store16(t.inv, hs->disks[i].inv);
        hs->disks[i].radius = s.radius;
        hs->disks[i].inner_radius = s.inner_radius;
    }

    // --- primitive bounds (shape.rs:402-438) and Bvh::new (bvh.rs:38-56)
    // Resident build: the tree is built by cray_scene_upload on the GPU, triangle bounds included; the host only keeps the
    // boxes of the other shapes (their transformations live here) and — when a Distant / Infinite light needs the world
    // radius for its power (light.rs:170-177, scene.rs:42) — the union of all boxes.
    bool need_world = !resident;
    for (uint32_t i = 0; i < d->n_lights; i++)
        if (d->lights[i].kind == CRAY_LIGHT_DISTANT || d->lights[i].kind == CRAY_LIGHT_INFINITE) need_world = true;
    std::vector<Item> items(resident ? 0 : d->n_prims);
    box3 world;
    bool world_set = false;
    for (uint32_t i = 0; i < d->n_prims; i++) {
        const cray_prim& p = d->prims[i];
        box3 b;
        bool ok = true;
        if (p.shape_kind == CRAY_SHAPE_TRIANGLE) {
            ok = p.shape < d->n_triangles;
            if (ok && need_world) {
                const cray_triangle& t = d->triangles[p.shape];
                vec3 v0 = mk(t.v0.x, t.v0.y, t.v0.z);
                vec3 v1 = v0 + mk(t.e1.x, t.e1.y, t.e1.z), v2 = v0 + mk(t.e2.x, t.e2.y, t.e2.z);
                b = box_of(mk(min_nn(v1.x, min_nn(v2.x, v0.x)), min_nn(v1.y, min_nn(v2.y, v0.y)), min_nn(v1.z, min_nn(v2.z, v0.z))),
                           mk(max_nn(v1.x, max_nn(v2.x, v0.x)), max_nn(v1.y, max_nn(v2.y, v0.y)), max_nn(v1.z, max_nn(v2.z, v0.z))));
            }
        } else if (p.shape_kind == CRAY_SHAPE_SPHERE) {
            ok = p.shape < d->n_spheres;
            if (ok) {
                double r = hs->spheres[p.shape].radius;
                mat4 m; memcpy(m.m, hs->spheres[p.shape].m, sizeof(m.m));
                b = xf_box(m, box_of(mk(-r, -r, -r), mk(r, r, r)));
            }
        } else if (p.shape_kind == CRAY_SHAPE_DISK) {
            ok = p.shape < d->n_disks;
            if (ok) {
                double r = hs->disks[p.shape].radius;
                mat4 m; memcpy(m.m, hs->disks[p.shape].m, sizeof(m.m));
                b = xf_box(m, box_of(mk(-r, -r, 0.0), mk(r, r, 0.0)));
            }
        } else ok = false;
        if (!ok || (p.material >= (int32_t)d->n_materials) || (p.light >= (int32_t)d->n_lights) ||
            (p.material < 0 && p.light < 0)) {
            set_last_error("cray_host_scene_new: primitive %u has a bad shape/material/light index", i);
            delete hs;
            return CRAY_ERR_INVALID;
        }
        if (need_world) { world = world_set ? join(world, b) : b; world_set = true; }
        if (resident) {
            if (p.shape_kind != CRAY_SHAPE_TRIANGLE) {
                cray_prim_bound ob;
                ob.prim = i; ob.pad_ = 0;
                ob.bmin[0] = b.lo.x; ob.bmin[1] = b.lo.y; ob.bmin[2] = b.lo.z; ob.bmax[0] = b.hi.x; ob.bmax[1] = b.hi.y; ob.bmax[2] = b.hi.z;
                hs->other_bounds.push_back(ob);
            }
            continue;
        }
        items[i].prim = i;
        items[i].box = b;
        items[i].centroid = mk((b.lo.x + b.hi.x) * 0.5, (b.lo.y + b.hi.y) * 0.5, (b.lo.z + b.hi.z) * 0.5);
    }
    auto t_bvh = std::chrono::steady_clock::now();
    if (resident) {
        // nothing here: cray_scene_upload builds the tree (flat.build_on_device)
    } else if (bvh_ctx) {
        // Bvh::new on the GPU (cray_bvh_build_sah, cray.h): same tree as Builder::sah below
        std::vector<double> pb((size_t)d->n_prims * 6);
        for (uint32_t i = 0; i < d->n_prims; i++) {
            const box3& b = items[i].box;
            double* o = &pb[(size_t)i * 6];
            o[0] = b.lo.x; o[1] = b.lo.y; o[2] = b.lo.z; o[3] = b.hi.x; o[4] = b.hi.y; o[5] = b.hi.z;
        }
        hs->nodes.resize((size_t)d->n_prims * 2 - 1);
        hs->refs.resize(d->n_prims);
        uint32_t n_nodes = 0;
        int rc = cray_bvh_build_sah(bvh_ctx, pb.data(), d->n_prims, hs->nodes.data(), (uint32_t)hs->nodes.size(), &n_nodes,
                                    hs->refs.data(), &hs->gpu_build);
        if (rc != CRAY_OK) { delete hs; return rc; }
        hs->nodes.resize(n_nodes);
    } else {
        Builder bld;
        bld.nodes.reserve((size_t)d->n_prims * 2);
        bld.refs.reserve(d->n_prims);
        if (split_method == CRAY_SPLIT_MEDIAN) bld.median(items.data(), items.size());
        else bld.sah(items.data(), items.size());
        if (bld.error) {
            set_last_error("Bvh::new would panic in the reference (code %d: 1 zero surface area, 2 non-finite cost, 3 empty partition)", bld.error);
            delete hs;
            return CRAY_ERR_BUILD;
        }
        hs->nodes.swap(bld.nodes);
        hs->refs.swap(bld.refs);
    }
    hs->bvh_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_bvh).count();

    // --- LightSampler::new (light.rs:187-200) with Light::power (:170-177), world_radius (scene.rs:42)
    double world_radius = need_world ? len(world.hi - world.lo) * 0.5 : 0.0;   // only read for Distant / Infinite lights
    hs->cdf.resize(d->n_lights);
    double total_power = 0.0;
    for (uint32_t i = 0; i < d->n_lights; i++) {
        const cray_light& l = d->lights[i];
        rgb c = mkc(l.c.r, l.c.g, l.c.b), power;
        if (l.kind == CRAY_LIGHT_POINT) power = c * 4.0 * kPi;
        else if (l.kind == CRAY_LIGHT_DISTANT || l.kind == CRAY_LIGHT_INFINITE) power = c * kPi * world_radius * world_radius;
        else {
            if (l.prim < 0 || (uint32_t)l.prim >= d->n_prims) { set_last_error("area light %u: bad prim index", i); delete hs; return CRAY_ERR_INVALID; }
            const cray_prim& p = d->prims[l.prim];
            double a;  // Shape::area (shape.rs:504-514); sphere is PI r^2 as in the reference
            if (p.shape_kind == CRAY_SHAPE_SPHERE) a = kPi * square(hs->spheres[p.shape].radius);
            else if (p.shape_kind == CRAY_SHAPE_DISK) a = kPi * (square(hs->disks[p.shape].radius) - square(hs->disks[p.shape].inner_radius));
            else {
                const cray_triangle& t = d->triangles[p.shape];
                a = len(cross(mk(t.e1.x, t.e1.y, t.e1.z), mk(t.e2.x, t.e2.y, t.e2.z))) / 2.0;
            }
            power = c * kPi * a;
        }
        double avg = (power.r + power.g + power.b) / 3.0;
        total_power += avg;
        hs->cdf[i] = total_power;
    }
    for (auto& v : hs->cdf) v = v / total_power;

    // --- lights.iter().position(|l| l == light) (path_integrator.rs:116): first light equal by value
    hs->first_equal.resize(d->n_lights);
    {
        std::unordered_map<std::string, int32_t> seen;
        seen.reserve(d->n_lights * 2);
        std::vector<double> key;
        for (uint32_t i = 0; i < d->n_lights; i++) {
            const cray_light& l = d->lights[i];
            key.clear();
            key.push_back((double)l.kind);
            key.push_back(l.c.r); key.push_back(l.c.g); key.push_back(l.c.b);
            if (l.kind == CRAY_LIGHT_POINT || l.kind == CRAY_LIGHT_DISTANT) { key.push_back(l.v.x); key.push_back(l.v.y); key.push_back(l.v.z); }
            else if (l.kind == CRAY_LIGHT_AREA) {
                const cray_prim& p = d->prims[l.prim];
                key.push_back((double)p.shape_kind);
                if (p.shape_kind == CRAY_SHAPE_TRIANGLE) {
                    const double* t = (const double*)&d->triangles[p.shape];
                    key.insert(key.end(), t, t + 24);
                } else {
                    const cray_xf_shape& s = p.shape_kind == CRAY_SHAPE_SPHERE ? hs->spheres[p.shape] : hs->disks[p.shape];
                    key.push_back(s.radius); key.push_back(s.inner_radius);
                    key.insert(key.end(), s.m, s.m + 16);
                    key.insert(key.end(), s.inv, s.inv + 16);
                }
            }
            for (auto& x : key) if (x == 0.0) x = 0.0;  // -0.0 == +0.0 under PartialEq
            std::string sk((const char*)key.data(), key.size() * sizeof(double));
            auto f = seen.find(sk);
            if (f == seen.end()) { seen.emplace(std::move(sk), (int32_t)i); hs->first_equal[i] = (int32_t)i; }
            else hs->first_equal[i] = f->second;
        }
    }

    // --- Camera::new (camera.rs:56-76) + get_camera_from_raster_transformation (:25-53)
    cray_flat_scene& f = hs->flat;
    memset(&f, 0, sizeof(f));
    {
        const cray_camera_desc& c = d->camera;
        xform screen_from_camera, world_from_camera;
        bool ok = true;
        if (c.type == CRAY_CAMERA_PERSPECTIVE) ok = perspective(c.fov, 1e-2, 1000.0, screen_from_camera);  // :87-92
        else screen_from_camera = orthographic(0.0, 1.0);                                                 // :114-117
        ok = ok && look_at(mk(c.origin.x, c.origin.y, c.origin.z), mk(c.target.x, c.target.y, c.target.z), mk(c.up.x, c.up.y, c.up.z), world_from_camera);
        if (!ok) { set_last_error("camera matrix is singular (reference: unwrap on None)"); delete hs; return CRAY_ERR_BUILD; }
        double fw = (double)c.film_width;
        double fh = (double)c.film_width;  // sic: camera.rs:30 uses film.width for the height
        double sw, sh;
        if (fw > fh) { sw = fw / fh; sh = 1.0; } else { sw = 1.0; sh = fh / fw; }
        xform screen_from_raster = compose(scale(2.0 * sw / fw, -2.0 * sh / fh, 1.0), translate(-fw / 2.0, -fh / 2.0, 0.0));
        xform camera_from_raster = compose(swapped(screen_from_camera), screen_from_raster);
        store16(camera_from_raster.fwd, f.camera_from_raster);
        store16(world_from_camera.fwd, f.world_from_camera);
        f.camera_type = c.type;
        f.film_width = c.film_width; f.film_height = c.film_height;
        f.lens_radius = c.lens_radius; f.focal_distance = c.focal_distance;
    }
    f.abi_version = CRAY_ABI_VERSION;
    f.max_depth = d->max_depth; f.num_samples = d->num_samples;
    f.n_nodes = (uint32_t)hs->nodes.size(); f.nodes = hs->nodes.data();
    f.n_prim_refs = (uint32_t)hs->refs.size(); f.prim_refs = hs->refs.data();
    f.n_prims = d->n_prims; f.prims = d->prims;
    f.n_triangles = d->n_triangles; f.triangles = d->triangles;
    f.n_spheres = d->n_spheres; f.spheres = hs->spheres.data();
    f.n_disks = d->n_disks; f.disks = hs->disks.data();
    f.n_materials = d->n_materials; f.materials = d->materials;
    f.n_bxdfs = d->n_bxdfs; f.bxdfs = d->bxdfs;
    f.n_textures = d->n_textures; f.textures = d->textures;
    f.n_images = d->n_images; f.images = d->images;
    f.image_pool_bytes = d->image_pool_bytes; f.image_pool = d->image_pool;
    f.n_lights = d->n_lights; f.lights = d->lights;
    f.light_cdf = hs->cdf.data();
    f.first_equal_light = hs->first_equal.data();
    f.build_on_device = resident ? 1u : 0u;
    f.n_other_bounds = (uint32_t)hs->other_bounds.size();
    f.other_bounds = hs->other_bounds.data();
    hs->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *out = hs;
    return CRAY_OK;
}

extern "C" const cray_flat_scene* cray_host_scene_flat(const cray_host_scene* s) { return s ? &s->flat : nullptr; }
extern "C" double cray_host_scene_build_seconds(const cray_host_scene* s) { return s ? s->build_seconds : 0.0; }
extern "C" double cray_host_scene_bvh_seconds(const cray_host_scene* s, cray_bvh_build_stats* gpu) {
    if (!s) return 0.0;
    if (gpu) *gpu = s->gpu_build;
    return s->bvh_seconds;
}
extern "C" void cray_host_scene_free(cray_host_scene* s) { delete s; }

extern "C" void cray_host_sincos(double x, double* s, double* c) {
    double sv, cv;
    cray::sincos_cr(x, sv, cv);
    if (s) *s = sv;
    if (c) *c = cv;
}

extern "C" void cray_host_chacha_block(const uint32_t* key /* [8] */, const uint32_t* w12_15 /* [4] */, int double_rounds, uint32_t* out /* [16] */) {
    cray::chacha_block(key, w12_15[0], w12_15[1], w12_15[2], w12_15[3], double_rounds, out);
}
extern "C" void cray_host_independent_draws(uint64_t seed, uint64_t x, uint64_t y, uint64_t sample_index, uint32_t first, double* out /* [8] */) {
    uint32_t key[8];
    cray::indep_key(cray::indep_pixel_hash(seed, x, y, sample_index), key);
    cray::indep_draws(key, first, out);
}

extern "C" uint64_t cray_host_sincos_fast_check(const double* x, uint64_t n, double* stats /* [3] */) {
    using namespace cray;
    uint64_t bad = 0, slow = 0;
    double worst = 0.0;   // largest |short evaluation - double-double evaluation| / |value| seen, in units of 2^-64 (= kSinCosEps)
    for (uint64_t i = 0; i < n; i++) {
        double s1, c1, s2, c2;
        sincos_cr(x[i], s1, c1);
        sincos_cr_dd(x[i], s2, c2);
        if (memcmp(&s1, &s2, 8) != 0 || memcmp(&c1, &c2, 8) != 0) bad++;
        const SinCosArg A = sincos_reduce(x[i]);
        double sv, cv;
        if (!sincos_fast_core(A, sv, cv)) slow++;
        // deviation of the candidates from the double-double values (both as hi + lo pairs)
        double s_hi, s_lo, c_hi, c_lo;
        sincos_fast_parts(A, s_hi, s_lo, c_hi, c_lo);
        const dd l = A.l, S = A.S, C = A.C;
        (void)l; (void)S; (void)C;
        // recompute the double-double pair through the reference evaluation's own arithmetic
        const dd l2 = dd_mul(A.l, A.l);
        const double ts = -0x1.a01a01a01a01ap-13 + l2.hi * (0x1.71de3a556c734p-19 + l2.hi * -0x1.ae64567f544e4p-26);
        const double tc = -0x1.6c16c16c16c17p-10 + l2.hi * (0x1.a01a01a01a01ap-16 + l2.hi * (-0x1.27e4fb7789f5cp-22 + l2.hi * 0x1.1eed8eff8d898p-29));
        dd ps = dd_add(dd{0x1.1111111111111p-7, 0x1.1111111111111p-63}, dd_mul_d(l2, ts));
        ps = dd_add(dd{-0x1.5555555555555p-3, -(0x1.5555555555555p-57)}, dd_mul(l2, ps));
        ps = dd_add(dd{1.0, 0.0}, dd_mul(l2, ps));
        const dd sl = dd_mul(A.l, ps);
        dd pc = dd_add(dd{0x1.5555555555555p-5, 0x1.5555555555555p-59}, dd_mul_d(l2, tc));
        pc = dd_add(dd{-0.5, 0.0}, dd_mul(l2, pc));
        const dd cl = dd_add(dd{1.0, 0.0}, dd_mul(l2, pc));
        const dd sr = dd_add(dd_mul(A.S, cl), dd_mul(A.C, sl));
        const dd cr = dd_add(dd_mul(A.C, cl), dd_neg(dd_mul(A.S, sl)));
        const double ds = fabs((s_hi - sr.hi) + (s_lo - sr.lo)), dc = fabs((c_hi - cr.hi) + (c_lo - cr.lo));
        if (sr.hi != 0.0) worst = fmax(worst, ds / fabs(sr.hi) * 0x1p64);
        if (cr.hi != 0.0) worst = fmax(worst, dc / fabs(cr.hi) * 0x1p64);
    }
    if (stats) { stats[0] = (double)slow; stats[1] = worst; stats[2] = (double)n; }
    return bad;
}

extern "C" uint64_t cray_host_div_fast_mismatches(const double* a, const double* d, uint64_t n) {
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; i++) {
        const double aa = fabs(a[i]);
        if (!cray::div_fast_ok(d[i]) || !(a[i] == 0.0 || (aa >= 0x1p-553 && aa <= 0x1p501))) continue;  // a = bound - origin
        const double y = 1.0 / d[i];
        const double q = cray::div_fast(a[i], d[i], y), ref = a[i] / d[i];
        if (memcmp(&q, &ref, 8) != 0 && !(q == 0.0 && ref == 0.0)) bad++;  // the sign of a zero quotient is immaterial to the slab test
    }
    return bad;
}

extern "C" uint64_t cray_host_child_key_mismatches(const double* lo, const double* hi, const double* o, const double* d, uint64_t n,
                                                   uint64_t* n_checked) {
    using namespace cray;
    uint64_t bad = 0, checked = 0;
    for (uint64_t i = 0; i < n; i++) {
        const double *l = lo + 3 * i, *h = hi + 3 * i, *oo = o + 3 * i, *dd = d + 3 * i;
        bool ok = true;
        for (int k = 0; k < 3; k++) ok = ok && div_fast_ok(dd[k]) && div_range_ok(oo[k]) && div_range_ok(l[k]) && div_range_ok(h[k]) && l[k] <= h[k];
        if (!ok) continue;
        checked++;
        const vec3 ov = mk(oo[0], oo[1], oo[2]), dv = mk(dd[0], dd[1], dd[2]);
        const vec3 rd = mk(1.0 / dd[0], 1.0 / dd[1], 1.0 / dd[2]);
        const double a = child_key(l, h, ov, dv), b = child_key_fast(l, h, ov, dv, rd);
        const double code = child_key_code(l, h, ov, dv, rd), cc = canonical_key(code);   // what the kernels compute: same decisions
        if (memcmp(&a, &b, 8) != 0 || memcmp(&a, &cc, 8) != 0) { bad++; continue; }
        // ... and the comparison the traversal makes agrees for every positive ray.tmax, the candidates themselves included
        const double ts[6] = {kEps, 2.0 * kEps, a, nextafter(a, inf64()), 1.0, inf64()};
        for (double t : ts)
            if (t > 0.0 && ((a < t) != (code < t))) { bad++; break; }
    }
    if (n_checked) *n_checked = checked;
    return bad;
}

// Certified f32 culling of the triangle test (cray_math.h tri_cull32) against the literal f64 test: cray_cull_check.h.
extern "C" uint64_t cray_host_tri_cull_violations(const double* v0, const double* e1, const double* e2, const double* o, const double* d, const double* tmax,
                                                  uint64_t n, uint64_t* counts) {
    return cray::tri_cull_violations(v0, e1, e2, o, d, tmax, n, counts);
}

extern "C" uint64_t cray_host_hyb_key_violations(const double* lo, const double* hi, const double* o, const double* d, const double* tmax,
                                                 uint64_t n, uint64_t* counts /* [4]: resolve, visit, cull, out-of-range rays */) {
    using namespace cray;
    uint64_t bad = 0, cnt[4] = {0, 0, 0, 0};
    // boxes i and i + 1 are the two children of one node, tested by the ray and tmax of sample i through hyb_pair (what the
    // kernel runs at a node); box i again alone through hyb_key + hyb_status (what it runs at a pop)
    for (uint64_t i = 0; i + 1 < n; i++) {
        const double *oo = o + 3 * i, *dd = d + 3 * i;
        bool box_ok = true, fast = true;
        for (int c = 0; c < 2; c++) {
            const double *l = lo + 3 * (i + c), *h = hi + 3 * (i + c);
            box_ok = box_ok && hyb_scene_ok(l, h);
            for (int k = 0; k < 3; k++) { box_ok = box_ok && l[k] <= h[k]; fast = fast && div_range_ok(l[k]) && div_range_ok(h[k]); }
        }
        for (int k = 0; k < 3; k++) fast = fast && div_fast_ok(dd[k]) && div_range_ok(oo[k]);
        if (!box_ok) continue;   // such a scene never runs the hybrid kernel
        const vec3 ov = mk(oo[0], oo[1], oo[2]), dv = mk(dd[0], dd[1], dd[2]);
        const vec3 rd = mk(1.0 / dd[0], 1.0 / dd[1], 1.0 / dd[2]);
        const HybRay hr = hyb_ray(ov, dv, rd, fast);
        if (hr.a != hr.a) cnt[3]++;
        float t_lo, t_hi;
        hyb_tmax(tmax[i], t_lo, t_hi);
        hyb_f2 l32[3], h32[3];
        for (int k = 0; k < 3; k++)
            for (int c = 0; c < 2; c++) { l32[k][c] = f32_down(lo[3 * (i + c) + k]); h32[k][c] = f32_up(hi[3 * (i + c) + k]); }
        // hyb_pair once per order of the two children: the near child's decision and the far child's decision + key estimate
        int st_[2][2]; float kc_[2] = {0.f, 0.f};   // st_[c][0]: child c as the near one, [1]: as the far one
        for (int rf = 0; rf < 2; rf++) {
            const HybPair hp = hyb_pair(l32[0], l32[1], l32[2], h32[0], h32[1], h32[2], hr.px, hr.py, hr.pz, hr.rx, hr.ry, hr.rz, hr.a, t_lo, t_hi, rf != 0);
            const int cn = rf, cf = 1 - rf;
            st_[cn][0] = hp.cull_n ? kHybCull : (hp.visit_n ? kHybVisit : kHybResolve);
            st_[cf][1] = hp.cull_f ? kHybCull : (hp.visit_f ? kHybVisit : kHybResolve);
            kc_[cf] = hp.kc_f;
        }
        for (int c = 0; c < 2; c++) {
            const double key = child_key(lo + 3 * (i + c), hi + 3 * (i + c), ov, dv);   // the literal restatement of the reference
            const bool accept = key < tmax[i];
            if (st_[c][0] != st_[c][1]) bad++;   // a child is classified the same way wherever it stands in the order
            const int st = st_[c][0];
            if (c == 0) cnt[st]++;
            if ((st == kHybVisit && !accept) || (st == kHybCull && accept)) bad++;
            // the deferred form: the encoded key against another (smaller) tmax, as at a pop
            const double tm2 = c == 0 ? tmax[i] : tmax[i] * 0.5;
            float a_lo, a_hi;
            hyb_tmax(tm2, a_lo, a_hi);
            const int st2 = hyb_status(kc_[c], hr.a, a_lo, a_hi);
            const bool accept2 = key < tm2;
            if ((st2 == kHybVisit && !accept2) || (st2 == kHybCull && accept2)) bad++;
        }
        // the single-box classifier must encode the same key
        float l1[3] = {l32[0][0], l32[1][0], l32[2][0]}, h1[3] = {h32[0][0], h32[1][0], h32[2][0]};
        const float kc1 = hyb_key(l1, h1, hr);
        if (memcmp(&kc1, &kc_[0], 4) != 0 && !(kc1 != kc1 && kc_[0] != kc_[0])) bad++;
    }
    if (counts) for (int k = 0; k < 4; k++) counts[k] = cnt[k];
    return bad;
}
