// cray_device.h — device-side data layout in HBM and the per-path state (SoA).
//
// Layout decisions (DESIGN.md "Data layout in HBM"):
//  * BVH: one 128-byte record per *interior* node holding BOTH children's bounds, so
//    one dependent 128-B (one cache line) fetch yields two slab tests.  Leaves have no
//    record: a child reference with the top bit set encodes (first leaf slot, count).
//  * Triangles are re-ordered into leaf order ("slots"); a slot is one 80-byte record
//    (v0,e1,e2 + primitive id) = five 16-B loads per lane.
//  * Shading data (normals / uvs) stays in primitive order, touched once per closest hit.
//  * Path state is SoA over path slots so that wave loads are coalesced — and it is RE-COMPACTED every bounce (round 3): k_shade
//    reads the state of bounce b from one buffer and writes the surviving paths' state into the other at
//    slot = tile * tile_size + rank-within-tile, so the survivors of a tile are contiguous again (a sparse survivor set in a
//    fixed per-path layout makes every 8-B lane access its own 64-B sector: the texture addresser, not the VALU, then
//    bounds k_shade at the later bounces).  Shadow rays get their own compacted buffer; L stays per original path.
#pragma once

#include <cstddef>

#include "../../include/cray.h"
#include "cray_math.h"

namespace cray {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNoRef = 0xffffffffu;
constexpr int kStackDepth = 96;
#ifndef CRAY_SHADE_TILE
#define CRAY_SHADE_TILE 2048
#endif
#ifndef CRAY_SHADE_LDS_TABLES
#define CRAY_SHADE_LDS_TABLES 40960
#endif
constexpr uint32_t kShadeLdsTables = CRAY_SHADE_LDS_TABLES;  // LDS bytes k_shade may fill with the small shading tables (2 blocks of 16 + 40 KB per CU)
constexpr uint32_t kShadeTile = CRAY_SHADE_TILE;  // paths per block-level queue flush in k_shade (8 x 256)
#ifndef CRAY_LDS_STACK
#define CRAY_LDS_STACK 12
#endif
constexpr int kLdsStack = CRAY_LDS_STACK;  // entries per lane kept in LDS (12 B each)
// Round 5: the launches that read the certified-f32 records run FIVE waves per SIMD.  Their loop needs 105 vector registers at four
// waves and fits the 96 of five with no spill (bounce 0) or two (mixed); five resident blocks need a block's LDS under 32 KB, hence 10
// stack entries per lane there (30 KB) and room for four sphere / disk records instead of eight (1 KB; cray_kernels.h).  configs[2]:
// bounce 0 22.0 -> 19.9 ms, mixed 107.3 -> 99.1, configs[3] 745 -> 692 ms (profiles/r05_five_waves_ab.log; 9 entries + eight records:
// mixed 100.9, 8 entries: 104-107).  The f64-record instantiations (124-126 registers) stay at four: at 96
// registers they spill 26-46 (configs[1], which runs on them, 13.1 -> 15.3 ms).
#ifndef CRAY_LDS_STACK_HYB
#define CRAY_LDS_STACK_HYB 10
#endif
#ifndef CRAY_TRACE_WAVES_HYB
#define CRAY_TRACE_WAVES_HYB 5
#endif
#ifndef CRAY_TRACE_WAVES
#define CRAY_TRACE_WAVES 4
#endif
constexpr int trace_lds_stack(int hyb) { return hyb ? CRAY_LDS_STACK_HYB : CRAY_LDS_STACK; }
constexpr int trace_waves(int hyb) { return hyb ? CRAY_TRACE_WAVES_HYB : CRAY_TRACE_WAVES; }   // waves per SIMD = resident blocks per CU

// child reference: interior -> index into DevScene::inner;
// leaf -> kLeafBit | first_slot << 3 | (count - 1), count in 1..8
CRAY_HD bool ref_is_leaf(uint32_t r) { return (r & kLeafBit) != 0; }
CRAY_HD uint32_t ref_leaf_first(uint32_t r) { return (r & ~kLeafBit) >> 3; }
CRAY_HD uint32_t ref_leaf_count(uint32_t r) { return (r & 7u) + 1u; }

struct alignas(128) InnerNode {
    double lo0[3], hi0[3];  // left child's Bounds
    double lo1[3], hi1[3];  // right child's Bounds
    uint32_t ref0, ref1;    // left / right child reference
    uint32_t axis;          // split axis of THIS node (orders the children, bvh.rs:92-98)
    uint32_t pad_;
    double pad2_[2];
};
static_assert(sizeof(InnerNode) == 128, "InnerNode must be one 128-B line");

struct alignas(16) LeafSlot {
    double v0[3], e1[3], e2[3];  // Shape::Triangle geometry; sphere / disk slots: zeros, and the shape index in the bits of v0[0]
    uint32_t prim;               // index into prims[]
    uint32_t kind;               // CRAY_SHAPE_*
};
static_assert(sizeof(LeafSlot) == 80, "LeafSlot is five 16-B loads");

// ---- "fast" mode (cray_render_params.precision = CRAY_PRECISION_F32_TRAVERSAL): the same tree, f32 records.
// Bounds are rounded OUTWARD (lo down, hi up), so a box never shrinks; triangles are rounded to nearest.  NOT bit-exact with
// the reference: reported separately.
struct alignas(64) InnerNode32 {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    uint32_t ref0, ref1, axis, pad_;
};
static_assert(sizeof(InnerNode32) == 64, "InnerNode32 is four 16-B loads");
struct alignas(16) LeafSlot32 {
    float v0[3], e1[3], e2[3];
    uint32_t prim, kind, shape;   // shape: index into spheres[] / disks[] for those kinds
};
static_assert(sizeof(LeafSlot32) == 48, "LeafSlot32 is three 16-B loads");

// ---- certified f32 culling of the exact traversal (cray_math.h hyb_pair): the two children's bounds, rounded outward to f32
// and INTERLEAVED by child — (lo[axis][child], hi[axis][child]) — so that a 64-bit register pair holds one coordinate of both
// boxes and the slab arithmetic of the pair is packed f32 math.
struct alignas(64) InnerNodeH {
    float lo[3][2], hi[3][2];
    uint32_t ref0, ref1, axis, pad_;
};
static_assert(sizeof(InnerNodeH) == 64, "InnerNodeH is four 16-B loads");
// ---- round 5: the certified-f32 records of a scene live in ONE arena — n_inner InnerNodeH, then the leaf slots (LeafSlot, 80 B, as in
// slots[]) — and a child reference of an InnerNodeH IS the byte offset (a multiple of 16) of what a lane fetches next:
//     interior child:  offset of its InnerNodeH
//     leaf child:      offset of its first slot | kHLeaf | (count - 1),  count in 1..8
// so the fetch address of an interior node or a leaf slot is arena + (ref & ~15): one instruction where the three arrays of rounds
// 2-4 took seventeen (three kinds of index, three bases behind selects).  k_trace_mixed 111.4 -> 108.0 ms, bounce 0 23.2 -> 22.4.
constexpr uint32_t kHLeaf = 8u;
CRAY_HD bool href_is_leaf(uint32_t r) { return (r & kHLeaf) != 0; }
CRAY_HD uint32_t href_off(uint32_t r) { return r & ~15u; }
CRAY_HD uint32_t href_more(uint32_t r) { return r & 7u; }   // slots of the leaf after this one

struct TriShade {
    double n0[3], n01[3], n02[3];
    double uv0[2], uv01[2], uv02[2];
};

// One entry per light; area lights carry their own shape's geometry so that
// Shape::sample / Shape::pdf_from need no second lookup (src/light.rs:114-131, shape.rs:445-514).
struct DevLight {
    int32_t kind;        // CRAY_LIGHT_*
    int32_t shape_kind;  // area lights: CRAY_SHAPE_*
    uint32_t shape;      // index into spheres[] / disks[]
    uint32_t pad_;
    double v[3];         // Point origin / Distant direction
    double c[3];         // intensity / emittance
    double v0[3], e1[3], e2[3];  // triangle emitters
    double area;         // Shape::area() (sphere: PI r^2 as in the reference, shape.rs:506)
};

struct DevScene {
    // Scene / Camera
    uint32_t max_depth, num_samples, film_w, film_h;
    int32_t camera_type;
    uint32_t n_lights;
    double lens_radius, focal_distance;
    double camera_from_raster[16], world_from_camera[16];
    // BVH
    double root_lo[3], root_hi[3];
    uint32_t root_ref, n_inner;
    uint32_t bounds_in_div_range;  // every node bound is 0 or within [2^-500, 2^500]: div_fast is exact
    uint32_t root_ref_h;           // root_ref in the arena's encoding (set with innerh)
    const InnerNode* inner;
    const LeafSlot* slots;
    const InnerNode32* inner32;   // fast mode only (built on first use)
    const InnerNodeH* innerh;     // certified f32 culling only (built on first use): the ARENA — n_inner InnerNodeH, then the leaf slots
    const LeafSlot32* slots32;
    // primitives
    const cray_prim* prims;
    const TriShade* tri_shade;  // indexed by prims[].shape for triangles
    const cray_xf_shape* spheres;
    const cray_xf_shape* disks;
    // materials
    const cray_material* materials;
    const cray_bxdf* bxdfs;
    const cray_texture* textures;
    const cray_image* images;
    const uint8_t* pool;
    const double* gamma_lut;  // (c/255)^2.2 for c in 0..255 (Color::from_rgb, color.rs:39-46)
    uint32_t n_materials, n_bxdfs, n_textures, n_images, n_spheres, n_disks;
    uint32_t shade_stage_shapes, pad_tab_;   // the sphere / disk tables fit the staging area as well
    uint32_t shade_tables_bytes;  // > 0: materials + bxdfs + textures + lights + light tables fit k_shade's LDS staging area (bytes)
    // lights
    const DevLight* lights;
    const double* light_cdf;
    const int32_t* first_equal_light;
    // sampler
    const uint16_t* sobol;  // [64][16][4] bit-reversed direction vectors
};

// A VIEW of the path state, SoA: three index spaces share one struct of pointers.
//   live slots   (ox .. hv, hprim, br .. bb, prev_pdf, hash, flags, p0): the paths alive at one bounce, compacted per k_shade
//                tile; two such buffers alternate (k_shade reads bounce b from one, writes bounce b + 1 into the other).  At
//                bounce 0 the live slot of a path is its original index (k_raygen), p0 is not read.
//   shadow slots (sox .. stmax, cr .. cb, sp0, sprim): the shadow rays of one bounce, compacted the same way.
//   original path index p0 = pixel-in-pass * spp_pass + sample (lr, lg, lb): what k_film sums.
struct PathState {
    double *ox, *oy, *oz, *dx, *dy, *dz;   // current ray (tmax is +inf for path segments)
    double *br, *bg, *bb;                  // beta
    double *lr, *lg, *lb;                  // L, per ORIGINAL path
    double* prev_pdf;                      // prev_bsdf_pdf
    double *ht, *hu, *hv;                  // closest hit: distance, triangle barycentrics
    int32_t* hprim;                        // closest hit primitive (-1 = miss); before the trace of a bounce: the primitive the segment starts on
    double *sox, *soy, *soz, *sdx, *sdy, *sdz, *stmax;  // shadow ray
    double *cr, *cg, *cb;                  // NEE contribution added if the shadow ray is unoccluded
    uint32_t* hash;                        // per-pixel Sobol seed (SipHash-1-3)
    uint32_t* flags;                       // bit0: is_specular_bounce
    uint32_t* p0;                          // live slot -> original path index
    uint32_t* sp0;                         // shadow slot -> original path index (where the NEE term is added)
    int32_t* sprim;                        // shadow slot -> primitive the shadow ray starts on (f32 fast mode only reads it)
};

struct Counters {
    unsigned long long closest_rays, shadow_rays;
    unsigned long long closest_nodes, closest_prims, shadow_nodes, shadow_prims;
    unsigned long long closest_tri, shadow_tri;
    unsigned long long nonfinite, stack_overflow, shadow_skipped, closest_hits;
    // The words a bounce zeroes sit between the two queue lengths, so that ONE 16-byte memset does it for either parity:
    // bounce b zeroes n_active[(b + 1) & 1] (the queue k_shade fills), n_shadow, trace_head and shade_head, and keeps n_active[b & 1].
    unsigned int n_active0, n_shadow, trace_head, shade_head, n_active1, pad_head_;
    unsigned long long tail_helped, tail_again;   // small launches: parts of closest-hit rays walked by helpers; rays walked again alone
    unsigned long long diag[32];  // k_trace lane-occupancy diagnostics (CRAY_TRACE_DIAG builds only)
    // Third level of the traversal stack (LDS -> scratch -> here): entries kStackDepth.. of every lane, in global
    // memory as [entry][global thread].  Allocated by the runtime only after a frame overflowed the first two levels.
    uint32_t* deep_ref;
    double* deep_key;
    unsigned int deep_depth;
    // Small launches (round 4, trace_body TAIL): a mixed launch of fewer than tail_rays rays (0: never) is traced by the
    // instantiation that lets idle lanes take over parts of unfinished rays of BOTH kinds; tail_res holds what the helpers of a
    // closest-hit ray found: [global thread of the ray's lane][kTailSlots][t, u, v, primitive | certified << 32].
    unsigned int tail_rays;
    double* tail_res;
};
constexpr uint32_t kTailSlots = 4;   // helpers one closest-hit ray can have had by the time it ends

}  // namespace cray
