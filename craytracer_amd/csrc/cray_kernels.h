// cray_kernels.h — the gfx950 kernels of the wavefront path tracer.
//
//   k_raygen        craytracer.rs:148-156 (render_pixel: start_pixel, film+lens samples, Camera::sample)
//   k_trace<ANY>    bvh.rs:58-104 (closest) / :106-147 (any hit) + bounds.rs:46-88 + shape.rs:157-400
//   k_shade         path_integrator.rs:54-212 minus the two BVH queries
//   k_film          craytracer.rs:175-188 (f64 batch sum -> f32 add), k_resolve :253-259
//
// One loop iteration of estimate_Li is split at its two BVH queries:
//   trace_closest -> shade (emission, NEE set-up, BSDF sample, roulette) -> trace_any (adds the
//   NEE term if unoccluded) -> next bounce.
// The additions to L therefore happen in the reference's order, and every path's result is
// independent of scheduling (outputs go to per-path slots, queues only carry indices).
#pragma once

#include "cray_shading.h"

namespace cray {

constexpr int kBlock = 256;

// child_key / child_key_fast (the slab test of one child box, split into its ray.tmax-independent part) live in
// cray_math.h so that the host can test them against each other.

// ---------------------------------------------------------------------------------
// Scene::intersect / Scene::intersects over a queue of paths — persistent wavefront tracer.
//
// Rays of one launch need between ~5 and several hundred node visits; with one ray per lane
// for the lifetime of a wave the 64-wide VALU ran at 15-19 % lane utilisation (rocprofv3
// SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, profiles/r01_v1_*).  Here every lane is a small
// state machine: a lane whose ray is finished pulls the next ray index from a device-wide
// queue head (one atomic per wave, ballot + prefix), and each loop iteration performs at most
// one interior-node step followed by at most one leaf step, so lanes re-converge every
// iteration.  Results do not depend on which lane traces which ray.
//
//  ANY = false: ray = path segment (tmax = +inf); writes the hit record.
//  ANY = true : ray = the path's shadow ray; adds the stored NEE term to L when unoccluded
//               (path_integrator.rs:141-163).
//  COUNT      : also count popped nodes / primitive tests exactly as the reference visits them.
// ---------------------------------------------------------------------------------
// (waves per SIMD of the traversal launches: trace_waves(HYB), cray_device.h)
#define CRAY_TRACE_EU
//  MODE 2 (mixed): ONE launch traces the path segments of bounce b+1 (positions [0, n_closest) of a virtual queue) and
//  the shadow rays of bounce b (the rest).  The two depend only on k_shade of bounce b, not on each other
//  (the shadow rays add to L, the segments write hit records), and together they have one drain phase instead
//  of two — the lanes of a draining any-hit launch were 84 % idle for a fifth of its iterations.
enum { kTraceClosest = 0, kTraceAny = 1, kTraceMixed = 2 };
constexpr uint32_t kTraceLdsShapes = 8;   // sphere + disk records (272 B each) a traversal block stages in LDS
constexpr uint32_t kTraceLdsShapesHyb = 4;   // ... in the five-waves instantiations: five blocks of 30 KB of stack + 1 KB of these fit a CU's 160 KB
constexpr uint32_t trace_lds_shapes(int hyb) { return hyb ? kTraceLdsShapesHyb : kTraceLdsShapes; }
// the slab summary of a child box the timed and the counting kernels use (cray_math.h): encoded special values by default,
// -DCRAY_KEY_PLAIN=1 for the +-inf form (A/B builds)
#ifdef CRAY_KEY_PLAIN
#define CRAY_CHILD_KEY child_key_fast
#else
#define CRAY_CHILD_KEY child_key_code
#endif
//  HYB        : certified f32 culling (cray_math.h hyb_pair): interior nodes are read as 64-B f32 records, every decision the f32
//               enclosure cannot certify is retaken from the f64 record in a RESOLVE step of the lane (cray_trace_step_hyb.inc).
//               Same hits, same counters.
//  SHAPES_LDS : the scene's few sphere / disk records are staged in LDS by every block (the host picks this instantiation when they fit).
//  TAIL       : the instantiation for SMALL mixed launches (fewer than Counters::tail_rays rays): such a launch is all drain — every
//               lane holds one or two rays from the start and the launch lasts as long as its longest ray — so here idle lanes
//               take over parts of unfinished rays of both kinds.  Shadow rays as in the any-hit launch (STEAL).  A CLOSEST-hit ray
//               hands the bottom entry of its stack to a helper, which walks that subtree with the ray's tmax of that moment —
//               never below the tmax the reference has when it gets there, so the helper enters a superset of the reference's
//               nodes and accepts a superset of its hits.  Its best hit is the reference's answer for the subtree if every box on
//               the path from the stolen node to the hit's leaf has a key below the hit distance (then the reference, whatever
//               its tmax, enters them all): the helper tracks "keys non-decreasing along my path" (one bit per deferred child)
//               and certifies a hit when that holds and the leaf's box key is below the hit.  The ray's own lane walks the
//               reference's order with the reference's tmax (what it gave away comes LAST in that order), then folds the helpers'
//               results in traversal order — latest helper first, a tie keeps the earlier hit, as the reference's strict `<` does.
//               A helper's hit that would win without being certified — the reference's leak cases — makes the lane walk THAT
//               part again itself, at its place in the order and with the tmax the fold has reached: the reference's own walk
//               of it.  Helpers do not hand work on.
enum : uint32_t { kTfThief = 1u, kTfMono = 2u, kTfCert = 4u, kTfWait = 8u, kTfNoDonate = 16u };   // + bits 8..10 helpers so far, 12..13 this helper's slot
// The path state's ~30 array pointers are needed when a lane takes a ray or hands back a result — once in thirty to fifty
// iterations.  Held in scalar registers for the whole loop (a kernel argument passed by value is loaded once, at the top) they
// pushed the pointers the loop needs in EVERY iteration out into lanes of a vector register, a v_readlane each time.  So the
// body reads them where it needs them, from the kernel-argument segment itself: ps_here() makes the pointer opaque to the
// compiler at that place, which keeps the loads there instead of hoisted in front of the loop.
typedef const PathState __attribute__((address_space(4))) * PsArg;
__device__ __forceinline__ PsArg ps_here(PsArg a) {
    asm volatile("" : "+s"(a));
    return a;
}
// (k_trace and k_trace_mixed both start with DevScene sc, PathState ps)
constexpr size_t kPsArgOffset = (sizeof(DevScene) + alignof(PathState) - 1) / alignof(PathState) * alignof(PathState);
__device__ __forceinline__ PsArg ps_kernarg() {
    return (PsArg)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kPsArgOffset);
}
// What ties ps_kernarg() to the kernel signatures: a kernel with the same leading arguments as k_trace / k_trace_mixed (DevScene,
// PathState) and k_shade (DevScene, PathState, PathState) compares what it reads from the kernel-argument segment at the
// hand-computed offsets with the by-value arguments themselves.  cray_ctx_create launches it once and refuses the context on a
// mismatch — a reordered signature or another by-value lowering would otherwise show as a memory fault in the middle of a frame.
__global__ void k_kernarg_check(DevScene sc, PathState ps, PathState po, uint32_t* __restrict__ bad) {
    const PsArg a = ps_kernarg();
    const PsArg b = (PsArg)((const char __attribute__((address_space(4)))*)a + sizeof(PathState));
    uint32_t e = 0;
    e |= (a->ox != ps.ox || a->hprim != ps.hprim || a->sprim != ps.sprim || a->lr != ps.lr) ? 1u : 0u;
    e |= (b->ox != po.ox || b->hprim != po.hprim || b->sprim != po.sprim || b->lr != po.lr) ? 2u : 0u;
    e |= sc.max_depth != 0x5a5a5a5au ? 4u : 0u;
    if (threadIdx.x == 0 && blockIdx.x == 0) *bad = e;
}

template <int MODE, bool COUNT, int HYB, bool SHAPES_LDS = false, bool TAIL = false, bool DEEP = false>
__device__ __forceinline__ void trace_body(const DevScene& sc, const PsArg ps_arg, const uint32_t* __restrict__ queue, const uint32_t n_first,
                                           const uint32_t* __restrict__ queue_b, const uint32_t n_b, const double* __restrict__ closest_tmax,
                                           Counters* ctr, unsigned int* work_head, unsigned int refill_min) {
    static_assert(!(COUNT && MODE == kTraceMixed), "traversal counting uses the separate launches");
    constexpr bool ANY = MODE == kTraceAny;   // for the counting code, which never runs mixed
    // ---- the drain of a launch: idle lanes take over parts of the shadow rays that are still running (round 4).
    // Once the queue is empty a launch lasts as long as its longest rays: 300-400 dependent node visits on a chip that empties,
    // ~0.7 ms per launch whatever its size (3 % of the whole frame, a fifth of an eighth of it).  An any-hit query never changes
    // ray.tmax (bvh.rs:106-147: `intersects` borrows the ray immutably), so the nodes it visits form a fixed tree — every node
    // all of whose ancestors pass `key < tmax` — and its answer is the OR over the primitives in the leaves of that tree:
    // independent of the order, hence of WHO walks which part.  A deferred child on a lane's stack is the root of a part nobody
    // has started; an idle lane of the same wave takes the BOTTOM entry (the shallowest deferral: the largest part) together
    // with a copy of the ray and walks it with its own stack.  The workers of one ray share a counter and an "occluded" flag in
    // LDS; a hit anywhere stops them all, and the worker that finishes last adds the NEE term when nobody hit
    // (path_integrator.rs:141-163).  Closest-hit queries are NOT split: their tmax shrinks with every accepted hit, and where the
    // reference's slab test and triangle test disagree in the last bit (scenes/rounding-error.cry) the result depends on the order.
    // Not used by the counting builds (the reference's early exit defines their counters) nor with HYB.
    // Built into the any-hit launch only (the last bounce of a pass): inside the loop of the mixed launches the same code cost
    // 9 % of the bulk (two spilled registers, compares in every iteration) for -0.2 ms of tail per launch — their tails are
    // mostly closest-hit rays once a launch is small (profiles/r04_experiments.md).  The opt-in small-launch instantiation
    // (TAIL, below) has it as well, together with the certified split of closest-hit rays.
    static_assert(!TAIL || (MODE == kTraceMixed && !COUNT && HYB == 0), "the small-launch instantiation reads f64 records");
    static_assert(HYB == 0 || HYB == 1, "records: 0 f64, 1 certified f32 culling");
    static_assert(!DEEP || (HYB == 0 && !SHAPES_LDS && !TAIL), "the instantiations with the third stack level read f64 records");
    constexpr bool STEAL = !COUNT && !HYB && (MODE == kTraceAny || TAIL);
    const bool steal_on = STEAL && (refill_min & 0x8000u) != 0;
    unsigned int age = 0, age_min = 0;   // TAIL: iterations this lane's segment has been walked / before it may hand parts out
    if (MODE == kTraceMixed && !COUNT) {   // a mixed launch runs in one of two instantiations, by its size (both are launched)
        const unsigned int tr = ctr->tail_rays & 0xffffffu;
        age_min = (ctr->tail_rays >> 24) * 4u;
        if (TAIL ? (tr == 0u || n_first + n_b >= tr) : (tr != 0u && n_first + n_b < tr)) return;
    }
    const uint32_t n = n_first + n_b;
    bool is_any = MODE == kTraceAny;          // per lane in mixed mode
#define CRAY_ANY_LANE (MODE == kTraceMixed ? is_any : (MODE == kTraceAny))
    const unsigned int lane = __lane_id();
    unsigned long long n_nodes = 0, n_prims = 0, n_tri = 0;
    unsigned int overflow = 0;
#ifdef CRAY_TRACE_DIAG
    unsigned long long dg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    constexpr int kLds = trace_lds_stack(HYB);   // stack entries per lane in LDS (cray_device.h)
    // Traversal stack: the bottom kLds entries of every lane live in LDS ([entry][thread], so a
    // wave's access is conflict-free), deeper entries (rare) spill to scratch.  An entry is 12 B: the child reference and its
    // f64 key, or (HYB) the reference, the encoded f32 key estimate and parent | side << 31 for a later exact test.
    // (one array [entry][word][thread] — reference, f64 key as two words / HYB: key estimate, parent — so that the pop loop walks
    // it with ONE address register and immediate offsets)
    __shared__ uint32_t lds_st[kLds * 3 * kBlock];
#define CRAY_LDS_REF(i_, t_) lds_st[((i_) * 3) * kBlock + (t_)]
#define CRAY_LDS_W0(i_, t_) lds_st[((i_) * 3 + 1) * kBlock + (t_)]
#define CRAY_LDS_W1(i_, t_) lds_st[((i_) * 3 + 2) * kBlock + (t_)]
    uint32_t sst[(kStackDepth - kLds) * 3];   // (scratch level: the three words of an entry side by side, one address, one load)
    const unsigned int tid = threadIdx.x;
    // work sharing in the drain (STEAL): per ray that has been split — indexed by the thread that fetched it from the queue, its
    // "owner" — the number of workers still walking and whether one of them found an occluder; `steal_pair` matches the k-th idle
    // lane of a wave with its k-th donor.  All three are only ever touched by the lanes of ONE wave (its own 64 entries).
    __shared__ uint32_t grp_cnt[STEAL ? kBlock : 1], grp_occ[STEAL ? kBlock : 1], steal_pair[STEAL ? kBlock : 1];
    // The sphere / disk records (2 x 16 f64 of transform + 2 radii each: 272 B) of a scene that has only a few of them sit in LDS
    // for the launch: a lane that reaches such a slot — in 13.5 % of configs[2]'s wave-iterations one does, the ground sphere —
    // would otherwise make the whole wave wait for a second, dependent fetch from global memory inside the iteration
    // (CRAY_TRACE_DIAG, profiles/r04_experiments.md: k_trace_mixed 137.0 -> 131.5 ms on f64 records, 131.9 -> 128.9 with f32 culling).
    // Its own instantiation: through one generic pointer for both cases the scenes that do NOT fit paid 9 % (flat loads).
    __shared__ double lds_shapes[SHAPES_LDS ? trace_lds_shapes(HYB) * (sizeof(cray_xf_shape) / 8) : 1];
    const cray_xf_shape* spheres = sc.spheres;
    const cray_xf_shape* disks = sc.disks;
    // (the host launches this instantiation only when 0 < n_spheres + n_disks <= trace_lds_shapes(HYB), and sets bit 14 of refill_min.
    // For the f64 instantiations the bit is tested, so that the compiler cannot prove where the records are: with pointers it
    // KNOWS to be LDS k_trace_mixed<0> came out with two registers spilled in its loop and the gain was gone — 138.0 ms against
    // 131.4 with this test and 137.1 without any staging, profiles/r04_lds_shapes_ab.log.)
    if (SHAPES_LDS && (HYB != 0 || (refill_min & 0x4000u) != 0)) {
        constexpr uint32_t per = sizeof(cray_xf_shape) / 8;
        const double* gs = reinterpret_cast<const double*>(sc.spheres);
        const double* gd = reinterpret_cast<const double*>(sc.disks);
        for (uint32_t i = threadIdx.x; i < sc.n_spheres * per; i += kBlock) lds_shapes[i] = gs[i];
        for (uint32_t i = threadIdx.x; i < sc.n_disks * per; i += kBlock) lds_shapes[sc.n_spheres * per + i] = gd[i];
        __syncthreads();
        spheres = reinterpret_cast<const cray_xf_shape*>(lds_shapes);
        disks = reinterpret_cast<const cray_xf_shape*>(lds_shapes + sc.n_spheres * per);
    }
    int sbase = 0;              // bottom of this lane's stack: entries [sbase, sp) are its own, [0, sbase) were given away
    uint32_t owner = tid;       // thread whose grp_cnt / grp_occ entry this lane's ray uses
    bool shared = false;        // this lane's ray has (had) other workers
    double kcur = 0.0;          // TAIL, helper of a closest-hit ray: the key of the node `cur`
    uint32_t tfl = 0;           // TAIL: kTf* flags of this lane
    unsigned int n_help = 0, n_again = 0;   // TAIL: parts of segments this lane handed to helpers / rays it walked again (cray_stats.tail_split)
    // (r_: reference, w0_ / w1_: the two payload words)
    // (third level, DEEP instantiations only: the code of a level that is almost never there cost the loop 1.5 % by being there,
    // profiles/r05_experiments.md — the runtime launches those after a frame reported an overflow, and they read the f64 records)
#define CRAY_DEEP_PUSH(r_, w0_, w1_)                                                       \
        else if (DEEP && sp < kStackDepth + (int)ctr->deep_depth) {                        \
            const size_t at_ = (size_t)(sp - kStackDepth) * ((size_t)gridDim.x * kBlock) + (size_t)blockIdx.x * kBlock + tid; \
            ctr->deep_ref[at_] = (r_);                                                     \
            ctr->deep_key[at_] = __hiloint2double((int)(w1_), (int)(w0_)); sp++;           \
        } else overflow = 1;
#define CRAY_DEEP_POP(r_, w0_, w1_)                                                        \
        else if (DEEP) {                                                                   \
            const size_t at_ = (size_t)(sp - kStackDepth) * ((size_t)gridDim.x * kBlock) + (size_t)blockIdx.x * kBlock + tid; \
            const double dk_ = ctr->deep_key[at_];                                         \
            r_ = ctr->deep_ref[at_]; w0_ = (uint32_t)__double2loint(dk_); w1_ = (uint32_t)__double2hiint(dk_);  \
        } else { r_ = 0u; w0_ = 0u; w1_ = 0u; }
#define CRAY_PUSH_W(r_, w0_, w1_)                                                          \
    do {                                                                                   \
        if (sp < kLds) { CRAY_LDS_REF(sp, tid) = (r_); CRAY_LDS_W0(sp, tid) = (w0_); CRAY_LDS_W1(sp, tid) = (w1_); sp++; } \
        else if (sp < kStackDepth) { sst[(sp - kLds) * 3] = (r_); sst[(sp - kLds) * 3 + 1] = (w0_); sst[(sp - kLds) * 3 + 2] = (w1_); sp++; }      \
        CRAY_DEEP_PUSH(r_, w0_, w1_)                                                       \
    } while (0)
#define CRAY_PUSH(r_, k_) CRAY_PUSH_W(r_, (uint32_t)__double2loint(k_), (uint32_t)__double2hiint(k_))
#define CRAY_PUSH_H(r_, kc_, par_) CRAY_PUSH_W(r_, __float_as_uint(kc_), (par_))
#define CRAY_POP_W(r_, w0_, w1_)                                                           \
    do {                                                                                   \
        if (sp < kLds) { r_ = CRAY_LDS_REF(sp, tid); w0_ = CRAY_LDS_W0(sp, tid); w1_ = CRAY_LDS_W1(sp, tid); } \
        else if (sp < kStackDepth) { r_ = sst[(sp - kLds) * 3]; w0_ = sst[(sp - kLds) * 3 + 1]; w1_ = sst[(sp - kLds) * 3 + 2]; }               \
        CRAY_DEEP_POP(r_, w0_, w1_)                                                        \
    } while (0)
    int sp = 0;
    bool active = false, exhausted = false;
    bool pending = false;  // ray finished, result still in registers (written at the next refill)
    unsigned int res_base = 0, res_left = 0;  // wave-uniform reserve of queue positions
#ifndef CRAY_CHUNK_MAX
#define CRAY_CHUNK_MAX 512u   // round 3 (f64 records): 256 best by ~0.5 %.  Round 4 (f32 culling; whole frame, bounce 0 / mixed): 128: 29.0 /
                              // 127.3 ms, 256: 26.5 / 126.7, 512: 25.95 / 126.6, 1024: 26.1 / 127.8, 2048: 26.5 / 130.3, 4096: 27.6 / 135.2.  But what
                              // a wave holds in reserve is what a SMALL launch waits for (an eighth of the frame, bounce 0: 3.8 ms at 256, 7.0 at
                              // 1024), hence at least 16 chunks per wave below, not 4 (profiles/r04_chunk_size_ab.log)
#endif
    // (the f64 instantiations keep round 3's rule — at least four chunks per wave, at most 256 positions: with the new one
    // configs[1]'s mixed launches were 2.5 % slower, 7.97 -> 8.17 ms)
    unsigned int chunk = n / (gridDim.x * (kBlock / 64) * (HYB ? 16u : 4u));
    chunk = chunk < 64u ? 64u : (chunk > (HYB ? CRAY_CHUNK_MAX : 256u) ? (HYB ? CRAY_CHUNK_MAX : 256u) : chunk);
    uint32_t p = 0, cur = 0;
    ray_t ray = mkray(mk(0, 0, 0), mk(0, 0, 1));
    vec3 rd = mk(0, 0, 1);   // 1 / ray.d per axis (exact-division helper)
    bool fast_div = false;   // operands of this ray are inside div_fast's proven range
    uint32_t dneg = 0;       // bit k: ray.d[k] < 0 (the child order at a node split on axis k)
    double hit_t = 0.0, hit_u = 0.0, hit_v = 0.0;
    int32_t hit_prim = -1;
    // HYB: the f32 view of the ray, [t_lo, t_hi] around ray.tmax, and the RESOLVE state: `cur` was reached on an uncertified
    // decision, `res` = parent | side << 31 names the f64 bounds that decide it
    float h_px = 0.f, h_py = 0.f, h_pz = 0.f, h_rx = 1.f, h_ry = 1.f, h_rz = 1.f, h_a = 0.f;   // HybRay, kept as scalars (an aggregate in the
                                                                                              // lane state ends up partly in LDS)
    float t_lo = 0.f, t_hi = 0.f;
    uint32_t res = 0;
    bool resolve = false;

    if constexpr (HYB != 0) {
        unsigned long long active_m = 0ull, resolve_m = 0ull;   // the mixed / any-hit launches' `active` and `resolve` between iterations (step file)
        // (how the loop is spelled: see the head of cray_trace_step.inc)
        auto step = [&]() __attribute__((always_inline)) -> bool {
#define CRAY_STEP_BREAK return true
#define CRAY_STEP_CONTINUE return false
#include "cray_trace_step_hyb.inc"
#undef CRAY_STEP_BREAK
#undef CRAY_STEP_CONTINUE
            return false;
        };
        for (;;)
            if (step()) break;
    } else {
        for (;;) {
#define CRAY_STEP_DRAIN (STEAL && steal_on && exhausted)
#define CRAY_STEP_BREAK break
#define CRAY_STEP_CONTINUE continue
#include "cray_trace_step.inc"
#undef CRAY_STEP_DRAIN
#undef CRAY_STEP_BREAK
#undef CRAY_STEP_CONTINUE
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (MODE == kTraceMixed) {
            atomicAdd(&ctr->shadow_rays, (unsigned long long)n_first);
            atomicAdd(&ctr->closest_rays, (unsigned long long)n_b);
        } else {
            atomicAdd(ANY ? &ctr->shadow_rays : &ctr->closest_rays, (unsigned long long)n);
        }
    }
    if (COUNT) {
        if (n_nodes) atomicAdd(ANY ? &ctr->shadow_nodes : &ctr->closest_nodes, n_nodes);
        if (n_prims) atomicAdd(ANY ? &ctr->shadow_prims : &ctr->closest_prims, n_prims);
        if (n_tri) atomicAdd(ANY ? &ctr->shadow_tri : &ctr->closest_tri, n_tri);
    }
    if (overflow) atomicAdd(&ctr->stack_overflow, 1ull);
    if (TAIL) {
        if (n_help) atomicAdd(&ctr->tail_helped, (unsigned long long)n_help);
        if (n_again) atomicAdd(&ctr->tail_again, (unsigned long long)n_again);
    }
#ifdef CRAY_TRACE_DIAG
    if (lane == 0)
        for (int k = 0; k < 16; k++) atomicAdd(&ctr->diag[(ANY ? 16 : 0) + k], dg[k]);
#endif
#undef CRAY_PUSH
#undef CRAY_PUSH_H
#undef CRAY_PUSH_W
#undef CRAY_POP_W
#undef CRAY_DEEP_PUSH
#undef CRAY_DEEP_POP
#undef CRAY_LDS_REF
#undef CRAY_LDS_W0
#undef CRAY_LDS_W1
#undef CRAY_ANY_LANE
}

template <bool ANY, bool COUNT, int HYB, bool SHAPES_LDS = false, bool DEEP = false>
__global__ void CRAY_TRACE_EU __launch_bounds__(kBlock, trace_waves(HYB)) k_trace(DevScene sc, PathState ps, const uint32_t* __restrict__ queue,
                                                  const unsigned int* __restrict__ n_ptr, uint32_t n_fixed,
                                                  const double* __restrict__ closest_tmax, Counters* ctr, unsigned int* work_head, unsigned int refill_min) {
    trace_body<ANY ? kTraceAny : kTraceClosest, COUNT, HYB, SHAPES_LDS, false, DEEP>(sc, ps_kernarg(), queue, n_ptr ? *n_ptr : n_fixed, nullptr, 0u, closest_tmax, ctr, work_head, refill_min);
}

// shadow rays of one bounce (any_queue) + path segments of the next (closest_queue) in one persistent launch
template <int HYB, bool SHAPES_LDS = false, bool TAIL = false, bool DEEP = false>
__global__ void CRAY_TRACE_EU __launch_bounds__(kBlock, trace_waves(HYB)) k_trace_mixed(DevScene sc, PathState ps, const uint32_t* __restrict__ any_queue,
                                                  const unsigned int* __restrict__ n_any_ptr, const uint32_t* __restrict__ closest_queue,
                                                  const unsigned int* __restrict__ n_closest_ptr, Counters* ctr, unsigned int* work_head,
                                                  unsigned int refill_min) {
    trace_body<kTraceMixed, false, HYB, SHAPES_LDS, TAIL, DEEP>(sc, ps_kernarg(), any_queue, *n_any_ptr, closest_queue, *n_closest_ptr, nullptr, ctr, work_head, refill_min);
}

// ---------------------------------------------------------------------------------------------------------------
// "Fast" traversal: f32 node and triangle records (64-B / 48-B, four loads per visit instead of seven), f32 slab and
// Moller-Trumbore arithmetic.  NOT the reference's arithmetic: hits near silhouettes and shadow-ray leaks fall differently,
// so films differ from the f64 path by a small RMSE that shrinks with the sample count (reported separately).
// Same tree, same traversal order and the same persistent per-lane state machine as trace_body.  Spheres and disks (the
// r = 1e5 ground sphere cancels catastrophically in f32) are still tested in f64.  A ray never tests the triangle it starts
// on (`skip`): f32 hit points lie ~1e-5 off their surface, which the reference's 1e-9 epsilon cannot absorb.
// ---------------------------------------------------------------------------------------------------------------
#ifndef CRAY_TRACE32_WAVES
#define CRAY_TRACE32_WAVES 4
#endif
__device__ __forceinline__ float child_key32(const float* lo, const float* hi, const float o[3], const float rd[3], float t_lo) {
    float tmin = -__builtin_huge_valf(), tmax = __builtin_huge_valf();
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float t0 = (lo[a] - o[a]) * rd[a], t1 = (hi[a] - o[a]) * rd[a];
        tmin = fmaxf(tmin, fminf(t0, t1));   // fminf / fmaxf drop a NaN (0 * inf: origin on a slab of a parallel ray)
        tmax = fminf(tmax, fmaxf(t0, t1));
    }
    tmax *= 1.0000004f;                      // 2 ulp: keep the test conservative under f32 rounding
    return (tmin <= tmax && tmax > t_lo) ? tmin : __builtin_huge_valf();   // accepted iff key < ray.tmax
}

template <int MODE>
__device__ __forceinline__ void trace_body32(const DevScene& sc, const PathState& ps, const uint32_t* __restrict__ queue, const uint32_t n_first,
                                             const uint32_t* __restrict__ queue_b, const uint32_t n_b, Counters* ctr, unsigned int* work_head,
                                             unsigned int refill_min, uint32_t first_bounce) {
    const uint32_t n = n_first + n_b;
    bool is_any = MODE == kTraceAny;
#define CRAY_ANY_LANE (MODE == kTraceMixed ? is_any : (MODE == kTraceAny))
    const unsigned int lane = __lane_id(), tid = threadIdx.x;
    unsigned int overflow = 0;
    __shared__ uint32_t lds_ref[kLdsStack * kBlock];
    __shared__ float lds_key[kLdsStack * kBlock];
    uint32_t sref[kStackDepth - kLdsStack];
    float skey[kStackDepth - kLdsStack];
    int sp = 0;
    bool active = false, exhausted = false, pending = false;
    unsigned int res_base = 0, res_left = 0;
    unsigned int chunk = n / (gridDim.x * (kBlock / 64) * 4u);
    chunk = chunk < 64u ? 64u : (chunk > 512u ? 512u : chunk);
    uint32_t p = 0, cur = 0;
    int32_t skip = -1;                  // the triangle this ray starts on
    ray_t ray = mkray(mk(0, 0, 0), mk(0, 0, 1));   // f64 ray: spheres / disks, and the exact t of their hits
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 1.f}, rd[3] = {0.f, 0.f, 1.f};
    float tmax = 0.f, t_lo = 0.f;       // f32 view of ray.tmax, and the near limit (relative to the origin's magnitude)
    double hit_t = 0.0, hit_u = 0.0, hit_v = 0.0;
    int32_t hit_prim = -1;

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const bool do_refill = (unsigned int)__popcll(idle) >= refill_min && !exhausted;
        if ((do_refill || idle == ~0ull) && pending) {
            if (CRAY_ANY_LANE) {
                const uint32_t l = ps.sp0[p];   // p is a shadow slot; L lives per original path
                ps.lr[l] = ps.lr[l] + ps.cr[p];
                ps.lg[l] = ps.lg[l] + ps.cg[p];
                ps.lb[l] = ps.lb[l] + ps.cb[p];
            } else {
                ps.ht[p] = hit_t; ps.hu[p] = hit_u; ps.hv[p] = hit_v; ps.hprim[p] = hit_prim;
            }
            pending = false;
        }
        if (do_refill) {
            if (res_left == 0) {
                const unsigned int leader = __ffsll((long long)idle) - 1;
                unsigned int base = 0;
                if (lane == leader) base = atomicAdd(work_head, chunk);
                base = __shfl(base, leader);
                if (base >= n) { exhausted = true; }
                else { res_base = base; res_left = n - base < chunk ? n - base : chunk; }
            }
            const unsigned int rank = (unsigned int)__popcll(idle & ((1ull << lane) - 1ull));
            const bool take = !active && rank < res_left;
            const unsigned int mine = res_base + rank;
            const unsigned int taken = (unsigned int)__popcll(__ballot(take));
            res_base += taken; res_left -= taken;
            if (take) {
                if (MODE == kTraceMixed) {
                    is_any = mine >= n_b;   // path segments first, shadow rays last (as in trace_body)
                    if (is_any) p = queue[mine - n_b];
                    else p = queue_b ? queue_b[mine] : mine;
                } else {
                    p = queue ? queue[mine] : mine;
                }
                if (CRAY_ANY_LANE) {
                    ray.o = mk(ps.sox[p], ps.soy[p], ps.soz[p]);
                    ray.d = mk(ps.sdx[p], ps.sdy[p], ps.sdz[p]);
                    ray.tmax = ps.stmax[p];
                } else {
                    ray.o = mk(ps.ox[p], ps.oy[p], ps.oz[p]);
                    ray.d = mk(ps.dx[p], ps.dy[p], ps.dz[p]);
                    ray.tmax = inf64();
                }
                skip = first_bounce ? -1 : (CRAY_ANY_LANE ? ps.sprim[p] : ps.hprim[p]);   // shadow rays and continued segments start on the last hit
                o[0] = (float)ray.o.x; o[1] = (float)ray.o.y; o[2] = (float)ray.o.z;
                d[0] = (float)ray.d.x; d[1] = (float)ray.d.y; d[2] = (float)ray.d.z;
                rd[0] = 1.0f / d[0]; rd[1] = 1.0f / d[1]; rd[2] = 1.0f / d[2];
                // a shadow ray ends 1e-9 before its light (light.rs:125-128): in f32 that margin must be relative
                tmax = CRAY_ANY_LANE ? __double2float_rd(ray.tmax) * 0.99998f : __builtin_huge_valf();
                t_lo = 1e-4f * fmaxf(1.0f, fmaxf(fabsf(o[0]), fmaxf(fabsf(o[1]), fabsf(o[2]))));
                hit_t = 0.0; hit_u = 0.0; hit_v = 0.0; hit_prim = -1;
                sp = 0;
                float rlo[3] = {__double2float_rd(sc.root_lo[0]), __double2float_rd(sc.root_lo[1]), __double2float_rd(sc.root_lo[2])};
                float rhi[3] = {__double2float_ru(sc.root_hi[0]), __double2float_ru(sc.root_hi[1]), __double2float_ru(sc.root_hi[2])};
                if (child_key32(rlo, rhi, o, rd, 0.0f) < tmax) { cur = sc.root_ref; active = true; }
                else pending = true;
            }
        }
        if (!__any(active)) {
            if (exhausted) break;
            continue;
        }
        bool need_pop = false, finished = false, occluded = false;
        // one record fetch per iteration: the interior node (64 B) or the first slot of the leaf (48 B), four 16-B loads
        const bool at_leaf = ref_is_leaf(cur);
        float4 r0, r1, r2, r3;
        r0 = r1 = r2 = r3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) {
            const float4* rec = at_leaf ? reinterpret_cast<const float4*>(sc.slots32 + ref_leaf_first(cur))
                                        : reinterpret_cast<const float4*>(sc.inner32 + cur);
            r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3];
        }
        if (active && !at_leaf) {
            const float lo0[3] = {r0.x, r0.y, r0.z}, hi0[3] = {r0.w, r1.x, r1.y};
            const float lo1[3] = {r1.z, r1.w, r2.x}, hi1[3] = {r2.y, r2.z, r2.w};
            const uint32_t ref0 = __float_as_uint(r3.x), ref1 = __float_as_uint(r3.y), axis = __float_as_uint(r3.z);
            const float k0 = child_key32(lo0, hi0, o, rd, 0.0f), k1 = child_key32(lo1, hi1, o, rd, 0.0f);
            const bool right_first = (axis == 0 ? d[0] : (axis == 1 ? d[1] : d[2])) < 0.0f;
            const uint32_t near = right_first ? ref1 : ref0, far = right_first ? ref0 : ref1;
            const float kn = right_first ? k1 : k0, kf = right_first ? k0 : k1;
            const bool an = kn < tmax, af = kf < tmax;
            if (an) {
                if (af) {
                    if (sp < kLdsStack) { lds_ref[sp * kBlock + tid] = far; lds_key[sp * kBlock + tid] = kf; sp++; }
                    else if (sp < kStackDepth) { sref[sp - kLdsStack] = far; skey[sp - kLdsStack] = kf; sp++; }
                    else overflow = 1;
                }
                cur = near;
            } else if (af) {
                cur = far;
            } else {
                need_pop = true;
            }
        } else if (active) {
            // one slot per iteration, as in trace_body: the remaining slots of a leaf are fetched by the next iterations' record fetch
            const uint32_t s_prim = __float_as_uint(r2.y), s_kind = __float_as_uint(r2.z);
            if (s_kind == CRAY_SHAPE_TRIANGLE) {
                // Moller-Trumbore as in shape.rs:216-262, in f32
                const float v0[3] = {r0.x, r0.y, r0.z}, e1[3] = {r0.w, r1.x, r1.y}, e2[3] = {r1.z, r1.w, r2.x};
                const float P[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
                const float det = P[0] * e1[0] + P[1] * e1[1] + P[2] * e1[2];
                if ((int32_t)s_prim != skip && det != 0.0f) {
                    const float inv = 1.0f / det;
                    const float T[3] = {o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]};
                    const float u = (P[0] * T[0] + P[1] * T[1] + P[2] * T[2]) * inv;
                    const float Q[3] = {T[1] * e1[2] - T[2] * e1[1], T[2] * e1[0] - T[0] * e1[2], T[0] * e1[1] - T[1] * e1[0]};
                    const float v = (Q[0] * d[0] + Q[1] * d[1] + Q[2] * d[2]) * inv;
                    const float t = (Q[0] * e2[0] + Q[1] * e2[1] + Q[2] * e2[2]) * inv;
                    if (u >= 0.0f && u <= 1.0f && v >= 0.0f && u + v <= 1.0f && t > t_lo && t < tmax) {
                        if (CRAY_ANY_LANE) occluded = true;
                        else {
                            tmax = t; ray.tmax = (double)t;
                            hit_t = (double)t; hit_u = (double)u; hit_v = (double)v; hit_prim = (int32_t)s_prim;
                        }
                    }
                }
            } else {
                const uint32_t shp = __float_as_uint(r2.w);   // LeafSlot32::shape
                const bool hit = s_kind == CRAY_SHAPE_SPHERE ? sphere_hit(sc.spheres[shp], ray, CRAY_ANY_LANE, nullptr)
                                                             : disk_hit(sc.disks[shp], ray, CRAY_ANY_LANE, nullptr);
                if (hit) {
                    if (CRAY_ANY_LANE) occluded = true;
                    else {
                        hit_t = ray.tmax; hit_prim = (int32_t)s_prim;
                        tmax = __double2float_ru(ray.tmax);
                    }
                }
            }
            if (CRAY_ANY_LANE && occluded) finished = true;
            else if (ref_leaf_count(cur) > 1) cur += 7u;
            else need_pop = true;
        }
        if (active && need_pop) {
            for (;;) {
                if (sp == 0) { finished = true; break; }
                --sp;
                float key;
                uint32_t ref;
                if (sp < kLdsStack) { key = lds_key[sp * kBlock + tid]; ref = lds_ref[sp * kBlock + tid]; }
                else { key = skey[sp - kLdsStack]; ref = sref[sp - kLdsStack]; }
                if (key < tmax) { cur = ref; break; }
            }
        }
        if (active && finished) {
            pending = CRAY_ANY_LANE ? !occluded : true;
            active = false;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (MODE == kTraceMixed) {
            atomicAdd(&ctr->shadow_rays, (unsigned long long)n_first);
            atomicAdd(&ctr->closest_rays, (unsigned long long)n_b);
        } else {
            atomicAdd(MODE == kTraceAny ? &ctr->shadow_rays : &ctr->closest_rays, (unsigned long long)n);
        }
    }
    if (overflow) atomicAdd(&ctr->stack_overflow, 1ull);
#undef CRAY_ANY_LANE
}

template <int MODE>
__global__ void __launch_bounds__(kBlock, CRAY_TRACE32_WAVES) k_trace32(DevScene sc, PathState ps, const uint32_t* __restrict__ queue_a, const unsigned int* __restrict__ n_a_ptr,
                                                                        uint32_t n_a_fixed, const uint32_t* __restrict__ queue_b, const unsigned int* __restrict__ n_b_ptr,
                                                                        Counters* ctr, unsigned int* work_head, unsigned int refill_min, uint32_t first_bounce) {
    const uint32_t n_a = n_a_ptr ? *n_a_ptr : n_a_fixed;
    trace_body32<MODE>(sc, ps, queue_a, n_a, queue_b, (MODE == kTraceMixed && n_b_ptr) ? *n_b_ptr : 0u, ctr, work_head, refill_min, first_bounce);
}

// f64 device layout -> the f32 records of the fast mode (bounds rounded outward, triangles to nearest)
__global__ void __launch_bounds__(kBlock) k_make_inner32(const InnerNode* __restrict__ in, uint32_t n, InnerNode32* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const InnerNode a = in[i];
    InnerNode32 o;
    for (int k = 0; k < 3; k++) {
        o.lo0[k] = __double2float_rd(a.lo0[k]); o.hi0[k] = __double2float_ru(a.hi0[k]);
        o.lo1[k] = __double2float_rd(a.lo1[k]); o.hi1[k] = __double2float_ru(a.hi1[k]);
    }
    o.ref0 = a.ref0; o.ref1 = a.ref1; o.axis = a.axis; o.pad_ = 0;
    out[i] = o;
}
// The arena of the certified f32 culling (cray_device.h): InnerNodeH records with references in the arena's encoding; the leaf
// slots behind them are copied from slots[] as they are.  Derived on the device from the f64 layout.
__host__ __device__ __forceinline__ uint32_t href_of(uint32_t ref, uint32_t leaf_base) {
    if (!ref_is_leaf(ref)) return ref * (uint32_t)sizeof(InnerNodeH);
    return (leaf_base + ref_leaf_first(ref) * (uint32_t)sizeof(LeafSlot)) | kHLeaf | (ref_leaf_count(ref) - 1u);
}
__global__ void __launch_bounds__(kBlock) k_make_innerh(const InnerNode* __restrict__ in, uint32_t n, InnerNodeH* __restrict__ out, uint32_t leaf_base) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const InnerNode a = in[i];
    InnerNodeH o;
    for (int k = 0; k < 3; k++) {
        o.lo[k][0] = f32_down(a.lo0[k]); o.hi[k][0] = f32_up(a.hi0[k]);
        o.lo[k][1] = f32_down(a.lo1[k]); o.hi[k][1] = f32_up(a.hi1[k]);
    }
    o.ref0 = href_of(a.ref0, leaf_base); o.ref1 = href_of(a.ref1, leaf_base); o.axis = a.axis; o.pad_ = 0;
    out[i] = o;
}
__global__ void __launch_bounds__(kBlock) k_make_slots32(const LeafSlot* __restrict__ in, uint32_t n, LeafSlot32* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const LeafSlot a = in[i];
    LeafSlot32 o;
    for (int k = 0; k < 3; k++) { o.v0[k] = (float)a.v0[k]; o.e1[k] = (float)a.e1[k]; o.e2[k] = (float)a.e2[k]; }
    o.prim = a.prim; o.kind = a.kind;
    o.shape = a.kind == CRAY_SHAPE_TRIANGLE ? 0u : (uint32_t)__double2loint(a.v0[0]);   // sphere / disk index (LeafSlot::v0[0] carries it as bits)
    if (a.kind != CRAY_SHAPE_TRIANGLE) o.v0[0] = 0.f;
    out[i] = o;
}

// ---------------------------------------------------------------------------------------------------------------
// Tile cost probe (round 4): the bounce-0 launch ends when its slowest pixels end, and a pixel whose camera rays graze the
// mesh needs ten times the node visits of one that looks at the sky.  Started last, such a pixel is the tail of the launch
// (1.0-1.8 ms of an eighth of configs[2]); started first it is hidden behind everything else.  One wave per tile of a rank's
// list walks 64 probe rays (an 8 x 8 grid of pixel centres) through the tree the way a camera ray would and reports
// max << 32 | sum of the probes' node visits + primitive tests; the host orders the rank's tiles by it, most expensive
// first (ensure_tile_order).  A scheduling hint only: which order the pixels are rendered in enters no result — the probe
// need not (and does not) reproduce the reference's traversal bit for bit, it uses the plain slab test and a 64-entry stack.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_tile_probe(DevScene sc, const uint32_t* __restrict__ rect /* [n][4]: x0, y0, w, h */, uint32_t n_tiles,
                                                   unsigned long long* __restrict__ cost) {
    const uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    const uint32_t lane = threadIdx.x, gx = lane & 7u, gy = lane >> 3;
    const uint32_t x0 = rect[4 * t], y0 = rect[4 * t + 1], w = rect[4 * t + 2], h = rect[4 * t + 3];
    uint32_t px = x0 + ((2u * gx + 1u) * w) / 16u, py = y0 + ((2u * gy + 1u) * h) / 16u;
    px = px < x0 + w ? px : x0 + w - 1u; py = py < y0 + h ? py : y0 + h - 1u;
    ray_t ray = camera_ray(sc, 0.5, 0.5, 0.5, 0.5, px, py);
    uint32_t stack[64];
    double keys[64];
    int sp = 0;
    uint32_t cur = sc.root_ref, visits = 0;
    bool go = child_key(sc.root_lo, sc.root_hi, ray.o, ray.d) < ray.tmax;
    while (go) {
        bool pop = true;
        if (!ref_is_leaf(cur)) {
            const InnerNode& nd = sc.inner[cur];
            visits++;
            const double k0 = child_key(nd.lo0, nd.hi0, ray.o, ray.d), k1 = child_key(nd.lo1, nd.hi1, ray.o, ray.d);
            const bool right_first = comp(ray.d, (int)nd.axis) < 0.0;
            const uint32_t near = right_first ? nd.ref1 : nd.ref0, far = right_first ? nd.ref0 : nd.ref1;
            const double kn = right_first ? k1 : k0, kf = right_first ? k0 : k1;
            const bool an = kn < ray.tmax, af = kf < ray.tmax;
            if (an) {
                if (af && sp < 64) { stack[sp] = far; keys[sp] = kf; sp++; }
                cur = near; pop = false;
            } else if (af) { cur = far; pop = false; }
        } else {
            const uint32_t first = ref_leaf_first(cur), cnt = ref_leaf_count(cur);
            for (uint32_t i = 0; i < cnt; i++) {
                const LeafSlot& sl = sc.slots[first + i];
                visits++;
                if (sl.kind == CRAY_SHAPE_TRIANGLE) {
                    double tt, uu, vv;
                    if (tri_test(mk(sl.v0[0], sl.v0[1], sl.v0[2]), mk(sl.e1[0], sl.e1[1], sl.e1[2]), mk(sl.e2[0], sl.e2[1], sl.e2[2]), ray, tt, uu, vv)) ray.tmax = tt;
                } else {
                    const uint32_t shp = (uint32_t)__double2loint(sl.v0[0]);
                    if (sl.kind == CRAY_SHAPE_SPHERE) (void)sphere_hit(sc.spheres[shp], ray, false, nullptr);
                    else (void)disk_hit(sc.disks[shp], ray, false, nullptr);
                }
            }
        }
        if (pop) {
            for (;;) {
                if (sp == 0) { go = false; break; }
                --sp;
                if (keys[sp] < ray.tmax) { cur = stack[sp]; break; }
            }
        }
    }
    uint32_t mx = visits, sum = visits;
    for (int o = 32; o; o >>= 1) {
        const uint32_t a = (uint32_t)__shfl_xor((int)mx, o), b = (uint32_t)__shfl_xor((int)sum, o);
        mx = a > mx ? a : mx; sum += b;
    }
    if (lane == 0) cost[t] = ((unsigned long long)mx << 32) | (unsigned long long)sum;
}

// render_pixel up to the camera ray (craytracer.rs:148-156) for every path of a pass.
// path p -> pixel pix_list[px0 + p / spp_pass], sample s_lo + p % spp_pass.  (Sample-major order was
// measured: k_film gets trivially coalesced, but the traversal and k_shade lose the coherence of the 16
// samples of one pixel sitting in adjacent lanes: +13 ms closest, +8 ms shade per frame.)
__global__ void __launch_bounds__(kBlock) k_raygen(DevScene sc, PathState ps, const uint32_t* __restrict__ pix_list, uint32_t px0,
                                                   uint32_t n_paths, uint32_t spp_pass, uint32_t s_lo, uint64_t seed, uint32_t uni_nx, uint32_t uni_ny,
                                                   uint32_t independent) {
    // Sobol set 0 (dimensions 0-3) folded into nibble tables once per block, as in k_shade (cray_shading.h sobol4_lut)
    __shared__ uint2 l_sob[64];
    if (threadIdx.x < 64u) sobol_fill_lut(sc.sobol, 0u, threadIdx.x, l_sob);
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_paths; p += stride) {
        const uint32_t pix = pix_list[px0 + p / spp_pass];
        const uint32_t s = s_lo + p % spp_pass;
        const uint32_t x = pix % sc.film_w, y = pix / sc.film_w;
        const uint32_t h = pixel_seed(seed, x, y);  // SobolSampler::start_pixel
        double u[4];
        if (independent) {  // IndependentSampler (sampling.rs:102-146): the first four draws of the pixel sample's own generator
            uint32_t key[8], w[16];
            indep_key(indep_pixel_hash(seed, x, y, s), key);
            chacha_block(key, 0u, 0u, 0u, 0u, 6, w);
            for (int j = 0; j < 4; j++) u[j] = indep_unit(w[2 * j], w[2 * j + 1]);
        } else if (uni_nx) {  // UniformSampler::sample_2d (sampling.rs:185-192): the centre of slot (s % nx, s / nx), for film and lens alike
            u[0] = u[2] = ((double)(s % uni_nx) + 0.5) / (double)uni_nx;
            u[1] = u[3] = ((double)(s / uni_nx) + 0.5) / (double)uni_ny;
        } else {
            sobol4_lut(l_sob, s, 0, h, u);  // dims 0,1 film; 2,3 lens (always drawn, craytracer.rs:153-154)
        }
        ray_t r = camera_ray(sc, u[0], u[1], u[2], u[3], x, y);
        ps.ox[p] = r.o.x; ps.oy[p] = r.o.y; ps.oz[p] = r.o.z;
        ps.dx[p] = r.d.x; ps.dy[p] = r.d.y; ps.dz[p] = r.d.z;
        // beta = WHITE, prev_bsdf_pdf = 0 and is_specular_bounce = true (path_integrator.rs:44-52) are not stored:
        // k_shade knows them for bounce 0
        ps.lr[p] = 0.0; ps.lg[p] = 0.0; ps.lb[p] = 0.0;
        ps.hash[p] = h;
    }
}

// Wave-aggregated reservation of one position per CALLING lane in a block-level counter (LDS): one atomic per wave
// (ballot of the lanes that are here + popcount + lane prefix).  Called from divergent code by exactly the lanes that append.
__device__ __forceinline__ uint32_t lds_reserve(unsigned int* count) {
    const unsigned long long mask = __ballot(1);
    const unsigned int lane = __lane_id();
    const unsigned int leader = __ffsll((long long)mask) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned int)__popcll(mask));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// (A material sort — per-class queues filled by a counting sort before k_shade — was measured in round 1 and made k_shade
// 1.6-2x slower on all three configs: shading is bound by path-state traffic, which class queues scatter.  The code is gone;
// profiles/r01_experiments.md has the numbers.)

// The body of one estimate_Li iteration between the two BVH queries (path_integrator.rs:56-211).
#ifndef CRAY_SHADE_WAVES
#define CRAY_SHADE_WAVES 2
#endif
// MODE bit 0: simple_integrator::estimate_Li (src/simple_integrator.rs:36-143: no MIS, no roulette, seven samples per segment)
//      bit 1: UniformSampler (src/sampling.rs:154-194) instead of SobolSampler; uni_nx / uni_ny are its two slot counts.
// The reference's `main` runs mode 0 (path integrator + Sobol, craytracer.rs:159-160, 361); the others are its selectable
// alternatives and run on the all-features instantiation only.
enum { kModeSimple = 1, kModeUniform = 2, kModeLdsTables = 4, kModeIndependent = 8 };
// Waves per SIMD the instantiation is compiled for.  Two everywhere (the double-double sin / cos and the emitters' 4x4 transforms
// keep the live set above 168 registers) except the textured instantiations of the default mode — the staircase.cry class — whose
// dependent texel fetches gain more from a third wave than its 32 spilled registers cost (round 3, after the Sobol nibble tables
// freed ~40 registers: k_shade 190 -> 177 ms at configs[3]; the matte / conductor instantiations fit 168 without a spill and do not gain).
constexpr int shade_waves(uint32_t F, int MODE) {
#ifdef CRAY_SHADE_WAVES_FORCE
    return CRAY_SHADE_WAVES_FORCE;
#else
    return ((MODE & ~kModeLdsTables) == 0 && CRAY_HAS(F, SF_TEX_IMAGE) && F != SF_ALL) ? 3 : CRAY_SHADE_WAVES;
#endif
}
template <uint32_t F, int MODE = 0>
__global__ void __launch_bounds__(kBlock, shade_waves(F, MODE)) k_shade(DevScene sc, PathState ps, PathState po, const uint32_t* __restrict__ queue,
                                                  const unsigned int* __restrict__ n_ptr, uint32_t n_fixed, uint32_t bounce,
                                                  uint32_t spp_pass, uint32_t s_lo, uint32_t* __restrict__ next_queue,
                                                  unsigned int* next_count, uint32_t* __restrict__ shadow_queue,
                                                  unsigned int* shadow_count, Counters* ctr, uint32_t trace_all_shadow,
                                                  uint32_t uni_nx, uint32_t uni_ny, const uint32_t* __restrict__ pix_list, uint32_t px0, uint64_t seed) {
    constexpr bool kSimple = (MODE & kModeSimple) != 0, kUniform = (MODE & kModeUniform) != 0, kIndependent = (MODE & kModeIndependent) != 0;
    const uint32_t n = n_ptr ? *n_ptr : n_fixed;
    // (the two path-state views — sixty array pointers — are read from the kernel-argument segment where a path uses them, as
    // in trace_body: held in scalar registers across the tile loop they were spilled into vector-register lanes, 343 v_readlane
    // and 222 v_writelane in this kernel)
    const PsArg ps_arg = ps_kernarg();
    const PsArg po_arg = (PsArg)((const char __attribute__((address_space(4)))*)ps_arg + sizeof(PathState));
    // `ps` is the state of this bounce (live slots named by `queue`, or the identity at bounce 0), `po` the view the survivors
    // are written to: the OTHER live buffer, the shadow buffer and L (cray_device.h).  A block walks tiles of `tsz` queue
    // positions; a survivor's new live slot is tile * tsz + its rank among the tile's survivors (one LDS atomic per wave), its
    // shadow ray's slot likewise — so what a tile leaves behind is contiguous whatever the survival rate, the next
    // kernels read whole lines again, and the queue entries of a tile are consecutive integers.  The queues themselves
    // are appended with ONE global atomic per queue and tile (one atomic per wave on a device-wide counter, ~88 returning
    // atomics/us per word, was half of this kernel's time in round 1).
    __shared__ unsigned int c_shadow, c_next, c_skip, c_hit, g_shadow, g_next;
    // The small shading tables (materials -> lobes -> textures, lights and their CDF) are walked by DEPENDENT loads: five to eight
    // round trips per path, each a trip to L2 for a wave that has only one other wave to hide behind (2 waves / SIMD).  When the
    // runtime found that they fit, every block copies them into LDS once and the walk reads LDS (through generic pointers).
    // (its own instantiation, MODE & kModeLdsTables: generic pointers make every table access a flat load, which costs the scenes
    // whose tables do not fit 7 % of this kernel)
    constexpr bool kLdsTables = (MODE & kModeLdsTables) != 0;
    __shared__ alignas(16) unsigned char l_tab[kLdsTables ? kShadeLdsTables : 16u];
    if (kLdsTables) {
        uint32_t off = 0;
        auto stage = [&](const void* src, uint32_t bytes) -> const void* {
            const uint32_t* s4 = static_cast<const uint32_t*>(src);
            uint32_t* d4 = reinterpret_cast<uint32_t*>(l_tab + off);
            for (uint32_t w = threadIdx.x; w < bytes / 4; w += kBlock) d4[w] = s4[w];
            const void* at = l_tab + off;
            off += (bytes + 15u) & ~15u;
            return at;
        };
        const uint32_t nl = sc.n_lights;
        sc.materials = static_cast<const cray_material*>(stage(sc.materials, sc.n_materials * (uint32_t)sizeof(cray_material)));
        sc.bxdfs = static_cast<const cray_bxdf*>(stage(sc.bxdfs, sc.n_bxdfs * (uint32_t)sizeof(cray_bxdf)));
        sc.textures = static_cast<const cray_texture*>(stage(sc.textures, sc.n_textures * (uint32_t)sizeof(cray_texture)));
        sc.lights = static_cast<const DevLight*>(stage(sc.lights, nl * (uint32_t)sizeof(DevLight)));
        sc.light_cdf = static_cast<const double*>(stage(sc.light_cdf, nl * 8u));
        sc.first_equal_light = static_cast<const int32_t*>(stage(sc.first_equal_light, nl * 4u));
        if (CRAY_HAS(F, SF_HIT_SPHERE | SF_HIT_DISK | SF_AREA_SPHERE | SF_AREA_DISK) && sc.shade_stage_shapes) {   // hit / emitter transforms
            sc.spheres = static_cast<const cray_xf_shape*>(stage(sc.spheres, sc.n_spheres * (uint32_t)sizeof(cray_xf_shape)));
            sc.disks = static_cast<const cray_xf_shape*>(stage(sc.disks, sc.n_disks * (uint32_t)sizeof(cray_xf_shape)));
        }
        if (CRAY_HAS(F, SF_TEX_IMAGE)) {   // image descriptors and the 8-bit -> linear table sit behind the texel fetch
            sc.images = static_cast<const cray_image*>(stage(sc.images, sc.n_images * (uint32_t)sizeof(cray_image)));
            sc.gamma_lut = static_cast<const double*>(stage(sc.gamma_lut, 256u * 8u));
        }
        __syncthreads();
    }
    // the two Sobol sets this bounce draws from (dimensions 4 + 8 b .. 11 + 8 b), folded into nibble tables once per block
    constexpr bool kSobolLut = !kSimple && !kUniform && !kIndependent;
    __shared__ uint2 l_sob[kSobolLut ? 128 : 1];
    if (kSobolLut) {
        if (threadIdx.x < 128u) sobol_fill_lut(sc.sobol, 1u + 2u * bounce + (threadIdx.x >> 6), threadIdx.x & 63u, l_sob + (threadIdx.x & 64u));
        __syncthreads();
    }
    // tile size: kShadeTile for the big launches, smaller (down to one pass of the block) when the launch would otherwise
    // leave blocks without a tile — the late bounces of a frame, every bounce of an eighth of a frame
    uint32_t per_tile = n / (gridDim.x * kBlock * 2u);
    per_tile = per_tile < 1u ? 1u : (per_tile > kShadeTile / kBlock ? kShadeTile / kBlock : per_tile);
    const uint32_t tsz = per_tile * kBlock;
    const uint32_t n_tiles = (n + tsz - 1) / tsz;
    // Tiles are handed out dynamically (the first one per block is its index, the rest come from Counters::shade_head): the grid
    // is exactly what the chip holds at this instantiation's occupancy, so no block waits for a slot while others run a round ahead
    // (a grid of 4 blocks per CU at 3 resident ones cost 15 % of this kernel), and a block that drew cheap tiles draws more of them.
    __shared__ unsigned int s_tile;
    for (uint32_t tile = blockIdx.x; tile < n_tiles;) {
      if (threadIdx.x == 0) { c_shadow = 0; c_next = 0; c_skip = 0; c_hit = 0; }
      __syncthreads();
      const uint32_t tile_base = tile * tsz;
      for (uint32_t k = 0; k < per_tile; k++) {
        const uint32_t i = tile_base + k * kBlock + threadIdx.x;
        bool skip_shadow = false, was_hit = false;
        if (i < n) {
            const uint32_t p = queue ? queue[i] : i;        // live slot of this bounce
            const PsArg psk = ps_here(ps_arg);
            const uint32_t p0 = queue ? psk->p0[p] : i;       // original path: L, the Sobol sample index, the pixel
            // L is read and written only by the paths that add emission in this bounce (a light was hit or the ray
            // escaped to an Infinite light); for all others the 48 B of traffic per path are skipped
            rgb L = mkc(0, 0, 0);
            bool L_loaded = false;
            auto add_L = [&](rgb term) {
                if (!L_loaded) { const PsArg pok = ps_here(po_arg); L = mkc(pok->lr[p0], pok->lg[p0], pok->lb[p0]); L_loaded = true; }
                L = L + term;
            };
            const int32_t hp = psk->hprim[p];
            // path state is loaded where it is first needed: an escaped path needs none of it unless the scene has an
            // Infinite light, prev_pdf / the specular flag only matter where emission is weighted
            rgb beta = mkc(0, 0, 0);
            double prev_pdf = 0.0;
            bool specular_bounce = false;

            if (hp < 0) {
                // escaped: Light::Le is non-black only for Infinite lights (light.rs:161-168)
                bool state_loaded = false;
                for (uint32_t li = 0; CRAY_HAS(F, SF_LIGHT_INFINITE) && li < sc.n_lights; li++) {
                    const DevLight& l = sc.lights[li];
                    if (l.kind != CRAY_LIGHT_INFINITE) continue;
                    if (!state_loaded) {
                        beta = bounce == 0 ? mkc(1, 1, 1) : mkc(psk->br[p], psk->bg[p], psk->bb[p]);
                        prev_pdf = bounce == 0 ? 0.0 : psk->prev_pdf[p];
                        specular_bounce = bounce == 0 || (psk->flags[p] & 1u) != 0;
                        state_loaded = true;
                    }
                    rgb Le = mkc(l.c[0], l.c[1], l.c[2]);
                    if (specular_bounce) {
                        add_L(beta * Le);  // :64-67
                    } else if (!kSimple && !black(Le)) {  // :68-88; pdf_Li of Infinite = 1/(4 pi) (light.rs:140); simple_integrator.rs:57-61 adds nothing
                        double light_pdf = (kInvPi / 4.0) * light_select_pdf(sc, li);
                        double w = power_heuristic(light_pdf, prev_pdf);
                        add_L(beta * Le * w);
                    }
                }
                if (L_loaded) { const PsArg pok = ps_here(po_arg); pok->lr[p0] = L.r; pok->lg[p0] = L.g; pok->lb[p0] = L.b; }
            } else {
                was_hit = true;
                const ray_t ray = mkray(mk(psk->ox[p], psk->oy[p], psk->oz[p]), mk(psk->dx[p], psk->dy[p], psk->dz[p]));
                const vec3 w_o = flip(ray.d);
                beta = bounce == 0 ? mkc(1, 1, 1) : mkc(psk->br[p], psk->bg[p], psk->bb[p]);  // camera rays: beta = WHITE
                const cray_prim pr = sc.prims[hp];
                const int32_t mat = pr.light >= 0 ? -1 : pr.material;
                const bool need_uv = CRAY_HAS(F, SF_TEX_CHECKER | SF_TEX_IMAGE) && mat >= 0 && sc.materials[mat].pad_ != 0;
                const SurfPoint sp = surface_at<F>(sc, pr, ray, psk->ht[p], psk->hu[p], psk->hv[p], need_uv);
                const vec3 n_s = sp.normal, x = sp.location;

                // PathSegmentSamples::from (path_integrator.rs:26-36): dims 4+8k .. 11+8k
                const uint32_t sidx = s_lo + p0 % spp_pass;
                const uint32_t h = psk->hash[p];
                double sa[4], sb[4];
                if (kIndependent) {  // IndependentSampler: draws 4 + 8 b .. (4 + 7 b .. for simple_integrator) of the pixel sample's generator
                    const uint32_t pix = pix_list[px0 + p0 / spp_pass];
                    uint32_t key[8];
                    indep_key(indep_pixel_hash(seed, pix % sc.film_w, pix / sc.film_w, sidx), key);
                    double dr[8];
                    indep_draws(key, 4u + (kSimple ? 7u : 8u) * bounce, dr);
                    sa[0] = dr[0]; sa[1] = dr[1]; sa[2] = dr[2]; sa[3] = dr[3];
                    sb[0] = dr[4]; sb[1] = dr[5]; sb[2] = dr[6]; sb[3] = kSimple ? 0.0 : dr[7];
                } else if (kUniform) {  // every 1-D draw is (s + 0.5) / (nx ny), every 2-D draw the slot centre (sampling.rs:180-192)
                    const double u1 = ((double)sidx + 0.5) / (double)(uni_nx * uni_ny);
                    const double ux = ((double)(sidx % uni_nx) + 0.5) / (double)uni_nx, uy = ((double)(sidx / uni_nx) + 0.5) / (double)uni_ny;
                    sa[0] = u1; sa[1] = ux; sa[2] = uy; sa[3] = u1;
                    sb[0] = u1; sb[1] = ux; sb[2] = uy; sb[3] = u1;
                } else if (kSimple) {  // seven draws per segment: dimensions 4 + 7 b .. 10 + 7 b straddle the 4-D sets
                    const uint32_t first = 4 + 7 * bounce, off = first & 3;   // uniform over the launch
                    double flat[12];                                         // the (at most) three 4-D sets the seven draws touch
                    sobol4(sc.sobol, sidx, (first >> 2), h, flat);
                    sobol4(sc.sobol, sidx, (first >> 2) + 1, h, flat + 4);
                    if (off + 7 > 8) sobol4(sc.sobol, sidx, (first >> 2) + 2, h, flat + 8);
                    else { flat[8] = flat[9] = flat[10] = flat[11] = 0.0; }
                    double v7[7];
#pragma unroll
                    for (uint32_t j = 0; j < 7; j++) {
                        double pick = flat[j];
#pragma unroll
                        for (uint32_t o = 1; o < 4; o++) pick = off == o ? flat[j + o] : pick;
                        v7[j] = pick;
                    }
                    sa[0] = v7[0]; sa[1] = v7[1]; sa[2] = v7[2]; sa[3] = v7[3];
                    sb[0] = v7[4]; sb[1] = v7[5]; sb[2] = v7[6]; sb[3] = 0.0;
                } else {
                    sobol4_lut(l_sob, sidx, 1 + 2 * bounce, h, sa);       // material 1D, material 2D, light index
                    sobol4_lut(l_sob + 64, sidx, 2 + 2 * bounce, h, sb);  // light 1D, light 2D, roulette
                }

                // emission at the hit (:106-126)
                if (pr.light >= 0) {
                    const DevLight& l = sc.lights[pr.light];
                    rgb Le = mkc(l.c[0], l.c[1], l.c[2]);
                    if (!black(Le)) {
                        specular_bounce = bounce == 0 || (psk->flags[p] & 1u) != 0;  // camera rays count as specular (:50)
                        if (specular_bounce) {
                            add_L(beta * Le);
                        } else if (!kSimple) {  // simple_integrator.rs:84-86 has no MIS branch
                            double lp = light_shape_pdf_from<F>(sc, l, x, n_s, w_o);
                            double light_pdf = lp * light_select_pdf(sc, (uint32_t)sc.first_equal_light[pr.light]);
                            prev_pdf = psk->prev_pdf[p];  // bounce > 0 here
                            double w = power_heuristic(light_pdf, prev_pdf);
                            add_L(beta * Le * w);
                        }
                    }
                }

                // next-event estimation (:129-164): build the shadow ray and the term it gates
                // A material without a diffuse lobe makes the gated term exactly zero: the light is not even
                // sampled (unless traversal is being counted).  Deviation from the reference only where it
                // would panic anyway: a non-finite Li / pdf factor times that zero (NaN) on an unoccluded ray.
                if (!kSimple && !trace_all_shadow && !material_has_diffuse_lobe(sc, mat)) {
                    skip_shadow = true;
                } else {
                    double sel_pdf;
                    const uint32_t li = light_select<F>(sc, sa[3], sel_pdf);
                    const DevLight& l = sc.lights[li];
                    vec3 w_i = mk(0, 0, 1);
                    rgb Li = mkc(l.c[0], l.c[1], l.c[2]);
                    double lpdf = 0.0, s_tmax = inf64();
                    bool delta = false;
                    constexpr uint32_t kArea = SF_AREA_TRI | SF_AREA_SPHERE | SF_AREA_DISK;
                    if (CRAY_HAS(F, SF_LIGHT_POINT) && (!CRAY_HAS(F, SF_LIGHT_DISTANT | SF_LIGHT_INFINITE | kArea) || l.kind == CRAY_LIGHT_POINT)) {  // light.rs:65-79
                        vec3 op = mk(l.v[0], l.v[1], l.v[2]) - x;
                        double d2 = len2(op);
                        double dist = sqrt(d2);
                        w_i = op / dist;
                        if (dist > kEps) s_tmax = dist;  // update_max_distance on a fresh ray
                        Li = Li / d2;
                        delta = true;
                    } else if (CRAY_HAS(F, SF_LIGHT_DISTANT) && (!CRAY_HAS(F, SF_LIGHT_INFINITE | kArea) || l.kind == CRAY_LIGHT_DISTANT)) {  // :80-96
                        w_i = mk(l.v[0], l.v[1], l.v[2]);
                        delta = true;
                    } else if (CRAY_HAS(F, SF_LIGHT_INFINITE) && (!CRAY_HAS(F, kArea) || l.kind == CRAY_LIGHT_INFINITE)) {  // :97-113
                        vec3 nn = sb[0] < 0.5 ? mk(1, 0, 0) : mk(-1, 0, 0);
                        vec3 r = sample_sphere(sb[1], sb[2]);
                        w_i = dot(r, nn) > 0.0 ? r : flip(r);  // sample_hemisphere
                        lpdf = kInvPi / 4.0;
                    } else if (CRAY_HAS(F, kArea)) {  // Area, :114-131 + Shape::sample_from (shape.rs:472-484)
                        vec3 pt = light_shape_sample<F>(sc, l, sb[1], sb[2]);
                        w_i = unit(pt - x);
                        lpdf = light_shape_pdf_from<F>(sc, l, x, n_s, w_i);
                        double dist = len(pt - x);
                        double tm = dist - kEps;
                        if (tm > kEps) s_tmax = tm;
                    }
                    rgb f = material_f<F>(sc, mat, w_o, w_i, n_s, sp.u, sp.v);
                    double cos_t = fabs(dot(w_i, n_s));
                    rgb contrib = mkc(0, 0, 0);
                    bool queried = true;  // does the reference call Scene::intersects for this segment at all?
                    if (kSimple) {
                        // simple_integrator.rs:102-111: Delta -> 1.0; the shadow ray is only cast when light_pdf > 0
                        // (`&&` short-circuit) and the term is beta * Li * f * cos / p_select / light_pdf, no MIS weight
                        const double light_pdf = delta ? 1.0 : lpdf;
                        queried = light_pdf > 0.0;
                        if (queried) contrib = beta * Li * f * cos_t / sel_pdf / light_pdf;
                    } else if (!delta) {
                        if (lpdf > 0.0) {
                            double light_pdf = lpdf * sel_pdf;
                            double bsdf_pdf = 0.0;
                            if (!material_pdf<F>(sc, mat, w_o, w_i, n_s, bsdf_pdf)) bsdf_pdf = 0.0;
                            double w = power_heuristic(light_pdf, bsdf_pdf);
                            contrib = beta * Li * f * cos_t * w / light_pdf;
                        }
                    } else {
                        contrib = beta * Li * f * cos_t / sel_pdf;
                    }
                    // The reference queries Scene::intersects unconditionally (:141).  When the term it
                    // gates is exactly zero (pdf 0, black f, light behind the surface) the answer cannot
                    // change L (L + 0 == L), so the query is skipped unless traversal is being counted.
                    const bool want_shadow = queried && (trace_all_shadow || !black(contrib));
                    skip_shadow = queried && !want_shadow;
                    if (want_shadow) {
                        const uint32_t q = tile_base + lds_reserve(&c_shadow);   // shadow slot: contiguous per tile
                        const PsArg pok = ps_here(po_arg);
                        pok->sox[q] = x.x; pok->soy[q] = x.y; pok->soz[q] = x.z;
                        pok->sdx[q] = w_i.x; pok->sdy[q] = w_i.y; pok->sdz[q] = w_i.z;
                        pok->stmax[q] = s_tmax;
                        pok->cr[q] = contrib.r; pok->cg[q] = contrib.g; pok->cb[q] = contrib.b;
                        pok->sp0[q] = p0; pok->sprim[q] = hp;
                    }
                }

                // BSDF sample, throughput update, roulette (:167-206)
                LobeSample ls;
                // an area light's own surface is a black Lambertian (primitive.rs:40-46): its sampled f is BLACK, the
                // path ends (`if f.is_black() { break }`, path_integrator.rs:176-178) whatever direction was drawn
                bool go = mat >= 0 && material_sample<F>(sc, mat, sa[0], sa[1], sa[2], w_o, n_s, sp.u, sp.v, ls);
                if (go && black(ls.f)) go = false;
                double bsdf_pdf = 0.0;
                if (go) {
                    bsdf_pdf = ls.delta ? 1.0 : ls.pdf;
                    if (bsdf_pdf == 0.0) go = false;
                }
                if (go) {
                    double cos_t = fabs(dot(ls.w_i, n_s));
                    beta = beta * ls.f * cos_t / bsdf_pdf;
                    if (bounce > 0 && !kSimple) {  // Russian roulette: path integrator only (path_integrator.rs:197-206)
                        double m = max_nn(beta.r, max_nn(beta.g, beta.b));
                        if (m < 1.0) {
                            double q = 1.0 - m;
                            if (sb[3] < q) go = false;
                            else beta = beta / (1.0 - q);
                        }
                    }
                }
                if (go) {
                    if (!finite3(beta)) atomicAdd(&ctr->nonfinite, 1ull);  // reference: assert!(beta.is_finite())
                    // the loop condition of the next iteration (:54)
                    go = (bounce + 1 < sc.max_depth) && !black(beta);
                }
                if (L_loaded) { const PsArg pok = ps_here(po_arg); pok->lr[p0] = L.r; pok->lg[p0] = L.g; pok->lb[p0] = L.b; }
                if (go) {
                    const uint32_t q = tile_base + lds_reserve(&c_next);   // live slot of the next bounce: contiguous per tile
                    const PsArg pok = ps_here(po_arg);
                    pok->ox[q] = x.x; pok->oy[q] = x.y; pok->oz[q] = x.z;  // Ray::new(location, w_i): no offset
                    pok->dx[q] = ls.w_i.x; pok->dy[q] = ls.w_i.y; pok->dz[q] = ls.w_i.z;
                    pok->br[q] = beta.r; pok->bg[q] = beta.g; pok->bb[q] = beta.b;
                    pok->prev_pdf[q] = bsdf_pdf;
                    pok->flags[q] = ls.specular ? 1u : 0u;
                    pok->hash[q] = h; pok->p0[q] = p0; pok->hprim[q] = hp;
                }
            }
        }
        {
            const unsigned long long sk = __ballot(skip_shadow);
            if (sk != 0 && __lane_id() == (unsigned int)(__ffsll((long long)sk) - 1)) atomicAdd(&c_skip, (unsigned int)__popcll(sk));
            const unsigned long long hk = __ballot(was_hit);
            if (hk != 0 && __lane_id() == (unsigned int)(__ffsll((long long)hk) - 1)) atomicAdd(&c_hit, (unsigned int)__popcll(hk));
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
          g_shadow = c_shadow ? atomicAdd(shadow_count, c_shadow) : 0u;
          g_next = c_next ? atomicAdd(next_count, c_next) : 0u;
          if (c_skip) atomicAdd(&ctr->shadow_skipped, (unsigned long long)c_skip);
          if (c_hit) atomicAdd(&ctr->closest_hits, (unsigned long long)c_hit);
          s_tile = atomicAdd(&ctr->shade_head, 1u) + gridDim.x;
      }
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < c_shadow; j += kBlock) shadow_queue[g_shadow + j] = tile_base + j;
      for (uint32_t j = threadIdx.x; j < c_next; j += kBlock) next_queue[g_next + j] = tile_base + j;
      tile = s_tile;
      __syncthreads();
    }
}

// The instantiations of k_shade: cray_scene_upload computes the scene's feature mask (cray_shading.h) and the launch
// uses the variant with the fewest features among those that cover it; the last one covers everything.
constexpr uint32_t kHitAny = SF_HIT_TRI | SF_HIT_SPHERE | SF_HIT_DISK;
constexpr uint32_t kAreaAny = SF_AREA_TRI | SF_AREA_SPHERE | SF_AREA_DISK;
#ifdef CRAY_VARIANTS_OVERRIDE   // compile-time experiments: hipcc -DCRAY_VARIANTS_OVERRIDE=mask,mask,...
constexpr uint32_t kShadeVariants[] = {CRAY_VARIANTS_OVERRIDE, SF_ALL};
#else
constexpr uint32_t kShadeVariants[] = {
    // constant-textured matte surfaces under area lights (Cornell-box class)
    kHitAny | kAreaAny | SF_MANY_LIGHTS,
    // + conductors, one disk light (dragon.cry: metal mesh on a matte ground)
    kHitAny | SF_CONDUCTOR | SF_AREA_DISK,
    // constant-textured plastics / metals / matte under any number of area lights
    kHitAny | SF_OREN_NAYAR | SF_CONDUCTOR | SF_SPEC_BRDF | SF_MULTI_LOBE | kAreaAny | SF_MANY_LIGHTS,
    // every lobe and texture kind under point + area lights (staircase.cry class: OBJ/MTL interiors)
    kHitAny | SF_TEX_CHECKER | SF_TEX_IMAGE | SF_OREN_NAYAR | SF_CONDUCTOR | SF_SPEC_BRDF | SF_SPEC_BTDF | SF_FRESNEL_SPEC | SF_MULTI_LOBE |
        SF_LIGHT_POINT | SF_AREA_TRI | SF_AREA_DISK | SF_MANY_LIGHTS,
    SF_ALL,
};
#endif
constexpr int kNumShadeVariants = (int)(sizeof(kShadeVariants) / sizeof(kShadeVariants[0]));

// render_tile's accumulation (craytracer.rs:175-188): per pixel, each sample batch is summed in
// f64 in sample order, cast to f32 and added into the f32 film; batches in ascending order.
// One wave per block handles 64 consecutive pixels of the pass.  A pixel's samples are contiguous in the
// radiance arrays (path = pixel * spp_pass + sample), so a lane-per-pixel loop would read with a stride of
// spp_pass * 8 B; instead the wave stages kFilmChunk samples of its 64 pixels through LDS with whole-line
// loads (4 pixels x 128 B per load instruction) and each lane then sums ITS pixel's samples in ascending
// order — the same f64 batch sums and f32 adds as before.
constexpr uint32_t kFilmChunk = 16;
__global__ void __launch_bounds__(64) k_film(PathState ps, const uint32_t* __restrict__ pix_list, uint32_t px0, uint32_t n_pix,
                                             uint32_t spp_pass, uint32_t s_lo, uint32_t batch, float* __restrict__ film, Counters* ctr) {
    __shared__ double stage[64 * (kFilmChunk + 1)];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_groups = (n_pix + 63) / 64;
    for (uint32_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint32_t pl0 = grp * 64, pl = pl0 + lane;
        const bool live = pl < n_pix;
        const uint32_t pix = live ? pix_list[px0 + pl] : 0u;
        float f[3] = {0.f, 0.f, 0.f};
        if (live) { f[0] = film[3 * (size_t)pix]; f[1] = film[3 * (size_t)pix + 1]; f[2] = film[3 * (size_t)pix + 2]; }
        double acc[3] = {0.0, 0.0, 0.0};
        bool bad = false;
        for (uint32_t j0 = 0; j0 < spp_pass; j0 += kFilmChunk) {
            const uint32_t nj = spp_pass - j0 < kFilmChunk ? spp_pass - j0 : kFilmChunk;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const double* __restrict__ src = ch == 0 ? ps.lr : (ch == 1 ? ps.lg : ps.lb);
                __syncthreads();
                for (uint32_t i = 0; i < 64 * kFilmChunk / 64; i++) {
                    const uint32_t q = i * (64 / kFilmChunk) + lane / kFilmChunk, jj = lane % kFilmChunk;
                    if (pl0 + q < n_pix && jj < nj) stage[q * (kFilmChunk + 1) + jj] = src[(size_t)(pl0 + q) * spp_pass + j0 + jj];
                }
                __syncthreads();
                if (live) {
                    for (uint32_t jj = 0; jj < nj; jj++) {
                        const double v = stage[lane * (kFilmChunk + 1) + jj];
                        if (!isfinite(v)) bad = true;
                        acc[ch] += v;
                        const uint32_t j = j0 + jj, sidx = s_lo + j;
                        if (((sidx + 1) % batch) == 0 || j + 1 == spp_pass) { f[ch] += (float)acc[ch]; acc[ch] = 0.0; }
                    }
                }
            }
        }
        if (live) {
            film[3 * (size_t)pix] = f[0]; film[3 * (size_t)pix + 1] = f[1]; film[3 * (size_t)pix + 2] = f[2];
            if (bad) atomicAdd(&ctr->nonfinite, 1ull);
        }
    }
}

// on_finish (craytracer.rs:253-259): `*pixel /= num_samples as f32`
__global__ void __launch_bounds__(kBlock) k_resolve(const float* __restrict__ film, float* __restrict__ out, size_t n, float num_samples) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = film[i] / num_samples;
}

// test hook: fill cray_hit records after a closest-hit trace of externally supplied rays
__global__ void __launch_bounds__(kBlock) k_hit_records(DevScene sc, PathState ps, uint32_t n, cray_hit* __restrict__ hits) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        cray_hit h;
        h.hit = ps.hprim[p] >= 0 ? 1 : 0;
        h.prim = ps.hprim[p];
        h.t = ps.ht[p];
        for (int k = 0; k < 3; k++) { h.location[k] = 0.0; h.normal[k] = 0.0; }
        h.uv[0] = 0.0; h.uv[1] = 0.0;
        if (h.hit) {
            ray_t ray = mkray(mk(ps.ox[p], ps.oy[p], ps.oz[p]), mk(ps.dx[p], ps.dy[p], ps.dz[p]));
            SurfPoint sp = surface_at(sc, sc.prims[h.prim], ray, ps.ht[p], ps.hu[p], ps.hv[p], true);
            h.location[0] = sp.location.x; h.location[1] = sp.location.y; h.location[2] = sp.location.z;
            h.normal[0] = sp.normal.x; h.normal[1] = sp.normal.y; h.normal[2] = sp.normal.z;
            h.uv[0] = sp.u; h.uv[1] = sp.v;
        }
        hits[p] = h;
    }
}

}  // namespace cray
