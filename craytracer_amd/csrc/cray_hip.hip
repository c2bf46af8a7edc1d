// cray_hip.hip — runtime and C ABI (include/cray.h) of the MI355X render backend.
//
// Host side of the wavefront loop: uploads the flattened scene in the device layout,
// then for every pass (a block of pixels x sample batches that fits the path-state
// buffers) runs   raygen ; { trace_closest ; shade ; trace_any } x max_depth ; film
// on one HIP stream without any host round trip: queue lengths live in device memory and
// the kernels read them, so a whole frame is a fixed launch sequence.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl.so.1 is dlopen'ed by the first cray_comm_* call

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "cray_bvh_build.h"
#include "cray_kernels.h"
#include "sobol_rev_vectors.h"

namespace cray {

static thread_local char g_err[512] = "";
void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_last_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return CRAY_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace cray

using namespace cray;

struct cray_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 256;
    // idle lanes a wave waits for before it fetches new rays: the coherent camera rays of bounce 0 finish together (late refills
    // cost little and keep neighbouring pixels in one wave), the incoherent later bounces refill earlier (profiles/r02_experiments.md)
    unsigned int refill_min = 28, refill_min_b0 = 64, refill_min_any = 28;
    // the launches that read certified f32 culling refill a little earlier (their iteration is shorter, an idle lane costs relatively
    // more): 20 / 20 is 1 % faster than 28 / 28 on configs[2]; the f64 launches of configs[3] lose 2 % at 20 / 20
    // (profiles/r04_refill_sweep_f32_culling*.log).  At five waves per SIMD (round 5) 24 / 24: configs[2] mixed 99.1 -> 98.3 ms, configs[3] 475.8 -> 472.8
    // (16: 102.3, 28: 101.2 on the nine-entry layout; profiles/r05_five_waves_thresholds_ab.log)
    unsigned int refill_min_hyb = 24, refill_min_any_hyb = 24;
    // lanes of a wave that must be at a leaf slot before the leaf step runs (mixed / any-hit launches; 0 or 1: no waiting); 12 since the
    // five-waves layout (10: configs[2] mixed 99.1 against 98.8, configs[3] 475.8 against 471.9; configs[1] is flat from 6 to 14)
    unsigned int leaf_min = 12;
    unsigned int lds_shapes = 1;   // the few sphere / disk records of a scene staged in LDS by the traversal launches (CRAY_LDS_SHAPES=0: read global memory)
    unsigned int steal = 1;   // work sharing among the lanes of a wave in the drain of the shadow-ray launches (trace_body, STEAL); CRAY_STEAL=0 turns it off
    int trace_blocks_per_cu = trace_waves(0);       // resident blocks per CU of the persistent traversal launches: f64 records (cray_device.h)
    int trace_blocks_per_cu_hyb = trace_waves(1);   // ... certified-f32 records
    int trace32_blocks_per_cu = 4;
    int shade_blocks_per_cu = 0;   // 0: the occupancy of the instantiation that runs (launch_shade)
    // path-state pool
    size_t capacity = 0;
    std::vector<void*> state_allocs;
    PathState ps{};       // view 0: live buffer 0 + the shadow buffer + L (k_raygen, bounce 0, the cray_trace hook)
    PathState ps1{};      // view 1: live buffer 1 + the same shadow buffer and L
    uint32_t* queue[2] = {nullptr, nullptr};
    uint32_t* shadow_queue = nullptr;
    // third stack level (Counters::deep_*), allocated after a frame overflowed LDS + scratch
    double* tail_res = nullptr;     // results of the helpers of closest-hit rays in small mixed launches (trace_body TAIL)
    unsigned int tail_seg = 1;      // CRAY_TAIL_SEG=0: the small-launch instantiation shares shadow rays only
    unsigned int tail_age = 16;     // CRAY_TAIL_AGE: a segment hands parts out only after 4 x this many iterations (long rays only)
    // mixed launches below this many rays run in the TAIL instantiation (CRAY_TAIL_RAYS); 0 = never, the default: exact and
    // tested, but as measured it costs more than it gives (the f64 instantiation is slower than the f32-culling one
    // it replaces, and a helper's winning hit near the ray's origin is often uncertifiable, which walks the ray twice)
    unsigned int tail_rays = 0;
    uint32_t* deep_ref = nullptr;
    double* deep_key = nullptr;
    unsigned int deep_depth = 0;
    size_t deep_threads = 0;
    int hybrid = -1;    // records the exact traversal reads: 0 f64, 1 certified f32 culling (same results); -1 (default) =
                        // per scene and per launch kind, whichever the scene's probe passes show to be faster (choose_trace_records).
                        // CRAY_HYBRID=0/1 pins it.
    int records_b0 = -1, records_rest = -1;   // CRAY_RECORDS_B0 / CRAY_RECORDS_REST (0 / 1): pin the records per launch kind instead of timing
    int mix_trace = 1;  // shadow rays of bounce b and segments of bounce b+1 in one launch (CRAY_MIX_TRACE=0 disables)
    int log_queues = 0; // CRAY_LOG_QUEUES=1 (diagnostics): after every bounce, sync and print the queue lengths to stderr
    Counters* counters = nullptr;
    uint32_t* pix_list = nullptr;
    size_t pix_capacity = 0;
    // the same pixels in the order they are RENDERED: the rank's tiles sorted by the cost of their camera rays, most expensive
    // first (ensure_tile_order, k_tile_probe).  pix_list stays the canonical order the gather packs and sends in.
    uint32_t* pix_render = nullptr;
    size_t pix_render_capacity = 0;
    const uint32_t* pix_order = nullptr;          // what run_pass reads: pix_render or pix_list
    uint64_t order_scene = 0;                     // uid of the scene, and the pixel list, pix_render was computed for
    uint64_t order_key[6] = {0, 0, 0, 0, 0, 0};
    bool order_valid = false;
    int tile_order = 1;                           // CRAY_TILE_ORDER=0: render in the canonical tile order
    uint64_t pix_key[6] = {0, 0, 0, 0, 0, 0};  // (W, H, tile w, tile h, rank, world) of the list in pix_list
    size_t pix_count = 0;
    bool pix_count_valid = false;
    float* film = nullptr;
    size_t film_floats = 0;
    float* out_stage = nullptr;  // resolved film on the device when the caller's out_rgb is host memory (no per-frame hipMalloc)
    size_t out_stage_floats = 0;
    // event pool for per-family kernel timing
    std::vector<hipEvent_t> events;
    // multi-GPU (cray_comm_*): one communicator per context, Film tiles gathered to rank 0
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    double* comm_scratch = nullptr;      // 64 doubles on the device (barrier word, host-value all-reduces, scene header)
    float* packed = nullptr;             // this rank's tiles, packed (ranks != 0)
    size_t packed_floats = 0;
    float* gathered = nullptr;           // rank 0: every rank's packed tiles, concatenated in rank order
    size_t gathered_floats = 0;
    uint32_t* all_pix = nullptr;         // rank 0: pixel index of every position of `gathered`
    size_t all_pix_capacity = 0;
    uint64_t all_key[5] = {0, 0, 0, 0, 0};  // (W, H, tile w, tile h, world) of all_pix
    bool all_valid = false;
    std::vector<size_t> rank_offset;     // world + 1 prefix sums of the per-rank pixel counts
};

static std::atomic<uint64_t> g_scene_uid{1};
struct cray_scene {
    uint64_t uid = g_scene_uid.fetch_add(1);   // never reused (a freed scene's address may be): what per-scene caches of a ctx are keyed on
    cray_ctx* ctx = nullptr;
    DevScene dev{};
    std::vector<void*> allocs;
    std::vector<size_t> alloc_bytes;  // parallel to allocs, in the order of scene_arrays()
    std::vector<void*> extra_allocs;  // per-rank extras that are not part of a broadcast (the f32 records of the fast mode)
    uint32_t n_slots = 0;             // leaf slots (without the pad slot)
    uint64_t bytes = 0;
    uint32_t n_prims = 0;
    cray_bvh_build_stats build_stats{};  // resident build only
    uint32_t features = SF_ALL;  // what the scene can make k_shade do (cray_shading.h)
    int shade_variant = kNumShadeVariants - 1;
    bool hybrid_ok = false;  // the scene is inside the range the certified f32 culling is proven for (cray_math.h hyb_scene_ok)
    // Which records the traversal launches of this scene read (0 f64, 1 certified f32 culling): the bounce-0 launch (coherent
    // camera rays) and the others separately.  Both kinds of records give the reference's hits bit for bit, so the choice is a
    // matter of speed only and it depends on the scene (axis-aligned scenes need the exact retake in most box decisions).  With
    // ctx->hybrid = -1 a scene's first render that is big enough to time runs a few small PROBE passes on both kinds before its
    // first frame (probe_trace_records) and the faster kind per launch kind is kept for the scene's lifetime in chosen_*; use_* is
    // what the CURRENT call reads (counting frames and the f32 fast mode read f64 records whatever was chosen).
    int use_b0 = 0, use_rest = 0;
    int chosen_b0 = -1, chosen_rest = -1;   // -1: not chosen yet
    bool needs_deep = false;       // a frame of this scene overflowed LDS + scratch: its launches run in the instantiations with the third stack level
    double tune_ms[2][2] = {{0.0, 0.0}, {0.0, 0.0}};   // [records][kind]: best time of the probe passes' bounce-0 launch / other traversal launches
    double probe_ms = 0.0;         // wall time the probe passes took (once per scene)
};

namespace {

template <class T>
int upload(cray_scene* s, const T* host, size_t n, const T** out) {
    *out = nullptr;
    size_t bytes = n * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);  // keep a valid pointer for empty tables
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    s->allocs.push_back(d);
    s->alloc_bytes.push_back(bytes);
    s->bytes += bytes;
    if (n) HIP_TRY(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return CRAY_OK;
}

// Bytes of path state per path of a pass: two live buffers (13 f64 + 4 words each), the shadow buffer (10 f64 + 2 words),
// L (3 f64) and three queues.
constexpr size_t kBytesPerPath = 2 * (10 * 8 + 4 * 4) + 3 * 8 + (10 * 8 + 2 * 4) + 3 * 8 + 3 * 4;   // two live buffers, the hit record, the shadow buffer, L, three queues

int ensure_state(cray_ctx* c, size_t capacity) {
    if (c->capacity >= capacity) return CRAY_OK;
    for (void* p : c->state_allocs) (void)hipFree(p);
    c->state_allocs.clear();
    c->capacity = 0;
    auto alloc = [&](size_t bytes, void** out) -> int {
        HIP_TRY(hipMalloc(out, bytes));
        c->state_allocs.push_back(*out);
        return CRAY_OK;
    };
    // k_shade compacts per tile: slots run up to (number of tiles) x (tile size) <= paths + one tile
    const size_t slots = capacity + kShadeTile;
    int r;
    for (PathState* v : {&c->ps, &c->ps1}) {
        double** live[] = {&v->ox, &v->oy, &v->oz, &v->dx, &v->dy, &v->dz, &v->br, &v->bg, &v->bb, &v->prev_pdf};
        for (double** f : live)
            if ((r = alloc(slots * sizeof(double), (void**)f))) return r;
        if ((r = alloc(slots * 4, (void**)&v->hprim))) return r;
        if ((r = alloc(slots * 4, (void**)&v->hash))) return r;
        if ((r = alloc(slots * 4, (void**)&v->flags))) return r;
        if ((r = alloc(slots * 4, (void**)&v->p0))) return r;
    }
    // the hit record (distance, barycentrics) exists once: the traversal of bounce b writes it at the live slots of bounce b, k_shade(b)
    // reads it there, and the traversal of bounce b + 1 — the next writer, at the slots of the OTHER live buffer — starts after
    // k_shade(b) has ended.  (hprim stays per buffer: k_shade writes the next bounce's start primitive while other blocks still read
    // this bounce's hit primitive.)  340 instead of 364 B per path.
    double** shared[] = {&c->ps.ht, &c->ps.hu, &c->ps.hv, &c->ps.lr, &c->ps.lg, &c->ps.lb, &c->ps.sox, &c->ps.soy, &c->ps.soz, &c->ps.sdx, &c->ps.sdy, &c->ps.sdz, &c->ps.stmax,
                         &c->ps.cr, &c->ps.cg, &c->ps.cb};
    for (double** f : shared)
        if ((r = alloc(slots * sizeof(double), (void**)f))) return r;
    if ((r = alloc(slots * 4, (void**)&c->ps.sp0))) return r;
    if ((r = alloc(slots * 4, (void**)&c->ps.sprim))) return r;
    // the second view shares the shadow buffer and L
    c->ps1.ht = c->ps.ht; c->ps1.hu = c->ps.hu; c->ps1.hv = c->ps.hv;
    c->ps1.lr = c->ps.lr; c->ps1.lg = c->ps.lg; c->ps1.lb = c->ps.lb;
    c->ps1.sox = c->ps.sox; c->ps1.soy = c->ps.soy; c->ps1.soz = c->ps.soz; c->ps1.sdx = c->ps.sdx; c->ps1.sdy = c->ps.sdy; c->ps1.sdz = c->ps.sdz;
    c->ps1.stmax = c->ps.stmax; c->ps1.cr = c->ps.cr; c->ps1.cg = c->ps.cg; c->ps1.cb = c->ps.cb; c->ps1.sp0 = c->ps.sp0; c->ps1.sprim = c->ps.sprim;
    if ((r = alloc(slots * 4, (void**)&c->queue[0]))) return r;
    if ((r = alloc(slots * 4, (void**)&c->queue[1]))) return r;
    if ((r = alloc(slots * 4, (void**)&c->shadow_queue))) return r;
    c->capacity = capacity;
    return CRAY_OK;
}

// zero the device counters, keeping the description of the third stack level
int reset_counters(cray_ctx* c) {
    HIP_TRY(hipMemsetAsync(c->counters, 0, sizeof(Counters), c->stream));
    if (c->deep_depth || c->tail_res) {
        struct { uint32_t* r; double* k; unsigned int d, tail_rays; double* tail_res; } v{c->deep_ref, c->deep_key, c->deep_depth, c->tail_res ? (c->tail_rays | (c->tail_age << 24)) : 0u, c->tail_res};
        static_assert(sizeof(v) == sizeof(Counters) - offsetof(Counters, deep_ref), "deep_* and tail_* are the tail of Counters");
        HIP_TRY(hipMemcpyAsync(&c->counters->deep_ref, &v, sizeof(v), hipMemcpyHostToDevice, c->stream));
    }
    return CRAY_OK;
}

// The result table of the small-launch instantiation (trace_body TAIL): kTailSlots x 4 doubles per thread of the largest
// traversal grid.  Bit 30 of a child reference must be free for its helpers' bookkeeping: scenes beyond 2^27 leaf slots or
// 2^30 nodes (and CRAY_TAIL_RAYS=0) run every mixed launch in the ordinary instantiation.
int ensure_tail(cray_ctx* c, const cray_scene* s) {
#ifndef CRAY_WITH_EXPERIMENTS
    (void)s;
    c->tail_rays = 0;   // the small-launch instantiation (DESIGN.md 3.1 TAIL: exact, tested, slower) is compiled with -DCRAY_WITH_EXPERIMENTS only
#endif
    const bool want = c->tail_rays != 0 && c->steal && c->mix_trace && s->n_slots < (1u << 27) && s->dev.n_inner < (1u << 30);
    if (!want) {
        if (c->tail_res) { (void)hipFree(c->tail_res); c->tail_res = nullptr; }
        return CRAY_OK;
    }
    if (c->tail_res) return CRAY_OK;
    const size_t threads = (size_t)c->n_cu * (size_t)(c->trace_blocks_per_cu > 8 ? c->trace_blocks_per_cu : 8) * kBlock;
    HIP_TRY(hipMalloc((void**)&c->tail_res, threads * kTailSlots * 4 * sizeof(double)));
    return CRAY_OK;
}

// The reference's traversal has no depth limit (recursion / Vec); here a ray may keep kStackDepth nodes pending in
// LDS + scratch.  When a frame reports an overflow the runtime adds this third level in HBM and renders again.
constexpr unsigned int kDeepDepth = 2000;
int ensure_deep(cray_ctx* c) {
    if (c->deep_depth) return CRAY_OK;
    const size_t threads = (size_t)c->n_cu * (size_t)(c->trace_blocks_per_cu > 8 ? c->trace_blocks_per_cu : 8) * kBlock;
    HIP_TRY(hipMalloc((void**)&c->deep_ref, threads * kDeepDepth * sizeof(uint32_t)));
    hipError_t e = hipMalloc((void**)&c->deep_key, threads * kDeepDepth * sizeof(double));
    if (e != hipSuccess) { (void)hipFree(c->deep_ref); c->deep_ref = nullptr; set_last_error("hipMalloc of the deep traversal stack failed: %s", hipGetErrorString(e)); return CRAY_ERR_HIP; }
    c->deep_threads = threads;
    c->deep_depth = kDeepDepth;
    return CRAY_OK;
}

enum { FAM_CLOSEST = 0, FAM_ANY = 1, FAM_SHADE = 2, FAM_OTHER = 3, FAM_MIXED = 4, FAM_COUNT = 5 };
struct EventTimer {
    cray_ctx* c;
    size_t used = 0;
    struct Span { size_t a, b; int family; };
    std::vector<Span> spans;
    explicit EventTimer(cray_ctx* ctx) : c(ctx) {}
    int begin(int family) {
        while (c->events.size() < used + 2) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            c->events.push_back(e);
        }
        HIP_TRY(hipEventRecord(c->events[used], c->stream));
        spans.push_back(Span{used, used + 1, family});
        used += 2;
        return CRAY_OK;
    }
    int end() {
        HIP_TRY(hipEventRecord(c->events[spans.back().b], c->stream));
        return CRAY_OK;
    }
    int collect(double ms[FAM_COUNT], uint32_t launches[FAM_COUNT]) {
        for (int i = 0; i < FAM_COUNT; i++) { ms[i] = 0.0; launches[i] = 0; }
        for (const Span& s : spans) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, c->events[s.a], c->events[s.b]));
            ms[s.family] += t;
            launches[s.family] += 1;
        }
        return CRAY_OK;
    }
};

int grid_for(const cray_ctx* c, size_t n, int blocks_per_cu) {
    size_t want = (n + kBlock - 1) / kBlock;
    size_t cap = (size_t)c->n_cu * blocks_per_cu;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

// resident blocks per CU of a kernel, as the runtime computes it from the code object's registers and LDS.  The grid-stride kernels
// size their grids as a multiple of it: a grid that is not one leaves a last round with part of the chip idle.
int resident_blocks(const void* kernel, int threads, int fallback) {
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kernel, threads, 0) != hipSuccess || o < 1) { (void)hipGetLastError(); o = fallback; }
    return o;
}

void fill_stats(const Counters& h, cray_stats* st) {
    st->closest_rays = h.closest_rays; st->shadow_rays = h.shadow_rays + h.shadow_skipped; st->shadow_skipped = h.shadow_skipped;
    st->closest_nodes = h.closest_nodes; st->closest_prims = h.closest_prims;
    st->shadow_nodes = h.shadow_nodes; st->shadow_prims = h.shadow_prims;
    st->closest_tri_tests = h.closest_tri; st->shadow_tri_tests = h.shadow_tri;
    st->nonfinite = h.nonfinite; st->stack_overflow = h.stack_overflow; st->closest_hits = h.closest_hits;
    st->tail_split = (uint32_t)(h.tail_helped > 0xffffffull ? 0xffffffull : h.tail_helped) | ((uint32_t)(h.tail_again > 255ull ? 255ull : h.tail_again) << 24);
#ifdef CRAY_TRACE_DIAG
    for (int a = 0; a < 2; a++) {
        const unsigned long long* g = h.diag + 16 * a;
        if (!g[0]) continue;
        fprintf(stderr, "diag %s: wave-iterations %llu, active lanes/iter %.1f (interior %.1f, leaf %.1f, resolve %.2f in %.1f%% of the iterations); exhausted iters %.1f%% at %.1f lanes; refills %llu x %.1f lanes\n",
                a ? "any" : "closest", g[0], (double)g[1] / g[0], (double)g[6] / g[0], (double)g[7] / g[0], (double)g[8] / g[0], 100.0 * g[9] / g[0],
                100.0 * g[2] / g[0], g[2] ? (double)g[3] / g[2] : 0.0, g[4], g[4] ? (double)g[5] / g[4] : 0.0);
        fprintf(stderr, "diag %s: iterations with a lane at a leaf %.1f%%, at a sphere / disk slot %.1f%% (%.2f lanes), at a leaf of several slots %.1f%%\n",
                a ? "any" : "closest", 100.0 * g[10] / g[0], 100.0 * g[11] / g[0], (double)g[13] / g[0], 100.0 * g[12] / g[0]);
        fprintf(stderr, "diag %s (certified-f32 launches read the second line as): every lane at an interior node on the SAME node in %.1f%% of the iterations; of %.1f such lanes %.1f share the first one's node\n",
                a ? "any" : "closest", 100.0 * g[11] / g[0], (double)g[13] / g[0], (double)g[12] / g[0]);
        fprintf(stderr, "diag %s: pop loop entered in %.1f%% of the iterations, %.2f trips per entry (the wave runs the maximum over its lanes)\n",
                a ? "any" : "closest", 100.0 * g[15] / g[0], g[15] ? (double)g[14] / g[15] : 0.0);
    }
#endif
}

}  // namespace

extern "C" const char* cray_last_error(void) { return g_err; }

static void comm_release(cray_ctx* c);  // cray_comm section at the end of this file

namespace {
struct DevMem {  // frees its allocations on every exit path
    std::vector<void*> ptrs;
    ~DevMem() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t get(T** out, size_t count) {
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(d);
        *out = (T*)d;
        return e;
    }
    void release(void* p) {  // hand an allocation over to another owner
        for (auto& q : ptrs) if (q == p) { q = ptrs.back(); ptrs.pop_back(); return; }
    }
};
struct BvhOnDevice { cray_bvh_node* d_nodes; uint32_t* d_order; uint32_t n_nodes; cray_bvh_build_stats stats; };
int bvh_build_device(cray_ctx* c, const double* d_box, uint32_t n, DevMem& mem, BvhOnDevice* r);
}  // namespace

extern "C" int cray_ctx_create(int device_id, void* stream, cray_ctx** out) {
    if (!out) { set_last_error("cray_ctx_create: out is null"); return CRAY_ERR_INVALID; }
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
        set_last_error("no HIP device visible: this backend has no CPU fallback");
        return CRAY_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n_dev) { set_last_error("device %d out of range (%d devices)", device_id, n_dev); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device_id));
    cray_ctx* c = new cray_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { HIP_TRY(hipStreamCreate(&c->stream)); c->own_stream = true; }
    HIP_TRY(hipMalloc((void**)&c->counters, sizeof(Counters)));
    // experiment knobs, clamped to the ranges the kernels are written for (refill_min > 64 would leave the persistent
    // waves of k_trace spinning without ever fetching a ray; a grid of 0 blocks is a launch error)
    auto env_int = [](const char* name, int lo, int hi, int dflt) {
        const char* e = getenv(name);
        if (!e || !*e) return dflt;
        const int v = atoi(e);
        return v < lo ? lo : (v > hi ? hi : v);
    };
    c->mix_trace = env_int("CRAY_MIX_TRACE", 0, 1, c->mix_trace);
    c->hybrid = env_int("CRAY_HYBRID", -1, 1, c->hybrid);
    c->records_b0 = env_int("CRAY_RECORDS_B0", -1, 1, c->records_b0);
    c->records_rest = env_int("CRAY_RECORDS_REST", -1, 1, c->records_rest);
    c->log_queues = env_int("CRAY_LOG_QUEUES", 0, 1, 0);
    c->refill_min = (unsigned int)env_int("CRAY_REFILL_MIN", 1, 64, (int)c->refill_min);
    c->refill_min_b0 = (unsigned int)env_int("CRAY_REFILL_MIN_B0", 1, 64, (int)c->refill_min_b0);
    c->refill_min_any = (unsigned int)env_int("CRAY_REFILL_MIN_ANY", 1, 64, (int)c->refill_min_any);
    // (an explicit CRAY_REFILL_MIN / _ANY applies to both kinds of launches: the sweeps set those)
    c->refill_min_hyb = (unsigned int)env_int("CRAY_REFILL_MIN", 1, 64, (int)c->refill_min_hyb);
    c->refill_min_any_hyb = (unsigned int)env_int("CRAY_REFILL_MIN_ANY", 1, 64, (int)c->refill_min_any_hyb);
    c->steal = (unsigned int)env_int("CRAY_STEAL", 0, 1, (int)c->steal);
    c->lds_shapes = (unsigned int)env_int("CRAY_LDS_SHAPES", 0, 1, (int)c->lds_shapes);
    c->tile_order = env_int("CRAY_TILE_ORDER", 0, 1, c->tile_order);
    c->leaf_min = (unsigned int)env_int("CRAY_LEAF_MIN", 0, 63, (int)c->leaf_min);
    c->tail_rays = (unsigned int)env_int("CRAY_TAIL_RAYS", 0, (1 << 24) - 1, (int)c->tail_rays);
    c->tail_seg = (unsigned int)env_int("CRAY_TAIL_SEG", 0, 1, (int)c->tail_seg);
    c->tail_age = (unsigned int)env_int("CRAY_TAIL_AGE", 0, 255, (int)c->tail_age);
    c->trace_blocks_per_cu = env_int("CRAY_TRACE_BLOCKS_PER_CU", 1, 16, c->trace_blocks_per_cu);
    c->trace_blocks_per_cu_hyb = env_int("CRAY_TRACE_BLOCKS_PER_CU_HYB", 1, 16, c->trace_blocks_per_cu_hyb);
    c->shade_blocks_per_cu = env_int("CRAY_SHADE_BLOCKS_PER_CU", 0, 64, c->shade_blocks_per_cu);
    c->trace32_blocks_per_cu = env_int("CRAY_TRACE32_BLOCKS_PER_CU", 1, 16, c->trace32_blocks_per_cu);
    {
        // the kernels read their path-state pointers straight from the kernel-argument segment (cray_kernels.h ps_kernarg): check once
        // that the offsets computed there are where the compiler put the arguments
        DevScene probe_sc;
        memset(&probe_sc, 0, sizeof(probe_sc));
        probe_sc.max_depth = 0x5a5a5a5au;
        PathState a, b;
        memset(&a, 0, sizeof(a));
        memset(&b, 0, sizeof(b));
        a.ox = reinterpret_cast<double*>(0x1000); a.hprim = reinterpret_cast<int32_t*>(0x2000); a.sprim = reinterpret_cast<int32_t*>(0x3000); a.lr = reinterpret_cast<double*>(0x4000);
        b.ox = reinterpret_cast<double*>(0x5000); b.hprim = reinterpret_cast<int32_t*>(0x6000); b.sprim = reinterpret_cast<int32_t*>(0x7000); b.lr = reinterpret_cast<double*>(0x8000);
        uint32_t* d_bad = reinterpret_cast<uint32_t*>(&c->counters->pad_head_);
        uint32_t bad = ~0u;
        hipLaunchKernelGGL(k_kernarg_check, dim3(1), dim3(64), 0, c->stream, probe_sc, a, b, d_bad);
        hipError_t err = hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, c->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
        if (err != hipSuccess || bad != 0u) {
            set_last_error("kernel-argument layout check failed (%s, code %u): the path-state views are not where cray_kernels.h ps_kernarg() expects them",
                           err != hipSuccess ? hipGetErrorString(err) : "mismatch", bad);
            cray_ctx_destroy(c);
            return CRAY_ERR_HIP;
        }
        (void)hipMemsetAsync(d_bad, 0, sizeof(uint32_t), c->stream);
    }
    *out = c;
    return CRAY_OK;
}

extern "C" void cray_ctx_destroy(cray_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (void* p : c->state_allocs) (void)hipFree(p);
    if (c->counters) (void)hipFree(c->counters);
    if (c->deep_ref) (void)hipFree(c->deep_ref);
    if (c->tail_res) (void)hipFree(c->tail_res);
    if (c->deep_key) (void)hipFree(c->deep_key);
    if (c->pix_list) (void)hipFree(c->pix_list);
    if (c->pix_render) (void)hipFree(c->pix_render);
    if (c->film) (void)hipFree(c->film);
    if (c->out_stage) (void)hipFree(c->out_stage);
    comm_release(c);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void cray_scene_free(cray_scene* s) {
    if (!s) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    for (void* p : s->allocs) (void)hipFree(p);
    for (void* p : s->extra_allocs) (void)hipFree(p);
    delete s;
}
extern "C" uint64_t cray_scene_device_bytes(const cray_scene* s) { return s ? s->bytes : 0; }
extern "C" void cray_scene_build_stats(const cray_scene* s, cray_bvh_build_stats* out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    if (s) *out = s->build_stats;
}
extern "C" void cray_scene_info(const cray_scene* s, uint32_t* w, uint32_t* h, uint32_t* ns, uint32_t* depth) {
    if (!s) return;
    if (w) *w = s->dev.film_w;
    if (h) *h = s->dev.film_h;
    if (ns) *ns = s->dev.num_samples;
    if (depth) *depth = s->dev.max_depth;
}

// Direction vectors every later cray_scene_upload copies to the device: the built-in table or the caller's
// (cray_set_sobol_vectors).
static uint16_t g_sobol_override[CRAY_SOBOL_SETS * CRAY_SOBOL_BITS * 4];
static const uint16_t* g_sobol_table = &CRAY_SOBOL_REV_VECTORS[0][0][0];
extern "C" int cray_set_sobol_vectors(const uint16_t* rev_vectors) {
    if (!rev_vectors) { g_sobol_table = &CRAY_SOBOL_REV_VECTORS[0][0][0]; return CRAY_OK; }
    memcpy(g_sobol_override, rev_vectors, sizeof(g_sobol_override));
    g_sobol_table = g_sobol_override;
    return CRAY_OK;
}

// the leanest instantiation of k_shade that covers `features`
static int pick_shade_variant(uint32_t features) {
    int best = kNumShadeVariants - 1;
    for (int i = 0; i < kNumShadeVariants; i++)
        if ((kShadeVariants[i] & features) == features && __builtin_popcount(kShadeVariants[i]) < __builtin_popcount(kShadeVariants[best])) best = i;
    return best;
}

// ---- resident build: Bvh::new on the device and the traversal layout derived from its output in place -------------
namespace {

// Shape::bounds of a triangle (shape.rs:402-438 via Bounds::new of the three vertices): min / max of v0, v0 + e1, v0 + e2 in the
// host's operand order (cray_host.cpp); other shapes get their host-computed box scattered in afterwards.
__global__ void __launch_bounds__(kBlock) k_prim_bounds(const cray_prim* __restrict__ prims, const cray_triangle* __restrict__ tris, uint32_t n, uint32_t n_tris,
                                                        double* __restrict__ box, unsigned int* __restrict__ err) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const cray_prim p = prims[i];
    double* o = box + (size_t)i * 6;
    if (p.shape_kind != CRAY_SHAPE_TRIANGLE) { for (int k = 0; k < 6; k++) o[k] = 0.0; return; }
    if (p.shape >= n_tris) { atomicOr(err, 1u); return; }
    const cray_triangle t = tris[p.shape];
    const double v0[3] = {t.v0.x, t.v0.y, t.v0.z};
    const double v1[3] = {t.v0.x + t.e1.x, t.v0.y + t.e1.y, t.v0.z + t.e1.z}, v2[3] = {t.v0.x + t.e2.x, t.v0.y + t.e2.y, t.v0.z + t.e2.z};
    bool finite = true;
    for (int k = 0; k < 3; k++) {
        o[k] = min_nn(v1[k], min_nn(v2[k], v0[k]));
        o[3 + k] = max_nn(v1[k], max_nn(v2[k], v0[k]));
        finite = finite && isfinite(o[k]) && isfinite(o[3 + k]);
    }
    if (!finite) atomicOr(err, 2u);
}
__global__ void __launch_bounds__(kBlock) k_other_bounds(const cray_prim_bound* __restrict__ ob, uint32_t n_other, uint32_t n, double* __restrict__ box, unsigned int* __restrict__ err) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_other) return;
    const cray_prim_bound b = ob[i];
    if (b.prim >= n) { atomicOr(err, 4u); return; }
    double* o = box + (size_t)b.prim * 6;
    for (int k = 0; k < 3; k++) { o[k] = b.bmin[k]; o[3 + k] = b.bmax[k]; }
}
__global__ void __launch_bounds__(kBlock) k_flag_interior(const cray_bvh_node* __restrict__ nodes, uint32_t n_nodes, uint8_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_nodes) flag[i] = nodes[i].is_leaf ? 0 : 1;
}
// one 128-B record per interior node, at its rank among the interior nodes in DFS pre-order (T = exclusive scan of the flags)
__global__ void __launch_bounds__(kBlock) k_make_inner(const cray_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ T, uint32_t n_nodes,
                                                       InnerNode* __restrict__ inner, unsigned int* __restrict__ err, unsigned int* __restrict__ out_of_div_range) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    const cray_bvh_node nd = nodes[i];
    bool in_range = true;
    for (int k = 0; k < 3; k++) in_range = in_range && div_range_ok(nd.bmin[k]) && div_range_ok(nd.bmax[k]);
    if (!in_range) atomicOr(out_of_div_range, 1u);
    if (nd.is_leaf) { if (nd.count < 1 || nd.count > 8) atomicOr(err, 8u); return; }
    if (nd.left >= n_nodes || nd.right >= n_nodes || nd.axis < 0 || nd.axis > 2) { atomicOr(err, 16u); return; }
    const cray_bvh_node l = nodes[nd.left], r = nodes[nd.right];
    InnerNode o;
    for (int k = 0; k < 3; k++) { o.lo0[k] = l.bmin[k]; o.hi0[k] = l.bmax[k]; o.lo1[k] = r.bmin[k]; o.hi1[k] = r.bmax[k]; }
    o.ref0 = l.is_leaf ? (kLeafBit | (l.first << 3) | (l.count - 1)) : T[nd.left];
    o.ref1 = r.is_leaf ? (kLeafBit | (r.first << 3) | (r.count - 1)) : T[nd.right];
    o.axis = (uint32_t)nd.axis; o.pad_ = 0; o.pad2_[0] = 0.0; o.pad2_[1] = 0.0;
    inner[T[i]] = o;
}
__global__ void __launch_bounds__(kBlock) k_make_slots(const uint32_t* __restrict__ order, const cray_prim* __restrict__ prims, const cray_triangle* __restrict__ tris,
                                                       uint32_t n, LeafSlot* __restrict__ slots) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j > n) return;
    LeafSlot s;
    for (int k = 0; k < 3; k++) { s.v0[k] = 0.0; s.e1[k] = 0.0; s.e2[k] = 0.0; }
    s.prim = 0; s.kind = 0;
    if (j < n) {  // slot n is the zero pad the unified 112-B record fetch may read
        const uint32_t pi = order[j];
        const cray_prim p = prims[pi];
        s.prim = pi; s.kind = (uint32_t)p.shape_kind;
        if (p.shape_kind == CRAY_SHAPE_TRIANGLE) {
            const cray_triangle t = tris[p.shape];
            s.v0[0] = t.v0.x; s.v0[1] = t.v0.y; s.v0[2] = t.v0.z;
            s.e1[0] = t.e1.x; s.e1[1] = t.e1.y; s.e1[2] = t.e1.z;
            s.e2[0] = t.e2.x; s.e2[1] = t.e2.y; s.e2[2] = t.e2.z;
        } else {
            s.v0[0] = __longlong_as_double((long long)p.shape);   // the sphere / disk index, so that k_trace needs no prims[] lookup
        }
    }
    slots[j] = s;
}
__global__ void __launch_bounds__(kBlock) k_make_trishade(const cray_triangle* __restrict__ tris, uint32_t n_tris, TriShade* __restrict__ shade) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_tris) return;
    const cray_triangle t = tris[i];
    TriShade o;
    o.n0[0] = t.n0.x; o.n0[1] = t.n0.y; o.n0[2] = t.n0.z;
    o.n01[0] = t.n01.x; o.n01[1] = t.n01.y; o.n01[2] = t.n01.z;
    o.n02[0] = t.n02.x; o.n02[1] = t.n02.y; o.n02[2] = t.n02.z;
    for (int k = 0; k < 2; k++) { o.uv0[k] = t.uv0[k]; o.uv01[k] = t.uv01[k]; o.uv02[k] = t.uv02[k]; }
    shade[i] = o;
}

// Registers the four BVH-side arrays of the scene in cray_scene_upload's order (inner, slots, prims, tri_shade).
int build_resident(cray_ctx* c, const cray_flat_scene* f, cray_scene* s) {
    using namespace cray::bvhb;
    const uint32_t n = f->n_prims, n_tris = f->n_triangles;
    if (n >= (1u << 28)) { set_last_error("too many primitives for the 28-bit leaf slot index"); return CRAY_ERR_UNSUPPORTED; }
    hipStream_t st = c->stream;
    DevScene& d = s->dev;
    DevMem mem;
    cray_triangle* d_tris; cray_prim* d_prims; double* d_box; unsigned int* d_err; cray_prim_bound* d_other;
    HIP_TRY(mem.get(&d_tris, n_tris));
    HIP_TRY(mem.get(&d_prims, n));
    HIP_TRY(mem.get(&d_box, (size_t)n * 6));
    HIP_TRY(mem.get(&d_err, 2));
    HIP_TRY(mem.get(&d_other, f->n_other_bounds));
    if (n_tris) HIP_TRY(hipMemcpyAsync(d_tris, f->triangles, (size_t)n_tris * sizeof(cray_triangle), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_prims, f->prims, (size_t)n * sizeof(cray_prim), hipMemcpyHostToDevice, st));
    if (f->n_other_bounds) HIP_TRY(hipMemcpyAsync(d_other, f->other_bounds, (size_t)f->n_other_bounds * sizeof(cray_prim_bound), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_err, 0, 2 * sizeof(unsigned int), st));
    const dim3 blk(kBlock), grid_n((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_prim_bounds, grid_n, blk, 0, st, (const cray_prim*)d_prims, (const cray_triangle*)d_tris, n, n_tris, d_box, d_err);
    if (f->n_other_bounds) hipLaunchKernelGGL(k_other_bounds, dim3((f->n_other_bounds + kBlock - 1) / kBlock), blk, 0, st, (const cray_prim_bound*)d_other, f->n_other_bounds, n, d_box, d_err);
    unsigned int err[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(err, d_err, sizeof(err), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (err[0]) { set_last_error("resident build: bad primitive table (code %u: 1 triangle index, 2 non-finite bound, 4 other_bounds index)", err[0]); return CRAY_ERR_INVALID; }

    BvhOnDevice bvh{};
    int rc = bvh_build_device(c, d_box, n, mem, &bvh);   // Bvh::new: the reference's tree, DFS pre-order, on the device
    if (rc != CRAY_OK) return rc;
    const uint32_t n_nodes = bvh.n_nodes;

    uint8_t* d_flag; uint32_t *d_T, *d_tile, *d_total;
    const uint32_t n_tiles = (n_nodes + kScanTile - 1) / kScanTile;
    HIP_TRY(mem.get(&d_flag, n_nodes));
    HIP_TRY(mem.get(&d_T, (size_t)n_nodes + 1));
    HIP_TRY(mem.get(&d_tile, n_tiles));
    HIP_TRY(mem.get(&d_total, 1));
    const dim3 grid_nodes((n_nodes + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_flag_interior, grid_nodes, blk, 0, st, (const cray_bvh_node*)bvh.d_nodes, n_nodes, d_flag);
    hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(kTB), 0, st, (const uint8_t*)d_flag, d_T, d_tile, n_nodes);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kTB), 0, st, d_tile, n_tiles, d_total);
    hipLaunchKernelGGL(k_scan_add, dim3((n_nodes + kTB - 1) / kTB), dim3(kTB), 0, st, d_T, (const uint32_t*)d_tile, (const uint32_t*)d_total, n_nodes);
    uint32_t n_inner = 0;
    cray_bvh_node root;
    HIP_TRY(hipMemcpyAsync(&n_inner, d_T + n_nodes, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&root, bvh.d_nodes, sizeof(root), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));

    InnerNode* d_inner; LeafSlot* d_slots; TriShade* d_shade;
    HIP_TRY(mem.get(&d_inner, n_inner ? n_inner : 1));
    HIP_TRY(mem.get(&d_slots, (size_t)n + 1));
    HIP_TRY(mem.get(&d_shade, n_tris));
    if (!n_inner) HIP_TRY(hipMemsetAsync(d_inner, 0, sizeof(InnerNode), st));
    hipLaunchKernelGGL(k_make_inner, grid_nodes, blk, 0, st, (const cray_bvh_node*)bvh.d_nodes, (const uint32_t*)d_T, n_nodes, d_inner, d_err, d_err + 1);
    hipLaunchKernelGGL(k_make_slots, dim3((n + 1 + kBlock - 1) / kBlock), blk, 0, st, (const uint32_t*)bvh.d_order, (const cray_prim*)d_prims, (const cray_triangle*)d_tris, n, d_slots);
    if (n_tris) hipLaunchKernelGGL(k_make_trishade, dim3((n_tris + kBlock - 1) / kBlock), blk, 0, st, (const cray_triangle*)d_tris, n_tris, d_shade);
    HIP_TRY(hipMemcpyAsync(err, d_err, sizeof(err), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    if (err[0]) { set_last_error("resident build: malformed tree (code %u)", err[0]); return CRAY_ERR_INVALID; }

    for (int k = 0; k < 3; k++) { d.root_lo[k] = root.bmin[k]; d.root_hi[k] = root.bmax[k]; }
    d.root_ref = root.is_leaf ? (kLeafBit | (root.first << 3) | (root.count - 1)) : 0u;   // the root is interior record 0
    d.n_inner = n_inner;
    d.bounds_in_div_range = err[1] ? 0u : 1u;
    s->build_stats = bvh.stats;
    // hand the four arrays over to the scene, in cray_scene_upload's order
    auto adopt = [&](void* ptr, size_t bytes) { mem.release(ptr); s->allocs.push_back(ptr); s->alloc_bytes.push_back(bytes); s->bytes += bytes; };
    adopt(d_inner, (size_t)(n_inner ? n_inner : 1) * sizeof(InnerNode)); d.inner = d_inner;
    adopt(d_slots, ((size_t)n + 1) * sizeof(LeafSlot)); d.slots = d_slots;
    adopt(d_prims, (size_t)n * sizeof(cray_prim)); d.prims = d_prims;
    adopt(d_shade, (size_t)(n_tris ? n_tris : 1) * sizeof(TriShade)); d.tri_shade = d_shade;
    return CRAY_OK;
}

}  // namespace

// Flattened reference-topology BVH -> device layout (cray_device.h).
extern "C" int cray_scene_upload(cray_ctx* c, const cray_flat_scene* f, cray_scene** out) {
    if (!c || !f || !out) { set_last_error("cray_scene_upload: null argument"); return CRAY_ERR_INVALID; }
    *out = nullptr;
    if (f->abi_version != CRAY_ABI_VERSION) { set_last_error("flat scene abi_version %u != %u", f->abi_version, CRAY_ABI_VERSION); return CRAY_ERR_INVALID; }
    const bool resident = f->build_on_device != 0;   // Bvh::new runs on the GPU inside this call (build_resident below)
    if ((!resident && f->n_nodes == 0) || f->n_prims == 0 || f->n_lights == 0) { set_last_error("scene needs nodes, primitives and lights"); return CRAY_ERR_INVALID; }
    if (resident && (f->n_nodes != 0 || f->n_prim_refs != 0)) { set_last_error("build_on_device: the flat scene must not carry a tree"); return CRAY_ERR_INVALID; }
    if (4 + 8 * (uint64_t)f->max_depth > 256) { set_last_error("max_depth %u needs more than sobol_burley's 256 dimensions", f->max_depth); return CRAY_ERR_UNSUPPORTED; }
    if (f->n_prims >= (1u << 28)) { set_last_error("too many primitives for the 28-bit leaf slot index"); return CRAY_ERR_UNSUPPORTED; }
    HIP_TRY(hipSetDevice(c->device));

    // ---- storage order of the interior records and of the leaf slots.  Traversal order is defined by the child
    // references, so any order gives the same hits and counters; only locality differs.  Default: the flat scene's DFS
    // pre-order for the records and leaf (prim_refs) order for the slots.  CRAY_BVH_LAYOUT=<treelet>[:<top>] (experiments,
    // profiles/r02_experiments.md): the first <top> records breadth-first from the root, then treelets of <treelet>
    // records (breadth-first inside a treelet, treelets depth-first), leaf slots in the order the records reference them.
    uint32_t lay_treelet = 0, lay_top = 0;
    if (const char* ev = getenv("CRAY_BVH_LAYOUT")) {
        lay_treelet = (uint32_t)strtoul(ev, nullptr, 10);
        if (const char* c2 = strchr(ev, ':')) lay_top = (uint32_t)strtoul(c2 + 1, nullptr, 10);
    }
    for (uint32_t i = 0; i < f->n_nodes; i++) {
        const cray_bvh_node& nd = f->nodes[i];
        if (!nd.is_leaf && (nd.left >= f->n_nodes || nd.right >= f->n_nodes || nd.axis < 0 || nd.axis > 2)) { set_last_error("BVH node %u malformed", i); return CRAY_ERR_INVALID; }
        if (nd.is_leaf) {
            if (nd.count < 1 || nd.count > 8) { set_last_error("BVH leaf with %u primitives (supported: 1..8)", nd.count); return CRAY_ERR_UNSUPPORTED; }
            if ((uint64_t)nd.first + nd.count > f->n_prim_refs) { set_last_error("BVH leaf range out of bounds"); return CRAY_ERR_INVALID; }
        }
    }
    std::vector<uint32_t> inner_index(f->n_nodes, kNoRef);   // flat node -> interior record
    std::vector<uint32_t> leaf_first(f->n_nodes, kNoRef);    // flat leaf node -> first slot
    uint32_t n_inner = 0, n_slots = 0;
    if (lay_treelet == 0) {
        for (uint32_t i = 0; i < f->n_nodes; i++) {
            if (!f->nodes[i].is_leaf) inner_index[i] = n_inner++;
            else leaf_first[i] = f->nodes[i].first;
        }
        n_slots = f->n_prim_refs;
    } else {
        std::vector<uint32_t> roots, fifo;
        if (!f->nodes[0].is_leaf) roots.push_back(0);
        else { leaf_first[0] = 0; n_slots = f->nodes[0].count; }
        bool first = true;
        while (!roots.empty()) {
            const uint32_t r = roots.back();
            roots.pop_back();
            const uint32_t budget = first && lay_top ? lay_top : lay_treelet;
            first = false;
            fifo.clear();
            fifo.push_back(r);
            size_t head = 0;
            uint32_t taken = 0;
            while (head < fifo.size() && taken < budget) {
                const uint32_t nidx = fifo[head++];
                if (inner_index[nidx] != kNoRef) { set_last_error("BVH is not a tree (node %u reached twice)", nidx); return CRAY_ERR_INVALID; }
                inner_index[nidx] = n_inner++;
                taken++;
                for (uint32_t ch : {f->nodes[nidx].left, f->nodes[nidx].right}) {
                    if (!f->nodes[ch].is_leaf) fifo.push_back(ch);
                    else if (leaf_first[ch] == kNoRef) { leaf_first[ch] = n_slots; n_slots += f->nodes[ch].count; }
                }
            }
            for (size_t k = fifo.size(); k > head; k--) roots.push_back(fifo[k - 1]);  // the frontier: next treelets, leftmost first
        }
        if (n_slots != f->n_prim_refs) { set_last_error("BVH leaves cover %u of %u primitive references", n_slots, f->n_prim_refs); return CRAY_ERR_INVALID; }
    }

    // ---- leaf slots; interior records with both children's bounds
    std::vector<LeafSlot> slots((size_t)f->n_prim_refs + 1);  // +1: the unified 112-B record fetch of k_trace reads past an 80-B slot
    memset(&slots[f->n_prim_refs], 0, sizeof(LeafSlot));
    auto fill_slot = [&](uint32_t dst, uint32_t src_ref) -> int {
        uint32_t pi = f->prim_refs[src_ref];
        if (pi >= f->n_prims) { set_last_error("prim_refs[%u] out of range", src_ref); return CRAY_ERR_INVALID; }
        const cray_prim& p = f->prims[pi];
        LeafSlot& s = slots[dst];
        memset(&s, 0, sizeof(s));
        s.prim = pi;
        s.kind = (uint32_t)p.shape_kind;
        if (p.shape_kind == CRAY_SHAPE_TRIANGLE) {
            if (p.shape >= f->n_triangles) { set_last_error("primitive %u: bad triangle index", pi); return CRAY_ERR_INVALID; }
            const cray_triangle& t = f->triangles[p.shape];
            s.v0[0] = t.v0.x; s.v0[1] = t.v0.y; s.v0[2] = t.v0.z;
            s.e1[0] = t.e1.x; s.e1[1] = t.e1.y; s.e1[2] = t.e1.z;
            s.e2[0] = t.e2.x; s.e2[1] = t.e2.y; s.e2[2] = t.e2.z;
        } else if ((p.shape_kind == CRAY_SHAPE_SPHERE && p.shape >= f->n_spheres) || (p.shape_kind == CRAY_SHAPE_DISK && p.shape >= f->n_disks) ||
                   (p.shape_kind != CRAY_SHAPE_SPHERE && p.shape_kind != CRAY_SHAPE_DISK)) {
            set_last_error("primitive %u: bad shape", pi);
            return CRAY_ERR_INVALID;
        } else {
            const uint64_t bits = p.shape;   // the sphere / disk index, so that k_trace needs no prims[] lookup
            memcpy(&s.v0[0], &bits, 8);
        }
        return CRAY_OK;
    };
    if (lay_treelet == 0) {
        for (uint32_t i = 0; i < f->n_prim_refs; i++) { int e0 = fill_slot(i, i); if (e0) return e0; }
    } else {
        for (uint32_t i = 0; i < f->n_nodes; i++) {
            const cray_bvh_node& nd = f->nodes[i];
            if (!nd.is_leaf) continue;
            if (leaf_first[i] == kNoRef) { set_last_error("BVH leaf %u is not referenced", i); return CRAY_ERR_INVALID; }
            for (uint32_t k = 0; k < nd.count; k++) { int e0 = fill_slot(leaf_first[i] + k, nd.first + k); if (e0) return e0; }
        }
    }
    auto make_ref = [&](uint32_t node, uint32_t* ref) -> int {
        if (node >= f->n_nodes) { set_last_error("BVH child index %u out of range", node); return CRAY_ERR_INVALID; }
        const cray_bvh_node& nd = f->nodes[node];
        if (!nd.is_leaf) {
            if (inner_index[node] == kNoRef) { set_last_error("BVH node %u is not reachable from the root", node); return CRAY_ERR_INVALID; }
            *ref = inner_index[node];
            return CRAY_OK;
        }
        *ref = kLeafBit | (leaf_first[node] << 3) | (nd.count - 1);
        return CRAY_OK;
    };
    std::vector<InnerNode> inner(n_inner ? n_inner : 1);
    memset(inner.data(), 0, inner.size() * sizeof(InnerNode));
    for (uint32_t i = 0; i < f->n_nodes; i++) {
        const cray_bvh_node& nd = f->nodes[i];
        if (nd.is_leaf || inner_index[i] == kNoRef) continue;
        InnerNode& o = inner[inner_index[i]];
        const cray_bvh_node& l = f->nodes[nd.left];
        const cray_bvh_node& r = f->nodes[nd.right];
        for (int k = 0; k < 3; k++) { o.lo0[k] = l.bmin[k]; o.hi0[k] = l.bmax[k]; o.lo1[k] = r.bmin[k]; o.hi1[k] = r.bmax[k]; }
        int e;
        if ((e = make_ref(nd.left, &o.ref0))) return e;
        if ((e = make_ref(nd.right, &o.ref1))) return e;
        o.axis = (uint32_t)nd.axis;
    }

    cray_scene* s = new cray_scene();
    s->ctx = c;
    s->n_prims = f->n_prims;
    DevScene& d = s->dev;
    d.max_depth = f->max_depth; d.num_samples = f->num_samples;
    d.film_w = f->film_width; d.film_h = f->film_height;
    d.camera_type = f->camera_type; d.n_lights = f->n_lights;
    d.lens_radius = f->lens_radius; d.focal_distance = f->focal_distance;
    memcpy(d.camera_from_raster, f->camera_from_raster, sizeof(d.camera_from_raster));
    memcpy(d.world_from_camera, f->world_from_camera, sizeof(d.world_from_camera));
    int e = CRAY_OK;
    if (!resident) {
        for (int k = 0; k < 3; k++) { d.root_lo[k] = f->nodes[0].bmin[k]; d.root_hi[k] = f->nodes[0].bmax[k]; }
        e = make_ref(0, &d.root_ref);
    }
    d.n_inner = n_inner;
    d.bounds_in_div_range = 1;
    for (uint32_t i = 0; i < f->n_nodes && d.bounds_in_div_range; i++)
        for (int k = 0; k < 3; k++)
            if (!div_range_ok(f->nodes[i].bmin[k]) || !div_range_ok(f->nodes[i].bmax[k])) d.bounds_in_div_range = 0;

    // triangle shading records, in triangle-table order
    std::vector<TriShade> shade(resident ? 0 : f->n_triangles);
    for (uint32_t i = 0; i < f->n_triangles && !resident; i++) {
        const cray_triangle& t = f->triangles[i];
        TriShade& o = shade[i];
        o.n0[0] = t.n0.x; o.n0[1] = t.n0.y; o.n0[2] = t.n0.z;
        o.n01[0] = t.n01.x; o.n01[1] = t.n01.y; o.n01[2] = t.n01.z;
        o.n02[0] = t.n02.x; o.n02[1] = t.n02.y; o.n02[2] = t.n02.z;
        for (int k = 0; k < 2; k++) { o.uv0[k] = t.uv0[k]; o.uv01[k] = t.uv01[k]; o.uv02[k] = t.uv02[k]; }
    }
    // lights with their emitter geometry and Shape::area (shape.rs:504-514)
    std::vector<DevLight> lights(f->n_lights);
    for (uint32_t i = 0; i < f->n_lights && !e; i++) {
        const cray_light& l = f->lights[i];
        DevLight& o = lights[i];
        memset(&o, 0, sizeof(o));
        o.kind = l.kind;
        o.v[0] = l.v.x; o.v[1] = l.v.y; o.v[2] = l.v.z;
        o.c[0] = l.c.r; o.c[1] = l.c.g; o.c[2] = l.c.b;
        if (l.kind != CRAY_LIGHT_AREA) continue;
        if (l.prim < 0 || (uint32_t)l.prim >= f->n_prims) { set_last_error("area light %u: bad primitive", i); e = CRAY_ERR_INVALID; break; }
        const cray_prim& p = f->prims[l.prim];
        o.shape_kind = p.shape_kind; o.shape = p.shape;
        if ((p.shape_kind == CRAY_SHAPE_TRIANGLE && p.shape >= f->n_triangles) || (p.shape_kind == CRAY_SHAPE_SPHERE && p.shape >= f->n_spheres) ||
            (p.shape_kind == CRAY_SHAPE_DISK && p.shape >= f->n_disks) || p.shape_kind < CRAY_SHAPE_SPHERE || p.shape_kind > CRAY_SHAPE_DISK) {
            set_last_error("area light %u: bad shape", i); e = CRAY_ERR_INVALID; break;
        }
        if (p.shape_kind == CRAY_SHAPE_TRIANGLE) {
            const cray_triangle& t = f->triangles[p.shape];
            o.v0[0] = t.v0.x; o.v0[1] = t.v0.y; o.v0[2] = t.v0.z;
            o.e1[0] = t.e1.x; o.e1[1] = t.e1.y; o.e1[2] = t.e1.z;
            o.e2[0] = t.e2.x; o.e2[1] = t.e2.y; o.e2[2] = t.e2.z;
            o.area = len(cross(mk(t.e1.x, t.e1.y, t.e1.z), mk(t.e2.x, t.e2.y, t.e2.z))) / 2.0;
        } else if (p.shape_kind == CRAY_SHAPE_SPHERE) {
            o.area = kPi * square(f->spheres[p.shape].radius);
        } else {
            o.area = kPi * (square(f->disks[p.shape].radius) - square(f->disks[p.shape].inner_radius));
        }
    }
    for (uint32_t i = 0; i < f->n_prims && !e; i++) {
        const cray_prim& p = f->prims[i];
        if (p.material >= (int32_t)f->n_materials || p.light >= (int32_t)f->n_lights || (p.material < 0 && p.light < 0)) {
            set_last_error("primitive %u: bad material/light index", i);
            e = CRAY_ERR_INVALID;
        }
    }
    // material / texture / image tables are indexed by the kernels (and below on the host): check every index once
    for (uint32_t i = 0; i < f->n_materials && !e; i++) {
        const cray_material& m = f->materials[i];
        const int64_t nb = m.is_bsdf ? m.n_bxdfs : 1;
        if (m.n_bxdfs < 0 || m.first_bxdf < 0 || (nb > 0 && (int64_t)m.first_bxdf + nb > (int64_t)f->n_bxdfs)) {
            set_last_error("material %u: lobes [%d, %d + %lld) outside the %u bxdfs", i, m.first_bxdf, m.first_bxdf, (long long)nb, f->n_bxdfs);
            e = CRAY_ERR_INVALID;
        }
    }
    for (uint32_t i = 0; i < f->n_bxdfs && !e; i++) {
        const cray_bxdf& bx = f->bxdfs[i];
        if (bx.kind < CRAY_BXDF_LAMBERTIAN || bx.kind > CRAY_BXDF_FRESNEL_SPECULAR || bx.tex_a >= (int32_t)f->n_textures || bx.tex_b >= (int32_t)f->n_textures) {
            set_last_error("bxdf %u: bad kind or texture index", i);
            e = CRAY_ERR_INVALID;
        }
    }
    for (uint32_t i = 0; i < f->n_textures && !e; i++) {
        const cray_texture& t = f->textures[i];
        if (t.kind < CRAY_TEX_CONSTANT || t.kind > CRAY_TEX_IMAGE || (t.kind == CRAY_TEX_IMAGE && (t.image < 0 || t.image >= (int32_t)f->n_images))) {
            set_last_error("texture %u: bad kind or image index", i);
            e = CRAY_ERR_INVALID;
        }
    }
    for (uint32_t i = 0; i < f->n_images && !e; i++) {
        const cray_image& im = f->images[i];
        const uint64_t bytes = (uint64_t)im.width * im.height * 3;
        if (im.width == 0 || im.height == 0 || im.offset > f->image_pool_bytes || bytes > f->image_pool_bytes - im.offset) {
            set_last_error("image %u: %ux%u at offset %llu does not fit the %llu-byte pool", i, im.width, im.height, (unsigned long long)im.offset, (unsigned long long)f->image_pool_bytes);
            e = CRAY_ERR_INVALID;
        }
    }
    double gamma_lut[256];  // Color::from_rgb (color.rs:39-46): (c/255).powf(2.2), libm pow like the reference
    for (int i = 0; i < 256; i++) gamma_lut[i] = pow((double)i / 255.0, 2.2);

    if (resident && !e) {
        // What fill_slot checks for a scene that brings its tree, checked here for one that does not: the kernels index
        // spheres[] / disks[] / triangles[] with prims[].shape unchecked, and a sphere / disk without a box of its own would
        // silently get the all-zero box of k_prim_bounds.
        std::vector<uint8_t> boxes(f->n_prims, 0);
        for (uint32_t i = 0; i < f->n_other_bounds && !e; i++) {
            const cray_prim_bound& b = f->other_bounds[i];
            if (b.prim >= f->n_prims) { set_last_error("other_bounds[%u]: primitive %u out of range", i, b.prim); e = CRAY_ERR_INVALID; break; }
            if (f->prims[b.prim].shape_kind == CRAY_SHAPE_TRIANGLE) { set_last_error("other_bounds[%u]: primitive %u is a triangle (its box is computed on the device)", i, b.prim); e = CRAY_ERR_INVALID; break; }
            if (boxes[b.prim]++) { set_last_error("other_bounds[%u]: primitive %u already has a box", i, b.prim); e = CRAY_ERR_INVALID; break; }
            for (int k = 0; k < 3; k++)
                if (!std::isfinite(b.bmin[k]) || !std::isfinite(b.bmax[k])) { set_last_error("other_bounds[%u]: non-finite box of primitive %u", i, b.prim); e = CRAY_ERR_INVALID; }
        }
        for (uint32_t i = 0; i < f->n_prims && !e; i++) {
            const cray_prim& p = f->prims[i];
            const bool tri = p.shape_kind == CRAY_SHAPE_TRIANGLE, sph = p.shape_kind == CRAY_SHAPE_SPHERE, dsk = p.shape_kind == CRAY_SHAPE_DISK;
            if (tri && p.shape >= f->n_triangles) { set_last_error("primitive %u: bad triangle index", i); e = CRAY_ERR_INVALID; }
            else if ((!tri && !sph && !dsk) || (sph && p.shape >= f->n_spheres) || (dsk && p.shape >= f->n_disks)) { set_last_error("primitive %u: bad shape", i); e = CRAY_ERR_INVALID; }
            else if (!tri && !boxes[i]) { set_last_error("primitive %u: a sphere / disk needs its box in other_bounds (build_on_device)", i); e = CRAY_ERR_INVALID; }
        }
    }
    if (resident) {
        if (!e) e = build_resident(c, f, s);   // inner, slots, prims, tri_shade: built on the device, in this order
    } else {
        if (!e) e = upload(s, inner.data(), inner.size(), &d.inner);
        if (!e) e = upload(s, slots.data(), slots.size(), &d.slots);
        if (!e) e = upload(s, f->prims, (size_t)f->n_prims, &d.prims);
        if (!e) e = upload(s, shade.data(), shade.size(), &d.tri_shade);
    }
    if (!e) e = upload(s, f->spheres, (size_t)f->n_spheres, &d.spheres);
    if (!e) e = upload(s, f->disks, (size_t)f->n_disks, &d.disks);
    // device copy of the materials with pad_ = "some lobe reads a non-constant texture": only then does k_shade
    // need the (u, v) of a sphere / disk hit (atan2 + acos per hit otherwise computed for nothing)
    std::vector<cray_material> mats(f->materials, f->materials + f->n_materials);
    for (cray_material& m : mats) {
        m.pad_ = 0;
        if (e) break;
        const int nb = m.is_bsdf ? m.n_bxdfs : 1;
        for (int i = 0; i < nb; i++) {
            const cray_bxdf& bx = f->bxdfs[m.first_bxdf + i];
            for (int32_t t : {bx.tex_a, bx.tex_b})
                if (t >= 0 && f->textures[t].kind != CRAY_TEX_CONSTANT) m.pad_ = 1;
        }
    }
    if (!e) e = upload(s, mats.data(), mats.size(), &d.materials);
    if (!e) e = upload(s, f->bxdfs, (size_t)f->n_bxdfs, &d.bxdfs);
    if (!e) e = upload(s, f->textures, (size_t)f->n_textures, &d.textures);
    if (!e) e = upload(s, f->images, (size_t)f->n_images, &d.images);
    if (!e) e = upload(s, f->image_pool, (size_t)f->image_pool_bytes, &d.pool);
    if (!e) e = upload(s, gamma_lut, (size_t)256, &d.gamma_lut);
    if (!e) e = upload(s, lights.data(), lights.size(), &d.lights);
    if (!e) e = upload(s, f->light_cdf, (size_t)f->n_lights, &d.light_cdf);
    if (!e) e = upload(s, f->first_equal_light, (size_t)f->n_lights, &d.first_equal_light);
    if (!e) e = upload(s, g_sobol_table, (size_t)CRAY_SOBOL_SETS * CRAY_SOBOL_BITS * 4, &d.sobol);
    if (e) { cray_scene_free(s); return e; }
    // k_shade walks materials -> lobes -> textures and the light tables by dependent loads: staged in LDS when they fit
    d.n_materials = (uint32_t)mats.size(); d.n_bxdfs = f->n_bxdfs; d.n_textures = f->n_textures; d.n_images = f->n_images;
    {
        auto pad16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
        const size_t need = pad16(mats.size() * sizeof(cray_material)) + pad16((size_t)f->n_bxdfs * sizeof(cray_bxdf)) +
                            pad16((size_t)f->n_textures * sizeof(cray_texture)) + pad16((size_t)f->n_lights * sizeof(DevLight)) +
                            pad16((size_t)f->n_lights * 8) + pad16((size_t)f->n_lights * 4) +
                            pad16((size_t)f->n_images * sizeof(cray_image)) + 256 * 8;   // (the last two only where textures read images)
        const char* ev = getenv("CRAY_SHADE_LDS");   // experiments: 0 keeps the tables in global memory
        d.shade_tables_bytes = (need <= kShadeLdsTables && !(ev && ev[0] == '0')) ? (uint32_t)need : 0u;
        const size_t shapes = pad16((size_t)f->n_spheres * sizeof(cray_xf_shape)) + pad16((size_t)f->n_disks * sizeof(cray_xf_shape));
        d.n_spheres = f->n_spheres; d.n_disks = f->n_disks;
        d.shade_stage_shapes = (d.shade_tables_bytes && need + shapes <= kShadeLdsTables) ? 1u : 0u;
        if (d.shade_stage_shapes) d.shade_tables_bytes += (uint32_t)shapes;
    }
    // what this scene can make k_shade do -> the leanest instantiation that covers it
    uint32_t feat = 0;
    for (uint32_t i = 0; i < f->n_textures; i++)
        feat |= f->textures[i].kind == CRAY_TEX_CHECKERBOARD ? SF_TEX_CHECKER : (f->textures[i].kind == CRAY_TEX_IMAGE ? SF_TEX_IMAGE : 0u);
    for (uint32_t i = 0; i < f->n_bxdfs; i++) {
        static const uint32_t lobe_bit[6] = {0u, SF_OREN_NAYAR, SF_CONDUCTOR, SF_SPEC_BRDF, SF_SPEC_BTDF, SF_FRESNEL_SPEC};
        feat |= lobe_bit[f->bxdfs[i].kind];
    }
    for (uint32_t i = 0; i < f->n_materials; i++)
        if (f->materials[i].is_bsdf && f->materials[i].n_bxdfs != 1) feat |= SF_MULTI_LOBE;
    for (uint32_t i = 0; i < f->n_lights; i++) {
        const int k = f->lights[i].kind;
        if (k == CRAY_LIGHT_POINT) feat |= SF_LIGHT_POINT;
        else if (k == CRAY_LIGHT_DISTANT) feat |= SF_LIGHT_DISTANT;
        else if (k == CRAY_LIGHT_INFINITE) feat |= SF_LIGHT_INFINITE;
        else {
            const int sk = f->prims[f->lights[i].prim].shape_kind;
            feat |= sk == CRAY_SHAPE_TRIANGLE ? SF_AREA_TRI : (sk == CRAY_SHAPE_SPHERE ? SF_AREA_SPHERE : SF_AREA_DISK);
        }
    }
    if (f->n_lights > 1) feat |= SF_MANY_LIGHTS;
    for (uint32_t i = 0; i < f->n_prims; i++) {
        const int sk = f->prims[i].shape_kind;
        feat |= sk == CRAY_SHAPE_TRIANGLE ? SF_HIT_TRI : (sk == CRAY_SHAPE_SPHERE ? SF_HIT_SPHERE : SF_HIT_DISK);
    }
    s->features = feat;
    s->n_slots = resident ? f->n_prims : f->n_prim_refs;   // every primitive sits in exactly one leaf (Bvh::new partitions them)
    s->shade_variant = pick_shade_variant(feat);
    if (const char* ev = getenv("CRAY_SHADE_VARIANT")) {  // experiments: force an instantiation that still covers the scene
        const int v = atoi(ev);
        if (v >= 0 && v < kNumShadeVariants && (kShadeVariants[v] & feat) == feat) s->shade_variant = v;
    }
    *out = s;
    return CRAY_OK;
}

extern "C" void cray_render_params_default(cray_render_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->tile_width = 64; p->tile_height = 64; p->sample_batch = 8;  // craytracer.rs:232-234
    p->rank = 0; p->world_size = 1;
}

namespace {

struct PassPlan { uint32_t px0, n_pix, s_lo, s_hi; };

// The tile shard.  The reference hands its tiles to whichever worker thread is free (craytracer.rs:271-291); a static shard has to
// spread the expensive image regions itself.  Rounds 2-4 gave tile t (numbered ty outer, tx inner, like generate_tiles,
// craytracer.rs:32-33) to rank t % world: with tiles_x = 4 (mod 8) a rank then owns two of the eight column residues and nothing
// of the others, and one rank of eight carried 3 % more of the dragon than the mean.  Round 5: tile (tx, ty) belongs to rank
// (tx + s ty) % world with s the smallest stride >= 2 that is coprime with world (3 for eight ranks): in every row a rank owns
// every world-th tile, and the phase walks through ALL residues from row to row.  walk_rank_tiles visits a rank's tiles row by
// row, left to right — the order of its pixel list, of cray_film_pack and of what cray_render_gather sends.
uint64_t shard_stride(uint64_t world) {
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
    for (uint64_t s = 2; s + 1 < world; s++)
        if (gcd(s, world) == 1) return s;
    return world > 2 ? world - 1 : 1;
}
template <class F>
void walk_rank_tiles(uint64_t tiles_x, uint64_t tiles_y, uint64_t rank, uint64_t world, F&& f) {
    const uint64_t s = shard_stride(world);
    for (uint64_t ty = 0; ty < tiles_y; ty++) {
        const uint64_t phase = (s * ty) % world;
        for (uint64_t tx = (rank + world - phase) % world; tx < tiles_x; tx += world) f(tx, ty);
    }
}

// pixels of the tiles this rank owns, tile by tile, row-major inside a tile
std::vector<uint32_t> rank_pixels(uint32_t W, uint32_t H, const cray_render_params& p) {
    std::vector<uint32_t> pix;
    // 64-bit tile arithmetic: a tile edge near 2^32 must not wrap `W + tw - 1` or `tx + tw` (one tile then covers the film)
    const uint64_t tw = p.tile_width, th = p.tile_height;
    const uint64_t tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th;
    walk_rank_tiles(tiles_x, tiles_y, p.rank, p.world_size, [&](uint64_t txi, uint64_t tyi) {
        const uint64_t tx = txi * tw, ty = tyi * th;
        const uint64_t x1 = tx + tw < W ? tx + tw : W, y1 = ty + th < H ? ty + th : H;
        for (uint64_t y = ty; y < y1; y++)
            for (uint64_t x = tx; x < x1; x++) pix.push_back((uint32_t)(y * W + x));
    });
    return pix;
}

}  // namespace

// The shard map itself, for hosts with their own transport and for tests without a GPU (no context needed): the pixels
// rank `rank` of `world_size` renders, in the order cray_film_pack packs them / cray_render_gather sends them.
extern "C" int cray_tile_pixels(uint32_t W, uint32_t H, uint32_t tw, uint32_t th, uint32_t rank, uint32_t world, uint32_t* out, uint64_t capacity,
                                uint64_t* n_pixels) {
    if (!n_pixels || W == 0 || H == 0 || tw == 0 || th == 0 || world == 0 || rank >= world || (uint64_t)W * H >= (1ull << 32)) {
        set_last_error("cray_tile_pixels: bad argument");
        return CRAY_ERR_INVALID;
    }
    cray_render_params p;
    cray_render_params_default(&p);
    p.tile_width = tw; p.tile_height = th; p.rank = rank; p.world_size = world;
    if (!out) {   // the size query: tile areas summed, no map built
        const uint64_t tiles_x = ((uint64_t)W + tw - 1) / tw, tiles_y = ((uint64_t)H + th - 1) / th;
        uint64_t cnt = 0;
        walk_rank_tiles(tiles_x, tiles_y, rank, world, [&](uint64_t txi, uint64_t tyi) {
            const uint64_t tx = txi * tw, ty = tyi * th;
            cnt += ((tx + tw < W ? tx + tw : W) - tx) * ((ty + th < H ? ty + th : H) - ty);
        });
        *n_pixels = cnt;
        return CRAY_OK;
    }
    try {   // up to 16 GB of indices for a film near 2^32 pixels: nothing may be thrown across the C boundary
        const std::vector<uint32_t> pix = rank_pixels(W, H, p);
        *n_pixels = pix.size();
        if (capacity < pix.size()) { set_last_error("cray_tile_pixels: %llu pixels do not fit a buffer of %llu", (unsigned long long)pix.size(), (unsigned long long)capacity); return CRAY_ERR_INVALID; }
        if (!pix.empty()) memcpy(out, pix.data(), pix.size() * sizeof(uint32_t));
    } catch (const std::exception& ex) {
        set_last_error("cray_tile_pixels: %s", ex.what());
        return CRAY_ERR_INVALID;
    }
    return CRAY_OK;
}

namespace {

// The 64-B f32 interior records of the fast mode (bounds rounded outward), derived on the device from the f64 layout.
int ensure_inner32(cray_ctx* c, cray_scene* s) {
    if (s->dev.inner32) return CRAY_OK;
    InnerNode32* i32 = nullptr;
    const uint32_t n_inner = s->dev.n_inner ? s->dev.n_inner : 1u;
    HIP_TRY(hipMalloc((void**)&i32, (size_t)n_inner * sizeof(InnerNode32)));
    s->extra_allocs.push_back(i32);
    hipLaunchKernelGGL(k_make_inner32, dim3((n_inner + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, s->dev.inner, n_inner, i32);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    s->bytes += (size_t)n_inner * sizeof(InnerNode32);
    s->dev.inner32 = i32;
    return CRAY_OK;
}
// Exact traversal with certified f32 culling: decided per scene (range of the bounds, size of the arena), records derived on first
// use: ONE allocation, n_inner InnerNodeH followed by one LeafRecH per leaf slot (cray_device.h), references = byte offsets.
size_t arena_bytes(const cray_scene* s) {
    const uint64_t n_inner = s->dev.n_inner ? s->dev.n_inner : 1u, n_slots = (uint64_t)s->n_slots + 1u;   // + the zero pad slot of the f64 layout
    return (size_t)(n_inner * sizeof(InnerNodeH) + n_slots * sizeof(LeafSlot));
}
bool hybrid_possible(const cray_scene* s) {
    // (a reference is a 32-bit offset into the arena: scenes beyond ~30 M triangles read the f64 records)
    return s->dev.bounds_in_div_range && hyb_scene_ok(s->dev.root_lo, s->dev.root_hi) && arena_bytes(s) < ((size_t)1 << 32);
}
int ensure_hybrid(cray_ctx* c, cray_scene* s, int level) {
    if (level <= 0 || !s->hybrid_ok) return CRAY_OK;
    if (s->dev.innerh) return CRAY_OK;
    const uint32_t n_inner = s->dev.n_inner ? s->dev.n_inner : 1u, n_slots = s->n_slots + 1u;
    const size_t bytes = arena_bytes(s);
    char* arena = nullptr;
    HIP_TRY(hipMalloc((void**)&arena, bytes));
    s->extra_allocs.push_back(arena);
    const uint32_t leaf_base = n_inner * (uint32_t)sizeof(InnerNodeH);
    hipLaunchKernelGGL(k_make_innerh, dim3((n_inner + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, s->dev.inner, n_inner, reinterpret_cast<InnerNodeH*>(arena), leaf_base);
    HIP_TRY(hipMemcpyAsync(arena + leaf_base, s->dev.slots, (size_t)n_slots * sizeof(LeafSlot), hipMemcpyDeviceToDevice, c->stream));   // the slots as they are
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    s->bytes += bytes;
    s->dev.innerh = reinterpret_cast<const InnerNodeH*>(arena);
    s->dev.root_ref_h = href_of(s->dev.root_ref, leaf_base);
    return CRAY_OK;
}
// The records this call's traversal launches read.  Pinned by CRAY_HYBRID / ctx->hybrid >= 0 or CRAY_RECORDS_B0 / _REST; otherwise
// the scene's choice (cray_scene::chosen_*).  `want_probe` (out): no choice exists yet and this call is big enough to make one —
// the caller runs probe_trace_records before its first pass.  Never touches the choice itself.
int choose_trace_records(cray_ctx* c, cray_scene* s, bool counting, size_t n_paths, bool* want_probe) {
    if (want_probe) *want_probe = false;
    s->hybrid_ok = hybrid_possible(s);
    if (!s->hybrid_ok || s->needs_deep) { s->use_b0 = s->use_rest = 0; return CRAY_OK; }   // (the instantiations with the third stack level read f64 records)
    if (c->hybrid >= 0) {
        s->use_b0 = s->use_rest = c->hybrid;
    } else if (counting) {
        s->use_b0 = s->use_rest = 0;
    } else if (c->records_b0 >= 0 && c->records_rest >= 0) {
        s->use_b0 = c->records_b0; s->use_rest = c->records_rest;   // pinned per launch kind (profiling passes: tools/profile_round.sh)
    } else if (s->chosen_b0 >= 0) {
        s->use_b0 = s->chosen_b0; s->use_rest = s->chosen_rest;
    } else if (n_paths < ((size_t)1 << 21) || !want_probe) {
        s->use_b0 = s->use_rest = 0;   // too small to tell anything: f64 records, nothing chosen
    } else {
        s->use_b0 = s->use_rest = 1;   // both kinds of records are needed for the probe
        *want_probe = true;
    }
    int e = ensure_hybrid(c, s, s->use_b0);
    if (!e) e = ensure_hybrid(c, s, s->use_rest);
    return e;
}
// The f32 triangle records of the fast mode, derived the first time a fast frame is asked for.
int ensure_fast_layout(cray_ctx* c, cray_scene* s) {
    if (int e = ensure_inner32(c, s)) return e;
    if (s->dev.slots32) return CRAY_OK;
    LeafSlot32* s32 = nullptr;
    const uint32_t n_slots = s->n_slots + 1u;
    HIP_TRY(hipMalloc((void**)&s32, ((size_t)n_slots + 1) * sizeof(LeafSlot32)));   // + one more: the 4-load fetch reads 16 B past a slot
    s->extra_allocs.push_back(s32);
    HIP_TRY(hipMemsetAsync(s32, 0, ((size_t)n_slots + 1) * sizeof(LeafSlot32), c->stream));
    hipLaunchKernelGGL(k_make_slots32, dim3((n_slots + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, s->dev.slots, n_slots, s32);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    s->bytes += ((size_t)n_slots + 1) * sizeof(LeafSlot32);
    s->dev.slots32 = s32;
    return CRAY_OK;
}

// One k_shade launch.  The grid is what the chip holds of THIS instantiation (the occupancy the runtime reports for its registers and
// LDS, 2 - 4 blocks per CU): the kernel hands its tiles out dynamically, so more blocks than that would only queue behind the resident
// ones (CRAY_SHADE_BLOCKS_PER_CU > 0 overrides the count for experiments).
template <uint32_t F, int MODE, class... A>
void launch_shade(const cray_ctx* c, size_t n_work, hipStream_t st, A... args) {
    static const int occ = resident_blocks(reinterpret_cast<const void*>(&k_shade<F, MODE>), kBlock, 2);
    const int per_cu = c->shade_blocks_per_cu > 0 ? c->shade_blocks_per_cu : occ;
    hipLaunchKernelGGL((k_shade<F, MODE>), dim3(grid_for(c, n_work, per_cu)), dim3(kBlock), 0, st, args...);
}

template <int I>
struct ShadeLaunch {
    template <class... A>
    static void go(int variant, int mode, bool lds_tables, const cray_ctx* c, size_t n_work, hipStream_t st, A... args) {
        if (mode != 0) {  // the selectable alternatives of the reference (simple integrator, uniform sampler): all-features kernel
            if (mode == kModeSimple) launch_shade<SF_ALL, kModeSimple>(c, n_work, st, args...);
            else if (mode == kModeUniform) launch_shade<SF_ALL, kModeUniform>(c, n_work, st, args...);
            else if (mode == kModeIndependent) launch_shade<SF_ALL, kModeIndependent>(c, n_work, st, args...);
            else if (mode == (kModeSimple | kModeIndependent)) launch_shade<SF_ALL, kModeSimple | kModeIndependent>(c, n_work, st, args...);
            else launch_shade<SF_ALL, kModeSimple | kModeUniform>(c, n_work, st, args...);
            return;
        }
        if (variant == I) {
            if (lds_tables) launch_shade<kShadeVariants[I], kModeLdsTables>(c, n_work, st, args...);
            else launch_shade<kShadeVariants[I], 0>(c, n_work, st, args...);
        } else ShadeLaunch<I + 1>::go(variant, mode, lds_tables, c, n_work, st, args...);
    }
};
template <>
struct ShadeLaunch<kNumShadeVariants> {
    template <class... A>
    static void go(int, int, bool, const cray_ctx*, size_t, hipStream_t, A...) {}
};

// The traversal launches: which records they read (v: 0 f64, 1 certified f32 culling) and whether the scene's few
// sphere / disk records are staged in LDS (shp: those instantiations get bit 14 of refill_min) pick the instantiation.
// Counting launches exist for v 0 and 1 only; the f64 any-hit launch keeps its LDS for the work sharing.
// deep: the context has the third stack level (a frame overflowed LDS + scratch): the instantiations that carry it, on f64 records.
template <bool ANY, bool COUNT, class... A>
void launch_trace(int v, bool shp, bool shp_hyb, bool deep, int grid, int grid_hyb, hipStream_t st, A... a) {
    if (v == 1) shp = shp_hyb;   // (the five-waves instantiations have room for fewer sphere / disk records in LDS: kTraceLdsShapesHyb)
    const dim3 g((!deep && v == 1) ? grid_hyb : grid), b(kBlock);   // (the certified-f32 instantiations run five blocks per CU, the others four: trace_waves)
    if (deep) { hipLaunchKernelGGL((k_trace<ANY, COUNT, 0, false, true>), g, b, 0, st, a...); return; }
    if constexpr (COUNT) {
        if (v == 1) hipLaunchKernelGGL((k_trace<ANY, true, 1>), g, b, 0, st, a...);
        else hipLaunchKernelGGL((k_trace<ANY, true, 0>), g, b, 0, st, a...);
    } else {
        if (v == 1 && shp) hipLaunchKernelGGL((k_trace<ANY, false, 1, true>), g, b, 0, st, a...);
        else if (v == 1) hipLaunchKernelGGL((k_trace<ANY, false, 1>), g, b, 0, st, a...);
        else if (shp && !ANY) hipLaunchKernelGGL((k_trace<false, false, 0, true>), g, b, 0, st, a...);
        else hipLaunchKernelGGL((k_trace<ANY, false, 0>), g, b, 0, st, a...);
    }
}
template <class... A>
void launch_mixed(int v, bool shp, bool shp_hyb, bool tail, bool deep, int grid, int grid_hyb, hipStream_t st, A... a) {
    if (v == 1) shp = shp_hyb;
    const dim3 g((!deep && v == 1) ? grid_hyb : grid), b(kBlock);
    if (deep) { hipLaunchKernelGGL((k_trace_mixed<0, false, false, true>), g, b, 0, st, a...); return; }
    // a mixed launch is launched twice when the small-launch instantiation is on (Counters::tail_rays != 0): each of the two
    // returns at once unless the launch has its size — the host does not know the queue lengths
#ifdef CRAY_WITH_EXPERIMENTS
    if (tail) {
        if (shp) hipLaunchKernelGGL((k_trace_mixed<0, true, true>), g, b, 0, st, a...);
        else hipLaunchKernelGGL((k_trace_mixed<0, false, true>), g, b, 0, st, a...);
    }
#else
    (void)tail;
#endif
    if (v == 1 && shp) hipLaunchKernelGGL((k_trace_mixed<1, true>), g, b, 0, st, a...);
    else if (v == 1) hipLaunchKernelGGL(k_trace_mixed<1>, g, b, 0, st, a...);
    else if (shp) hipLaunchKernelGGL((k_trace_mixed<0, true>), g, b, 0, st, a...);
    else hipLaunchKernelGGL(k_trace_mixed<0>, g, b, 0, st, a...);
}
bool shapes_fit_lds(const cray_ctx* c, const DevScene& d) {
    return c->lds_shapes && d.n_spheres + d.n_disks > 0 && d.n_spheres + d.n_disks <= kTraceLdsShapes;
}

int run_pass(cray_ctx* c, cray_scene* s, const cray_render_params& prm, const PassPlan& pp, EventTimer* tm) {
    const DevScene& d = s->dev;
    const uint32_t spp_pass = pp.s_hi - pp.s_lo;
    const uint32_t n_paths = pp.n_pix * spp_pass;
    hipStream_t st = c->stream;
    Counters* ctr = c->counters;
    const bool count = prm.count_traversal != 0;

    if (tm) { int e = tm->begin(FAM_OTHER); if (e) return e; }
    const uint32_t uni_nx = prm.sampler == CRAY_SAMPLER_UNIFORM ? prm.uniform_nx : 0u, uni_ny = prm.sampler == CRAY_SAMPLER_UNIFORM ? prm.uniform_ny : 0u;
    const uint32_t independent = prm.sampler == CRAY_SAMPLER_INDEPENDENT ? 1u : 0u;
    const int mode = (prm.integrator == CRAY_INTEGRATOR_SIMPLE ? kModeSimple : 0) | (uni_nx ? kModeUniform : 0) | (independent ? kModeIndependent : 0);
    static const int occ_raygen = resident_blocks(reinterpret_cast<const void*>(&k_raygen), kBlock, 4);
    hipLaunchKernelGGL(k_raygen, dim3(grid_for(c, n_paths, 4 * occ_raygen)), dim3(kBlock), 0, st, d, c->ps, c->pix_order, pp.px0, n_paths, spp_pass, pp.s_lo, prm.seed, uni_nx, uni_ny, independent);
    if (tm) { int e = tm->end(); if (e) return e; }

    // Launch sequence of a pass.  The shadow rays of bounce b and the path segments of bounce b+1 both depend on
    // k_shade(b) only, so they share ONE persistent launch (k_trace_mixed); with traversal counting every query
    // kind keeps its own launch so that the counters stay per kind.
    const bool mixed = !count && c->mix_trace;
    const bool fast = prm.precision == CRAY_PRECISION_F32_TRAVERSAL;   // check_render_args refuses it together with counting
    const bool deep = s->needs_deep && c->deep_depth != 0;   // this scene needs the third stack level: its instantiations (f64 records) trace the pass
    const bool shp = shapes_fit_lds(c, d) && !deep;
    const bool shp_hyb = shp && d.n_spheres + d.n_disks <= kTraceLdsShapesHyb;
    const unsigned int shp_bit = (shp ? 0x4000u : 0u) | (c->tail_seg ? 0x2000u : 0u);   // + whether small launches split segments: both ride in refill_min
    const unsigned int trace_all = prm.count_traversal == 1 ? 1u : 0u;  // 2 = count, but keep skipping zero-term shadow rays
    for (uint32_t b = 0; b < d.max_depth; b++) {
        // live state of bounce b sits in view b & 1 (k_raygen wrote view 0), k_shade moves the survivors to the other one
        const PathState& ps_b = (b & 1) ? c->ps1 : c->ps;
        const PathState& ps_n = (b & 1) ? c->ps : c->ps1;
        const uint32_t* q = b == 0 ? nullptr : c->queue[b & 1];
        const unsigned int* nq = b == 0 ? nullptr : ((b & 1) ? &ctr->n_active1 : &ctr->n_active0);
        uint32_t* q_next = c->queue[(b + 1) & 1];
        unsigned int* n_next = ((b + 1) & 1) ? &ctr->n_active1 : &ctr->n_active0;
        // an upper bound of the live paths is not known on the host: size the grids for the pass
        const int g_trace = grid_for(c, n_paths, c->trace_blocks_per_cu);  // persistent: 4 blocks x 4 waves per CU at 4 waves/SIMD
        const int g_trace_hyb = grid_for(c, n_paths, c->trace_blocks_per_cu_hyb);   // 5 x 4 with the certified-f32 records
        // (a SMALL bounce-0 launch — a rank's share of a sharded frame — is mostly ramp and drain, and a fifth block per CU only adds waves to
        // both: an eighth of configs[2], 16.6 M paths: 3.04 ms on four blocks, 3.36 on five; a quarter: 5.8 either way; the whole frame 21.9 / 19.9)
        const int g_b0_hyb = (n_paths < ((size_t)24 << 20) && c->trace_blocks_per_cu_hyb > c->trace_blocks_per_cu) ? grid_for(c, n_paths, c->trace_blocks_per_cu) : g_trace_hyb;
        const int g_trace32 = grid_for(c, n_paths, c->trace32_blocks_per_cu);

        if (!mixed || b == 0) {
            HIP_TRY(hipMemsetAsync(&ctr->trace_head, 0, sizeof(unsigned int), st));
            if (tm) { int e = tm->begin(FAM_CLOSEST); if (e) return e; }
            const int v_closest = b == 0 ? s->use_b0 : s->use_rest;
            if (count) launch_trace<false, true>(v_closest, shp, shp_hyb, deep, g_trace, g_trace_hyb, st, d, ps_b, q, nq, n_paths, (const double*)nullptr, ctr, &ctr->trace_head, c->refill_min);
            else if (fast) hipLaunchKernelGGL((k_trace32<kTraceClosest>), dim3(g_trace32), dim3(kBlock), 0, st, d, ps_b, q, nq, n_paths, (const uint32_t*)nullptr, (const unsigned int*)nullptr, ctr, &ctr->trace_head, b == 0 ? c->refill_min_b0 : c->refill_min, b == 0 ? 1u : 0u);
            else launch_trace<false, false>(v_closest, shp, shp_hyb, deep, g_trace, b == 0 ? g_b0_hyb : g_trace_hyb, st, d, ps_b, q, nq, n_paths, (const double*)nullptr, ctr, &ctr->trace_head, (b == 0 ? c->refill_min_b0 : c->refill_min) | shp_bit);
            if (tm) { int e = tm->end(); if (e) return e; }
        }

        // n_next, n_shadow, trace_head and shade_head in one fill: four consecutive words of Counters for either parity (cray_device.h)
        static_assert(offsetof(Counters, n_shadow) == offsetof(Counters, n_active0) + 4 && offsetof(Counters, n_active1) == offsetof(Counters, n_shadow) + 12,
                      "the words a bounce zeroes are consecutive");
        HIP_TRY(hipMemsetAsync(((b + 1) & 1) ? &ctr->n_shadow : &ctr->n_active0, 0, 4 * sizeof(unsigned int), st));
        if (tm) { int e = tm->begin(FAM_SHADE); if (e) return e; }
        ShadeLaunch<0>::go(s->shade_variant, mode, d.shade_tables_bytes != 0, c, (size_t)n_paths, st, d, ps_b, ps_n, q, nq, n_paths, b, spp_pass, pp.s_lo, q_next, n_next,
                           c->shadow_queue, &ctr->n_shadow, ctr, trace_all, uni_nx, uni_ny, c->pix_order, pp.px0, prm.seed);
        if (tm) { int e = tm->end(); if (e) return e; }

        if (mixed && b + 1 < d.max_depth) {
            if (tm) { int e = tm->begin(FAM_MIXED); if (e) return e; }
            if (fast) hipLaunchKernelGGL((k_trace32<kTraceMixed>), dim3(g_trace32), dim3(kBlock), 0, st, d, ps_n, (const uint32_t*)c->shadow_queue, (const unsigned int*)&ctr->n_shadow, 0u,
                                         (const uint32_t*)q_next, (const unsigned int*)n_next, ctr, &ctr->trace_head, c->refill_min, 0u);
            else launch_mixed(s->use_rest, shp, shp_hyb, c->tail_res != nullptr, deep, g_trace, g_trace_hyb, st, d, ps_n, (const uint32_t*)c->shadow_queue, (const unsigned int*)&ctr->n_shadow, (const uint32_t*)q_next,
                              (const unsigned int*)n_next, ctr, &ctr->trace_head,
                              (s->use_rest ? c->refill_min_hyb | (c->refill_min_any_hyb << 16) : c->refill_min | (c->refill_min_any << 16)) | (c->leaf_min << 7) | (c->steal ? 0x8000u : 0u) | shp_bit);
            if (tm) { int e = tm->end(); if (e) return e; }
        } else {
            if (tm) { int e = tm->begin(FAM_ANY); if (e) return e; }
            if (count) launch_trace<true, true>(s->use_rest, shp, shp_hyb, deep, g_trace, g_trace_hyb, st, d, ps_n, (const uint32_t*)c->shadow_queue, (const unsigned int*)&ctr->n_shadow, 0u, (const double*)nullptr, ctr, &ctr->trace_head, c->refill_min);
            else if (fast) hipLaunchKernelGGL((k_trace32<kTraceAny>), dim3(g_trace32), dim3(kBlock), 0, st, d, ps_n, (const uint32_t*)c->shadow_queue, (const unsigned int*)&ctr->n_shadow, 0u,
                                              (const uint32_t*)nullptr, (const unsigned int*)nullptr, ctr, &ctr->trace_head, c->refill_min, 0u);
            // (the any-hit launch of the last bounce is all drain: the f64 instantiation, which shares work between lanes, beats
            // the f32 culling there — 0.22 against 0.47 ms at an eighth of configs[2])
            else launch_trace<true, false>((s->use_rest == 1 && c->steal) ? 0 : s->use_rest, shp, shp_hyb, deep, g_trace, g_trace_hyb, st, d, ps_n, (const uint32_t*)c->shadow_queue, (const unsigned int*)&ctr->n_shadow, 0u, (const double*)nullptr, ctr, &ctr->trace_head, c->refill_min_any | (c->steal ? 0x8000u : 0u) | shp_bit);
            if (tm) { int e = tm->end(); if (e) return e; }
        }
        if (c->log_queues) {   // diagnostics only: a host round trip per bounce
            unsigned int nn = 0, ns = 0;
            HIP_TRY(hipMemcpyAsync(&nn, n_next, sizeof(nn), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&ns, &ctr->n_shadow, sizeof(ns), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            fprintf(stderr, "[cray] bounce %u: %u paths continue, %u shadow rays traced (pass of %u paths)\n", b, nn, ns, n_paths);
            if (count) {   // running totals of the traversal counters (differences between lines = this bounce's queries)
                Counters hc;
                HIP_TRY(hipMemcpy(&hc, ctr, sizeof(hc), hipMemcpyDeviceToHost));
                fprintf(stderr, "[cray]   so far: closest %llu rays %llu nodes %llu prims; shadow %llu rays %llu nodes %llu prims\n", hc.closest_rays, hc.closest_nodes,
                        hc.closest_prims, hc.shadow_rays, hc.shadow_nodes, hc.shadow_prims);
            }
        }
    }
    if (tm) { int e = tm->begin(FAM_OTHER); if (e) return e; }
    {
        static const int occ_film = resident_blocks(reinterpret_cast<const void*>(&k_film), 64, 16);
        const size_t groups = ((size_t)pp.n_pix + 63) / 64, cap = (size_t)c->n_cu * occ_film * 2;  // one wave per block
        const int g_film = (int)(groups < cap ? (groups ? groups : 1) : cap);
        hipLaunchKernelGGL(k_film, dim3(g_film), dim3(64), 0, st, c->ps, c->pix_order, pp.px0, pp.n_pix, spp_pass, pp.s_lo,
                           prm.sample_batch, c->film, ctr);
    }
    if (tm) { int e = tm->end(); if (e) return e; }
    HIP_TRY(hipGetLastError());
    return CRAY_OK;
}

int check_render_args(cray_ctx* c, cray_scene* s, const cray_render_params* p) {
    if (!c || !s || !p) { set_last_error("cray_render: null argument"); return CRAY_ERR_INVALID; }
    if (s->ctx != c) { set_last_error("scene was uploaded through a different context"); return CRAY_ERR_INVALID; }
    if (p->tile_width == 0 || p->tile_height == 0 || p->sample_batch == 0 || p->world_size == 0 || p->rank >= p->world_size) {
        set_last_error("cray_render: bad tile/batch/rank parameters");
        return CRAY_ERR_INVALID;
    }
    if (p->precision > CRAY_PRECISION_F32_TRAVERSAL) { set_last_error("cray_render: unknown precision"); return CRAY_ERR_INVALID; }
    if (p->precision != CRAY_PRECISION_F64 && p->count_traversal) {
        set_last_error("cray_render: the traversal counters are defined by the reference's f64 traversal; not available in the f32 fast mode");
        return CRAY_ERR_INVALID;
    }
    if (p->integrator > CRAY_INTEGRATOR_SIMPLE || p->sampler > CRAY_SAMPLER_INDEPENDENT) { set_last_error("cray_render: unknown integrator / sampler"); return CRAY_ERR_INVALID; }
    if (p->sampler == CRAY_SAMPLER_UNIFORM) {
        // UniformSampler::num_samples() = nx * ny is what `render` divides by (craytracer.rs:235, 255): it must be the scene's
        if (p->uniform_nx == 0 || p->uniform_ny == 0 || (uint64_t)p->uniform_nx * p->uniform_ny != s->dev.num_samples) {
            set_last_error("cray_render: UniformSampler %u x %u does not give the scene's %u samples", p->uniform_nx, p->uniform_ny, s->dev.num_samples);
            return CRAY_ERR_INVALID;
        }
    } else if (p->sampler == CRAY_SAMPLER_SOBOL && s->dev.num_samples > 65536) { set_last_error("sobol_burley indexes at most 2^16 samples"); return CRAY_ERR_UNSUPPORTED; }
    if (p->integrator == CRAY_INTEGRATOR_SIMPLE && p->sampler == CRAY_SAMPLER_SOBOL && 4 + 7 * (uint64_t)s->dev.max_depth > 256) {
        set_last_error("max_depth %u needs more than sobol_burley's 256 dimensions", s->dev.max_depth);
        return CRAY_ERR_UNSUPPORTED;
    }
    return CRAY_OK;
}

// plan passes: whole sample batches x pixel blocks, ascending batches outermost per pixel block
std::vector<PassPlan> plan(size_t capacity, uint32_t n_pix, uint32_t s_begin, uint32_t s_end, uint32_t batch) {
    std::vector<PassPlan> out;
    if (n_pix == 0 || s_end <= s_begin) return out;
    size_t per_batch_all = (size_t)n_pix * batch;
    if (per_batch_all <= capacity) {
        uint32_t k = (uint32_t)(capacity / per_batch_all);  // batches per pass
        uint32_t s = s_begin;
        while (s < s_end) {
            uint32_t first_batch_end = (s / batch + 1) * batch;
            uint32_t hi = first_batch_end + (k - 1) * batch;
            if (hi > s_end) hi = s_end;
            out.push_back(PassPlan{0, n_pix, s, hi});
            s = hi;
        }
    } else {
        uint32_t chunk = (uint32_t)(capacity / batch);
        for (uint32_t px = 0; px < n_pix; px += chunk) {
            uint32_t np = n_pix - px < chunk ? n_pix - px : chunk;
            uint32_t s = s_begin;
            while (s < s_end) {
                uint32_t hi = (s / batch + 1) * batch;
                if (hi > s_end) hi = s_end;
                out.push_back(PassPlan{px, np, s, hi});
                s = hi;
            }
        }
    }
    return out;
}

}  // namespace

namespace {

template <class T>
int ensure_buffer(T** buf, size_t* have, size_t want) {
    if (*have >= want && *buf) return CRAY_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *have = 0;
    HIP_TRY(hipMalloc((void**)buf, (want ? want : 1) * sizeof(T)));
    *have = want;
    return CRAY_OK;
}

// The pixel list of (film, tiles, rank, world) stays on the device between frames.
int ensure_pix_list(cray_ctx* c, uint32_t W, uint32_t H, const cray_render_params& prm) {
    const uint64_t pix_key[6] = {W, H, prm.tile_width, prm.tile_height, prm.rank, prm.world_size};
    if (c->pix_count_valid && memcmp(pix_key, c->pix_key, sizeof(pix_key)) == 0) return CRAY_OK;
    std::vector<uint32_t> pix;
    try { pix = rank_pixels(W, H, prm); } catch (const std::exception& ex) { set_last_error("tile map: %s", ex.what()); return CRAY_ERR_INVALID; }
    c->pix_count_valid = false;
    int e = ensure_buffer(&c->pix_list, &c->pix_capacity, pix.size());
    if (e) return e;
    if (!pix.empty()) HIP_TRY(hipMemcpyAsync(c->pix_list, pix.data(), pix.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // pix is a local vector
    memcpy(c->pix_key, pix_key, sizeof(pix_key));
    c->pix_count = pix.size();
    c->pix_count_valid = true;
    c->order_valid = false;
    return CRAY_OK;
}

// The rank's tiles in the order of their cost (k_tile_probe), most expensive first: the pixel list the kernels of a frame read
// (c->pix_order).  Computed once per (scene, pixel list) — ~1 ms: a probe launch, 8 bytes per tile to the host, a sort, the
// permuted list back — and kept; frames too small for a tail to matter and lists of few tiles render in the canonical order.
int ensure_tile_order(cray_ctx* c, cray_scene* s, const cray_render_params& prm, size_t n_paths) {
    c->pix_order = c->pix_list;
    const uint32_t W = s->dev.film_w, H = s->dev.film_h;
    if (!c->tile_order || n_paths < ((size_t)1 << 21) || s->dev.n_inner == 0) return CRAY_OK;
    if (c->order_valid && c->order_scene == s->uid && memcmp(c->order_key, c->pix_key, sizeof(c->order_key)) == 0) { c->pix_order = c->pix_render; return CRAY_OK; }
    c->order_valid = false;
    try {
        // the rank's tiles, as rank_pixels walks them
        std::vector<uint32_t> rect, start;
        const uint64_t tw = prm.tile_width, th = prm.tile_height;
        const uint64_t tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th;
        uint64_t at = 0;
        walk_rank_tiles(tiles_x, tiles_y, prm.rank, prm.world_size, [&](uint64_t txi, uint64_t tyi) {
            const uint64_t tx = txi * tw, ty = tyi * th;
            const uint64_t x1 = tx + tw < W ? tx + tw : W, y1 = ty + th < H ? ty + th : H;
            rect.push_back((uint32_t)tx); rect.push_back((uint32_t)ty); rect.push_back((uint32_t)(x1 - tx)); rect.push_back((uint32_t)(y1 - ty));
            start.push_back((uint32_t)at);
            at += (x1 - tx) * (y1 - ty);
        });
        const size_t n_tiles = start.size();
        if (n_tiles < 64 || at != c->pix_count) return CRAY_OK;
        start.push_back((uint32_t)at);
        DevMem mem;
        uint32_t* d_rect;
        unsigned long long* d_cost;
        HIP_TRY(mem.get(&d_rect, rect.size()));
        HIP_TRY(mem.get(&d_cost, n_tiles));
        HIP_TRY(hipMemcpyAsync(d_rect, rect.data(), rect.size() * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_tile_probe, dim3((unsigned int)n_tiles), dim3(64), 0, c->stream, s->dev, (const uint32_t*)d_rect, (uint32_t)n_tiles, d_cost);
        std::vector<unsigned long long> cost(n_tiles);
        HIP_TRY(hipMemcpyAsync(cost.data(), d_cost, n_tiles * 8, hipMemcpyDeviceToHost, c->stream));
        std::vector<uint32_t> canon(c->pix_count);
        HIP_TRY(hipMemcpyAsync(canon.data(), c->pix_list, c->pix_count * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> idx(n_tiles);
        for (size_t i = 0; i < n_tiles; i++) idx[i] = (uint32_t)i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
        std::vector<uint32_t> out;
        out.reserve(c->pix_count);
        for (uint32_t i : idx) out.insert(out.end(), canon.begin() + start[i], canon.begin() + start[i + 1]);
        int e = ensure_buffer(&c->pix_render, &c->pix_render_capacity, out.size());
        if (e) return e;
        HIP_TRY(hipMemcpyAsync(c->pix_render, out.data(), out.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));   // `out` is a local vector
    } catch (const std::exception& ex) { set_last_error("tile order: %s", ex.what()); return CRAY_ERR_INVALID; }
    c->order_scene = s->uid;
    memcpy(c->order_key, c->pix_key, sizeof(c->order_key));
    c->order_valid = true;
    c->pix_order = c->pix_render;
    return CRAY_OK;
}

// Which records are faster for THIS scene, measured before its first frame (round 5; rounds 4 timed a scene's first two frames, so a
// host that renders one frame per process — the reference's own usage, craytracer.rs:336-372 — never got the faster records).
// A probe pass is an ordinary pass over a sample of the rank's pixels (every k-th of its list, 2 Mi paths) whose film is thrown
// away: the launch sequence, the queues and the path pool are the frame's own.  f32 culling first (unmeasured: first launches of the
// instantiations, cold caches), then f64 / f32 / f64 / f32, the best time per kind of records and launch kind counts; the f32
// culling has to win by more than 1 % to replace the plain records.  ~3 % of one configs[2] frame, once per scene.
// Ranks probe their own pixels: their choices may differ, their films cannot (both kinds of records are exact).
__global__ void __launch_bounds__(kBlock) k_stride_list(const uint32_t* __restrict__ in, uint32_t n_out, uint32_t stride, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_out) out[i] = in[(size_t)i * stride];
}
int probe_trace_records(cray_ctx* c, cray_scene* s, const cray_render_params& prm, uint32_t n_pix_rank, uint32_t s_begin, uint32_t s_end) {
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spp = s_end - s_begin;
    if (spp > prm.sample_batch && prm.sample_batch) spp = prm.sample_batch;
    size_t want_paths = (size_t)1 << 21;
    if (want_paths > c->capacity) want_paths = c->capacity;
    uint32_t n_pix = (uint32_t)(want_paths / spp);
    if (n_pix > n_pix_rank) n_pix = n_pix_rank;
    if (n_pix == 0) { s->use_b0 = s->use_rest = 0; return CRAY_OK; }
    const uint32_t stride = n_pix_rank / n_pix;
    DevMem mem;
    uint32_t* d_list;
    HIP_TRY(mem.get(&d_list, n_pix));
    hipLaunchKernelGGL(k_stride_list, dim3((n_pix + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->pix_order, n_pix, stride, d_list);
    const uint32_t* const frame_order = c->pix_order;
    c->pix_order = d_list;
    const PassPlan pp{0u, n_pix, s_begin, s_begin + spp};
    double best[2][2] = {{1e300, 1e300}, {1e300, 1e300}};
    bool overflow = false;
    int e = CRAY_OK;
    static const int order[5] = {1, 0, 1, 0, 1};
    for (int k = 0; k < 5 && !e && !overflow; k++) {
        const int v = order[k];
        s->use_b0 = s->use_rest = v;
        if ((e = reset_counters(c))) break;
        EventTimer timer(c);
        if ((e = run_pass(c, s, prm, pp, &timer))) break;
        Counters h;
        if (hipMemcpyAsync(&h, c->counters, sizeof(h), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            set_last_error("probe pass failed: %s", hipGetErrorString(hipGetLastError()));
            e = CRAY_ERR_HIP;
            break;
        }
        double ms[FAM_COUNT]; uint32_t launches[FAM_COUNT];
        if ((e = timer.collect(ms, launches))) break;
        overflow = h.stack_overflow != 0;   // (the frame itself will report it and come back with the deeper stack: choose then)
        if (k == 0) continue;
        const double t_b0 = ms[FAM_CLOSEST], t_rest = ms[FAM_MIXED] + ms[FAM_ANY];
        if (t_b0 < best[v][0]) best[v][0] = t_b0;
        if (t_rest < best[v][1]) best[v][1] = t_rest;
    }
    c->pix_order = frame_order;
    if (e) return e;
    if (overflow) { s->use_b0 = s->use_rest = 0; return CRAY_OK; }
    for (int v = 0; v < 2; v++) for (int k = 0; k < 2; k++) s->tune_ms[v][k] = best[v][k];
    s->chosen_b0 = best[1][0] < 0.99 * best[0][0] ? 1 : 0;
    s->chosen_rest = (best[0][1] > 0.0 && best[1][1] < 0.99 * best[0][1]) ? 1 : 0;
    s->use_b0 = s->chosen_b0; s->use_rest = s->chosen_rest;
    s->probe_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (c->log_queues)
        fprintf(stderr, "[cray] traversal records chosen for this scene from %u probe pixels x %u samples in %.1f ms: bounce 0 %s (%.3f vs %.3f ms), other launches %s (%.3f vs %.3f ms)\n",
                n_pix, spp, s->probe_ms, s->chosen_b0 ? "f32 culling" : "f64", best[1][0], best[0][0], s->chosen_rest ? "f32 culling" : "f64", best[1][1], best[0][1]);
    return CRAY_OK;
}

// Every pass of this rank's share of the frame, accumulated into c->film (sums over samples, not yet divided by
// num_samples); fills everything of `stats` but `seconds`.  Local to the rank: no collective in here, so a rank that
// has to repeat its frame with the deep traversal stack does not disturb the others.
int render_local(cray_ctx* c, cray_scene* s, const cray_render_params* prm, cray_stats* stats) {
    const DevScene& d = s->dev;
    const uint32_t W = d.film_w, H = d.film_h;
    uint32_t s_begin = prm->sample_begin, s_end = prm->sample_end;
    if (s_begin == 0 && s_end == 0) s_end = d.num_samples;
    if (s_end > d.num_samples) s_end = d.num_samples;
    int e;
    if ((e = ensure_pix_list(c, W, H, *prm))) return e;
    if (prm->precision == CRAY_PRECISION_F32_TRAVERSAL && (e = ensure_fast_layout(c, s))) return e;
    const size_t n_pix_rank = c->pix_count;
    bool want_probe = false;   // no records chosen for this scene yet, and this frame is big enough to choose them
    if ((e = choose_trace_records(c, s, prm->count_traversal != 0 || prm->precision != CRAY_PRECISION_F64,
                                  n_pix_rank * (size_t)(s_end > s_begin ? s_end - s_begin : 0), &want_probe))) return e;
    if ((e = ensure_tail(c, s))) return e;
    if ((e = ensure_tile_order(c, s, *prm, n_pix_rank * (size_t)(s_end > s_begin ? s_end - s_begin : 0)))) return e;
    const size_t film_floats = (size_t)W * H * 3;
    if ((e = ensure_buffer(&c->film, &c->film_floats, film_floats))) return e;
    // Paths in flight per pass.  Fewer, larger passes are faster (every launch of a pass ends in a drain phase, and late bounces
    // fill the chip better with more paths): configs[2] 295 ms at 32 Mi paths, 278 at 64 Mi, 267 with the whole frame (132.7 M
    // paths, 31 GB of state) in one pass; configs[3] 1 468 -> 1 340 ms.  HBM is there to be used: by default the pool may take
    // up to 72 % of the memory that is free (or already held by this pool), 364 B per path (two live buffers since round 3:
    // configs[3]'s 531 M paths still fit one pass).
    size_t capacity = (size_t)prm->max_paths_in_flight;
    if (!capacity) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        // a share of what is free, never a floor above it: a GPU this process shares with another tenant (a torch caching
        // allocator, another ctx or rank on the same device, a smaller-memory part) gets a smaller pool and more passes instead
        // of a failed allocation.  32 Mi paths only when the runtime cannot say what is free.  Hosts that share a GPU on
        // purpose set max_paths_in_flight.
        capacity = free_b ? (size_t)(0.72 * (double)(free_b + c->capacity * kBytesPerPath) / (double)kBytesPerPath) : ((size_t)32 << 20);
    }
    const size_t need = n_pix_rank * (s_end > s_begin ? s_end - s_begin : 0);
    if (need < capacity) capacity = need;
    if (capacity < prm->sample_batch) capacity = prm->sample_batch;
    if (capacity >= ((size_t)1 << 32)) capacity = ((size_t)1 << 32) - 1;
    while ((e = ensure_state(c, capacity))) {
        // the adaptive default asked for more than the allocator gives (fragmentation, another tenant): halve and retry
        if (prm->max_paths_in_flight || capacity <= (size_t)prm->sample_batch * 4096) return e;
        (void)hipGetLastError();
        capacity /= 2;
    }
    const std::vector<PassPlan> passes = plan(c->capacity, (uint32_t)n_pix_rank, s_begin, s_end, prm->sample_batch);
    if (want_probe && (e = probe_trace_records(c, s, *prm, (uint32_t)n_pix_rank, s_begin, s_end))) return e;
    const uint32_t used_records = (uint32_t)(s->use_b0 | (s->use_rest << 4));   // what THIS call's launches read (cray_stats.trace_records)

    for (int attempt = 0;; attempt++) {
        HIP_TRY(hipMemsetAsync(c->film, 0, film_floats * sizeof(float), c->stream));
        if ((e = reset_counters(c))) return e;
        EventTimer timer(c);
        EventTimer* tm = stats ? &timer : nullptr;
        for (const PassPlan& pp : passes)
            if ((e = run_pass(c, s, *prm, pp, tm))) return e;
        Counters h;
        HIP_TRY(hipMemcpyAsync(&h, c->counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        hipError_t err = hipStreamSynchronize(c->stream);
        if (err != hipSuccess) { set_last_error("render failed: %s", hipGetErrorString(err)); return CRAY_ERR_HIP; }
        double ms[FAM_COUNT]; uint32_t launches[FAM_COUNT];
        if (tm && (e = timer.collect(ms, launches))) return e;
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            fill_stats(h, stats);
            stats->paths = need;
            stats->trace_records = used_records;
            stats->trace_mixed_ms = ms[FAM_MIXED]; stats->trace_mixed_launches = launches[FAM_MIXED];
            stats->trace_closest_ms = ms[FAM_CLOSEST]; stats->trace_any_ms = ms[FAM_ANY];
            stats->shade_ms = ms[FAM_SHADE]; stats->other_ms = ms[FAM_OTHER];
            stats->trace_closest_launches = launches[FAM_CLOSEST]; stats->trace_any_launches = launches[FAM_ANY];
            stats->shade_launches = launches[FAM_SHADE];
        }
        if (!h.stack_overflow) return CRAY_OK;
        // a ray needed more pending nodes than the traversal stack holds: its result is not the reference's
        if (!s->needs_deep && attempt == 0) {
            // first time for this scene: add the third stack level (HBM, once per context) and render the frame again in the
            // instantiations that carry it
            if ((e = ensure_deep(c))) return e;
            s->needs_deep = true;
            continue;
        }
        set_last_error("BVH deeper than the %u-entry traversal stack: %llu lane(s) overflowed; the film is not valid",
                       (unsigned)kStackDepth + c->deep_depth, h.stack_overflow);
        return CRAY_ERR_UNSUPPORTED;
    }
}

// Device destination of a finished film: the caller's device pointer, or the context's staging buffer when the caller
// handed host memory (copied out by finish_output).
int output_target(cray_ctx* c, const cray_render_params* prm, float* out_rgb, size_t film_floats, float** dst) {
    if (prm->out_is_device) { *dst = out_rgb; return CRAY_OK; }
    int e = ensure_buffer(&c->out_stage, &c->out_stage_floats, film_floats);
    *dst = c->out_stage;
    return e;
}
int finish_output(cray_ctx* c, const cray_render_params* prm, float* out_rgb, size_t film_floats) {
    if (!prm->out_is_device && out_rgb)
        HIP_TRY(hipMemcpyAsync(out_rgb, c->out_stage, film_floats * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    hipError_t err = hipStreamSynchronize(c->stream);
    if (err != hipSuccess) { set_last_error("render failed: %s", hipGetErrorString(err)); return CRAY_ERR_HIP; }
    return CRAY_OK;
}

}  // namespace

extern "C" int cray_render(cray_ctx* c, cray_scene* s, const cray_render_params* prm, float* out_rgb, cray_stats* stats) {
    int e = check_render_args(c, s, prm);
    if (e) return e;
    if (!out_rgb) { set_last_error("cray_render: out_rgb is null"); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    const size_t film_floats = (size_t)s->dev.film_w * s->dev.film_h * 3;
    float* dst = nullptr;
    if ((e = output_target(c, prm, out_rgb, film_floats, &dst))) return e;
    auto t0 = std::chrono::steady_clock::now();
    if ((e = render_local(c, s, prm, stats))) return e;
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(c, film_floats, 8)), dim3(kBlock), 0, c->stream, c->film, dst, film_floats, (float)s->dev.num_samples);
    if ((e = finish_output(c, prm, out_rgb, film_floats))) return e;
    if (stats) stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return CRAY_OK;
}

// per-path radiance: out_L[(pixel_in_row_major * n + j)][3], n = sample_end - sample_begin (one batch at most)
extern "C" int cray_render_samples(cray_ctx* c, cray_scene* s, const cray_render_params* prm, double* out_L) {
    int e = check_render_args(c, s, prm);
    if (e) return e;
    if (!out_L) { set_last_error("cray_render_samples: out_L is null"); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    const DevScene& d = s->dev;
    uint32_t s_begin = prm->sample_begin, s_end = prm->sample_end;
    if (s_begin == 0 && s_end == 0) s_end = d.num_samples;
    if (s_end > d.num_samples || s_end <= s_begin || prm->world_size != 1) { set_last_error("cray_render_samples: bad sample range / needs world_size 1"); return CRAY_ERR_INVALID; }
    const uint32_t n = s_end - s_begin;
    const size_t n_pix = (size_t)d.film_w * d.film_h;
    if (n_pix * n >= ((size_t)1 << 32)) { set_last_error("cray_render_samples: too many paths"); return CRAY_ERR_UNSUPPORTED; }
    std::vector<uint32_t> pix(n_pix);
    for (size_t i = 0; i < n_pix; i++) pix[i] = (uint32_t)i;
    c->pix_count_valid = false;  // this call overwrites the cached per-rank pixel list
    if (pix.size() > c->pix_capacity) {
        if (c->pix_list) (void)hipFree(c->pix_list);
        c->pix_list = nullptr; c->pix_capacity = 0;
        HIP_TRY(hipMalloc((void**)&c->pix_list, pix.size() * 4));
        c->pix_capacity = pix.size();
    }
    const size_t film_floats = n_pix * 3;
    if (film_floats > c->film_floats) {
        if (c->film) (void)hipFree(c->film);
        c->film = nullptr; c->film_floats = 0;
        HIP_TRY(hipMalloc((void**)&c->film, film_floats * sizeof(float)));
        c->film_floats = film_floats;
    }
    if ((e = ensure_state(c, n_pix * n))) return e;
    // the records THIS scene's launches read (what an earlier cray_render left in use_* belongs to that call): pinned, chosen, or f64
    if ((e = choose_trace_records(c, s, prm->count_traversal != 0 || prm->precision != CRAY_PRECISION_F64, n_pix * n, nullptr))) return e;
    if ((e = ensure_tail(c, s))) return e;
    HIP_TRY(hipMemcpyAsync(c->pix_list, pix.data(), pix.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->film, 0, film_floats * sizeof(float), c->stream));
    if ((e = reset_counters(c))) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    cray_render_params p2 = *prm;
    p2.sample_batch = n > p2.sample_batch ? n : p2.sample_batch;
    PassPlan pp{0, (uint32_t)n_pix, s_begin, s_end};
    c->pix_order = c->pix_list;
    c->order_valid = false;
    if ((e = run_pass(c, s, p2, pp, nullptr))) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<double> r(n_pix * n), g(n_pix * n), b(n_pix * n);
    HIP_TRY(hipMemcpy(r.data(), c->ps.lr, r.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(g.data(), c->ps.lg, g.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b.data(), c->ps.lb, b.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < r.size(); i++) { out_L[3 * i] = r[i]; out_L[3 * i + 1] = g[i]; out_L[3 * i + 2] = b[i]; }
    return CRAY_OK;
}

extern "C" int cray_trace(cray_ctx* c, cray_scene* s, const cray_ray* rays, size_t n, cray_hit* hits, int mode, cray_stats* stats) {
    if (!c || !s || (!rays && n) || (!hits && n)) { set_last_error("cray_trace: null argument"); return CRAY_ERR_INVALID; }
    if (s->ctx != c) { set_last_error("scene was uploaded through a different context"); return CRAY_ERR_INVALID; }
    if (n >= ((size_t)1 << 30)) { set_last_error("cray_trace: too many rays in one call"); return CRAY_ERR_UNSUPPORTED; }
    if (mode < CRAY_TRACE_CLOSEST || mode > CRAY_TRACE_MIXED_TIMED) { set_last_error("cray_trace: unknown mode %d", mode); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return CRAY_OK; }
    const bool mixed = mode == CRAY_TRACE_MIXED_TIMED;
    const bool do_any = mode == CRAY_TRACE_ANY || mode == CRAY_TRACE_ANY_TIMED || mixed;
    const bool do_closest = mode == CRAY_TRACE_CLOSEST || mode == CRAY_TRACE_CLOSEST_TIMED || mixed;
    int e;
    if ((e = ensure_state(c, n))) return e;
    std::vector<double> col(n);
    PathState& ps = c->ps;
    for (int k = 0; k < 3; k++) {
        double* o_closest[3] = {ps.ox, ps.oy, ps.oz}; double* d_closest[3] = {ps.dx, ps.dy, ps.dz};
        double* o_any[3] = {ps.sox, ps.soy, ps.soz}; double* d_any[3] = {ps.sdx, ps.sdy, ps.sdz};
        for (size_t i = 0; i < n; i++) col[i] = rays[i].o[k];
        if (do_closest) HIP_TRY(hipMemcpy(o_closest[k], col.data(), n * 8, hipMemcpyHostToDevice));
        if (do_any) HIP_TRY(hipMemcpy(o_any[k], col.data(), n * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < n; i++) col[i] = rays[i].d[k];
        if (do_closest) HIP_TRY(hipMemcpy(d_closest[k], col.data(), n * 8, hipMemcpyHostToDevice));
        if (do_any) HIP_TRY(hipMemcpy(d_any[k], col.data(), n * 8, hipMemcpyHostToDevice));
    }
    for (size_t i = 0; i < n; i++) col[i] = rays[i].tmax;
    HIP_TRY(hipMemcpy(ps.stmax, col.data(), n * 8, hipMemcpyHostToDevice));
    if ((e = ensure_tail(c, s))) return e;   // (before the counters are reset: they carry the small-launch threshold and table)
    if ((e = reset_counters(c))) return e;
    // any-hit resolution adds `contribution` to L: use L as the "unoccluded" flag (0 + 1)
    if (do_any) {
        for (size_t i = 0; i < n; i++) col[i] = 0.0;
        HIP_TRY(hipMemcpy(ps.lr, col.data(), n * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ps.lg, col.data(), n * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ps.lb, col.data(), n * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ps.cg, col.data(), n * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ps.cb, col.data(), n * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < n; i++) col[i] = 1.0;
        HIP_TRY(hipMemcpy(ps.cr, col.data(), n * 8, hipMemcpyHostToDevice));
        std::vector<uint32_t> iota0(n);   // shadow slot i belongs to "path" i
        for (size_t i = 0; i < n; i++) iota0[i] = (uint32_t)i;
        HIP_TRY(hipMemcpy(ps.sp0, iota0.data(), n * 4, hipMemcpyHostToDevice));
    }
    const int g = grid_for(c, n, 8);
    Counters* ctr = c->counters;
    // the per-ray hook reads the records the context is pinned to (CRAY_HYBRID); a context that chooses per scene reads f64 here
    s->hybrid_ok = hybrid_possible(s);
    const int level = (c->hybrid > 0 && s->hybrid_ok) ? c->hybrid : 0;
    if ((e = ensure_hybrid(c, s, level))) return e;
    const bool shp = shapes_fit_lds(c, s->dev);   // the timed instantiations the frame loop would launch for this scene
    const bool shp_hyb = shp && s->dev.n_spheres + s->dev.n_disks <= kTraceLdsShapesHyb;
    hipEvent_t ev_[2] = {nullptr, nullptr};   // the launch's time for `stats` (trace_mixed_ms / trace_any_ms / trace_closest_ms)
    if (stats) { HIP_TRY(hipEventCreate(&ev_[0])); HIP_TRY(hipEventCreate(&ev_[1])); }
    struct EvFree { hipEvent_t* e; ~EvFree() { for (int i = 0; i < 2; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_free_{ev_};
    if (mixed) {
        // k_trace_mixed as the frame loop launches it: positions [0, n) of the virtual queue are the shadow rays of paths
        // 0..n-1 (an identity queue), positions [n, 2n) the path segments of the same paths (tmax = +inf, like Ray::new)
        std::vector<uint32_t> iota(n);
        for (size_t i = 0; i < n; i++) iota[i] = (uint32_t)i;
        const unsigned int cnt = (unsigned int)n;
        HIP_TRY(hipMemcpy(c->shadow_queue, iota.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(&ctr->n_shadow, &cnt, sizeof(cnt), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(&ctr->n_active0, &cnt, sizeof(cnt), hipMemcpyHostToDevice));
        if (stats) HIP_TRY(hipEventRecord(ev_[0], c->stream));
        launch_mixed(level, shp, shp_hyb, c->tail_res != nullptr, s->needs_deep && c->deep_depth != 0, g, g, c->stream, s->dev, ps, (const uint32_t*)c->shadow_queue, (const unsigned int*)&ctr->n_shadow, (const uint32_t*)nullptr,
                     (const unsigned int*)&ctr->n_active0, ctr, &ctr->trace_head, c->refill_min | (c->refill_min_any << 16) | (c->steal ? 0x8000u : 0u) | (shp ? 0x4000u : 0u) | (c->tail_seg ? 0x2000u : 0u));
    } else {
#define CRAY_TRACE_GO(ANY_, COUNT_, TMAX_)                                                                                       \
    launch_trace<ANY_, COUNT_>(level, shp, shp_hyb, s->needs_deep && c->deep_depth != 0, g, g, c->stream, s->dev, ps, (const uint32_t*)nullptr, (const unsigned int*)nullptr, (uint32_t)n, TMAX_, ctr, \
                               &ctr->trace_head, c->refill_min | (c->steal ? 0x8000u : 0u) | (shp ? 0x4000u : 0u))
        // closest hit with caller-supplied tmax: rays whose tmax is finite go through the same kernel via stmax
        if (stats) HIP_TRY(hipEventRecord(ev_[0], c->stream));
        if (mode == CRAY_TRACE_ANY) CRAY_TRACE_GO(true, true, (const double*)nullptr);
        else if (mode == CRAY_TRACE_ANY_TIMED) CRAY_TRACE_GO(true, false, (const double*)nullptr);
        else if (mode == CRAY_TRACE_CLOSEST) CRAY_TRACE_GO(false, true, (const double*)ps.stmax);
        else CRAY_TRACE_GO(false, false, (const double*)ps.stmax);
#undef CRAY_TRACE_GO
    }
    if (stats) HIP_TRY(hipEventRecord(ev_[1], c->stream));
    cray_hit* closest_out = mixed ? hits + n : hits;
    if (do_any) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(col.data(), ps.lr, n * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) {
            memset(&hits[i], 0, sizeof(cray_hit));
            hits[i].hit = col[i] == 0.0 ? 1 : 0;
            hits[i].prim = -1;
        }
    }
    if (do_closest) {
        cray_hit* d_hits = nullptr;
        HIP_TRY(hipMalloc((void**)&d_hits, n * sizeof(cray_hit)));
        hipLaunchKernelGGL(k_hit_records, dim3(g), dim3(kBlock), 0, c->stream, s->dev, ps, (uint32_t)n, d_hits);
        hipError_t err = hipStreamSynchronize(c->stream);
        if (err == hipSuccess) err = hipMemcpy(closest_out, d_hits, n * sizeof(cray_hit), hipMemcpyDeviceToHost);
        (void)hipFree(d_hits);
        if (err != hipSuccess) { set_last_error("cray_trace failed: %s", hipGetErrorString(err)); return CRAY_ERR_HIP; }
    }
    Counters h;
    HIP_TRY(hipMemcpy(&h, c->counters, sizeof(h), hipMemcpyDeviceToHost));
    if (h.stack_overflow) {
        if (!s->needs_deep) {
            if ((e = ensure_deep(c))) return e;
            s->needs_deep = true;
            return cray_trace(c, s, rays, n, hits, mode, stats);
        }
        set_last_error("BVH deeper than the %u-entry traversal stack", (unsigned)kStackDepth + c->deep_depth);
        return CRAY_ERR_UNSUPPORTED;
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        fill_stats(h, stats);
        float ms_ = 0.f;
        HIP_TRY(hipEventSynchronize(ev_[1]));
        HIP_TRY(hipEventElapsedTime(&ms_, ev_[0], ev_[1]));
        if (mixed) { stats->trace_mixed_ms = ms_; stats->trace_mixed_launches = 1; }
        else if (do_any) { stats->trace_any_ms = ms_; stats->trace_any_launches = 1; }
        else { stats->trace_closest_ms = ms_; stats->trace_closest_launches = 1; }
    }
    return CRAY_OK;
}

// ---------------------------------------------------------------------------------------------
// Bvh::new(primitives, SplitMethod::SAH) on the device (cray_bvh_build.h): same nodes, same leaf order.
// ---------------------------------------------------------------------------------------------
namespace {
// Bvh::new on device memory: d_box [n][6] -> DFS pre-order nodes + leaf order, both left on the device in `mem`.
int bvh_build_device(cray_ctx* c, const double* d_box, uint32_t n, DevMem& mem, BvhOnDevice* r) {
    using namespace cray::bvhb;
    hipStream_t st = c->stream;
    uint32_t *d_order, *d_T, *d_tmp, *d_active[2], *d_small, *d_tile, *d_total, *d_reccnt;
    int32_t* d_slot_of; uint8_t *d_bidx, *d_flag;
    TopNode* d_top; SmallRec* d_pool; Slot* d_slots[2]; Ctl* d_ctl; cray_bvh_node* d_out;
    const size_t max_active = (size_t)n / (kSmall + 1) + 2;
    const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
    DevMem tmp;  // scratch of the build, released when it returns; the results live in `mem`
    HIP_TRY(mem.get(&d_order, n));
    HIP_TRY(mem.get(&d_out, (size_t)2 * n));
    HIP_TRY(tmp.get(&d_slot_of, n));
    HIP_TRY(tmp.get(&d_bidx, n));
    HIP_TRY(tmp.get(&d_flag, n));
    HIP_TRY(tmp.get(&d_T, (size_t)n + 1));
    HIP_TRY(tmp.get(&d_tmp, n));
    HIP_TRY(tmp.get(&d_tile, n_tiles));
    HIP_TRY(tmp.get(&d_total, 1));
    HIP_TRY(tmp.get(&d_active[0], max_active));
    HIP_TRY(tmp.get(&d_active[1], max_active));
    HIP_TRY(tmp.get(&d_slots[0], max_active));
    HIP_TRY(tmp.get(&d_slots[1], max_active));
    HIP_TRY(tmp.get(&d_small, n));
    HIP_TRY(tmp.get(&d_reccnt, n));
    HIP_TRY(tmp.get(&d_top, (size_t)2 * n));
    HIP_TRY(tmp.get(&d_pool, (size_t)2 * n));
    HIP_TRY(tmp.get(&d_ctl, 1));
    hipEvent_t ev0, ev1;
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, st));

    const dim3 blk(kTB), grid_n((n + kTB - 1) / kTB);
    hipLaunchKernelGGL(k_init, grid_n, blk, 0, st, d_order, d_slot_of, n, n > kSmall ? 0 : -1);
    hipLaunchKernelGGL(k_root, dim3(1), dim3(1), 0, st, d_top, d_slots[0], d_active[0], d_small, d_ctl, n);
    uint32_t n_active = n > kSmall ? 1u : 0u, levels = 0;
    Ctl h{};
    int cur = 0;
    int rc = CRAY_OK;
    auto sync_ctl = [&](Ctl* dst) -> int {
        HIP_TRY(hipMemcpyAsync(dst, d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return CRAY_OK;
    };
    while (n_active > 0 && rc == CRAY_OK) {
        if (++levels > 4096) { set_last_error("cray_bvh_build_sah: tree deeper than 4096 levels above the %u-primitive subtrees", kSmall); rc = CRAY_ERR_UNSUPPORTED; break; }
        const dim3 grid_a((n_active + kTB - 1) / kTB);
        // only the root reduces its bounds from the primitives; below it they come from the parent's SAH buckets (k_children)
        if (levels == 1) hipLaunchKernelGGL(k_bounds, grid_n, blk, 0, st, d_box, d_order, d_slot_of, d_slots[cur], n);
        hipLaunchKernelGGL(k_setup, grid_a, blk, 0, st, d_top, d_slots[cur], n_active, d_ctl);
        hipLaunchKernelGGL(k_buckets, grid_n, blk, 0, st, d_box, d_order, d_slot_of, d_slots[cur], d_bidx, n);
        hipLaunchKernelGGL(k_choose, grid_a, blk, 0, st, d_slots[cur], n_active, d_ctl);
        hipLaunchKernelGGL(k_flags, grid_n, blk, 0, st, d_slot_of, d_slots[cur], d_bidx, d_flag, n);
        hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), blk, 0, st, d_flag, d_T, d_tile, n);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), blk, 0, st, d_tile, n_tiles, d_total);
        hipLaunchKernelGGL(k_scan_add, grid_n, blk, 0, st, d_T, d_tile, d_total, n);
        hipLaunchKernelGGL(k_pairs, grid_n, blk, 0, st, d_slot_of, d_slots[cur], d_top, d_flag, d_T, d_tmp, n);
        hipLaunchKernelGGL(k_swap, grid_n, blk, 0, st, d_slot_of, d_slots[cur], d_top, d_flag, d_T, d_tmp, d_order, d_bidx, n);
        hipLaunchKernelGGL(k_children, grid_a, blk, 0, st, d_top, d_slots[cur], d_slots[cur ^ 1], n_active, d_T, d_active[cur ^ 1], d_small, d_ctl);
        hipLaunchKernelGGL(k_reslot, grid_n, blk, 0, st, d_slot_of, d_slots[cur], d_top, n);
        if ((rc = sync_ctl(&h))) break;
        if (h.error) break;
        n_active = h.n_next;
        if (n_active > max_active) { set_last_error("cray_bvh_build_sah: internal error (active list overflow)"); rc = CRAY_ERR_INVALID; break; }
        const unsigned int zero = 0;
        if (hipMemcpyAsync(&d_ctl->n_next, &zero, sizeof(zero), hipMemcpyHostToDevice, st) != hipSuccess) { set_last_error("hipMemcpyAsync failed"); rc = CRAY_ERR_HIP; break; }
        cur ^= 1;
    }
    uint32_t n_leaves = 0;
    if (rc == CRAY_OK && !h.error && (rc = sync_ctl(&h)) == CRAY_OK) {
        hipError_t e = hipMemsetAsync(d_flag, 0, n, st);
        hipLaunchKernelGGL(k_small, dim3((h.n_small + 63) / 64), dim3(64), 0, st, d_box, d_order, d_top, d_small, h.n_small, d_pool, d_reccnt, d_flag, d_ctl);
        hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), blk, 0, st, d_flag, d_T, d_tile, n);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), blk, 0, st, d_tile, n_tiles, d_total);
        hipLaunchKernelGGL(k_scan_add, grid_n, blk, 0, st, d_T, d_tile, d_total, n);
        hipLaunchKernelGGL(k_emit_top, dim3((h.n_top + kTB - 1) / kTB), blk, 0, st, d_top, h.n_top, d_T, d_out);
        hipLaunchKernelGGL(k_emit_small, dim3((h.n_small + kTB - 1) / kTB), blk, 0, st, d_top, d_small, h.n_small, d_pool, d_reccnt, d_T, d_out);
        if (e == hipSuccess) e = hipEventRecord(ev1, st);
        Ctl h2{};
        if (e == hipSuccess) e = hipMemcpyAsync(&n_leaves, d_T + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&h2, d_ctl, sizeof(h2), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_last_error("Bvh::new on the device failed: %s", hipGetErrorString(e)); rc = CRAY_ERR_HIP; }
        h.error = h2.error;
        if (rc == CRAY_OK && !h.error) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev0, ev1);
            r->d_nodes = d_out; r->d_order = d_order; r->n_nodes = 2 * n_leaves - 1;
            r->stats.device_seconds = ms * 1e-3;
            r->stats.levels = levels; r->stats.top_nodes = h.n_top; r->stats.small_subtrees = h.n_small; r->stats.leaves = n_leaves;
        }
    }
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    if (rc != CRAY_OK) return rc;
    if (h.error) {
        set_last_error("Bvh::new would panic in the reference (code %u: 1 zero surface area, 2 non-finite cost, 3 empty partition)", h.error);
        return CRAY_ERR_BUILD;
    }
    return CRAY_OK;
}
}  // namespace

extern "C" int cray_bvh_build_sah(cray_ctx* c, const double* prim_bounds, uint32_t n, cray_bvh_node* out_nodes, uint32_t node_capacity,
                                  uint32_t* out_n_nodes, uint32_t* out_prim_refs, cray_bvh_build_stats* stats) {
    if (!c || !prim_bounds || !out_nodes || !out_n_nodes || !out_prim_refs) { set_last_error("cray_bvh_build_sah: null argument"); return CRAY_ERR_INVALID; }
    if (n == 0) { set_last_error("Bvh::new: no primitives"); return CRAY_ERR_BUILD; }
    if (n >= (1u << 30)) { set_last_error("cray_bvh_build_sah: too many primitives"); return CRAY_ERR_UNSUPPORTED; }
    for (size_t i = 0; i < (size_t)n * 6; i++)
        if (!std::isfinite(prim_bounds[i])) { set_last_error("cray_bvh_build_sah: primitive %zu has a non-finite bound", i / 6); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    auto t_begin = std::chrono::steady_clock::now();
    DevMem mem;
    double* d_box;
    HIP_TRY(mem.get(&d_box, (size_t)n * 6));
    HIP_TRY(hipMemcpyAsync(d_box, prim_bounds, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    BvhOnDevice r{};
    int rc = bvh_build_device(c, d_box, n, mem, &r);
    if (rc != CRAY_OK) return rc;
    if (r.n_nodes > node_capacity) { set_last_error("cray_bvh_build_sah: %u nodes do not fit the caller's %u", r.n_nodes, node_capacity); return CRAY_ERR_INVALID; }
    HIP_TRY(hipMemcpy(out_nodes, r.d_nodes, (size_t)r.n_nodes * sizeof(cray_bvh_node), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_prim_refs, r.d_order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *out_n_nodes = r.n_nodes;
    if (stats) {
        *stats = r.stats;
        stats->total_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return CRAY_OK;
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU: tile shard + gather of Film tiles over RCCL (include/cray.h "multi-GPU").
// The reference's workers are threads adding tiles into one Mutex<Vec<f32>> (craytracer.rs:245, 271-291, 182-188);
// here the workers are GPUs, each with the pixels of its tiles complete, and the merge is one gather to rank 0.
// ---------------------------------------------------------------------------------------------
namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;   // optional (reporting only)
};
static_assert(sizeof(ncclUniqueId) == CRAY_COMM_ID_BYTES, "cray_comm_id is an ncclUniqueId");

// librccl.so.1: the copy already in the process (a PyTorch host has loaded its own) or the ROCm one on the library path
Rccl* rccl() {
    static Rccl api;
    static bool tried = false;
    if (api.handle) return &api;
    if (tried) return nullptr;
    tried = true;
    // CRAY_RCCL_LIB=<path>: a particular build of the collective library (the tests use it to put several ranks on one GPU
    // through a shared-memory stand-in, tests/mock_rccl/)
    const char* forced = getenv("CRAY_RCCL_LIB");
    for (const char* name : {forced, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (!name || !name[0]) continue;
        api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.handle || name == forced) break;   // a forced library that does not load is an error, not a fallback
    }
    if (!api.handle) { set_last_error("cannot load librccl.so.1: %s", dlerror()); return nullptr; }
    bool ok = true;
    auto sym = [&](const char* n) { void* f = dlsym(api.handle, n); if (!f) { set_last_error("librccl: missing symbol %s", n); ok = false; } return f; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.GetVersion = (decltype(api.GetVersion))dlsym(api.handle, "ncclGetVersion");
    if (!ok) { dlclose(api.handle); api.handle = nullptr; return nullptr; }
    return &api;
}

#define RCCL_TRY(R, expr)                                                                          \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            set_last_error("%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(r_), __FILE__, __LINE__); \
            return CRAY_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

// packed[3 i + k] = film[3 pix[i] + k] / div   (div = num_samples resolves the film on the way; x / 1.0f == x)
__global__ void __launch_bounds__(kBlock) k_pack_tiles(const float* __restrict__ film, const uint32_t* __restrict__ pix, size_t n_pix, float div,
                                                       float* __restrict__ packed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = n_pix * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t q = i / 3;
        packed[i] = film[(size_t)pix[q] * 3 + (i - q * 3)] / div;
    }
}
// out[3 all_pix[q] + k] = gathered[3 q + k]: the rank-ordered concatenation back into the row-major film
__global__ void __launch_bounds__(kBlock) k_unpack_tiles(const float* __restrict__ gathered, const uint32_t* __restrict__ all_pix, size_t n_pix,
                                                         float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = n_pix * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t q = i / 3;
        out[(size_t)all_pix[q] * 3 + (i - q * 3)] = gathered[i];
    }
}

// rank 0's map of the gathered buffer: the pixel lists of ranks 0..world-1 one after the other
int ensure_all_pix(cray_ctx* c, uint32_t W, uint32_t H, uint32_t tw, uint32_t th, uint32_t world) {
    const uint64_t key[5] = {W, H, tw, th, world};
    if (c->all_valid && memcmp(key, c->all_key, sizeof(key)) == 0) return CRAY_OK;
    c->all_valid = false;
    std::vector<uint32_t> all;
    try {
        all.reserve((size_t)W * H);
        c->rank_offset.assign(world + 1, 0);
        cray_render_params p;
        cray_render_params_default(&p);
        p.tile_width = tw; p.tile_height = th; p.world_size = world;
        for (uint32_t r = 0; r < world; r++) {
            p.rank = r;
            const std::vector<uint32_t> mine = rank_pixels(W, H, p);
            all.insert(all.end(), mine.begin(), mine.end());
            c->rank_offset[r + 1] = all.size();
        }
    } catch (const std::exception& ex) { set_last_error("tile map: %s", ex.what()); return CRAY_ERR_INVALID; }
    if (all.size() != (size_t)W * H) { set_last_error("internal error: tile lists cover %zu of %zu pixels", all.size(), (size_t)W * H); return CRAY_ERR_INVALID; }
    int e = ensure_buffer(&c->all_pix, &c->all_pix_capacity, all.size());
    if (e) return e;
    HIP_TRY(hipMemcpyAsync(c->all_pix, all.data(), all.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(c->all_key, key, sizeof(key));
    c->all_valid = true;
    return CRAY_OK;
}

// Everything of the gather that can fail on ONE rank alone (the tile map, the buffers): done before the ranks agree on
// their status, so that no rank enters the point-to-point exchange while another has already returned an error.
int gather_prepare(cray_ctx* c, uint32_t W, uint32_t H, uint32_t tw, uint32_t th) {
    if (!rccl() || !c->comm) { set_last_error("no communicator: call cray_comm_init first"); return CRAY_ERR_INVALID; }
    const uint32_t world = (uint32_t)c->comm_world, rank = (uint32_t)c->comm_rank;
    const size_t n_mine = c->pix_count;
    int e;
    if (rank == 0) {
        if ((e = ensure_all_pix(c, W, H, tw, th, world))) return e;
        if (c->rank_offset[1] != n_mine) { set_last_error("internal error: rank 0 owns %zu pixels, the map says %zu", n_mine, c->rank_offset[1]); return CRAY_ERR_INVALID; }
        if ((e = ensure_buffer(&c->gathered, &c->gathered_floats, (size_t)W * H * 3))) return e;
    } else {
        if ((e = ensure_buffer(&c->packed, &c->packed_floats, n_mine * 3))) return e;
    }
    return CRAY_OK;
}

// Pack this rank's tiles out of `film` (dividing by `div`), move them to rank 0, and there rebuild the row-major film
// in `dst` (device).  Collective over the ctx's communicator; c->pix_list must hold this rank's pixel list and
// gather_prepare + comm_agree must have succeeded on every rank.
int gather_tiles(cray_ctx* c, uint32_t W, uint32_t H, uint32_t tw, uint32_t th, const float* film, float div, float* dst) {
    Rccl* R = rccl();
    const uint32_t world = (uint32_t)c->comm_world, rank = (uint32_t)c->comm_rank;
    const size_t n_mine = c->pix_count;
    // An error INSIDE the exchange must not leave this rank's group open or its remaining receives unposted: a receive that was
    // never posted strands its sender, and an unclosed group poisons the next collective.  So every receive is posted whatever
    // an earlier one returned, the group is always closed, and the first error is what the call returns.
    ncclResult_t first = ncclSuccess;
    const char* what = "";
#define CRAY_KEEP_FIRST(expr) do { const ncclResult_t r_ = (expr); if (r_ != ncclSuccess && first == ncclSuccess) { first = r_; what = #expr; } } while (0)
    if (rank == 0) {
        const size_t total = (size_t)W * H;
        // rank 0's own tiles go straight to the head of the gathered buffer
        if (n_mine) hipLaunchKernelGGL(k_pack_tiles, dim3(grid_for(c, n_mine * 3, 8)), dim3(kBlock), 0, c->stream, film, c->pix_list, n_mine, div, c->gathered);
        CRAY_KEEP_FIRST(R->GroupStart());
        for (uint32_t r = 1; r < world; r++) {
            const size_t cnt = c->rank_offset[r + 1] - c->rank_offset[r];
            if (cnt) CRAY_KEEP_FIRST(R->Recv(c->gathered + c->rank_offset[r] * 3, cnt * 3, ncclFloat, (int)r, c->comm, c->stream));
        }
        CRAY_KEEP_FIRST(R->GroupEnd());
        if (first == ncclSuccess)
            hipLaunchKernelGGL(k_unpack_tiles, dim3(grid_for(c, total * 3, 8)), dim3(kBlock), 0, c->stream, (const float*)c->gathered, (const uint32_t*)c->all_pix, total, dst);
    } else if (n_mine) {
        hipLaunchKernelGGL(k_pack_tiles, dim3(grid_for(c, n_mine * 3, 8)), dim3(kBlock), 0, c->stream, film, c->pix_list, n_mine, div, c->packed);
        CRAY_KEEP_FIRST(R->Send(c->packed, n_mine * 3, ncclFloat, 0, c->comm, c->stream));
    }
#undef CRAY_KEEP_FIRST
    if (first != ncclSuccess) {
        set_last_error("gather of Film tiles: %s failed on rank %u: %s", what, rank, R->GetErrorString(first));
        (void)hipGetLastError();
        return CRAY_ERR_HIP;
    }
    HIP_TRY(hipGetLastError());
    return CRAY_OK;
}

// DevScene's device arrays in the order cray_scene_upload allocates them (= cray_scene::allocs)
constexpr int kSceneArrays = 16;
void scene_arrays(DevScene& d, const void** out[kSceneArrays]) {
    int i = 0;
    out[i++] = (const void**)&d.inner; out[i++] = (const void**)&d.slots; out[i++] = (const void**)&d.prims; out[i++] = (const void**)&d.tri_shade;
    out[i++] = (const void**)&d.spheres; out[i++] = (const void**)&d.disks; out[i++] = (const void**)&d.materials; out[i++] = (const void**)&d.bxdfs;
    out[i++] = (const void**)&d.textures; out[i++] = (const void**)&d.images; out[i++] = (const void**)&d.pool; out[i++] = (const void**)&d.gamma_lut;
    out[i++] = (const void**)&d.lights; out[i++] = (const void**)&d.light_cdf; out[i++] = (const void**)&d.first_equal_light; out[i++] = (const void**)&d.sobol;
}

struct SceneHeader {
    DevScene dev;  // root's copy; the pointers are replaced on the receiving side
    uint64_t bytes[kSceneArrays];
    uint32_t n_prims, magic, features;
    int32_t shade_variant;
    uint32_t n_slots, pad_;
};

}  // namespace

static void comm_release(cray_ctx* c) {
    if (c->comm) { if (Rccl* R = rccl()) (void)R->CommDestroy(c->comm); c->comm = nullptr; }
    if (c->comm_scratch) (void)hipFree(c->comm_scratch);
    if (c->packed) (void)hipFree(c->packed);
    if (c->gathered) (void)hipFree(c->gathered);
    if (c->all_pix) (void)hipFree(c->all_pix);
    c->comm_scratch = nullptr; c->packed = nullptr; c->gathered = nullptr; c->all_pix = nullptr;
}

// A plain 16-B-per-lane streaming read of `bytes` of HBM: what the chip delivers to a kernel that only reads
// (MI355X_MICROARCH.md: ~6.3 TB/s of the 8 TB/s spec), measured with HIP events on the ctx's stream.
__global__ void __launch_bounds__(kBlock) k_stream_read(const double2* __restrict__ src, size_t n, double* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2 v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 123.456) *sink = acc;  // keeps the loads alive; never true for the zero-filled buffer
}

extern "C" void cray_ctx_pool_info(const cray_ctx* c, uint64_t* pool_bytes, uint64_t* paths) {
    if (pool_bytes) *pool_bytes = c ? (uint64_t)c->capacity * kBytesPerPath : 0;
    if (paths) *paths = c ? (uint64_t)c->capacity : 0;
}
extern "C" void cray_scene_records_info(const cray_scene* s, int32_t chosen[2], double* probe_ms, double probe_kernel_ms[4]) {
    if (chosen) { chosen[0] = s ? s->chosen_b0 : -1; chosen[1] = s ? s->chosen_rest : -1; }
    if (probe_ms) *probe_ms = s ? s->probe_ms : 0.0;
    if (probe_kernel_ms) for (int v = 0; v < 2; v++) for (int k = 0; k < 2; k++) probe_kernel_ms[2 * v + k] = s ? s->tune_ms[v][k] : 0.0;
}

extern "C" int cray_measure_stream_read(cray_ctx* c, uint64_t bytes, int repeats, double* gb_per_s) {
    if (!c || !gb_per_s || bytes < (1u << 20) || repeats < 1) { set_last_error("cray_measure_stream_read: bad argument"); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    DevMem mem;
    double2* buf;
    double* sink;
    const size_t n = bytes / sizeof(double2);
    HIP_TRY(mem.get(&buf, n));
    HIP_TRY(mem.get(&sink, 1));
    HIP_TRY(hipMemsetAsync(buf, 0, n * sizeof(double2), c->stream));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const dim3 grid(c->n_cu * 8);
    hipLaunchKernelGGL(k_stream_read, grid, dim3(kBlock), 0, c->stream, (const double2*)buf, n, sink);  // warm-up
    HIP_TRY(hipEventRecord(e0, c->stream));
    for (int i = 0; i < repeats; i++) hipLaunchKernelGGL(k_stream_read, grid, dim3(kBlock), 0, c->stream, (const double2*)buf, n, sink);
    HIP_TRY(hipEventRecord(e1, c->stream));
    hipError_t err = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (err != hipSuccess) { set_last_error("cray_measure_stream_read failed: %s", hipGetErrorString(err)); return CRAY_ERR_HIP; }
    *gb_per_s = (double)(n * sizeof(double2)) * repeats / (ms * 1e-3) / 1e9;
    return CRAY_OK;
}

extern "C" int cray_comm_unique_id(cray_comm_id* out) {
    if (!out) { set_last_error("cray_comm_unique_id: out is null"); return CRAY_ERR_INVALID; }
    Rccl* R = rccl();
    if (!R) return CRAY_ERR_UNSUPPORTED;
    ncclUniqueId id;
    RCCL_TRY(R, R->GetUniqueId(&id));
    memcpy(out->bytes, &id, sizeof(id));
    return CRAY_OK;
}

extern "C" int cray_comm_init(cray_ctx* c, const cray_comm_id* id, int rank, int world) {
    if (!c || !id) { set_last_error("cray_comm_init: null argument"); return CRAY_ERR_INVALID; }
    if (world < 1 || rank < 0 || rank >= world) { set_last_error("cray_comm_init: rank %d of %d", rank, world); return CRAY_ERR_INVALID; }
    if (c->comm) { set_last_error("cray_comm_init: the context already has a communicator"); return CRAY_ERR_INVALID; }
    Rccl* R = rccl();
    if (!R) return CRAY_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->comm_scratch) HIP_TRY(hipMalloc((void**)&c->comm_scratch, 64 * sizeof(double) > sizeof(SceneHeader) ? 64 * sizeof(double) : sizeof(SceneHeader)));
    ncclUniqueId nid;
    memcpy(&nid, id->bytes, sizeof(nid));
    RCCL_TRY(R, R->CommInitRank(&c->comm, world, nid, rank));
    c->comm_rank = rank; c->comm_world = world;
    return CRAY_OK;
}

extern "C" int cray_comm_rank(const cray_ctx* c) { return c && c->comm ? c->comm_rank : 0; }
extern "C" int cray_comm_world_size(const cray_ctx* c) { return c && c->comm ? c->comm_world : 1; }

extern "C" int cray_comm_allreduce_f64(cray_ctx* c, double* v, int n, int op) {
    if (!c || !v || n < 1 || n > 64 || op < 0 || op > 2) { set_last_error("cray_comm_allreduce_f64: bad argument"); return CRAY_ERR_INVALID; }
    if (!c->comm) return CRAY_OK;  // one rank: the values are the result
    Rccl* R = rccl();
    if (!R) return CRAY_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->comm_scratch, v, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    RCCL_TRY(R, R->AllReduce(c->comm_scratch, c->comm_scratch, (size_t)n, ncclDouble, ops[op], c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(v, c->comm_scratch, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CRAY_OK;
}

extern "C" int cray_comm_barrier(cray_ctx* c) {
    if (!c) { set_last_error("cray_comm_barrier: ctx is null"); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    if (!c->comm) { HIP_TRY(hipStreamSynchronize(c->stream)); return CRAY_OK; }
    double one = 1.0;
    return cray_comm_allreduce_f64(c, &one, 1, CRAY_REDUCE_SUM);
}

extern "C" int cray_comm_describe(cray_ctx* c, cray_comm_info* out) {
    if (!c || !out) { set_last_error("cray_comm_describe: null argument"); return CRAY_ERR_INVALID; }
    memset(out, 0, sizeof(*out));
    out->world_size = cray_comm_world_size(c); out->rank = cray_comm_rank(c); out->ranks_seen = 1;
    if (!c->comm) return CRAY_OK;
    Rccl* R = rccl();
    if (!R) return CRAY_ERR_UNSUPPORTED;
    int v = 0;
    if (R->GetVersion && R->GetVersion(&v) == ncclSuccess) out->rccl_version = v;
    Dl_info info;
    if (dladdr((const void*)R->Send, &info) && info.dli_fname) snprintf(out->library, sizeof(out->library), "%s", info.dli_fname);
    double one = 1.0;
    const int e = cray_comm_allreduce_f64(c, &one, 1, CRAY_REDUCE_SUM);
    if (e) return e;
    out->ranks_seen = (int32_t)(one + 0.5);
    return CRAY_OK;
}

// Every rank passes the return code of the work it did alone (0 or a negative CRAY_ERR_*); every rank gets the worst of
// them.  Real RCCL has no timeout: a rank that returned early from a collective sequence leaves its peers waiting in
// ncclRecv / ncclBroadcast for ever, so the ranks agree BEFORE any transfer whose other end might be missing.
// `shape` (optional, n_shape <= 8 values): numbers every rank must pass identically for the exchange to line up — film size, tile
// shape, sample count.  Ranks that disagree would post receives and sends of different lengths, so a mismatch is an error on all.
static int comm_agree(cray_ctx* c, int local, const double* shape = nullptr, int n_shape = 0) {
    if (!c->comm || c->comm_world == 1) return local;
    char mine[sizeof(g_err)];
    memcpy(mine, g_err, sizeof(mine));   // the all-reduce below may overwrite the thread's message
    double v[1 + 2 * 8];
    v[0] = (double)local;
    if (n_shape > 8) n_shape = 8;
    for (int i = 0; i < n_shape; i++) { v[1 + 2 * i] = shape[i]; v[2 + 2 * i] = -shape[i]; }   // MIN of x and of -x: min and max in one reduction
    const int e = cray_comm_allreduce_f64(c, v, 1 + 2 * n_shape, CRAY_REDUCE_MIN);
    if (e) return e;                     // the collective itself failed: nothing to agree with
    if (local) { memcpy(g_err, mine, sizeof(mine)); return local; }
    if (v[0] != 0.0) { set_last_error("another rank failed with code %d before the exchange (its own message has the detail)", (int)v[0]); return (int)v[0]; }
    for (int i = 0; i < n_shape; i++)
        if (v[1 + 2 * i] != -v[2 + 2 * i]) {
            set_last_error("the ranks disagree on argument %d of the exchange (film size / tile shape / samples): %g here, between %g and %g over the ranks",
                           i, shape[i], v[1 + 2 * i], -v[2 + 2 * i]);
            return CRAY_ERR_INVALID;
        }
    return CRAY_OK;
}

extern "C" int cray_scene_broadcast(cray_ctx* c, cray_scene* mine, int root, cray_scene** out) {
    if (!c || !out) { set_last_error("cray_scene_broadcast: null argument"); return CRAY_ERR_INVALID; }
    *out = nullptr;
    const int rank = cray_comm_rank(c), world = cray_comm_world_size(c);
    int bad = CRAY_OK;
    if (root < 0 || root >= world) { set_last_error("cray_scene_broadcast: root %d of %d", root, world); bad = CRAY_ERR_INVALID; }
    else if ((rank == root) != (mine != nullptr)) { set_last_error("cray_scene_broadcast: pass the scene on the root and NULL elsewhere"); bad = CRAY_ERR_INVALID; }
    else if (mine && (mine->ctx != c || mine->allocs.size() != (size_t)kSceneArrays)) { set_last_error("cray_scene_broadcast: scene does not belong to this context"); bad = CRAY_ERR_INVALID; }
    if (world == 1) { if (!bad) *out = mine; return bad; }
    // From here on every exit goes through an agreement of the ranks: a rank that returned on its own would leave the others
    // inside ncclBroadcast for ever.  (world > 1 means cray_comm_init succeeded, so the collective library is loaded.)
    Rccl* R = rccl();
    if (!R) return CRAY_ERR_UNSUPPORTED;
    if (!bad && hipSetDevice(c->device) != hipSuccess) { set_last_error("cray_scene_broadcast: hipSetDevice(%d) failed", c->device); (void)hipGetLastError(); bad = CRAY_ERR_HIP; }
    {
        // `root` is part of what must line up: ranks inside ncclBroadcast with different roots wait for each other for ever
        const double shape[1] = {(double)root};
        if ((bad = comm_agree(c, bad, shape, 1))) return bad;
    }
    SceneHeader h;
    memset(&h, 0, sizeof(h));
    if (mine) {
        h.dev = mine->dev; h.n_prims = mine->n_prims; h.magic = 0x43524159u;
        h.features = mine->features; h.shade_variant = mine->shade_variant; h.n_slots = mine->n_slots;
        for (int i = 0; i < kSceneArrays; i++) h.bytes[i] = mine->alloc_bytes[i];
        if (hipMemcpyAsync(c->comm_scratch, &h, sizeof(h), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            // the root still takes part in the header broadcast (a zeroed header: no magic), so that the receivers come back
            set_last_error("cray_scene_broadcast: the header copy failed on the root");
            (void)hipGetLastError();
            (void)hipMemsetAsync(c->comm_scratch, 0, sizeof(h), c->stream);
            bad = CRAY_ERR_HIP;
        }
    }
    {
        const ncclResult_t r = R->Broadcast(c->comm_scratch, c->comm_scratch, sizeof(h), ncclChar, root, c->comm, c->stream);
        if (r != ncclSuccess && !bad) { set_last_error("cray_scene_broadcast: ncclBroadcast of the header failed: %s", R->GetErrorString(r)); bad = CRAY_ERR_HIP; }
    }
    if (!bad) {
        hipError_t err = hipMemcpyAsync(&h, c->comm_scratch, sizeof(h), hipMemcpyDeviceToHost, c->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
        if (err != hipSuccess) { set_last_error("cray_scene_broadcast: reading the header failed: %s", hipGetErrorString(err)); (void)hipGetLastError(); bad = CRAY_ERR_HIP; }
    }
    if (!bad && h.magic != 0x43524159u) { set_last_error("cray_scene_broadcast: bad header from root"); bad = CRAY_ERR_INVALID; }
    cray_scene* s = mine;
    if (!bad && !mine) {
        s = new (std::nothrow) cray_scene();
        if (!s) { set_last_error("cray_scene_broadcast: out of host memory"); bad = CRAY_ERR_HIP; }
    }
    if (!bad && !mine) {
        s->ctx = c; s->dev = h.dev; s->n_prims = h.n_prims;
        s->features = h.features; s->shade_variant = h.shade_variant;
        s->n_slots = h.n_slots;
        s->dev.inner32 = nullptr; s->dev.slots32 = nullptr; s->dev.innerh = nullptr;   // the fast-mode records are derived per rank on first use
        const void** fields[kSceneArrays];
        scene_arrays(s->dev, fields);
        for (int i = 0; i < kSceneArrays && !bad; i++) {
            void* d = nullptr;
            hipError_t err = hipMalloc(&d, h.bytes[i]);
            if (err != hipSuccess) {
                set_last_error("cray_scene_broadcast: hipMalloc(%llu) failed: %s", (unsigned long long)h.bytes[i], hipGetErrorString(err));
                (void)hipGetLastError();
                bad = CRAY_ERR_HIP;
                break;
            }
            s->allocs.push_back(d); s->alloc_bytes.push_back(h.bytes[i]); s->bytes += h.bytes[i];
            *fields[i] = d;
        }
    }
    // a rank whose header did not arrive or that could not allocate its copy tells the others before the big transfers start:
    // every rank returns the error
    if ((bad = comm_agree(c, bad))) {
        if (!mine && s) cray_scene_free(s);
        return bad;
    }
    // the big arrays (2.4 GB at 7.2 M triangles) go root HBM -> peer HBM over xGMI, one broadcast per array.  A broadcast that
    // fails here does not end the sequence on this rank (the others are inside the same sequence); the first error is returned.
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < kSceneArrays; i++) {
        const ncclResult_t r = R->Broadcast(s->allocs[i], s->allocs[i], (size_t)h.bytes[i], ncclChar, root, c->comm, c->stream);
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    const hipError_t serr = hipStreamSynchronize(c->stream);
    if (first != ncclSuccess || serr != hipSuccess) {
        set_last_error("cray_scene_broadcast: the array transfers failed: %s", first != ncclSuccess ? R->GetErrorString(first) : hipGetErrorString(serr));
        (void)hipGetLastError();
        if (!mine) cray_scene_free(s);
        return CRAY_ERR_HIP;
    }
    *out = s;
    return CRAY_OK;
}

extern "C" int cray_render_gather(cray_ctx* c, cray_scene* s, const cray_render_params* prm_in, float* out_rgb, cray_stats* stats) {
    if (!c || !prm_in) { set_last_error("cray_render_gather: null argument"); return CRAY_ERR_INVALID; }
    if (!c->comm || c->comm_world == 1) {
        cray_render_params p1 = *prm_in;
        p1.rank = 0; p1.world_size = 1;
        return cray_render(c, s, &p1, out_rgb, stats);
    }
    cray_render_params prm = *prm_in;
    prm.rank = (uint32_t)c->comm_rank; prm.world_size = (uint32_t)c->comm_world;
    HIP_TRY(hipSetDevice(c->device));
    // Whatever this rank does alone — argument checks, allocations, the render itself (out of memory, a traversal stack
    // that overflows, a HIP error) — is done first and its status agreed with the other ranks; only then does the gather start.
    int e = check_render_args(c, s, &prm);
    if (!e && c->comm_rank == 0 && !out_rgb) { set_last_error("cray_render_gather: out_rgb is null on rank 0"); e = CRAY_ERR_INVALID; }
    size_t film_floats = 0;
    float* dst = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    if (!e) {
        film_floats = (size_t)s->dev.film_w * s->dev.film_h * 3;
        if (c->comm_rank == 0) e = output_target(c, &prm, out_rgb, film_floats, &dst);
    }
    if (!e) e = render_local(c, s, &prm, stats);
    if (!e) e = gather_prepare(c, s->dev.film_w, s->dev.film_h, prm.tile_width, prm.tile_height);
    {
        const double shape[5] = {s ? (double)s->dev.film_w : 0.0, s ? (double)s->dev.film_h : 0.0, (double)prm.tile_width, (double)prm.tile_height,
                                 s ? (double)s->dev.num_samples : 0.0};
        if ((e = comm_agree(c, e, shape, 5))) return e;
    }
    const DevScene& d = s->dev;
    if ((e = gather_tiles(c, d.film_w, d.film_h, prm.tile_width, prm.tile_height, c->film, (float)d.num_samples, dst))) return e;
    if ((e = finish_output(c, &prm, c->comm_rank == 0 ? out_rgb : nullptr, film_floats))) return e;
    if (stats) stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return CRAY_OK;
}

extern "C" int cray_film_gather(cray_ctx* c, uint32_t W, uint32_t H, uint32_t tw, uint32_t th, const float* local_film, float* out_rgb, int out_is_device) {
    if (!c || !local_film || W == 0 || H == 0 || tw == 0 || th == 0) { set_last_error("cray_film_gather: bad argument"); return CRAY_ERR_INVALID; }
    if (!c->comm) { set_last_error("cray_film_gather: call cray_comm_init first"); return CRAY_ERR_INVALID; }
    if (c->comm_rank == 0 && !out_rgb) { set_last_error("cray_film_gather: out_rgb is null on rank 0"); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    cray_render_params prm;
    cray_render_params_default(&prm);
    prm.tile_width = tw; prm.tile_height = th; prm.rank = (uint32_t)c->comm_rank; prm.world_size = (uint32_t)c->comm_world;
    prm.out_is_device = out_is_device ? 1u : 0u;
    int e = ensure_pix_list(c, W, H, prm);
    const size_t film_floats = (size_t)W * H * 3;
    float* dst = nullptr;
    if (!e && c->comm_rank == 0) e = output_target(c, &prm, out_rgb, film_floats, &dst);
    if (!e) e = gather_prepare(c, W, H, tw, th);
    {
        const double shape[4] = {(double)W, (double)H, (double)tw, (double)th};
        if ((e = comm_agree(c, e, shape, 4))) return e;
    }
    if ((e = gather_tiles(c, W, H, tw, th, local_film, 1.0f, dst))) return e;
    return finish_output(c, &prm, c->comm_rank == 0 ? out_rgb : nullptr, film_floats);
}

extern "C" int cray_film_pack(cray_ctx* c, uint32_t W, uint32_t H, uint32_t tw, uint32_t th, uint32_t rank, uint32_t world, const float* film,
                              float* packed, uint64_t* n_pixels) {
    if (!c || !film || !packed || !n_pixels || W == 0 || H == 0 || tw == 0 || th == 0 || world == 0 || rank >= world) { set_last_error("cray_film_pack: bad argument"); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    cray_render_params prm;
    cray_render_params_default(&prm);
    prm.tile_width = tw; prm.tile_height = th; prm.rank = rank; prm.world_size = world;
    int e;
    if ((e = ensure_pix_list(c, W, H, prm))) return e;
    const size_t film_floats = (size_t)W * H * 3, n_mine = c->pix_count;
    DevMem mem;
    float *d_film, *d_packed;
    HIP_TRY(mem.get(&d_film, film_floats));
    HIP_TRY(mem.get(&d_packed, n_mine * 3));
    HIP_TRY(hipMemcpyAsync(d_film, film, film_floats * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (n_mine) hipLaunchKernelGGL(k_pack_tiles, dim3(grid_for(c, n_mine * 3, 8)), dim3(kBlock), 0, c->stream, (const float*)d_film, (const uint32_t*)c->pix_list, n_mine, 1.0f, d_packed);
    if (n_mine) HIP_TRY(hipMemcpyAsync(packed, d_packed, n_mine * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    *n_pixels = n_mine;
    return CRAY_OK;
}

extern "C" int cray_film_unpack(cray_ctx* c, uint32_t W, uint32_t H, uint32_t tw, uint32_t th, uint32_t world, const float* gathered, float* out) {
    if (!c || !gathered || !out || W == 0 || H == 0 || tw == 0 || th == 0 || world == 0) { set_last_error("cray_film_unpack: bad argument"); return CRAY_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    int e;
    if ((e = ensure_all_pix(c, W, H, tw, th, world))) return e;
    const size_t total = (size_t)W * H;
    DevMem mem;
    float *d_in, *d_out;
    HIP_TRY(mem.get(&d_in, total * 3));
    HIP_TRY(mem.get(&d_out, total * 3));
    HIP_TRY(hipMemcpyAsync(d_in, gathered, total * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_unpack_tiles, dim3(grid_for(c, total * 3, 8)), dim3(kBlock), 0, c->stream, (const float*)d_in, (const uint32_t*)c->all_pix, total, d_out);
    HIP_TRY(hipMemcpyAsync(out, d_out, total * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return CRAY_OK;
}
