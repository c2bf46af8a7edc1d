// cray_bvh_build.h — Bvh::new(primitives, SplitMethod::SAH) on the GPU (reference src/bvh.rs:38-56,
// 234-336; util::partition_by src/util.rs:4-26).  Included by cray_hip.hip.
//
// The reference's builder is a sequential top-down recursion, but every decision it takes is a function
// of order-independent quantities — unions of bounds (min/max, exact), integer counts, and a handful of
// f64 operations per node (surface areas, 11 costs) that gfx950 rounds like the CPU — except for the
// permutation util::partition_by leaves behind, which fixes the order of primitives inside the leaves.
// That permutation has a closed form: the two-pointer loop swaps the i-th misplaced `false` from the left
// with the i-th misplaced `true` from the right and touches nothing else.  So the SAME tree (node bounds,
// split axes, topology, leaf order) comes out of a data-parallel build:
//
//   phase L  nodes with more than kSmall primitives, one tree level per iteration, element-parallel:
//            bounds/centroid bounds (LDS-privatised atomics on order-preserving u64 keys), 12 SAH buckets,
//            per-node costs, then the partition through one global prefix sum of the predicate;
//   phase S  every subtree of <= kSmall primitives is finished by ONE thread running the reference
//            recursion verbatim (explicit stack);
//   emit     DFS pre-order index of a node = 2 * (leaves left of its segment) + (left turns on its root
//            path), so one prefix sum over the leaf starts places every node without a tree walk.
//
// The only freedom left is the sign of a zero in a bound when +0.0 and -0.0 both occur in a node (min/max
// of equal values); it never changes a comparison.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/cray.h"
#include "cray_math.h"

namespace cray {
namespace bvhb {

constexpr int kBk = 12;                 // NUM_BUCKETS, bvh.rs:235
constexpr double kTravCost = 1.0 / 8.0; // TRAVERSAL_TO_INTERSECTION_COST_RATIO, bvh.rs:236
constexpr uint32_t kMaxLeaf = 4;        // MAX_LEAF_PRIMITIVES, bvh.rs:237
#ifndef CRAY_BVH_SMALL
#define CRAY_BVH_SMALL 64
#endif
constexpr uint32_t kSmall = CRAY_BVH_SMALL;  // subtrees up to this size are built by one thread
constexpr int kTB = 256;

struct TopNode {  // a node of phase L (all interior) or the root of a phase-S subtree
    double lo[3], hi[3];
    uint32_t begin, end, split, lefts;
    int32_t axis, slot;   // slot in the active list of its level, -1 for a small-subtree root
    int32_t left, right;  // top ids of the children
};
struct SmallRec {  // node of a phase-S subtree, local DFS order at pool[2 * root.begin + k]
    double lo[3], hi[3];
    uint32_t begin, end, split, lefts;
    int32_t axis, is_leaf;
};
struct Slot {  // scratch of one active phase-L node
    unsigned long long bkey[6];  // bounds: min-keys of lo, max-keys of hi
    unsigned long long ckey[6];  // centroid bounds
    unsigned long long bk[kBk][6];  // per SAH bucket: bounds of its primitives
    unsigned long long ck[kBk][6];  // per SAH bucket: bounds of their centroids (the children's centroid bounds come from these)
    unsigned int bc[kBk];
    double c_lo, c_hi, total_area;
    int32_t axis, best;
    uint32_t node, pad;
};
struct Ctl {
    unsigned int n_top, n_small, n_next, error;
};

// order-preserving map f64 -> u64 (total order, -0 < +0), so unsigned atomic min/max reduce doubles
__device__ __forceinline__ unsigned long long enc(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ double area6(const double* lo, const double* hi) {  // bounds.rs:29-32
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
__device__ __forceinline__ int widest6(const double* lo, const double* hi) {  // bounds.rs:36-45
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx > dy && dx > dz) return 0;
    return dy > dz ? 1 : 2;
}
__device__ __forceinline__ int bucket_of(double c, double c_lo, double c_hi) {  // bvh.rs:258-263
    const double off = (c - c_lo) / (c_hi - c_lo);
    const uint64_t idx = to_u64_sat((double)kBk * off);
    return (int)(idx < (uint64_t)(kBk - 1) ? idx : (uint64_t)(kBk - 1));
}

// bvh.rs:283-313: cost of splitting after bucket s, s = 0..10; returns the first minimum.
__device__ inline int sah_best(unsigned int used, const double (*blo)[3], const double (*bhi)[3], const unsigned int* bcnt,
                               double total_area, double* best_cost, bool* nonfinite) {
    double cost[kBk - 1];
    for (int s = 0; s < kBk - 1; s++) {
        double c = kTravCost;
        for (int side = 0; side < 2; side++) {
            const int b0 = side ? s + 1 : 0, b1 = side ? kBk : s + 1;
            bool any = false;
            double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            unsigned long long mc = 0;
            for (int b = b0; b < b1; b++) {
                if (!((used >> b) & 1u)) continue;
                if (any) {
                    for (int k = 0; k < 3; k++) { lo[k] = min_nn(lo[k], blo[b][k]); hi[k] = max_nn(hi[k], bhi[b][k]); }
                    mc += bcnt[b];
                } else {
                    any = true;
                    for (int k = 0; k < 3; k++) { lo[k] = blo[b][k]; hi[k] = bhi[b][k]; }
                    mc = bcnt[b];
                }
            }
            if (any) c += (double)mc * area6(lo, hi) / total_area;
        }
        if (!isfinite(c)) *nonfinite = true;
        cost[s] = c;
    }
    int best = 0;
    for (int s = 0; s < kBk - 1; s++)
        if (cost[s] < cost[best]) best = s;
    *best_cost = cost[best];
    return best;
}

// ---------------------------------------------------------------- phase L kernels
__global__ void k_init(uint32_t* order, int32_t* slot_of, uint32_t n, int32_t root_slot) {
    const uint32_t i = blockIdx.x * kTB + threadIdx.x;
    if (i < n) { order[i] = i; slot_of[i] = root_slot; }
}

// bkey / ckey of the root are reduced from the primitives (k_bounds); every other node gets them from its parent's
// buckets in k_children, so this only resets them when asked to
__device__ __forceinline__ void slot_reset(Slot& s, uint32_t node, bool reset_bounds) {
    if (reset_bounds)
        for (int k = 0; k < 3; k++) { s.bkey[k] = ~0ull; s.bkey[3 + k] = 0ull; s.ckey[k] = ~0ull; s.ckey[3 + k] = 0ull; }
    for (int b = 0; b < kBk; b++) {
        for (int k = 0; k < 3; k++) { s.bk[b][k] = ~0ull; s.bk[b][3 + k] = 0ull; s.ck[b][k] = ~0ull; s.ck[b][3 + k] = 0ull; }
        s.bc[b] = 0;
    }
    s.node = node;
}

__global__ void k_root(TopNode* top, Slot* slots, uint32_t* active, uint32_t* small_list, Ctl* ctl, uint32_t n) {
    TopNode t{};
    t.begin = 0; t.end = n; t.lefts = 0; t.left = t.right = -1;
    ctl->n_top = 1; ctl->error = 0; ctl->n_next = 0;
    if (n > kSmall) { t.slot = 0; active[0] = 0; slot_reset(slots[0], 0, true); ctl->n_small = 0; }
    else { t.slot = -1; small_list[0] = 0; ctl->n_small = 1; }
    top[0] = t;
}

// bounds and centroid bounds of every active node (bvh.rs:239, 252-255)
__global__ void __launch_bounds__(kTB) k_bounds(const double* __restrict__ box, const uint32_t* __restrict__ order,
                                                const int32_t* __restrict__ slot_of, Slot* slots, uint32_t n) {
    __shared__ unsigned long long l_key[12];
    const uint32_t base = blockIdx.x * kTB, pos = base + threadIdx.x;
    const uint32_t last = base + kTB - 1 < n - 1 ? base + kTB - 1 : n - 1;
    const int32_t s_first = slot_of[base], s_last = slot_of[last];
    const bool uniform = s_first >= 0 && s_first == s_last;  // segments are contiguous
    const int32_t s = pos < n ? slot_of[pos] : -1;
    if (!uniform && s < 0) return;
    if (uniform && !__syncthreads_or(s >= 0)) return;
    unsigned long long key[12];
    if (s >= 0) {
        const double* b = box + 6 * (size_t)order[pos];
        for (int k = 0; k < 3; k++) {
            const double c = (b[k] + b[3 + k]) * 0.5;  // centroid as in Bvh::new (bvh.rs:44-49)
            key[k] = enc(b[k]); key[3 + k] = enc(b[3 + k]);
            key[6 + k] = enc(c); key[9 + k] = enc(c);
        }
    }
    if (uniform) {
        if (threadIdx.x < 12) l_key[threadIdx.x] = (threadIdx.x % 6) < 3 ? ~0ull : 0ull;
        __syncthreads();
        if (s >= 0)
            for (int k = 0; k < 12; k++) {
                if ((k % 6) < 3) atomicMin(&l_key[k], key[k]); else atomicMax(&l_key[k], key[k]);
            }
        __syncthreads();
        if (threadIdx.x < 12) {
            Slot& sl = slots[s_first];
            unsigned long long* dst = threadIdx.x < 6 ? &sl.bkey[threadIdx.x] : &sl.ckey[threadIdx.x - 6];
            if ((threadIdx.x % 6) < 3) atomicMin(dst, l_key[threadIdx.x]); else atomicMax(dst, l_key[threadIdx.x]);
        }
    } else {
        Slot& sl = slots[s];
        for (int k = 0; k < 6; k++) {
            if (k < 3) { atomicMin(&sl.bkey[k], key[k]); atomicMin(&sl.ckey[k], key[6 + k]); }
            else { atomicMax(&sl.bkey[k], key[k]); atomicMax(&sl.ckey[k], key[6 + k]); }
        }
    }
}

// per active node: bounds -> node, surface-area assert, split axis (bvh.rs:239-256)
__global__ void k_setup(TopNode* top, Slot* slots, uint32_t n_active, Ctl* ctl) {
    const uint32_t a = blockIdx.x * kTB + threadIdx.x;
    if (a >= n_active) return;
    Slot& s = slots[a];
    TopNode& t = top[s.node];
    double clo[3], chi[3];
    for (int k = 0; k < 3; k++) {
        t.lo[k] = dec(s.bkey[k]); t.hi[k] = dec(s.bkey[3 + k]);
        clo[k] = dec(s.ckey[k]); chi[k] = dec(s.ckey[3 + k]);
    }
    s.total_area = area6(t.lo, t.hi);
    if (!(s.total_area > 0.0)) atomicMax(&ctl->error, 1u);  // assert!, bvh.rs:245
    const int axis = widest6(clo, chi);
    s.axis = axis; t.axis = axis;
    s.c_lo = clo[axis]; s.c_hi = chi[axis];
}

// SAH buckets of every active node (bvh.rs:266-281); remembers each primitive's bucket for the partition
__global__ void __launch_bounds__(kTB) k_buckets(const double* __restrict__ box, const uint32_t* __restrict__ order,
                                                 const int32_t* __restrict__ slot_of, Slot* slots, uint8_t* __restrict__ bidx, uint32_t n) {
    __shared__ unsigned long long l_key[kBk][6], l_ckey[kBk][6];
    __shared__ unsigned int l_cnt[kBk];
    const uint32_t base = blockIdx.x * kTB, pos = base + threadIdx.x;
    const uint32_t last = base + kTB - 1 < n - 1 ? base + kTB - 1 : n - 1;
    const int32_t s_first = slot_of[base], s_last = slot_of[last];
    const bool uniform = s_first >= 0 && s_first == s_last;
    const int32_t s = pos < n ? slot_of[pos] : -1;
    if (!uniform && s < 0) return;
    if (uniform && !__syncthreads_or(s >= 0)) return;
    int b = 0;
    unsigned long long key[6], ckey[3];
    if (s >= 0) {
        const Slot& sl = slots[s];
        const double* bx = box + 6 * (size_t)order[pos];
        const int axis = sl.axis;
        double cen[3];
        for (int k = 0; k < 3; k++) { cen[k] = (bx[k] + bx[3 + k]) * 0.5; ckey[k] = enc(cen[k]); }
        b = bucket_of(cen[axis], sl.c_lo, sl.c_hi);
        bidx[pos] = (uint8_t)b;
        for (int k = 0; k < 6; k++) key[k] = enc(bx[k]);
    }
    if (uniform) {
        if (threadIdx.x < kBk * 6) {
            l_key[threadIdx.x / 6][threadIdx.x % 6] = (threadIdx.x % 6) < 3 ? ~0ull : 0ull;
            l_ckey[threadIdx.x / 6][threadIdx.x % 6] = (threadIdx.x % 6) < 3 ? ~0ull : 0ull;
        }
        if (threadIdx.x < kBk) l_cnt[threadIdx.x] = 0;
        __syncthreads();
        if (s >= 0) {
            for (int k = 0; k < 3; k++) {
                atomicMin(&l_key[b][k], key[k]); atomicMax(&l_key[b][3 + k], key[3 + k]);
                atomicMin(&l_ckey[b][k], ckey[k]); atomicMax(&l_ckey[b][3 + k], ckey[k]);
            }
            atomicAdd(&l_cnt[b], 1u);
        }
        __syncthreads();
        if (threadIdx.x < kBk * 6) {
            const int bb = threadIdx.x / 6, k = threadIdx.x % 6;
            if (l_cnt[bb]) {
                Slot& sl = slots[s_first];
                if (k < 3) { atomicMin(&sl.bk[bb][k], l_key[bb][k]); atomicMin(&sl.ck[bb][k], l_ckey[bb][k]); }
                else { atomicMax(&sl.bk[bb][k], l_key[bb][k]); atomicMax(&sl.ck[bb][k], l_ckey[bb][k]); }
                if (k == 0) atomicAdd(&sl.bc[bb], l_cnt[bb]);
            }
        }
    } else {
        Slot& sl = slots[s];
        for (int k = 0; k < 3; k++) {
            atomicMin(&sl.bk[b][k], key[k]); atomicMax(&sl.bk[b][3 + k], key[3 + k]);
            atomicMin(&sl.ck[b][k], ckey[k]); atomicMax(&sl.ck[b][3 + k], ckey[k]);
        }
        atomicAdd(&sl.bc[b], 1u);
    }
}

// per active node: costs and the split bucket (bvh.rs:283-319). A phase-L node has more than
// MAX_LEAF_PRIMITIVES primitives, so it never becomes a leaf.
__global__ void k_choose(Slot* slots, uint32_t n_active, Ctl* ctl) {
    const uint32_t a = blockIdx.x * kTB + threadIdx.x;
    if (a >= n_active) return;
    Slot& s = slots[a];
    double blo[kBk][3], bhi[kBk][3];
    unsigned int used = 0;
    for (int b = 0; b < kBk; b++) {
        if (s.bc[b]) used |= 1u << b;
        for (int k = 0; k < 3; k++) { blo[b][k] = dec(s.bk[b][k]); bhi[b][k] = dec(s.bk[b][3 + k]); }
    }
    double best_cost;
    bool nonfinite = false;
    s.best = sah_best(used, blo, bhi, s.bc, s.total_area, &best_cost, &nonfinite);
    if (nonfinite) atomicMax(&ctl->error, 2u);  // assert!(cost.is_finite()), bvh.rs:304
}

// predicate of partition_by (bvh.rs:322-325) for every position
__global__ void k_flags(const int32_t* __restrict__ slot_of, const Slot* __restrict__ slots, const uint8_t* __restrict__ bidx,
                        uint8_t* __restrict__ flag, uint32_t n) {
    const uint32_t pos = blockIdx.x * kTB + threadIdx.x;
    if (pos >= n) return;
    const int32_t s = slot_of[pos];
    flag[pos] = s >= 0 ? (uint8_t)((int)bidx[pos] <= slots[s].best) : (uint8_t)0;
}

// ---- exclusive prefix sum of a u8 array into T[0..n] (T[n] = total): three small kernels
constexpr uint32_t kScanPer = 4, kScanTile = kTB * kScanPer;
__global__ void __launch_bounds__(kTB) k_scan_tiles(const uint8_t* __restrict__ in, uint32_t* __restrict__ T, uint32_t* __restrict__ tile_sum, uint32_t n) {
    __shared__ uint32_t wave_tot[kTB / 64];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
    uint32_t v[kScanPer], sum = 0;
    for (uint32_t k = 0; k < kScanPer; k++) { v[k] = base + k < n ? in[base + k] : 0u; sum += v[k]; }
    // inclusive scan of `sum` across the block
    uint32_t x = sum;
    const unsigned int lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
    if (lane == 63) wave_tot[w] = x;
    __syncthreads();
    uint32_t off = 0;
    for (unsigned int i = 0; i < w; i++) off += wave_tot[i];
    uint32_t run = off + x - sum;  // exclusive prefix of this thread inside the tile
    for (uint32_t k = 0; k < kScanPer; k++) { if (base + k < n) T[base + k] = run; run += v[k]; }
    if (threadIdx.x == kTB - 1) tile_sum[blockIdx.x] = off + x;
}
__global__ void __launch_bounds__(kTB) k_scan_sums(uint32_t* tile_sum, uint32_t n_tiles, uint32_t* total) {
    __shared__ uint32_t wave_tot[kTB / 64];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += kTB) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_sum[i] : 0u;
        uint32_t x = v;
        const unsigned int lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
        if (lane == 63) wave_tot[w] = x;
        __syncthreads();
        uint32_t off = carry;
        for (unsigned int k = 0; k < w; k++) off += wave_tot[k];
        if (i < n_tiles) tile_sum[i] = off + x - v;
        __syncthreads();
        if (threadIdx.x == kTB - 1) carry = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(kTB) k_scan_add(uint32_t* __restrict__ T, const uint32_t* __restrict__ tile_sum, const uint32_t* __restrict__ total, uint32_t n) {
    const uint32_t i = blockIdx.x * kTB + threadIdx.x;
    if (i < n) T[i] += tile_sum[i / kScanTile];
    if (i == 0) T[n] = *total;
}

// partition_by as a permutation (util.rs:4-26): the k-th misplaced `false` from the left (position < split)
// trades places with the k-th misplaced `true` from the right; everything else stays.
__global__ void k_pairs(const int32_t* __restrict__ slot_of, const Slot* __restrict__ slots, const TopNode* __restrict__ top,
                        const uint8_t* __restrict__ flag, const uint32_t* __restrict__ T, uint32_t* __restrict__ tmp_true, uint32_t n) {
    const uint32_t pos = blockIdx.x * kTB + threadIdx.x;
    if (pos >= n) return;
    const int32_t s = slot_of[pos];
    if (s < 0) return;
    const TopNode& t = top[slots[s].node];
    const uint32_t split = t.begin + (T[t.end] - T[t.begin]);
    if (flag[pos] && pos >= split) tmp_true[t.begin + (T[t.end] - T[pos + 1])] = pos;
}
__global__ void k_swap(const int32_t* __restrict__ slot_of, const Slot* __restrict__ slots, const TopNode* __restrict__ top,
                       const uint8_t* __restrict__ flag, const uint32_t* __restrict__ T, const uint32_t* __restrict__ tmp_true,
                       uint32_t* __restrict__ order, uint8_t* __restrict__ bidx, uint32_t n) {
    const uint32_t pos = blockIdx.x * kTB + threadIdx.x;
    if (pos >= n) return;
    const int32_t s = slot_of[pos];
    if (s < 0) return;
    const TopNode& t = top[slots[s].node];
    const uint32_t split = t.begin + (T[t.end] - T[t.begin]);
    if (!flag[pos] && pos < split) {
        const uint32_t k = (pos - t.begin) - (T[pos] - T[t.begin]);  // falses before pos in the segment
        const uint32_t other = tmp_true[t.begin + k];
        const uint32_t a = order[pos], b = order[other];
        order[pos] = b; order[other] = a;
    }
}

// per active node: children (bvh.rs:327-335); large ones go to the next level, small ones to phase S
__global__ void k_children(TopNode* top, Slot* slots, Slot* next_slots, uint32_t n_active, const uint32_t* __restrict__ T,
                           uint32_t* next_active, uint32_t* small_list, Ctl* ctl) {
    const uint32_t a = blockIdx.x * kTB + threadIdx.x;
    if (a >= n_active) return;
    const uint32_t id = slots[a].node;
    TopNode& t = top[id];
    const uint32_t split = t.begin + (T[t.end] - T[t.begin]);
    t.split = split;
    if (split == t.begin || split == t.end) { atomicMax(&ctl->error, 3u); return; }  // assert!, bvh.rs:327-328
    const uint32_t c0 = atomicAdd(&ctl->n_top, 2u);
    t.left = (int32_t)c0; t.right = (int32_t)c0 + 1;
    for (int side = 0; side < 2; side++) {
        TopNode c{};
        c.begin = side ? split : t.begin;
        c.end = side ? t.end : split;
        c.lefts = t.lefts + (side ? 0u : 1u);
        c.left = c.right = -1;
        if (c.end - c.begin > kSmall) {
            const uint32_t sl = atomicAdd(&ctl->n_next, 1u);
            c.slot = (int32_t)sl;
            next_active[sl] = c0 + side;
            Slot& ns = next_slots[sl];
            slot_reset(ns, c0 + side, false);
            // the child's bounds / centroid bounds are the unions over the buckets on its side of the split:
            // min / max of the order-preserving keys (empty buckets hold the neutral elements)
            const Slot& ps = slots[a];
            const int b0 = side ? ps.best + 1 : 0, b1 = side ? kBk : ps.best + 1;
            for (int k = 0; k < 3; k++) {
                unsigned long long blo = ~0ull, bhi = 0ull, clo = ~0ull, chi = 0ull;
                for (int b = b0; b < b1; b++) {
                    blo = ps.bk[b][k] < blo ? ps.bk[b][k] : blo; bhi = ps.bk[b][3 + k] > bhi ? ps.bk[b][3 + k] : bhi;
                    clo = ps.ck[b][k] < clo ? ps.ck[b][k] : clo; chi = ps.ck[b][3 + k] > chi ? ps.ck[b][3 + k] : chi;
                }
                ns.bkey[k] = blo; ns.bkey[3 + k] = bhi; ns.ckey[k] = clo; ns.ckey[3 + k] = chi;
            }
        } else {
            c.slot = -1;
            small_list[atomicAdd(&ctl->n_small, 1u)] = c0 + side;
        }
        top[c0 + side] = c;
    }
}

__global__ void k_reslot(int32_t* __restrict__ slot_of, const Slot* __restrict__ slots, const TopNode* __restrict__ top, uint32_t n) {
    const uint32_t pos = blockIdx.x * kTB + threadIdx.x;
    if (pos >= n) return;
    const int32_t s = slot_of[pos];
    if (s < 0) return;
    const TopNode& t = top[slots[s].node];
    if (t.left < 0) { slot_of[pos] = -1; return; }  // build error: stop
    slot_of[pos] = top[pos < t.split ? t.left : t.right].slot;
}

// ---------------------------------------------------------------- phase S: one thread per small subtree
__global__ void __launch_bounds__(64) k_small(const double* __restrict__ box, uint32_t* __restrict__ order, const TopNode* __restrict__ top,
                                              const uint32_t* __restrict__ small_list, uint32_t n_small, SmallRec* __restrict__ pool,
                                              uint32_t* __restrict__ rec_count, uint8_t* __restrict__ leaf_flag, Ctl* ctl) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_small) return;
    const TopNode& root = top[small_list[i]];
    struct Ent { uint32_t b, e, lefts; };
    Ent st[kSmall + 1];
    uint8_t bk[kSmall];
    int sp = 0;
    st[sp++] = Ent{root.begin, root.end, root.lefts};
    SmallRec* out = pool + 2 * (size_t)root.begin;
    uint32_t cnt = 0;
    while (sp) {
        const Ent en = st[--sp];
        const uint32_t b0 = en.b, n = en.e - en.b;
        SmallRec rec{};
        rec.begin = en.b; rec.end = en.e; rec.split = en.b; rec.lefts = en.lefts;
        {
            const double* bx = box + 6 * (size_t)order[b0];
            for (int k = 0; k < 3; k++) { rec.lo[k] = bx[k]; rec.hi[k] = bx[3 + k]; }
            for (uint32_t j = 1; j < n; j++) {
                bx = box + 6 * (size_t)order[b0 + j];
                for (int k = 0; k < 3; k++) { rec.lo[k] = min_nn(rec.lo[k], bx[k]); rec.hi[k] = max_nn(rec.hi[k], bx[3 + k]); }
            }
        }
        bool leaf = n <= 1;
        int axis = 0, best = 0;
        if (!leaf) {
            const double total_area = area6(rec.lo, rec.hi);
            if (!(total_area > 0.0)) { atomicMax(&ctl->error, 1u); leaf = true; }
            else {
                double clo[3], chi[3];
                for (uint32_t j = 0; j < n; j++) {
                    const double* bx = box + 6 * (size_t)order[b0 + j];
                    for (int k = 0; k < 3; k++) {
                        const double c = (bx[k] + bx[3 + k]) * 0.5;
                        clo[k] = j ? min_nn(clo[k], c) : c;
                        chi[k] = j ? max_nn(chi[k], c) : c;
                    }
                }
                axis = widest6(clo, chi);
                const double c_lo = clo[axis], c_hi = chi[axis];
                double blo[kBk][3], bhi[kBk][3];
                unsigned int bcnt[kBk];
                unsigned int used = 0;
                for (int q = 0; q < kBk; q++) bcnt[q] = 0;
                for (uint32_t j = 0; j < n; j++) {
                    const double* bx = box + 6 * (size_t)order[b0 + j];
                    const int q = bucket_of((bx[axis] + bx[3 + axis]) * 0.5, c_lo, c_hi);
                    bk[j] = (uint8_t)q;
                    if ((used >> q) & 1u) {
                        for (int k = 0; k < 3; k++) { blo[q][k] = min_nn(blo[q][k], bx[k]); bhi[q][k] = max_nn(bhi[q][k], bx[3 + k]); }
                        bcnt[q]++;
                    } else {
                        used |= 1u << q;
                        for (int k = 0; k < 3; k++) { blo[q][k] = bx[k]; bhi[q][k] = bx[3 + k]; }
                        bcnt[q] = 1;
                    }
                }
                double best_cost;
                bool nonfinite = false;
                best = sah_best(used, blo, bhi, bcnt, total_area, &best_cost, &nonfinite);
                if (nonfinite) atomicMax(&ctl->error, 2u);
                if ((double)n <= best_cost && n <= kMaxLeaf) leaf = true;  // bvh.rs:316-319
            }
        }
        uint32_t split = 0;
        if (!leaf) {
            uint32_t l = 0, r = n - 1;  // util::partition_by, pred = bucket <= best
            while (l != r) {
                while (l < r && (int)bk[l] <= best) l++;
                while (r > l && !((int)bk[r] <= best)) r--;
                const uint32_t to = order[b0 + l]; order[b0 + l] = order[b0 + r]; order[b0 + r] = to;
                const uint8_t tb = bk[l]; bk[l] = bk[r]; bk[r] = tb;
            }
            split = (int)bk[l] <= best ? l + 1 : l;
            if (split == 0 || split == n) { atomicMax(&ctl->error, 3u); leaf = true; }
        }
        if (leaf) {
            rec.is_leaf = 1;
            leaf_flag[b0] = 1;
            out[cnt++] = rec;
            continue;
        }
        rec.is_leaf = 0; rec.axis = axis; rec.split = b0 + split;
        out[cnt++] = rec;
        st[sp++] = Ent{b0 + split, en.e, en.lefts};      // right, visited after the whole left subtree
        st[sp++] = Ent{b0, b0 + split, en.lefts + 1u};   // left
    }
    rec_count[i] = cnt;
}

// ---------------------------------------------------------------- emit in DFS pre-order
__device__ __forceinline__ void emit(cray_bvh_node* out, const double* lo, const double* hi, uint32_t begin, uint32_t end, uint32_t split,
                                     uint32_t lefts, int axis, bool is_leaf, const uint32_t* __restrict__ leaf_rank) {
    const uint32_t me = 2u * leaf_rank[begin] + lefts;
    cray_bvh_node nd;
    for (int k = 0; k < 3; k++) { nd.bmin[k] = lo[k]; nd.bmax[k] = hi[k]; }
    nd.left = is_leaf ? 0u : me + 1u;
    nd.right = is_leaf ? 0u : 2u * leaf_rank[split] + lefts;
    nd.first = is_leaf ? begin : 0u;
    nd.count = is_leaf ? end - begin : 0u;
    nd.axis = is_leaf ? 0 : axis;
    nd.is_leaf = is_leaf ? 1 : 0;
    out[me] = nd;
}
__global__ void k_emit_top(const TopNode* __restrict__ top, uint32_t n_top, const uint32_t* __restrict__ leaf_rank, cray_bvh_node* out) {
    const uint32_t i = blockIdx.x * kTB + threadIdx.x;
    if (i >= n_top) return;
    const TopNode& t = top[i];
    if (t.left < 0) return;  // root of a small subtree: emitted by k_emit_small
    emit(out, t.lo, t.hi, t.begin, t.end, t.split, t.lefts, t.axis, false, leaf_rank);
}
__global__ void k_emit_small(const TopNode* __restrict__ top, const uint32_t* __restrict__ small_list, uint32_t n_small,
                             const SmallRec* __restrict__ pool, const uint32_t* __restrict__ rec_count,
                             const uint32_t* __restrict__ leaf_rank, cray_bvh_node* out) {
    const uint32_t i = blockIdx.x * kTB + threadIdx.x;
    if (i >= n_small) return;
    const SmallRec* recs = pool + 2 * (size_t)top[small_list[i]].begin;
    const uint32_t cnt = rec_count[i];
    for (uint32_t k = 0; k < cnt; k++) {
        const SmallRec& r = recs[k];
        emit(out, r.lo, r.hi, r.begin, r.end, r.split, r.lefts, r.axis, r.is_leaf != 0, leaf_rank);
    }
}

}  // namespace bvhb
}  // namespace cray
