// Microbenchmark behind the record-fetch design of k_trace: a persistent wave whose lanes chase pointers through
// random 128-byte records (the BVH node records) with ~150 dependent f64 operations per step, 4 waves per SIMD.
//   A  every lane loads its own record with 7 x 16-B loads (what k_trace does): 7 x 64 distinct-line requests per wave step
//   B  8 lanes load one record with ONE 16-B load each (8 records per load instruction, coalesced per record), the data
//      goes through LDS to the owning lane: 8 x 8 line requests per wave step + 8 ds_write_b128 + 7 ds_read_b128
//   C  (round 3) the four lanes of a quad load one record's 64-B half with ONE 16-B load each (coalesced: one L1 access per half
//      instead of four), the 4 x 4 blocks are transposed inside the quad with DPP moves (no LDS): 8 load instructions of 16
//      accesses per wave step instead of 7 of 64, paid with ~64 more VALU instructions
//   D  calibration: per-lane loads of only the first 4 x 16 B of the record (what a 64-B record would cost)
//   hipcc --offload-arch=gfx950 -O3 -o fetch_pattern tools/fetch_pattern.hip && ./fetch_pattern
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kBlock = 256;
constexpr int kRow = 9;  // 16-B units per staged record row (8 + 1 pad: bank-conflict-free reads)

__device__ __forceinline__ double work(double x, const double2* r, int n_ops) {
    double a = x;
    for (int i = 0; i < n_ops; i += 14) {
#pragma unroll
        for (int k = 0; k < 7; k++) { a = fma(a, r[k].x, r[k].y); a = fma(a, 0.999, r[k].x); }
    }
    return a;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }

template <int MODE, int WAVES = 4>
__global__ void __launch_bounds__(kBlock, WAVES) chase(const double2* __restrict__ table, uint32_t n_rec, uint32_t hot_n, uint32_t hot, uint32_t steps, int n_ops, double* out,
                                                   unsigned long long* sink, uint32_t* last_idx) {
    __shared__ double2 stage[MODE == 1 ? (kBlock / 64) * 64 * kRow : 1];   // only mode B stages through LDS (37 KB would cap the other modes at 4 blocks per CU)
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    double2* st = stage + wave * 64 * kRow;
    uint32_t idx = (blockIdx.x * kBlock + threadIdx.x) * 2654435761u % n_rec;
    double acc = 1.0;
    for (uint32_t s = 0; s < steps; s++) {
        double2 r[7];
        if (MODE == 0) {
            const double2* rec = table + (size_t)idx * 8;
#pragma unroll
            for (int k = 0; k < 7; k++) r[k] = rec[k];
        } else if (MODE == 3) {
            const double2* rec = table + (size_t)idx * 8;
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = rec[k];
            r[4] = r[0]; r[5] = r[1]; r[6] = r[2];
            r[6].y = r[3].y;   // calibration table: the successor also sits in the fourth unit
        } else if (MODE == 4) {
            // as C, with the select folded into the DPP move (v_cndmask_b32_dpp): 2 instead of 4 VALU instructions per exchanged dword pair
            const unsigned int j = lane & 3u;
            uint4 L[4][2];
            const uint32_t oi[4] = {dpp<0x00>(idx), dpp<0x55>(idx), dpp<0xAA>(idx), dpp<0xFF>(idx)};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4* rec = reinterpret_cast<const uint4*>(table + (size_t)oi[k] * 8) + j;
                L[k][0] = rec[0];
                L[k][1] = rec[4];
            }
#define XCH4(A, B, KEEP_A, CTRL)                                                                       \
            {                                                                                           \
                uint4 na_, nb_;                                                                         \
                asm volatile("s_nop 4\n\ts_mov_b64 vcc, %16\n\t"                                       \
                             "v_cndmask_b32_dpp %0, %12, %8, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"  \
                             "v_cndmask_b32_dpp %1, %13, %9, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"  \
                             "v_cndmask_b32_dpp %2, %14, %10, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                             "v_cndmask_b32_dpp %3, %15, %11, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                             "s_not_b64 vcc, vcc\n\t"                                                   \
                             "v_cndmask_b32_dpp %4, %8, %12, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"  \
                             "v_cndmask_b32_dpp %5, %9, %13, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"  \
                             "v_cndmask_b32_dpp %6, %10, %14, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                             "v_cndmask_b32_dpp %7, %11, %15, vcc " CTRL " row_mask:0xf bank_mask:0xf"      \
                             : "=&v"(na_.x), "=&v"(na_.y), "=&v"(na_.z), "=&v"(na_.w), "=&v"(nb_.x), "=&v"(nb_.y), "=&v"(nb_.z), "=&v"(nb_.w) \
                             : "v"((A).x), "v"((A).y), "v"((A).z), "v"((A).w), "v"((B).x), "v"((B).y), "v"((B).z), "v"((B).w), "s"(KEEP_A)    \
                             : "vcc", "scc");                                                           \
                (A) = na_; (B) = nb_;                                                                   \
            }
            const unsigned long long keep_even = 0x5555555555555555ull, keep_low = 0x3333333333333333ull;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                XCH4(L[0][h], L[1][h], keep_even, "quad_perm:[1,0,3,2]")
                XCH4(L[2][h], L[3][h], keep_even, "quad_perm:[1,0,3,2]")
                XCH4(L[0][h], L[2][h], keep_low, "quad_perm:[2,3,0,1]")
                XCH4(L[1][h], L[3][h], keep_low, "quad_perm:[2,3,0,1]")
            }
#undef XCH4
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const uint4 v = L[k & 3][k >> 2];
                r[k].x = __hiloint2double((int)v.y, (int)v.x);
                r[k].y = __hiloint2double((int)v.w, (int)v.z);
            }
        } else if (MODE == 2) {
            // quad-cooperative fetch: load (k, h) reads bytes [64 h, 64 h + 64) of the record of quad lane k, 16 B per lane
            const unsigned int j = lane & 3u;
            uint4 L[4][2];
            const uint32_t oi[4] = {dpp<0x00>(idx), dpp<0x55>(idx), dpp<0xAA>(idx), dpp<0xFF>(idx)};   // quad_perm:[k,k,k,k]
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4* rec = reinterpret_cast<const uint4*>(table + (size_t)oi[k] * 8) + j;
                L[k][0] = rec[0];
                L[k][1] = rec[4];
            }
            // 4 x 4 transpose of 16-B blocks inside the quad: two butterfly stages of select + DPP move per dword
            const bool odd = (lane & 1u) != 0, hi = (lane & 2u) != 0;
#define XCH(A, B, SEL, CTRL)                                                                              \
            {                                                                                              \
                const uint32_t a_ = (A), b_ = (B);                                                         \
                const uint32_t pa_ = dpp<CTRL>(a_);                                                        \
                const uint32_t pb_ = dpp<CTRL>(b_);                                                        \
                (A) = (SEL) ? pb_ : a_;                                                                    \
                (B) = (SEL) ? b_ : pa_;                                                                    \
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int a = 0; a < 4; a += 2) {
                    XCH(L[a][h].x, L[a + 1][h].x, odd, 0xB1) XCH(L[a][h].y, L[a + 1][h].y, odd, 0xB1)
                    XCH(L[a][h].z, L[a + 1][h].z, odd, 0xB1) XCH(L[a][h].w, L[a + 1][h].w, odd, 0xB1)
                }
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    XCH(L[m][h].x, L[m + 2][h].x, hi, 0x4E) XCH(L[m][h].y, L[m + 2][h].y, hi, 0x4E)
                    XCH(L[m][h].z, L[m + 2][h].z, hi, 0x4E) XCH(L[m][h].w, L[m + 2][h].w, hi, 0x4E)
                }
            }
#undef XCH
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const uint4 v = L[k & 3][k >> 2];
                r[k].x = __hiloint2double((int)v.y, (int)v.x);
                r[k].y = __hiloint2double((int)v.w, (int)v.z);
            }
        } else {
#pragma unroll
            for (int rd = 0; rd < 8; rd++) {
                const unsigned int owner = rd * 8 + (lane >> 3), piece = lane & 7u;
                const uint32_t oidx = __shfl(idx, owner);
                const double2 v = table[(size_t)oidx * 8 + piece];
                st[owner * kRow + piece] = v;
            }
            __builtin_amdgcn_s_waitcnt(0);  // wave-local hand-off through LDS
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 7; k++) r[k] = st[lane * kRow + k];
            __builtin_amdgcn_wave_barrier();
        }
        acc = work(acc, r, n_ops);
        // next record: data dependent (the payload holds a random successor in r[6].y's bits)
        {   // data dependent AND different per lane and step (a pure function of the record would make all chains
            // fall into the same few short cycles of the random functional graph, i.e. into the caches)
            uint32_t hsh = (uint32_t)(__double_as_longlong(r[6].y) & 0xffffffffull) ^ ((blockIdx.x * kBlock + threadIdx.x) * 0x9E3779B9u + s * 0x85EBCA6Bu);
            hsh ^= hsh >> 16; hsh *= 0x7feb352du; hsh ^= hsh >> 15; hsh *= 0x846ca68bu; hsh ^= hsh >> 16;
            // `hot` of every 8 steps stay inside the first hot_n records (the top of the tree, cache resident)
            idx = ((hsh >> 24) & 7u) < hot ? (hsh % hot_n) : (hsh % n_rec);
        }
        if (acc == 0.12345) idx ^= 1u;
    }
    if (acc == 123.456) out[0] = acc;
    last_idx[blockIdx.x * kBlock + threadIdx.x] = idx;   // the chain's end: equal in every mode (checked on the host)
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1ull);
}

int main() {
    const uint32_t n_rec = 6u << 20;  // 6 Mi records x 128 B = 768 MiB (the dragon's interior-node array is 0.81 GB)
    double2* table; double* out; unsigned long long* sink;
    hipMalloc(&table, (size_t)n_rec * 128); hipMalloc(&out, 8); hipMalloc(&sink, 8); hipMemset(sink, 0, 8);
    std::vector<double2> h((size_t)n_rec * 8);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < (size_t)n_rec; i++) {
        for (int k = 0; k < 8; k++) { h[i * 8 + k].x = 1.0 + 1e-9 * (double)(i & 1023); h[i * 8 + k].y = 1e-12; }
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        long long bits = (long long)(x & 0xffffffffull);
        double d; __builtin_memcpy(&d, &bits, 8);
        h[i * 8 + 6].y = d;  // successor index in the low 32 bits (a denormal double)
        h[i * 8 + 3].y = d;  // (mode D reads it from the first half)
    }
    hipMemcpy(table, h.data(), (size_t)n_rec * 128, hipMemcpyHostToDevice);
    const int grid = 256 * 4;
    constexpr int kMaxGrid = 256 * 8;   // the occupancy rows launch up to 8 blocks per CU: every launch writes last[blockIdx.x * kBlock + threadIdx.x]
    uint32_t* last; hipMalloc(&last, (size_t)kMaxGrid * kBlock * 4);
    std::vector<uint32_t> ref((size_t)grid * kBlock), got((size_t)grid * kBlock);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t steps = 2000;
    for (int cfg = 0; cfg < 6; cfg++) {
        const int n_ops = cfg < 3 ? 154 : 308;
        const uint32_t hot = (cfg % 3) * 3;          // 0, 3, 6 of 8 steps hit the 4 MiB hot set
        const uint32_t hot_n = 32768;
        for (int mode = 0; mode < 5; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(grid), dim3(kBlock), 0, 0, table, n_rec, hot_n, hot, steps, n_ops, out, sink, last);
                else if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(grid), dim3(kBlock), 0, 0, table, n_rec, hot_n, hot, steps, n_ops, out, sink, last);
                else if (mode == 4) hipLaunchKernelGGL(chase<4>, dim3(grid), dim3(kBlock), 0, 0, table, n_rec, hot_n, hot, steps, n_ops, out, sink, last);
                else if (mode == 3) hipLaunchKernelGGL(chase<3>, dim3(grid), dim3(kBlock), 0, 0, table, n_rec, hot_n, hot, steps, n_ops, out, sink, last);
                else hipLaunchKernelGGL(chase<1>, dim3(grid), dim3(kBlock), 0, 0, table, n_rec, hot_n, hot, steps, n_ops, out, sink, last);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    hipMemcpy(got.data(), last, got.size() * 4, hipMemcpyDeviceToHost);
                    if (mode == 0) ref = got;
                    else if (got != ref) printf("  MISMATCH: mode %d ends its chains elsewhere than mode A\n", mode);
                }
                if (rep == 1) printf("hot %u/8 ops/step %3d  mode %s  %.2f ms  %.2f G lane-steps/s\n", hot, n_ops, mode == 0 ? "A per-lane        " : mode == 1 ? "B cooperative+LDS " : mode == 2 ? "C quad + DPP      " : mode == 4 ? "E quad + cndmask_dpp" : "D per-lane 4 loads", ms,
                                     (double)grid * kBlock * steps / (ms * 1e-3) / 1e9);
            }
        }
    }
    // occupancy: the per-lane fetch (mode A) with 2 / 4 / 8 waves per SIMD resident (grid = 256 CUs x waves blocks of 4 waves)
    for (uint32_t hot : {0u, 3u}) {
        for (int waves : {2, 4, 8}) {
            float best = 1e9f;
            const int g = 256 * waves;
            if (g > kMaxGrid) { printf("grid too large\n"); return 1; }
            const uint32_t st = 2000u * 4u / (uint32_t)waves;   // the same number of lane-steps in every row
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                if (waves == 8) hipLaunchKernelGGL((chase<0, 8>), dim3(g), dim3(kBlock), 0, 0, table, n_rec, 32768u, hot, st, 154, out, sink, last);
                else hipLaunchKernelGGL((chase<0, 4>), dim3(g), dim3(kBlock), 0, 0, table, n_rec, 32768u, hot, st, 154, out, sink, last);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("occupancy: hot %u/8, %d waves per SIMD: %.2f ms  %.2f G lane-steps/s\n", hot, waves, best, (double)g * kBlock * st / (best * 1e-3) / 1e9);
        }
    }
    // launch floor: the same chase with few steps and few blocks (what a nearly empty bounce of the traversal looks like):
    // time = fixed cost of a launch that touches a 768 MB table + steps x (dependent fetch + work)
    for (int g : {1024, 64}) {
        for (uint32_t st : {25u, 50u, 100u, 200u, 400u}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(chase<0>, dim3(g), dim3(kBlock), 0, 0, table, n_rec, 32768u, 0u, st, 154, out, sink, last);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("floor: %4d blocks, %3u dependent steps: %.3f ms = %.2f us per step\n", g, st, best, best * 1e3 / st);
        }
    }
    return 0;
}
