// Microbenchmark behind the record-fetch design of k_trace: a persistent wave whose lanes chase pointers through
// random 128-byte records (the BVH node records) with ~150 dependent f64 operations per step, 4 waves per SIMD.
//   A  every lane loads its own record with 7 x 16-B loads (what k_trace does): 7 x 64 distinct-line requests per wave step
//   B  8 lanes load one record with ONE 16-B load each (8 records per load instruction, coalesced per record), the data
//      goes through LDS to the owning lane: 8 x 8 line requests per wave step + 8 ds_write_b128 + 7 ds_read_b128
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

template <int MODE>
__global__ void __launch_bounds__(kBlock, 4) chase(const double2* __restrict__ table, uint32_t n_rec, uint32_t hot_n, uint32_t hot, uint32_t steps, int n_ops, double* out,
                                                   unsigned long long* sink) {
    __shared__ double2 stage[(kBlock / 64) * 64 * kRow];
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
    }
    hipMemcpy(table, h.data(), (size_t)n_rec * 128, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 4;
    const uint32_t steps = 2000;
    for (int cfg = 0; cfg < 6; cfg++) {
        const int n_ops = cfg < 3 ? 154 : 308;
        const uint32_t hot = (cfg % 3) * 3;          // 0, 3, 6 of 8 steps hit the 4 MiB hot set
        const uint32_t hot_n = 32768;
        for (int mode = 0; mode < 2; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(grid), dim3(kBlock), 0, 0, table, n_rec, hot_n, hot, steps, n_ops, out, sink);
                else hipLaunchKernelGGL(chase<1>, dim3(grid), dim3(kBlock), 0, 0, table, n_rec, hot_n, hot, steps, n_ops, out, sink);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) printf("hot %u/8 ops/step %3d  mode %s  %.2f ms  %.2f G lane-steps/s\n", hot, n_ops, mode ? "B cooperative+LDS" : "A per-lane       ", ms,
                                     (double)grid * kBlock * steps / (ms * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
