// Calibration of rocprofv3's FETCH_SIZE for THIS access pattern (MI355X_MICROARCH.md "HBM":
// calibrate on a known byte count in your own access pattern): every lane reads whole 128-byte
// records (8 x 16-B loads) at random record indices of a table much larger than the 256 MiB
// Infinity Cache, each record exactly once.  Known bytes = n_records * 128.
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch tools/calib_fetch.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void gather128(const double2* __restrict__ table, const uint32_t* __restrict__ idx, size_t n, double* out) {
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2* rec = table + (size_t)idx[i] * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) { double2 v = rec[k]; acc += v.x + v.y; }
    }
    if (acc == 123.456) out[0] = acc;
}
__global__ void stream16(const double2* __restrict__ table, size_t n16, double* out) {
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { double2 v = table[i]; acc += v.x + v.y; }
    if (acc == 123.456) out[0] = acc;
}
int main() {
    const size_t n = (size_t)24 << 20;  // 24 Mi records x 128 B = 3 GiB
    double2* table; uint32_t* idx; double* out;
    hipMalloc(&table, n * 128); hipMalloc(&idx, n * 4); hipMalloc(&out, 8);
    hipMemset(table, 0, n * 128);
    std::vector<uint32_t> h(n); std::iota(h.begin(), h.end(), 0u);
    std::mt19937_64 rng(1); std::shuffle(h.begin(), h.end(), rng);
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gather128, dim3(4096), dim3(256), 0, 0, table, idx, n, out);
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, table, n * 8, out);
    hipDeviceSynchronize();
    printf("gather128: %zu records, %zu bytes; idx %zu bytes; stream16: %zu bytes\n", n, n * 128, n * 4, n * 128);
    return 0;
}
