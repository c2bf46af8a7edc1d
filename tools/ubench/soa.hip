// 8-byte vs 16-byte per-lane accesses over the same bytes: is the texture addresser what limits k_shade / k_raygen?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct A8 { double* a[6]; double* b[6]; };
struct A16 { double2* a[3]; double2* b[3]; };
__global__ void __launch_bounds__(256) k8(A8 p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v[6];
        for (int k = 0; k < 6; k++) v[k] = p.a[k][i];
        for (int k = 0; k < 6; k++) p.b[k][i] = v[k] + v[(k + 1) % 6];
    }
}
__global__ void __launch_bounds__(256) k16(A16 p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double2 v[3];
        for (int k = 0; k < 3; k++) v[k] = p.a[k][i];
        for (int k = 0; k < 3; k++) p.b[k][i] = make_double2(v[k].x + v[(k + 1) % 3].y, v[k].y + v[(k + 1) % 3].x);
    }
}
__global__ void __launch_bounds__(256) w8(A8 p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        for (int k = 0; k < 6; k++) p.b[k][i] = (double)i + k;
}
__global__ void __launch_bounds__(256) w16(A16 p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        for (int k = 0; k < 3; k++) p.b[k][i] = make_double2((double)i + k, (double)i - k);
}
int main() {
    const size_t n = 64u << 20;
    double* pool;
    CK(hipMalloc(&pool, n * 8 * 12));
    CK(hipMemset(pool, 0, n * 8 * 12));
    A8 p8; A16 p16;
    for (int k = 0; k < 6; k++) { p8.a[k] = pool + n * k; p8.b[k] = pool + n * (6 + k); }
    for (int k = 0; k < 3; k++) { p16.a[k] = (double2*)(pool + 2 * n * k); p16.b[k] = (double2*)(pool + n * 6 + 2 * n * k); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {256 * 4, 256 * 8, 256 * 16}) {
        for (int which = 0; which < 4; which++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(k8, dim3(grid), dim3(256), 0, 0, p8, n);
                else if (which == 1) hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, p16, n);
                else if (which == 2) hipLaunchKernelGGL(w8, dim3(grid), dim3(256), 0, 0, p8, n);
                else hipLaunchKernelGGL(w16, dim3(grid), dim3(256), 0, 0, p16, n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double bytes = (which < 2 ? 12.0 : 6.0) * n * 8;
            printf("grid %5d %-28s %7.3f ms  %7.1f GB/s\n", grid, which == 0 ? "read 6 + write 6 x 8 B" : which == 1 ? "read 3 + write 3 x 16 B" : which == 2 ? "write 6 x 8 B" : "write 3 x 16 B", best, bytes / best / 1e6);
        }
    }
    return 0;
}
