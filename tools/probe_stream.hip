// Measurement aid: achievable device stream-copy bandwidth (16 B per lane, random data), SURVEY 8(d).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void __launch_bounds__(256) copy16(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) read16(const double2* __restrict__ a, double2* __restrict__ out, size_t n) {
    double2 s = make_double2(0, 0);
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) { double2 v = a[i]; s.x += v.x; s.y += v.y; }
    if (s.x == 1.2345) out[0] = s;
}
int main() {
    const size_t n = size_t(3200) << 16;  // 3.2 GiB-ish of double2 = 16 B each
    double2 *a, *b;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    std::vector<double> r(1 << 24); std::mt19937_64 g(1);
    for (auto& x : r) x = double(g() >> 11) / 9007199254740992.0;
    for (size_t off = 0; off < n * 2; off += r.size()) CK(hipMemcpy((double*)a + off, r.data(), std::min(r.size(), n * 2 - off) * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {2048, 8192, 65536}) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy  grid %6d: %.3f ms, %.0f GB/s (read+write)\n", grid, ms / 10, 2.0 * n * 16 / (ms / 10 * 1e-3) / 1e9);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("read  grid %6d: %.3f ms, %.0f GB/s\n", grid, ms / 10, 1.0 * n * 16 / (ms / 10 * 1e-3) / 1e9);
    }
    return 0;
}
