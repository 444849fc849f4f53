// Measurement aid: which structural ingredient of k_sweep_psi costs time beyond the memory floor?
// Same access pattern as probe_floor.hip (EPT=2, 512 edges per workgroup, rows of 10 edges), plus:
//   V0 nothing; V1 26 KB LDS per workgroup (occupancy of the real kernel); V2 V1 + b -> LDS, barrier, row-lane loop over
//   LDS, barrier, phase 3 re-reads LDS; V3 V2 + the real arithmetic volume (2*16 FMAs + 10 divisions per edge).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int V>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ nbr, const double2* __restrict__ psi, double2* __restrict__ M,
                                             double2* __restrict__ psi_new, uint32_t n_edges, uint32_t n_rows) {
    constexpr int EPT = 2;
    __shared__ double sb[V >= 1 ? 512 * 4 : 1];
    __shared__ double sA[V >= 1 ? 256 * 4 : 1];
    __shared__ unsigned short pad[V >= 1 ? 1024 : 1];
    const uint32_t base = blockIdx.x * 256 * EPT + threadIdx.x;
    uint32_t l[EPT]; double2 a[EPT][2], m[EPT][2];
#pragma unroll
    for (int j = 0; j < EPT; ++j) { uint32_t k = base + j * 256; l[j] = k < n_edges ? nbr[k] : 0; }
#pragma unroll
    for (int j = 0; j < EPT; ++j) { a[j][0] = psi[size_t(l[j]) * 2]; a[j][1] = psi[size_t(l[j]) * 2 + 1]; }
#pragma unroll
    for (int j = 0; j < EPT; ++j) { uint32_t k = base + j * 256; if (k < n_edges) { m[j][0] = M[size_t(k) * 2]; m[j][1] = M[size_t(k) * 2 + 1]; } }
    if (V >= 2) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int le = j * 256 + threadIdx.x;
            double b0 = a[j][0].x + m[j][0].x, b1 = a[j][0].y + m[j][0].y, b2 = a[j][1].x + m[j][1].x, b3 = a[j][1].y + m[j][1].y;
            if (V >= 3) {
                double w = 1.0000001;
#pragma unroll
                for (int r = 0; r < 8; ++r) { b0 = fma(b0, w, b1); b1 = fma(b1, w, b2); b2 = fma(b2, w, b3); b3 = fma(b3, w, b0); }
                const double t = (a[j][0].x + 2.0) / (b0 + 3.0) + (a[j][0].y + 2.0) / (b1 + 3.0) + (a[j][1].x + 2.0) / (b2 + 3.0) + (a[j][1].y + 2.0) / (b3 + 3.0);
                b0 += 1.0 / (t + 1.0);
            }
            sb[le * 4] = b0; sb[le * 4 + 1] = b1; sb[le * 4 + 2] = b2; sb[le * 4 + 3] = b3;
            pad[le] = (unsigned short)(le / 10);
        }
        __syncthreads();
        if (threadIdx.x < 52) {  // ~51 rows of 10 edges per workgroup
            double A0 = 1, A1 = 1, A2 = 1, A3 = 1;
            const int es = threadIdx.x * 10;
            for (int e = es; e < es + 10 && e < 512; ++e) { A0 *= sb[e * 4] + 1.0; A1 *= sb[e * 4 + 1] + 1.0; A2 *= sb[e * 4 + 2] + 1.0; A3 *= sb[e * 4 + 3] + 1.0; }
            sA[threadIdx.x * 4] = A0; sA[threadIdx.x * 4 + 1] = A1; sA[threadIdx.x * 4 + 2] = A2; sA[threadIdx.x * 4 + 3] = A3;
            const uint32_t row = blockIdx.x * 51 + threadIdx.x;
            if (row < n_rows && threadIdx.x < 51) { psi_new[size_t(row) * 2] = make_double2(A0, A1); psi_new[size_t(row) * 2 + 1] = make_double2(A2, A3); }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        uint32_t k = base + j * 256;
        if (k < n_edges) {
            double c0 = 0.5, c1 = 0.5, c2 = 0.5, c3 = 0.5;
            if (V >= 2) {
                const int le = j * 256 + threadIdx.x; const int r = pad[le];
                c0 = sA[r * 4] * 1e-300 + sb[le * 4] * 1e-300 + 0.5; c1 = sA[r * 4 + 1] * 1e-300 + 0.5; c2 = sA[r * 4 + 2] * 1e-300 + 0.5; c3 = sA[r * 4 + 3] * 1e-300 + 0.5;
                if (V >= 3) { c0 = c0 / (sb[le * 4] + 2.0) + 0.5; c1 = c1 / (sb[le * 4 + 1] + 2.0) + 0.5; c2 = c2 / (sb[le * 4 + 2] + 2.0) + 0.5; c3 = c3 / (sb[le * 4 + 3] + 2.0) + 0.5; c0 *= 1.0 / (c0 + c1 + c2 + c3); }
            }
            M[size_t(k) * 2] = make_double2(m[j][0].x * c0 + a[j][0].x * 0.5, m[j][0].y * c1 + a[j][0].y * 0.5);
            M[size_t(k) * 2 + 1] = make_double2(m[j][1].x * c2 + a[j][1].x * 0.5, m[j][1].y * c3 + a[j][1].y * 0.5);
            if (V < 2 && k % 10 == 0 && k / 10 < n_rows) { psi_new[size_t(k / 10) * 2] = a[j][0]; psi_new[size_t(k / 10) * 2 + 1] = a[j][1]; }
        }
    }
    if (V == 1 && threadIdx.x == 0 && n_edges == 7) { sb[0] = 1; sA[0] = sb[0]; pad[0] = 1; psi_new[0] = make_double2(sA[0], pad[0]); }
}
int main() {
    const uint32_t N = 10000000, E = 100000000;
    std::vector<uint32_t> h(E);
    std::mt19937_64 rng(1);
    const uint32_t G = N / 4;
    for (uint32_t k = 0; k < E; ++k) {
        uint32_t row = k / 10, g = row / G;
        uint64_t r = rng();
        uint32_t tg = ((r & 0xffff) < 0.77 * 65536) ? g : uint32_t((g + 1 + ((r >> 16) % 3)) % 4);
        h[k] = tg * G + uint32_t((r >> 20) % G);
    }
    uint32_t* nbr; double2 *psi, *M, *psin;
    CK(hipMalloc(&nbr, size_t(E) * 4)); CK(hipMalloc(&psi, size_t(N) * 32)); CK(hipMalloc(&M, size_t(E) * 32)); CK(hipMalloc(&psin, size_t(N) * 32));
    CK(hipMemcpy(nbr, h.data(), size_t(E) * 4, hipMemcpyHostToDevice));
    CK(hipMemset(psi, 0, size_t(N) * 32)); CK(hipMemset(M, 0, size_t(E) * 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* name) {
        dim3 grid((E + 511) / 512);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, nbr, psi, M, psin, E, N);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, nbr, psi, M, psin, E, N);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.3f ms/launch\n", name, ms / 10);
    };
    run(probe<0>, "V0 memory ops only         ");
    run(probe<1>, "V1 + 26 KB LDS footprint    ");
    run(probe<2>, "V2 + LDS round trip, 2 barriers, row phase");
    run(probe<3>, "V3 + arithmetic volume      ");
    return 0;
}
