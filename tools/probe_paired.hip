// Measurement aid: memory floor of a "paired record" layout for the message sweep at C3 size (random data).
// One 64-byte record per undirected edge = [m(lo->hi) | m(hi->lo)], stored with the lower endpoint's rows.
// Per directed edge: 4-B record index (stream), ONE 64-B record read (owner side: stream; other side: random),
// one 32-B half-record write into a second buffer (owner side: strided stream; other side: random). Per row: 32-B write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int EPT>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ rec, const double2* __restrict__ A, double2* __restrict__ B,
                                             double2* __restrict__ psi_new, uint32_t n_edges, uint32_t n_rows) {
    const uint32_t base = blockIdx.x * 256 * EPT + threadIdx.x;
    uint32_t r[EPT]; double2 v[EPT][4];
#pragma unroll
    for (int j = 0; j < EPT; ++j) { uint32_t k = base + j * 256; r[j] = k < n_edges ? rec[k] : 0; }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const double2* p = A + size_t(r[j] >> 1) * 4;
        v[j][0] = p[0]; v[j][1] = p[1]; v[j][2] = p[2]; v[j][3] = p[3];
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        uint32_t k = base + j * 256;
        if (k < n_edges) {
            const uint32_t side = r[j] & 1;
            double2* q = B + size_t(r[j] >> 1) * 4 + side * 2;
            q[0] = make_double2(v[j][0].x * 0.5 + v[j][2].x * 0.5, v[j][0].y * 0.5 + v[j][2].y * 0.5);
            q[1] = make_double2(v[j][1].x * 0.5 + v[j][3].x * 0.5, v[j][1].y * 0.5 + v[j][3].y * 0.5);
            if (k % 10 == 0 && k / 10 < n_rows) { psi_new[size_t(k / 10) * 2] = v[j][0]; psi_new[size_t(k / 10) * 2 + 1] = v[j][1]; }
        }
    }
}
int main() {
    const uint32_t N = 10000000, E = 100000000, U = E / 2;  // U undirected edges = records
    // directed edge k of row k/10: the first 5 edges of a row go to lower neighbours (records owned elsewhere: random,
    // with planted-partition locality), the last 5 are the row's own records (contiguous: record ids row*5 .. row*5+4)
    std::vector<uint32_t> h(E);
    std::mt19937_64 rng(1);
    const uint32_t G = N / 4;
    for (uint32_t k = 0; k < E; ++k) {
        uint32_t row = k / 10, j = k % 10, g = row / G;
        if (j >= 5) { h[k] = ((row * 5 + (j - 5)) << 1) | 0u; continue; }
        uint64_t r = rng();
        uint32_t tg = ((r & 0xffff) < 0.77 * 65536) ? g : uint32_t((g + 1 + ((r >> 16) % 3)) % 4);
        uint32_t owner = tg * G + uint32_t((r >> 20) % G);
        h[k] = ((owner * 5 + uint32_t((r >> 45) % 5)) << 1) | 1u;
    }
    uint32_t* rec; double2 *A, *B, *psin;
    CK(hipMalloc(&rec, size_t(E) * 4)); CK(hipMalloc(&A, size_t(U) * 64)); CK(hipMalloc(&B, size_t(U) * 64)); CK(hipMalloc(&psin, size_t(N) * 32));
    CK(hipMemcpy(rec, h.data(), size_t(E) * 4, hipMemcpyHostToDevice));
    std::vector<double> rd(1 << 24);
    for (auto& x : rd) x = double(rng() >> 11) / 9007199254740992.0;
    for (size_t off = 0; off < size_t(U) * 8; off += rd.size()) {
        size_t n = std::min(rd.size(), size_t(U) * 8 - off);
        CK(hipMemcpy((double*)A + off, rd.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy((double*)B + off, rd.data(), n * 8, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, int ept, const char* name) {
        dim3 grid((E + 256 * ept - 1) / (256 * ept));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, rec, w & 1 ? B : A, w & 1 ? A : B, psin, E, N);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, rec, it & 1 ? B : A, it & 1 ? A : B, psin, E, N);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = double(E) * (3 * 32 + 4) + double(N) * 40;
        printf("%s: %.3f ms/launch, algorithmic %.0f GB/s (%.1f%% of 8 TB/s)\n", name, ms / 10, bytes / (ms / 10 * 1e-3) / 1e9, bytes / (ms / 10 * 1e-3) / 8e12 * 100);
    };
    run(probe<1>, 1, "paired records EPT=1"); run(probe<2>, 2, "paired records EPT=2"); run(probe<4>, 4, "paired records EPT=4");
    return 0;
}
