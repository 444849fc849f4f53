// Measurement aid (not product code): memory floor of the marginal-gather sweep's access pattern at C3 size.
// Per directed edge: 4-B index (stream), random 32-B gather from an N*4-double table with planted-partition
// locality (77 % of targets in the row's own quarter), 32-B own message read + write (stream); per row a 32-B write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int EPT, int LOOKUP = 0>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ nbr, const double2* __restrict__ psi, double2* __restrict__ M,
                                             double2* __restrict__ psi_new, uint32_t n_edges, uint32_t n_rows,
                                             const uint32_t* __restrict__ blk_e0 = nullptr) {
    // LOOKUP: the workgroup's edge offset comes from a table (cold scalar load) instead of blockIdx arithmetic
    const uint32_t base = (LOOKUP ? blk_e0[blockIdx.x] : blockIdx.x * 256 * EPT) + threadIdx.x;
    if (LOOKUP) n_edges = min(n_edges, blk_e0[blockIdx.x + 1]);  // ragged segment end, as the real kernel
    uint32_t l[EPT]; double2 a[EPT][2], m[EPT][2];
#pragma unroll
    for (int j = 0; j < EPT; ++j) { uint32_t k = base + j * 256; l[j] = k < n_edges ? nbr[k] : 0; }
#pragma unroll
    for (int j = 0; j < EPT; ++j) { a[j][0] = psi[size_t(l[j]) * 2]; a[j][1] = psi[size_t(l[j]) * 2 + 1]; }
#pragma unroll
    for (int j = 0; j < EPT; ++j) { uint32_t k = base + j * 256; if (k < n_edges) { m[j][0] = M[size_t(k) * 2]; m[j][1] = M[size_t(k) * 2 + 1]; } }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        uint32_t k = base + j * 256;
        if (k < n_edges) {
            M[size_t(k) * 2] = make_double2(m[j][0].x * 0.5 + a[j][0].x * 0.5, m[j][0].y * 0.5 + a[j][0].y * 0.5);
            M[size_t(k) * 2 + 1] = make_double2(m[j][1].x * 0.5 + a[j][1].x * 0.5, m[j][1].y * 0.5 + a[j][1].y * 0.5);
            if (k % 10 == 0 && k / 10 < n_rows) { psi_new[size_t(k / 10) * 2] = a[j][0]; psi_new[size_t(k / 10) * 2 + 1] = a[j][1]; }
        }
    }
}
// own message kept as Q-1 = 3 components (24-byte records, contiguous per wave): -8 B read, -8 B write per edge
template <int EPT>
__global__ void __launch_bounds__(256) probe3(const uint32_t* __restrict__ nbr, const double2* __restrict__ psi, double2* __restrict__ M2,
                                              double2* __restrict__ psi_new, uint32_t n_edges, uint32_t n_rows,
                                              const uint32_t* __restrict__ blk_e0 = nullptr) {
    double* __restrict__ M = reinterpret_cast<double*>(M2);
#ifdef PROBE_XCD  // every XCD (workgroup i runs on XCD i % 8) takes a contiguous eighth of the segments, as k_sweep_psi does
    const uint32_t per = (gridDim.x + 7) / 8;
    const uint32_t bid = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (bid >= gridDim.x) return;
#else
    const uint32_t bid = blockIdx.x;
#endif
    const uint32_t base = blk_e0[bid] + threadIdx.x;
    n_edges = min(n_edges, blk_e0[bid + 1]);
    uint32_t l[EPT]; double2 a[EPT][2]; double m[EPT][3];
#pragma unroll
    for (int j = 0; j < EPT; ++j) { uint32_t k = base + j * 256; l[j] = k < n_edges ? nbr[k] : 0; }
#pragma unroll
    for (int j = 0; j < EPT; ++j) { a[j][0] = psi[size_t(l[j]) * 2]; a[j][1] = psi[size_t(l[j]) * 2 + 1]; }
#pragma unroll
    for (int j = 0; j < EPT; ++j) { uint32_t k = base + j * 256; if (k < n_edges) { m[j][0] = M[size_t(k) * 3]; m[j][1] = M[size_t(k) * 3 + 1]; m[j][2] = M[size_t(k) * 3 + 2]; } }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        uint32_t k = base + j * 256;
        if (k < n_edges) {
            M[size_t(k) * 3] = m[j][0] * 0.5 + a[j][0].x * 0.5;
            M[size_t(k) * 3 + 1] = m[j][1] * 0.5 + a[j][0].y * 0.5;
            M[size_t(k) * 3 + 2] = m[j][2] * 0.5 + a[j][1].x * 0.5 + a[j][1].y * 1e-9;
            if (k % 10 == 0 && k / 10 < n_rows) { psi_new[size_t(k / 10) * 2] = a[j][0]; psi_new[size_t(k / 10) * 2 + 1] = a[j][1]; }
        }
    }
}
// same, records staged through LDS so that global accesses are whole 16-byte lanes over the workgroup's contiguous span
template <int EPT>
__global__ void __launch_bounds__(256) probe3_lds(const uint32_t* __restrict__ nbr, const double2* __restrict__ psi, double2* __restrict__ M2,
                                                  double2* __restrict__ psi_new, uint32_t n_edges, uint32_t n_rows,
                                                  const uint32_t* __restrict__ blk_e0 = nullptr) {
    __shared__ double stage[256 * EPT * 3];
    double* __restrict__ M = reinterpret_cast<double*>(M2);
    const uint32_t e0 = blk_e0[blockIdx.x];
    const uint32_t e1 = min(n_edges, blk_e0[blockIdx.x + 1]);
    const uint32_t cnt = e1 - e0, nd = cnt * 3;
    uint32_t l[EPT]; double2 a[EPT][2];
#pragma unroll
    for (int j = 0; j < EPT; ++j) { uint32_t k = threadIdx.x + j * 256; l[j] = k < cnt ? nbr[e0 + k] : 0; }
#pragma unroll
    for (int j = 0; j < EPT; ++j) { a[j][0] = psi[size_t(l[j]) * 2]; a[j][1] = psi[size_t(l[j]) * 2 + 1]; }
    const double* src = M + size_t(e0) * 3;
    for (uint32_t i = threadIdx.x; i < nd; i += 256) stage[i] = src[i];
    __syncthreads();
    double o[EPT][3];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        uint32_t k = threadIdx.x + j * 256;
        if (k < cnt) {
            o[j][0] = stage[k * 3] * 0.5 + a[j][0].x * 0.5;
            o[j][1] = stage[k * 3 + 1] * 0.5 + a[j][0].y * 0.5;
            o[j][2] = stage[k * 3 + 2] * 0.5 + a[j][1].x * 0.5 + a[j][1].y * 1e-9;
            if ((e0 + k) % 10 == 0 && (e0 + k) / 10 < n_rows) { psi_new[size_t((e0 + k) / 10) * 2] = a[j][0]; psi_new[size_t((e0 + k) / 10) * 2 + 1] = a[j][1]; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j) { uint32_t k = threadIdx.x + j * 256; if (k < cnt) { stage[k * 3] = o[j][0]; stage[k * 3 + 1] = o[j][1]; stage[k * 3 + 2] = o[j][2]; } }
    __syncthreads();
    double* dst = M + size_t(e0) * 3;
    for (uint32_t i = threadIdx.x; i < nd; i += 256) dst[i] = stage[i];
}
int main(int argc, char** argv) {
    const uint32_t N = 10000000;
    uint32_t E = 100000000;
    std::vector<uint32_t> h(E);
    std::mt19937_64 rng(1);
    const uint32_t G = N / 4;
    for (uint32_t k = 0; k < E; ++k) {
        uint32_t row = k / 10, g = row / G;
        uint64_t r = rng();
        uint32_t tg = ((r & 0xffff) < 0.77 * 65536) ? g : uint32_t((g + 1 + ((r >> 16) % 3)) % 4);
        h[k] = tg * G + uint32_t((r >> 20) % G);
    }
    if (argc > 1 && argv[1][0] && argv[1][0] != '-') {  // the real graph's neighbour array (uint32 binary dumped by tools/dump_nbr.py)
        FILE* f = fopen(argv[1], "rb");
        if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
        fseek(f, 0, SEEK_END); E = uint32_t(ftell(f) / 4); fseek(f, 0, SEEK_SET);
        h.resize(E);
        if (fread(h.data(), 4, E, f) != E) return 1;
        fclose(f);
        printf("using %u real neighbour indices from %s\n", E, argv[1]);
    }
    const uint32_t SEG = (argc > 2) ? uint32_t(atoi(argv[2])) : 512;  // edges per workgroup when looked up (real segments: ~505)
    std::vector<uint32_t> hb((E + SEG - 1) / SEG + 1);
    for (size_t b = 0; b < hb.size(); ++b) hb[b] = uint32_t(b * SEG);
    printf("lookup segments of %u edges\n", SEG);
    uint32_t* blk; CK(hipMalloc(&blk, hb.size() * 4)); CK(hipMemcpy(blk, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    uint32_t* nbr; double2 *psi, *M, *psin;
    CK(hipMalloc(&nbr, size_t(E) * 4)); CK(hipMalloc(&psi, size_t(N) * 32)); CK(hipMalloc(&M, size_t(E) * 32)); CK(hipMalloc(&psin, size_t(N) * 32));
    CK(hipMemcpy(nbr, h.data(), size_t(E) * 4, hipMemcpyHostToDevice));
    CK(hipMemset(psi, 0, size_t(N) * 32)); CK(hipMemset(M, 0, size_t(E) * 32));
    if (!(argc > 3 && argv[3][0] == 'z')) {  // random data by default: zero-filled tables read 9 % faster (pass 'z' to see it)
        std::vector<double> r(1 << 24);
        for (auto& x : r) x = double(rng() >> 11) / 9007199254740992.0;
        for (size_t off = 0; off < size_t(N) * 4; off += r.size()) CK(hipMemcpy((double*)psi + off, r.data(), std::min(r.size(), size_t(N) * 4 - off) * 8, hipMemcpyHostToDevice));
        for (size_t off = 0; off < size_t(E) * 4; off += r.size()) CK(hipMemcpy((double*)M + off, r.data(), std::min(r.size(), size_t(E) * 4 - off) * 8, hipMemcpyHostToDevice));
        printf("tables filled with random data\n");
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, int ept, const char* name) {
        dim3 grid(ept == 2 ? uint32_t(hb.size() - 1) : (E + 256 * ept - 1) / (256 * ept));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, nbr, psi, M, psin, E, N, blk);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, nbr, psi, M, psin, E, N, blk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = double(E) * (3 * 32 + 4) + double(N) * 40;
        printf("%s: %.3f ms/launch, algorithmic %.0f GB/s (%.1f%% of 8 TB/s)\n", name, ms / 10, bytes / (ms / 10 * 1e-3) / 1e9, bytes / (ms / 10 * 1e-3) / 8e12 * 100);
    };
    run(probe<2>, 2, "probe EPT=2                     ");
    run(probe<2, 1>, 2, "probe EPT=2 + bounds table lookup");
    run(probe3<2>, 2, "probe EPT=2, 24-B own messages   ");
    run(probe3_lds<2>, 2, "probe EPT=2, 24-B msgs via LDS   ");
    run(probe<1>, 1, "probe EPT=1                     ");
    run(probe<4>, 4, "probe EPT=4                     ");
    return 0;
}
