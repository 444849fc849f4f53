// Measurement aid (not product code): what a random 32-byte row gather costs on MI355X, alone, by how it is issued and by the
// kind of memory the table lives in. 1e8 gathers from a table of 1e7 rows of 32 bytes (320 MB, the C3 marginal table), indices
// with the planted-partition locality of the C3 graph. Question behind it: the sweep's gathers cost one 64-byte fabric request
// per 32-byte row (TCC_EA0_RDREQ_32B = 0); can the same rows be had for 32-byte requests?
//   hipcc -O3 --offload-arch=gfx950 tools/probe_gather32.hip -o tools/probe_gather32 && tools/probe_gather32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

// MODE 0: a lane loads its row as two 16-byte loads. MODE 1: a lane PAIR loads a row with ONE 16-byte load per lane (two rows
// per pair and trip, halves exchanged by a shuffle). NT: non-temporal loads.
template <int MODE, bool NT>
__global__ void __launch_bounds__(256) gather(const uint32_t *__restrict__ idx, const v2d *__restrict__ tab, uint32_t n, double *__restrict__ out) {
    const uint32_t k0 = (blockIdx.x * 256 + threadIdx.x) * 2;  // two gathers per lane
    double acc = 0.0;
    if (k0 + 1 < n) {
        const uint32_t l0 = idx[k0], l1 = idx[k0 + 1];
        if (MODE == 0) {
            v2d a0, a1, b0, b1;
            if (NT) { a0 = __builtin_nontemporal_load(tab + size_t(l0) * 2); a1 = __builtin_nontemporal_load(tab + size_t(l0) * 2 + 1);
                      b0 = __builtin_nontemporal_load(tab + size_t(l1) * 2); b1 = __builtin_nontemporal_load(tab + size_t(l1) * 2 + 1); }
            else { a0 = tab[size_t(l0) * 2]; a1 = tab[size_t(l0) * 2 + 1]; b0 = tab[size_t(l1) * 2]; b1 = tab[size_t(l1) * 2 + 1]; }
            acc = a0.x + a0.y + a1.x + a1.y + b0.x + b0.y + b1.x + b1.y;
        } else {
            // the pair (lane, lane ^ 1) serves rows l0, l1 of the even lane first, then those of the odd lane: 4 rows, 4 trips of
            // one 16-byte load per lane; lane parity picks the half
            const int odd = threadIdx.x & 1;
            const uint32_t m0 = __shfl_xor(l0, 1), m1 = __shfl_xor(l1, 1);
            const uint32_t r[4] = {odd ? m0 : l0, odd ? m1 : l1, odd ? l0 : m0, odd ? l1 : m1};  // rows of the even lane, then of the odd lane
            v2d h[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) h[t] = NT ? __builtin_nontemporal_load(tab + size_t(r[t]) * 2 + odd) : tab[size_t(r[t]) * 2 + odd];
            // every lane now holds one half of four rows; the sum over both halves of its own two rows needs the partner's halves
            double mine = 0.0, theirs = 0.0;
            mine += odd ? (h[2].x + h[2].y + h[3].x + h[3].y) : (h[0].x + h[0].y + h[1].x + h[1].y);
            theirs += odd ? (h[0].x + h[0].y + h[1].x + h[1].y) : (h[2].x + h[2].y + h[3].x + h[3].y);
            acc = mine + __shfl_xor(theirs, 1);
        }
    }
    // keep the loads alive without adding traffic: one value per workgroup
    __shared__ double s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < 256; ++i) t += s[i]; out[blockIdx.x] = t; }
}

template <int MODE, bool NT>
static double run(const char *what, const uint32_t *idx, const v2d *tab, uint32_t n, double *out, double *check) {
    const uint32_t grid = (n / 2 + 255) / 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((gather<MODE, NT>), dim3(grid), dim3(256), 0, 0, idx, tab, n, out);
    CK(hipEventRecord(a));
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((gather<MODE, NT>), dim3(grid), dim3(256), 0, 0, idx, tab, n, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<double> h(grid);
    CK(hipMemcpy(h.data(), out, size_t(grid) * 8, hipMemcpyDeviceToHost));
    double t = 0.0; for (double x : h) t += x;
    printf("%-58s %.3f ms per pass  (%.2f G gathers/s, %.2f TB/s at 64 B each, %.2f TB/s at 32 B)  checksum %.6e%s\n", what, ms / 5, n / (ms / 5) * 1e-6,
           n * 64.0 / (ms / 5) * 1e-9, n * 32.0 / (ms / 5) * 1e-9, t, (*check != 0.0 && fabs(t - *check) > 1e-6 * fabs(*check)) ? "  MISMATCH" : "");
    if (*check == 0.0) *check = t;
    return ms / 5;
}

int main() {
    const uint32_t N = 10000000, E = 100000000, G = N / 4;
    std::vector<uint32_t> h(E);
    std::mt19937_64 rng(1);
    for (uint32_t k = 0; k < E; ++k) {
        const uint32_t g = (k / 10) / G;
        const uint64_t r = rng();
        const uint32_t tg = ((r & 0xffff) < 0.77 * 65536) ? g : uint32_t((g + 1 + ((r >> 16) % 3)) % 4);
        h[k] = tg * G + uint32_t((r >> 20) % G);
    }
    uint32_t *idx; double *out;
    CK(hipMalloc(&idx, size_t(E) * 4)); CK(hipMemcpy(idx, h.data(), size_t(E) * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, size_t(E / 512 + 2) * 8));
    std::vector<double> r(size_t(N) * 4);
    for (auto &x : r) x = double(rng() >> 11) / 9007199254740992.0;
    struct { const char *name; unsigned flags; } kinds[] = {{"hipMalloc", 0xffffffffu}, {"hipExtMallocWithFlags(Uncached)", hipDeviceMallocUncached},
                                                           {"hipExtMallocWithFlags(Finegrained)", hipDeviceMallocFinegrained}};
    double check = 0.0;
    for (auto &kd : kinds) {
        v2d *tab = nullptr;
        hipError_t e = kd.flags == 0xffffffffu ? hipMalloc(&tab, size_t(N) * 32) : hipExtMallocWithFlags(reinterpret_cast<void **>(&tab), size_t(N) * 32, kd.flags);
        if (e != hipSuccess) { printf("%s: %s\n", kd.name, hipGetErrorString(e)); continue; }
        CK(hipMemcpy(tab, r.data(), size_t(N) * 32, hipMemcpyHostToDevice));
        char what[160];
        snprintf(what, sizeof what, "%s, row per lane", kd.name);                     run<0, false>(what, idx, tab, E, out, &check);
        snprintf(what, sizeof what, "%s, row per lane, non-temporal", kd.name);      run<0, true>(what, idx, tab, E, out, &check);
        snprintf(what, sizeof what, "%s, row per lane PAIR", kd.name);                run<1, false>(what, idx, tab, E, out, &check);
        snprintf(what, sizeof what, "%s, row per lane PAIR, non-temporal", kd.name); run<1, true>(what, idx, tab, E, out, &check);
        CK(hipFree(tab));
    }
    return 0;
}
