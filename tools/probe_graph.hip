// Measurement aid: what a chain of DEPENDENT short kernels costs per link when it is launched kernel by kernel on a stream
// and when the same chain is replayed as a hipGraph (is the gap between dependent kernels a host cost or a device cost?).
//   hipcc -O2 --offload-arch=gfx950 tools/probe_graph.hip -o tools/probe_graph && tools/probe_graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_busy(double *p, int iters) {
    double v = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;
    if (v == 12345.678) p[0] = v;  // never
}
int main() {
    double *d;
    CHK(hipMalloc(&d, 4096));
    CHK(hipMemset(d, 0, 4096));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int chain = 200;
    for (int blocks : {1, 256 * 8}) {
        for (int iters : {200, 20000}) {
            auto run_stream = [&]() { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s, d, iters); };
            run_stream();
            CHK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            run_stream();
            CHK(hipStreamSynchronize(s));
            const double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / chain;
            hipGraph_t g;
            hipGraphExec_t ge;
            CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            run_stream();
            CHK(hipStreamEndCapture(s, &g));
            CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CHK(hipGraphLaunch(ge, s));
            CHK(hipStreamSynchronize(s));
            t0 = std::chrono::steady_clock::now();
            CHK(hipGraphLaunch(ge, s));
            CHK(hipStreamSynchronize(s));
            const double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / chain;
            // one kernel alone, for the net duration
            hipEvent_t a, b;
            CHK(hipEventCreate(&a));
            CHK(hipEventCreate(&b));
            CHK(hipEventRecord(a, s));
            hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s, d, iters);
            CHK(hipEventRecord(b, s));
            CHK(hipStreamSynchronize(s));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, a, b));
            std::printf("blocks %5d iters %6d: per link on a stream %.2f us, in a graph %.2f us (one kernel between events: %.2f us)\n", blocks, iters,
                        us_stream, us_graph, ms * 1e3);
            CHK(hipGraphExecDestroy(ge));
            CHK(hipGraphDestroy(g));
        }
    }
    return 0;
}
