// Micro-probe: cost of a dependent kernel boundary on this box (eager vs hipGraph, 1..4 streams).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 0.f; }
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int CH = 100, REP = 200;
    float* d; hipMalloc(&d, 64 << 20);
    for (int ns = 1; ns <= 4; ns *= 2) {
        std::vector<hipStream_t> st(ns);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (int mode = 0; mode < 2; ++mode) {       // 0 trivial kernel, 1 256-WG touching 256 KB
            // eager
            for (int w = 0; w < 2; ++w) {
                double t0 = now();
                for (int r = 0; r < REP; ++r)
                    for (int s = 0; s < ns; ++s)
                        for (int i = 0; i < CH; ++i) {
                            if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[s], d);
                            else hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st[s], d + s * (1 << 20), 65536);
                        }
                for (auto& s : st) hipStreamSynchronize(s);
                double dt = now() - t0;
                if (w) printf("streams %d mode %d eager : %.3f us per kernel per stream, %.3f us aggregate\n", ns, mode,
                              dt / (REP * CH) * 1e6, dt / (REP * CH * ns) * 1e6);
            }
            // graph
            std::vector<hipGraphExec_t> ge(ns);
            for (int s = 0; s < ns; ++s) {
                hipGraph_t g;
                hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal);
                for (int i = 0; i < CH; ++i) {
                    if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[s], d);
                    else hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st[s], d + s * (1 << 20), 65536);
                }
                hipStreamEndCapture(st[s], &g);
                hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0);
                hipGraphDestroy(g);
            }
            for (int w = 0; w < 2; ++w) {
                double t0 = now();
                for (int r = 0; r < REP; ++r)
                    for (int s = 0; s < ns; ++s) hipGraphLaunch(ge[s], st[s]);
                for (auto& s : st) hipStreamSynchronize(s);
                double dt = now() - t0;
                if (w) printf("streams %d mode %d graph : %.3f us per kernel per stream, %.3f us aggregate\n", ns, mode,
                              dt / (REP * CH) * 1e6, dt / (REP * CH * ns) * 1e6);
            }
            for (auto& g : ge) hipGraphExecDestroy(g);
        }
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
