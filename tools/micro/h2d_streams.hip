// How fast do N page-locked host buffers of a frame's size reach HBM?  One hipMemcpyAsync each on ONE stream, the same spread over
// 2 / 4 / 8 streams (do copies on different streams run on different SDMA engines?), hipMemcpyBatchAsync, and a copy kernel
// reading the host memory through its device-mapped address.   hipcc --offload-arch=gfx950 -O3 -o /tmp/h2d tools/micro/h2d_streams.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u4;
__global__ void pull(const u4* const* src, u4* const* dst, size_t n16) {
    const u4* s = src[blockIdx.y];
    u4* d = dst[blockIdx.y];
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d[i] = __builtin_nontemporal_load(s + i);
}

int main() {
    const int NB = 32;
    for (size_t bytes : {921600ul, 6220800ul}) {
        std::vector<void*> h(NB), d(NB);
        for (int i = 0; i < NB; ++i) { CK(hipHostMalloc(&h[i], bytes, hipHostMallocDefault)); CK(hipMalloc(&d[i], bytes)); memset(h[i], i, bytes); }
        hipStream_t st[8];
        for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        auto run = [&](const char* name, auto fn) {
            fn(); hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 5; ++r) fn();
            hipDeviceSynchronize();
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%8zu B x %d  %-34s %6.1f GB/s  %6.1f us per buffer\n", bytes, NB, name, 5.0 * NB * bytes / s / 1e9, s / (5.0 * NB) * 1e6);
            return 0;
        };
        for (int ns : {1, 2, 4, 8}) {
            char nm[64]; snprintf(nm, sizeof nm, "hipMemcpyAsync on %d stream(s)", ns);
            run(nm, [&] { for (int i = 0; i < NB; ++i) (void)hipMemcpyAsync(d[i], h[i], bytes, hipMemcpyHostToDevice, st[i % ns]); });
        }
        {
            std::vector<size_t> sz(8, bytes);
            run("hipMemcpyBatchAsync x8, 1 stream", [&] {
                for (int i = 0; i < NB; i += 8) { size_t fail = 0; hipError_t e = hipMemcpyBatchAsync(&d[i], &h[i], sz.data(), 8, nullptr, nullptr, 0, &fail, st[0]); if (e != hipSuccess) { printf("batch: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); return; } }
            });
        }
        {
            void **dsrc, **ddst;
            std::vector<void*> hm(NB);
            for (int i = 0; i < NB; ++i) CK(hipHostGetDevicePointer(&hm[i], h[i], 0));
            CK(hipMalloc(&dsrc, NB * 8)); CK(hipMalloc(&ddst, NB * 8));
            CK(hipMemcpy(dsrc, hm.data(), NB * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(ddst, d.data(), NB * 8, hipMemcpyHostToDevice));
            for (int wg : {8, 32, 128}) {
                char nm[64]; snprintf(nm, sizeof nm, "copy kernel, %d WGs per buffer", wg);
                run(nm, [&] { for (int i = 0; i < NB; i += 8) hipLaunchKernelGGL(pull, dim3(wg, 8), dim3(256), 0, st[0], (const u4* const*)dsrc + i, (u4* const*)ddst + i, bytes / 16); });
            }
        }
        for (int i = 0; i < NB; ++i) { (void)hipHostFree(h[i]); (void)hipFree(d[i]); }
    }
    return 0;
}
