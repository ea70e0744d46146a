// Micro-benchmark: how fast ONE compute unit pulls L2-resident bytes into registers -- the floor of every kernel whose workgroups each
// stream the same weights (wz_k_mbconv_hp on the 10x10 maps: 72 workgroups x 1.2 - 1.8 MB of split weights per launch).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cu_stream.hip -o /tmp/cu_stream && /tmp/cu_stream
// W workgroups of 512 threads (eight waves, one workgroup per CU: 96 KiB of LDS asked for) each read the SAME `bytes` with 16-byte loads per lane,
// eight loads in flight per wave, and fold them into one word.  Printed: launch duration (HIP events, 20 launches back to back, empty bracket
// subtracted), bytes per workgroup / duration = GB/s per CU, and the sum over the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_stream(const u32x4* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
    extern __shared__ unsigned char lds[];
    u32x4 acc = {0u, 0u, 0u, 0u};
    // a wave walks its own eighth of the range (like a wave of the block kernel walks its own chunks), 1 KiB per instruction
    const size_t per_wave = n16 / 8;
    const u32x4* p = src + (threadIdx.x >> 6) * per_wave + (threadIdx.x & 63);
    for (size_t i = 0; i + 8 * 64 <= per_wave; i += 8 * 64) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[i + k * 64];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = lds[threadIdx.x];   // (never true: keeps the loads)
}

int main() {
    const size_t sizes[] = {640u << 10, 1280u << 10, 1843u << 10};
    const int wgs[] = {1, 36, 72, 144, 256};
    unsigned char* d = nullptr;
    unsigned* sink = nullptr;
    hipMalloc(&d, 4 << 20);
    hipMemset(d, 1, 4 << 20);
    hipMalloc(&sink, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("%10s %6s %12s %14s %14s\n", "bytes/wg", "wgs", "us/launch", "GB/s per CU", "TB/s chip");
    for (size_t bytes : sizes)
        for (int w : wgs) {
            const size_t n16 = bytes / 16;
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_stream, dim3(w), dim3(512), 96 * 1024, s, (const u32x4*)d, n16, sink);
            hipStreamSynchronize(s);
            hipEventRecord(e0, s);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_stream, dim3(w), dim3(512), 96 * 1024, s, (const u32x4*)d, n16, sink);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / 20.0;
            printf("%10zu %6d %12.2f %14.1f %14.2f\n", bytes, w, us, bytes / us * 1e-3, bytes * (double)w / us * 1e-6);
        }
    return 0;
}
