// What does a dependent kernel boundary cost on THIS stack, and which ingredient of the engine's environment makes it cost more?
// (the engine measures 2.7 - 3.0 us between dependent kernels of one stream; /opt/skills/guides/MI355X_MICROARCH.md's
// `boundary` row says 1.45 us between trivial kernels, 1.7 - 1.9 between streaming ones, + dirty bytes / 6 TB/s.)
//
// A chain of N dependent launches on one stream; every workgroup stamps the constant 100 MHz clock (s_memrealtime: one clock for all
// XCDs) when it enters and when it leaves; the GAP of a boundary = earliest entry of launch k + 1 - latest exit of launch k.
// Variants, one ingredient at a time:
//   grid / block shape, bytes the predecessor leaves dirty, a page-locked + mapped host range in the process, the LAST kernel of the
//   chain writing into mapped host memory (the engine's rows), a kernel-argument block of 1.3 KB (the grouped launches pass whole
//   descriptor tables by value), 150 KB of dynamic LDS (the whole-CU workgroups), eager launches vs a captured hipGraph.
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/boundary.hip -o /tmp/boundary && /tmp/boundary
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Big { unsigned char pad[1280]; };

// stamps[launch][0 .. 255] entry per workgroup bucket (min is taken on the host), [256 .. 511] exit
__global__ void k_chain(unsigned long long* stamps, int launch, float* dirty, size_t dirty_floats, float* host_out, int work_iters) {
    const unsigned long long t0 = wall_clock64();
    extern __shared__ float lds[];
    const unsigned id = blockIdx.x;
    float acc = 0.f;
    // optional streaming work: every thread writes its share of `dirty`
    const size_t n = dirty_floats, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)id * blockDim.x + threadIdx.x; i < n; i += stride) dirty[i] = (float)(i + launch);
    for (int i = 0; i < work_iters; ++i) acc = acc * 1.0001f + (float)i;
    if (host_out && id == 0 && threadIdx.x < 64) host_out[threadIdx.x] = acc + (float)launch;
    if (work_iters < 0) lds[threadIdx.x] = acc;   // (keeps the LDS allocation alive)
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* s = stamps + (size_t)launch * 512;
        if (id < 256) s[id] = t0;
        s[256 + (id & 255)] = wall_clock64();
    }
}

__global__ void k_chain_big(Big b, unsigned long long* stamps, int launch) {
    const unsigned long long t0 = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* s = stamps + (size_t)launch * 512;
        if (blockIdx.x < 256) s[blockIdx.x] = t0 + (b.pad[launch & 1023] & 0);
        s[256 + (blockIdx.x & 255)] = wall_clock64();
    }
}

// sixteen DIFFERENT kernels of ~20 KB of straight-line code each, launched round-robin: every boundary lands on code that is not in the
// instruction caches (the engine's 30 launches are 30 different kernels; the chains above re-run one kernel)
template <int V>
__global__ void k_distinct(unsigned long long* stamps, int launch, const float* src, float* dst) {
    float x = src[threadIdx.x & 63];                    // (an argument-dependent load in front of the stamp, like a real kernel's first instruction)
    const unsigned long long t0 = wall_clock64();
#pragma unroll
    for (int i = 0; i < 1200; ++i) x = x * (1.0f + 1e-7f * (float)(i * 17 + V)) + (float)(i ^ V) * 1e-9f;
    if (x == 123.456f) dst[threadIdx.x] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* s = stamps + (size_t)launch * 512;
        if (blockIdx.x < 256) s[blockIdx.x] = t0;
        s[256 + (blockIdx.x & 255)] = wall_clock64();
    }
}
typedef void (*distinct_fn)(unsigned long long*, int, const float*, float*);
template <int... V> static void fill_distinct(distinct_fn* t, std::integer_sequence<int, V...>) { int i = 0; ((t[i++] = k_distinct<V>), ...); }

struct Cfg {
    const char* name;
    int grid, block;
    size_t lds;
    size_t dirty_bytes;
    bool host_tail;    // the chain's last kernel writes into mapped host memory
    bool big_args;
    bool graph;
    int work;
    int distinct;      // > 0: that many different kernels round-robin (k_distinct)
};

static double run(const Cfg& c, hipStream_t s, unsigned long long* d_stamps, unsigned long long* h_stamps, float* d_dirty, float* h_mapped_dev, int N, int reps,
                  double* p_exit_to_exit) {
    std::vector<double> gaps, e2e;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    auto enqueue = [&]() {
        static distinct_fn table[16];
        static bool filled = false;
        if (!filled) { fill_distinct(table, std::make_integer_sequence<int, 16>()); filled = true; }
        for (int k = 0; k < N; ++k) {
            if (c.distinct > 0) {
                hipLaunchKernelGGL(table[k % c.distinct], dim3(c.grid), dim3(c.block), 0, s, d_stamps, k, (const float*)d_dirty, d_dirty + 1024);
            } else if (c.big_args) {
                Big b;
                memset(&b, 0, sizeof(b));
                hipLaunchKernelGGL(k_chain_big, dim3(c.grid), dim3(c.block), 0, s, b, d_stamps, k);
            } else {
                float* ho = (c.host_tail && k == N - 1) ? h_mapped_dev : nullptr;
                hipLaunchKernelGGL(k_chain, dim3(c.grid), dim3(c.block), c.lds, s, d_stamps, k, d_dirty, c.dirty_bytes / 4, ho, c.work);
            }
        }
    };
    if (c.graph) {
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue();
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    }
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipMemsetAsync(d_stamps, 0, (size_t)N * 512 * 8, s));
        CK(hipStreamSynchronize(s));
        if (c.graph) CK(hipGraphLaunch(exec, s)); else enqueue();
        CK(hipStreamSynchronize(s));
        if (r < 2) continue;
        CK(hipMemcpy(h_stamps, d_stamps, (size_t)N * 512 * 8, hipMemcpyDeviceToHost));
        unsigned long long prev_exit = 0;
        for (int k = 0; k < N; ++k) {
            unsigned long long en = ~0ull, ex = 0;
            for (int i = 0; i < 256; ++i) {
                const unsigned long long a = h_stamps[(size_t)k * 512 + i], b = h_stamps[(size_t)k * 512 + 256 + i];
                if (a && a < en) en = a;
                if (b > ex) ex = b;
            }
            if (k > 1 && k < N - 1) {   // (the first boundary follows the memset / the host's launch, the last may be the host-writing kernel)
                gaps.push_back((double)(en - prev_exit) * 0.01);
                e2e.push_back((double)(ex - prev_exit) * 0.01);
            }
            prev_exit = ex;
        }
    }
    if (exec) CK(hipGraphExecDestroy(exec));
    if (graph) CK(hipGraphDestroy(graph));
    std::sort(gaps.begin(), gaps.end());
    std::sort(e2e.begin(), e2e.end());
    *p_exit_to_exit = e2e[e2e.size() / 2];
    return gaps[gaps.size() / 2];
}

int main(int argc, char** argv) {
    const int N = 64, reps = 20;
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *d_stamps, *h_stamps;
    CK(hipMalloc((void**)&d_stamps, (size_t)N * 512 * 8));
    h_stamps = (unsigned long long*)malloc((size_t)N * 512 * 8);
    float* d_dirty;
    CK(hipMalloc((void**)&d_dirty, 64u << 20));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    const Cfg base[] = {
        {"trivial, 256 WGs x 256 thr, eager", 256, 256, 0, 0, false, false, false, 0},
        {"trivial, 256 WGs x 256 thr, hipGraph", 256, 256, 0, 0, false, false, true, 0},
        {"trivial, 72 WGs x 512 thr + 150 KB LDS, hipGraph", 72, 512, 150 * 1024, 0, false, false, true, -1},
        {"trivial, 2816 WGs x 256 thr, hipGraph", 2816, 256, 0, 0, false, false, true, 0},
        {"trivial, 8 WGs x 1024 thr, hipGraph", 8, 1024, 0, 0, false, false, true, 0},
        {"1.3 KB of kernel arguments, 256 WGs, hipGraph", 256, 256, 0, 0, false, true, true, 0},
        {"predecessor leaves 1 MB dirty, hipGraph", 256, 256, 0, 1u << 20, false, false, true, 0},
        {"predecessor leaves 4 MB dirty, hipGraph", 256, 256, 0, 4u << 20, false, false, true, 0},
        {"predecessor leaves 16 MB dirty, hipGraph", 1024, 256, 0, 16u << 20, false, false, true, 0},
        {"~9 us of ALU work per launch, hipGraph", 256, 256, 0, 0, false, false, true, 500},
        {"~9 us of ALU work per launch, eager (the host runs ahead)", 256, 256, 0, 0, false, false, false, 500},
        {"~9 us of work + 4 MB dirty per launch, eager", 256, 256, 0, 4u << 20, false, false, false, 500},
        {"~9 us of work + 4 MB dirty per launch, hipGraph", 256, 256, 0, 4u << 20, false, false, true, 500},
        {"last kernel of the chain writes mapped host memory, hipGraph", 256, 256, 0, 0, true, false, true, 0},
        {"ONE kernel of 20 KB of code + a leading argument load, hipGraph", 256, 256, 0, 0, false, false, true, 0, 1},
        {"16 DIFFERENT kernels of 20 KB of code round-robin, hipGraph", 256, 256, 0, 0, false, false, true, 0, 16},
    };
    printf("# tools/micro/boundary.hip: gap between dependent launches of one stream (earliest entry of k + 1 - latest exit of k, median over %d boundaries x %d runs),\n"
           "# and exit-to-exit (gap + the kernel itself), us, 100 MHz device clock\n", N - 3, reps);
    for (int phase = 0; phase < 3; ++phase) {
        float* h_mapped = nullptr;
        float* h_mapped_dev = nullptr;
        void* reg = nullptr;
        if (phase == 0) printf("## plain process: no page-locked host memory\n");
        if (phase == 1) {
            printf("## + 64 MB page-locked by hipHostRegister(mapped) and 1 MB of hipHostMalloc(mapped) alive in the process (the engine's frame arenas and row blocks)\n");
            reg = aligned_alloc(4096, 64u << 20);
            memset(reg, 1, 64u << 20);
            CK(hipHostRegister(reg, 64u << 20, hipHostRegisterMapped));
            CK(hipHostMalloc((void**)&h_mapped, 1u << 20, hipHostMallocMapped));
            CK(hipHostGetDevicePointer((void**)&h_mapped_dev, h_mapped, 0));
        }
        if (phase == 2) {
            printf("## + GPU_MAX_HW_QUEUES=8 style load: three more streams alive and idle\n");
            hipStream_t extra[3];
            for (auto& x : extra) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
            reg = aligned_alloc(4096, 64u << 20);
            memset(reg, 1, 64u << 20);
            CK(hipHostRegister(reg, 64u << 20, hipHostRegisterMapped));
            CK(hipHostMalloc((void**)&h_mapped, 1u << 20, hipHostMallocMapped));
            CK(hipHostGetDevicePointer((void**)&h_mapped_dev, h_mapped, 0));
        }
        for (const Cfg& c : base) {
            if (c.host_tail && !h_mapped_dev) continue;
            double e2e = 0;
            const double gap = run(c, s, d_stamps, h_stamps, d_dirty, h_mapped_dev, N, reps, &e2e);
            printf("%-66s gap %5.2f us   exit-to-exit %6.2f us\n", c.name, gap, e2e);
        }
        if (reg) { CK(hipHostUnregister(reg)); free(reg); }
        if (h_mapped) CK(hipHostFree(h_mapped));
    }
    return 0;
}
