// Micro-benchmark (not part of the product): where do workgroups land, and what does a barrier among a few workgroups cost
// when they share an XCD's L2 -- and when they do not?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_barrier tools/micro/xcd_barrier.hip && /tmp/xcd_barrier
// Each cluster of C workgroups runs R rounds of: every thread writes a word (plain store), s_waitcnt, one thread adds 1 to the
// cluster's counter (agent-scope atomic, relaxed) and spins until it reaches C * round, then every thread reads a word
// written by ANOTHER member with an L1-bypassing load (sc0) and checks it.  Clusters are formed either from workgroups with the
// same blockIdx % 8 (same XCD under round-robin dispatch) or from neighbours (8 different XCDs for C = 4: ids 4k .. 4k+3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// MODE: 0 = plain stores, loads sc1 (agent scope: past L1);  1 = stores sc1 (written through L2), loads sc1;
//       2 = stores sc0 sc1, loads sc0 sc1 (system scope)
template <bool SAME_XCD, int MODE>
__global__ __launch_bounds__(256) void k_barrier(int C, int R, unsigned* counters, unsigned* data, unsigned* xcc_out,
                                                 unsigned long long* cycles, unsigned* errors) {
    const int wg = blockIdx.x;
    int cluster, member;
    if (SAME_XCD) {   // members wg = x + 8 * (C * c + m)
        const int x = wg & 7, q = wg >> 3;
        cluster = (q / C) * 8 + x;
        member = q % C;
    } else {
        cluster = wg / C;
        member = wg % C;
    }
    if (threadIdx.x == 0) xcc_out[wg] = xcc_id();
    unsigned* cnt = counters + cluster;
    // two slots per member, alternating: a member that runs ahead into round r + 1 does not overwrite what a slower one still reads
    unsigned* mine0 = data + ((size_t)cluster * C + member) * 512;
    unsigned* other0 = data + ((size_t)cluster * C + (member + 1) % C) * 512;
    __shared__ unsigned go;
    const long long t0 = __builtin_readcyclecounter();
    unsigned bad = 0;
    for (int r = 1; r <= R; ++r) {
        unsigned* const mine = mine0 + (r & 1) * 256;
        unsigned* const other = other0 + (r & 1) * 256;
        const unsigned val = (unsigned)(r * 1000 + member);
        if (MODE == 0) mine[threadIdx.x] = val;
        else if (MODE == 1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(mine + threadIdx.x), "v"(val) : "memory");
        else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(mine + threadIdx.x), "v"(val) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(C * r)) {}
            go = r;
        }
        __syncthreads();
        unsigned v;
        if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(other + threadIdx.x) : "memory");
        else asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(other + threadIdx.x) : "memory");
        if (v != (unsigned)(r * 1000 + (member + 1) % C)) ++bad;
        __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[wg] = (unsigned long long)(t1 - t0);
    if (bad) atomicAdd(errors, bad);
}

int main() {
    const int C = 4, R = 200;
    for (int mode = 0; mode < 3; ++mode)
    for (int same = 1; same >= 0; --same) {
        for (int clusters : {8, 32}) {
            const int wgs = clusters * C;
            unsigned *counters, *data, *xcc, *errors;
            unsigned long long* cycles;
            hipMalloc(&counters, 4096); hipMemset(counters, 0, 4096);
            hipMalloc(&data, (size_t)wgs * 512 * 4); hipMemset(data, 0, (size_t)wgs * 512 * 4);
            hipMalloc(&xcc, wgs * 4); hipMalloc(&errors, 4); hipMemset(errors, 0, 4);
            hipMalloc(&cycles, wgs * 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
#define LAUNCH(S, M) hipLaunchKernelGGL((k_barrier<S, M>), dim3(wgs), dim3(256), 0, 0, C, R, counters, data, xcc, cycles, errors)
            if (same) { if (mode == 0) LAUNCH(true, 0); else if (mode == 1) LAUNCH(true, 1); else LAUNCH(true, 2); }
            else { if (mode == 0) LAUNCH(false, 0); else if (mode == 1) LAUNCH(false, 1); else LAUNCH(false, 2); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned> hx(wgs); std::vector<unsigned long long> hc(wgs); unsigned herr = 0;
            hipMemcpy(hx.data(), xcc, wgs * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hc.data(), cycles, wgs * 8, hipMemcpyDeviceToHost);
            hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
            int rr = 0; unsigned long long mx = 0;
            for (int i = 0; i < wgs; ++i) { rr += hx[i] == (unsigned)(i & 7); mx = hc[i] > mx ? hc[i] : mx; }
            printf("mode %d %s  clusters %2d x %d workgroups: %.2f us per barrier round (kernel %.1f us / %d rounds; %.0f cycles), workgroup i on XCD i %% 8: %d of %d, stale reads %u\n",
                   mode, same ? "same-XCD " : "cross-XCD", clusters, C, ms * 1000.0 / R, ms * 1000.0, R, (double)mx / R, rr, wgs, herr);
            hipFree(counters); hipFree(data); hipFree(xcc); hipFree(errors); hipFree(cycles);
        }
    }
    return 0;
}
