// Shared by k_conv.hip (fp16 engine) and k_f32.hip (fp32 engine).
#pragma once
#include "wz_common.h"

// --------------------------------------------------------------------------------------------
// Register-staged variant of the LDS-tiled implicit GEMM (same tile, same XCD-aware order, same LDS image of the
// activations).  Why: the `global_load_lds` path measured ~16 bytes per clock per CU no matter how it was driven (more
// buffers in flight, dedicated producer waves, full-line sources -- tools/conv_probe.py), i.e. ~2 000 cycles for the
// 32 KiB of a K step against 512 cycles of MFMA.  Here nothing goes through the DMA engine:
//   * waves are laid out 1 (pixels) x 4 (channels): a wave owns NW/2 channel tiles for all 128 pixels, so its weight
//     fragments are needed by no other wave -- they are loaded straight into VGPRs (the packed layout IS fragment order:
//     one coalesced 1 KiB load per fragment), two K steps ahead, and never touch LDS;
//   * the activation tile (shared by the four waves) is loaded in full 128-byte lines into VGPRs one step ahead and
//     written to LDS with `ds_write_b128` (lane-linear image, chunk swizzle applied on the source side) -- 16 KiB per
//     step instead of 32 KiB, through the ordinary vector-memory path.
// One `__syncthreads()` per K step, two 16 KiB LDS buffers.
// --------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int uint4_t;
#ifndef WZ_RS_STAMPS
#define WZ_RS_STAMPS 0
#endif
#define WZ_RS_TM 128   // pixels per workgroup tile

// F32: the -p 32 engine.  Same tile, same LDS image (a K step is still 128 bytes per pixel = 32 fp32 channels, a
// fragment still 1 KiB = one tile x 16 channels), `a.kc` / `a.kchunks` count 16-channel chunks, and each (tile, tile,
// chunk) is four `mfma_f32_16x16x4f32` consuming component j of the two float4 fragments (k = k0 + 4g + j, the packing of
// k_f32.hip).  At 32 cycles per MFMA a step is 4 096 cycles of matrix work: this variant IS MFMA-bound.
// EPI: static `apply(a, m, n4, v)` (final epilogue) and `partials(a)` (fp32 split-K workspace).

// SPEC: eight waves; 0..3 only compute (and fetch their own weight fragments), 4..7 only move the activation tile
// (global -> VGPR -> LDS).  A wave issues in order, so with four waves the ~1 000 cycles of loads, waits and LDS writes
// of a step sit in front of its ~750 cycles of fragment reads and MFMAs; split over two waves per SIMD they overlap.
// L = index of this workgroup within the convolution's own grid (blockIdx.x, or blockIdx.x minus the entry's first
// workgroup in a grouped launch, where entries start at multiples of 8 so that L & 7 is still the XCD).
template <int KS, int NW, bool SPEC, bool F32, class EPI>
__device__ __forceinline__ void wz_conv_rs_body(const WzConvArgs& a, unsigned char* smem, const int L) {
    constexpr int EB = F32 ? 4 : 2;   // bytes per element
    constexpr int taps = KS * KS;
    constexpr int NTW = NW / 2;   // 16-channel tiles per wave
    constexpr unsigned OOB = 0x7ffffff0u;   // buffer offset beyond every tensor: the load returns zeros
    const int n_tiles = a.n_pad >> 4;
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
    const int wave = wave8 & 3;
    const bool mover = !SPEC || wave8 >= 4, worker = !SPEC || wave8 < 4;
    const int r16 = lane & 15, g = lane >> 4;
    int bx, by, bz;
    {   // XCD-aware tile order, as in wz_k_conv_lds
        const int total = a.grid_m * a.grid_n * a.splitk;
        if (L >= total) return;   // padding workgroup of a grouped launch
        const int xcd = L & 7, slot = L >> 3;
        const int qd = total >> 3, rm = total & 7;
        int V = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
        if (a.order == 1) V = L;
        bx = V % a.grid_m;
        const int rest = V / a.grid_m;
        by = rest % a.grid_n;
        bz = rest / a.grid_n;
    }
    const int m_base = bx * WZ_RS_TM;
    const int nt_w = a.nt_base + by * (2 * NW) + wave * NTW;   // first channel tile of this wave

    // Addressing is the expensive part of an implicit GEMM step if done naively (a first version spent 900 of its
    // 2 400 cycles per step on 64-bit address arithmetic and bounds tests): both operands are read through buffer
    // descriptors with 32-bit offsets, everything that depends on the K step is wave-uniform (SGPR offset), and what
    // depends on the lane is computed once: the byte offset of the lane's pixel/chunk at tap (0, 0) and a 9-bit mask of
    // the taps that fall inside the frame.  An out-of-frame lane gets an out-of-range offset, for which the hardware
    // returns zeros -- no zero page, no branch.
    const int hw = a.hout * a.wout;
    const int n_frames = (a.M + hw - 1) / hw;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.in, 0, n_frames * a.hin * a.win * a.cin * EB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, n_tiles * taps * a.kc * 1024, 0x00020000);

    // activation staging: instruction i of this wave = pixels wave*32 + i*8 + (lane >> 3), 16-byte chunk (lane & 7)
    // stored at slot chunk ^ ((P >> 1) & 7) of the pixel's 128 bytes (lane-linear LDS write, swizzled source)
    int pix_off[4];
    unsigned tapmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m_base + wave * 32 + i * 8 + (lane >> 3);
        const bool mv = m < a.M;
        const int mm = mv ? m : 0;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / a.wout, ox = rem - oy * a.wout;
        const int iy0 = oy * a.stride - a.pad_t, ix0 = ox * a.stride - a.pad_l;
        const int chunk = (lane & 7) ^ ((i * 4 + (lane >> 4)) & 7);
        pix_off[i] = ((b * a.hin + iy0) * a.win + ix0) * a.cin * EB + chunk * 16;
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < taps; ++t) {
            const int iy = iy0 + t / KS, ix = ix0 + t % KS;
            if (mv && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win) mask |= 1u << t;
        }
        tapmask[i] = mask;
    }

    const int nsteps = a.kchunks >> 1;
    const int per = (nsteps + a.splitk - 1) / a.splitk;
    const int s0 = bz * per, s1 = min(s0 + per, nsteps);

    // K order: channel pair outermost, filter tap innermost (the nine taps of a channel pair read almost the same lines)
    auto load_b = [&](int s, uint4_t (&r)[4]) {
        const int t = (KS == 1) ? 0 : s % taps;
        const int c = (KS == 1) ? s * 2 : (s / taps) * 2;
        const int ky = (KS == 1) ? 0 : t / KS, kx = (KS == 1) ? 0 : t - ky * KS;
        const int tap_off = (ky * a.win + kx) * a.cin * EB;   // bytes, wave-uniform
        const int soff = c * 64;                              // a chunk (32 halves / 16 floats) = 64 bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned voff = ((tapmask[i] >> t) & 1u) ? (unsigned)(pix_off[i] + tap_off) : OOB;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, soff, 0);
        }
    };
    auto store_b = [&](int buf, const uint4_t (&r)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<uint4_t*>(smem + buf * 16384 + (wave * 4 + i) * 1024 + lane * 16) = r[i];
    };
    auto load_a = [&](int s, half8_t (&f)[NTW][2]) {
        const int t = (KS == 1) ? 0 : s % taps;
        const int c = (KS == 1) ? s * 2 : (s / taps) * 2;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            // fragment (tile, tap, chunk) = 1 KiB at ((tile * taps + tap) * kc + chunk) * 1 KiB; tiles past the end read zeros
            const unsigned soff = nt_w + nt < n_tiles ? (unsigned)(((nt_w + nt) * taps + t) * a.kc + c) * 1024u : OOB;
            const uint4_t lo = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, soff, 0);
            const uint4_t hi = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16 + 1024, soff, 0);
            f[nt][0] = __builtin_bit_cast(half8_t, lo);
            f[nt][1] = __builtin_bit_cast(half8_t, hi);
        }
    };

    float4_t acc[8][NTW];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf, const half8_t (&f)[NTW][2]) {
        const unsigned char* base = smem + buf * 16384;
        // all sixteen activation fragments of the step first, then the MFMAs back to back: left to itself the
        // scheduler recycles two fragment registers and exposes the LDS latency sixteen times per step
        half8_t fb[2][8];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)   // pixel P = mt*16 + r16, chunk kc*4 + g, slot swizzled by (P >> 1) & 7
                fb[kc][mt] = *reinterpret_cast<const half8_t*>(base + (mt * 16 + r16) * 128 + (((kc * 4 + g) ^ ((r16 >> 1) & 7)) * 16));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    if constexpr (F32) {
                        const float4_t wf = __builtin_bit_cast(float4_t, f[nt][kc]), xf = __builtin_bit_cast(float4_t, fb[kc][mt]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], xf[j], acc[mt][nt], 0, 0, 0);
                    } else {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[nt][kc], fb[kc][mt], acc[mt][nt], 0, 0, 0);
                    }
                }
        __builtin_amdgcn_sched_barrier(0);
    };

    // Two K steps per trip so that the two weight-fragment sets and the two LDS buffers have static names; loads past
    // the last step are clamped to it (a redundant reload instead of a branch), an odd last step runs after the loop.
    // Half-step (parity p): LDS buffer p and fragment set p hold its step, `rb` holds the activations of the next step
    // (loaded one half-step ago), fragment set p ^ 1 the next weights (loaded one and a half half-steps ago).
    uint4_t rb[4];
    half8_t fa0[NTW][2], fa1[NTW][2];
    const int last = s1 - 1, pairs = (s1 - s0) >> 1;
    if (s0 < s1) {
        if (mover) {
            load_b(s0, rb);
            store_b(0, rb);
            load_b(min(s0 + 1, last), rb);
        }
        if (worker) {
            load_a(s0, fa0);
            load_a(min(s0 + 1, last), fa1);
        }
        __syncthreads();
        int s = s0;
        // phase cycle counts for tools/rs_probe.py: compiled in only with -DWZ_RS_STAMPS=1 (they cost registers)
        const bool stamp = WZ_RS_STAMPS && a.dbg && (threadIdx.x & 255) == 0 && L == 0;
        long long cy[5] = {0, 0, 0, 0, 0};
        for (int p = 0; p < pairs; ++p, s += 2) {
            const long long c0 = stamp ? clock64() : 0;
            long long c1 = c0, c2 = c0, c3 = c0;
            if (mover) {
                store_b(1, rb);   // buffer 1 was last read before the previous barrier
                c1 = stamp ? clock64() : 0;
                load_b(min(s + 2, last), rb);
                c2 = c3 = stamp ? clock64() : 0;
            }
            if (worker) {
                if (SPEC) c1 = c2 = stamp ? clock64() : 0;
                compute(0, fa0);
                c3 = stamp ? clock64() : 0;
                load_a(min(s + 2, last), fa0);
            }
            const long long c4 = stamp ? clock64() : 0;
            __syncthreads();
            if (stamp) {
                const long long c5 = clock64();
                cy[0] += c1 - c0; cy[1] += c2 - c1; cy[2] += c3 - c2; cy[3] += c4 - c3; cy[4] += c5 - c4;
            }
            if (mover) {
                store_b(0, rb);
                load_b(min(s + 3, last), rb);
            }
            if (worker) {
                compute(1, fa1);
                load_a(min(s + 3, last), fa1);
            }
            __syncthreads();
        }
        if (worker && ((s1 - s0) & 1)) compute(0, fa0);
        if (stamp && pairs > 0) {
            unsigned long long* const d = a.dbg + (threadIdx.x == 0 ? 0 : 8);   // slots 8.. = a mover wave
#pragma unroll
            for (int i = 0; i < 5; ++i) d[i] = (unsigned long long)(cy[i] / pairs);
            d[5] = (unsigned long long)pairs;
        }
    }
    if (!worker) return;

    if constexpr (!SPEC && !F32) {
        if (a.splitk > 1 && a.inline_reduce) {
            // in-launch reduction (WzConvArgs::inline_reduce): every K slice publishes its partial tile write-through and
            // takes a ticket on the tile's counter; the workgroup that finds the other slices already there sums them
            // all in slice order (wz_k_splitk_reduce's arithmetic: bit-identical) and finishes the head's outputs
            float* const ws = a.ws;   // (a.out is where the FINISHED columns go)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int m = m_base + mt * 16 + r16;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int n4 = (nt_w + nt) * 16 + g * 4;
                    if (m < a.M && n4 < a.n_pad) EPI::publish(ws + ((size_t)bz * a.M + m) * a.n_pad + n4, acc[mt][nt]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                       // every wave's stores have landed (and nobody reads the LDS tiles any more)
            int* const flag = reinterpret_cast<int*>(smem);
            if (threadIdx.x == 0) {
                int32_t* const tk = a.tickets + (by * a.grid_m + bx);
                const int t = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == a.splitk - 1) {
                    __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                        // this CU reads the slabs fresh
                }
                *flag = t;
            }
            __syncthreads();
            if (*flag != a.splitk - 1) return;
            // slice by slice, all of a slice's fragments requested before the first is added (one memory latency per
            // slice, not per fragment); the sum order per element is slice 0, 1, 2 ... as in wz_k_splitk_reduce
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < a.splitk; ++z) {
                float4_t pz[8][NTW];
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const int m = m_base + mt * 16 + r16;
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const int n4 = (nt_w + nt) * 16 + g * 4;
                        pz[mt][nt] = (m < a.M && n4 < a.n_pad)
                                         ? *reinterpret_cast<const float4_t*>(ws + ((size_t)z * a.M + m) * a.n_pad + n4)
                                         : (float4_t){0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[mt][nt][r] += pz[mt][nt][r];
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) EPI::finish(a, m_base + mt * 16 + r16, (nt_w + nt) * 16 + g * 4, acc[mt][nt]);
            return;
        }
    }

#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = m_base + mt * 16 + r16;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n4 = (nt_w + nt) * 16 + g * 4;
            if (a.splitk > 1) {
                if (m < a.M && n4 < a.n_pad)
                    *reinterpret_cast<float4_t*>(EPI::partials(a) + ((size_t)bz * a.M + m) * a.n_pad + n4) = acc[mt][nt];
            } else {
                EPI::apply(a, m, n4, acc[mt][nt]);
            }
        }
    }
}

