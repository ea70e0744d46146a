// Inverted-residual block on the SMALL maps (19x19, 10x10): the expanded channels are split over the
// WAVEFRONTS of a workgroup and summed through LDS -- no partial sums in HBM, no reduce launch.
//
// On these maps there are few pixels (2 888 / 800 at batch 8) and many expanded channels (384 ... 960), so the
// parallelism has to come from the channels.  k_mbconv.hip spreads channel groups over workgroups and needs a
// second launch to add their fp32 partials; here a workgroup is 8 waves on ONE 4x4 pixel tile:
//
//   * every wave loads the tile's halo (6x6 pixels at stride 1, 9x9 at stride 2) as MFMA B fragments (the
//     same 36 / 81 pixels for all waves: L1 hits) and walks ITS 32-channel chunks (wave, wave+8, ...): expand
//     MFMA -> its own few KiB of LDS -> depthwise 3x3 in fp32 -> project MFMA into its own fp32 accumulators,
//     with no workgroup barrier in the loop (the structure of k_mbconv_wave.hip);
//   * at the end the 8 accumulator sets meet in LDS and are added in the fixed order wave 0 .. 7 (deterministic),
//     + bias, + residual, fp16 store.
//
// The sum order differs from the per-layer kernels' (chunk 0, 1, 2, ... in one accumulator), so outputs agree
// with them to fp32 rounding of the sum, i.e. an fp16 ulp here and there (same as k_mbconv.hip's channel groups).
#include "wz_common.h"


// CS_WAVES: waves per workgroup / tile (8; 4: half a CU per workgroup, a longer chunk walk)
template <bool EXPAND, int MPW, int MQW, int KCI, int NTO, int CS_WAVES = 8>
__global__ __launch_bounds__(CS_WAVES * 64) void wz_k_mbconv_cs(const WzMbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wz_cs_smem[];
    WZ_LANE_STAMP(a.dbg);
    constexpr int CE = 32, ES = CE + 8;
    constexpr int EBYTES = EXPAND ? MPW * 16 * ES * 2 : 0;
    constexpr int NTC = NTO > 10 ? 10 : NTO;                        // output tiles reduced per round
    constexpr int RED_BYTES = CS_WAVES * MQW * NTC * 1024;
    constexpr int REGION = (CS_WAVES * EBYTES > RED_BYTES) ? CS_WAVES * EBYTES : RED_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    half_t* const E = reinterpret_cast<half_t*>(wz_cs_smem + wave * EBYTES);
    float* const red = reinterpret_cast<float*>(wz_cs_smem);                    // overlays the chunk buffers after the loop
    half_t* const wd_l = reinterpret_cast<half_t*>(wz_cs_smem + REGION);         // [9][cmid_pad]
    float* const bd_l = reinterpret_cast<float*>(wd_l + 9 * a.cmid_pad);         // [cmid_pad]
    float* const be_l = bd_l + a.cmid_pad;                                       // [cmid_pad] expand bias

    {   // staged once: depthwise weights, depthwise bias, expand bias
        const int c8s = a.cmid_pad >> 3;
        for (int i = threadIdx.x; i < 9 * c8s; i += CS_WAVES * 64)
            *reinterpret_cast<half8_t*>(wd_l + i * 8) = *reinterpret_cast<const half8_t*>(a.wd + (size_t)i * 8);
        for (int i = threadIdx.x; i < (a.cmid_pad >> 2); i += CS_WAVES * 64) {
            *reinterpret_cast<float4_t*>(bd_l + i * 4) = *reinterpret_cast<const float4_t*>(a.bd + i * 4);
            if (EXPAND)
                *reinterpret_cast<float4_t*>(be_l + i * 4) = (i * 4 < a.nmid_pad) ? *reinterpret_cast<const float4_t*>(a.be + i * 4)
                                                                                  : (float4_t){0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- the workgroup's tile
    // (a.nsplit > 1: that many workgroups per tile, each with NTO of the block's output tiles -- block 16; the expand and depthwise
    //  stages are repeated per group, which costs nothing on a launch of 72 tiles and spares the fp32 partial sums in HBM)
    const int tiles = a.tiles_x * a.tiles_y;
    const int bid = (int)blockIdx.x / a.nsplit, nt0 = ((int)blockIdx.x % a.nsplit) * NTO;
    const int b = bid / tiles, t = bid - b * tiles;
    const int tyi = t / a.tiles_x;
    const int oy0 = tyi * a.th, ox0 = (t - tyi * a.tiles_x) * a.tw;
    const int s = a.stride;
    const int hw_ = (a.tw - 1) * s + 3, hh_ = (a.th - 1) * s + 3;
    const int P = hh_ * hw_, Q = a.th * a.tw;
    const int iy_base = oy0 * s - a.pad_t, ix_base = ox0 * s - a.pad_l;
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    int hp0[MQW], opix[MQW];
#pragma unroll
    for (int j = 0; j < MQW; ++j) {
        const int q = j * 16 + r16;
        const int qc = q < Q ? q : Q - 1;
        const int qy = qc / a.tw, qx = qc - qy * a.tw;
        hp0[j] = qy * s * hw_ + qx * s;
        const int oy = oy0 + qy, ox = ox0 + qx;
        opix[j] = (q < Q && oy < a.hout && ox < a.wout) ? (b * a.hout + oy) * a.wout + ox : -1;
    }

    half8_t xf[EXPAND ? MPW : 1][EXPAND ? KCI : 1];
    bool inimg[EXPAND ? MPW : 1];
    // a.out2: the block also STORES its expanded tensor (block 13: the first SSD feature map) -- every input pixel by the tile that owns
    // it: the first th * s rows / tw * s columns of a tile's halo are its own, the rest is the next tile's (tiles step by th * s)
    int own2[EXPAND ? MPW : 1];   // pixel offset into out2, or -1
    if constexpr (EXPAND) {
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
            const int p = i * 16 + r16;
            const int hy = p / hw_, hx = p - hy * hw_;
            const int iy = iy_base + hy, ix = ix_base + hx;
            const bool ok = p < P && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
            inimg[i] = ok;
            own2[i] = (a.out2 && ok && nt0 == 0 && hy < a.th * s && hx < a.tw * s) ? (b * a.hin + iy) * a.win + ix : -1;
            const half_t* src = a.in + ((size_t)(b * a.hin + (ok ? iy : 0)) * a.win + (ok ? ix : 0)) * a.cin0;
#pragma unroll
            for (int c = 0; c < KCI; ++c) {
                const int k0 = c * 32 + g * 8;
                xf[i][c] = (ok && k0 < a.cin0) ? *reinterpret_cast<const half8_t*>(src + k0) : zero8;
            }
        }
    }

    float4_t acc[MQW][NTO];
#pragma unroll
    for (int j = 0; j < MQW; ++j)
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) acc[j][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int nk32 = a.cmid_pad >> 5;
    const int ntiles_e = EXPAND ? (a.nmid_pad >> 4) : 1;
    half8_t wa[2][EXPAND ? KCI : 1];
    auto load_wa = [&](int ps) {
        if constexpr (EXPAND) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int tn = min(ps * 2 + nt, ntiles_e - 1);
                const half_t* wsrc = a.we + ((size_t)tn * a.kc0 * 64 + lane) * 8;
#pragma unroll
                for (int c = 0; c < KCI; ++c) wa[nt][c] = *reinterpret_cast<const half8_t*>(wsrc + (size_t)c * 512);
            }
        }
    };
    if (wave < nk32) load_wa(wave);
    __syncthreads();   // staged depthwise weights / biases visible

    for (int ps = wave; ps < nk32; ps += CS_WAVES) {
        const int ce0 = ps * CE;
        if constexpr (EXPAND) {
            half8_t wa_c[2][KCI];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int c = 0; c < KCI; ++c) wa_c[nt][c] = wa[nt][c];
            if (ps + CS_WAVES < nk32) load_wa(ps + CS_WAVES);   // in flight under this pass
            // ---- expand: E[p][ce] = in-frame ? relu6(sum_k X[p][k] We[k][ce] + be[ce]) : 0
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bool have = ce0 + nt * 16 < a.nmid_pad;
                const float4_t bv = *reinterpret_cast<const float4_t*>(be_l + ce0 + nt * 16 + g * 4);
#pragma unroll
                for (int i = 0; i < MPW; ++i) {
                    float4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < KCI; ++c) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa_c[nt][c], xf[i][c], d, 0, 0, 0);
                    const half4_t o = wz_relu6_pack(d, bv, inimg[i] && have);
                    *reinterpret_cast<half4_t*>(E + (i * 16 + r16) * ES + nt * 16 + g * 4) = o;
                    if (own2[i] >= 0 && ce0 + nt * 16 + g * 4 < a.cmid)
                        *reinterpret_cast<half4_t*>(a.out2 + (size_t)own2[i] * a.cmid + ce0 + nt * 16 + g * 4) = o;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- depthwise (lane = output pixel x 8 channels) feeding the project MFMAs
        {
            const int coff = ce0 + g * 8;
            half8_t wt[9];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) wt[tp] = *reinterpret_cast<const half8_t*>(wd_l + tp * a.cmid_pad + coff);
            const float4_t b0 = *reinterpret_cast<const float4_t*>(bd_l + coff);
            const float4_t b1 = *reinterpret_cast<const float4_t*>(bd_l + coff + 4);
#pragma unroll
            for (int j = 0; j < MQW; ++j) {
                float d[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { d[r] = b0[r]; d[4 + r] = b1[r]; }
                if constexpr (EXPAND) {
                    const half_t* ep = E + hp0[j] * ES + g * 8;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const half8_t x = *reinterpret_cast<const half8_t*>(ep + (ky * hw_ + kx) * ES);
#pragma unroll
                            for (int r = 0; r < 8; ++r) d[r] = fmaf((float)x[r], (float)wt[ky * 3 + kx][r], d[r]);
                        }
                } else {   // no expand stage: the taps come from global memory (zero outside the frame)
                    const int hy0 = hp0[j] / hw_, hx0 = hp0[j] - hy0 * hw_;
                    const int cload = min(coff, a.cmid - 8);
                    const bool cok = coff < a.cmid;
                    half8_t x[9];
                    bool okx[9];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int iy = iy_base + hy0 + ky, ix = ix_base + hx0 + kx;
                            okx[ky * 3 + kx] = cok && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
                            const int cy = min(max(iy, 0), a.hin - 1), cx = min(max(ix, 0), a.win - 1);
                            x[ky * 3 + kx] = *reinterpret_cast<const half8_t*>(
                                a.in + ((size_t)(b * a.hin + cy) * a.win + cx) * a.cmid + cload);
                        }
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const half8_t xv = okx[tp] ? x[tp] : zero8;
#pragma unroll
                        for (int r = 0; r < 8; ++r) d[r] = fmaf((float)xv[r], (float)wt[tp][r], d[r]);
                    }
                }
                half8_t bf;
#pragma unroll
                for (int r = 0; r < 8; ++r) bf[r] = (half_t)fminf(fmaxf(d[r], 0.0f), 6.0f);
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    const half8_t wp = *reinterpret_cast<const half8_t*>(a.wp + ((size_t)((nt0 + nt) * a.kc + ps) * 64 + lane) * 8);
                    acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp, bf, acc[j][nt], 0, 0, 0);
                }
            }
        }
        if constexpr (EXPAND) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // (the next pass's E stores stay behind these reads)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }

    // ---- the 8 waves' accumulators meet in LDS (over the chunk buffers), NTC output tiles per round;
    //      (j, nt) of a round is summed by wave (j*NTC + nt) % 8 in the order wave 0 .. 7
    for (int n0 = 0; n0 < NTO; n0 += NTC) {
        __syncthreads();   // chunk buffers (first round) / previous round's slabs are no longer read
#pragma unroll
        for (int j = 0; j < MQW; ++j)
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt)
                if (nt >= n0 && nt < n0 + NTC)
                    *reinterpret_cast<float4_t*>(red + ((size_t)((wave * MQW + j) * NTC + (nt - n0)) * 64 + lane) * 4) = acc[j][nt];
        __syncthreads();
        for (int pr = wave; pr < MQW * NTC; pr += CS_WAVES) {
            const int j = pr / NTC, ntl = pr - j * NTC;
            const int nt = n0 + ntl;
            if (nt >= NTO) continue;
            float4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < CS_WAVES; ++w) {
                const float4_t pz = *reinterpret_cast<const float4_t*>(red + ((size_t)((w * MQW + j) * NTC + ntl) * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += pz[r];
            }
            // epilogue for (pixel of slot j, channels nt*16 + g*4 ..)
            int op = -1;
#pragma unroll
            for (int jj = 0; jj < MQW; ++jj)
                if (jj == j) op = opix[jj];
            const int n4 = (nt0 + nt) * 16 + g * 4;
            if (op < 0 || n4 >= a.cout) continue;
            const float4_t bv = *reinterpret_cast<const float4_t*>(a.bp + n4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
            const size_t o = (size_t)op * a.cout + n4;
            if (a.res) {
                const half4_t rv = *reinterpret_cast<const half4_t*>(a.res + o);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
            }
            const half4_t hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4_t*>(a.out + o) = hv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
static int wz_cs_env(const char* name, int dflt) {
    const char* e = wz_dev_getenv(name);
    return (e && atoi(e) > 0) ? atoi(e) : dflt;
}

template <bool EXPAND, int MPW, int MQW, int KCI, int NTO, int CS_WAVES = 8>
static int wz_cs_launch(WzMbArgs a, int n, hipStream_t s, bool prepare) {
    constexpr int EB = EXPAND ? MPW * 16 * 40 * 2 : 0;
    constexpr int NTC = NTO > 10 ? 10 : NTO;
    constexpr int RED = CS_WAVES * MQW * NTC * 1024;
    const size_t region = (size_t)(CS_WAVES * EB > RED ? CS_WAVES * EB : RED);
    const size_t lds = region + (size_t)a.cmid_pad * (9 * 2 + 2 * 4);
    auto k = wz_k_mbconv_cs<EXPAND, MPW, MQW, KCI, NTO, CS_WAVES>;
    if (prepare) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return lds <= 160 * 1024 ? 0 : -1;
    }
    WZ_LAUNCH(k, dim3(a.tiles_x * a.tiles_y * n * a.nsplit), dim3(CS_WAVES * 64), lds, s, a);
    return 1;
}

// Serves the blocks with 19x19 outputs (blocks 6 .. 12); the 10x10 ones on request (WZ_MB_CS_MIN_W).  -2: does not apply (caller falls back to wz_launch_mbconv).
int wz_launch_mbconv_cs(const WzMbArgs& a0, int n, hipStream_t s, bool prepare) {
    static const int enabled = wz_cs_env("WZ_MB_CS", 1);
    // On the 10x10 maps (blocks 13 .. 16: 9 tiles per frame, 18 - 30 chunks) this kernel takes longer alone than channel groups
    // over 256 workgroups + a reduce launch (block 16: 31 us against 10 + 4) -- every workgroup streams all of the block's
    // 0.6 - 0.9 MB of weights -- but it occupies 72 CUs instead of all of them, writes no fp32 partial sums (80 MB per batch
    // out and back) and needs no reduce launches (31 graph nodes instead of 35): 49.1 k -> 50.1 k frames/s with four lanes in flight,
    // p50 0.380 -> 0.394 ms (profiles/r03_wave_counts_*; round 1 measured the same trade at +1 %, round 2 at +0.6 %).  Default since
    // round 3; WZ_MB_CS_MIN_W=11 brings the channel-group kernel back for the 10x10 maps.
    static const int min_w = wz_cs_env("WZ_MB_CS_MIN_W", wz_latency_schedule() ? 11 : 1);
    if (a0.stem || a0.wout > 19 || ((enabled != 1 || a0.wout < min_w) && !a0.has_out2)) return -2;   // (a block with a second output runs here or nowhere)
    const int nto = a0.n_pad / 16;
    // ... except block 16 (320 output channels, 0.9 MB of weights per workgroup: 31 us alone against 9 + 4): it stays on the channel-group
    // kernel -- 50.0 k frames/s either way, p50 0.380 instead of 0.395 ms (WZ_MB_CS_MAX_NTO=20: on this kernel as well)
    static const int max_nto = wz_cs_env("WZ_MB_CS_MAX_NTO", 10);   // blocks with more output tiles than this stay on the channel-group kernel
    // ... until late round 3: block 16 runs on this kernel as TWO workgroups per tile with 10 output tiles each (WZ_MB_CS_SPLIT16, default 1;
    // the expand and depthwise stages are done twice -- on 72 tiles that costs nothing): 15.8 + 2.8 us and a kernel boundary -> 12.7 us,
    // no fp32 partial sums of any block in HBM any more, 31 graph nodes; 51.8 k -> 52.8 k frames/s, p50 0.383 -> 0.376 ms
    // (profiles/r03_four_waves_per_simd.txt (e)).  0: the channel-group kernel + reduce as before.
    static const int split16 = [] {
        const char* e = wz_dev_getenv("WZ_MB_CS_SPLIT16");   // (0 is a value here, unlike with wz_cs_env)
        return (e && e[0]) ? atoi(e) : (wz_latency_schedule() ? 0 : 1);
    }();
    if (!prepare && nto > max_nto && !(split16 == 1 && nto == 20 && a0.kc0 == 5 && a0.stride == 1)) return -2;
    WzMbArgs a = a0;
    a.nsplit = 1;
    a.th = 4; a.tw = 4;
    a.tiles_y = (a.hout + a.th - 1) / a.th;
    a.tiles_x = (a.wout + a.tw - 1) / a.tw;
    static const int nw10 = wz_cs_env("WZ_MB_CS_NW", 8);   // waves per tile on the 10x10 maps (4 or 8)
    if (a.cin0 == 0) {   // no expand stage (block 13)
        if (nto == 10) {
            if (prepare) (void)wz_cs_launch<false, 1, 1, 1, 10, 4>(a, n, s, true);
            if (!prepare && nw10 == 4) return wz_cs_launch<false, 1, 1, 1, 10, 4>(a, n, s, false);
            return wz_cs_launch<false, 1, 1, 1, 10>(a, n, s, prepare);
        }
        return -2;
    }
    if (a.stride == 2) {   // halo 9 x 9 = 81 pixels -> 6 m-tiles
        if (a.kc0 == 1 && nto == 4) return wz_cs_launch<true, 6, 1, 1, 4>(a, n, s, prepare);
        if (a.kc0 == 3 && nto == 10) return wz_cs_launch<true, 6, 1, 3, 10>(a, n, s, prepare);   // block 13 with its expand stage (19x19 -> 10x10)
        return -2;
    }
    // stride 1: halo 6 x 6 = 36 pixels -> 3 m-tiles
#define CS_CASE(K, N) if (a.kc0 == K && nto == N) return wz_cs_launch<true, 3, 1, K, N>(a, n, s, prepare)
    CS_CASE(2, 4);
    CS_CASE(2, 6);
    CS_CASE(3, 6);
    if (a.kc0 == 5 && nto == 10) {
        if (prepare) (void)wz_cs_launch<true, 3, 1, 5, 10, 4>(a, n, s, true);
        if (!prepare && nw10 == 4) return wz_cs_launch<true, 3, 1, 5, 10, 4>(a, n, s, false);
    }
    CS_CASE(5, 10);
    if (!prepare && split16 == 1 && a.kc0 == 5 && nto == 20) {
        a.nsplit = 2;
        return wz_cs_launch<true, 3, 1, 5, 10>(a, n, s, false);
    }
    CS_CASE(5, 20);
#undef CS_CASE
    return -2;
}
