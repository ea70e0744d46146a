// Inverted-residual block, one WAVEFRONT per pixel tile (the large maps: 150x150 ... 38x38).
//
// k_mbconv.hip gives a tile to a 4-wave workgroup and walks the expanded channels through one LDS buffer
// with a workgroup barrier per chunk; on the large maps that is a chain of ~1 us phases per workgroup with
// the CU mostly waiting.  Here every wave owns a small tile (4x8 outputs at stride 1, 4x4 at stride 2)
// and its own few KiB of LDS, and runs the same three stages with NO workgroup barrier in the loop: waves
// drift apart and cover each other's latencies, and 16-20 of them fit on a CU.
//
//   * halo pixels -> MFMA B fragments in registers, once;
//   * per 32-channel chunk: expand MFMA -> fp16 chunk in the wave's LDS region -> depthwise 3x3 in fp32
//     (each lane = one output pixel x 8 channels = one B fragment of the project MFMA) -> project MFMA
//     into fp32 accumulators that live in registers across all chunks;
//   * the chunk's GEMM weight fragments are prefetched from L2 one chunk ahead; depthwise weights and the
//     biases are staged in LDS once per workgroup (the only barrier of the kernel).
//
// Rounding points and accumulation orders are those of k_conv.hip / k_mbconv.hip: the outputs are
// bit-identical (tests/test_gpu_parity.py::test_fused_blocks_equal_unfused_layers).
#include "wz_common.h"

// MPW: halo m-tiles (16 pixels) per wave, MQW: output m-tiles per wave, KCI: K chunks of the expand conv,
// NTO: 16-column tiles of the project output, NKK: 32-channel chunks per pass (their two dependency chains --
// expand -> LDS -> depthwise -> project -- are independent, so a pass of 2 costs about the latency of 1).
// STEM: the block is the network's first one and its "expand" stage is the stem convolution (3x3 s2, 3 -> 32,
// K = 27 padded to 32): the B fragment of a halo pixel is the im2col of its 3x3x3 window, gathered straight from
// the normalised 300x300x4 input -- the 150x150x32 stem output never exists in HBM.
template <int MPW, int MQW, int KCI, int NTO, int NKK, bool STEM>
__global__ __launch_bounds__(256, 4) void wz_k_mbconv_wave(const WzMbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wz_mbw_smem[];
    WZ_LANE_STAMP(a.dbg);
    constexpr int CE = 32 * NKK, ES = CE + 8;   // NKK 32-channel K chunks of the project conv per pass
    constexpr int EBYTES = MPW * 16 * ES * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    half_t* const E = reinterpret_cast<half_t*>(wz_mbw_smem + wave * EBYTES);
    half_t* const wd_l = reinterpret_cast<half_t*>(wz_mbw_smem + 4 * EBYTES);   // [9][cmid_pad]
    float* const bd_l = reinterpret_cast<float*>(wd_l + 9 * a.cmid_pad);         // [cmid_pad]
    float* const be_l = bd_l + a.cmid_pad;                                       // [cmid_pad] expand bias

    // ---- staged once per workgroup: depthwise weights, depthwise bias, expand bias
    {
        const int c8s = a.cmid_pad >> 3;
        for (int i = threadIdx.x; i < 9 * c8s; i += 256)
            *reinterpret_cast<half8_t*>(wd_l + i * 8) = *reinterpret_cast<const half8_t*>(a.wd + (size_t)i * 8);
        for (int i = threadIdx.x; i < (a.cmid_pad >> 2); i += 256) {
            *reinterpret_cast<float4_t*>(bd_l + i * 4) = *reinterpret_cast<const float4_t*>(a.bd + i * 4);
            *reinterpret_cast<float4_t*>(be_l + i * 4) = (i * 4 < a.nmid_pad) ? *reinterpret_cast<const float4_t*>(a.be + i * 4)
                                                                              : (float4_t){0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- this wave's tile
    const int tiles = a.tiles_x * a.tiles_y;
    const int wt = blockIdx.x * 4 + wave;
    const bool live = wt < tiles * a.nb;              // wave-uniform; dead waves still take the barrier below
    const int wtc = live ? wt : 0;
    const int b = wtc / tiles, t = wtc - b * tiles;
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
        opix[j] = (live && q < Q && oy < a.hout && ox < a.wout) ? (b * a.hout + oy) * a.wout + ox : -1;
    }

    // halo pixels of this lane: input channels as B fragments
    half8_t xf[MPW][KCI];
    bool inimg[MPW];
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int p = i * 16 + r16;
        const int hy = p / hw_, hx = p - hy * hw_;
        const int iy = iy_base + hy, ix = ix_base + hx;
        const bool ok = live && p < P && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
        inimg[i] = ok;
        if constexpr (STEM) {
            // k = g*8 + j = (ky*3 + kx)*3 + c: this lane needs taps tb .. tb+3 with tb = (g*8)/3 = {0, 2, 5, 8}
            const int tb = (g * 8) / 3;
            half4_t tp[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tq = tb + q;
                const int ky = tq / 3, kx = tq - ky * 3;
                const int sy = iy * 2 - a.spad_t + ky, sx = ix * 2 - a.spad_l + kx;   // stem: stride 2 on the input image
                const bool in = ok && tq < 9 && sy >= 0 && sy < a.sin_h && sx >= 0 && sx < a.sin_w;
                const int cy = min(max(sy, 0), a.sin_h - 1), cx = min(max(sx, 0), a.sin_w - 1);
                const half4_t v = *reinterpret_cast<const half4_t*>(a.in + ((size_t)(b * a.sin_h + cy) * a.sin_w + cx) * 4);
                tp[q] = in ? v : (half4_t){0, 0, 0, 0};
            }
            half8_t x;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // element j of lane group g: tap (g*8+j)/3 - tb, channel (g*8+j)%3, zero for k >= 27; the four cases are
                // compile-time constants, g picks one
                half_t e[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int k = gg * 8 + j;
                    e[gg] = k < 27 ? tp[k / 3 - (gg * 8) / 3][k % 3] : (half_t)0.0f;
                }
                x[j] = g == 0 ? e[0] : g == 1 ? e[1] : g == 2 ? e[2] : e[3];
            }
            xf[i][0] = x;
        } else {
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

    // GEMM weight fragments of a pass: 2 NKK expand channel tiles x KCI and NKK x NTO project tiles, straight from L2
    const int nk32 = a.cmid_pad >> 5;                       // 32-channel chunks in all
    const int npass = (nk32 + NKK - 1) / NKK;
    const int ntiles_e = a.nmid_pad >> 4;
    half8_t wa[2 * NKK][KCI], wp[NKK][NTO];
    auto load_weights = [&](int ps) {
#pragma unroll
        for (int nt = 0; nt < 2 * NKK; ++nt) {
            const int tn = min(ps * 2 * NKK + nt, ntiles_e - 1);   // beyond the packed tiles: clamped, never used (see `have`)
            const half_t* wsrc = a.we + ((size_t)tn * a.kc0 * 64 + lane) * 8;
#pragma unroll
            for (int c = 0; c < KCI; ++c) wa[nt][c] = *reinterpret_cast<const half8_t*>(wsrc + (size_t)c * 512);
        }
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int kg = min(ps * NKK + kk, nk32 - 1);
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt)
                wp[kk][nt] = *reinterpret_cast<const half8_t*>(a.wp + ((size_t)(nt * a.kc + kg) * 64 + lane) * 8);
        }
    };
    load_weights(0);
    __syncthreads();   // staged depthwise weights / biases visible; the only workgroup barrier
    if (!live) return;

    for (int ps = 0; ps < npass; ++ps) {
        half8_t wa_c[2 * NKK][KCI], wp_c[NKK][NTO];
#pragma unroll
        for (int nt = 0; nt < 2 * NKK; ++nt)
#pragma unroll
            for (int c = 0; c < KCI; ++c) wa_c[nt][c] = wa[nt][c];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) wp_c[kk][nt] = wp[kk][nt];
        if (ps + 1 < npass) load_weights(ps + 1);   // in flight under this pass
        const int ce0 = ps * CE;
        // ---- expand: E[p][ce] = in-frame ? relu6(sum_k X[p][k] We[k][ce] + be[ce]) : 0
#pragma unroll
        for (int nt = 0; nt < 2 * NKK; ++nt) {
            const bool have = ce0 + nt * 16 < a.nmid_pad;    // this 16-channel tile exists (wave-uniform)
            const float4_t bv = *reinterpret_cast<const float4_t*>(be_l + min(ce0 + nt * 16, a.cmid_pad - 16) + g * 4);
#pragma unroll
            for (int i = 0; i < MPW; ++i) {
                float4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < KCI; ++c) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa_c[nt][c], xf[i][c], d, 0, 0, 0);
                const half4_t o = wz_relu6_pack(d, bv, inimg[i] && have);
                *reinterpret_cast<half4_t*>(E + (i * 16 + r16) * ES + nt * 16 + g * 4) = o;
            }
        }
        // the wave's own LDS writes are ordered before its reads by the LDS queue; keep the compiler from
        // moving the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- depthwise (lane = output pixel x 8 channels) feeding the project MFMAs
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            if (ps * NKK + kk < nk32) {   // wave-uniform (the last pass may hold fewer chunks)
                const int coff = ce0 + kk * 32 + g * 8;
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
                    const half_t* ep = E + hp0[j] * ES + kk * 32 + g * 8;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const half8_t x = *reinterpret_cast<const half8_t*>(ep + (ky * hw_ + kx) * ES);
#pragma unroll
                            for (int r = 0; r < 8; ++r) d[r] = fmaf((float)x[r], (float)wt[ky * 3 + kx][r], d[r]);
                        }
                    half8_t bf;
#pragma unroll
                    for (int r = 0; r < 8; ++r) bf[r] = (half_t)fminf(fmaxf(d[r], 0.0f), 6.0f);
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt)
                        acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp_c[kk][nt], bf, acc[j][nt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (the next pass's E stores stay behind these reads)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---- epilogue
#pragma unroll
    for (int j = 0; j < MQW; ++j) {
        if (opix[j] < 0) continue;
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            const int n4 = nt * 16 + g * 4;
            if (n4 >= a.cout) continue;
            const float4_t bv = *reinterpret_cast<const float4_t*>(a.bp + n4);
            float4_t v = acc[j][nt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
            const size_t o = (size_t)opix[j] * a.cout + n4;
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
static int wz_mbw_env(const char* name, int dflt) {
    const char* e = wz_dev_getenv(name);
    return (e && atoi(e) > 0) ? atoi(e) : dflt;
}

template <int MPW, int MQW, int KCI, int NTO, int NKK, bool STEM = false>
static int wz_mbw_launch(WzMbArgs a, int n, hipStream_t s, bool prepare) {
    a.nb = n;
    const size_t lds = (size_t)4 * MPW * 16 * (32 * NKK + 8) * 2 + (size_t)a.cmid_pad * (9 * 2 + 2 * 4);
    auto k = wz_k_mbconv_wave<MPW, MQW, KCI, NTO, NKK, STEM>;
    if (prepare) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return lds <= 160 * 1024 ? 0 : -1;
    }
    const int waves = a.tiles_x * a.tiles_y * n;
    WZ_LAUNCH(k, dim3((waves + 3) / 4), dim3(256), lds, s, a);
    return 1;
}

// Serves a block iff it has an expand stage with one K chunk (cin0 <= 32), few project tiles and a map large
// enough that pixel tiles alone fill the GPU.  Returns -2 when it does not apply (the caller falls back to
// wz_launch_mbconv), -1 on an unsupported shape, else 1.
int wz_launch_mbconv_wave(const WzMbArgs& a0, int n, hipStream_t s, bool prepare) {
    static const int enabled = wz_mbw_env("WZ_MB_WAVE", 1);
    static const int min_w = wz_mbw_env("WZ_MB_WAVE_MIN_W", 38);
    const int nto = a0.n_pad / 16;
    if (a0.stem) {   // stem + first block: only this kernel implements it
        if (a0.kc0 != 1 || a0.stride != 1 || nto != 2) return -1;
        WzMbArgs a = a0;
        a.nsplit = 1;
        static const int big = wz_mbw_env("WZ_MB_WAVE_STEM_TILE", 1);   // 1: 4x8 outputs per wave, 2: 8x8, 3: 4x4
        a.th = big == 2 ? 8 : 4; a.tw = big == 3 ? 4 : 8;
        a.tiles_y = (a.hout + a.th - 1) / a.th;
        a.tiles_x = (a.wout + a.tw - 1) / a.tw;
        if (big == 2) return wz_mbw_launch<7, 4, 1, 2, 1, true>(a, n, s, prepare);   // halo 10 x 10 = 100 pixels
        if (big == 3) return wz_mbw_launch<3, 1, 1, 2, 1, true>(a, n, s, prepare);   // 4x4 outputs, halo 6 x 6 = 36
        return wz_mbw_launch<4, 2, 1, 2, 1, true>(a, n, s, prepare);
    }
    if (enabled != 1 || a0.cin0 == 0 || a0.kc0 != 1 || a0.wout < min_w) return -2;
    if (nto != 2 && nto != 4) return -2;
    WzMbArgs a = a0;
    a.nsplit = 1;
    if (a.stride == 1) { a.th = 4; a.tw = 8; } else { a.th = 4; a.tw = 4; }
    a.tiles_y = (a.hout + a.th - 1) / a.th;
    a.tiles_x = (a.wout + a.tw - 1) / a.tw;
    static const int nkk = wz_mbw_env("WZ_MB_WAVE_NKK", 1);   // measured: 2 chunks per pass cost occupancy and gain nothing
    if (nkk == 2) {
        if (a.stride == 1)   // halo 6 x 10 = 60 pixels -> 4 m-tiles, 32 outputs -> 2 m-tiles
            return nto == 2 ? wz_mbw_launch<4, 2, 1, 2, 2>(a, n, s, prepare) : wz_mbw_launch<4, 2, 1, 4, 2>(a, n, s, prepare);
        // stride 2: halo 9 x 9 = 81 pixels -> 6 m-tiles, 16 outputs -> 1 m-tile
        return nto == 2 ? wz_mbw_launch<6, 1, 1, 2, 2>(a, n, s, prepare) : wz_mbw_launch<6, 1, 1, 4, 2>(a, n, s, prepare);
    }
    if (a.stride == 1)
        return nto == 2 ? wz_mbw_launch<4, 2, 1, 2, 1>(a, n, s, prepare) : wz_mbw_launch<4, 2, 1, 4, 1>(a, n, s, prepare);
    return nto == 2 ? wz_mbw_launch<6, 1, 1, 2, 1>(a, n, s, prepare) : wz_mbw_launch<6, 1, 1, 4, 1>(a, n, s, prepare);
}
