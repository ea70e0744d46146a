// One inverted-residual block of MobileNet-v2 as ONE launch on gfx950:
//   [1x1 expand + BN + ReLU6] -> depthwise 3x3 + BN + ReLU6 -> 1x1 project + BN [+ residual]
//
// In the reference the block is ~10 graph nodes inside `sess.run` (`watsor/detection/tensorflow_cpu.py:114-115`;
// layer shapes SURVEY.md Appendix A).  Unfused, the 6x-expanded tensor makes two round trips through HBM
// (22.0 + 14.9 of the 51.3 MB/frame of activation traffic); here it never leaves the CU:
//
//   * a workgroup owns a TH x TW tile of output pixels of one frame and the (TH-1)s+3 x (TW-1)s+3 halo of
//     input pixels under it; the halo's input channels are loaded ONCE, straight into MFMA B-operand
//     fragments (lane = pixel, 8 consecutive channels = one 16-byte load) and stay in registers;
//   * the expanded channels are walked in chunks of CE: the expand GEMM (v_mfma_f32_16x16x32_f16, weights =
//     A operand) writes the chunk for every halo pixel to LDS as fp16 (zero outside the frame: TF pads the
//     depthwise INPUT), one barrier, then every lane computes the depthwise 3x3 of 8 channels of one
//     output pixel from LDS -- which is exactly the B-operand fragment of the project GEMM -- and feeds
//     it to the project MFMAs; the project accumulators (fp32) live in registers across all chunks;
//   * the chunk buffer in LDS is double-buffered, so there is one barrier per chunk;
//   * epilogue: + bias, + residual, fp16 NHWC store.
//
// Rounding points (fp16 after expand, fp16 after depthwise, fp32 accumulation in the same order) are the
// same as in the unfused kernels of k_conv.hip, so both programs produce bit-identical tensors
// (tests/test_gpu_parity.py::test_fused_blocks_equal_unfused_layers).
//
// Without an expand stage (block 0, and block 13 whose expand output is an SSD feature map and has to
// exist in HBM anyway) the depthwise taps are gathered from global memory and no LDS is used.
#include "wz_common.h"

template <int MP, int KCI>
struct MbHalo {
    half8_t xf[MP][KCI];
    bool inimg[MP];
};

// CE: expanded channels per chunk; MP: halo m-tiles (16 pixels) per wave; MQ: output m-tiles per wave;
// KCI: 32-channel K chunks of the expand conv; NTO: 16-column tiles of the project output.
template <bool EXPAND, int CE, int MP, int MQ, int KCI, int NTO>
__global__ __launch_bounds__(256) void wz_k_mbconv(const WzMbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wz_mb_smem[];
    half_t* const e_base = reinterpret_cast<half_t*>(wz_mb_smem);
    constexpr int ES = CE + 8;   // LDS row stride in halfs: 16-byte aligned rows, breaks the power-of-two stride

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    // diagnostics (a.dbg != nullptr only under wz_debug_mbconv): phase timestamps of the first and the last workgroup
    const bool stamp = !WZ_LANE_STAMPS && a.dbg && threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1);
    WZ_LANE_STAMP(a.dbg);
    unsigned long long* const dbg = a.dbg + (blockIdx.x == 0 ? 0 : 8);
#define MB_STAMP(i) do { if (stamp) dbg[i] = wall_clock64(); } while (0)
    MB_STAMP(0);
    if (stamp) dbg[5] = clock64();   // shader cycles: (dbg[6] - dbg[5]) / (dbg[4] - dbg[0]) * 100 = effective MHz
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles, t = blockIdx.x - b * tiles;
    const int tyi = t / a.tiles_x;
    const int oy0 = tyi * a.th, ox0 = (t - tyi * a.tiles_x) * a.tw;
    const int s = a.stride;
    const int hw_ = (a.tw - 1) * s + 3, hh_ = (a.th - 1) * s + 3;
    const int P = hh_ * hw_, Q = a.th * a.tw;

    const int iy_base = oy0 * s - a.pad_t, ix_base = ox0 * s - a.pad_l;
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- output pixels of this lane (one per owned output m-tile)
    int hp0[MQ];        // halo index of the top-left depthwise tap
    int opix[MQ];       // flat output pixel index in the frame batch, or -1
#pragma unroll
    for (int j = 0; j < MQ; ++j) {
        const int q = (wave + 4 * j) * 16 + r16;
        const int qc = q < Q ? q : Q - 1;
        const int qy = qc / a.tw, qx = qc - qy * a.tw;
        hp0[j] = qy * s * hw_ + qx * s;
        const int oy = oy0 + qy, ox = ox0 + qx;
        opix[j] = (q < Q && oy < a.hout && ox < a.wout) ? (b * a.hout + oy) * a.wout + ox : -1;
    }

    float4_t acc[MQ][NTO];
#pragma unroll
    for (int j = 0; j < MQ; ++j)
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) acc[j][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};

    if constexpr (EXPAND) {
        // ---- halo pixels of this lane: input channels as B fragments, loaded once
        MbHalo<MP, KCI> h;
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            const int p = (wave + 4 * i) * 16 + r16;
            const int hy = p / hw_, hx = p - hy * hw_;
            const int iy = iy_base + hy, ix = ix_base + hx;
            const bool ok = p < P && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
            h.inimg[i] = ok;
            const half_t* src = a.in + ((size_t)(b * a.hin + (ok ? iy : 0)) * a.win + (ok ? ix : 0)) * a.cin0;
#pragma unroll
            for (int c = 0; c < KCI; ++c) {
                const int k0 = c * 32 + g * 8;
                h.xf[i][c] = (ok && k0 < a.cin0) ? *reinterpret_cast<const half8_t*>(src + k0) : zero8;
            }
        }
        // halfs per chunk buffer: one row per halo pixel SLOT of the template (MP m-tiles per wave), so that the
        // expand stage needs no "does this m-tile exist" branch and the compiler can interleave the MP chains
        constexpr int ebuf = MP * 64 * ES;
        const int nchunks = (a.cmid + CE - 1) / CE;
        // this workgroup's share of the expanded channels (blockIdx.y = group; a.cpg chunks per group)
        const int ch_begin = blockIdx.y * a.cpg, ch_end = min(ch_begin + a.cpg, nchunks);
        const int c_begin = ch_begin * CE;
        const int cw = min(a.cpg * CE, a.cmid_pad - c_begin);   // multiple of 32
        // depthwise weights + bias of these channels -> LDS (read after the first barrier below)
        half_t* const wd_l = e_base + a.ebufs * ebuf;                         // [9][cw]
        float* const bd_l = reinterpret_cast<float*>(wd_l + 9 * a.cpg * CE);   // [cw]
        float* const be_l = bd_l + a.cpg * CE;                                 // [cw] expand bias
        {
            const int c8s = cw >> 3;
            for (int i = threadIdx.x; i < 9 * c8s; i += 256) {
                const int tp = i / c8s, c8 = i - tp * c8s;
                *reinterpret_cast<half8_t*>(wd_l + tp * cw + c8 * 8) =
                    *reinterpret_cast<const half8_t*>(a.wd + (size_t)tp * a.cmid_pad + c_begin + c8 * 8);
            }
            for (int i = threadIdx.x; i < (cw >> 2); i += 256) {
                *reinterpret_cast<float4_t*>(bd_l + i * 4) = *reinterpret_cast<const float4_t*>(a.bd + c_begin + i * 4);
                // (the expand bias has nmid_pad >= cmid entries; nmid_pad is a multiple of 32 like cmid_pad)
                *reinterpret_cast<float4_t*>(be_l + i * 4) =
                    (c_begin + i * 4 < a.nmid_pad) ? *reinterpret_cast<const float4_t*>(a.be + c_begin + i * 4)
                                                   : (float4_t){0.f, 0.f, 0.f, 0.f};
            }
        }
        // expand + project weights of these channels -> LDS by DMA (global_load_lds, 1 KiB MFMA fragment per
        // wave instruction, no staging registers): every global latency of the workgroup -- halo pixels,
        // depthwise weights, GEMM weights -- is in flight at once and paid once, at the barrier below.
        // (a.stage == 0 when the slice does not fit: the fragments are then read from L2 where they are used.)
        unsigned char* const we_l = reinterpret_cast<unsigned char*>(be_l + a.cpg * CE);   // [cw/16][KCI] KiB
        const int nkg = cw >> 5;                                                           // K chunks of the project conv
        unsigned char* const wp_l = we_l + (size_t)(a.cpg * CE / 16) * KCI * 1024;         // [NTO][nkg] KiB
        if (a.stage) {
            const int nwe = (cw >> 4) * KCI;
            const half_t* wsrc = a.we + ((size_t)(c_begin >> 4) * a.kc0 * 64 + lane) * 8;
            for (int fr = wave; fr < nwe; fr += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)fr * 512),
                                                 (__attribute__((address_space(3))) void*)(we_l + fr * 1024), 16, 0, 0);
            const int nwp = NTO * nkg;
            for (int fr = wave; fr < nwp; fr += 4) {
                const int nt = fr / nkg, kl = fr - nt * nkg;
                const half_t* src = a.wp + ((size_t)(nt * a.kc + (c_begin >> 5) + kl) * 64 + lane) * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(wp_l + fr * 1024), 16, 0, 0);
            }
        }
        __syncthreads();   // staged biases / weights are visible (the barrier's vmcnt(0) also covers the halo loads above)
        MB_STAMP(1);
        for (int ch = ch_begin; ch < ch_end; ++ch) {
            half_t* const E = e_base + ((ch - ch_begin) & (a.ebufs - 1)) * ebuf;
            const int ce0 = ch * CE;
            const int nte = min(CE, a.nmid_pad - ce0) >> 4;   // 16-channel tiles of this chunk
            // ---- expand: E[p][ce] = in-frame ? relu6(sum_k X[p][k] We[k][ce] + be[ce]) : 0
            for (int nt = 0; nt < nte; ++nt) {
                half8_t wa[KCI];
                if (a.stage) {
                    const unsigned char* wl = we_l + ((size_t)(((ce0 - c_begin) >> 4) + nt) * KCI * 64 + lane) * 16;
#pragma unroll
                    for (int c = 0; c < KCI; ++c) wa[c] = *reinterpret_cast<const half8_t*>(wl + c * 1024);
                } else {
                    const half_t* wsrc = a.we + ((size_t)((ce0 >> 4) + nt) * a.kc0 * 64 + lane) * 8;
#pragma unroll
                    for (int c = 0; c < KCI; ++c) wa[c] = *reinterpret_cast<const half8_t*>(wsrc + (size_t)c * 512);
                }
                const float4_t bv = *reinterpret_cast<const float4_t*>(be_l + (ce0 - c_begin) + nt * 16 + g * 4);
#pragma unroll
                for (int i = 0; i < MP; ++i) {
                    const int mt = wave + 4 * i;   // slots beyond the halo hold zeros (inimg false)
                    float4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < KCI; ++c)
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c], h.xf[i][c], d, 0, 0, 0);
                    const half4_t o = wz_relu6_pack(d, bv, h.inimg[i]);
                    *reinterpret_cast<half4_t*>(E + (mt * 16 + r16) * ES + nt * 16 + g * 4) = o;
                }
            }
            __syncthreads();
            if (ch == ch_begin) MB_STAMP(2);
            // ---- depthwise on the chunk (lane = output pixel x 8 channels) feeding the project MFMAs
            const int nkk = (nte * 16 + 31) >> 5;
            for (int kk = 0; kk < nkk; ++kk) {
                const int coff = ce0 - c_begin + kk * 32 + g * 8;   // < cw
                half8_t wt[9];
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) wt[tp] = *reinterpret_cast<const half8_t*>(wd_l + tp * cw + coff);
                const float4_t b0 = *reinterpret_cast<const float4_t*>(bd_l + coff);
                const float4_t b1 = *reinterpret_cast<const float4_t*>(bd_l + coff + 4);
                const int kg = (ce0 >> 5) + kk;            // K chunk of the project conv
#pragma unroll
                for (int j = 0; j < MQ; ++j) {
                    {   // (output m-tile slots beyond the tile compute on a clamped pixel and are never stored)
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
                        if (a.stage) {
                            const unsigned char* wl = wp_l + ((size_t)(kg - (c_begin >> 5)) * 64 + lane) * 16;
#pragma unroll
                            for (int nt = 0; nt < NTO; ++nt) {
                                const half8_t wp = *reinterpret_cast<const half8_t*>(wl + (size_t)nt * nkg * 1024);
                                acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp, bf, acc[j][nt], 0, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int nt = 0; nt < NTO; ++nt) {
                                const half8_t wp = *reinterpret_cast<const half8_t*>(
                                    a.wp + ((size_t)(nt * a.kc + kg) * 64 + lane) * 8);
                                acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp, bf, acc[j][nt], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            // two chunk buffers: no barrier here -- the next chunk's expand writes the OTHER buffer, and a wave
            // can only reach the chunk after that (same buffer again) through the barrier above, i.e. after
            // every wave has finished reading this one.  One buffer (less LDS, more workgroups per CU):
            if (a.ebufs == 1 && ch + 1 < ch_end) __syncthreads();
        }
    } else {
        // ---- no expand stage: depthwise taps gathered from global memory (zero outside the frame).
        // Every load of a K step is issued before its first use and none sits behind a branch (out-of-frame
        // taps load a clamped address and are zeroed afterwards), so a step costs ONE exposed memory latency.
        const int kk_begin = blockIdx.y * a.cpg, kk_end = min(kk_begin + a.cpg, a.kc);
        int hy0[MQ], hx0[MQ];
#pragma unroll
        for (int j = 0; j < MQ; ++j) {
            hy0[j] = hp0[j] / hw_;
            hx0[j] = hp0[j] - hy0[j] * hw_;
        }
        for (int kk = kk_begin; kk < kk_end; ++kk) {
            const int cbase = kk * 32 + g * 8;
            const int cload = min(cbase, a.cmid - 8);          // channel groups beyond cmid: clamped load, zeroed below
            const bool cok = cbase < a.cmid;
            half8_t wt[9], x[MQ][9], wp[NTO];
            bool okx[MQ][9];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
                wt[tp] = *reinterpret_cast<const half8_t*>(a.wd + (size_t)tp * a.cmid_pad + cbase);
            const float4_t b0 = *reinterpret_cast<const float4_t*>(a.bd + cbase);
            const float4_t b1 = *reinterpret_cast<const float4_t*>(a.bd + cbase + 4);
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt)
                wp[nt] = *reinterpret_cast<const half8_t*>(a.wp + ((size_t)(nt * a.kc + kk) * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < MQ; ++j)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int iy = iy_base + hy0[j] + ky, ix = ix_base + hx0[j] + kx;
                        okx[j][ky * 3 + kx] = cok && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
                        const int cy = min(max(iy, 0), a.hin - 1), cx = min(max(ix, 0), a.win - 1);
                        x[j][ky * 3 + kx] = *reinterpret_cast<const half8_t*>(
                            a.in + ((size_t)(b * a.hin + cy) * a.win + cx) * a.cmid + cload);
                    }
#pragma unroll
            for (int j = 0; j < MQ; ++j) {
                float d[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { d[r] = b0[r]; d[4 + r] = b1[r]; }
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const half8_t xv = okx[j][tp] ? x[j][tp] : zero8;
#pragma unroll
                    for (int r = 0; r < 8; ++r) d[r] = fmaf((float)xv[r], (float)wt[tp][r], d[r]);
                }
                half8_t bf;
#pragma unroll
                for (int r = 0; r < 8; ++r) bf[r] = (half_t)fminf(fmaxf(d[r], 0.0f), 6.0f);
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
                    acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[nt], bf, acc[j][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane holds output channels nt*16 + g*4 .. +3 of its output pixel
    MB_STAMP(3);
    if (a.nsplit > 1) {   // raw fp32 partial sums of this channel group; wz_k_splitk_reduce finishes the block
#pragma unroll
        for (int j = 0; j < MQ; ++j) {
            if (opix[j] < 0) continue;
            float* o = a.ws + ((size_t)blockIdx.y * a.M + opix[j]) * a.n_pad + g * 4;
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) *reinterpret_cast<float4_t*>(o + nt * 16) = acc[j][nt];
        }
        MB_STAMP(4);
        if (stamp) dbg[6] = clock64();
        return;
    }
#pragma unroll
    for (int j = 0; j < MQ; ++j) {
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
    MB_STAMP(4);
    if (stamp) dbg[6] = clock64();
#undef MB_STAMP
}

// ---------------------------------------------------------------------------------------------
// host side: tile choice and dispatch
// ---------------------------------------------------------------------------------------------
struct MbCfg {
    int th, tw, ce, mp, mq, nsplit, cpg;
};

static int wz_mb_env(const char* name, int dflt) {
    const char* e = wz_dev_getenv(name);
    return (e && atoi(e) > 0) ? atoi(e) : dflt;
}

// Tile of output pixels per workgroup and the split of the expanded channels over blockIdx.y.
// Output m-tiles (16 pixels) per wave MQ <= 2, halo m-tiles per wave MP <= 5.  The small late layers
// (19x19, 10x10) have few tiles and many channel chunks: a serial walk over the chunks is one exposed
// memory latency after the other, so the chunks are spread over workgroups instead (fp32 partial sums,
// summed in a fixed order by wz_k_splitk_reduce).
static MbCfg wz_mb_choose(const WzMbArgs& a, int n) {
    MbCfg c;
    int th = 8, tw = 8;
    if (a.wout <= 19 && a.stride == 1) { th = 5; tw = 10; }   // 10x10: 2 tiles, 19x19: 4 x 2 tiles per frame
    else if (a.wout <= 10) { th = 5; tw = 10; }
    c.th = wz_mb_env("WZ_MB_TH", th);
    c.tw = wz_mb_env("WZ_MB_TW", tw);
    if (a.wout <= 10) { c.th = wz_mb_env("WZ_MB_TH10", c.th); c.tw = wz_mb_env("WZ_MB_TW10", c.tw); }
    else if (a.wout <= 19) { c.th = wz_mb_env("WZ_MB_TH19", c.th); c.tw = wz_mb_env("WZ_MB_TW19", c.tw); }
    else if (a.wout <= 38) { c.th = wz_mb_env("WZ_MB_TH38", c.th); c.tw = wz_mb_env("WZ_MB_TW38", c.tw); }
    else if (a.wout <= 75) { c.th = wz_mb_env("WZ_MB_TH75", c.th); c.tw = wz_mb_env("WZ_MB_TW75", c.tw); }
    else { c.th = wz_mb_env("WZ_MB_TH150", c.th); c.tw = wz_mb_env("WZ_MB_TW150", c.tw); }
    if (c.th > a.hout) c.th = a.hout;
    if (c.tw > a.wout) c.tw = a.wout;
    for (;;) {
        const int P = ((c.th - 1) * a.stride + 3) * ((c.tw - 1) * a.stride + 3), Q = c.th * c.tw;
        const int np = (P + 15) / 16, nq = (Q + 15) / 16;
        c.mp = a.cin0 ? (np + 3) / 4 : 1;
        c.mq = (nq + 3) / 4;
        if ((c.mp <= 5 && c.mq <= 2) || (c.th == 1 && c.tw == 1)) break;
        if (c.th >= c.tw) c.th = (c.th + 1) / 2; else c.tw = (c.tw + 1) / 2;   // requested tile too large: halve it
    }
    c.ce = (c.mp > 3) ? 32 : 64;
    const int units = a.cin0 ? (a.cmid + c.ce - 1) / c.ce : a.kc;   // chunks (expand) or 32-channel K steps (no expand)
    const int tiles = ((a.hout + c.th - 1) / c.th) * ((a.wout + c.tw - 1) / c.tw) * n;
    // workgroups wanted per launch: half the chip.  256 (round 1 / 2) gives the shortest launch (block 16: 9.1 + 4.4 us against
    // 15.8 + 2.8) -- and 50.9 k frames/s against 52.0 k with four lanes in flight: the other lanes' launches want the CUs
    // (profiles/r03_wave_counts_and_cu_footprints.txt)
    const int target = wz_mb_env("WZ_MB_WGS", wz_latency_schedule() ? 256 : 128);
    int nsplit = (target + tiles - 1) / tiles;
    const int max_split = wz_mb_env("WZ_MB_MAXSPLIT", 16);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit > units) nsplit = units;
    if (nsplit < 1 || !a.ws) nsplit = 1;
    while (nsplit > 1 && (size_t)nsplit * a.M * a.n_pad * 4 > a.ws_bytes) --nsplit;
    c.cpg = (units + nsplit - 1) / nsplit;
    c.nsplit = (units + c.cpg - 1) / c.cpg;
    return c;
}

template <bool EXPAND, int CE, int MP, int MQ, int KCI, int NTO>
static int wz_mb_launch(WzMbArgs a, const MbCfg& c, int n, hipStream_t s, bool prepare) {
    a.th = c.th;
    a.tw = c.tw;
    a.tiles_y = (a.hout + c.th - 1) / c.th;
    a.tiles_x = (a.wout + c.tw - 1) / c.tw;
    a.nsplit = c.nsplit;
    a.cpg = c.cpg;
    size_t lds = 0;
    a.stage = 0;
    if (EXPAND) {
        const int P = ((c.th - 1) * a.stride + 3) * ((c.tw - 1) * a.stride + 3);
        (void)P;
        a.ebufs = (c.cpg > 1 && wz_mb_env("WZ_MB_EBUFS", 1) == 2) ? 2 : 1;
        lds = (size_t)a.ebufs * MP * 64 * (CE + 8) * sizeof(half_t)              // chunk buffer(s)
              + (size_t)c.cpg * CE * (9 * sizeof(half_t) + 2 * sizeof(float));   // depthwise weights + bias, expand bias
        const size_t wbytes = (size_t)(c.cpg * CE / 16) * KCI * 1024 + (size_t)NTO * (c.cpg * CE / 32) * 1024;
        if (lds + wbytes <= (size_t)wz_mb_env("WZ_MB_STAGE_KB", 96) * 1024) {    // GEMM weights of the group fit in LDS
            a.stage = 1;
            lds += wbytes;
        }
    }
    auto k = wz_k_mbconv<EXPAND, CE, MP, MQ, KCI, NTO>;
    if (prepare) {   // kernel attributes cannot be set while a stream is capturing: done once at engine creation
        if (EXPAND)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
        return lds <= 160 * 1024 ? 0 : -1;
    }
    WZ_LAUNCH(k, dim3(a.tiles_x * a.tiles_y * n, c.nsplit), dim3(256), lds, s, a);
    return c.nsplit;
}

// (KCI, NTO) pairs of SSD-MobileNet-v2: expand K chunks x project output tiles
template <bool EXPAND, int CE, int MP, int MQ>
static int wz_mb_dispatch(const WzMbArgs& a, const MbCfg& c, int n, hipStream_t s, bool prepare) {
    const int nto = a.n_pad / 16, kci = EXPAND ? a.kc0 : 1;
#define WZ_MB_CASE(K, N) \
    if (kci == K && nto == N) return wz_mb_launch<EXPAND, CE, MP, MQ, K, N>(a, c, n, s, prepare)
    WZ_MB_CASE(1, 2);
    WZ_MB_CASE(1, 4);
    WZ_MB_CASE(1, 10);
    if constexpr (EXPAND) {
        WZ_MB_CASE(2, 4);
        WZ_MB_CASE(2, 6);
        WZ_MB_CASE(3, 6);
        WZ_MB_CASE(5, 10);
        WZ_MB_CASE(5, 20);
    }
#undef WZ_MB_CASE
    return -1;
}

// prepare = true: validate the shape and set kernel attributes, launch nothing (engine creation; n = the
// largest batch).  Returns -1 when the block's shape has no instantiation (the engine then refuses the
// file), else the number of channel groups launched: > 1 means a.ws holds that many fp32 partial tensors
// [group][M][n_pad] and the caller has to enqueue wz_k_splitk_reduce.
int wz_launch_mbconv(const WzMbArgs& a, int n, hipStream_t s, bool prepare) {
    const MbCfg c = wz_mb_choose(a, n);
    if (c.mq < 1 || c.mq > 2 || c.mp < 1 || c.mp > 5) return -1;
    if (a.cin0 == 0) {
        if (c.mq == 1) return wz_mb_dispatch<false, 32, 1, 1>(a, c, n, s, prepare);
        return wz_mb_dispatch<false, 32, 1, 2>(a, c, n, s, prepare);
    }
    if (c.mp <= 2 && c.mq == 1) return wz_mb_dispatch<true, 64, 2, 1>(a, c, n, s, prepare);
    if (c.mp <= 3) return wz_mb_dispatch<true, 64, 3, 2>(a, c, n, s, prepare);
    if (c.mq == 1) return wz_mb_dispatch<true, 32, 5, 1>(a, c, n, s, prepare);
    return wz_mb_dispatch<true, 32, 5, 2>(a, c, n, s, prepare);
}
