// The 10x10 inverted-residual blocks of the ROBUST program (expanded_conv_13 .. 16: 96 -> 576 -> 160 at stride 2 from the 19x19 map,
// 160 -> 960 -> 160 / 320 on 10x10; inside `sess.run` of watsor/detection/tensorflow_cpu.py:113-115) as TWO GEMM-shaped launches per
// block instead of one (the decomposition profiles/r05_blocks_13_16_tile_4x8.txt names in its last paragraph).
//
// Why.  The one-launch form (k_mbconv_hp.hip, lean chunk-split builds) gives a 4 x 4 output tile to a workgroup that walks ALL of the
// block's expanded channels: 72 workgroups at batch 8, each streaming the block's 1.2 - 1.8 MB of split weights through one CU's
// vector-memory path (~130 GB/s: 9.6 - 10.2 us of a 16 - 19 us launch, profiles/r05_cu_stream_microbench.txt), each repeating the expand
// GEMM on a 6 x 6 halo padded to 48 pixels for 16 outputs (3 x the matrix work), on 28 % of the CUs.  Here
//
// Since late round 6 this is the ONLY form of these blocks (the one-launch lean builds, their channel groups over workgroups and the ticketed
// cross-workgroup sum left k_mbconv_hp.hip): it is the faster one at every batch size under both schedules.
//
//   launch A  wz_k_hp2_expdw   workgroup = (frame, band of output rows, group of NW 32-channel chunks), one chunk per WAVE:
//             the band's input pixels (hi + lo fragments) are fetched once per workgroup and shared through LDS; a wave expands
//             them for its 32 channels (three-term split products on v_mfma_f32_16x16x32_f16, relu6 / 6 -> unorm16 chunk buffer in
//             LDS exactly as in the one-launch kernel), runs the depthwise 3x3 in fp32 on the buffer and stores relu6 of the result,
//             split into hi + lo halves, as B FRAGMENTS of the project GEMM: D[pixel / 16][chunk][hi | lo][64 lanes][8 halves]
//             (3 MB at batch 8, it stays in L2 / Infinity Cache).  A workgroup streams NW x 20 KB of weights + <= 40 KB of input.
//             Block 13 also stores its expanded tensor (the first SSD feature map, plain fp16) from here: WzMbArgs::out2.
//   launch B  wz_k_hp2_proj    a plain split-operand GEMM over the whole batch's pixels: workgroup = MT x NT tiles of 16 pixels x 16
//             channels, K dealt out over the 8 waves (both operands arrive as whole 1 KiB fragments, straight from L2 into
//             registers, every load of a wave in flight before its first MFMA), partial tiles summed through LDS in wave order,
//             then the epilogue of the one-launch kernel (bias, residual pair, hi + lo split or [hi | hi]).  A workgroup streams
//             (MT + NT) x 2 KiB per chunk: 240 KB for 2 x 2.
//
// Same arithmetic at the same rounding points as the one-launch form (linear unorm16 chunk buffer, fp32 depthwise, three-term
// products); only the fp32 summation order of the project stage differs (K over 8 waves here), fixed from run to run.
#include "k_hp_ops.h"

#define HP2_ES 40   // unorm16 per pixel of the chunk buffer: 32 channels + 8 of padding (as in k_mbconv_hp.hip)

// S: stride; KCI: 32-channel K chunks of the expand conv; NW: waves per workgroup; OHR: output rows per band;
// MPW / MQW: 16-pixel tiles that cover a band's in-frame input pixels / its output pixels; TAP: the expanded tensor is a second output.
// PAIR: TWO waves per chunk (NW / 2 chunks per workgroup) -- wave 2c + h expands the 16-channel tile h of chunk c for all of the band's pixels into the
// chunk's shared buffer and, behind a second barrier, runs the depthwise stage for every other output tile: twice the workgroups, half the serial
// chain per wave.  For the launches that leave the chip half empty (128 workgroups of 4 waves at batch 8: a wave per SIMD on half of the CUs).
template <int S, int KCI, int NW, int OHR, int MPW, int MQW, bool TAP, bool PAIR = false>
__global__ __launch_bounds__(NW * 64, 2) void wz_k_hp2_expdw(const WzMbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wz_hp2_smem[];
    WZ_LANE_STAMP(a.dbg);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r16 = lane & 15, g = lane >> 4;
    const int nk32 = a.cmid_pad >> 5;
    constexpr int CPWG = PAIR ? NW / 2 : NW;              // chunks per workgroup
    constexpr int NTW = PAIR ? 1 : 2;                     // 16-channel tiles of the chunk a wave expands
    const int cgw = (nk32 + CPWG - 1) / CPWG;             // chunk groups
    const int nr = (a.hout + OHR - 1) / OHR;              // bands per frame
    int id = (int)blockIdx.x;
    const int cgp = id % cgw;
    id /= cgw;
    const int r = id % nr, b = id / nr;
    const int ci = PAIR ? wave >> 1 : wave, half = PAIR ? wave & 1 : 0;
    const int ps = cgp * CPWG + ci;                       // this wave's chunk
    const bool havec = ps < nk32;
    const int psc = havec ? ps : nk32 - 1;
    const int ce0 = psc * 32;
    const int oy0 = r * OHR, oh = min(OHR, a.hout - oy0);
    const int ey0 = oy0 * S - a.pad_t;                    // frame row of the chunk buffer's row 0
    const int iy0 = max(ey0, 0), iy1 = min(ey0 + (oh - 1) * S + 3, a.hin);
    const int npix = (iy1 - iy0) * a.win;                 // in-frame input pixels of the band: one contiguous run of rows
    const int EW = (a.wout - 1) * S + 3;
    constexpr int EH = (OHR - 1) * S + 3;
    const int ebytes = (EH * EW * HP2_ES * 2 + 15) & ~15;
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- the band's input pixels as B fragments (hi, lo), fetched once per workgroup: fragment f = (i * KCI + c) * 2 + (0: hi, 1: lo)
    // by wave f % NW, parked at f KiB
    constexpr int NFRAG = MPW * KCI * 2;
    constexpr int PERW = (NFRAG + NW - 1) / NW;
    half8_t part[PERW];
    {
        const half_t* const rbase = a.in + (size_t)((b * a.hin + iy0) * a.win) * (2 * a.cin0);
#pragma unroll
        for (int k = 0; k < PERW; ++k) {
            const int f = wave + k * NW;
            part[k] = zero8;
            if (f < NFRAG) {
                const int i = f / (KCI * 2), c = (f >> 1) % KCI, lo = f & 1;
                const int p = i * 16 + r16;
                const int k0 = c * 32 + g * 8;
                if (p < npix && k0 < a.cin0) part[k] = *reinterpret_cast<const half8_t*>(rbase + (size_t)p * (2 * a.cin0) + (lo ? a.cin0 : 0) + k0);
            }
        }
    }
    // ---- this wave's weights: expand fragments (2 n-tiles x KCI, hi and lo), depthwise taps and both biases, all in flight together
    half8_t wah[NTW][KCI], wal[NTW][KCI];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const size_t off = ((size_t)(psc * 2 + nt + half) * a.kc0 * 64 + lane) * 8;
#pragma unroll
        for (int c = 0; c < KCI; ++c) {
            wah[nt][c] = *reinterpret_cast<const half8_t*>(a.we + off + (size_t)c * 512);
            wal[nt][c] = *reinterpret_cast<const half8_t*>(a.we_lo + off + (size_t)c * 512);
        }
    }
    float4_t bv[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) bv[nt] = *reinterpret_cast<const float4_t*>(a.be + ce0 + (nt + half) * 16 + g * 4);
    const float* const wd32 = reinterpret_cast<const float*>(a.wd);      // [9][cmid_pad], already * 6 / 65535
    const int coff = ce0 + g * 8;
    float4_t wt0[9], wt1[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        wt0[tp] = *reinterpret_cast<const float4_t*>(wd32 + (size_t)tp * a.cmid_pad + coff);
        wt1[tp] = *reinterpret_cast<const float4_t*>(wd32 + (size_t)tp * a.cmid_pad + coff + 4);
    }
    const float4_t b0 = *reinterpret_cast<const float4_t*>(a.bd + coff);
    const float4_t b1 = *reinterpret_cast<const float4_t*>(a.bd + coff + 4);

    // ---- this wave's chunk buffer: zero (padding stays zero: code 0 = value 0), then the halo fragments are parked
    unsigned char* const Eb = wz_hp2_smem + (size_t)NFRAG * 1024 + (size_t)ci * ebytes;
    unsigned short* const E = reinterpret_cast<unsigned short*>(Eb);
    for (int o = (half * 64 + lane) * 16; o < ebytes; o += (PAIR ? 2048 : 1024)) *reinterpret_cast<wz_u32x4_t*>(Eb + o) = (wz_u32x4_t){0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < PERW; ++k) {
        const int f = wave + k * NW;
        if (f < NFRAG) *reinterpret_cast<half8_t*>(wz_hp2_smem + (size_t)f * 1024 + lane * 16) = part[k];
    }
    __syncthreads();   // the band's fragments are in LDS, the chunk buffers zeroed
    if (!PAIR && !havec) return;

    // ---- expand: E[pixel][ce] = unorm16(clamp((sum_k X[pixel][k] We[k][ce] + be[ce]) / 6, 0, 1)) for the band's in-frame pixels
    const float rcp_win = 1.0f / (float)a.win;   // p < 96, win <= 19: floor((p + 0.5) / win) is exact in fp32
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        if (i * 16 >= npix || !havec) break;     // (uniform)
        half8_t fh[KCI], fl[KCI];
#pragma unroll
        for (int c = 0; c < KCI; ++c) {
            fh[c] = *reinterpret_cast<const half8_t*>(wz_hp2_smem + (size_t)((i * KCI + c) * 2) * 1024 + lane * 16);
            fl[c] = *reinterpret_cast<const half8_t*>(wz_hp2_smem + (size_t)((i * KCI + c) * 2 + 1) * 1024 + lane * 16);
        }
        float4_t d[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) d[nt] = bv[nt];
#pragma unroll
        for (int c = 0; c < KCI; ++c)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) d[nt] = WZ_HP_MFMA(wal[nt][c], fh[c], d[nt]);
#pragma unroll
        for (int c = 0; c < KCI; ++c)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) d[nt] = WZ_HP_MFMA(wah[nt][c], fl[c], d[nt]);
#pragma unroll
        for (int c = 0; c < KCI; ++c)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) d[nt] = WZ_HP_MFMA(wah[nt][c], fh[c], d[nt]);
        const int p = i * 16 + r16;
        const bool keep = p < npix;
        const int py = (int)(((float)p + 0.5f) * rcp_win), px = p - py * a.win;
        const int iy = iy0 + py;
        unsigned short* const erow = E + ((iy - ey0) * EW + px + a.pad_l) * HP2_ES + g * 4;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            wz_u32x2_t o;
            o[0] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pknorm_u16(d[nt][0], d[nt][1]));
            o[1] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pknorm_u16(d[nt][2], d[nt][3]));
            if (keep) *reinterpret_cast<wz_u32x2_t*>(erow + (nt + half) * 16) = o;
            if constexpr (TAP) {
                // the second output: relu6 of the expanded value as plain fp16 (d carries the 1 / 6 of the chunk buffer), for the rows this band
                // OWNS -- [oy0 * S, (oy0 + OHR) * S): every input row exactly once over the bands
                if (keep && a.out2 && iy >= oy0 * S && iy < (oy0 + OHR) * S) {
                    half4_t t2;
#pragma unroll
                    for (int q = 0; q < 4; ++q) t2[q] = (half_t)(6.0f * fminf(fmaxf(d[nt][q], 0.0f), 1.0f));
                    *reinterpret_cast<half4_t*>(a.out2 + (size_t)((b * a.hin + iy) * a.win + px) * a.cmid + ce0 + (nt + half) * 16 + g * 4) = t2;
                }
            }
        }
    }
    if constexpr (PAIR) {
        __syncthreads();   // both halves of every chunk buffer are written
        if (!havec) return;
    } else {
        // the wave's own LDS writes are ordered before its reads by the LDS queue; keep the compiler from moving the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---- depthwise 3x3 in fp32 (lane = output pixel x 8 channels), relu6, split, -> D as B fragments of the project GEMM
    half_t* const D = reinterpret_cast<half_t*>(a.ws);
    const float rcp_wout = 1.0f / (float)a.wout;
    const int nout = oh * a.wout;
#pragma unroll
    for (int j = half; j < MQW; j += (PAIR ? 2 : 1)) {
        if (j * 16 >= nout) break;               // (uniform)
        const int q = j * 16 + r16;
        const bool valid = q < nout;
        const int qc = valid ? q : 0;
        const int qy = (int)(((float)qc + 0.5f) * rcp_wout), qx = qc - qy * a.wout;
        const unsigned short* const ep = E + (qy * S * EW + qx * S) * HP2_ES + g * 8;
        wz_f32x2_t dd[4];
        dd[0] = __builtin_shufflevector(b0, b0, 0, 1);
        dd[1] = __builtin_shufflevector(b0, b0, 2, 3);
        dd[2] = __builtin_shufflevector(b1, b1, 0, 1);
        dd[3] = __builtin_shufflevector(b1, b1, 2, 3);
        wz_u32x4_t tq[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) tq[t] = *reinterpret_cast<const wz_u32x4_t*>(ep + ((t / 3) * EW + t % 3) * HP2_ES);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wz_f32x2_t x[4];
            wz_hp_unpack<false>(tq[t], x);
            wz_hp_fma8(dd, x, wt0[t], wt1[t]);
        }
        float v[8];
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) v[q8] = fminf(fmaxf(dd[q8 >> 1][q8 & 1], 0.0f), 6.0f);
        half8_t bh, bl;
        wz_hp_split(v, bh, bl);
        if (valid) {
            const int P = (b * a.hout + oy0 + qy) * a.wout + qx;
            half_t* const dst = D + ((size_t)(P >> 4) * nk32 + ps) * 1024 + (g * 16 + (P & 15)) * 8;
            *reinterpret_cast<half8_t*>(dst) = bh;
            *reinterpret_cast<half8_t*>(dst + 512) = bl;
        }
    }
}

// out[pixel][n] = sum_k Wp[n][k] D[pixel][k] (+ bias, + residual): MT x NT tiles per workgroup, K over the NW waves (CPW chunks each at most).
// Workgroups are dealt round-robin over the 8 XCDs (id % 8): the m-groups are dealt the same way, so that an XCD's L2 holds a
// disjoint eighth of D and every m-group's NT-groups meet in one L2.
template <int MT, int NT, int NW, int CPW>
__global__ __launch_bounds__(NW * 64, 1) void wz_k_hp2_proj(const WzMbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wz_hp2_smem[];
    WZ_LANE_STAMP(a.dbg);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r16 = lane & 15, g = lane >> 4;
    const int nk32 = a.cmid_pad >> 5;
    const int mtiles = (a.M + 15) >> 4;
    const int mgs = (mtiles + MT - 1) / MT, ngs = (a.n_pad >> 4) / NT;
    const int id = (int)blockIdx.x;
    const int xcd = id & 7, rest = id >> 3;
    const int ng = rest % ngs, mg = (rest / ngs) * 8 + xcd;
    if (mg >= mgs) return;                                // (whole workgroup, in front of every barrier)
    const half_t* const D = reinterpret_cast<const half_t*>(a.ws);

    half8_t dh[CPW][MT], dl[CPW][MT], wh[CPW][NT], wl[CPW][NT];
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const int ps = min(wave + k * NW, nk32 - 1);      // beyond the block's chunks: clamped, the products are skipped below
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int tm = min(mg * MT + mt, mtiles - 1);
            const half_t* const src = D + ((size_t)tm * nk32 + ps) * 1024 + lane * 8;
            dh[k][mt] = *reinterpret_cast<const half8_t*>(src);
            dl[k][mt] = *reinterpret_cast<const half8_t*>(src + 512);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const size_t off = ((size_t)((ng * NT + nt) * a.kc + ps) * 64 + lane) * 8;
            wh[k][nt] = *reinterpret_cast<const half8_t*>(a.wp + off);
            wl[k][nt] = *reinterpret_cast<const half8_t*>(a.wp_lo + off);
        }
    }
    // the epilogue's bias and residual (wave t < MT * NT finishes tile t): on their way while the chunks are multiplied
    const int ostride = a.hp_out ? 2 * a.cout : a.cout;
    const bool fin_wave = wave < MT * NT;
    const int fmt = fin_wave ? wave / NT : 0, fnt = fin_wave ? wave - fmt * NT : 0;
    const int P = (mg * MT + fmt) * 16 + r16;
    const int n4 = (ng * NT + fnt) * 16 + g * 4;
    const bool on = fin_wave && P < a.M && n4 < a.cout;
    float4_t sbias = {0.f, 0.f, 0.f, 0.f};
    half4_t srh = {0, 0, 0, 0}, srl = {0, 0, 0, 0};
    if (on) {
        sbias = *reinterpret_cast<const float4_t*>(a.bp + n4);
        if (a.res) {
            const half_t* rp = a.res + (size_t)P * (2 * a.cout) + n4;
            srh = *reinterpret_cast<const half4_t*>(rp);
            srl = *reinterpret_cast<const half4_t*>(rp + a.cout);
        }
    }
    float4_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        if (wave + k * NW >= nk32) break;                 // (uniform)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = WZ_HP_MFMA(wl[k][nt], dh[k][mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = WZ_HP_MFMA(wh[k][nt], dl[k][mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = WZ_HP_MFMA(wh[k][nt], dh[k][mt], acc[mt][nt]);
    }
    // the waves' partial tiles meet in LDS; tile t is summed by wave t in the order wave 0 .. NW - 1 (fixed: bit-identical run to run)
    float* const red = reinterpret_cast<float*>(wz_hp2_smem);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<float4_t*>(red + ((size_t)(wave * MT * NT + mt * NT + nt) * 64 + lane) * 4) = acc[mt][nt];
    __syncthreads();
    if (!on) return;
    float4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const float4_t pz = *reinterpret_cast<const float4_t*>(red + ((size_t)(w * MT * NT + wave) * 64 + lane) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += pz[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += sbias[q];
    if (a.res) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += (float)srh[q] + (float)srl[q];
    }
    half4_t oh, ol;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        oh[q] = (half_t)v[q];
        ol[q] = (half_t)(v[q] - (float)oh[q]);
    }
    half_t* const dst = a.out + (size_t)P * ostride + n4;
    *reinterpret_cast<half4_t*>(dst) = oh;
    if (a.hp_out) *reinterpret_cast<half4_t*>(dst + a.cout) = a.hp_out == 2 ? oh : ol;   // 2: [hi | hi] for a consumer whose WEIGHTS are split
}

// ---------------------------------------------------------------------------------------------
static int wz_hp2_env(const char* name, int dflt) {
    const char* e = wz_dev_getenv(name);
    return (e && e[0] && atoi(e) >= 0) ? atoi(e) : dflt;
}

template <int S, int KCI, int NW, int OHR, int MPW, int MQW, bool TAP, bool PAIR = false>
static int wz_hp2_launch_a(const WzMbArgs& a, int n, hipStream_t s, bool prepare) {
    auto k = wz_k_hp2_expdw<S, KCI, NW, OHR, MPW, MQW, TAP, PAIR>;
    constexpr int CPWG = PAIR ? NW / 2 : NW;
    const int EW = (a.wout - 1) * S + 3, EH = (OHR - 1) * S + 3;
    const size_t ebytes = ((size_t)EH * EW * HP2_ES * 2 + 15) & ~(size_t)15;
    const size_t lds = (size_t)MPW * KCI * 2 * 1024 + (size_t)CPWG * ebytes;
    // every band's in-frame input pixels fit MPW tiles, its outputs MQW tiles
    const int nr = (a.hout + OHR - 1) / OHR;
    for (int r = 0; r < nr; ++r) {
        const int oy0 = r * OHR, oh = a.hout - oy0 < OHR ? a.hout - oy0 : OHR;
        const int ey0 = oy0 * S - a.pad_t;
        const int iy0 = ey0 > 0 ? ey0 : 0, iy1 = ey0 + (oh - 1) * S + 3 < a.hin ? ey0 + (oh - 1) * S + 3 : a.hin;
        if ((iy1 - iy0) * a.win > MPW * 16 || oh * a.wout > MQW * 16) return -1;
    }
    if (a.pad_t < 0 || a.pad_t > 1 || a.pad_l < 0 || a.pad_l > 1 || a.win + a.pad_l > EW) return -1;   // the chunk buffer's row holds every input column + padding
    if (TAP && (nr * OHR * S < a.hin || S - 1 > 2 - a.pad_t)) return -1;                                       // the bands' owned rows cover the input map
    if (lds > 160 * 1024) return -1;
    if (prepare) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return 0;
    }
    const int nk32 = a.cmid_pad >> 5, cgw = (nk32 + CPWG - 1) / CPWG;
    WzMbArgs b = a;
    b.nb = n;
    WZ_LAUNCH(k, dim3(n * nr * cgw), dim3(NW * 64), lds, s, b);
    return 1;
}

template <int MT, int NT, int NW, int CPW>
static int wz_hp2_launch_b(const WzMbArgs& a, int n, hipStream_t s, bool prepare) {
    auto k = wz_k_hp2_proj<MT, NT, NW, CPW>;
    const int nk32 = a.cmid_pad >> 5;
    if ((a.n_pad >> 4) % NT || (nk32 + NW - 1) / NW > CPW) return -1;
    const size_t lds = (size_t)NW * MT * NT * 1024;
    if (prepare) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return 0;
    }
    const int mtiles = (n * a.hout * a.wout + 15) >> 4, mgs = (mtiles + MT - 1) / MT, ngs = (a.n_pad >> 4) / NT;
    WzMbArgs b = a;
    b.nb = n;
    b.M = n * a.hout * a.wout;
    b.dbg = a.dbg2;
    WZ_LAUNCH(k, dim3(8 * ((mgs + 7) / 8) * ngs), dim3(NW * 64), lds, s, b);
    return 1;
}

// bytes of D (the depthwise output as project fragments) for n frames of this block
static size_t wz_hp2_d_bytes(const WzMbArgs& a, int n) {
    return (size_t)((n * a.hout * a.wout + 15) >> 4) * (size_t)(a.cmid_pad >> 5) * 2048;
}

// 1: this block takes the two-launch form -- every 10x10 split block with the linear chunk buffer does, at every batch size (from one frame
// up it is the faster form: 33 against 41 us for the four blocks at batch 1, 42 against 65 at batch 8, profiles/r06_hp2_by_batch_size.txt);
// then wz_launch_mbconv_hp2 enqueues both launches, or one of them (`phase`)
int wz_mbconv_hp2_applies(const WzMbArgs& a, int n) {
    if (!a.hp || a.qenc || a.stem || a.cin0 == 0 || a.wout > 10 || a.hout > 10 || !a.ws || !a.we_lo || !a.wp_lo) return 0;
    if (a.nmid_pad != a.cmid_pad || (a.cmid_pad & 31) || a.kc != (a.cmid_pad >> 5) || a.cmid != a.cmid_pad || (a.cin0 & 31) || a.kc0 * 32 != a.cin0) return 0;
    if (wz_hp2_d_bytes(a, n) > (a.ws_bytes >> 1)) return 0;
    if (a.stride == 1 && a.kc0 == 5 && !a.has_out2) return 1;
    if (a.stride == 2 && a.kc0 == 3) return 1;
    return 0;
}

// phase 0: both launches; 1: launch A only; 2: launch B only (the engine's stage timer brackets them separately).  prepare: kernel attributes.
int wz_launch_mbconv_hp2(const WzMbArgs& a, int n, hipStream_t s, bool prepare, int phase) {
    static const int nw1 = wz_hp2_env("WZ_HP2_NW1", 4), nw2 = wz_hp2_env("WZ_HP2_NW2", 4);
    static const int mt = wz_hp2_env("WZ_HP2_MT", 1);
    // Two waves per chunk (PAIR: twice the workgroups, half the serial chain per wave -- launch A 7.0 -> 5.2 us at batch 8, 6.1 -> 3.8 us at batch 1) when the
    // batch runs ALONE on the chip (WzMbArgs::lone: every other lane idle, run_batch) and the one-wave-per-chunk launch leaves at least half of the CUs empty
    // (<= 128 workgroups of four waves: batch <= 8 on the 10x10 maps, <= 5 for block 13's bands), or when it leaves three quarters of them empty whatever the
    // other lanes do (<= 64).  With four lanes in flight at batch 8 the pairs cost 1 % of the throughput (49.6 against 50.2 k frames/s: twice the workgroups
    // hold the CUs), hence the condition.  Both shapes compute bit-identical tensors (a chunk's two 16-channel tiles are independent MFMA chains either way),
    // so a batch's rows do not depend on which one ran (tests/test_gpu_variants.py: graph against kernel-by-kernel launches).
    static const int pair_wgs = wz_hp2_env("WZ_HP2_PAIR_WGS", 128), pair_always_wgs = wz_hp2_env("WZ_HP2_PAIR_ALWAYS_WGS", 64);
    const int nk32_ = a.cmid_pad >> 5;   // pixel tiles per workgroup of launch B: 1 (122 registers: two workgroups per CU) or 2
    int ra = 1, rb = 1;
    if (prepare || phase != 2) {
        if (a.stride == 1 && a.kc0 == 5) {
            if (prepare) {
                (void)wz_hp2_launch_a<1, 5, 3, 5, 4, 4, false>(a, n, s, true);
                (void)wz_hp2_launch_a<1, 5, 4, 5, 4, 4, false>(a, n, s, true);
                (void)wz_hp2_launch_a<1, 5, 4, 5, 4, 4, false, true>(a, n, s, true);
                (void)wz_hp2_launch_a<1, 5, 6, 5, 4, 4, false>(a, n, s, true);
                ra = wz_hp2_launch_a<1, 5, 5, 5, 4, 4, false>(a, n, s, true);
            } else
                if (nw1 == 4 && n * 2 * ((nk32_ + 3) / 4) <= (a.lone ? pair_wgs : pair_always_wgs)) ra = wz_hp2_launch_a<1, 5, 4, 5, 4, 4, false, true>(a, n, s, false);
                else
                ra = nw1 == 3 ? wz_hp2_launch_a<1, 5, 3, 5, 4, 4, false>(a, n, s, false)
                   : nw1 == 5 ? wz_hp2_launch_a<1, 5, 5, 5, 4, 4, false>(a, n, s, false)
                   : nw1 == 6 ? wz_hp2_launch_a<1, 5, 6, 5, 4, 4, false>(a, n, s, false)
                              : wz_hp2_launch_a<1, 5, 4, 5, 4, 4, false>(a, n, s, false);
        } else if (a.stride == 2 && a.kc0 == 3) {
            if (prepare) {
                (void)wz_hp2_launch_a<2, 3, 3, 2, 6, 2, true>(a, n, s, true);
                (void)wz_hp2_launch_a<2, 3, 4, 2, 6, 2, true>(a, n, s, true);
                (void)wz_hp2_launch_a<2, 3, 4, 2, 6, 2, true, true>(a, n, s, true);
                ra = wz_hp2_launch_a<2, 3, 6, 2, 6, 2, true>(a, n, s, true);
            } else
                if (nw2 == 4 && n * 5 * ((nk32_ + 3) / 4) <= (a.lone ? pair_wgs : pair_always_wgs)) ra = wz_hp2_launch_a<2, 3, 4, 2, 6, 2, true, true>(a, n, s, false);
                else
                ra = nw2 == 3 ? wz_hp2_launch_a<2, 3, 3, 2, 6, 2, true>(a, n, s, false)
                   : nw2 == 6 ? wz_hp2_launch_a<2, 3, 6, 2, 6, 2, true>(a, n, s, false)
                              : wz_hp2_launch_a<2, 3, 4, 2, 6, 2, true>(a, n, s, false);
        } else
            return -1;
        if (ra < 0) return -1;
    }
    if (prepare || phase != 1) {
        const bool c3 = ((a.cmid_pad >> 5) + 7) / 8 <= 3;   // chunks per wave: 3 (block 13: 18 chunks) or 4 (30)
        if (prepare) {
            (void)wz_hp2_launch_b<1, 2, 8, 4>(a, n, s, true);
            (void)wz_hp2_launch_b<1, 2, 8, 3>(a, n, s, true);
            (void)wz_hp2_launch_b<2, 2, 8, 3>(a, n, s, true);
            rb = wz_hp2_launch_b<2, 2, 8, 4>(a, n, s, true);
        } else if (c3)
            rb = mt == 1 ? wz_hp2_launch_b<1, 2, 8, 3>(a, n, s, false) : wz_hp2_launch_b<2, 2, 8, 3>(a, n, s, false);
        else
            rb = mt == 1 ? wz_hp2_launch_b<1, 2, 8, 4>(a, n, s, false) : wz_hp2_launch_b<2, 2, 8, 4>(a, n, s, false);
        if (rb < 0) return -1;
    }
    return prepare ? 0 : 1;
}
