// Inverted-residual block with SPLIT matrix operands: the blocks of the `-p 16` program that decide whether the
// detector's scores stay within 1e-3 of the fp32 reference (`watsor/detection/tensorflow_cpu.py:79-90` reads fp32
// scores; north star: "box scores within 1e-3 of the CPU reference").
//
// Plain fp16 storage misses that bar by 3x on this network, and tools/err_budget.py shows where the error comes
// from: the stem and blocks 0 .. 12 (the 150x150 ... 19x19 maps), every operand about equally -- the weights, the
// block inputs, the expanded tensor that feeds the depthwise conv, and its output that feeds the project conv.
// Blocks 13 .. 16, Conv_1, the extras and the heads (85 % of the FLOPs) contribute 2e-4 together and stay plain.
// So these blocks keep fp16 MFMA but carry ~22 significand bits through every operand:
//
//   * a value v travels as the pair hi = RN16(v), lo = RN16(v - hi); a product W.x is evaluated as
//     Wlo.xhi + Whi.xlo + Whi.xhi on v_mfma_f32_16x16x32_f16 (fp32 accumulate; Wlo.xlo ~ 2^-22 is dropped):
//     3 MFMAs for one -- affordable because these blocks are VALU / LDS bound, not matrix bound;
//   * weights are split offline (watsor_amd/engine.py), block inputs / outputs / residuals are stored in HBM as
//     pair tensors (c hi halves, then c lo halves per pixel), the depthwise output is split in registers;
//   * the expanded tensor is a relu6 output, i.e. in [0, 6]: it goes to LDS as unorm16 of v / 6
//     (v_cvt_pknorm_u16_f32: clamp + scale + pack in one instruction, absolute step 9e-5 -- 10 to 40 times finer
//     than fp16 above 0.5, and the same 2 bytes), the 1/6 folded into the expand weights and the 6/65535 into the
//     depthwise weights, which are fp32 and come straight from L2 into registers;
//   * depthwise 3x3 in fp32 on v_cvt_f32_u32 (SDWA word select) + v_fma_f32.
//
// Two shapes of the same body (the structure of k_mbconv_wave.hip / k_mbconv_cs.hip):
//   CS = false  one WAVEFRONT per pixel tile, 4 independent waves per workgroup (150x150 ... 38x38 maps);
//               at stride 1 the lane's two output pixels are vertically adjacent, so the 4 x 3 taps under them
//               are read once for both (12 LDS reads instead of 18);
//   CS = true   8 waves on ONE 4x4 tile, the expanded channels dealt out over the waves, partial accumulators
//               summed through LDS in wave order (19x19 maps).
#include "wz_common.h"

#include "k_hp_ops.h"

#define HP_CS_WAVES 8
#ifndef WZ_HP_STAMPS
#define WZ_HP_STAMPS 0   // 1: cycle counts of the first workgroup's wave 0 into WzMbArgs::dbg (tools/hp_probe.py)
#endif
#ifndef WZ_HP_RR3
#define WZ_HP_RR3 1      // 168-register builds, 4 x 8 tiles: halo rows per tap request group (1, 2 or 4)
#endif
#ifndef WZ_HP_ROWS3
#define WZ_HP_ROWS3 1    // 168-register builds, 4 x 4 tiles: tap rows per request group (1 or 3)
#endif
#ifndef WZ_HP_STAMP_LAST
#define WZ_HP_STAMP_LAST 0   // 1: ... of the LAST workgroup instead (one that starts on a CU another workgroup has run on)
#endif

// MPW: halo m-tiles (16 pixels) per wave, MQW: output m-tiles per wave, KCI: 32-channel K chunks of the expand conv,
// NTO: 16-column tiles of the project output.  STEM: the "expand" stage is the stem convolution gathered from the
// 300x300 input pair tensor (8 halves per pixel: r g b 0 hi | r g b 0 lo), as in k_mbconv_wave.hip.
// NW: wavefronts per workgroup (CS: they share one tile and deal its 32-channel chunks out among themselves).
// OCC: wavefronts per SIMD the kernel is compiled for (2: 256 registers, everything prefetched into them; 3 / 4: 168 / 128
// registers -- depthwise weights read from LDS where they are used, taps requested a row at a time -- for the blocks
// whose waves spend their time waiting rather than issuing).
// ONEPASS (CS only): the workgroup has at least as many waves as the block has chunks, every wave walks at most ONE -- the
// halo fragments and the expand weights are then dead after the expand stage and the kernel fits 3 waves per SIMD.
// SH (CS only): the halo fragments are fetched ONCE per workgroup -- each wave loads its share of the MPW x KCI x 2 fragments,
// they meet in LDS, every wave reads all of them back -- instead of once per wave (8 or 12 times the same 13 .. 18 KiB through
// the vector memory path of one CU: the prologue of the 19x19 blocks was bound by exactly that, profiles/r02zq_*).
// QE: the chunk buffer holds the 16-bit FLOAT form of v / 6 (above) instead of unorm16 of v / 6 (the ROBUST program, WzMbArgs::qenc): a
// relative step -- what channels of very different scale need.  Round 3 kept square roots there (one v_sqrt_f32
// per stored value, one multiply per tap): coarser for small channels and slower to encode.
// LEAN (CS + SH, one output m-tile; MQW = 1): the shapes behind the 19x19 maps (block 13: 96 -> 576 -> 160 at stride 2; blocks 14 .. 16:
// 160 -> 960 -> 160 / 320 on 10x10) do not fit 256 registers the way the others are written -- MPW x KCI x 2 halo fragments alone are
// 144 / 120 of them -- so the halo fragments of ONE m-tile at a time come back from LDS inside the expand stage, and a workgroup
// finishes NTO of the block's n-tiles: blockIdx.x % nsplit picks which (the expand and depthwise stages are repeated per group:
// these launches have a third of a chip's worth of workgroups, the repeat costs no time and halves the accumulators).
// (Rounds 4 / 5 also dealt a 10x10 block's chunks over several WORKGROUPS per tile -- channel groups with a ticketed last-arriver sum -- and stored
// block 13's second output from here; since round 6 the 10x10 blocks of the robust program run as two GEMM-shaped launches at every batch size
// (k_mbconv_hp2.hip: faster from one frame up, profiles/r06_hp2_by_batch_size.txt) and both mechanisms left this kernel.)
template <int NW, bool CS, bool STEM, int MPW, int MQW, int KCI, int NTO, int OCC = 2, bool ONEPASS = false, bool SH = false,
          bool QE = false, bool LEAN = false>
__global__ __launch_bounds__(NW * 64, OCC) void wz_k_mbconv_hp(const WzMbArgs a) {
    static_assert(!LEAN || (CS && SH && !ONEPASS && !STEM), "lean: chunk-split, shared halo");
    static_assert(!LEAN || MQW == 1, "lean builds: 4 x 4 tiles (the 4 x 8 lean builds of round 5 lost their A/B: profiles/r05_blocks_13_16_tile_4x8.txt)");
    extern __shared__ __attribute__((aligned(16))) unsigned char wz_hp_smem[];
    WZ_LANE_STAMP(a.dbg);
    const long long t_entry = WZ_HP_STAMPS ? __builtin_readcyclecounter() : 0;
    constexpr bool LDSW = CS || OCC > 2;                 // depthwise weights staged in LDS (else: registers, from L2)
    constexpr bool PRE = OCC <= 2;                       // all taps of an output requested before the first is used
    constexpr int ES = 40;                               // unorm16 per row of the chunk buffer: 32 channels + 8 of padding
    constexpr int EBYTES = MPW * 16 * ES * 2;
    // rounds of the accumulators' trip through LDS
    constexpr int RROUNDS = (LEAN && OCC >= 4 && NTO % 2 == 0 && NTO >= 6) ? 2 : 1;
    constexpr int RED_BYTES = CS ? NW * (MQW * NTO / RROUNDS) * 1024 : 0;
    constexpr int REGION = (NW * EBYTES > RED_BYTES) ? NW * EBYTES : RED_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    unsigned short* const E = reinterpret_cast<unsigned short*>(wz_hp_smem + wave * EBYTES);
    float* const bd_l = reinterpret_cast<float*>(wz_hp_smem + REGION);   // [cmid_pad] depthwise bias
    float* const be_l = bd_l + a.cmid_pad;                               // [cmid_pad] expand bias (already / 6)
    const float* const wd32 = reinterpret_cast<const float*>(a.wd);      // [9][cmid_pad], already * 6 / 65535
    // CS: 8 waves share a CU and 256 registers each -- the depthwise weights are staged in LDS once and read where they
    // are used (broadcast reads); !CS: they come from L2 into registers at the top of a pass (LDS is the busy unit there)
    float* const wd_l = be_l + a.cmid_pad;                               // CS only: [9][cmid_pad]

    // Order of the prologue: what has to travel furthest goes first.  The halo pixels and the first expand fragments are requested
    // (into registers) before the biases and depthwise weights are staged into LDS, whose loads are all issued before the first
    // of them is stored -- in-kernel stamps (tools/hp_probe.py, profiles/r02zp_*) had the old order (staging loop by loop, then
    // the index arithmetic with its integer divisions, then the halo) at 4 000 .. 9 500 cycles before the last load was even
    // issued, 30 .. 40 % of a 19x19 block's launch.

    // ---- the tile of this wave (CS: of this workgroup); wave-uniform, kept in scalar registers
    const int tiles = a.tiles_x * a.tiles_y;
    const int ngrp = LEAN ? a.nsplit : 1;                 // workgroups per tile, each with NTO of the n-tiles
    const int bidx = (int)blockIdx.x;                     // (tile, n-group)
    const int nt0 = LEAN ? (bidx % ngrp) * NTO : 0;
    const int wt = LEAN ? bidx / ngrp : CS ? (int)blockIdx.x : (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave);
    const bool live = wt < tiles * a.nb;                  // wave-uniform; dead waves still take the barrier below
    const int wtc = live ? wt : 0;
    const int b = wtc / tiles, t = wtc - b * tiles;
    const int tyi = t / a.tiles_x;
    const int oy0 = tyi * a.th, ox0 = (t - tyi * a.tiles_x) * a.tw;
    const int s = a.stride;
    const int hw_ = (a.tw - 1) * s + 3, hh_ = (a.th - 1) * s + 3;
    const int P = hh_ * hw_;
    const int iy_base = oy0 * s - a.pad_t, ix_base = ox0 * s - a.pad_l;
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    int hp0[MQW], opix[MQW];
#pragma unroll
    for (int j = 0; j < MQW; ++j) {
        int qy, qx;
        if constexpr (MQW == 2) {      // 4 x 8 tile at stride 1: output j of a lane sits right below output j - 1
            qx = r16 & 7;
            qy = ((r16 >> 3) << 1) + j;
        } else {                       // 4 x 4 tile
            qy = r16 >> 2;
            qx = r16 & 3;
        }
        hp0[j] = qy * s * hw_ + qx * s;
        const int oy = oy0 + qy, ox = ox0 + qx;
        opix[j] = (live && oy < a.hout && ox < a.wout) ? (b * a.hout + oy) * a.wout + ox : -1;
    }

    // ---- halo pixels of this lane: input channels as B fragments, hi and lo
    half8_t xh[MPW][KCI], xl[MPW][KCI];
    bool inimg[MPW];
    const float rcp_hw = 1.0f / (float)hw_;   // p < 96, hw_ <= 10: floor((p + 0.5) / hw_) is exact in fp32 (no integer division)
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int p = i * 16 + r16;
        const int hy = (int)(((float)p + 0.5f) * rcp_hw), hx = p - hy * hw_;
        const int iy = iy_base + hy, ix = ix_base + hx;
        const bool ok = live && p < P && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
        inimg[i] = ok;
        if constexpr (STEM) {
            // K order of the stem GEMM in this program (engine.py: stem_k_rows): k = tap*4 + c for taps 0 .. 7, i.e. lane group
            // g holds the 4-channel pixels of taps 2g and 2g+1 exactly as they lie in the input pair tensor, and the ninth
            // tap's three channels sit in the always-zero fourth-channel slots of taps 0, 1, 2 (k = 3, 7, 11: groups 0 and 1)
            auto tap = [&](int tq, bool want) -> half8_t {
                const int ky = tq / 3, kx = tq - ky * 3;
                const int sy = iy * 2 - a.spad_t + ky, sx = ix * 2 - a.spad_l + kx;   // stem: stride 2 on the input image
                const bool in = want && ok && sy >= 0 && sy < a.sin_h && sx >= 0 && sx < a.sin_w;
                const int cy = min(max(sy, 0), a.sin_h - 1), cx = min(max(sx, 0), a.sin_w - 1);
                const half8_t v = *reinterpret_cast<const half8_t*>(a.in + ((size_t)(b * a.sin_h + cy) * a.sin_w + cx) * 8);
                return in ? v : zero8;
            };
            const half8_t va = tap(2 * g, true), vb = tap(2 * g + 1, true), v8 = tap(8, g <= 1);
            const half_t z16 = (half_t)0.0f;
            const half_t e0h = g == 0 ? v8[0] : g == 1 ? v8[2] : z16, e1h = g == 0 ? v8[1] : z16;
            const half_t e0l = g == 0 ? v8[4] : g == 1 ? v8[6] : z16, e1l = g == 0 ? v8[5] : z16;
            xh[i][0] = (half8_t){va[0], va[1], va[2], e0h, vb[0], vb[1], vb[2], e1h};
            xl[i][0] = (half8_t){va[4], va[5], va[6], e0l, vb[4], vb[5], vb[6], e1l};
        } else if constexpr (!SH) {
            const half_t* src = a.in + ((size_t)(b * a.hin + (ok ? iy : 0)) * a.win + (ok ? ix : 0)) * (2 * a.cin0);
#pragma unroll
            for (int c = 0; c < KCI; ++c) {
                const int k0 = c * 32 + g * 8;
                const bool kok = ok && k0 < a.cin0;
                xh[i][c] = kok ? *reinterpret_cast<const half8_t*>(src + k0) : zero8;
                xl[i][c] = kok ? *reinterpret_cast<const half8_t*>(src + a.cin0 + k0) : zero8;
            }
        }
    }
    // SH: fragment f = (i * KCI + c) * 2 + (0: hi, 1: lo) is fetched by wave f % NW and parked at f KiB of the halo area
    constexpr int NFRAG = MPW * KCI * 2;
    unsigned char* const halo_l = wz_hp_smem + REGION + (size_t)a.cmid_pad * (LDSW ? 44 : 8);
    // (its LDS stores wait for the loads: they come behind the ISSUE of the staging loads and of the first expand weights below -- the
    //  prologue of these kernels used to take two memory round trips one after the other, `s_waitcnt vmcnt(0)` in front of the halo's
    //  ds_write and again in front of the staged biases', round 4)
    constexpr int PERW = SH ? (NFRAG + NW - 1) / NW : 1;
    half8_t part[PERW];
    if constexpr (SH) {
        static_assert(!STEM && CS, "shared halo: chunk-split kernels only");
#pragma unroll
        for (int k = 0; k < PERW; ++k) {
            const int f = wave + k * NW;     // wave-uniform
            part[k] = zero8;
            if (f < NFRAG) {
                const int i = f / (KCI * 2), c = (f >> 1) % KCI, lo = f & 1;
                const int p = i * 16 + r16;
                const int hy = (int)(((float)p + 0.5f) * rcp_hw), hx = p - hy * hw_;
                const int iy = iy_base + hy, ix = ix_base + hx;
                const int k0 = c * 32 + g * 8;
                const bool kok = live && p < P && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win && k0 < a.cin0;
                const half_t* src = a.in + ((size_t)(b * a.hin + (kok ? iy : 0)) * a.win + (kok ? ix : 0)) * (2 * a.cin0);
                if (kok) part[k] = *reinterpret_cast<const half8_t*>(src + (lo ? a.cin0 : 0) + k0);
            }
        }
    }
    auto park_halo = [&]() {
        if constexpr (SH) {
#pragma unroll
            for (int k = 0; k < PERW; ++k) {
                const int f = wave + k * NW;
                if (f < NFRAG) *reinterpret_cast<half8_t*>(halo_l + (size_t)f * 1024 + lane * 16) = part[k];
            }
        }
    };

    // a tile whose halo lies inside the frame needs no per-pixel masking of the expanded values (wave-uniform; rows of
    // the last m-tile beyond the halo are never read by the depthwise stage, whatever they hold)
    bool interior = true;
#pragma unroll
    for (int i = 0; i < MPW; ++i) interior = interior && __builtin_amdgcn_ballot_w64(!inimg[i] && i * 16 + r16 < P) == 0;

    float4_t acc[MQW][NTO];
#pragma unroll
    for (int j = 0; j < MQW; ++j)
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) acc[j][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int nk32 = a.cmid_pad >> 5;                     // 32-channel chunks in all
    const int ntiles_e = a.nmid_pad >> 4;
    constexpr int STEP = CS ? NW : 1;
    const int c_lo = 0, c_hi = nk32;                      // the workgroup's chunks: all of the block's
    const int ps0 = CS ? c_lo + wave : 0;
    half8_t wah[2][KCI], wal[2][KCI];
    auto load_wa = [&](int ps) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int tn = min(ps * 2 + nt, ntiles_e - 1);   // beyond the packed tiles: clamped, never used (see `have`)
            const size_t off = ((size_t)tn * a.kc0 * 64 + lane) * 8;
#pragma unroll
            for (int c = 0; c < KCI; ++c) {
                wah[nt][c] = *reinterpret_cast<const half8_t*>(a.we + off + (size_t)c * 512);
                wal[nt][c] = *reinterpret_cast<const half8_t*>(a.we_lo + off + (size_t)c * 512);
            }
        }
    };
    constexpr bool WA_AHEAD = !((LEAN || !CS) && OCC >= 4);   // the next pass's expand fragments requested a stage ahead (4 waves per SIMD: at the
                                                     // top of the pass instead -- the other waves cover the wait, the registers are not there)
    if (!SH && WA_AHEAD && ps0 < c_hi) load_wa(ps0);
    {   // biases (and, LDSW, depthwise weights) into LDS: every load of a thread in flight before its first store
        constexpr int NT = NW * 64;
        constexpr int WD_IT = LEAN ? (NW >= 8 ? 5 : 9) : NW >= 8 ? 3 : 8;   // 9 * cmid_pad / 4 float4s over NT threads: at most this many each (checked by the launcher)
        const int nb4 = a.cmid_pad >> 2;
        float4_t sw[WD_IT], sb, se;
        if constexpr (LDSW) {
#pragma unroll
            for (int k = 0; k < WD_IT; ++k) {
                const int i = (int)threadIdx.x + k * NT;
                sw[k] = i < 9 * nb4 ? *reinterpret_cast<const float4_t*>(wd32 + (size_t)i * 4) : (float4_t){0.f, 0.f, 0.f, 0.f};
            }
        }
        const int ib = (int)threadIdx.x;   // nb4 <= NT: one float4 of each bias per thread
        const bool hb = ib < nb4;
        sb = hb ? *reinterpret_cast<const float4_t*>(a.bd + ib * 4) : (float4_t){0.f, 0.f, 0.f, 0.f};
        se = (hb && ib * 4 < a.nmid_pad) ? *reinterpret_cast<const float4_t*>(a.be + ib * 4) : (float4_t){0.f, 0.f, 0.f, 0.f};
        if constexpr (SH) {   // shared halo: the first expand weights are requested LAST (nothing in front of the barrier waits for them) ...
            __builtin_amdgcn_sched_barrier(0);
            if (WA_AHEAD && ps0 < c_hi) load_wa(ps0);
            __builtin_amdgcn_sched_barrier(0);
            park_halo();      // ... and the halo fragments, requested first, are parked now
        }
        if constexpr (LDSW) {
#pragma unroll
            for (int k = 0; k < WD_IT; ++k) {
                const int i = (int)threadIdx.x + k * NT;
                if (i < 9 * nb4) *reinterpret_cast<float4_t*>(wd_l + i * 4) = sw[k];
            }
        }
        if (hb) {
            *reinterpret_cast<float4_t*>(bd_l + ib * 4) = sb;
            *reinterpret_cast<float4_t*>(be_l + ib * 4) = se;
        }
    }
    const long long t_issued = WZ_HP_STAMPS ? __builtin_readcyclecounter() : 0;
    __syncthreads();   // staged biases (SH: and halo fragments) visible; the only workgroup barrier in front of the loop
    if (!CS && !live) return;

    long long t_loop = 0, t_first = 0, cy_expand = 0, cy_dw = 0, cy_proj = 0;
    if (WZ_HP_STAMPS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (stamped build only: the halo and the first weights have landed)
        t_loop = __builtin_readcyclecounter();
    }

    for (int ps = ps0; ps < c_hi; ps += STEP) {
        const long long tc0 = WZ_HP_STAMPS ? __builtin_readcyclecounter() : 0;
        const int ce0 = ps * 32;
        const int coff = ce0 + g * 8;
        // operands of the later phases, in flight under the expand stage: project fragments, depthwise weights
        half8_t wph[NTO], wpl[NTO];
        auto load_wp = [&]() {
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                const size_t off = ((size_t)((nt0 + nt) * a.kc + ps) * 64 + lane) * 8;
                wph[nt] = *reinterpret_cast<const half8_t*>(a.wp + off);
                wpl[nt] = *reinterpret_cast<const half8_t*>(a.wp_lo + off);
            }
        };
        constexpr bool WP_LATE = (OCC > 2 && CS) || (LEAN && OCC >= 2) || OCC >= 4;   // 3 waves per SIMD: the project fragments are requested behind the expand stage
        if constexpr (!WP_LATE) load_wp();        // (they have the depthwise stage to land) instead of holding 8 x NTO registers through it
        if constexpr (!WA_AHEAD) load_wa(ps);
        float4_t wt0[9], wt1[9];   // (LDSW: unused, the weights are read from LDS where they are needed)
        if constexpr (!LDSW) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                wt0[tp] = *reinterpret_cast<const float4_t*>(wd32 + (size_t)tp * a.cmid_pad + coff);
                wt1[tp] = *reinterpret_cast<const float4_t*>(wd32 + (size_t)tp * a.cmid_pad + coff + 4);
            }
        }
        if constexpr (SH && !LEAN) {   // the halo fragments come back from LDS for every pass: they are dead behind the expand stage, which is
                              // what lets a multi-pass wave fit 168 registers (3 waves per SIMD, 12 per tile)
#pragma unroll
            for (int i = 0; i < MPW; ++i)
#pragma unroll
                for (int c = 0; c < KCI; ++c) {
                    xh[i][c] = *reinterpret_cast<const half8_t*>(halo_l + (size_t)((i * KCI + c) * 2) * 1024 + lane * 16);
                    xl[i][c] = *reinterpret_cast<const half8_t*>(halo_l + (size_t)((i * KCI + c) * 2 + 1) * 1024 + lane * 16);
                }
        }
        __builtin_amdgcn_sched_barrier(0);   // (the loads above stay above: they have the whole expand stage to land)
        // ---- expand: E[p][ce] = in-frame ? unorm16(clamp((sum_k X[p][k] We[k][ce] + be[ce]) / 6, 0, 1)) : 0
        // one 16 x 4 tile of expanded values of a lane -> the chunk buffer (QE: in the float form)
        auto put = [&](float4_t d, int i, int nt, bool keep) {
            wz_u32x2_t o;
            if constexpr (QE) {
                o = wz_hp_fenc4(d);
            } else {
                o[0] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pknorm_u16(d[0], d[1]));
                o[1] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pknorm_u16(d[2], d[3]));
            }
            if (!keep) o[0] = o[1] = 0u;
            *reinterpret_cast<wz_u32x2_t*>(E + (i * 16 + r16) * ES + nt * 16 + g * 4) = o;
        };
        if constexpr (LEAN) {
            // one pixel tile at a time: its KCI x 2 fragments come from the shared halo, feed both 16-channel tiles (two independent
            // accumulator chains, issued term by term) and are dead again
            float4_t bv[2];
            bool have[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                have[nt] = ce0 + nt * 16 < a.nmid_pad;
                bv[nt] = *reinterpret_cast<const float4_t*>(be_l + ce0 + nt * 16 + g * 4);
            }
#pragma unroll
            for (int i = 0; i < MPW; ++i) {
                half8_t fh[KCI], fl[KCI];
#pragma unroll
                for (int c = 0; c < KCI; ++c) {
                    fh[c] = *reinterpret_cast<const half8_t*>(halo_l + (size_t)((i * KCI + c) * 2) * 1024 + lane * 16);
                    fl[c] = *reinterpret_cast<const half8_t*>(halo_l + (size_t)((i * KCI + c) * 2 + 1) * 1024 + lane * 16);
                }
                float4_t d[2] = {bv[0], bv[1]};
#pragma unroll
                for (int c = 0; c < KCI; ++c)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) d[nt] = WZ_HP_MFMA(wal[nt][c], fh[c], d[nt]);
#pragma unroll
                for (int c = 0; c < KCI; ++c)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) d[nt] = WZ_HP_MFMA(wah[nt][c], fl[c], d[nt]);
#pragma unroll
                for (int c = 0; c < KCI; ++c)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) d[nt] = WZ_HP_MFMA(wah[nt][c], fh[c], d[nt]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) put(d[nt], i, nt, (interior || inimg[i]) && have[nt]);
            }
        } else {
        // per 16-channel tile the MPW pixel tiles are MPW independent accumulator chains: the three terms are issued term by
        // term across them, so that no MFMA waits for the one in front of it
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const bool have = ce0 + nt * 16 < a.nmid_pad;    // this 16-channel tile exists (wave-uniform)
            const float4_t bv = *reinterpret_cast<const float4_t*>(be_l + ce0 + nt * 16 + g * 4);
            float4_t d[MPW];
#pragma unroll
            for (int i = 0; i < MPW; ++i) {
                d[i] = bv;                        // the bias is the accumulators' initial value (the MFMA's C operand)
#pragma unroll
                for (int c = 0; c < KCI; ++c) d[i] = WZ_HP_MFMA(wal[nt][c], xh[i][c], d[i]);
            }
            __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler re-serialises the chains to save registers)
#pragma unroll
            for (int i = 0; i < MPW; ++i)
#pragma unroll
                for (int c = 0; c < KCI; ++c) d[i] = WZ_HP_MFMA(wah[nt][c], xl[i][c], d[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MPW; ++i)
#pragma unroll
                for (int c = 0; c < KCI; ++c) d[i] = WZ_HP_MFMA(wah[nt][c], xh[i][c], d[i]);
            __builtin_amdgcn_sched_barrier(0);
            if (interior && have) {
#pragma unroll
                for (int i = 0; i < MPW; ++i) put(d[i], i, nt, true);
            } else {
#pragma unroll
                for (int i = 0; i < MPW; ++i) put(d[i], i, nt, inimg[i] && have);
            }
        }
        }
        if constexpr (WP_LATE) load_wp();
        if constexpr (!ONEPASS && WA_AHEAD)
            if (ps + STEP < c_hi) load_wa(ps + STEP);   // next pass's expand fragments, in flight under the depthwise stage
        // the wave's own LDS writes are ordered before its reads by the LDS queue; keep the compiler from moving the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const long long tc1 = WZ_HP_STAMPS ? __builtin_readcyclecounter() : 0;   // (s_memtime waits lgkmcnt(0): the chunk's LDS stores have landed)
        // ---- depthwise (lane = output pixel x 8 channels), fp32
        const float4_t b0 = *reinterpret_cast<const float4_t*>(bd_l + coff);
        const float4_t b1 = *reinterpret_cast<const float4_t*>(bd_l + coff + 4);
        wz_f32x2_t dd[MQW][4];
#pragma unroll
        for (int j = 0; j < MQW; ++j) {
            if constexpr (QE) {   // float form: the tap sum starts at zero and is scaled back + biased at the end (wz_hp_dw_finish)
                dd[j][0] = dd[j][1] = dd[j][2] = dd[j][3] = (wz_f32x2_t){0.f, 0.f};
            } else {
                dd[j][0] = __builtin_shufflevector(b0, b0, 0, 1);
                dd[j][1] = __builtin_shufflevector(b0, b0, 2, 3);
                dd[j][2] = __builtin_shufflevector(b1, b1, 0, 1);
                dd[j][3] = __builtin_shufflevector(b1, b1, 2, 3);
            }
        }
        auto W0 = [&](int tp) -> float4_t {
            if constexpr (LDSW) return *reinterpret_cast<const float4_t*>(wd_l + tp * a.cmid_pad + coff);
            else return wt0[tp];
        };
        auto W1 = [&](int tp) -> float4_t {
            if constexpr (LDSW) return *reinterpret_cast<const float4_t*>(wd_l + tp * a.cmid_pad + coff + 4);
            else return wt1[tp];
        };
        if constexpr (MQW == 2) {
            // rows hp0 .. hp0 + 3 of the halo serve both outputs: row rr is tap row rr of output 0 and rr - 1 of output 1.
            // All 12 taps are requested before the first one is used: one LDS latency instead of twelve.
            const unsigned short* ep = E + hp0[0] * ES + g * 8;
            constexpr int RR = PRE ? 4 : WZ_HP_RR3;   // halo rows requested per group (3 waves per SIMD: WZ_HP_RR3)
#pragma unroll
            for (int r0 = 0; r0 < 4; r0 += RR) {
                wz_u32x4_t tq[RR * 3];
#pragma unroll
                for (int t = 0; t < RR * 3; ++t)
                    tq[t] = *reinterpret_cast<const wz_u32x4_t*>(ep + ((r0 + t / 3) * hw_ + t % 3) * ES);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < RR * 3; ++t) {
                    const int rr = r0 + t / 3, kx = t % 3;
                    wz_f32x2_t x[4];
                    wz_hp_unpack<QE>(tq[t], x);
                    if (rr < 3) wz_hp_fma8(dd[0], x, W0(rr * 3 + kx), W1(rr * 3 + kx));
                    if (rr > 0) wz_hp_fma8(dd[1], x, W0((rr - 1) * 3 + kx), W1((rr - 1) * 3 + kx));
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < MQW; ++j) {
                const unsigned short* ep = E + hp0[j] * ES + g * 8;
                // the taps are requested ahead of their use, all nine at once where the registers allow it
                constexpr int ROWS = KCI >= 3 ? 1 : PRE ? 3 : (CS ? 1 : WZ_HP_ROWS3);   // tap rows per request group
#pragma unroll
                for (int k0 = 0; k0 < 3; k0 += ROWS) {
                    wz_u32x4_t tq[ROWS * 3];
#pragma unroll
                    for (int t = 0; t < ROWS * 3; ++t)
                        tq[t] = *reinterpret_cast<const wz_u32x4_t*>(ep + ((k0 + t / 3) * hw_ + t % 3) * ES);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < ROWS * 3; ++t) {
                        const int tp = k0 * 3 + t;
                        wz_f32x2_t x[4];
                        wz_hp_unpack<QE>(tq[t], x);
                        wz_hp_fma8(dd[j], x, W0(tp), W1(tp));
                    }
                }
            }
        }
        const long long tc2 = WZ_HP_STAMPS ? __builtin_readcyclecounter() : 0;
        // ---- relu6, split, project: acc += Wlo.dhi + Whi.dlo + Whi.dhi
#pragma unroll
        for (int j = 0; j < MQW; ++j) {
            wz_hp_dw_finish<QE>(dd[j], b0, b1);
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = fminf(fmaxf(dd[j][r >> 1][r & 1], 0.0f), 6.0f);
            half8_t bh, bl;
            wz_hp_split(v, bh, bl);
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) acc[j][nt] = WZ_HP_MFMA(wpl[nt], bh, acc[j][nt]);
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) acc[j][nt] = WZ_HP_MFMA(wph[nt], bl, acc[j][nt]);
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) acc[j][nt] = WZ_HP_MFMA(wph[nt], bh, acc[j][nt]);
        }
        if (WZ_HP_STAMPS) {
            const long long tc3 = __builtin_readcyclecounter();
            cy_expand += tc1 - tc0; cy_dw += tc2 - tc1; cy_proj += tc3 - tc2;
            if (ps == ps0) t_first = tc3;
        }
        if constexpr (ONEPASS) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (the next pass's E stores stay behind these reads)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const long long t_chunks = WZ_HP_STAMPS ? __builtin_readcyclecounter() : 0;

    // ---- epilogue: + bias, + residual (a pair tensor like the input), store as a pair or as one half.  The bias and residual of
    // every output of a lane are requested before the first one is used (one memory latency, not one per output tile: the
    // residual blocks' epilogue took 5 500 cycles against 2 000 without a residual, profiles/r02zp_*).
    const int ostride = a.hp_out ? 2 * a.cout : a.cout;
    struct Side { float4_t bias; half4_t rh, rl; };
    auto side = [&](int op, int n4) {
        Side sd;
        sd.bias = *reinterpret_cast<const float4_t*>(a.bp + n4);
        if (a.res) {
            const half_t* rp = a.res + (size_t)op * (2 * a.cout) + n4;
            sd.rh = *reinterpret_cast<const half4_t*>(rp);
            sd.rl = *reinterpret_cast<const half4_t*>(rp + a.cout);
        } else {
            sd.rh = sd.rl = (half4_t){0, 0, 0, 0};
        }
        return sd;
    };
    auto finish = [&](float4_t v, const Side& sd, int op, int n4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += sd.bias[r];
        if (a.res) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)sd.rh[r] + (float)sd.rl[r];
        }
        half4_t oh, ol;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            oh[r] = (half_t)v[r];
            ol[r] = (half_t)(v[r] - (float)oh[r]);
        }
        half_t* const dst = a.out + (size_t)op * ostride + n4;
        *reinterpret_cast<half4_t*>(dst) = oh;
        if (a.hp_out) *reinterpret_cast<half4_t*>(dst + a.cout) = a.hp_out == 2 ? oh : ol;   // 2: [hi | hi] for a consumer whose WEIGHTS are split
    };

    if constexpr (!CS) {
        Side sd[MQW][NTO];
#pragma unroll
        for (int j = 0; j < MQW; ++j)
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                const int n4 = nt * 16 + g * 4;
                const bool on = opix[j] >= 0 && n4 < a.cout;
                sd[j][nt] = side(on ? opix[j] : 0, on ? n4 : 0);   // (an unused slot reads output 0: valid memory, result dropped)
            }
#pragma unroll
        for (int j = 0; j < MQW; ++j) {
            if (opix[j] < 0) continue;
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                const int n4 = nt * 16 + g * 4;
                if (n4 >= a.cout) continue;
                finish(acc[j][nt], sd[j][nt], opix[j], n4);
            }
        }
    } else if constexpr (RROUNDS > 1) {
        // lean builds at 4 waves per SIMD: the accumulators cross LDS in RROUNDS rounds of NTO / RROUNDS n-tiles, so that the area they
        // need (NW KiB per n-tile) stays under the chunk buffers and two workgroups fit the LDS of a CU; n-tile t of a round is
        // summed by wave t, in the order wave 0 .. NW - 1 as everywhere
        constexpr int NH = NTO / RROUNDS;
        static_assert(NTO % RROUNDS == 0 && NH <= NW && MQW == 1, "rounds of the lean reduction");
        Side sd[RROUNDS];
        const bool fin_wave = wave < NH;
#pragma unroll
        for (int r = 0; r < RROUNDS; ++r) {
            const int n4 = (nt0 + r * NH + (fin_wave ? wave : 0)) * 16 + g * 4;
            const bool on = fin_wave && opix[0] >= 0 && n4 < a.cout;
            sd[r] = side(on ? opix[0] : 0, on ? n4 : 0);
        }
        float* const red = reinterpret_cast<float*>(wz_hp_smem);
#pragma unroll
        for (int r = 0; r < RROUNDS; ++r) {
            __syncthreads();   // every wave is done with its chunk buffer / with the previous round's partials
#pragma unroll
            for (int t = 0; t < NH; ++t)
                *reinterpret_cast<float4_t*>(red + ((size_t)(wave * NH + t) * 64 + lane) * 4) = acc[0][r * NH + t];
            __syncthreads();
            if (fin_wave) {
                float4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const float4_t pz = *reinterpret_cast<const float4_t*>(red + ((size_t)(w * NH + wave) * 64 + lane) * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += pz[q];
                }
                const int n4 = (nt0 + r * NH + wave) * 16 + g * 4;
                if (opix[0] >= 0 && n4 < a.cout) finish(v, sd[r], opix[0], n4);
            }
        }
    } else {
        // the 8 waves' accumulators meet in LDS (over the chunk buffers); tile (j, nt) is summed by wave
        // (j*NTO + nt) % 8 in the order wave 0 .. 7 (deterministic)
        constexpr int PER = (MQW * NTO + NW - 1) / NW;   // tiles a wave finishes
        Side sd[PER];
        int sop[PER], sn4[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {   // their bias and residual: on their way while the accumulators cross LDS
            const int pr = wave + k * NW;
            const int j = pr / NTO, nt = pr - j * NTO;
            int op = -1;
#pragma unroll
            for (int jj = 0; jj < MQW; ++jj)
                if (jj == j) op = opix[jj];
            const int n4 = (nt0 + nt) * 16 + g * 4;
            const bool on = pr < MQW * NTO && op >= 0 && n4 < a.cout;
            sop[k] = on ? op : -1;
            sn4[k] = n4;
            sd[k] = side(on ? op : 0, on ? n4 : 0);
        }
        float* const red = reinterpret_cast<float*>(wz_hp_smem);
        __syncthreads();   // every wave is done with its chunk buffer
#pragma unroll
        for (int j = 0; j < MQW; ++j)
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt)
                *reinterpret_cast<float4_t*>(red + ((size_t)((wave * MQW + j) * NTO + nt) * 64 + lane) * 4) = acc[j][nt];
        __syncthreads();
        float4_t vs[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int pr = wave + k * NW;
            vs[k] = (float4_t){0.f, 0.f, 0.f, 0.f};
            if (pr >= MQW * NTO) break;
            const int j = pr / NTO, nt = pr - j * NTO;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float4_t pz = *reinterpret_cast<const float4_t*>(red + ((size_t)((w * MQW + j) * NTO + nt) * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) vs[k][r] += pz[r];
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int pr = wave + k * NW;
            if (pr >= MQW * NTO) break;
            if (sop[k] < 0) continue;
            finish(vs[k], sd[k], sop[k], sn4[k]);
        }
    }
    if (WZ_HP_STAMPS && a.dbg && blockIdx.x == (WZ_HP_STAMP_LAST ? gridDim.x - 1 : 0) && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t_end = __builtin_readcyclecounter();
        a.dbg[0] = (unsigned long long)(t_issued - t_entry);   // index math, staging loops, halo + weight loads issued
        a.dbg[1] = (unsigned long long)(t_loop - t_issued);    // ... landed, workgroup barrier
        a.dbg[2] = (unsigned long long)(t_first - t_loop);     // first chunk
        a.dbg[3] = (unsigned long long)(t_chunks - t_loop);    // all chunks of this wave
        a.dbg[4] = (unsigned long long)(t_end - t_chunks);     // (reduce through LDS,) epilogue, stores landed
        a.dbg[5] = (unsigned long long)((c_hi - ps0 + STEP - 1) / STEP);
        a.dbg[6] = (unsigned long long)cy_expand;   // per-phase sums over this wave's chunks (each stamp is an lgkmcnt(0): phases do not overlap here)
        a.dbg[7] = (unsigned long long)cy_dw;
        a.dbg[9] = (unsigned long long)cy_proj;
    }
}

// ---------------------------------------------------------------------------------------------
static int wz_hp_env(const char* name, int dflt) {
    const char* e = wz_dev_getenv(name);
    return (e && e[0] && atoi(e) >= 0) ? atoi(e) : dflt;
}

template <int NW, bool CS, bool STEM, int MPW, int MQW, int KCI, int NTO, int OCC = 2, bool ONEPASS = false, bool SH = false,
          bool QE = false, bool LEAN = false>
static int wz_hp_launch(WzMbArgs a, int n, hipStream_t s, bool prepare) {
    if (ONEPASS && (a.cmid_pad >> 5) > NW) return -1;
    if (QE != (a.qenc != 0)) return -1;
    if (a.has_out2) return -1;   // (a second output -- block 13's expanded tensor -- is stored by the two-launch form: k_mbconv_hp2.hip)
    a.nb = n;
    if (MQW == 2) { a.th = 4; a.tw = 8; } else { a.th = 4; a.tw = 4; }
    a.tiles_y = (a.hout + a.th - 1) / a.th;
    a.tiles_x = (a.wout + a.tw - 1) / a.tw;
    a.nsplit = LEAN ? (a.n_pad / 16 + NTO - 1) / NTO : 1;   // lean: workgroups per tile, NTO n-tiles each
    if (!LEAN && a.n_pad / 16 != NTO) return -1;
    if (LEAN && a.nsplit * NTO != a.n_pad / 16) return -1;
    constexpr int EB = MPW * 16 * 40 * 2;
    constexpr int RED = CS ? NW * (MQW * NTO / ((LEAN && OCC >= 4 && NTO % 2 == 0 && NTO >= 6) ? 2 : 1)) * 1024 : 0;
    const size_t region = (size_t)(NW * EB > RED ? NW * EB : RED);
    const size_t lds = region + (size_t)a.cmid_pad * ((CS || OCC > 2) ? 8 + 36 : 8) + (SH ? (size_t)MPW * KCI * 2 * 1024 : 0);
    if ((a.cmid_pad >> 2) > NW * 64 || 9 * (a.cmid_pad >> 2) > (LEAN ? (NW >= 8 ? 5 : 9) : NW >= 8 ? 3 : 8) * NW * 64) return -1;   // the staging code's fixed trip counts
    auto k = wz_k_mbconv_hp<NW, CS, STEM, MPW, MQW, KCI, NTO, OCC, ONEPASS, SH, QE, LEAN>;
    if (prepare) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return lds <= 160 * 1024 ? 0 : -1;
    }
    const int tiles = a.tiles_x * a.tiles_y * n;
    WZ_LAUNCH(k, dim3(CS ? tiles * a.nsplit : (tiles + 3) / 4), dim3(NW * 64), lds, s, a);
    return 1;
}

// The ROBUST program (WzMbArgs::qenc, `python -m watsor_amd.engine --robust`): all 17 blocks on this kernel with the float-form chunk
// buffer, one launch shape per block shape -- the throughput defaults of the dispatcher below, plus the lean builds for blocks 13 .. 16.
static int wz_launch_mbconv_hp_q(WzMbArgs a, int n, hipStream_t s, bool prepare) {
    const int nto = a.n_pad / 16;
    a.nsplit = 1;
    if (a.nmid_pad != a.cmid_pad || (a.cmid_pad & 31) || a.kc != (a.cmid_pad >> 5) || !a.we_lo || !a.wp_lo) return -1;
    // One or two frames per batch: 4 x 4 tiles instead of 4 x 8 on the stride-1 blocks of the 75x75 and 38x38 maps (block 2; blocks 4 / 5), whose 4 x 8 grids are
    // 48 - 95 / 50 - 100 workgroups then.  Twice the waves, 36 halo pixels each instead of 60, the same arithmetic in the same order for every output pixel
    // (bit-identical tensors: a tile shape only decides which lane holds a pixel).  Batch 1: block 2 10.4 -> 7.4 us, blocks 4 / 5 5.0 / 4.5 -> 3.1 / 2.9 us, a single
    // frame's p50 0.269 -> 0.262 ms and +2.5 % frames/s from four lanes; batch 2 the same; from three frames up the 20 % more halo work costs what the fuller grid
    // gains (batch 4: p50 -2 us, -0.6 % frames/s; batch 6: -2 % frames/s), and block 0's 150x150 grid is large enough at batch 1 (4.9 -> 5.9 us with 4 x 4):
    // profiles/r06_small_batch_tiles_4x4.txt.  WZ_HP_TILES44=0 / 1: never / always (development library; read per launch).
    const int tiles44 = wz_hp_env("WZ_HP_TILES44", 2);
    auto small44 = [&]() { return tiles44 == 1 || (tiles44 == 2 && n <= 2); };
    if (a.stem) {
        if (!(a.kc0 == 1 && nto == 2 && a.stride == 1)) return -1;
        return wz_hp_launch<4, false, true, 4, 2, 1, 2, 3, false, false, true>(a, n, s, prepare);
    }
    if (a.cin0 == 0) return -1;
    if (a.wout > 19 && a.kc0 == 1 && nto == 2) {
        const int nk32 = a.cmid_pad >> 5;
        // One or two frames per batch, blocks 1 .. 3 (150 -> 75 stride 2, 75x75 stride 1, 75 -> 38 stride 2): the chunks of a 4 x 4 tile dealt over the waves of a
        // workgroup -- one chunk per wave, halo fetched once and shared through LDS, partial accumulators summed through LDS in wave order: the shape of the
        // 38x38 / 19x19 blocks -- instead of one wave per tile walking 3 - 5 chunks: 3 - 5 times the waves on grids of 25 - 95 workgroups.  Batch 1: blocks 1 / 2 / 3
        // 6.4 / 7.2 / 8.5 -> 4.3 / 3.9 / 3.7 us, a single frame's p50 0.263 -> 0.254 ms, +3 % frames/s from four lanes; batch 2: p50 0.272 -> 0.264 ms.  From
        // three frames up the repeated halo work and the accumulators' trip through LDS cost more CU time than the chip has to spare (batch 3: p50 -4 us, -5 %
        // frames/s; batch 4: -8 %): profiles/r06_small_batch_chunk_split.txt.  The sum over a tile's chunks is taken in another order than at batch >= 3 (fp32
        // partials in wave order instead of one running accumulator): tensors agree to fp32 rounding, not bit for bit -- as batch sizes differ elsewhere
        // (split-K choices follow the pixel count).  WZ_HP_SMALL_CS=0 / 1: never / always (development library; read per launch).
        const int small_cs = wz_hp_env("WZ_HP_SMALL_CS", 2);
        const bool scs = small_cs == 1 || (small_cs == 2 && n <= 2);
        if (a.stride == 2 && nk32 == 3) {
            if (prepare && wz_hp_launch<3, true, false, 6, 1, 1, 2, 3, true, true, true>(a, n, s, true) < 0) return -1;
            if (!prepare && scs) return wz_hp_launch<3, true, false, 6, 1, 1, 2, 3, true, true, true>(a, n, s, false);
        }
        if (a.stride == 2 && nk32 == 5) {
            if (prepare && wz_hp_launch<5, true, false, 6, 1, 1, 2, 3, true, true, true>(a, n, s, true) < 0) return -1;
            if (!prepare && scs) return wz_hp_launch<5, true, false, 6, 1, 1, 2, 3, true, true, true>(a, n, s, false);
        }
        if (a.stride == 1 && a.wout > 38 && nk32 == 5) {
            if (prepare && wz_hp_launch<5, true, false, 3, 1, 1, 2, 3, true, true, true>(a, n, s, true) < 0) return -1;
            if (!prepare && scs) return wz_hp_launch<5, true, false, 3, 1, 1, 2, 3, true, true, true>(a, n, s, false);
        }
        if (a.stride == 2) return wz_hp_launch<4, false, false, 6, 1, 1, 2, 3, false, false, true>(a, n, s, prepare);
        if (a.wout <= 38 && nk32 == 6) {
            if (prepare && wz_hp_launch<6, true, false, 3, 1, 1, 2, 4, false, true, true>(a, n, s, true) < 0) return -1;
            if (!prepare && small44()) return wz_hp_launch<6, true, false, 3, 1, 1, 2, 4, false, true, true>(a, n, s, false);
            return wz_hp_launch<6, true, false, 4, 2, 1, 2, 4, false, true, true>(a, n, s, prepare);
        }
        if (a.wout <= 38 && nk32 >= 4 && nk32 <= 5) return wz_hp_launch<3, true, false, 4, 2, 1, 2, 2, false, true, true>(a, n, s, prepare);
        if (a.wout > 38) {
            if (prepare && wz_hp_launch<4, false, false, 3, 1, 1, 2, 3, false, false, true>(a, n, s, true) < 0) return -1;
            if (!prepare && small44()) return wz_hp_launch<4, false, false, 3, 1, 1, 2, 3, false, false, true>(a, n, s, false);
        }
        return wz_hp_launch<4, false, false, 4, 2, 1, 2, 3, false, false, true>(a, n, s, prepare);
    }
    if (a.wout > 10) {
        if (a.stride == 2) return (a.kc0 == 1 && nto == 4) ? wz_hp_launch<6, true, false, 6, 1, 1, 4, 4, false, true, true, true>(a, n, s, prepare) : -1;
        if (a.kc0 == 2 && nto == 4) return wz_hp_launch<8, true, false, 3, 1, 2, 4, 4, false, true, true, true>(a, n, s, prepare);
        if (a.kc0 == 2 && nto == 6) return wz_hp_launch<8, true, false, 3, 1, 2, 6, 4, false, true, true, true>(a, n, s, prepare);
        if (a.kc0 == 3 && nto == 6) return wz_hp_launch<8, true, false, 3, 1, 3, 6, 4, false, true, true, true>(a, n, s, prepare);
        return -1;
    }
    return -1;   // (the 10x10 maps -- blocks 13 .. 16 -- keep the linear chunk buffer in the robust program: wz_launch_mbconv_hp below)
}

// Blocks 0 (with the stem) .. 12 of SSD-MobileNet-v2 300x300.  prepare: 0 = a kernel exists (its attributes are set),
// -1 = no kernel for this shape; launch: 1.
//   maps wider than WZ_HP_CS_MAX_W (default 19): one wavefront per tile, every wave walks all chunks;
//   19x19: 8 waves per 4x4 tile, 2 - 3 chunks per wave;
//   WZ_HP_CS_MAX_W=38 / 75 puts the 38x38 / 75x75 maps on workgroups of 5 / 6 waves per tile with ONE chunk per wave as
//     well.  Measured (profiles/r02c_*): the 38x38 stride-1 blocks get faster alone (13.6 -> 11.0 us: 100 tiles x 8 frames of
//     waves walking 6 chunks each leave most of the GPU idle) but every wave repeats the halo load and the accumulators
//     take a trip through LDS, and with four lanes in flight CU x time is what counts: 37.2 k frames/s against 40.2 k
//     with one wave per tile (75x75 on it as well: 35.0 k).
// The launch shapes that lost their A/B (rounds 2 .. 5: HISTORY.md part B, each with its profile) stay reachable through their knobs in the DEVELOPMENT
// library only: HP_DEV(...) compiles its argument there and to nothing in the product library, which instantiates the ~20 shapes its defaults can reach
// at some batch size and nothing else (tests/test_gpu_parity.py holds the two libraries' rows bit-equal).
#ifdef WZ_DEV_BUILD
#define HP_DEV(...) __VA_ARGS__
#else
#define HP_DEV(...)
#endif
int wz_launch_mbconv_hp(const WzMbArgs& a0, int n, hipStream_t s, bool prepare) {
    static const int cs_max_w = wz_hp_env("WZ_HP_CS_MAX_W", 19);
    static const int cs_few_wgs = wz_hp_env("WZ_HP_CS_FEW_WGS", 64);   // chunk-split when one wave per tile gives at most this many workgroups
    // WZ_HP_SH (default 1): chunk-split workgroups fetch the halo once and share it through LDS.  Measured at batch 8
    // (profiles/r02zt_*): the 19x19 blocks 8.0 -> 6.95 us (cmid 384) and 12.3 -> 10.5 us (cmid 576), 45.6 k -> 47.3 k frames/s.
    // WZ_HP_W12 (default 0): 12 waves per 19x19 tile (3 per SIMD, re-reading the halo from LDS every pass) instead of 8 -- the
    // cmid-384 blocks 6.95 -> 6.76 us, no gain in frames/s; the cmid-576 shape does not fit 168 registers (spills: 16 us) and stays on 8.
    static const int sh = wz_hp_env("WZ_HP_SH", 1);
    static const int w12 = wz_hp_env("WZ_HP_W12", 0);
    if (a0.qenc) return wz_launch_mbconv_hp_q(a0, n, s, prepare);
    const int nto = a0.n_pad / 16;
    WzMbArgs a = a0;
    a.nsplit = 1;
    if (a.nmid_pad != a.cmid_pad || (a.cmid_pad & 31) || a.kc != (a.cmid_pad >> 5) || !a.we_lo || !a.wp_lo) return -1;
    const int nk32 = a.cmid_pad >> 5;
    // waves per SIMD of the one-wave-per-tile kernels.  Measured (profiles/r02g_*, batch 8): 3 instead of 2 takes the stem
    // block from 22.4 to 18.1 us and the stride-2 blocks from 16.5 / 10.5 to 11.9 / 9.9 us (their waves wait for the halo
    // gather and for LDS, a third wave fills the gaps), leaves the 75x75 stride-1 block where it is and costs the 38x38
    // ones 0.7 us (too few waves to fill even two per SIMD); 4 spills and loses everywhere.  WZ_HP_OCC=2|3|4 forces one.
    static const int occ_env = wz_hp_env("WZ_HP_OCC", 0);
    const int occ = occ_env ? occ_env : (a.stem || a.stride == 2 || a.wout >= 75) ? 3 : 2;
    if (a.stem) {
        if (!(a.kc0 == 1 && nto == 2 && a.stride == 1)) return -1;
        if (prepare) {
            HP_DEV((void)wz_hp_launch<4, false, true, 4, 2, 1, 2, 4>(a, n, s, true); (void)wz_hp_launch<4, false, true, 4, 2, 1, 2>(a, n, s, true);)
            return wz_hp_launch<4, false, true, 4, 2, 1, 2, 3>(a, n, s, true);
        }
        HP_DEV(if (occ == 4) return wz_hp_launch<4, false, true, 4, 2, 1, 2, 4>(a, n, s, false);
               if (occ != 3) return wz_hp_launch<4, false, true, 4, 2, 1, 2>(a, n, s, false);)
        return wz_hp_launch<4, false, true, 4, 2, 1, 2, 3>(a, n, s, false);
    }
    if (a.cin0 == 0) return -1;
    if (a.wout <= 10) {
        // 10x10 maps (blocks 13 .. 16 of the ROBUST program; the default program runs them on wz_k_mbconv_cs): two GEMM-shaped launches per block,
        // k_mbconv_hp2.hip -- the engine asks wz_mbconv_hp2_applies() and enqueues them itself (one stage-timer slot each); this entry only
        // prepares their kernels at load time.  (Rounds 3 .. 5 ran them here: lean builds, 8 waves per 4 x 4 tile streaming all of the block's split
        // weights, channel groups over workgroups for few frames -- 65 us for the four at batch 8 against 42, 41 against 33 at batch 1.)
        return prepare ? wz_launch_mbconv_hp2(a, n, s, true, 0) : -1;
    }
    if (a.wout > 19 && a.kc0 == 1 && nto == 2) {
        // Few frames (a single camera's frame at a time is the reference's normal load, detector.py:102-112): one wave per tile
        // would leave most CUs empty and every wave walking 5 - 6 chunks, ~2 us each -- there the chunks go to the waves of a
        // workgroup instead (measured at batch 1, profiles/r02t_*).  With the GPU filled by the batch it is the other way
        // round (comment above wz_launch_mbconv_hp).
        const int tiles_s1 = ((a.hout + 3) / 4) * ((a.wout + (a.stride == 1 ? 7 : 3)) / (a.stride == 1 ? 8 : 4));
        const bool few = (tiles_s1 * n + 3) / 4 <= cs_few_wgs;
        // Stride-1 blocks on the 38x38 maps: THREE waves per tile, two chunks each, halo shared through LDS.  One wave per tile walks
        // six chunks in a row on a chip that 400 such waves leave almost empty (11.3 us per block); six waves with one chunk each
        // are fastest alone but every workgroup then fills most of a CU (8.0 - 9.0 us, 45.9 k frames/s against 47.3 k); three waves
        // take 6.7 - 7.0 us and two workgroups share a CU: 47.7 k frames/s, p50 0.378 -> 0.369 ms (profiles/r03_wave_counts_*).
        static const int cs_s1_max_w = wz_hp_env("WZ_HP_CS_S1_MAX_W", 38);   // the same threshold for the stride-1 blocks alone
        const bool cs = (prepare || a.wout <= cs_max_w || (a.stride == 1 && a.wout <= cs_s1_max_w) || few) && nk32 >= 4 && nk32 <= 6;
        if (a.stride == 1) {        // 4 x 8 tiles, halo 6 x 10 = 60 pixels
            static const int cs_nw = wz_hp_env("WZ_HP_CS_NW", 3);   // 2 / 3 / 4: that many waves per tile, several chunks each (shared halo); 0: one chunk per wave
            if (prepare) {
                // live with the defaults: 3 waves per tile (few frames on the 75x75 map), 5 / 6 waves at four per SIMD (38x38), one wave per tile at
                // three per SIMD (75x75 and wider) or two (a 38x38 block whose chunk count the chunk-split builds do not take)
                (void)wz_hp_launch<3, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, true);
                (void)wz_hp_launch<5, true, false, 4, 2, 1, 2, 4, false, true>(a, n, s, true);
                (void)wz_hp_launch<6, true, false, 4, 2, 1, 2, 4, false, true>(a, n, s, true);
                (void)wz_hp_launch<4, false, false, 4, 2, 1, 2, 3>(a, n, s, true);
                HP_DEV((void)wz_hp_launch<2, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, true);
                       (void)wz_hp_launch<4, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, true);
                       (void)wz_hp_launch<5, true, false, 4, 2, 1, 2>(a, n, s, true);
                       (void)wz_hp_launch<6, true, false, 4, 2, 1, 2>(a, n, s, true);
                       (void)wz_hp_launch<5, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, true);
                       (void)wz_hp_launch<6, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, true);
                       (void)wz_hp_launch<4, false, false, 4, 2, 1, 2, 4>(a, n, s, true);)
                return wz_hp_launch<4, false, false, 4, 2, 1, 2>(a, n, s, true);
            }
            static const int cs75_nw = wz_hp_env("WZ_HP_CS75_NW", 0);   // the 75x75 stride-1 block: 2 / 3 waves per tile (0: one wave per tile)
            static const int cs_occ4 = wz_hp_env("WZ_HP_CS_OCC4", 1);   // chunk-split stride-1 tiles: one chunk per wave at 128 registers (4 waves per SIMD)
            HP_DEV(if (!prepare && sh && a.wout > 38 && a.wout <= 75 && nk32 >= 4 && nk32 <= 6) {
                if (cs75_nw == 2) return wz_hp_launch<2, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, false);
                if (cs75_nw == 3) return wz_hp_launch<3, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, false);
                if (cs75_nw == 5 && nk32 <= 5) return wz_hp_launch<5, true, false, 4, 2, 1, 2, 4, false, true>(a, n, s, false);
            })
            (void)cs75_nw;
            if (!prepare && cs && sh && cs_occ4 && a.wout <= 38)
                return nk32 <= 5 ? wz_hp_launch<5, true, false, 4, 2, 1, 2, 4, false, true>(a, n, s, false)
                                 : wz_hp_launch<6, true, false, 4, 2, 1, 2, 4, false, true>(a, n, s, false);
            HP_DEV(if (cs && sh && cs_nw == 2) return wz_hp_launch<2, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, false);)
            if (cs && sh && cs_nw == 3) return wz_hp_launch<3, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, false);
            HP_DEV(if (cs && sh && cs_nw == 4) return wz_hp_launch<4, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, false);
                   if (cs && sh && nk32 <= 5) return wz_hp_launch<5, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, false);
                   if (cs && sh) return wz_hp_launch<6, true, false, 4, 2, 1, 2, 2, false, true>(a, n, s, false);
                   if (cs && nk32 <= 5) return wz_hp_launch<5, true, false, 4, 2, 1, 2>(a, n, s, false);
                   if (cs) return wz_hp_launch<6, true, false, 4, 2, 1, 2>(a, n, s, false);)
            if (occ == 3) return wz_hp_launch<4, false, false, 4, 2, 1, 2, 3>(a, n, s, false);
            HP_DEV(if (occ == 4) return wz_hp_launch<4, false, false, 4, 2, 1, 2, 4>(a, n, s, false);)
            return wz_hp_launch<4, false, false, 4, 2, 1, 2>(a, n, s, false);
        }
        // stride 2: 4 x 4 tiles, halo 9 x 9 = 81 pixels
        static const int cs2_nw = wz_hp_env("WZ_HP_CS2_NW", 0);   // stride-2 blocks on maps up to WZ_HP_CS2_MAX_W: that many waves per tile
        static const int cs2_max_w = wz_hp_env("WZ_HP_CS2_MAX_W", 0);
        if (prepare) {
            (void)wz_hp_launch<5, true, false, 6, 1, 1, 2, 2, false, true>(a, n, s, true);     // (few frames)
            HP_DEV((void)wz_hp_launch<2, true, false, 6, 1, 1, 2, 2, false, true>(a, n, s, true);
                   (void)wz_hp_launch<3, true, false, 6, 1, 1, 2, 2, false, true>(a, n, s, true);
                   (void)wz_hp_launch<5, true, false, 6, 1, 1, 2>(a, n, s, true);
                   (void)wz_hp_launch<4, false, false, 6, 1, 1, 2, 4>(a, n, s, true);
                   (void)wz_hp_launch<4, false, false, 6, 1, 1, 2>(a, n, s, true);)
            return wz_hp_launch<4, false, false, 6, 1, 1, 2, 3>(a, n, s, true);
        }
        HP_DEV(if (sh && cs2_nw == 3 && a.wout <= cs2_max_w && nk32 >= 4 && nk32 <= 6)
                   return wz_hp_launch<3, true, false, 6, 1, 1, 2, 2, false, true>(a, n, s, false);
               if (sh && cs2_nw == 2 && a.wout <= cs2_max_w && nk32 >= 3 && nk32 <= 6)
                   return wz_hp_launch<2, true, false, 6, 1, 1, 2, 2, false, true>(a, n, s, false);)
        (void)cs2_nw; (void)cs2_max_w;
        if (cs && sh && nk32 <= 5) return wz_hp_launch<5, true, false, 6, 1, 1, 2, 2, false, true>(a, n, s, false);
        HP_DEV(if (cs && nk32 <= 5) return wz_hp_launch<5, true, false, 6, 1, 1, 2>(a, n, s, false);
               if (occ == 4) return wz_hp_launch<4, false, false, 6, 1, 1, 2, 4>(a, n, s, false);
               if (occ != 3) return wz_hp_launch<4, false, false, 6, 1, 1, 2>(a, n, s, false);)
        return wz_hp_launch<4, false, false, 6, 1, 1, 2, 3>(a, n, s, false);
    }
    if (a.wout > 19) return -1;
    if (a.stride == 2) {
        if (a.kc0 == 1 && nto == 4) {
            static const int cs6_nw = wz_hp_env("WZ_HP_CS6_NW", wz_latency_schedule() ? 8 : 3);   // waves per tile of the 38x38 -> 19x19 block (6 chunks): 3 with two chunks
                                                                      // each (49.8 k -> 50.5 k frames/s; 8: two waves idle, a CU per workgroup)
            static const int cs6_lean4 = wz_hp_env("WZ_HP_CS6_LEAN4", 1);   // six waves (one chunk each) at 128 registers
            if (prepare) {
                HP_DEV((void)wz_hp_launch<HP_CS_WAVES, true, false, 6, 1, 1, 4, 2, false, true>(a, n, s, true);
                       (void)wz_hp_launch<3, true, false, 6, 1, 1, 4, 2, false, true>(a, n, s, true);
                       (void)wz_hp_launch<6, true, false, 6, 1, 1, 4, 2, false, true>(a, n, s, true);
                       (void)wz_hp_launch<HP_CS_WAVES, true, false, 6, 1, 1, 4>(a, n, s, true);)
                return wz_hp_launch<6, true, false, 6, 1, 1, 4, 4, false, true, false, true>(a, n, s, true);
            }
            HP_DEV(if (!(sh && cs6_lean4)) {
                if (sh && cs6_nw == 3) return wz_hp_launch<3, true, false, 6, 1, 1, 4, 2, false, true>(a, n, s, false);
                if (sh && cs6_nw == 6) return wz_hp_launch<6, true, false, 6, 1, 1, 4, 2, false, true>(a, n, s, false);
                if (sh) return wz_hp_launch<HP_CS_WAVES, true, false, 6, 1, 1, 4, 2, false, true>(a, n, s, false);
                return wz_hp_launch<HP_CS_WAVES, true, false, 6, 1, 1, 4>(a, n, s, false);
            })
            (void)cs6_nw; (void)cs6_lean4;
            return wz_hp_launch<6, true, false, 6, 1, 1, 4, 4, false, true, false, true>(a, n, s, false);
        }
        return -1;
    }
    // 12 chunks (cmid 384), WZ_HP_ONEPASS=1: 12 waves with one chunk each instead of 8 waves with up to two.  Measured
    // (profiles/r02r_*, A/B in one run): 8.97 against 8.0 us per block -- four more waves repeat the halo load and the
    // accumulators of twelve waves meet in LDS; the chunk walk is not what these launches wait for.  Off by default.
    static const int onepass = wz_hp_env("WZ_HP_ONEPASS", 0);
    HP_DEV(if (nk32 <= 12 && a.kc0 == 2 && (nto == 4 || nto == 6)) {
        if (prepare) {
            (void)wz_hp_launch<12, true, false, 3, 1, 2, 4, 3, true>(a, n, s, true);
            (void)wz_hp_launch<12, true, false, 3, 1, 2, 6, 3, true>(a, n, s, true);
        } else if (onepass == 1) {
            return nto == 4 ? wz_hp_launch<12, true, false, 3, 1, 2, 4, 3, true>(a, n, s, false)
                            : wz_hp_launch<12, true, false, 3, 1, 2, 6, 3, true>(a, n, s, false);
        }
    })
    (void)onepass;
    // 19x19 blocks: FOUR waves per tile (3 - 5 chunks each) instead of eight.  Alone a block gets slower (5.2 -> 6.5 us, 9.3 -> 11.5 us:
    // the chunk walk is longer) -- but a workgroup of four 256-register waves takes half a CU's register file, so two of them (of
    // this lane's launch or of another lane's) share a CU, where eight waves own it: 47.5 k -> 49.1 k frames/s with four lanes in
    // flight.  3 waves: 48.2 k; 5: 45.8 k; 6: 46.8 k (workgroups that neither fill a CU nor leave room for a second one);
    // WZ_HP_CS19_NW=8 is the lowest-latency setting (p50 0.372 against 0.380 ms).  profiles/r03_wave_counts_*.
    static const int cs19_nw = wz_hp_env("WZ_HP_CS19_NW", wz_latency_schedule() ? 8 : 4);
    // WZ_HP_CS19_LEAN4=1 (default, later in round 3): EIGHT waves per tile at 128 registers (4 per SIMD; halo fragments per pixel tile, no
    // weight prefetch, accumulators through LDS in two rounds): as many waves as the lowest-latency setting, the CU footprint of the
    // four-wave one -- 6.7 / 5.9 / 6.2 / 7.7 / 11.7 / 10.3 -> 5.7 / 5.0 / 5.2 / 6.3 / 10.1 / 9.0 us, 52.1 -> 52.5 k frames/s, p50 0.399 ->
    // 0.386 ms (profiles/r03_four_waves_per_simd.txt).  0: the wave counts below.
    static const int cs19_lean4 = wz_hp_env("WZ_HP_CS19_LEAN4", 1);
#define HP_CASE(K, N)                                                                                         \
    if (a.kc0 == K && nto == N) {                                                                             \
        if (prepare) {                                                                                        \
            HP_DEV((void)wz_hp_launch<3, true, false, 3, 1, K, N, 2, false, true>(a, n, s, true);            \
                   (void)wz_hp_launch<4, true, false, 3, 1, K, N, 2, false, true>(a, n, s, true);            \
                   (void)wz_hp_launch<5, true, false, 3, 1, K, N, 2, false, true>(a, n, s, true);            \
                   (void)wz_hp_launch<6, true, false, 3, 1, K, N, 2, false, true>(a, n, s, true);            \
                   (void)wz_hp_launch<HP_CS_WAVES, true, false, 3, 1, K, N, 2, false, true>(a, n, s, true);  \
                   if (K == 2) (void)wz_hp_launch<12, true, false, 3, 1, 2, N, 3, false, true>(a, n, s, true); \
                   (void)wz_hp_launch<HP_CS_WAVES, true, false, 3, 1, K, N>(a, n, s, true);)                 \
            return wz_hp_launch<8, true, false, 3, 1, K, N, 4, false, true, false, true>(a, n, s, true);     \
        }                                                                                                     \
        HP_DEV(if (!(sh && cs19_lean4)) {                                                                     \
            if (sh && cs19_nw == 3) { const int r = wz_hp_launch<3, true, false, 3, 1, K, N, 2, false, true>(a, n, s, false); if (r >= 0) return r; } \
            if (sh && cs19_nw == 5) { const int r = wz_hp_launch<5, true, false, 3, 1, K, N, 2, false, true>(a, n, s, false); if (r >= 0) return r; } \
            if (sh && cs19_nw == 4) { const int r = wz_hp_launch<4, true, false, 3, 1, K, N, 2, false, true>(a, n, s, false); if (r >= 0) return r; } \
            if (sh && cs19_nw == 6) { const int r = wz_hp_launch<6, true, false, 3, 1, K, N, 2, false, true>(a, n, s, false); if (r >= 0) return r; } \
            if (sh && w12 && K == 2) return wz_hp_launch<12, true, false, 3, 1, 2, N, 3, false, true>(a, n, s, false); \
            if (sh) return wz_hp_launch<HP_CS_WAVES, true, false, 3, 1, K, N, 2, false, true>(a, n, s, false); \
            return wz_hp_launch<HP_CS_WAVES, true, false, 3, 1, K, N>(a, n, s, false);                        \
        })                                                                                                    \
        return wz_hp_launch<8, true, false, 3, 1, K, N, 4, false, true, false, true>(a, n, s, false);        \
    }
    HP_CASE(2, 4);
    HP_CASE(2, 6);
    HP_CASE(3, 6);
#undef HP_CASE
    (void)cs19_nw; (void)cs19_lean4; (void)w12; (void)sh;
    return -1;
}
#undef HP_DEV
