// The six SSD heads as a WIDE implicit GEMM, one launch: one workgroup = 128 pixels x up to 320 output channels (the heads on the
// 3x3 ... 1x1 maps fill little of a pixel tile; what they need is their long K loop cut into slices like everybody else's).
//
// Why another tile kernel.  wz_k_conv_rs (k_conv_rs.h) gives a wave 128 pixels x 32 channels: per K step (64 input
// channels of one filter tap) it reads the whole 16 KiB activation tile from LDS for 32 MFMAs, i.e. per workgroup step
// 64 KiB of LDS reads (512 cycles at 128 B/clk), 32 KiB through the vector memory path (512 cycles at 64 B/clk) and
// 544 cycles of MFMA -- three pipes loaded alike, executed one after the other by four waves that run in lockstep between
// barriers (measured 1 750 cycles per step, profiles/r01zy_*): 0.2 of the MFMA peak, at any batch size
// (profiles/r02v_batch_sweep.txt).  The operand traffic per MFMA falls with the number of channels a wave owns:
//   * here a wave owns 128 pixels x 80 channels (five 16-channel tiles; accumulators: 160 registers, the kernel runs one
//     wave per SIMD with the unified 512-register file): 80 MFMAs (1 280 cycles) per 16 KiB of LDS reads -- LDS 512,
//     vector memory <= 900 (16 KiB activations + <= 40 KiB weights), MFMA 1 280 cycles per workgroup step;
//     -- since round 5 the DEFAULT is the same body with THREE tiles per wave (wz_k_conv_wide_group3: 96 accumulator registers, 248 in all, two
//     workgroups per CU and two waves per SIMD that cover each other's barrier waits): 48 MFMAs per 16 KiB of LDS reads, 1.5 - 2 x the activation
//     tiles through LDS, and a launch that is 27 % shorter (38.0 -> 27.9 us at batch 8) -- see wz_conv_wide_ntw() below;
//   * and the step is software-pipelined in two halves (the two 32-channel MFMA K chunks of the step): while the MFMAs
//     of one half run, the LDS fragment reads of the next half, the LDS write of the next step's activations and the
//     global loads of the step after are in flight.  One `s_barrier` per step, preceded by `lgkmcnt(0)` only: the weight
//     loads in flight are NOT drained (`__syncthreads()` would).
// Operands exactly as in wz_k_conv_rs: weights straight from L2 into registers (the packed layout is fragment order,
// a wave's fragments are needed by no other wave), activations global -> VGPR -> LDS in full 128-byte lines through a
// buffer descriptor (out-of-frame taps read zeros), same XOR-swizzled LDS image (conflict-free fragment reads), same
// K order (channel pair outermost, tap innermost), same XCD-aware tile order.
// Output: fp32 partial sums for wz_k_splitk_reduce_group, in fragment order [K slice][M / 16][n_pad / 16][64 lanes][4] (WzConvArgs::frag_ws;
// [K slice][M][n_pad] without it) -- always, also with one K slice: this
// kernel only serves the heads, whose epilogue (bias, scatter, box decode, candidate marking) lives in that launch.
#include "wz_common.h"

typedef __attribute__((ext_vector_type(4))) unsigned int uint4_t;

#define WZ_WIDE_TM 128
#ifndef WZ_WIDE_STAMPS
#define WZ_WIDE_STAMPS 0   // 1: cycle counts of the first workgroup's wave 0 into WzConvArgs::dbg (tools/wide_probe.py)
#endif

template <int KS, int NTW>
__device__ __forceinline__ void wz_conv_wide_body(const WzConvArgs& a, unsigned char* smem, const int L) {
    constexpr int taps = KS * KS;
    constexpr unsigned OOB = 0x7ffffff0u;   // buffer offset beyond every tensor: the load returns zeros
    const long long t_entry = WZ_WIDE_STAMPS ? __builtin_readcyclecounter() : 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    int bx, by, bz;
    {   // XCD-aware tile order: an XCD's workgroups are neighbours in (pixel tile, channel group, K slice) order, so the
        // workgroups that read the same weight slice share an L2
        const int total = a.grid_m * a.grid_n * a.splitk;
        if (L >= total) return;   // padding workgroup of the grouped launch
        const int xcd = L & 7, slot = L >> 3;
        const int qd = total >> 3, rm = total & 7;
        const int V = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
        bx = V % a.grid_m;
        const int rest = V / a.grid_m;
        by = rest % a.grid_n;
        bz = rest / a.grid_n;
    }
    const int m_base = bx * WZ_WIDE_TM;
    // the workgroup's channel tiles [g0, g0 + gcnt) dealt out over the four waves: q or q + 1 each
    const int g0 = by * a.nt_group;
    const int gcnt = min(a.nt_group, a.nt_live - g0);
    const int q = gcnt >> 2, r = gcnt & 3;
    const int nt_w = g0 + wave * q + min(wave, r);
    const int cnt = q + (wave < r ? 1 : 0);

    const int hw = a.hout * a.wout;
    const int n_frames = (a.M + hw - 1) / hw;
    // Activations are read through a descriptor whose base lies `bias` bytes BELOW the tensor: a lane's offset is then its
    // pixel at filter tap (0, 0) -- which may be above / left of the frame -- plus `bias`, never negative, and the tap's own offset
    // (wave-uniform) travels in the instruction's scalar offset.  No lane ever addresses the bytes below the tensor: the taps
    // that fall outside the frame get bit 31 set in their offset, which is beyond the descriptor's range (the load returns zeros).
    const int bias = (a.pad_t * a.win + a.pad_l) * a.cin * 2;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.in - bias), 0, n_frames * a.hin * a.win * a.cin * 2 + bias, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (a.n_pad >> 4) * taps * a.kc * 1024, 0x00020000);

    // activation staging (as wz_k_conv_rs): instruction i of this wave = pixels wave*32 + i*8 + (lane >> 3), 16-byte chunk
    // (lane & 7), stored at slot chunk ^ ((P >> 1) & 7) of the pixel's 128 bytes (lane-linear LDS write, swizzled source)
    unsigned pix_off[4];    // biased byte offset of the lane's pixel / chunk at tap (0, 0)
    unsigned tap_out[4];    // bit t set: tap t of that pixel lies outside the frame (all set for a pixel past the end of M); bit 31 always
    // (filled in by `pixel_table` below, after the first weight loads are on their way)
    auto pixel_table = [&]() {
        // The two integer divisions that turn a pixel index into (frame, row, column) cost ~70 VALU instructions per pixel: each of
        // the tile's 128 pixels is worked out ONCE, by the thread of that number, and handed to the lanes that stage it through
        // 1 KiB of LDS (a first version did all four of a lane's pixels in every lane: 600 VALU instructions, and 8 700 cycles from
        // kernel entry to the first MFMA -- profiles/r02ze_*).
        uint2* const tab = reinterpret_cast<uint2*>(smem + 2 * 16384);
        if (threadIdx.x < WZ_WIDE_TM) {
            const int m = m_base + (int)threadIdx.x;
            const bool mv = m < a.M;
            const int mm = mv ? m : 0;
            const int b = mm / hw, rem = mm - b * hw;
            const int oy = rem / a.wout, ox = rem - oy * a.wout;
            const int iy0 = oy * a.stride - a.pad_t, ix0 = ox * a.stride - a.pad_l;
            unsigned cols = 0, mask = 0;
#pragma unroll
            for (int k = 0; k < KS; ++k) cols |= ((unsigned)(ix0 + k) < (unsigned)a.win ? 1u : 0u) << k;
#pragma unroll
            for (int k = 0; k < KS; ++k) mask |= ((unsigned)(iy0 + k) < (unsigned)a.hin ? cols : 0u) << (KS * k);
            uint2 ent;
            ent.x = (unsigned)(((b * a.hin + iy0) * a.win + ix0) * a.cin * 2 + bias);
            ent.y = ~(mv ? mask : 0u);
            tab[threadIdx.x] = ent;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint2 ent = tab[wave * 32 + i * 8 + (lane >> 3)];
            const int chunk = (lane & 7) ^ ((i * 4 + (lane >> 4)) & 7);
            pix_off[i] = ent.x + (unsigned)(chunk * 16);
            tap_out[i] = ent.y;
        }
    };
    // weight fragments: lane's 16 bytes of the 1 KiB fragment; a tile slot this wave does not own reads zeros
    unsigned w_voff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) w_voff[nt] = nt < cnt ? (unsigned)(lane * 16) : OOB;

    const int nsteps = a.kchunks >> 1;
    const int per = (nsteps + a.splitk - 1) / a.splitk;
    const int s0 = bz * per, s1 = min(s0 + per, nsteps);
    const int last = s1 - 1;

    // (kernel arguments the loop needs, in registers: the asm barrier below is a compiler memory barrier and would have
    // them re-read from the argument segment every step)
    const int kc_ = a.kc, win_ = a.win, cin_ = a.cin;

    // first fragment of each of the wave's tile slots (a slot the wave does not own reads zeros through w_voff, from tile nt_w)
    unsigned w_base[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
        w_base[nt] = (unsigned)__builtin_amdgcn_readfirstlane((nt_w + (nt < cnt ? nt : 0)) * taps * kc_ * 1024);

    // The pieces of a step, one or two instructions each, so that the step below can place them between MFMAs.  A wave issues in
    // order and runs alone on its SIMD: every scalar or vector instruction between two MFMAs is an issue slot the matrix pipe may
    // wait for (a first version spent 68 SALU + 53 VALU per 80 MFMAs on `s % 9`, per-load selects and address sums: 46 % MFMA
    // busy, profiles/r02za_*).  So a position in K (filter tap, channel pair) is carried as running offsets, advanced once per step:
    //   wq       byte offset of the position's first weight fragment within a tile's fragments
    //   xs       scalar offset of the activation loads: the tap's offset within the frame + the channel pair's
    //   tbit,sh  the tap's bit in `tap_out` and the shift that moves it to bit 31 (a lane offset with bit 31 set reads zeros)
    // A position past the slice's end (the phantom second step of an odd slice's last pair) keeps the last real offsets and
    // selects bit 31 of `tap_out`, which is always set: its activations are zeros, whatever weights it multiplies them with.
    struct Pos { int idx, kx, ky; unsigned wq, xs, tbit, sh; };
    auto pos_at = [&](int s_) {
        Pos k;
        k.idx = min(s_, last);
        const int t = (KS == 1) ? 0 : k.idx % taps;
        const int c = (KS == 1) ? k.idx * 2 : (k.idx / taps) * 2;
        k.ky = t / KS;
        k.kx = t - k.ky * KS;
        k.wq = (unsigned)(t * kc_ + c) * 1024u;
        k.xs = (unsigned)((k.ky * win_ + k.kx) * cin_ * 2 + c * 64);
        k.tbit = s_ <= last ? 1u << t : 0x80000000u;
        k.sh = s_ <= last ? 31u - t : 0u;
        return k;
    };
    const unsigned w_tap = (unsigned)kc_ * 1024u, w_wrap = 2048u - (unsigned)(taps - 1) * w_tap;
    const unsigned x_tap = (unsigned)cin_ * 2u, x_row = (unsigned)(win_ - (KS - 1)) * x_tap;
    const unsigned x_wrap = 128u - (unsigned)(((KS - 1) * win_ + (KS - 1)) * cin_ * 2);
    auto advance = [&](Pos& k) {   // (written as branches on purpose: as a chain of selects the compiler turned it into a lookup table in scratch memory)
        if (k.idx < last) {
            ++k.idx;
            if (k.kx == KS - 1) {
                k.kx = 0;
                if (k.ky == KS - 1) {
                    k.ky = 0; k.wq += w_wrap; k.xs += x_wrap; k.tbit = 1u; k.sh = 31u;
                } else {
                    ++k.ky; k.wq += w_tap; k.xs += x_row; k.tbit <<= 1; --k.sh;
                }
            } else {
                ++k.kx; k.wq += w_tap; k.xs += x_tap; k.tbit <<= 1; --k.sh;
            }
        } else {   // the last real position, or past it: becomes / stays phantom
            k.tbit = 0x80000000u; k.sh = 0u;
        }
    };
    auto load_x1 = [&](const Pos& k, int i, uint4_t (&rb)[4]) {   // quarter i of this wave's share of the step's activation tile
        const unsigned voff = ((tap_out[i] & k.tbit) << k.sh) | pix_off[i];
        rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, k.xs, 0);
    };
    auto store_x1 = [&](int buf, int i, const uint4_t (&rb)[4]) {
        *reinterpret_cast<uint4_t*>(smem + buf * 16384 + (wave * 4 + i) * 1024 + lane * 16) = rb[i];
    };
    // K chunk h of tile slot nt: fragment (tile, tap, chunk) = 1 KiB at ((tile * taps + tap) * kc + chunk) * 1 KiB, the step's two
    // chunks are neighbours
    auto load_w1 = [&](unsigned wq, int nt, int h, half8_t (&f)[NTW][2]) {
        f[nt][h] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_voff[nt] + h * 1024u, w_base[nt] + wq, 0));
    };
    auto read_x1 = [&](int buf, int kc, int mt, half8_t (&fb)[8]) {   // pixel P = mt*16 + r16, chunk kc*4 + g, slot swizzled by (P >> 1) & 7
        fb[mt] = *reinterpret_cast<const half8_t*>(smem + buf * 16384 + (mt * 16 + r16) * 128 + (((kc * 4 + g) ^ ((r16 >> 1) & 7)) * 16));
    };
    float4_t acc[8][NTW];

    // Accumulators pinned to AGPRs ("+a"): left to the allocator, part of them lived in VGPRs and 228 copies per pair of steps
    // shuffled tiles between the two register files.  Consecutive MFMAs never share an accumulator (the same one comes back 40
    // MFMAs later).  `fence` = a scheduling barrier: the instruction order of a step is the program order written below.
    auto mfma = [&](int kc, int mt, int nt, const half8_t (&f)[NTW][2], const half8_t (&fb)[8]) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(f[nt][kc]), "v"(fb[mt]));
    };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
    const bool stamp = WZ_WIDE_STAMPS && a.dbg && threadIdx.x == 0 && L == 0;
    long long cy_work = 0, cy_sync = 0, t_prev = 0;
    auto lds_barrier = [&]() {
        if (WZ_WIDE_STAMPS) {   // work = barrier release -> arrival at the next barrier's wait; sync = that wait + the barrier
            const long long tA = __builtin_readcyclecounter();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const long long tB = __builtin_readcyclecounter();
            if (t_prev) { cy_work += tA - t_prev; cy_sync += tB - tA; }
            t_prev = tB;
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };

    uint4_t rb[4];
    half8_t fw0[NTW][2], fw1[NTW][2];   // weights of the even / odd steps of a pair
    half8_t fx0[8], fx1[8];             // activation fragments of the K chunk in hand / of the next one
    if (s0 >= s1) return;
    static_assert(NTW >= 3 && NTW <= 6, "the step below has slots for the loads of 2 .. 6 tiles per wave");

    // One step, parity P: LDS buffer P holds its activations, `fw` its weights, fx0 its first K chunk (read during the
    // previous step); `rb` holds the next step's activations.  A row = the NTW MFMAs of one pixel tile (16 cycles each); one
    // memory instruction goes behind the 2nd and one behind the 4th MFMA of a row, where it issues in the shadow of the matrix pipe.
    //   half 1: MFMAs of K chunk 0  ||  LDS reads of chunk 1; chunk-1 weights of the NEXT step (into the other set, whose
    //           chunk-1 registers the previous half freed); LDS writes of the next step's activations; activation loads of
    //           the step after
    //   barrier, preceded by lgkmcnt(0) only (the next step's LDS buffer is complete; nobody reads buffer P any more)
    //   half 2: MFMAs of K chunk 1  ||  LDS reads of the next step's chunk 0; chunk-0 weights of the step AFTER the next
    //           (into this step's own set, whose chunk-0 registers half 1 freed)
    // Every weight fragment is requested a step and a half before its first MFMA: the heads' weights are streamed once per batch
    // (21 MB, out of HBM or the Infinity Cache -- not an L2 hit), and with one step of distance the wave waited for them every
    // step (measured 2 700 cycles per step, profiles/r02y_*).
    // One step, parity P: LDS buffer P holds its activations, `fw` its weights, fx0 its first K chunk (read during the
    // previous step); `rb` holds the next step's activations.  A row = the NTW MFMAs of one pixel tile (16 cycles each); the
    // other instructions go behind the 1st, 2nd and last MFMA of a row, one piece per place, so that each issues in the shadow of
    // the matrix pipe.
    //   half 1: MFMAs of K chunk 0  ||  LDS reads of chunk 1 (one per row); chunk-1 weights of the NEXT step (into the other
    //           set, whose chunk-1 registers the previous half freed); LDS writes of the next step's activations; activation
    //           loads of the step after
    //   barrier, preceded by lgkmcnt(0) only (the next step's LDS buffer is complete; nobody reads buffer P any more)
    //   half 2: MFMAs of K chunk 1  ||  LDS reads of the next step's chunk 0 (one per row); chunk-0 weights of the step AFTER
    //           the next (into this step's own set, whose chunk-0 registers half 1 freed); the position counters move on
    // Every weight fragment is requested a step and a half before its first MFMA.
    unsigned wq1;   // weights of the next step
    Pos k2;         // the step after it
    auto step = [&](const int P, half8_t (&fw)[NTW][2], half8_t (&fw_next)[NTW][2]) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                mfma(0, mt, nt, fw, fx0);
                fence();
                if (nt == 0) {
                    read_x1(P, 1, mt, fx1);
                    fence();
                } else if (nt == 1 || nt == NTW - 1) {
                    const int j = mt * 2 + (nt == 1 ? 0 : 1);   // place
                    if (j < NTW) load_w1(wq1, j, 1, fw_next);
                    else if (j < NTW + 4) store_x1(P ^ 1, j - NTW, rb);
                    else if (j < NTW + 8) load_x1(k2, j - NTW - 4, rb);
                    fence();
                }
            }
        }
        lds_barrier();
        wq1 = k2.wq;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                mfma(1, mt, nt, fw, fx1);
                fence();
                if (nt == 0) {
                    read_x1(P ^ 1, 0, mt, fx0);
                    fence();
                } else if (nt == 1 || nt == NTW - 1) {
                    const int j = mt * 2 + (nt == 1 ? 0 : 1);
                    if (j < NTW) load_w1(wq1, j, 0, fw);
                    else if (j == NTW) advance(k2);
                    fence();
                }
            }
        }
    };

    {   // prologue, in the order of what has to travel furthest: the weights of the first two steps (HBM or the Infinity Cache),
        // then the pixel table and the first activation tiles (L2), the accumulators cleared while those are on their way
        const Pos k0 = pos_at(s0), k1 = pos_at(s0 + 1);
        k2 = pos_at(s0 + 2);
        wq1 = k1.wq;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) { load_w1(k0.wq, nt, 0, fw0); load_w1(k0.wq, nt, 1, fw0); }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) load_w1(k1.wq, nt, 0, fw1);
        fence();
        pixel_table();
#pragma unroll
        for (int i = 0; i < 4; ++i) load_x1(k0, i, rb);
        fence();
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};
        fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) store_x1(0, i, rb);
#pragma unroll
        for (int i = 0; i < 4; ++i) load_x1(k1, i, rb);
        lds_barrier();
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) read_x1(0, 0, mt, fx0);
    }
    const long long t_loop = WZ_WIDE_STAMPS ? __builtin_readcyclecounter() : 0;
    for (int s = s0; s < s1; s += 2) {   // always whole pairs: the second step of an odd slice's last pair multiplies zeros
        step(0, fw0, fw1);
        step(1, fw1, fw0);
    }

    const long long t_done = WZ_WIDE_STAMPS ? __builtin_readcyclecounter() : 0;
    if (stamp) {
        a.dbg[0] = (unsigned long long)cy_work;
        a.dbg[1] = (unsigned long long)cy_sync;
        a.dbg[2] = (unsigned long long)((s1 - s0 + 1) & ~1);
        a.dbg[3] = (unsigned long long)(t_loop - t_entry);
        a.dbg[4] = (unsigned long long)(t_done - t_loop);
    }
    // the compiler does not see MFMAs in the asm statements: keep the accumulator reads of the epilogue behind the last result
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    float* const ws = reinterpret_cast<float*>(a.out);
    if (a.frag_ws) {
        // fragment order: an accumulator tile is 1 KiB in one piece (lane l's four values at l * 16) -- whole lines per store instruction, where
        // [slice][pixel][column] order makes sixteen 64-byte pieces of it, one per pixel row (the epilogue of a workgroup was 9.9 k cycles
        // of a 65 k-cycle launch: profiles/r02zf_wide_heads_phase_cycles.txt).  The grouped reduce reads the fragments back the same way.
        const int mtt = (a.M + 15) >> 4, ntt = a.n_pad >> 4;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            if (nt >= cnt) break;   // wave-uniform
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int mtg = (m_base >> 4) + mt;
                if (mtg < mtt) *reinterpret_cast<float4_t*>(ws + (((size_t)bz * mtt + mtg) * ntt + (nt_w + nt)) * 256 + lane * 4) = acc[mt][nt];
            }
        }
    } else {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        if (nt >= cnt) break;   // wave-uniform
        const int n4 = (nt_w + nt) * 16 + g * 4;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int m = m_base + mt * 16 + r16;
            if (m < a.M) *reinterpret_cast<float4_t*>(ws + ((size_t)bz * a.M + m) * a.n_pad + n4) = acc[mt][nt];
        }
    }
    }
    if (WZ_WIDE_STAMPS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (stamp) a.dbg[5] = (unsigned long long)(__builtin_readcyclecounter() - t_done);
    }
}

__global__ __launch_bounds__(256, 1) void wz_k_conv_wide_group(const WzConvGroup g) {
    WZ_LANE_STAMP(g.stamp);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384 + WZ_WIDE_TM * 8];   // two activation tiles + the pixel table
    int e = 0;
    while (e + 1 < g.n && (int)blockIdx.x >= g.first[e + 1]) ++e;   // wave-uniform
    wz_conv_wide_body<3, 5>(g.a[e], smem, (int)blockIdx.x - g.first[e]);
}

// The same body with THREE channel tiles per wave (128 pixels x up to 192 channels per workgroup) at <= 256 registers: two workgroups share a
// CU, two waves a SIMD (round 5; WZ_WIDE_NTW=3).  Alone the five-tile form does more matrix work per LDS byte (80 against 48 MFMAs per 16 KiB of fragment
// reads); what this form is for is the four-lane case: a 344-register workgroup needs a WHOLE free CU and, with three other lanes' workgroups scattered over
// the chip, waits for one (the launch takes 55 us under load against 26.5 us alone: profiles/r05zz_lane_overlap_robust.txt), a 256-register one moves in
// beside whatever holds the other half.
__global__ __launch_bounds__(256, 2) void wz_k_conv_wide_group3(const WzConvGroup g) {
    WZ_LANE_STAMP(g.stamp);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384 + WZ_WIDE_TM * 8];
    int e = 0;
    while (e + 1 < g.n && (int)blockIdx.x >= g.first[e + 1]) ++e;   // wave-uniform
    wz_conv_wide_body<3, 3>(g.a[e], smem, (int)blockIdx.x - g.first[e]);
}

// channel tiles per wave of the build in use (5: one workgroup per CU; 3: two) -- the workgroup's tile is four times that
int wz_conv_wide_ntw() {
    // default 3 since round 5: the heads' launch 38.0 -> 27.9 us alone, a lone batch of eight 0.391 -> 0.379 ms, frames/s with four lanes unchanged
    // (47.7 - 47.8 k both ways; under load the launch takes ~55 us either way: profiles/r05_heads_three_tiles_per_wave.txt).  WZ_WIDE_NTW=5: the round-2 .. 4 build.
    static const int ntw = [] { const char* e = wz_dev_getenv("WZ_WIDE_NTW"); const int v = (e && e[0]) ? atoi(e) : 3; return v == 5 ? 5 : 3; }();
    return ntw;
}

// 3x3 heads with a K loop worth tiling (the conditions of the LDS-tiled kernels); the engine's WZ_CONV_WIDE=0 gives them back to
// wz_k_conv_rs
bool wz_conv_wide_applies(const WzConvArgs& a) {
    // whole 64-column tiles in the packed weights, whole 2-chunk K steps, 32-channel chunks.  Any number of pixels: the small
    // heads (3x3 ... 1x1 maps) fill little of a 128-pixel tile, but what they cost is the walk over K = 9 x cin, and that is cut
    // into slices here like everybody else's (measured: wz_k_conv_group took 18.7 us for them at batch 1, 7 us at batch 8)
    return a.ksize == 3 && a.out_mode != WZ_OUT_ACT && a.zeros && a.n_pad % 64 == 0 && a.kc % 2 == 0 && a.cin % 32 == 0 &&
           a.kchunks >= 16;
}

void wz_conv_wide_shape(const WzConvArgs& a, int* tiles, int* steps) {
    const int live = (a.cout + 15) >> 4;
    const int per = 4 * wz_conv_wide_ntw();
    const int groups = (live + per - 1) / per;
    *tiles = ((a.M + WZ_WIDE_TM - 1) / WZ_WIDE_TM) * groups;
    *steps = a.kchunks >> 1;
}

// K slices for a set of convolutions that will share one launch: every workgroup walks at most T steps, T chosen by a small cost
// model -- the launch lasts as long as its busiest CU (longest-first list schedule over `cus` CUs, ~1 500 cycles per step), and
// every slice costs a trip of its fp32 partial tile to HBM and back (~1 250 bytes per cycle for write + read).
int wz_choose_wide_T(const int* tiles, const int* steps, const long long* tile_bytes, int n, int cus) {
    int max_steps = 0;
    for (int i = 0; i < n; ++i) max_steps = steps[i] > max_steps ? steps[i] : max_steps;
    long long best_cost = -1;
    int best_T = max_steps > 0 ? max_steps : 1;
    for (int T = 4; T <= max_steps; ++T) {
        long long wgs = 0, work = 0, bytes = 0;
        int longest = 0;
        for (int i = 0; i < n; ++i) {
            const int sk = (steps[i] + T - 1) / T;
            const int dur = (steps[i] + sk - 1) / sk;
            wgs += (long long)tiles[i] * sk;
            work += (long long)tiles[i] * sk * dur;
            bytes += (long long)tiles[i] * sk * tile_bytes[i];
            longest = dur > longest ? dur : longest;
        }
        long long span = (work + cus - 1) / cus;
        if (wgs > cus) span += longest / 2;   // a second, ragged round
        if (span < longest) span = longest;
        const long long cost = span * 1500 + 2 * bytes / 1250;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_T = T; }
    }
    return best_T;
}

int wz_conv_wide_group_add(WzConvGroup& g, const WzConvArgs& a0) {
    if (g.n >= WZ_CONV_GROUP_MAX) return 0;
    const int i = g.n++;
    WzConvArgs& a = g.a[i];
    a = a0;
    a.nt_live = (a.cout + 15) >> 4;
    a.grid_n = (a.nt_live + 4 * wz_conv_wide_ntw() - 1) / (4 * wz_conv_wide_ntw());
    a.nt_group = (a.nt_live + a.grid_n - 1) / a.grid_n;
    a.grid_m = (a.M + WZ_WIDE_TM - 1) / WZ_WIDE_TM;
    g.gx[i] = g.gy[i] = 0;
    g.first[i + 1] = g.first[i] + ((a.grid_m * a.grid_n * a.splitk + 7) & ~7);   // entries start on an XCD boundary
    return 1;
}

void wz_launch_conv_wide_group(const WzConvGroup& g, hipStream_t s) {
    if (wz_conv_wide_ntw() == 3)
        WZ_LAUNCH(wz_k_conv_wide_group3, dim3(g.first[g.n]), dim3(256), 0, s, g);
    else
        WZ_LAUNCH(wz_k_conv_wide_group, dim3(g.first[g.n]), dim3(256), 0, s, g);
}
