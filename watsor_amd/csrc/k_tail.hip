// The extras behind the 5x5 map -- six convolutions on 5x5 ... 1x1 maps, 6 MFLOP per frame -- in ONE launch, one workgroup per frame.
//
// Each of them alone is a launch at the floor of what a launch costs here (3.0 .. 4.6 us: a kernel boundary inside the captured
// graph, a cold start of ~5 000 cycles until the first operands have arrived, a few hundred cycles of work), and a frame's tensors
// are at most 25 KiB: a workgroup keeps them in LDS from layer to layer, so the chain needs no hand-over between workgroups (which
// would cost as much as the launches it replaces, tools/micro/xcd_barrier.hip).  What bounds the launch is the weights: every
// workgroup streams all 1.5 MB of them (the frames are independent and nothing is shared between CUs but L2).
//   * 8 waves; a layer's 16-channel output tiles are dealt out over the waves (two per wave where there are sixteen: both walk
//     K together, which doubles the loads in flight), fp32 accumulators for the <= 25 pixels = 2 pixel tiles;
//   * weights straight from L2 into registers in MFMA fragment order (the packed layout: one 1 KiB load per fragment), four K steps
//     per request group, the next group in flight under the MFMAs of the current one;
//   * activations as MFMA B fragments out of LDS: pixel-major rows of cin + 8 halves (the 16 bytes of padding spread the pixels
//     over the banks), filter taps outside the map read as zeros;
//   * epilogue: + bias, relu6, fp16 -- into the other LDS buffer for the next layer, and to HBM where `out` is set: the engine sets it
//     for the tensors somebody outside the chain reads (the 3x3 outputs are SSD feature maps), see the comment at the call site.
// Numerics: the products and the fp32 accumulation of the layer-by-layer kernels, summed in another order (K is not split here).
//
// MEASURED (profiles/r02t2_*, batch 8 and batch 1 alike): 33.6 us for the chain against 3.7 + 5.0 + 3.5 + 4.9 + 3.2 + 3.4 = 23.5 us
// as six launches -- a workgroup pulls its 1.5 MB of weights through ONE CU's vector memory path, 8 .. 24 KiB in flight per wave,
// where a layer's own launch spreads them over a hundred CUs.  Off by default (WZ_TAIL_FUSE=1 turns it on); kept because it is the
// measured answer to "why not one launch for the tail".
#include "wz_common.h"

#define WZ_TAIL_BUF (25 * (512 + 8) * 2 + 256)   // bytes per LDS buffer: the largest tensor of the chain (5x5x512) with padded rows

// KS: filter size (1: K = cin in groups of four 32-channel chunks, cin a multiple of 256; 3: one request group per filter tap, cin <= 128).
// NT2: 16-channel tiles per wave walked together (1 or 2).  `zpad`: 16 zero bytes in LDS (what a tap outside the map reads).
template <int KS, int NT2>
__device__ __forceinline__ void wz_tail_layer(const WzConvArgs& a, const int frame, const half_t* __restrict__ xin, half_t* __restrict__ xout,
                                              const half_t* __restrict__ zpad) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int taps = KS * KS;
    const int kc = a.kc, steps = taps * kc;
    const int M = a.hout * a.wout;
    const int ntiles = (a.cout + 15) >> 4;
    const int rs_in = a.cin + 8, rs_out = a.cout + 8;

    // the lane's two pixels: where their filter windows start
    int iy0[2], ix0[2];
    bool pv[2];
    const float rcp_w = 1.0f / (float)a.wout;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = mt * 16 + r16;
        pv[mt] = p < M;
        const int oy = (int)(((float)p + 0.5f) * rcp_w), ox = p - oy * a.wout;   // p < 32: exact
        iy0[mt] = oy * a.stride - a.pad_t;
        ix0[mt] = ox * a.stride - a.pad_l;
    }
    // LDS address of the lane's fragment of pixel mt at filter tap (ky, kx), chunk 0 -- or of the zero pad
    auto tap_ptr = [&](int mt, int ky, int kx) -> const half_t* {
        const int iy = iy0[mt] + ky, ix = ix0[mt] + kx;
        const bool ok = pv[mt] && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
        return ok ? xin + (iy * a.win + ix) * rs_in + g * 8 : zpad;
    };

    for (int nt0 = wave * NT2; nt0 < ntiles; nt0 += 8 * NT2) {
        float4_t acc[NT2][2];
#pragma unroll
        for (int j = 0; j < NT2; ++j) acc[j][0] = acc[j][1] = (float4_t){0.f, 0.f, 0.f, 0.f};
        const half_t* wl[NT2];
#pragma unroll
        for (int j = 0; j < NT2; ++j) wl[j] = a.w + ((size_t)min(nt0 + j, ntiles - 1) * steps * 64 + lane) * 8;   // fragment q of the tile: + q * 512

        // One request group = up to four K steps of every tile of the wave; three groups in flight.  No branches in the loop: the
        // steps of a group are a compile-time count, a tap outside the map reads the zero pad (a first version with conditional loads
        // was cut into ~300 basic blocks by the compiler, one wait each: 50 us for the chain).
        half8_t wa[3][NT2][4];
        if constexpr (KS == 3) {   // group = filter tap (kc <= 4 chunks: checked by wz_tail_layer_ok)
            auto request = [&](int t, half8_t (&f)[NT2][4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < NT2; ++j) f[j][u] = *reinterpret_cast<const half8_t*>(wl[j] + (size_t)(t * kc + min(u, kc - 1)) * 512);
            };
            auto work = [&](int t, const half8_t (&f)[NT2][4]) {
                const half_t* p0 = tap_ptr(0, t / 3, t % 3);
                const half_t* p1 = tap_ptr(1, t / 3, t % 3);
                const int s0 = p0 == zpad ? 0 : 32, s1 = p1 == zpad ? 0 : 32;   // halves per chunk (the zero pad does not move)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool live = u < kc;   // wave-uniform; a chunk past kc re-read the last fragment: it meets zeros (a pointer select, no branch)
                    const half8_t x0 = *reinterpret_cast<const half8_t*>(live ? p0 + u * s0 : zpad);
                    const half8_t x1 = *reinterpret_cast<const half8_t*>(live ? p1 + u * s1 : zpad);
#pragma unroll
                    for (int j = 0; j < NT2; ++j) {
                        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[j][u], x0, acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[j][u], x1, acc[j][1], 0, 0, 0);
                    }
                }
            };
            request(0, wa[0]);
            request(1, wa[1]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 2 < 9) request(t + 2, wa[(t + 2) % 3]);
                work(t, wa[t % 3]);
                __builtin_amdgcn_sched_barrier(0);   // (or the scheduler hoists all 72 LDS reads of the nine taps: 288 registers)
            }
        } else {                   // 1x1: K = kc chunks, kc a multiple of 8: two groups per loop trip
            const half_t* p0 = tap_ptr(0, 0, 0);
            const half_t* p1 = tap_ptr(1, 0, 0);
            const int s0 = p0 == zpad ? 0 : 32, s1 = p1 == zpad ? 0 : 32;   // halves per chunk step (the zero pad does not move)
            auto request = [&](int c0, half8_t (&f)[NT2][4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < NT2; ++j) f[j][u] = *reinterpret_cast<const half8_t*>(wl[j] + (size_t)min(c0 + u, kc - 1) * 512);
            };
            auto work = [&](int c0, const half8_t (&f)[NT2][4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const half8_t x0 = *reinterpret_cast<const half8_t*>(p0 + (c0 + u) * s0);
                    const half8_t x1 = *reinterpret_cast<const half8_t*>(p1 + (c0 + u) * s1);
#pragma unroll
                    for (int j = 0; j < NT2; ++j) {
                        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[j][u], x0, acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[j][u], x1, acc[j][1], 0, 0, 0);
                    }
                }
            };
            request(0, wa[0]);
            for (int c0 = 0; c0 < kc; c0 += 8) {
                request(c0 + 4, wa[1]);
                work(c0, wa[0]);
                __builtin_amdgcn_sched_barrier(0);
                request(c0 + 8, wa[0]);   // (past the end: the last fragment again, never used)
                work(c0 + 4, wa[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int n4 = (nt0 + j) * 16 + g * 4;
            if (nt0 + j >= ntiles || n4 >= a.cout) continue;
            const float4_t bv = *reinterpret_cast<const float4_t*>(a.bias + n4);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int p = mt * 16 + r16;
                if (p >= M) continue;
                float4_t v = acc[j][mt];
                half4_t h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = v[r] + bv[r];
                    if (a.act == WZ_ACT_RELU6) x = fminf(fmaxf(x, 0.0f), 6.0f);
                    h[r] = (half_t)x;
                }
                *reinterpret_cast<half4_t*>(xout + p * rs_out + n4) = h;
                if (a.out) *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(a.out) + ((size_t)frame * M + p) * a.cout + n4) = h;
            }
        }
    }
}

__global__ __launch_bounds__(512) void wz_k_extras_tail(const WzTailArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WZ_TAIL_BUF + 64];
    const int frame = blockIdx.x;
    half_t* const buf0 = reinterpret_cast<half_t*>(smem);
    half_t* const buf1 = reinterpret_cast<half_t*>(smem + WZ_TAIL_BUF);
    half_t* const zpad = reinterpret_cast<half_t*>(smem + 2 * WZ_TAIL_BUF);
    if (threadIdx.x < 32) zpad[threadIdx.x] = (half_t)0.0f;
    // Every weight of the chain is touched once up front (one dword per 128-byte line, all of a thread's loads in flight together):
    // the frames' workgroups sit on different XCDs, each L2 meets these 1.5 MB for the first time, and a K loop that met them one
    // request group at a time paid a trip to HBM per group (measured: 50 us for the chain against 24 us for six launches).
    unsigned sink = 0;
    {
        constexpr int PER = 10;   // lines per thread and layer: 16 tiles x 9 taps x 4 chunks x 8 lines / 512 threads = 9 (wz_tail_layer_ok keeps it there)
        unsigned got[WZ_TAIL_MAX][PER];
#pragma unroll
        for (int li = 0; li < WZ_TAIL_MAX; ++li) {
            const WzConvArgs& a = A.l[li < A.n ? li : 0];
            const int lines = ((a.cout + 15) >> 4) * a.ksize * a.ksize * a.kc * 8;   // 1 KiB fragments = 8 lines each
            const unsigned* w32 = reinterpret_cast<const unsigned*>(a.w);
#pragma unroll
            for (int k = 0; k < PER; ++k)   // (a thread without a line of its own re-touches the last one: no branch, every load issued before the first is used)
                got[li][k] = __builtin_nontemporal_load(w32 + (size_t)min((int)threadIdx.x + k * 512, lines - 1) * 32);
        }
#pragma unroll
        for (int li = 0; li < WZ_TAIL_MAX; ++li)
#pragma unroll
            for (int k = 0; k < PER; ++k) sink += got[li][k];
    }
    {   // the chain's input, this frame's pixels, into padded rows
        const WzConvArgs& a = A.l[0];
        const int npx = a.hin * a.win, c8s = a.cin >> 3, rs = a.cin + 8;
        const half_t* src = a.in + (size_t)frame * npx * a.cin;
        for (int i = threadIdx.x; i < npx * c8s; i += 512) {
            const int px = i / c8s, c8 = i - px * c8s;
            *reinterpret_cast<half8_t*>(buf0 + px * rs + c8 * 8) = *reinterpret_cast<const half8_t*>(src + (size_t)px * a.cin + c8 * 8);
        }
    }
    if (sink == 0x9e3779b9u && frame < 0) buf1[threadIdx.x] = (half_t)1.0f;   // (never true: keeps the touches alive)
    __syncthreads();
    for (int li = 0; li < A.n; ++li) {
        const WzConvArgs& a = A.l[li];
        const half_t* const xin = (li & 1) ? buf1 : buf0;
        half_t* const xout = (li & 1) ? buf0 : buf1;
        const bool two = ((a.cout + 15) >> 4) > 8;
        if (a.ksize == 3) {
            if (two) wz_tail_layer<3, 2>(a, frame, xin, xout, zpad);
            else wz_tail_layer<3, 1>(a, frame, xin, xout, zpad);
        } else {
            if (two) wz_tail_layer<1, 2>(a, frame, xin, xout, zpad);
            else wz_tail_layer<1, 1>(a, frame, xin, xout, zpad);
        }
        __syncthreads();
    }
}

// a convolution the chain kernel can take as one of its layers (per frame: <= 32 output pixels, everything in one LDS buffer)
bool wz_tail_layer_ok(const WzConvArgs& a, int n_frames) {
    if (n_frames < 1 || a.M % n_frames) return false;
    const int m = a.M / n_frames;
    return a.out_mode == WZ_OUT_ACT && !a.res && m == a.hout * a.wout && m <= 32 && a.cin % 32 == 0 && a.kc * 32 == a.cin && a.cout % 16 == 0 &&
           (size_t)a.hin * a.win * (a.cin + 8) * 2 <= WZ_TAIL_BUF && (size_t)m * (a.cout + 8) * 2 <= WZ_TAIL_BUF && a.splitk <= 1 &&
           (a.cout + 15) / 16 <= 16 && ((a.ksize == 3 && a.kc <= 4) || (a.ksize == 1 && a.kc % 8 == 0)) &&
           ((a.cout + 15) / 16) * a.ksize * a.ksize * a.kc * 8 <= 10 * 512;
}

void wz_launch_extras_tail(const WzTailArgs& A, int n_frames, hipStream_t s) {
    WZ_LAUNCH(wz_k_extras_tail, dim3(n_frames), dim3(512), 0, s, A);
}
