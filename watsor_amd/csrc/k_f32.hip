// The `-p 32` engine: the same op program (one op per layer) with fp32 activations and fp32 weights.
//
// The reference's TensorRT engine builder has the same switch (`watsor/engine.py:77-80`, `-p {32,16}`:
// fp16 is the fast mode, fp32 the accurate one).  The fp16 engine of k_conv.hip / k_mbconv.hip lands
// within 3e-3 of the fp32 CPU detector's scores on the seeded random-init network (every one of its ~53
// layers rounds weights and activations to 11 bits); this engine is the one that meets the north-star's
// 1e-3 (measured ~1e-5): exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32, 157 TFLOP/s peak = 1/16 of
// the fp16 rate), fp32 accumulation, nothing rounded below fp32 after the fp16 input tensor.
//
//  wz_k_stem_f32   3x3 s2 on the fp16 input tensor (uint8 -> fp16 normalise is part of the model's input
//                  contract), fp32 output
//  wz_k_dw_f32     depthwise 3x3, thread = (pixel, 4 channels)
//  wz_k_conv_f32   1x1 / 3x3 implicit GEMM: D[n][m] += W[n][k] X[m][k], weights = A operand, pre-packed so
//                  that lane (r16, g) reads W[n = r16][k0 + 4g .. 4g+3] as ONE float4; the activations are
//                  read the same way (pixel r16, channels k0 + 4g ..), and the four MFMAs of a 16-channel
//                  chunk consume component t of both float4s (k = k0 + 4g + t)
//  wz_k_splitk_reduce_f32
#include <stdlib.h>

#include "wz_common.h"

// pair: the input tensor holds (r, g, b, 0) hi then (r, g, b, 0) lo per pixel (wz_k_preprocess<true>): x = hi + lo, i.e. the resized
// image to ~22 significant bits instead of fp16's 11 -- the input rounding was this engine's largest error on weights with a wide
// per-channel dynamic range (tests/test_gpu_stress.py: 5e-4 of the scores)
__global__ __launch_bounds__(256) void wz_k_stem_f32(const half_t* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     int total, int hin, int win, int hout, int wout, int pad_t,
                                                     int pad_l, int pair) {
    __shared__ float sw[27 * 32 + 32];
    for (int i = threadIdx.x; i < 27 * 32; i += 256) sw[i] = w[i];
    if (threadIdx.x < 32) sw[27 * 32 + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if (tid >= total) return;
    const int cg = tid & 3, pix = tid >> 2;
    const int ox = pix % wout;
    const int t2 = pix / wout;
    const int oy = t2 % hout, b = t2 / hout;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = sw[27 * 32 + cg * 8 + j];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - pad_t + ky;
        if (iy < 0 || iy >= hin) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - pad_l + kx;
            if (ix < 0 || ix >= win) continue;
            const size_t px = (size_t)(b * hin + iy) * win + ix;
            const half4_t p = *reinterpret_cast<const half4_t*>(in + px * (pair ? 8 : 4));
            half4_t pl = {0, 0, 0, 0};
            if (pair) pl = *reinterpret_cast<const half4_t*>(in + px * 8 + 4);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float x = (float)p[c] + (float)pl[c];
                const float* wr = sw + ((ky * 3 + kx) * 3 + c) * 32 + cg * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(x, wr[j], acc[j]);
            }
        }
    }
    float4_t o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o0[j] = fminf(fmaxf(acc[j], 0.0f), 6.0f);
        o1[j] = fminf(fmaxf(acc[4 + j], 0.0f), 6.0f);
    }
    float* o = out + (size_t)pix * 32 + cg * 8;
    *reinterpret_cast<float4_t*>(o) = o0;
    *reinterpret_cast<float4_t*>(o + 4) = o1;
}

__global__ __launch_bounds__(256) void wz_k_dw_f32(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   int total, int hin, int win, int c, int hout, int wout,
                                                   int stride, int pad_t, int pad_l, int act) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if (tid >= total) return;
    const int c4 = c >> 2;
    const int cg = tid % c4, pix = tid / c4;
    const int ox = pix % wout;
    const int t2 = pix / wout;
    const int oy = t2 % hout, b = t2 / hout;
    float4_t acc = *reinterpret_cast<const float4_t*>(bias + cg * 4);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride - pad_t + ky;
        if (iy < 0 || iy >= hin) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride - pad_l + kx;
            if (ix < 0 || ix >= win) continue;
            const float4_t x = *reinterpret_cast<const float4_t*>(in + ((size_t)(b * hin + iy) * win + ix) * c + cg * 4);
            const float4_t k = *reinterpret_cast<const float4_t*>(w + (size_t)(ky * 3 + kx) * c + cg * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(x[j], k[j], acc[j]);
        }
    }
    if (act == WZ_ACT_RELU6)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fminf(fmaxf(acc[j], 0.0f), 6.0f);
    *reinterpret_cast<float4_t*>(out + (size_t)pix * c + cg * 4) = acc;
}

__device__ __forceinline__ void wz_epilogue4_f32(const WzConvArgs& a, int m, int n4, float4_t v) {
    if (m >= a.M || n4 >= a.cout) return;
    const float4_t bv = *reinterpret_cast<const float4_t*>(a.bias + n4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x = v[r] + bv[r];
        if (a.act == WZ_ACT_RELU6) x = fminf(fmaxf(x, 0.0f), 6.0f);
        v[r] = x;
    }
    if (a.out_mode == WZ_OUT_ACT) {
        const size_t o = (size_t)m * a.cout + n4;
        if (a.res) {
            const float4_t rv = *reinterpret_cast<const float4_t*>(reinterpret_cast<const float*>(a.res) + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
        *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(a.out) + o) = v;
    } else {
        const int hw = a.hout * a.wout;
        const int b = m / hw, pix = m - b * hw;
        float* o;
        int cols, n0;
        if (a.out_mode == WZ_OUT_HEAD && n4 >= a.n_box) {
            cols = a.cout - a.n_box;
            n0 = n4 - a.n_box;
            o = a.out2 + (size_t)b * a.out2_batch_stride + a.out2_off;
        } else {
            cols = (a.out_mode == WZ_OUT_HEAD) ? a.n_box : a.cout;
            n0 = n4;
            o = reinterpret_cast<float*>(a.out) + (size_t)b * a.out_batch_stride + a.out_off;
        }
        o += (size_t)pix * cols + n0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n0 + r < cols) o[r] = v[r];
    }
}

// wave = 32 pixels x 32 channels (2 x 2 MFMA tiles); a.kc = 16-channel chunks per tap, a.kchunks = taps * kc.
// Register double buffer of U chunks each: the loads of the next U chunks are in flight under the 16 U MFMAs
// (32 cycles each on a SIMD) of the current ones.
template <int U>
struct F32Frags {
    float4_t xb[U][2], wa[U][2];
};

template <int KS, int U>
__device__ __forceinline__ void wz_f32_load(const WzConvArgs& a, F32Frags<U>& f, int& ql, const int q1, int& t, int& c,
                                            const int (&iy0)[2], const int (&ix0)[2], const int (&boff)[2],
                                            const bool (&mv)[2], const float* in, const float* wlane, const int nt0,
                                            const int g) {
    constexpr int taps = KS * KS;
    const float4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool live = ql < q1;   // wave-uniform
        const int ky = (KS == 1) ? 0 : t / KS, kx = (KS == 1) ? 0 : t - ky * KS;
        const bool cin_ok = (c * 16 + g * 4) < a.cin;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int iy = iy0[mt] + ky, ix = ix0[mt] + kx;
            const bool ok = live && cin_ok && mv[mt] && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
            f.xb[u][mt] = ok ? *reinterpret_cast<const float4_t*>(in + ((size_t)(boff[mt] + iy) * a.win + ix) * a.cin + c * 16 + g * 4)
                             : zero;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            f.wa[u][nt] = live ? *reinterpret_cast<const float4_t*>(wlane + ((size_t)((nt0 + nt) * taps + t) * a.kc + c) * 256)
                               : zero;
        ++ql;
        if (++c == a.kc) {
            c = 0;
            ++t;
        }
    }
}

template <int U>
__device__ __forceinline__ void wz_f32_mfma(const F32Frags<U>& f, float4_t (&acc)[2][2]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.wa[u][nt][j], f.xb[u][mt][j], acc[mt][nt], 0, 0, 0);
}

template <int KS>
__global__ __launch_bounds__(256) void wz_k_conv_f32(const WzConvArgs a) {
    constexpr int MT = 2, NT = 2, U = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int m_base = (blockIdx.x * 4 + wave) * (MT * 16);
    const int nt0 = blockIdx.y * NT;
    if (m_base >= a.M) return;
    const float* in = reinterpret_cast<const float*>(a.in);
    const float* wlane = reinterpret_cast<const float*>(a.w) + lane * 4;

    const int hw = a.hout * a.wout;
    int iy0[MT], ix0[MT], boff[MT];
    bool mv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16 + r16;
        mv[mt] = m < a.M;
        const int mm = mv[mt] ? m : 0;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / a.wout, ox = rem - oy * a.wout;
        iy0[mt] = oy * a.stride - a.pad_t;
        ix0[mt] = ox * a.stride - a.pad_l;
        boff[mt] = b * a.hin;
    }
    float4_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int per = (a.kchunks + a.splitk - 1) / a.splitk;
    const int q0 = blockIdx.z * per, q1 = min(q0 + per, a.kchunks);
    int ql = q0, t = (KS == 1) ? 0 : q0 / a.kc, c = (KS == 1) ? q0 : q0 - t * a.kc;
    F32Frags<U> fa, fb;
    wz_f32_load<KS, U>(a, fa, ql, q1, t, c, iy0, ix0, boff, mv, in, wlane, nt0, g);
    for (int q = q0; q < q1;) {
        wz_f32_load<KS, U>(a, fb, ql, q1, t, c, iy0, ix0, boff, mv, in, wlane, nt0, g);
        wz_f32_mfma(fa, acc);
        q += U;
        if (q >= q1) break;
        wz_f32_load<KS, U>(a, fa, ql, q1, t, c, iy0, ix0, boff, mv, in, wlane, nt0, g);
        wz_f32_mfma(fb, acc);
        q += U;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16 + r16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n4 = (nt0 + nt) * 16 + g * 4;
            if (a.splitk > 1) {
                if (m < a.M)
                    *reinterpret_cast<float4_t*>(a.ws + ((size_t)blockIdx.z * a.M + m) * a.n_pad + n4) = acc[mt][nt];
            } else {
                wz_epilogue4_f32(a, m, n4, acc[mt][nt]);
            }
        }
    }
}

// Split-K inside the workgroup for the extras chain, fp32 engine (see wz_k_conv_ws in k_conv.hip): the 8 waves of a
// workgroup share one 32-pixel x 32-channel tile, each walks an eighth of the 16-channel chunks, the accumulators meet
// in LDS and are summed in wave order.
template <int KS>
__global__ __launch_bounds__(512) void wz_k_conv_ws_f32(const WzConvArgs a) {
    constexpr int U = 2, WAVES = 8;
    __shared__ float4_t red[WAVES][4][64];   // 32 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int m_base = blockIdx.x * 32;
    const int nt0 = blockIdx.y * 2;
    const float* in = reinterpret_cast<const float*>(a.in);
    const float* wlane = reinterpret_cast<const float*>(a.w) + lane * 4;

    const int hw = a.hout * a.wout;
    int iy0[2], ix0[2], boff[2];
    bool mv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m_base + mt * 16 + r16;
        mv[mt] = m < a.M;
        const int mm = mv[mt] ? m : 0;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / a.wout, ox = rem - oy * a.wout;
        iy0[mt] = oy * a.stride - a.pad_t;
        ix0[mt] = ox * a.stride - a.pad_l;
        boff[mt] = b * a.hin;
    }
    float4_t acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int per = (a.kchunks + WAVES - 1) / WAVES;
    const int q0 = wave * per, q1 = min(q0 + per, a.kchunks);
    if (q0 < q1) {
        int ql = q0, t = (KS == 1) ? 0 : q0 / a.kc, c = (KS == 1) ? q0 : q0 - t * a.kc;
        F32Frags<U> fa, fb;
        wz_f32_load<KS, U>(a, fa, ql, q1, t, c, iy0, ix0, boff, mv, in, wlane, nt0, g);
        for (int q = q0; q < q1;) {
            wz_f32_load<KS, U>(a, fb, ql, q1, t, c, iy0, ix0, boff, mv, in, wlane, nt0, g);
            wz_f32_mfma(fa, acc);
            q += U;
            if (q >= q1) break;
            wz_f32_load<KS, U>(a, fa, ql, q1, t, c, iy0, ix0, boff, mv, in, wlane, nt0, g);
            wz_f32_mfma(fb, acc);
            q += U;
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) red[wave][mt * 2 + nt][lane] = acc[mt][nt];
    __syncthreads();
    if (wave < 4) {
        float4_t v = red[0][wave][lane];
#pragma unroll
        for (int z = 1; z < WAVES; ++z) {
            const float4_t p = red[z][wave][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += p[r];
        }
        wz_epilogue4_f32(a, m_base + (wave / 2) * 16 + r16, (nt0 + wave % 2) * 16 + g * 4, v);
    }
}

bool wz_conv_ws_f32_applies(const WzConvArgs& a) {
    static const bool on = [] { const char* e = wz_dev_getenv("WZ_CONV_WS"); return !(e && atoi(e) == 0); }();
    return on && a.out_mode == WZ_OUT_ACT && a.M <= 1024 && a.kchunks >= 16 && a.n_pad % 32 == 0;
}
void wz_launch_conv_ws_f32(const WzConvArgs& a, hipStream_t s) {
    const dim3 grid((a.M + 31) / 32, a.n_pad / 32);
    if (a.ksize == 1)
        WZ_LAUNCH(wz_k_conv_ws_f32<1>, grid, dim3(512), 0, s, a);
    else
        WZ_LAUNCH(wz_k_conv_ws_f32<3>, grid, dim3(512), 0, s, a);
}

__global__ __launch_bounds__(256) void wz_k_splitk_reduce_f32(const WzConvArgs a) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int n4s = a.n_pad >> 2;
    if (tid >= a.M * n4s) return;
    const int m = tid / n4s, n4 = (tid - m * n4s) * 4;
    float4_t v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < a.splitk; ++z) {
        const float4_t p = *reinterpret_cast<const float4_t*>(a.ws + ((size_t)z * a.M + m) * a.n_pad + n4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p[r];
    }
    wz_epilogue4_f32(a, m, n4, v);
}

void wz_launch_stem_f32(const half_t* in, const float* w, const float* bias, float* out, int n, int hin, int win,
                        int hout, int wout, int pad_t, int pad_l, hipStream_t s, bool pair) {
    const int total = n * hout * wout * 4;
    WZ_LAUNCH(wz_k_stem_f32, dim3((total + 255) / 256), dim3(256), 0, s, in, w, bias, out, total, hin, win,
                       hout, wout, pad_t, pad_l, pair ? 1 : 0);
}

void wz_launch_dw_f32(const float* in, const float* w, const float* bias, float* out, int n, int hin, int win, int c,
                      int hout, int wout, int stride, int pad_t, int pad_l, int act, hipStream_t s) {
    const int total = n * hout * wout * (c >> 2);
    WZ_LAUNCH(wz_k_dw_f32, dim3((total + 255) / 256), dim3(256), 0, s, in, w, bias, out, total, hin, win, c,
                       hout, wout, stride, pad_t, pad_l, act);
}

// ---- the register-staged LDS-tiled kernel (k_conv_rs.h) in fp32: the long-K convolutions
#include "k_conv_rs.h"

struct WzEpiF32 {
    static __device__ __forceinline__ void apply(const WzConvArgs& a, int m, int n4, float4_t v) { wz_epilogue4_f32(a, m, n4, v); }
    static __device__ __forceinline__ float* partials(const WzConvArgs& a) { return a.ws; }
};

template <int KS, int NW>
__global__ __launch_bounds__(256, 2) void wz_k_conv_rs_f32(const WzConvArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];
    wz_conv_rs_body<KS, NW, false, true, WzEpiF32>(a, smem, blockIdx.x);
}

// whole 64-column tiles, whole 32-channel K steps, enough pixels and a K loop long enough to pay for the tile
bool wz_conv_f32_use_rs(const WzConvArgs& a) {
    static const bool on = [] { const char* e = wz_dev_getenv("WZ_F32_RS"); return !(e && atoi(e) == 0); }();
    // (channel tiles past n_pad read zeros and are never stored: n_pad need not be a multiple of the tile)
    return on && a.kc % 2 == 0 && a.cin % 32 == 0 && a.M >= 128 && a.kchunks >= 8;
}

// This variant is MFMA-bound (4 096 cycles of fp32 MFMA per step against ~1 000 of loads and LDS traffic).  Measured
// targets 256 / 512 / 768 workgroups: 12.2 / 12.1 / 11.9 k frames/s -- a second workgroup per CU does not pay.
int wz_choose_splitk_rs_f32(int M, int n_pad, int kchunks) {
    static const int target = [] { const char* e = wz_dev_getenv("WZ_F32_WGS"); return (e && atoi(e) > 0) ? atoi(e) : 256; }();
    const int tn = wz_lds_nw(M, n_pad, kchunks) * 32;
    const int wgs = ((M + WZ_RS_TM - 1) / WZ_RS_TM) * ((n_pad + tn - 1) / tn);
    const int nsteps = kchunks / 2;
    int s = (target + wgs - 1) / wgs;
    if (s > nsteps / 4) s = nsteps / 4;   // >= 4 steps per split
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
}

// a.splitk > 1: partials go to a.ws and the reduce kernel is enqueued right behind
void wz_launch_conv_f32(const WzConvArgs& a0, hipStream_t s, bool reduce) {
    if (wz_conv_f32_use_rs(a0)) {
        WzConvArgs a = a0;
        const int nw = wz_lds_nw(a.M, a.n_pad, a.kchunks);
        a.grid_m = (a.M + WZ_RS_TM - 1) / WZ_RS_TM;
        a.grid_n = (a.n_pad + nw * 32 - 1) / (nw * 32);
        a.order = 0;
        const dim3 grid(a.grid_m * a.grid_n * a.splitk);
        if (nw == 4) {
            if (a.ksize == 1)
                WZ_LAUNCH((wz_k_conv_rs_f32<1, 4>), grid, dim3(256), 0, s, a);
            else
                WZ_LAUNCH((wz_k_conv_rs_f32<3, 4>), grid, dim3(256), 0, s, a);
        } else {
            if (a.ksize == 1)
                WZ_LAUNCH((wz_k_conv_rs_f32<1, 2>), grid, dim3(256), 0, s, a);
            else
                WZ_LAUNCH((wz_k_conv_rs_f32<3, 2>), grid, dim3(256), 0, s, a);
        }
        if (a.splitk > 1 && reduce) {
            const int total = a.M * (a.n_pad >> 2);
            WZ_LAUNCH(wz_k_splitk_reduce_f32, dim3((total + 255) / 256), dim3(256), 0, s, a);
        }
        return;
    }
    const WzConvArgs& a = a0;
    const int mtiles = (a.M + 31) / 32;
    dim3 grid((mtiles + 3) / 4, a.n_pad / 32, a.splitk);
    if (a.ksize == 1)
        WZ_LAUNCH(wz_k_conv_f32<1>, grid, dim3(256), 0, s, a);
    else
        WZ_LAUNCH(wz_k_conv_f32<3>, grid, dim3(256), 0, s, a);
    if (a.splitk > 1 && reduce) {
        const int total = a.M * (a.n_pad >> 2);
        WZ_LAUNCH(wz_k_splitk_reduce_f32, dim3((total + 255) / 256), dim3(256), 0, s, a);
    }
}
