// Convolution stack of SSD-MobileNet-v2 on gfx950: activations NHWC fp16 in HBM, fp32 accumulate.
//
// In the reference all of this is inside `sess.run` (`watsor/detection/tensorflow_cpu.py:114-115`)
// or inside the TensorRT engine (`tensorrt_gpu.py:150`); there is no reference kernel to mirror
// (SURVEY.md §2 "Native inventory").  Layer shapes: SURVEY.md Appendix A.
//
//  wz_k_stem   3x3 s2, 3(+1 pad)->32, BN folded, ReLU6            VALU, K = 27
//  wz_k_dw     depthwise 3x3 s1/s2, TF SAME, BN folded, ReLU6      VALU, HBM-bound (AI 1.8-4.5 flop/B)
//  wz_k_conv   1x1 and dense 3x3 as implicit GEMM on v_mfma_f32_16x16x32_f16:
//              D[n][m] = sum_k W[n][k] * X[m][k]; the weights are the MFMA "A" operand (pre-packed by
//              the engine builder in fragment order, one coalesced 1 KiB load per fragment) and the
//              activations the "B" operand (each lane gathers 8 consecutive channels = 16 B of one
//              input pixel), so a lane ends up with 4 consecutive output channels of one pixel and
//              stores them as one 8-byte word.  Epilogue fuses bias, ReLU6 and the residual add.
//  wz_k_splitk_reduce  deterministic (fixed order) reduction of split-K partials + the same epilogue.
#include "wz_common.h"

// --------------------------------------------------------------------------------------------
// stem: thread = (output pixel, group of 8 output channels)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wz_k_stem(const half_t* __restrict__ in, const float* __restrict__ w,
                                                 const float* __restrict__ bias, half_t* __restrict__ out,
                                                 int total, int hin, int win, int hout, int wout, int pad_t,
                                                 int pad_l) {
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
            const half4_t p = *reinterpret_cast<const half4_t*>(in + ((size_t)(b * hin + iy) * win + ix) * 4);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float x = (float)p[c];
                const float* wr = sw + ((ky * 3 + kx) * 3 + c) * 32 + cg * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(x, wr[j], acc[j]);
            }
        }
    }
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)fminf(fmaxf(acc[j], 0.0f), 6.0f);
    *reinterpret_cast<half8_t*>(out + (size_t)pix * 32 + cg * 8) = o;
}

void wz_launch_stem(const half_t* in, const float* w, const float* bias, half_t* out, int n, int hin, int win,
                    int hout, int wout, int pad_t, int pad_l, hipStream_t s) {
    const int total = n * hout * wout * 4;
    WZ_LAUNCH(wz_k_stem, dim3((total + 255) / 256), dim3(256), 0, s, in, w, bias, out, total, hin, win,
                       hout, wout, pad_t, pad_l);
}

// --------------------------------------------------------------------------------------------
// depthwise 3x3: thread = (output pixel, group of 8 channels); 16-byte loads/stores, channel-coalesced
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wz_k_dw(const half_t* __restrict__ in, const half_t* __restrict__ w,
                                               const float* __restrict__ bias, half_t* __restrict__ out,
                                               int total, int hin, int win, int c, int hout, int wout, int stride,
                                               int pad_t, int pad_l, int act) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if (tid >= total) return;
    const int c8 = c >> 3;
    const int cg = tid % c8, pix = tid / c8;
    const int ox = pix % wout;
    const int t2 = pix / wout;
    const int oy = t2 % hout, b = t2 / hout;
    float acc[8];
    {
        const float4_t b0 = *reinterpret_cast<const float4_t*>(bias + cg * 8);
        const float4_t b1 = *reinterpret_cast<const float4_t*>(bias + cg * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = b0[j]; acc[4 + j] = b1[j]; }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride - pad_t + ky;
        if (iy < 0 || iy >= hin) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride - pad_l + kx;
            if (ix < 0 || ix >= win) continue;
            const half8_t x = *reinterpret_cast<const half8_t*>(in + ((size_t)(b * hin + iy) * win + ix) * c + cg * 8);
            const half8_t k = *reinterpret_cast<const half8_t*>(w + (size_t)(ky * 3 + kx) * c + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf((float)x[j], (float)k[j], acc[j]);
        }
    }
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = acc[j];
        if (act == WZ_ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
        o[j] = (half_t)v;
    }
    *reinterpret_cast<half8_t*>(out + (size_t)pix * c + cg * 8) = o;
}

void wz_launch_dw(const half_t* in, const half_t* w, const float* bias, half_t* out, int n, int hin, int win,
                  int c, int hout, int wout, int stride, int pad_t, int pad_l, int act, hipStream_t s) {
    const int total = n * hout * wout * (c >> 3);
    WZ_LAUNCH(wz_k_dw, dim3((total + 255) / 256), dim3(256), 0, s, in, w, bias, out, total, hin, win, c,
                       hout, wout, stride, pad_t, pad_l, act);
}

// --------------------------------------------------------------------------------------------
// implicit-GEMM convolution on MFMA.  Workgroup = 4 waves; wave = (MT*16 pixels) x (NT*16 channels).
// --------------------------------------------------------------------------------------------

__device__ __forceinline__ void wz_epilogue4(const WzConvArgs& a, int m, int n4, float4_t v) {
    // v = 4 consecutive output channels n4..n4+3 of output pixel m (bias not yet added)
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
            const half4_t rv = *reinterpret_cast<const half4_t*>(a.res + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(a.out) + o) = h;
    } else {
        const int hw = a.hout * a.wout;
        const int b = m / hw, pix = m - b * hw;
        float* o;
        int cols, n0;
        if (a.out_mode == WZ_OUT_HEAD && n4 >= a.n_box) {   // n_box is a multiple of 4: no group straddles
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

// The reductions of several convolutions in one launch: a workgroup finds its entry from the prefix table, then does
// exactly what wz_k_splitk_reduce does (same order over the splits: bit-identical results).
// box decode + clip of one anchor from its finished encoding (ty, tx, th, tw): the arithmetic of wz_k_decode
// (k_post.hip, which is compiled without contraction -- hence the pragma), operation for operation
__device__ __forceinline__ void wz_decode_anchor(const float4_t e, const float4_t an, const WzPostConsts& k,
                                                 float* __restrict__ box_out, uint8_t* __restrict__ valid_out) {
#pragma clang fp contract(off)
    const float ty = e[0] / k.scale_y, tx = e[1] / k.scale_x, th = e[2] / k.scale_h, tw = e[3] / k.scale_w;
    const float w = expf(tw) * an[3];
    const float h = expf(th) * an[2];
    const float yc = ty * an[2] + an[0];
    const float xc = tx * an[3] + an[1];
    const float hh = h / 2.0f, hw = w / 2.0f;
    float ymin = yc - hh, xmin = xc - hw, ymax = yc + hh, xmax = xc + hw;
    if (k.clip_after) {   // the NMS takes the boxes as decoded; every anchor is a candidate (wz_k_nms clips what it keeps)
        *reinterpret_cast<float4_t*>(box_out) = (float4_t){ymin, xmin, ymax, xmax};
        *valid_out = 1;
        return;
    }
    ymin = fminf(fmaxf(ymin, 0.0f), 1.0f);
    xmin = fminf(fmaxf(xmin, 0.0f), 1.0f);
    ymax = fminf(fmaxf(ymax, 0.0f), 1.0f);
    xmax = fminf(fmaxf(xmax, 0.0f), 1.0f);
    const float area = (ymax - ymin) * (xmax - xmin);
    *reinterpret_cast<float4_t*>(box_out) = (float4_t){ymin, xmin, ymax, xmax};
    *valid_out = area > 0.0f ? 1 : 0;
}

// Everything that happens to four finished output columns n4 .. n4+3 of pixel m of an SSD head (v = the K sum, bias not yet
// added): the epilogue proper (bias, store into the box-encoding / class-logit buffers), and optionally the candidate
// marking for wz_k_nms and the box decode.  Shared by the grouped reduce launch and by the in-launch reduction.
__device__ __forceinline__ void wz_head_finish(const WzConvArgs& a, int m, int n4, float4_t v, bool list, bool decode,
                                               const float* __restrict__ hint_logit, uint32_t* __restrict__ cbits,
                                               int cbits_words, const WzPostConsts& pc, const float* __restrict__ anchors,
                                               float* __restrict__ boxes, uint8_t* __restrict__ valid) {
    wz_epilogue4(a, m, n4, v);
    if (m >= a.M) return;
    const int n_box = a.out_mode == WZ_OUT_HEAD ? a.n_box : (a.out_mode == WZ_OUT_BOX ? a.cout : 0);
    if (list && n4 >= n_box && n4 < a.cout && a.out_mode != WZ_OUT_BOX && a.out_mode != WZ_OUT_ACT) {
        // class logits of one anchor location, four at a time: the ones that can reach the frame's first score band
        // are listed for wz_k_nms (which re-derives score, validity and bin exactly as its own scan would)
        const int hw = a.hout * a.wout;
        const int b = m / hw, pix = m - b * hw;
        const int cols = a.cout - n_box, n0 = n4 - n_box;
        const long long off = a.out_mode == WZ_OUT_HEAD ? a.out2_off : a.out_off;
        const float lf = hint_logit[b];
        const float4_t bv = *reinterpret_cast<const float4_t*>(a.bias + n4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = v[r] + bv[r];   // what the epilogue stored
            if (n0 + r < cols && x >= lf) {
                const long long j = off + (long long)pix * cols + n0 + r;   // entry index within the frame's logits
                atomicOr(&cbits[(size_t)b * cbits_words + (size_t)(j >> 5)], 1u << (j & 31));   // result unused
            }
        }
    }
    if (decode && n4 < n_box) {   // columns n4 .. n4+3 = the encoding of anchor (pixel, n4 / 4)
        const float4_t bv = *reinterpret_cast<const float4_t*>(a.bias + n4);
        float4_t enc;
#pragma unroll
        for (int r = 0; r < 4; ++r) enc[r] = v[r] + bv[r];   // what the epilogue stored
        const int hw = a.hout * a.wout;
        const int b = m / hw, pix = m - b * hw;
        const int anchor = (int)(a.out_off >> 2) + pix * (n_box >> 2) + (n4 >> 2);
        const size_t i = (size_t)b * pc.num_anchors + anchor;
        wz_decode_anchor(enc, *reinterpret_cast<const float4_t*>(anchors + (size_t)anchor * 4), pc, boxes + i * 4, valid + i);
    }
}

// ---- in-launch split-K reduction (see WzConvArgs::inline_reduce) ----------------------------------------------------
// publish: write-through (sc1) stores, then the issuing wave waits for them (the compiler knows nothing of asm stores)
__device__ __forceinline__ void wz_store_partial_sc1(float* p, const float4_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void wz_wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the head's finish for the summed K slices of (m, n4) (summed in slice order: wz_k_splitk_reduce's arithmetic)
__device__ __forceinline__ void wz_inline_finish(const WzConvArgs& a, int m, int n4, const float4_t v) {
    if (m >= a.M || n4 >= a.n_pad) return;
    const WzHeadFinish& f = *a.fin;
    wz_head_finish(a, m, n4, v, (a.fin_flags & 2) != 0, (a.fin_flags & 1) != 0, f.hint_logit, f.cbits, f.cbits_words, f.pc,
                   f.anchors, f.boxes, f.valid);
}


// Tile configuration: a wave computes (MT*16 pixels) x (NT*16 channels); U K-chunks (32 channels each)
// per register buffer, two buffers in flight.
//   <2,2,4>  small layers / small N: many waves, 1 KiB of loads per MFMA
//   <4,4,2>  the 3x3 SSD heads (M >= 128, N >= 256, K in the thousands): 4x the L2->register reuse
template <int MT, int NT, int U>
struct ConvFrags {
    half8_t xa[U][MT];
    half8_t wf[U][NT];
};

// Issue the loads of the next U K-chunks (flattened index ql = tap * kc + c) into `f`.
// (t, c) walk the chunk order; chunks at or beyond q1 load nothing and contribute zeros.
template <int KS, int MT, int NT, int U>
__device__ __forceinline__ void wz_conv_load(const WzConvArgs& a, ConvFrags<MT, NT, U>& f, int& ql, const int q1, int& t,
                                             int& c, const int (&iy0)[MT], const int (&ix0)[MT],
                                             const int (&boff)[MT], const bool (&mv)[MT],
                                             const half_t* wlane, const int nt0, const int cg8) {
    constexpr int taps = KS * KS;
    const half8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool live = ql < q1;   // wave-uniform
        const int ky = (KS == 1) ? 0 : t / KS, kx = (KS == 1) ? 0 : t - ky * KS;
        const bool cin_ok = (c * 32 + cg8) < a.cin;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int iy = iy0[mt] + ky, ix = ix0[mt] + kx;
            const bool ok = live && cin_ok && mv[mt] && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
            f.xa[u][mt] = ok ? *reinterpret_cast<const half8_t*>(
                                   a.in + ((size_t)(boff[mt] + iy) * a.win + ix) * a.cin + c * 32 + cg8)
                             : zero;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            f.wf[u][nt] = live ? *reinterpret_cast<const half8_t*>(
                                     wlane + ((size_t)((nt0 + nt) * taps + t) * a.kc + c) * 512)
                               : zero;
        ++ql;
        if (++c == a.kc) {
            c = 0;
            ++t;
        }
    }
}

template <int MT, int NT, int U>
__device__ __forceinline__ void wz_conv_mfma(const ConvFrags<MT, NT, U>& f, float4_t (&acc)[MT][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wf[u][nt], f.xa[u][mt], acc[mt][nt], 0, 0, 0);
}

template <int KS, int MT, int NT, int U>
__device__ __forceinline__ void wz_conv_body(const WzConvArgs& a, int bx, int by, int bz) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int m_base = (bx * 4 + wave) * (MT * 16);
    const int nt0 = by * NT;
    if (m_base >= a.M) return;   // whole wave out of range (no barriers in this kernel)

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

    // K range of this split (flattened chunk index q = tap * kc + c)
    const int per = (a.kchunks + a.splitk - 1) / a.splitk;
    const int q0 = bz * per;
    const int q1 = min(q0 + per, a.kchunks);
    const half_t* wlane = a.w + (size_t)lane * 8;
    const int cg8 = g * 8;

    // software pipeline: while the MFMAs of one register buffer run, the loads of the other are in flight
    int ql = q0, t = (KS == 1) ? 0 : q0 / a.kc, c = (KS == 1) ? q0 : q0 - t * a.kc;
    ConvFrags<MT, NT, U> fa, fb;
    wz_conv_load<KS, MT, NT, U>(a, fa, ql, q1, t, c, iy0, ix0, boff, mv, wlane, nt0, cg8);
    for (int q = q0; q < q1;) {
        wz_conv_load<KS, MT, NT, U>(a, fb, ql, q1, t, c, iy0, ix0, boff, mv, wlane, nt0, cg8);
        wz_conv_mfma(fa, acc);
        q += U;
        if (q >= q1) break;
        wz_conv_load<KS, MT, NT, U>(a, fa, ql, q1, t, c, iy0, ix0, boff, mv, wlane, nt0, cg8);
        wz_conv_mfma(fb, acc);
        q += U;
    }

    // D layout: lane holds rows (n) g*4..g*4+3 of column (m) r16
    if (a.splitk > 1 && a.inline_reduce) {
        // in-launch reduction, one WAVE tile at a time (the waves of this kernel never meet): publish, take a ticket,
        // and the wave that finds the other K slices already there sums them all in slice order and finishes
        float* const ws = a.ws;   // (a.out is where the FINISHED columns go)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m_base + mt * 16 + r16;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (m < a.M) wz_store_partial_sc1(ws + ((size_t)bz * a.M + m) * a.n_pad + (nt0 + nt) * 16 + g * 4, acc[mt][nt]);
        }
        wz_wait_stores();
        int32_t* const tk = a.tickets + ((int)(by * a.grid_m + bx) * 4 + (int)(threadIdx.x >> 6));
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t != a.splitk - 1) return;
        if (lane == 0) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < a.splitk; ++z) {   // a slice's fragments are all requested before the first is added
            float4_t pz[MT][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m_base + mt * 16 + r16;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    pz[mt][nt] = m < a.M ? *reinterpret_cast<const float4_t*>(ws + ((size_t)z * a.M + m) * a.n_pad + (nt0 + nt) * 16 + g * 4)
                                         : (float4_t){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][nt][r] += pz[mt][nt][r];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wz_inline_finish(a, m_base + mt * 16 + r16, (nt0 + nt) * 16 + g * 4, acc[mt][nt]);
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16 + r16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n4 = (nt0 + nt) * 16 + g * 4;
            if (a.splitk > 1) {
                if (m < a.M)
                    *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(a.out) +
                                                 ((size_t)bz * a.M + m) * a.n_pad + n4) = acc[mt][nt];
            } else {
                wz_epilogue4(a, m, n4, acc[mt][nt]);
            }
        }
    }
}

template <int KS, int MT, int NT, int U>
__global__ __launch_bounds__(256) void wz_k_conv(const WzConvArgs a) {
    WZ_LANE_STAMP(a.dbg);
    wz_conv_body<KS, MT, NT, U>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Split-K inside the workgroup, for the convolutions of the SSD extras chain (10x10 ... 1x1 maps, K up to 2 304, each
// feeding the next one): the 8 waves of a workgroup share ONE 32-pixel x 32-channel tile and each walks an eighth of
// the K chunks; the eight accumulator sets meet in LDS and are summed in wave order (deterministic).  No partial sums
// in HBM and no reduce launch behind the convolution -- on this chain a kernel boundary costs as much as the kernel.
// WM x WN: the workgroup's tile is WM x WN such 32 x 32 tiles and K is cut 8 / (WM WN) ways; U: K chunks per load group (two groups in flight);
// WGS_PER_CU = 2 holds the kernel to 128 registers (U = 2: 109) so that two workgroups share a CU (U = 4: 173 registers, a CU per workgroup).
template <int KS, int WM = 1, int WN = 1, int U = 2, int WGS_PER_CU = 2>
__global__ __launch_bounds__(512, WGS_PER_CU) void wz_k_conv_ws(const WzConvArgs a) {
    WZ_LANE_STAMP(a.dbg);
    constexpr int MT = 2, NT = 2, WAVES = 8, KSPL = WAVES / (WM * WN);
    static_assert(KSPL * WM * WN == WAVES, "eight waves: sub-tiles x K slices");
    __shared__ float4_t red[WAVES][MT * NT][64];   // 32 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int sub = wave / KSPL, ks = wave - sub * KSPL;   // which 32 x 32 tile of the workgroup's, which K slice of it
    const int m_base = ((int)blockIdx.x * WM + sub / WN) * (MT * 16);
    const int nt0 = ((int)blockIdx.y * WN + sub % WN) * NT;
    const bool tile_live = m_base < a.M && nt0 * 16 < a.n_pad;   // (wave-uniform; a dead tile still takes the barrier)

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

    const int per = (a.kchunks + KSPL - 1) / KSPL;
    const int q0 = ks * per, q1 = min(q0 + per, a.kchunks);
    if (tile_live && q0 < q1) {
        const half_t* wlane = a.w + (size_t)lane * 8;
        const int cg8 = g * 8;
        int ql = q0, t = (KS == 1) ? 0 : q0 / a.kc, c = (KS == 1) ? q0 : q0 - t * a.kc;
        ConvFrags<MT, NT, U> fa, fb;
        wz_conv_load<KS, MT, NT, U>(a, fa, ql, q1, t, c, iy0, ix0, boff, mv, wlane, nt0, cg8);
        for (int q = q0; q < q1;) {
            wz_conv_load<KS, MT, NT, U>(a, fb, ql, q1, t, c, iy0, ix0, boff, mv, wlane, nt0, cg8);
            wz_conv_mfma(fa, acc);
            q += U;
            if (q >= q1) break;
            wz_conv_load<KS, MT, NT, U>(a, fa, ql, q1, t, c, iy0, ix0, boff, mv, wlane, nt0, cg8);
            wz_conv_mfma(fb, acc);
            q += U;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) red[wave][mt * NT + nt][lane] = acc[mt][nt];
    __syncthreads();
    // K slice ks of a tile finishes its 16 x 16 sub-tiles ks, ks + KSPL, ...: the sum over the slices in slice order, then the epilogue
    // (eight slices: waves 0 .. 3 one sub-tile each, as before; two slices: two each)
    if (tile_live) {
        for (int tl = ks; tl < MT * NT; tl += KSPL) {
            float4_t v = red[sub * KSPL][tl][lane];
#pragma unroll
            for (int z = 1; z < KSPL; ++z) {
                const float4_t p = red[sub * KSPL + z][tl][lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += p[r];
            }
            wz_epilogue4(a, m_base + (tl / NT) * 16 + r16, (nt0 + tl % NT) * 16 + g * 4, v);
        }
    }
}

// Several small 3x3 convolutions that do not depend on each other (the SSD heads on the 3x3 ... 1x1 maps) in one launch:
// a workgroup finds its entry from the prefix table and runs wz_k_conv's body on it.
__global__ __launch_bounds__(256) void wz_k_conv_group(const WzConvGroup g) {
    WZ_LANE_STAMP(g.stamp);
    int e = 0;
    while (e + 1 < g.n && (int)blockIdx.x >= g.first[e + 1]) ++e;   // wave-uniform
    const int L = (int)blockIdx.x - g.first[e];
    const int gx = g.gx[e], gy = g.gy[e];
    wz_conv_body<3, 2, 2, 4>(g.a[e], L % gx, (L / gx) % gy, L / (gx * gy));
}

__global__ __launch_bounds__(256) void wz_k_splitk_reduce(const WzConvArgs a, const float* __restrict__ ws) {
    WZ_LANE_STAMP(a.dbg);
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int n4s = a.n_pad >> 2;
    if (tid >= a.M * n4s) return;
    const int m = tid / n4s, n4 = (tid - m * n4s) * 4;
    float4_t v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < a.splitk; ++z) {
        const float4_t p = *reinterpret_cast<const float4_t*>(ws + ((size_t)z * a.M + m) * a.n_pad + n4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p[r];
    }
    wz_epilogue4(a, m, n4, v);
}

// The reductions of several convolutions in one launch: a workgroup finds its entry from the prefix table, then does
// exactly what wz_k_splitk_reduce does (same order over the splits: bit-identical results).
__global__ __launch_bounds__(256) void wz_k_splitk_reduce_group(const WzReduceGroup g) {
    WZ_LANE_STAMP(g.stamp);
    if (g.decode && !g.list) {   // first kernel of the histogram-based post chain: clear its per-frame scratch
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < g.n_frames * WZ_HIST_BINS) g.hist[i] = 0u;
        if (i < g.n_frames) g.count[i] = 0u;
        if (i < 2 * g.n_frames) g.band[i] = 0u;
    }
    int e = 0;
    while (e + 1 < g.n && (int)blockIdx.x >= g.first[e + 1]) ++e;   // wave-uniform
    const WzConvArgs& a = g.a[e];
    const float* __restrict__ ws = g.ws[e];
    const int tid = ((int)blockIdx.x - g.first[e]) * 256 + threadIdx.x;
    int m, n4;
    size_t off, zstride;
    if (a.frag_ws) {   // partials in fragment order (WzConvArgs::frag_ws): a wavefront reads one 1 KiB fragment per slice
        const int mtt = (a.M + 15) >> 4, ntt = a.n_pad >> 4;
        const int frag = tid >> 6, lane = tid & 63;
        if (frag >= mtt * ntt) return;
        const int fm = frag / ntt;
        m = fm * 16 + (lane & 15);
        n4 = (frag - fm * ntt) * 16 + (lane >> 4) * 4;
        if (m >= a.M) return;
        off = (size_t)frag * 256 + lane * 4;
        zstride = (size_t)mtt * ntt * 256;
    } else {
        const int n4s = a.n_pad >> 2;
        if (tid >= a.M * n4s) return;
        m = tid / n4s;
        n4 = (tid - m * n4s) * 4;
        off = (size_t)m * a.n_pad + n4;
        zstride = (size_t)a.M * a.n_pad;
    }
    if (n4 >= a.cout) return;   // padding columns: nothing is stored for them (and the wide tile kernel does not write their partials)
    // the slices in groups of four: a group's loads are all requested before the first add (one memory latency per four slices instead of one
    // per slice -- the launch is bound by exactly that chain); the adds stay in slice order 0, 1, 2 ...: bit-identical sums
    float4_t v = {0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < a.splitk; z0 += 4) {
        float4_t p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            p[k] = z0 + k < a.splitk ? *reinterpret_cast<const float4_t*>(ws + (size_t)(z0 + k) * zstride + off) : (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (z0 + k < a.splitk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += p[k][r];
            }
    }
    wz_head_finish(a, m, n4, v, g.list != 0, g.decode != 0, g.hint_logit, g.cbits, g.cbits_words, g.pc, g.anchors, g.boxes, g.valid);
}

// --------------------------------------------------------------------------------------------
// LDS-tiled implicit GEMM for the layers with a long K loop (the 3x3 SSD heads, the 3x3 extras, Conv_1):
// workgroup = 128 pixels x 64 channels, K step = 64 (two MFMA K chunks), 4 waves as 2 (pixels) x 2
// (channels), wave tile 64 x 32.  Both operands go global -> LDS with `global_load_lds_dwordx4`
// (no staging registers): the LDS image of a tile is a list of 1 KiB MFMA fragments, lane l's 16 bytes
// at l*16, which is exactly what the DMA writes (wave-uniform base + lane * 16) and what `ds_read_b128`
// reads back conflict-free.  Weight fragments are contiguous in the packed layout; an activation
// tile is staged pixel-major in full 128-byte lines (see the kernel) with out-of-frame pixels read from a
// zero page.  Two LDS buffers: the DMA of step s+1 runs under the MFMAs of step s, one barrier per step.
// --------------------------------------------------------------------------------------------
#define WZ_LDS_TM 128
// NW = 16-channel tiles per wave: 2 -> workgroup tile 128 x 64, 4 -> 128 x 128 (twice the MFMAs per DMA byte)
#define WZ_LDS_TN(NW) (2 * (NW) * 16)
#define WZ_LDS_BUF(NW) ((4 * (NW) + 16) * 1024)   // A fragments + 16 B fragments of 1 KiB

__device__ __forceinline__ void wz_glds16(const void* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// NBUF LDS buffers = NBUF - 1 K steps of DMA in flight per workgroup.  With one step in flight the kernel is bound by
// latency, not bandwidth (bytes in flight per CU / L2 latency); the waits are explicit `s_waitcnt vmcnt(n)` on the
// wave's own DMA instructions (NW + 4 per step, always issued, completing in order) followed by a bare `s_barrier`:
// `__syncthreads()` would drain every outstanding DMA.
template <int N>
__device__ __forceinline__ void wz_wait_dma_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// SPEC: producer / consumer wavefronts.  In-kernel cycle counts (tools/conv_probe.py) showed a K step of the plain
// variant costing 2 360 cycles per wave = 1 140 to ISSUE its eight DMA instructions (each stalls ~140 cycles in the
// address path while others are in flight) + 960 for the fragment reads and 32 MFMAs + 250 waiting -- serialised,
// because a wave issues in order and there is one wave per SIMD.  With SPEC the workgroup has eight waves: 4..7 only
// issue the DMAs of step s + D, 0..3 only compute step s; the two halves meet at one barrier per step.
template <int KS, int NW, int NBUF, bool SPEC>
__global__ __launch_bounds__(SPEC ? 512 : 256) void wz_k_conv_lds(const WzConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wz_cl_smem[];
    constexpr int taps = KS * KS;
    constexpr int ABYTES = 4 * NW * 1024, BUF = WZ_LDS_BUF(NW);
    const int n_tiles = a.n_pad >> 4;   // packed 16-channel tiles; a partial last workgroup tile stages zeros beyond
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const bool producer = !SPEC || threadIdx.x >= 256, consumer = !SPEC || threadIdx.x < 256;   // wave-uniform
    const int r16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // diagnostics (a.dbg != nullptr only in engines created with WZ_MB_DEBUG=1): phase timestamps (100 MHz) of the
    // first workgroup (slots 0..7) and of the last one (8..15)
    const bool stamp = !WZ_LANE_STAMPS && a.dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1);
    unsigned long long* const dbg = a.dbg + (blockIdx.x == 0 ? 0 : 8);
    WZ_LANE_STAMP(a.dbg);
#define CL_STAMP(i) do { if (stamp) dbg[i] = wall_clock64(); } while (0)
    CL_STAMP(0);
    // XCD-aware tile order: workgroup L runs on XCD L % 8 (each XCD has its own L2).  Renumber so that
    // the workgroups of one XCD are CONSECUTIVE tiles, pixel tile fastest: the tiles that stream the same
    // weight slice (same channel tile, same K split) then share one L2 instead of pulling it eight times.
    int bx, by, bz;
    {
        const int total = a.grid_m * a.grid_n * a.splitk;
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        const int qd = total >> 3, rm = total & 7;
        const int V = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
        bx = V % a.grid_m;
        const int rest = V / a.grid_m;
        by = rest % a.grid_n;
        bz = rest / a.grid_n;
    }
    const int m_base = bx * WZ_LDS_TM;
    const int nt0 = by * (2 * NW);

    // Activations are staged in FULL 128-byte lines: one DMA instruction = 8 pixels x 64 channels (a K step), lane l
    // fetching 16-byte chunk (l & 7) of pixel (l >> 3) -- 8 cache lines per instruction instead of the 16 half lines a
    // fragment-shaped gather (16 pixels x 64 bytes) touches, which halves the address-path work per byte.  The LDS
    // image is pixel-major, 128 bytes per pixel; a B fragment (16 pixels at a 128-byte stride) would hit two bank
    // groups 8 ways, so chunk j of pixel P is stored at slot j ^ ((P >> 1) & 7): the swizzle is applied to the SOURCE
    // address (the DMA writes lane-linearly) and again when the fragment is read.
    // This wave stages pixels [wave * 32, wave * 32 + 32) of the tile, 8 per instruction.
    const int hw = a.hout * a.wout;
    int iy0[4], ix0[4], boff[4];
    bool mv[4];
    const int dma_j = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m_base + wave * 32 + i * 8 + (lane >> 3);
        mv[i] = m < a.M;
        const int mm = mv[i] ? m : 0;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / a.wout, ox = rem - oy * a.wout;
        iy0[i] = oy * a.stride - a.pad_t;
        ix0[i] = ox * a.stride - a.pad_l;
        boff[i] = b * a.hin;
    }
    // (P >> 1) & 7 of the pixel this lane fetches in instruction i: P = wave*32 + i*8 + (lane>>3)  ->  (i*4 + (lane>>4)) & 7
    const int dma_swz = lane >> 4;   // + i * 4, & 7 below

    // K steps of this split (a step = 2 consecutive 32-channel chunks of one filter tap; kc is even)
    const int nsteps = a.kchunks >> 1;
    const int per = (nsteps + a.splitk - 1) / a.splitk;
    const int s0 = bz * per, s1 = min(s0 + per, nsteps);

    auto stage = [&](int s, int buf) {
        unsigned char* base = wz_cl_smem + buf * BUF;
        // K order: channel pair outermost, filter tap innermost -- the nine taps of a channel pair read (almost) the same
        // pixel lines, 21 KiB per tile, which then stay in the 32 KiB L1 instead of being fetched from L2 nine times
        // (tap-major order puts 9 steps x every workgroup's traffic between two uses of a line)
        const int t = (KS == 1) ? 0 : ((a.order & 4) ? (s * 2) / a.kc : s % taps);
        const int c = (KS == 1) ? s * 2 : ((a.order & 4) ? s * 2 - t * a.kc : (s / taps) * 2);
        const int ky = (KS == 1) ? 0 : t / KS, kx = (KS == 1) ? 0 : t - ky * KS;
        // A: this wave stages NW/2 channel tiles x (kc = 0/1): each 2 KiB contiguous in the packed weights
#pragma unroll
        for (int i = 0; i < NW / 2; ++i) {
            const int ntl = wave * (NW / 2) + i;
            const bool have = nt0 + ntl < n_tiles;   // wave-uniform
            const half_t* wsrc = have ? a.w + ((size_t)((nt0 + ntl) * taps + t) * a.kc + c) * 512 + lane * 8
                                      : a.zeros + lane * 8;
            wz_glds16(wsrc, base + (ntl * 2 + 0) * 1024);
            wz_glds16(have ? wsrc + 512 : wsrc, base + (ntl * 2 + 1) * 1024);
        }
        // B: 8 pixels x 128 bytes per instruction
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = mv[i] && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
            const int chunk = dma_j ^ ((i * 4 + dma_swz) & 7);
            const half_t* src = ok ? a.in + ((size_t)(boff[i] + iy) * a.win + ix) * a.cin + c * 32 + chunk * 8 : a.zeros;
            wz_glds16(src, base + ABYTES + (wave * 4 + i) * 1024);
        }
    };

    float4_t acc[4][NW];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) acc[mt][nt] = (float4_t){0.f, 0.f, 0.f, 0.f};

    constexpr int D = NBUF - 1, PER = NW + 4;   // prefetch distance; DMA instructions per wave per step
    if (producer) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (s0 + d < s1) stage(s0 + d, d);
    }
    CL_STAMP(1);
    long long cyc_wait = 0, cyc_stage = 0, cyc_loop = stamp ? clock64() : 0;
    int buf = 0;
    for (int s = s0; s < s1; ++s) {
        const long long c0 = stamp ? clock64() : 0;
        // step s has landed once at most the DMAs of the steps issued after it are outstanding (in-order completion);
        // past the barrier every wave is done reading the buffer of step s - 1, which step s + D reuses
        const int later = min(D - 1, s1 - 1 - s);
        if (NBUF == 2 || later == 0)
            wz_wait_dma_then_barrier<0>();
        else if (NBUF == 3 || later == 1)
            wz_wait_dma_then_barrier<PER>();
        else
            wz_wait_dma_then_barrier<2 * PER>();
        if (s == s0) CL_STAMP(2);
        const long long c1 = stamp ? clock64() : 0;
        if (producer && s + D < s1) stage(s + D, buf == 0 ? NBUF - 1 : buf - 1);
        if (stamp) {
            const long long c2 = clock64();
            cyc_wait += c1 - c0;
            cyc_stage += c2 - c1;
        }
        const unsigned char* base = wz_cl_smem + buf * BUF;
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        if (!consumer) continue;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            half8_t fa[NW], fb[4];
#pragma unroll
            for (int nt = 0; nt < NW; ++nt)
                fa[nt] = *reinterpret_cast<const half8_t*>(base + ((wn * NW + nt) * 2 + kc) * 1024 + lane * 16);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)   // pixel P = (wm*4 + mt)*16 + r16, chunk kc*4 + g, slot swizzled by (P >> 1) & 7
                fb[mt] = *reinterpret_cast<const half8_t*>(base + ABYTES + ((wm * 4 + mt) * 16 + r16) * 128 +
                                                           (((kc * 4 + g) ^ ((r16 >> 1) & 7)) * 16));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[nt], fb[mt], acc[mt][nt], 0, 0, 0);
        }
    }

    if (!consumer) return;
    CL_STAMP(3);
    if (stamp) {
        dbg[5] = (unsigned long long)cyc_wait;
        dbg[7] = (unsigned long long)cyc_stage;
        dbg[6] = (unsigned long long)(s1 - s0) | ((unsigned long long)(clock64() - cyc_loop) << 16);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m_base + (wm * 4 + mt) * 16 + r16;
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) {
            const int n4 = (nt0 + wn * NW + nt) * 16 + g * 4;
            if (a.splitk > 1) {
                if (m < a.M && n4 < a.n_pad)
                    *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(a.out) +
                                                 ((size_t)bz * a.M + m) * a.n_pad + n4) = acc[mt][nt];
            } else {
                wz_epilogue4(a, m, n4, acc[mt][nt]);
            }
        }
    }
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg[4] = wall_clock64();
    }
#undef CL_STAMP
}

#include "k_conv_rs.h"

struct WzEpiF16 {
    static __device__ __forceinline__ void apply(const WzConvArgs& a, int m, int n4, float4_t v) { wz_epilogue4(a, m, n4, v); }
    static __device__ __forceinline__ float* partials(const WzConvArgs& a) { return reinterpret_cast<float*>(a.out); }
    static constexpr bool INLINE = true;   // supports WzConvArgs::inline_reduce
    static __device__ __forceinline__ void publish(float* p, const float4_t v) { wz_store_partial_sc1(p, v); }
    static __device__ __forceinline__ void finish(const WzConvArgs& a, int m, int n4, const float4_t v) { wz_inline_finish(a, m, n4, v); }
};

template <int KS, int NW, bool SPEC>
__global__ __launch_bounds__(SPEC ? 512 : 256, 2) void wz_k_conv_rs(const WzConvArgs a) {   // <= 256 registers: two waves per SIMD
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];
    WZ_LANE_STAMP(a.dbg);
    wz_conv_rs_body<KS, NW, SPEC, false, WzEpiF16>(a, smem, blockIdx.x);
}

// The SSD heads served by the register-staged tile kernel (the two big ones with 128 x 128 tiles, the 5x5 one with
// 128 x 64) in one launch: the workgroups of the later entries fill the CUs the first one's last round leaves idle, two
// workgroups share a CU and cover each other's stalls, and the kernel boundaries between them go away
// (measured for the two big heads: 32 us together against 23 + 26 us apart).
__global__ __launch_bounds__(256, 2) void wz_k_conv_rs_group(const WzConvGroup g) {
    WZ_LANE_STAMP(g.stamp);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];
    int e = 0;
    while (e + 1 < g.n && (int)blockIdx.x >= g.first[e + 1]) ++e;   // wave-uniform
    const int L = (int)blockIdx.x - g.first[e];
    if (g.gx[e] == 4)   // channel tiles per wave pair of this entry (wave-uniform)
        wz_conv_rs_body<3, 4, false, false, WzEpiF16>(g.a[e], smem, L);
    else
        wz_conv_rs_body<3, 2, false, false, WzEpiF16>(g.a[e], smem, L);
}

static int wz_env_int(const char* name, int dflt) {
    const char* e = wz_dev_getenv(name);
    return (e && atoi(e) >= 0 && e[0]) ? atoi(e) : dflt;
}

// the LDS-tiled kernel needs whole 64-column tiles, whole 2-chunk K steps and enough pixels to fill a tile
bool wz_conv_use_lds(const WzConvArgs& a) {
    static const int min_m = wz_env_int("WZ_LDS_MIN_M", 128);
    static const int min_k = wz_env_int("WZ_LDS_MIN_KCHUNKS", 8);
    return a.zeros && a.n_pad % 64 == 0 && a.kc % 2 == 0 && a.M >= min_m && a.kchunks >= min_k && a.cin % 32 == 0;
}

// 128 x 128 workgroup tiles where there are enough channels and pixels for them to pay
int wz_lds_nw(int M, int n_pad, int kchunks) {
    static const int force = wz_env_int("WZ_LDS_NW", 0);
    if (force == 2 || force == 4) return force;
    return (n_pad >= 256 && M >= 512 && kchunks >= 64) ? 4 : 2;   // measured: the two big heads gain, Conv_1 (K = 320) loses
}

static int wz_lds_nbuf();
static int wz_lds_spec();
int wz_choose_splitk_lds(int M, int n_pad, int kchunks) {
    static const int target = wz_env_int("WZ_LDS_WGS", 192);   // measured 64 .. 512: fewer, longer K slices win (less partial-sum traffic); 192 best
    const int tn = WZ_LDS_TN(wz_lds_nw(M, n_pad, kchunks));
    const int wgs = ((M + WZ_LDS_TM - 1) / WZ_LDS_TM) * ((n_pad + tn - 1) / tn);
    const int nsteps = kchunks / 2;
    int s = (target + wgs - 1) / wgs;
    // with more than 80 KiB of LDS per workgroup only one fits a CU: a grid beyond 256 workgroups would need a second,
    // nearly empty round
    static const int rs_mode = wz_env_int("WZ_LDS_RS", 1);
    const bool one_per_cu = rs_mode ? wz_lds_spec() != 0 : wz_lds_nbuf() * WZ_LDS_BUF(wz_lds_nw(M, n_pad, kchunks)) > 80 * 1024;
    if (one_per_cu && s * wgs > 256) s = 256 / wgs;
    if (s > nsteps / 4) s = nsteps / 4;   // >= 4 steps per split
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
}

// measured (profiles/): the 64x64 tile wins only where M is large enough to keep >= 1 wave per SIMD
// busy through a long K loop (BoxPredictor_0: M = n*361); at M = n*100 it is latency-bound and loses.
static inline bool wz_conv_big(int M, int n_pad, int kchunks) { return n_pad % 64 == 0 && n_pad >= 256 && M >= 2048 && kchunks >= 64; }

int wz_choose_splitk(int M, int n_pad, int kchunks) {
    // Long K loops on few waves are latency-bound: split K until there is about one wave per SIMD
    // (1024), keeping >= 8 chunks per split so the extra reduce launch stays amortised.
    const bool big = wz_conv_big(M, n_pad, kchunks);
    const int tm = big ? 64 : 32, tn = big ? 64 : 32;
    const long waves = (long)((M + tm - 1) / tm) * (n_pad / tn);
    static const int target = [] { const char* e = wz_dev_getenv("WZ_SPLITK_WAVES"); return (e && atoi(e) > 0) ? atoi(e) : 1024; }();
    static const int max_split = [] { const char* e = wz_dev_getenv("WZ_SPLITK_MAX"); return (e && atoi(e) > 0) ? atoi(e) : 16; }();
    if (kchunks < 32 || waves >= target) return 1;
    int s = (int)(target / (waves > 0 ? waves : 1));
    const int max_by_k = kchunks / 8;
    if (s > max_by_k) s = max_by_k;
    if (s > max_split) s = max_split;
    return s < 1 ? 1 : s;
}

template <int MT, int NT, int U>
static void wz_launch_conv_cfg(const WzConvArgs& a, hipStream_t s) {
    const int mtiles = (a.M + MT * 16 - 1) / (MT * 16);
    dim3 grid((mtiles + 3) / 4, a.n_pad / (NT * 16), a.splitk);
    if (a.ksize == 1)
        WZ_LAUNCH((wz_k_conv<1, MT, NT, U>), grid, dim3(256), 0, s, a);
    else
        WZ_LAUNCH((wz_k_conv<3, MT, NT, U>), grid, dim3(256), 0, s, a);
}

static int wz_lds_nbuf() { return 2; }

static int wz_lds_spec() {
    static const int v = wz_env_int("WZ_LDS_SPEC", 0);
    return v;
}

template <int KS, int NW, int NBUF, bool SPEC>
static void wz_conv_lds_attr() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wz_k_conv_lds<KS, NW, NBUF, SPEC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * WZ_LDS_BUF(NW));
}
template <int KS, int NW>
static void wz_conv_lds_attrs() {
    wz_conv_lds_attr<KS, NW, 2, false>();
}
void wz_conv_init() {   // kernel attributes (before any stream capture)
    wz_conv_lds_attrs<1, 2>();
    wz_conv_lds_attrs<3, 2>();
    wz_conv_lds_attrs<1, 4>();
    wz_conv_lds_attrs<3, 4>();
}

// Default: the register-staged kernel (wz_k_conv_rs).  WZ_LDS_RS=0 selects the LDS-DMA kernel (two buffers), kept for
// comparison; its deeper rings (WZ_LDS_NBUF 3/4) and its producer/consumer form are instantiated only with
// -DWZ_LDS_VARIANTS=1 -- they measured no faster.
template <int KS, int NW>
static void wz_launch_conv_lds(const WzConvArgs& a, dim3 grid, hipStream_t s) {
    static const int rs = wz_env_int("WZ_LDS_RS", 1);
    if (rs && wz_lds_spec())
        WZ_LAUNCH((wz_k_conv_rs<KS, NW, true>), grid, dim3(512), 0, s, a);
    else if (rs)
        WZ_LAUNCH((wz_k_conv_rs<KS, NW, false>), grid, dim3(256), 0, s, a);
    else
        WZ_LAUNCH((wz_k_conv_lds<KS, NW, 2, false>), grid, dim3(256), 2 * WZ_LDS_BUF(NW), s, a);
}

void wz_launch_conv(const WzConvArgs& a0, hipStream_t s) {
    if (wz_conv_use_lds(a0)) {
        WzConvArgs a = a0;
        a.grid_m = (a.M + WZ_LDS_TM - 1) / WZ_LDS_TM;
        const int nw = wz_lds_nw(a.M, a.n_pad, a.kchunks);
        a.grid_n = (a.n_pad + WZ_LDS_TN(nw) - 1) / WZ_LDS_TN(nw);
        static const int order = wz_env_int("WZ_LDS_ORDER", 0);
        a.order = order;
        dim3 grid(a.grid_m * a.grid_n * a.splitk);
        if (nw == 4) {
            if (a.ksize == 1)
                wz_launch_conv_lds<1, 4>(a, grid, s);
            else
                wz_launch_conv_lds<3, 4>(a, grid, s);
        } else {
            if (a.ksize == 1)
                wz_launch_conv_lds<1, 2>(a, grid, s);
            else
                wz_launch_conv_lds<3, 2>(a, grid, s);
        }
        return;
    }
    const WzConvArgs& a = a0;
    if (wz_conv_big(a.M, a.n_pad, a.kchunks))
        wz_launch_conv_cfg<4, 4, 2>(a, s);
    else
        wz_launch_conv_cfg<2, 2, 4>(a, s);
}

// small maps, a K loop long enough for eight slices, whole 32-channel tiles
bool wz_conv_ws_applies(const WzConvArgs& a) {
    static const int on = wz_env_int("WZ_CONV_WS", 1);
    return on && a.out_mode == WZ_OUT_ACT && a.M <= 1024 && a.kchunks >= 8 && a.n_pad % 32 == 0;
}
void wz_launch_conv_ws(const WzConvArgs& a, hipStream_t s) {
    // Many pixels x many channels (Conv_1: 800 x 1 280 at batch 8): 1 000 tiles of 32 x 32 at 173 registers are one workgroup per CU and FOUR
    // rounds of workgroups that each wait out their operands' latency (13.3 us with split weights, K = 640).  32 x 64 tiles, K in four slices,
    // two chunks per load group (109 registers: two workgroups per CU) are 500 workgroups in ONE round: 6.9 us.  Measured beside it
    // (profiles/r04_conv1_tiles.txt): 32 x 32 at 109 registers 8.1 us, 64 x 64 (K in halves) 11.9 us, 64 x 64 at 173 registers 12.8 us (260
    // workgroups on 256 CUs: a second round for four of them).  WZ_CONV_WS64=0: the 32 x 32 tiles everywhere.
    static const int wide_tiles = wz_env_int("WZ_CONV_WS64", 1);
    if (wide_tiles && a.ksize == 1 && a.M >= 512 && a.n_pad >= 512 && a.n_pad % 64 == 0 && a.kchunks >= 8) {
        WZ_LAUNCH((wz_k_conv_ws<1, 1, 2>), dim3((a.M + 31) / 32, a.n_pad / 64), dim3(512), 0, s, a);
        return;
    }
    const dim3 grid((a.M + 31) / 32, a.n_pad / 32);
    // (two K chunks per load group, 109 registers, two workgroups per CU for every launch of this kernel since late round 4: with U = 4 and 173
    //  registers a workgroup owned its CU -- the extras chain's launches 4.5 / 6.0 -> 4.0 / 5.1 us and, with four lanes in flight, 46.9 -> 47.4 k
    //  frames/s: what a launch keeps other lanes' workgroups from using counts, profiles/r04_conv1_tiles.txt)
    if (a.ksize == 1)
        WZ_LAUNCH(wz_k_conv_ws<1>, grid, dim3(512), 0, s, a);
    else
        WZ_LAUNCH(wz_k_conv_ws<3>, grid, dim3(512), 0, s, a);
}

// the 3x3 convolutions wz_launch_conv would give to wz_k_conv<3, 2, 2, 4>
bool wz_conv_groupable(const WzConvArgs& a) {
    return a.ksize == 3 && !wz_conv_use_lds(a) && !wz_conv_big(a.M, a.n_pad, a.kchunks);
}
void wz_conv_group_add(WzConvGroup& g, const WzConvArgs& a) {
    const int i = g.n++;
    const int mtiles = (a.M + 31) / 32;
    g.a[i] = a;
    g.gx[i] = (mtiles + 3) / 4;
    g.gy[i] = a.n_pad / 32;
    g.first[i + 1] = g.first[i] + g.gx[i] * g.gy[i] * a.splitk;
    g.a[i].grid_m = g.gx[i];
    g.a[i].grid_n = g.gy[i];
    g.a[i].tickets = g.tickets ? g.tickets + g.ticket_off : nullptr;   // one counter per wave tile
    g.ticket_off += g.gx[i] * g.gy[i] * 4;
}
void wz_launch_conv_group(const WzConvGroup& g, hipStream_t s) {
    WZ_LAUNCH(wz_k_conv_group, dim3(g.first[g.n]), dim3(256), 0, s, g);
}

// the convolutions wz_launch_conv would give to wz_k_conv_rs<3, 4>
bool wz_conv_rs_groupable(const WzConvArgs& a) {
    static const int rs = wz_env_int("WZ_LDS_RS", 1);
    return rs && !wz_lds_spec() && a.ksize == 3 && wz_conv_use_lds(a);
}
// Returns the number of entries added (0: the group is full).  A head whose packed columns are an odd number of 64-column
// tiles (BoxPredictor_0: 320 = 2 x 128 + 64, BoxPredictor_1: 576 = 4 x 128 + 64) goes in as TWO entries -- the 128-column
// tiles and one 64-column tile behind them -- instead of rounding up to 128-column tiles (384 / 640 columns of MFMA work: 20 % /
// 11 % of it on padding).  Both write the same partial-sum slab ([z][M][n_pad], absolute columns) with the same split count.
int wz_conv_rs_group_add(WzConvGroup& g, const WzConvArgs& a0) {
    static const int split_n = wz_env_int("WZ_HEAD_SPLIT_N", 1);
    const int nw = wz_lds_nw(a0.M, a0.n_pad, a0.kchunks);
    const bool two = split_n && nw == 4 && (a0.n_pad % 128) == 64 && a0.n_pad > 128;
    if (g.n + (two ? 2 : 1) > WZ_CONV_GROUP_MAX) return 0;
    for (int part = 0; part < (two ? 2 : 1); ++part) {
        const int i = g.n++;
        WzConvArgs& a = g.a[i];
        a = a0;
        const int pnw = (two && part == 1) ? 2 : nw;
        a.grid_m = (a.M + WZ_LDS_TM - 1) / WZ_LDS_TM;
        a.nt_base = (two && part == 1) ? (a.n_pad / 128) * 8 : 0;
        a.grid_n = two ? (part == 0 ? a.n_pad / 128 : 1) : (a.n_pad + WZ_LDS_TN(nw) - 1) / WZ_LDS_TN(nw);
        a.order = 0;
        g.gx[i] = pnw;
        g.gy[i] = 0;
        g.first[i + 1] = g.first[i] + ((a.grid_m * a.grid_n * a.splitk + 7) & ~7);   // entries start on an XCD boundary
        a.tickets = g.tickets ? g.tickets + g.ticket_off : nullptr;                  // one counter per workgroup tile
        g.ticket_off += a.grid_m * a.grid_n;
    }
    return two ? 2 : 1;
}
void wz_launch_conv_rs_group(const WzConvGroup& g, hipStream_t s) {
    WZ_LAUNCH(wz_k_conv_rs_group, dim3(g.first[g.n]), dim3(256), 0, s, g);
}

void wz_reduce_group_add(WzReduceGroup& g, const WzConvArgs& a, const float* ws) {
    const int i = g.n++;
    g.a[i] = a;
    g.ws[i] = ws;
    const int threads = a.frag_ws ? ((a.M + 15) >> 4) * (a.n_pad >> 4) * 64 : a.M * (a.n_pad >> 2);
    g.first[i + 1] = g.first[i] + (threads + 255) / 256;
}
void wz_launch_splitk_reduce_group(const WzReduceGroup& g, hipStream_t s) {
    WZ_LAUNCH(wz_k_splitk_reduce_group, dim3(g.first[g.n]), dim3(256), 0, s, g);
}

void wz_launch_splitk_reduce(const WzConvArgs& a, const float* ws, hipStream_t s) {
    const int total = a.M * (a.n_pad >> 2);
    WZ_LAUNCH(wz_k_splitk_reduce, dim3((total + 255) / 256), dim3(256), 0, s, a, ws);
}
