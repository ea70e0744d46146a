// Two convolutions of the SSD extras chain in ONE launch: the 1x1 bottleneck (layer_19_1_Conv2d_k_1x1_*) and the 3x3 stride-2 convolution behind it
// (layer_19_2_Conv2d_k_3x3_s2_*) on the 5x5 / 3x3 / 2x2 maps -- layers inside `sess.run` of watsor/detection/tensorflow_cpu.py:113-115.
//
// Why.  These six launches are 0.1 % of the network's FLOPs and a tenth of a batch's chain: 1.1 - 2.2 us of kernel each plus a 1.5 - 2 us boundary alone,
// 3 - 7 us each with four lanes in flight (profiles/r06zz_lane_overlap_robust.txt) -- leaving the whole extras chain out was worth 11 % of the throughput in
// round 3's experiment (profiles/r03_lane_time_skip_experiments.txt).  One launch for all six with a workgroup per frame lost twice (k_tail.hip, rounds 2 and 5:
// a frame's workgroup streams 1.5 MB of weights through one CU).  A PAIR keeps the parallelism: workgroup = (frame, two 16-channel tiles of the 3x3's output);
//   stage 1  the 1x1 on ALL of the frame's pixels (<= 25: two pixel tiles), one 16-channel tile of its output per wave, straight from L2 into registers,
//            bias + relu6 + ONE fp16 rounding -- the rounding the unfused program applies when it stores the tensor -- into LDS (<= 7 KB); the workgroup of
//            tile 0 also stores it to its tensor in HBM (nothing reads it there but the tests and a buffer-keeping engine);
//   stage 2  the 3x3 stride-2 from LDS: its K (9 taps x cmid / 32 chunks) dealt over the 8 waves, partial tiles summed through LDS in wave order (fixed:
//            bit-identical run to run), bias + relu6, fp16 NHWC store (the SSD feature map the heads and the next pair read).
// The 1x1 is recomputed by each of the frame's 4 - 8 workgroups (256 MFMAs: nothing) and its weights (32 - 131 KB) are streamed by each: 0.1 - 0.2 MB per
// workgroup, 32 - 64 workgroups per launch at batch 8.
#include "wz_common.h"

struct WzPairArgs {
    WzConvArgs a;       // the 1x1 (in, w, bias, out = its tensor or nullptr, hin/win, cin, cout = cmid, kc, act)
    WzConvArgs b;       // the 3x3 stride 2 (w, bias, out, hout/wout, cout, n_pad, kc = cmid / 32, pad_t, pad_l, stride, act)
    int32_t n_frames, ntb;   // frames, 16-channel tiles of the 3x3's output per workgroup
};

#define PAIR_NW 8
#define PAIR_MID_PAD 8     // halves of padding per pixel row of the intermediate tensor in LDS

// KCA: K chunks of the 1x1 at most (registers: KCA weight fragments per wave); UPW: (tap, chunk) units of the 3x3 per wave at most.
// Everything a wave needs from memory is requested in its first instructions -- its share of the frame's pixels (fetched once per workgroup, shared
// through LDS), ALL of its 1x1 weight fragments and ALL of its 3x3 weight fragments -- so the launch pays one memory latency, not one per K step
// (a first version that loaded four chunks at a time and the 3x3's fragments unit by unit took 8.8 us for the 512-channel pair where the two
// launches it replaces take 1.7 + 2.4: profiles/r06_extras_pair_fusion.txt).
template <int NTB, int KCA, int UPW>
__global__ __launch_bounds__(PAIR_NW * 64, 1) void wz_k_extras_pair(const WzPairArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pair_smem[];
    WZ_LANE_STAMP(A.a.dbg);
    const WzConvArgs& a = A.a;
    const WzConvArgs& b = A.b;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r16 = lane & 15, g = lane >> 4;
    const int tiles_b = (b.n_pad >> 4) / NTB;
    const int f = (int)blockIdx.x / tiles_b, tb = (int)blockIdx.x - f * tiles_b;
    const int P = a.hin * a.win;                      // pixels of the frame on the 1x1's map (<= 32)
    const int cmid = a.cout, mrow = cmid + PAIR_MID_PAD;
    const int kca = a.kc;
    unsigned char* const xl = pair_smem;                                                        // [2 * kca] fragments of 1 KiB: the frame's pixels
    half_t* const mid = reinterpret_cast<half_t*>(pair_smem + (size_t)2 * KCA * 1024);           // [32][cmid + pad]
    float* const red = reinterpret_cast<float*>(pair_smem + (size_t)2 * KCA * 1024 + (size_t)32 * mrow * 2);   // [PAIR_NW][NTB][64][4]
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- every load of the wave, up front
    constexpr int XPW = (2 * KCA + PAIR_NW - 1) / PAIR_NW;     // pixel fragments (pixel tile mt, chunk c) -> index mt * kca + c, dealt over the waves
    half8_t xpart[XPW];
    const half_t* const x0 = a.in + (size_t)f * P * a.cin;
#pragma unroll
    for (int k = 0; k < XPW; ++k) {
        const int fi = wave + k * PAIR_NW;
        xpart[k] = zero8;
        if (fi < 2 * kca) {
            const int mt = fi / kca, c = fi - mt * kca;
            const int p = mt * 16 + r16;
            if (p < P && c * 32 + g * 8 < a.cin) xpart[k] = *reinterpret_cast<const half8_t*>(x0 + (size_t)p * a.cin + c * 32 + g * 8);
        }
    }
    const int nta = cmid >> 4;
    half8_t wa[KCA];
#pragma unroll
    for (int c = 0; c < KCA; ++c)
        wa[c] = (wave < nta && c < kca) ? *reinterpret_cast<const half8_t*>(a.w + (((size_t)wave * kca + c) * 64 + lane) * 8) : zero8;
    const int units = 9 * b.kc;
    half8_t wb[UPW][NTB];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int u = wave + i * PAIR_NW;
        const int uc = u < units ? u : 0;
        const int tap = uc / b.kc, c = uc - tap * b.kc;
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
            wb[i][nt] = u < units ? *reinterpret_cast<const half8_t*>(b.w + (((size_t)((tb * NTB + nt) * 9 + tap) * b.kc + c) * 64 + lane) * 8) : zero8;
    }
#pragma unroll
    for (int k = 0; k < XPW; ++k) {
        const int fi = wave + k * PAIR_NW;
        if (fi < 2 * kca) *reinterpret_cast<half8_t*>(xl + (size_t)fi * 1024 + lane * 16) = xpart[k];
    }
    __syncthreads();

    // ---- stage 1: mid[p][n] = act(sum_k Wa[n][k] X[p][k] + ba[n]), wave w owns 16-channel tile w (cmid <= 128: at most 8 tiles)
    if (wave < nta) {
        float4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < KCA; ++c) {
            if (c >= kca) continue;
            const half8_t xa = *reinterpret_cast<const half8_t*>(xl + (size_t)c * 1024 + lane * 16);
            const half8_t xb = *reinterpret_cast<const half8_t*>(xl + (size_t)(kca + c) * 1024 + lane * 16);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c], xa, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c], xb, acc[1], 0, 0, 0);
        }
        const int n4 = wave * 16 + g * 4;
        const float4_t bv = *reinterpret_cast<const float4_t*>(a.bias + n4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int p = mt * 16 + r16;
            half4_t o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[mt][r] + bv[r];
                if (a.act == WZ_ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
                o[r] = (half_t)v;
            }
            *reinterpret_cast<half4_t*>(mid + (size_t)p * mrow + n4) = o;     // (rows >= P hold act(bias): never read)
            if (tb == 0 && a.out && p < P && n4 < a.cout)
                *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(a.out) + ((size_t)f * P + p) * a.cout + n4) = o;
        }
    }
    __syncthreads();

    // ---- stage 2: out[q][n] = act(sum_{tap, k} Wb[n][tap][k] mid[pixel(q, tap)][k] + bb[n]); K units (tap, chunk) dealt over the waves
    const int Q = b.hout * b.wout;                    // output pixels of the frame (<= 16)
    const int q = r16 < Q ? r16 : 0;
    const int oy = q / b.wout, ox = q - oy * b.wout;
    float4_t acc2[NTB];
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) acc2[nt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int u = wave + i * PAIR_NW;
        if (u >= units) continue;
        const int tap = u / b.kc, c = u - tap * b.kc;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = oy * b.stride - b.pad_t + ky, ix = ox * b.stride - b.pad_l + kx;
        const bool in = r16 < Q && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win && c * 32 + g * 8 < cmid;
        const half8_t x = in ? *reinterpret_cast<const half8_t*>(mid + (size_t)(iy * a.win + ix) * mrow + c * 32 + g * 8) : zero8;
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[i][nt], x, acc2[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt)
        *reinterpret_cast<float4_t*>(red + ((size_t)(wave * NTB + nt) * 64 + lane) * 4) = acc2[nt];
    __syncthreads();
    if (wave >= NTB) return;
    float4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < PAIR_NW; ++w) {
        const float4_t pz = *reinterpret_cast<const float4_t*>(red + ((size_t)(w * NTB + wave) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += pz[r];
    }
    const int n4 = (tb * NTB + wave) * 16 + g * 4;
    if (r16 >= Q || n4 >= b.cout) return;
    const float4_t bv = *reinterpret_cast<const float4_t*>(b.bias + n4);
    half4_t o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = v[r] + bv[r];
        if (b.act == WZ_ACT_RELU6) t = fminf(fmaxf(t, 0.0f), 6.0f);
        o[r] = (half_t)t;
    }
    *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(b.out) + ((size_t)f * Q + r16) * b.cout + n4) = o;
}

// a: a 1x1 stride-1 convolution with an activation tensor as output; b: the 3x3 stride-2 convolution that reads it (and nothing else does)
bool wz_extras_pair_applies(const WzConvArgs& a, const WzConvArgs& b) {
    const char* const e = wz_dev_getenv("WZ_EXTRAS_PAIR");      // development library only: 0 = the two launches
    if (e && e[0] && atoi(e) == 0) return false;
    if (a.ksize != 1 || a.stride != 1 || a.out_mode != WZ_OUT_ACT || a.res || a.hin != a.hout || a.win != a.wout) return false;
    if (b.ksize != 3 || b.stride != 2 || b.out_mode != WZ_OUT_ACT || b.res || b.hin != a.hout || b.win != a.wout || b.cin != a.cout) return false;
    if (a.hin * a.win > 32 || b.hout * b.wout > 16) return false;
    if ((a.cout & 31) || a.cout > 128 || a.n_pad != a.cout || (a.cin & 31) || a.kc * 32 != a.cin || a.kc > 16 || b.kc * 32 != b.cin || 9 * b.kc > 5 * PAIR_NW) return false;
    if ((b.n_pad & 31) || b.cout != b.n_pad) return false;
    return true;
}

// enqueues ONE launch for both convolutions; n = frames
void wz_launch_extras_pair(const WzConvArgs& a, const WzConvArgs& b, int n, hipStream_t s) {
    WzPairArgs A;
    A.a = a;
    A.b = b;
    A.n_frames = n;
    A.ntb = 2;
    const int tiles_b = (b.n_pad >> 4) / 2;
    const int upw = (9 * b.kc + PAIR_NW - 1) / PAIR_NW;
    if (a.kc <= 8 && upw <= 3) {
        const size_t lds = (size_t)2 * 8 * 1024 + (size_t)32 * (a.cout + PAIR_MID_PAD) * 2 + (size_t)PAIR_NW * 2 * 1024;
        WZ_LAUNCH((wz_k_extras_pair<2, 8, 3>), dim3(n * tiles_b), dim3(PAIR_NW * 64), lds, s, A);
    } else if (a.kc <= 8) {
        const size_t lds = (size_t)2 * 8 * 1024 + (size_t)32 * (a.cout + PAIR_MID_PAD) * 2 + (size_t)PAIR_NW * 2 * 1024;
        WZ_LAUNCH((wz_k_extras_pair<2, 8, 5>), dim3(n * tiles_b), dim3(PAIR_NW * 64), lds, s, A);
    } else {
        const size_t lds = (size_t)2 * 16 * 1024 + (size_t)32 * (a.cout + PAIR_MID_PAD) * 2 + (size_t)PAIR_NW * 2 * 1024;
        WZ_LAUNCH((wz_k_extras_pair<2, 16, 5>), dim3(n * tiles_b), dim3(PAIR_NW * 64), lds, s, A);
    }
}
