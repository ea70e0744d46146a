// Resize + normalise: a frame of any resolution (packed RGB24, or NV12 / I420) -> 300x300 fp16 network input.
//
// Stands in for the first nodes of the TF graph the reference's CPU plugin runs on the full
// resolution frame (`watsor/detection/tensorflow_cpu.py:113-115`):
// ToFloat -> ResizeBilinear(align_corners=False, legacy: src = dst * (in/out), lo = floor,
// hi = min(ceil, in-1), lerp in fp32) -> (2/255)*x - 1.  The TRT plugin does the same two steps on
// the host (`tensorrt_gpu.py:67,179-180`).  Every fp32 operation is rounded once, in TF's order,
// so the result equals the oracle (oracle/preprocess.py) bit for bit before the final fp16 rounding.
//
// Roofline: HBM-bound streaming kernel.  Algorithmic bytes per frame = W*H*3 read + size*size*4*2
// written (the 4th channel is zero padding so the stem conv reads one aligned 8-byte pixel).
#pragma clang fp contract(off)
#include "wz_common.h"

// Pixel formats (WzFrameDesc::fmt; SURVEY 8f-3, the decoder side: `watsor/stream/ffmpeg.py:78-88` reads rawvideo frames
// of whatever `-pix_fmt` the decoder was told to write).  RGB24 is what the reference's schema asks for
// (`watsor/config/schema.py:161`); NV12 / I420 are what a video decoder produces natively and half the bytes per
// frame on PCIe.  For those the RGB value of a source pixel is worked out here, where ffmpeg's swscale would have
// done it on the host: 8-bit BT.601 limited range, the chroma sample of the pixel's 2x2 block (no chroma
// interpolation), the usual 8.8 fixed-point form
//     C = Y - 16, D = U - 128, E = V - 128
//     R = clip8((298 C + 409 E + 128) >> 8),  G = clip8((298 C - 100 D - 208 E + 128) >> 8),  B = clip8((298 C + 516 D + 128) >> 8)
// -- integer arithmetic, bit-exact against oracle/yuv.py; the bytes equal a host-side conversion by that formula, they
// are NOT pinned against a particular swscale code path (its C and SIMD converters differ from each other by one LSB).
__device__ __forceinline__ void wz_fetch_rgb(const WzFrameDesc& f, int x, int y, float (&c)[3]) {
    if (f.fmt == WZ_FMT_RGB24) {
        const uint8_t* p = f.rgb + ((size_t)y * f.w + x) * 3;
        c[0] = (float)p[0];
        c[1] = (float)p[1];
        c[2] = (float)p[2];
        return;
    }
    const int Y = f.rgb[(size_t)y * f.w + x];
    const size_t cw = (size_t)(f.w >> 1);
    const uint8_t* chroma = f.rgb + (size_t)f.w * f.h;
    int U, V;
    if (f.fmt == WZ_FMT_NV12) {
        const uint8_t* uv = chroma + ((size_t)(y >> 1) * cw + (x >> 1)) * 2;
        U = uv[0];
        V = uv[1];
    } else {   // I420: a U plane, then a V plane
        const uint8_t* up = chroma + (size_t)(y >> 1) * cw + (x >> 1);
        U = up[0];
        V = up[cw * (size_t)(f.h >> 1)];
    }
    const int C = Y - 16, D = U - 128, E = V - 128;
    c[0] = (float)min(max((298 * C + 409 * E + 128) >> 8, 0), 255);
    c[1] = (float)min(max((298 * C - 100 * D - 208 * E + 128) >> 8, 0), 255);
    c[2] = (float)min(max((298 * C + 516 * D + 128) >> 8, 0), 255);
}

// HP: the network input is stored as a hi + lo pair of halves per value (hi = RN16(v), lo = RN16(v - hi), both
// roundings and the subtraction exact-or-once-rounded fp32 operations): the first blocks of the `-p 16` program take
// both (k_mbconv_hp.hip), which removes the 2^-11 input rounding from the error budget of the scores.
// `keep`: where to leave a copy of the frame's descriptor for the kernels behind (wz_k_nms reads sizes and camera ids from it),
// or nullptr.  With `frames` pointing into page-locked HOST memory this replaces the descriptor copy in front of every batch
// (a graph node of its own, ~4 us of a batch's latency) by one 32-byte read over PCIe per workgroup, in flight together.
// `half_pixel`: the coordinate rule of graphs exported with ResizeBilinear(half_pixel_centers=True) -- src = (dst + 0.5) * scale - 0.5,
// three roundings (`HalfPixelScaler`) -- instead of the legacy src = dst * scale (WzBlobHeader::resize_mode; oracle/preprocess.py).
template <bool HP>
__global__ __launch_bounds__(256) void wz_k_preprocess(const WzFrameDesc* __restrict__ frames, int size,
                                                       half_t* __restrict__ out, WzFrameDesc* __restrict__ keep, int half_pixel) {
    const WzFrameDesc f = frames[blockIdx.y];
    if (keep && blockIdx.x == 0 && threadIdx.x == 0) keep[blockIdx.y] = f;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= size * size) return;
    const int oy = pix / size, ox = pix - oy * size;

    const float in_y = half_pixel ? ((float)oy + 0.5f) * f.scale_y - 0.5f : (float)oy * f.scale_y;
    const float fl_y = floorf(in_y);
    const int y_lo = max((int)fl_y, 0);
    const int y_hi = min((int)ceilf(in_y), f.h - 1);
    const float ly = in_y - fl_y;
    const float in_x = half_pixel ? ((float)ox + 0.5f) * f.scale_x - 0.5f : (float)ox * f.scale_x;
    const float fl_x = floorf(in_x);
    const int x_lo = max((int)fl_x, 0);
    const int x_hi = min((int)ceilf(in_x), f.w - 1);
    const float lx = in_x - fl_x;

    float tap[4][3];   // top-left, top-right, bottom-left, bottom-right
    wz_fetch_rgb(f, x_lo, y_lo, tap[0]);
    wz_fetch_rgb(f, x_hi, y_lo, tap[1]);
    wz_fetch_rgb(f, x_lo, y_hi, tap[2]);
    wz_fetch_rgb(f, x_hi, y_hi, tap[3]);
    half_t v[4], vl[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float tl = tap[0][c], tr = tap[1][c];
        const float bl = tap[2][c], br = tap[3][c];
        const float top = tl + (tr - tl) * lx;
        const float bot = bl + (br - bl) * lx;
        const float px = top + (bot - top) * ly;
        const float nv = (2.0f / 255.0f) * px - 1.0f;
        v[c] = (half_t)nv;   // round-to-nearest-even
        vl[c] = (half_t)(nv - (float)v[c]);
    }
    v[3] = vl[3] = (half_t)0.0f;
    if constexpr (HP) {
        const half8_t o = {v[0], v[1], v[2], v[3], vl[0], vl[1], vl[2], vl[3]};
        *reinterpret_cast<half8_t*>(out + ((size_t)blockIdx.y * size * size + pix) * 8) = o;
    } else {
        const half4_t o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<half4_t*>(out + ((size_t)blockIdx.y * size * size + pix) * 4) = o;
    }
}

void wz_launch_preprocess(const WzFrameDesc* d_frames, int n, int size, half_t* out, hipStream_t s, bool hp, WzFrameDesc* keep,
                          bool half_pixel) {
    dim3 grid((size * size + 255) / 256, n);
    if (hp)
        WZ_LAUNCH(wz_k_preprocess<true>, grid, dim3(256), 0, s, d_frames, size, out, keep, half_pixel ? 1 : 0);
    else
        WZ_LAUNCH(wz_k_preprocess<false>, grid, dim3(256), 0, s, d_frames, size, out, keep, half_pixel ? 1 : 0);
}
