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
#include <string.h>

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

// The two horizontal taps of one source row.  RGB24: pixels x_lo and x_hi = x_lo (+1) are 6 contiguous bytes, so ONE 12-byte load
// from the 4-byte boundary below them replaces six byte loads (the kernel was bound by the number of vector-memory instructions:
// twelve byte loads per output pixel); the last few bytes of a frame, where those 12 bytes could reach past it, are read byte-wise.
__device__ __forceinline__ void wz_fetch_row_pair(const WzFrameDesc& f, int x_lo, int x_hi, int y, float (&a)[3], float (&b)[3]) {
    if (f.fmt == WZ_FMT_RGB24) {
        const size_t off = ((size_t)y * f.w + x_lo) * 3;
        const size_t base = off & ~(size_t)3;
        if (base + 12 <= (size_t)f.w * f.h * 3 && ((uintptr_t)f.rgb & 3) == 0) {
            typedef __attribute__((ext_vector_type(3))) unsigned int u3;   // (natural alignment 16; the address is only 4-byte aligned:
            u3 v;                                                          //  memcpy states that, and still compiles to one 12-byte load)
            __builtin_memcpy(&v, __builtin_assume_aligned(f.rgb + base, 4), 12);
            const unsigned sh = (unsigned)(off - base) * 8;                       // 0, 8, 16 or 24
            const unsigned long long lo64 = ((unsigned long long)v[1] << 32) | v[0];
            const unsigned long long hi64 = ((unsigned long long)v[2] << 32) | v[1];
            const unsigned long long w0 = lo64 >> sh;                             // bytes 0 .. 7 - sh/8 of the window
            const unsigned w1 = (unsigned)(hi64 >> sh);                           // bytes 4 .. 7 of the window (sh < 32)
            a[0] = (float)(w0 & 0xff);
            a[1] = (float)((w0 >> 8) & 0xff);
            a[2] = (float)((w0 >> 16) & 0xff);
            if (x_hi != x_lo) {
                b[0] = (float)((w0 >> 24) & 0xff);
                b[1] = (float)(w1 & 0xff);
                b[2] = (float)((w1 >> 8) & 0xff);
            } else {
                b[0] = a[0]; b[1] = a[1]; b[2] = a[2];
            }
            return;
        }
    }
    wz_fetch_rgb(f, x_lo, y, a);
    wz_fetch_rgb(f, x_hi, y, b);
}

// HP: the network input is stored as a hi + lo pair of halves per value (hi = RN16(v), lo = RN16(v - hi), both
// roundings and the subtraction exact-or-once-rounded fp32 operations): the first blocks of the `-p 16` program take
// both (k_mbconv_hp.hip), which removes the 2^-11 input rounding from the error budget of the scores.
// `keep`: where to leave a copy of the frame's descriptor for the kernels behind (wz_k_nms reads sizes and camera ids from it),
// or nullptr.  With `frames` pointing into page-locked HOST memory this replaces the descriptor copy in front of every batch
// (a graph node of its own, ~4 us of a batch's latency) by one 32-byte read over PCIe per workgroup, in flight together.
// `half_pixel`: the coordinate rule of graphs exported with ResizeBilinear(half_pixel_centers=True) -- src = (dst + 0.5) * scale - 0.5,
// three roundings (`HalfPixelScaler`) -- instead of the legacy src = dst * scale (WzBlobHeader::resize_mode; oracle/preprocess.py).
// `frames` == nullptr: the descriptors are the kernel's own ARGUMENTS (`pack`, up to WZ_DESC_PACK frames: scalar loads from the
// kernarg segment) -- neither a copy node in front of the batch nor a read over PCIe at the head of every workgroup; the engine
// rewrites them in the captured graph's node before every replay (wz_engine.hip: run_batch).
template <bool HP>
__global__ __launch_bounds__(256) void wz_k_preprocess(const WzFrameDesc* __restrict__ frames, const WzDescPack pack, int size,
                                                       half_t* __restrict__ out, WzFrameDesc* __restrict__ keep, int half_pixel) {
    const WzFrameDesc f = frames ? frames[blockIdx.y] : pack.d[blockIdx.y];
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
    wz_fetch_row_pair(f, x_lo, x_hi, y_lo, tap[0], tap[1]);
    wz_fetch_row_pair(f, x_lo, x_hi, y_hi, tap[2], tap[3]);
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
                          bool half_pixel, const WzFrameDesc* by_value) {
    dim3 grid((size * size + 255) / 256, n);
    WzDescPack pack;
    memset(&pack, 0, sizeof(pack));
    if (by_value && n <= WZ_DESC_PACK) {
        memcpy(pack.d, by_value, sizeof(WzFrameDesc) * n);
        d_frames = nullptr;
    }
    if (hp)
        WZ_LAUNCH(wz_k_preprocess<true>, grid, dim3(256), 0, s, d_frames, pack, size, out, keep, half_pixel ? 1 : 0);
    else
        WZ_LAUNCH(wz_k_preprocess<false>, grid, dim3(256), 0, s, d_frames, pack, size, out, keep, half_pixel ? 1 : 0);
}
const void* wz_preprocess_func(bool hp) {
    return hp ? reinterpret_cast<const void*>(wz_k_preprocess<true>) : reinterpret_cast<const void*>(wz_k_preprocess<false>);
}
