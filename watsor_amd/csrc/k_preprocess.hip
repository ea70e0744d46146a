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
                                                       half_t* __restrict__ out, WzFrameDesc* __restrict__ keep, int flags) {
    WZ_LANE_STAMP(keep ? reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(keep) - WZ_STAMP_PRE_BYTES) : nullptr);
    const int half_pixel = flags & 1;   // (bits 8 ..: the row-staged kernel's LDS budget, unused here; one argument list for both kernels)
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Row-staged form (round 4): one workgroup per OUTPUT ROW of one frame (or per few of them: see the kernel).  The source rows they tap (for NV12 / I420
// also their chroma rows) are fetched as whole, contiguous byte runs with 16-byte loads -- every 64-byte line of a row requested
// once, by consecutive lanes -- into LDS, and the 300 output pixels are computed from there with exactly the arithmetic above
// (same operations in the same order: bit-identical output, tests/test_gpu_parity.py).
//
// Why: the per-pixel form above asks for 12 bytes per lane at a stride of `scale_x * 3` bytes and for every source row once per
// output row that taps it -- fine out of HBM behind L2, wasteful when the frame lies in page-locked HOST memory and every request
// is a PCIe read: with this form the kernel can read a camera's frame where the decoder wrote it (`watsor/stream/share.py:35-41`,
// registered by wz_host_register; device-mapped address in WzFrameDesc::rgb) and no staging copy is made at all.  TF-legacy bilinear
// at a down-scale >= 2 never touches most rows: 1920x1080 -> 300x300 taps 600 of the 1080 rows (3.46 of 6.22 MB per RGB24 frame),
// NV12 600 luma + <= 600 chroma rows of 1620.  (SURVEY 8(d) "Host/PCIe side bound"; VERDICT r3 next #3.)
// LDS: [luma / RGB span | chroma span(s)], each run starting at the 16-byte boundary below its first byte.

// the three aligned dwords around byte offset `off` of an LDS run -> the 6 bytes of two adjacent RGB pixels (or of one, twice)
__device__ __forceinline__ void wz_lds_row_pair(const uint8_t* __restrict__ run, int off, bool two, float (&a)[3], float (&b)[3]) {
    const int base = off & ~3;
    const uint32_t* __restrict__ w = reinterpret_cast<const uint32_t*>(run + base);
    const uint32_t v0 = w[0], v1 = w[1], v2 = w[2];
    const unsigned sh = (unsigned)(off - base) * 8;
    const unsigned long long lo64 = ((unsigned long long)v1 << 32) | v0;
    const unsigned long long hi64 = ((unsigned long long)v2 << 32) | v1;
    const unsigned long long w0 = lo64 >> sh;
    const unsigned w1 = (unsigned)(hi64 >> sh);
    a[0] = (float)(w0 & 0xff);
    a[1] = (float)((w0 >> 8) & 0xff);
    a[2] = (float)((w0 >> 16) & 0xff);
    if (two) {
        b[0] = (float)((w0 >> 24) & 0xff);
        b[1] = (float)(w1 & 0xff);
        b[2] = (float)((w1 >> 8) & 0xff);
    } else {
        b[0] = a[0]; b[1] = a[1]; b[2] = a[2];
    }
}

__device__ __forceinline__ void wz_yuv_to_rgb(int Y, int U, int V, float (&c)[3]) {
    const int C = Y - 16, D = U - 128, E = V - 128;
    c[0] = (float)min(max((298 * C + 409 * E + 128) >> 8, 0), 255);
    c[1] = (float)min(max((298 * C - 100 * D - 208 * E + 128) >> 8, 0), 255);
    c[2] = (float)min(max((298 * C + 516 * D + 128) >> 8, 0), 255);
}

#define WZ_PRE_ROWS_THREADS 320   // 5 waves: 300 output pixels of a row, one per thread
#define WZ_PRE_ROWS_MAX 8         // output rows a workgroup takes at most
// A workgroup takes R consecutive output rows and stages the SPAN of source rows they tap -- one contiguous byte run (rows are packed),
// plus the span of chroma rows under it (NV12: one run, I420: a U and a V run).  R = 1 where the resize skips rows (vertical scale >= 2: the
// span is the two adjacent rows an output row taps, nothing is fetched that is not used); below that every source row is tapped, by one
// or two output rows, and R is as many rows as the LDS budget holds (up to 8): every source byte crosses PCIe once, in runs of 15 - 40 KiB
// (640x480: 5 output rows = 8 - 9 source rows; one row at a time read 1.25 x the frame in 1 920-byte runs).  The grid is one workgroup per
// output row for every frame of the batch; a frame's surplus workgroups leave at once.
template <bool HP>
__global__ __launch_bounds__(WZ_PRE_ROWS_THREADS) void wz_k_preprocess_rows(const WzFrameDesc* __restrict__ frames, const WzDescPack pack,
                                                                            int size, half_t* __restrict__ out,
                                                                            WzFrameDesc* __restrict__ keep, int flags) {
    WZ_LANE_STAMP(keep ? reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(keep) - WZ_STAMP_PRE_BYTES) : nullptr);
    extern __shared__ __attribute__((aligned(16))) uint8_t wz_pre_lds[];
    typedef unsigned int wz_u32x4 __attribute__((ext_vector_type(4)));   // (an array of HIP's uint4 structs is kept in scratch memory)
    const int half_pixel = flags & 1;
    const int budget = (flags >> 8) << 8;                            // LDS bytes of this launch
    const WzFrameDesc f = frames ? frames[blockIdx.y] : pack.d[blockIdx.y];
    if (keep && blockIdx.x == 0 && threadIdx.x == 0) keep[blockIdx.y] = f;
    const int tid = threadIdx.x;
    const int rowb = f.fmt == WZ_FMT_RGB24 ? f.w * 3 : f.w;          // bytes of a luma / RGB row
    const int cw = f.w >> 1;
    const int crowb = f.fmt == WZ_FMT_NV12 ? f.w : cw;               // bytes of a chroma row (NV12: interleaved; I420: per plane)

    auto in_row = [&](int oy) { return half_pixel ? ((float)oy + 0.5f) * f.scale_y - 0.5f : (float)oy * f.scale_y; };
    // LDS bytes of R output rows' spans, from above: (R - 1) scale + 3 source rows, half as many + 2 chroma rows per chroma run, 32 per run
    auto need = [&](int R) {
        const int rows = (int)((float)(R - 1) * f.scale_y) + 3;
        int b = rows * rowb + 32;
        if (f.fmt != WZ_FMT_RGB24) b += (f.fmt == WZ_FMT_NV12 ? 1 : 2) * (((rows >> 1) + 2) * crowb + 32);
        return b;
    };
    int R = 1;
    if (f.scale_y < 2.0f)
        while (R < WZ_PRE_ROWS_MAX && need(R + 1) <= budget) ++R;
    const int oy0 = (int)blockIdx.x * R;
    if (oy0 >= size) return;                                         // (the whole workgroup: no barrier has been passed)
    const int nrows = min(R, size - oy0);
    const int y0 = max((int)floorf(in_row(oy0)), 0);
    const int y1 = min((int)ceilf(in_row(oy0 + nrows - 1)), f.h - 1);
    const int cy0 = y0 >> 1, cy1 = y1 >> 1;

    // the byte runs: luma / RGB span, chroma span(s); each lands at the 16-byte boundary below its first byte
    // (Over-read: a run is fetched as the whole aligned 16-byte chunks that COVER it -- up to 15 bytes in front of its first and behind its
    //  last byte.  Every chunk holds at least one byte of the run and an aligned 16-byte chunk never straddles a page, so the extra bytes
    //  lie in a page the run itself lies in: inside what wz_host_register() page-locked (pinning is page-granular) or inside the lane's
    //  staging allocation, whatever the alignment of the registered range or of frame_stride.  They are never used.  ADVICE r4.)
    const uint8_t* const chroma = f.rgb + (size_t)f.w * f.h;
    const uint8_t* const src0 = f.rgb + (size_t)y0 * rowb;
    const uint8_t* const src1 = chroma + (size_t)cy0 * crowb;
    const uint8_t* const src2 = chroma + (size_t)cw * (f.h >> 1) + (size_t)cy0 * crowb;   // I420: the V plane
    const int nb0 = (y1 - y0 + 1) * rowb;
    const int nb1 = f.fmt != WZ_FMT_RGB24 ? (cy1 - cy0 + 1) * crowb : 0;
    const int nb2 = f.fmt == WZ_FMT_I420 ? nb1 : 0;
    const int sh0 = (int)(reinterpret_cast<uintptr_t>(src0) & 15), sh1 = (int)(reinterpret_cast<uintptr_t>(src1) & 15);
    const int sh2 = (int)(reinterpret_cast<uintptr_t>(src2) & 15);
    const int n0 = (sh0 + nb0 + 15) >> 4, n1 = nb1 ? (sh1 + nb1 + 15) >> 4 : 0, n2 = nb2 ? (sh2 + nb2 + 15) >> 4 : 0;   // 16-byte chunks
    const int lds1 = n0 * 16, lds2 = lds1 + n1 * 16;                 // where runs 1 and 2 start in LDS (run 0 at 0)
    {   // one list of chunks dealt out over the threads, a thread's loads (up to four per trip) requested before its first store: the
        // workgroup waits out one PCIe round trip per trip, not one per run
        const wz_u32x4* const g0 = reinterpret_cast<const wz_u32x4*>(src0 - sh0);
        const wz_u32x4* const g1 = reinterpret_cast<const wz_u32x4*>(src1 - sh1);
        const wz_u32x4* const g2 = reinterpret_cast<const wz_u32x4*>(src2 - sh2);
        const int total = n0 + n1 + n2;
        for (int c0 = 0; c0 < total; c0 += 4 * WZ_PRE_ROWS_THREADS) {
            wz_u32x4 v[4];
            bool on[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k * WZ_PRE_ROWS_THREADS + tid;
                on[k] = c < total;
                if (on[k]) v[k] = c < n0 ? g0[c] : c < n0 + n1 ? g1[c - n0] : g2[c - n0 - n1];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k * WZ_PRE_ROWS_THREADS + tid;
                if (on[k]) *reinterpret_cast<wz_u32x4*>(wz_pre_lds + (size_t)c * 16) = v[k];   // (the runs follow each other chunk by chunk)
            }
        }
    }
    __syncthreads();

    for (int r = 0; r < nrows; ++r) {
        const int oy = oy0 + r;
        const float in_y = in_row(oy);
        const float fl_y = floorf(in_y);
        const int y_lo = max((int)fl_y, 0);
        const int y_hi = min((int)ceilf(in_y), f.h - 1);
        const float ly = in_y - fl_y;
        for (int ox = tid; ox < size; ox += WZ_PRE_ROWS_THREADS) {
            const float in_x = half_pixel ? ((float)ox + 0.5f) * f.scale_x - 0.5f : (float)ox * f.scale_x;
            const float fl_x = floorf(in_x);
            const int x_lo = max((int)fl_x, 0);
            const int x_hi = min((int)ceilf(in_x), f.w - 1);
            const float lx = in_x - fl_x;
            float tap[4][3];   // top-left, top-right, bottom-left, bottom-right
            if (f.fmt == WZ_FMT_RGB24) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int y = q == 0 ? y_lo : y_hi;
                    wz_lds_row_pair(wz_pre_lds, sh0 + (y - y0) * rowb + x_lo * 3, x_hi != x_lo, tap[2 * q], tap[2 * q + 1]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int y = q == 0 ? y_lo : y_hi;
                    const uint8_t* const yrow = wz_pre_lds + sh0 + (y - y0) * rowb;
                    const int crow = ((y >> 1) - cy0) * crowb;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int x = t == 0 ? x_lo : x_hi;
                        int U, V;
                        if (f.fmt == WZ_FMT_NV12) {
                            const uint8_t* const uv = wz_pre_lds + lds1 + sh1 + crow + (x >> 1) * 2;
                            U = uv[0];
                            V = uv[1];
                        } else {
                            U = wz_pre_lds[lds1 + sh1 + crow + (x >> 1)];
                            V = wz_pre_lds[lds2 + sh2 + crow + (x >> 1)];
                        }
                        wz_yuv_to_rgb(yrow[x], U, V, tap[2 * q + t]);
                    }
                }
            }
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
            const size_t pix = (size_t)oy * size + ox;
            if constexpr (HP) {
                const half8_t o = {v[0], v[1], v[2], v[3], vl[0], vl[1], vl[2], vl[3]};
                *reinterpret_cast<half8_t*>(out + ((size_t)blockIdx.y * size * size + pix) * 8) = o;
            } else {
                const half4_t o = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<half4_t*>(out + ((size_t)blockIdx.y * size * size + pix) * 4) = o;
            }
        }
    }
}

// LDS bytes of the row-staged kernel for frames up to `max_w` wide: what one output row of the widest frame needs (3 RGB rows, or 3 luma rows
// and 2 x 3 chroma rows, + 32 per run), and at least 40 KiB so that frames whose resize skips no rows get several output rows per workgroup
size_t wz_preprocess_rows_lds(int max_w) {
    const size_t rgb = 3 * (size_t)max_w * 3 + 32;
    const size_t yuv = 3 * (size_t)max_w + 32 + 2 * (3 * (size_t)max_w + 32);
    size_t b = rgb > yuv ? rgb : yuv;
    if (b < 40 * 1024) b = 40 * 1024;
    return (b + 255) & ~(size_t)255;
}

// the kernels' last argument: bit 0 = half-pixel centres, bits 8 ..: the row-staged kernel's LDS budget in units of 256 bytes
int wz_preprocess_flags(bool half_pixel, int rows_lds) { return (half_pixel ? 1 : 0) | ((rows_lds >> 8) << 8); }

void wz_launch_preprocess(const WzFrameDesc* d_frames, int n, int size, half_t* out, hipStream_t s, bool hp, WzFrameDesc* keep,
                          bool half_pixel, const WzFrameDesc* by_value, int rows_lds) {
    dim3 grid((size * size + 255) / 256, n);
    WzDescPack pack;
    memset(&pack, 0, sizeof(pack));
    if (by_value && n <= WZ_DESC_PACK) {
        memcpy(pack.d, by_value, sizeof(WzFrameDesc) * n);
        d_frames = nullptr;
    }
    if (rows_lds > 0) {   // the row-staged form: one workgroup per output row
        dim3 rgrid(size, n);
        if (hp)
            WZ_LAUNCH(wz_k_preprocess_rows<true>, rgrid, dim3(WZ_PRE_ROWS_THREADS), rows_lds, s, d_frames, pack, size, out, keep, wz_preprocess_flags(half_pixel, rows_lds));
        else
            WZ_LAUNCH(wz_k_preprocess_rows<false>, rgrid, dim3(WZ_PRE_ROWS_THREADS), rows_lds, s, d_frames, pack, size, out, keep, wz_preprocess_flags(half_pixel, rows_lds));
        return;
    }
    if (hp)
        WZ_LAUNCH(wz_k_preprocess<true>, grid, dim3(256), 0, s, d_frames, pack, size, out, keep, half_pixel ? 1 : 0);
    else
        WZ_LAUNCH(wz_k_preprocess<false>, grid, dim3(256), 0, s, d_frames, pack, size, out, keep, half_pixel ? 1 : 0);
}
const void* wz_preprocess_func(bool hp, bool rows) {
    if (rows) return hp ? reinterpret_cast<const void*>(wz_k_preprocess_rows<true>) : reinterpret_cast<const void*>(wz_k_preprocess_rows<false>);
    return hp ? reinterpret_cast<const void*>(wz_k_preprocess<true>) : reinterpret_cast<const void*>(wz_k_preprocess<false>);
}
int wz_preprocess_rows_threads() { return WZ_PRE_ROWS_THREADS; }
