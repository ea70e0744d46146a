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

// ---------------------------------------------------------------------------------------------------------------------------------
// Row-staged form (round 4): one workgroup per OUTPUT ROW of one frame.  The two source rows that output row taps (for NV12 / I420
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
// LDS: [row y_lo | row y_hi | chroma rows], each run starting at the 16-byte boundary below its first byte.
struct WzRowRun {
    const uint8_t* src;   // first byte wanted
    int bytes;            // how many
    int lds;              // where the 16-byte granule holding `src` lands in LDS (multiple of 16)
};

__device__ __forceinline__ void wz_stage_run(const WzRowRun r, uint8_t* __restrict__ lds, int tid, int nthreads) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(r.src);
    const int shift = (int)(a & 15);
    const uint4* __restrict__ g = reinterpret_cast<const uint4*>(a - shift);     // (stays inside the granule of the first valid byte)
    const int chunks = (shift + r.bytes + 15) >> 4;
    uint4* __restrict__ d = reinterpret_cast<uint4*>(lds + r.lds);
    for (int c = tid; c < chunks; c += nthreads) d[c] = g[c];
}

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
template <bool HP>
__global__ __launch_bounds__(WZ_PRE_ROWS_THREADS) void wz_k_preprocess_rows(const WzFrameDesc* __restrict__ frames, const WzDescPack pack,
                                                                            int size, half_t* __restrict__ out,
                                                                            WzFrameDesc* __restrict__ keep, int half_pixel) {
    extern __shared__ __attribute__((aligned(16))) uint8_t wz_pre_lds[];
    const WzFrameDesc f = frames ? frames[blockIdx.y] : pack.d[blockIdx.y];
    if (keep && blockIdx.x == 0 && threadIdx.x == 0) keep[blockIdx.y] = f;
    const int oy = blockIdx.x, tid = threadIdx.x;

    const float in_y = half_pixel ? ((float)oy + 0.5f) * f.scale_y - 0.5f : (float)oy * f.scale_y;
    const float fl_y = floorf(in_y);
    const int y_lo = max((int)fl_y, 0);
    const int y_hi = min((int)ceilf(in_y), f.h - 1);
    const float ly = in_y - fl_y;

    // the byte runs of this output row
    const int rowb = f.fmt == WZ_FMT_RGB24 ? f.w * 3 : f.w;          // bytes of a luma / RGB row
    const int slot = (rowb + 31 + 15) & ~15;                         // LDS bytes reserved per run (shift <= 15, rounded up)
    // (fixed places, constant indices only: the array lives in registers.  0 / 1: rows y_lo / y_hi; 2 / 3: chroma of y_lo -- NV12's
    // interleaved row, or I420's U and V rows; 4 / 5: the same for y_hi when it lies in another chroma row; bytes = 0: unused)
    WzRowRun run[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) run[r] = {f.rgb, 0, 0};
    run[0] = {f.rgb + (size_t)y_lo * rowb, rowb, 0};
    run[1] = {f.rgb + (size_t)y_hi * rowb, rowb, slot};
    int c_lo = 2 * slot, c_hi = 2 * slot, v_off = 0;                 // LDS offsets of the chroma runs of y_lo / y_hi (+ v_off: the V run, I420)
    if (f.fmt != WZ_FMT_RGB24) {
        const uint8_t* chroma = f.rgb + (size_t)f.w * f.h;
        const int cw = f.w >> 1, cy_lo = y_lo >> 1, cy_hi = y_hi >> 1;
        if (f.fmt == WZ_FMT_NV12) {
            run[2] = {chroma + (size_t)cy_lo * f.w, f.w, c_lo};
            if (cy_hi != cy_lo) {
                c_hi = c_lo + slot;
                run[4] = {chroma + (size_t)cy_hi * f.w, f.w, c_hi};
            }
        } else {   // I420: U plane, then V plane, rows of w / 2 bytes
            const int cslot = (cw + 31 + 15) & ~15;
            v_off = cslot;
            const size_t vplane = (size_t)cw * (f.h >> 1);
            run[2] = {chroma + (size_t)cy_lo * cw, cw, c_lo};
            run[3] = {chroma + vplane + (size_t)cy_lo * cw, cw, c_lo + v_off};
            if (cy_hi != cy_lo) {
                c_hi = c_lo + 2 * cslot;
                run[4] = {chroma + (size_t)cy_hi * cw, cw, c_hi};
                run[5] = {chroma + vplane + (size_t)cy_hi * cw, cw, c_hi + v_off};
            }
        }
    }
    // All runs as ONE list of 16-byte chunks dealt out over the threads, a thread's loads (up to four per trip) requested before its first
    // store: a workgroup waits out one PCIe round trip, not one per run (run by run a 1080p NV12 row was four of them in a row).
    {
        // (a plain vector type for the chunks in flight: an array of HIP's uint4 structs is kept in scratch memory)
        typedef unsigned int wz_u32x4 __attribute__((ext_vector_type(4)));
        auto gran = [](const WzRowRun& r) { return reinterpret_cast<const wz_u32x4*>(reinterpret_cast<uintptr_t>(r.src) & ~(uintptr_t)15); };
        auto count = [](const WzRowRun& r) { return r.bytes ? (int)((reinterpret_cast<uintptr_t>(r.src) & 15) + r.bytes + 15) >> 4 : 0; };
        const int f1 = count(run[0]), f2 = f1 + count(run[1]), f3 = f2 + count(run[2]), f4 = f3 + count(run[3]), f5 = f4 + count(run[4]);
        const int total = f5 + count(run[5]);
        for (int c0 = 0; c0 < total; c0 += 4 * WZ_PRE_ROWS_THREADS) {
            wz_u32x4 v[4];
            int where[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k * WZ_PRE_ROWS_THREADS + tid;
                where[k] = -1;
                if (c < total) {
                    const wz_u32x4* gp = gran(run[0]);
                    int off = c, l = run[0].lds;
                    if (c >= f1) { gp = gran(run[1]); off = c - f1; l = run[1].lds; }
                    if (c >= f2) { gp = gran(run[2]); off = c - f2; l = run[2].lds; }
                    if (c >= f3) { gp = gran(run[3]); off = c - f3; l = run[3].lds; }
                    if (c >= f4) { gp = gran(run[4]); off = c - f4; l = run[4].lds; }
                    if (c >= f5) { gp = gran(run[5]); off = c - f5; l = run[5].lds; }
                    v[k] = gp[off];
                    where[k] = l + off * 16;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (where[k] >= 0) *reinterpret_cast<wz_u32x4*>(wz_pre_lds + where[k]) = v[k];
        }
    }
    __syncthreads();

    for (int ox = tid; ox < size; ox += WZ_PRE_ROWS_THREADS) {
        const float in_x = half_pixel ? ((float)ox + 0.5f) * f.scale_x - 0.5f : (float)ox * f.scale_x;
        const float fl_x = floorf(in_x);
        const int x_lo = max((int)fl_x, 0);
        const int x_hi = min((int)ceilf(in_x), f.w - 1);
        const float lx = in_x - fl_x;
        float tap[4][3];   // top-left, top-right, bottom-left, bottom-right
        if (f.fmt == WZ_FMT_RGB24) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int sh = (int)(reinterpret_cast<uintptr_t>(run[r].src) & 15);
                wz_lds_row_pair(wz_pre_lds + run[r].lds, sh + x_lo * 3, x_hi != x_lo, tap[2 * r], tap[2 * r + 1]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int sh = (int)(reinterpret_cast<uintptr_t>(run[r].src) & 15);
                const uint8_t* yrow = wz_pre_lds + run[r].lds + sh;
                const int cbase = r == 0 ? c_lo : c_hi;
                const int y = r == 0 ? y_lo : y_hi;
                const uint8_t* csrc = f.rgb + (size_t)f.w * f.h;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int x = t == 0 ? x_lo : x_hi;
                    int U, V;
                    if (f.fmt == WZ_FMT_NV12) {
                        const int csh = (int)(reinterpret_cast<uintptr_t>(csrc + (size_t)(y >> 1) * f.w) & 15);
                        const uint8_t* uv = wz_pre_lds + cbase + csh + (x >> 1) * 2;
                        U = uv[0];
                        V = uv[1];
                    } else {
                        const int cw = f.w >> 1;
                        const int ush = (int)(reinterpret_cast<uintptr_t>(csrc + (size_t)(y >> 1) * cw) & 15);
                        const int vsh = (int)(reinterpret_cast<uintptr_t>(csrc + (size_t)cw * (f.h >> 1) + (size_t)(y >> 1) * cw) & 15);
                        U = wz_pre_lds[cbase + ush + (x >> 1)];
                        V = wz_pre_lds[cbase + v_off + vsh + (x >> 1)];
                    }
                    wz_yuv_to_rgb(yrow[x], U, V, tap[2 * r + t]);
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

// LDS bytes of the row-staged kernel for frames up to `max_w` wide: 2 luma / RGB runs + up to 2 x (U + V) or 2 NV12 chroma runs
size_t wz_preprocess_rows_lds(int max_w) {
    const size_t rgb = 2 * (size_t)((max_w * 3 + 31 + 15) & ~15);
    const size_t yuv = 2 * (size_t)((max_w + 31 + 15) & ~15) + 4 * (size_t)((max_w + 31 + 15) & ~15);
    return rgb > yuv ? rgb : yuv;
}

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
            WZ_LAUNCH(wz_k_preprocess_rows<true>, rgrid, dim3(WZ_PRE_ROWS_THREADS), rows_lds, s, d_frames, pack, size, out, keep, half_pixel ? 1 : 0);
        else
            WZ_LAUNCH(wz_k_preprocess_rows<false>, rgrid, dim3(WZ_PRE_ROWS_THREADS), rows_lds, s, d_frames, pack, size, out, keep, half_pixel ? 1 : 0);
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
