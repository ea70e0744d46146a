// Internal declarations shared by the HIP translation units of libwatsor_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/watsor_hip.h"
#include "wz_program.h"

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 half4_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
#ifdef __HIPCC__
// relu6(d + bias) -> four halves, zeroed as a whole when the pixel is outside the frame / the tile does not exist:
// two packed conversions and two selects on the packed words (selecting per element before the conversion costs the
// compiler four single conversions, two packs and four selects).  Same values, same rounding.
__device__ __forceinline__ half4_t wz_relu6_pack(const float4_t d, const float4_t bv, bool keep) {
    typedef __attribute__((ext_vector_type(2))) unsigned int wz_uint2_t;
    half4_t o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (half_t)fminf(fmaxf(d[r] + bv[r], 0.0f), 6.0f);
    wz_uint2_t p = __builtin_bit_cast(wz_uint2_t, o);
    p[0] = keep ? p[0] : 0u;
    p[1] = keep ? p[1] : 0u;
    return __builtin_bit_cast(half4_t, p);
}
#endif

// One frame handed to the pre-processing kernel.
struct WzFrameDesc {
    const uint8_t* rgb;   // the frame (device): packed RGB24 h x w x 3, or NV12 / I420 (h x w luma, then the 2x2-subsampled chroma)
    int32_t w, h;
    float scale_x, scale_y;   // (float)w / (float)size, TF legacy ResizeBilinear scale
    int32_t cam;              // camera filter index or -1
    int32_t fmt;              // WZ_FMT_* (include/watsor_hip.h)
};

struct WzConvArgs {
    const half_t* in;
    const half_t* w;
    const float* bias;
    const half_t* res;     // residual (same shape as the fp16 output) or nullptr
    void* out;             // half NHWC, or float head buffer, or float split-K workspace
    int32_t M;             // n * hout * wout
    int32_t hin, win, cin;
    int32_t hout, wout, cout, n_pad;
    int32_t ksize, stride, pad_t, pad_l, kc;
    int32_t act, out_mode;
    int32_t splitk;        // >1: raw fp32 partials to workspace [z][M][n_pad]
    int32_t kchunks;       // ksize*ksize*kc
    int64_t out_batch_stride;   // head modes: floats per frame in the concat buffer
    int64_t out_off;            // head modes: first float of this feature map
    float* out2;                // WZ_OUT_HEAD: class-logit buffer (out = box-encoding buffer)
    int64_t out2_batch_stride, out2_off;
    int32_t n_box;              // WZ_OUT_HEAD: columns [0, n_box) are box encodings, the rest class logits
    const half_t* zeros;        // 4 KiB of zeros in HBM: source of out-of-frame lanes / absent tiles in the LDS-tiled kernel
    int32_t grid_m, grid_n;     // LDS-tiled kernel: pixel tiles x channel tiles (filled in by the launcher)
    float* ws;                  // fp32 engine: split-K workspace (the fp16 kernels get it through `out`)
    unsigned long long* dbg;    // WZ_MB_DEBUG=1: 16 slots of phase timestamps (LDS-tiled kernel), else nullptr
    int32_t order;              // tile order of the LDS-tiled kernels (experiment knob WZ_LDS_ORDER)
    // In-launch split-K reduction of the SSD heads (splitk > 1 && inline_reduce): every K slice publishes its fp32 partial tile
    // write-through, takes a ticket on the tile's counter, and the LAST arriver sums the slices in slice order (the order of
    // wz_k_splitk_reduce: bit-identical) and finishes the outputs -- no reduce launch behind the convolution.
    int32_t nt_base;            // tile kernel: first 16-channel tile this launch entry serves (a head split along N, see wz_conv_rs_group_add)
    int32_t nt_live, nt_group;  // wide tile kernel (k_conv_wide.hip): 16-channel tiles that hold real columns / tiles per workgroup
    int32_t inline_reduce;
    int32_t frag_ws;            // the K slices' partial sums lie in FRAGMENT order: [z][M / 16][n_pad / 16][64 lanes][4] (wide tile kernel -> grouped reduce)
    int32_t fin_flags;          // bit 0: decode the boxes, bit 1: mark the NMS candidates (see WzHeadFinish)
    int32_t* tickets;           // one counter per output tile of this convolution, zero between launches
    const struct WzHeadFinish* fin;   // device-resident, per lane
};


struct WzPostConsts {
    int32_t num_anchors, num_classes;   // classes incl. background
    int32_t max_total, max_per_class;
    float score_thr, iou_thr;
    float scale_y, scale_x, scale_h, scale_w;
    int32_t clip_after;   // 1: the per-class NMS sees the boxes as decoded, what it selects is clipped afterwards (WzBlobHeader::post_flags)
    int32_t _pad;
};

// What finishing a head output needs beyond the convolution's own arguments (static per lane, lives in HBM).
struct WzHeadFinish {
    const float* hint_logit;    // [n] see WzPostBuffers
    uint32_t* cbits;            // [n][cbits_words]
    int32_t cbits_words, _pad;
    WzPostConsts pc;
    const float* anchors;       // [A][4]
    float* boxes;               // [n][A][4] decoded + clipped
    uint8_t* valid;             // [n][A]
};

// One fused inverted-residual block (k_mbconv.hip).  cin/kc/n_pad/cout describe the project conv.
struct WzMbArgs {
    const half_t* in;      // block input, NHWC fp16: cin0 channels (expand) or cmid channels (no expand)
    const half_t* we;      // expand weights, MFMA A fragments [nmid_pad/16][kc0][64][8]
    const float* be;       // [nmid_pad]
    const half_t* wd;      // depthwise weights [9][cmid_pad]
    const float* bd;       // [cmid_pad]
    const half_t* wp;      // project weights, MFMA A fragments [n_pad/16][kc][64][8]
    const float* bp;       // [n_pad]
    const half_t* res;     // residual (shape of out) or nullptr
    half_t* out;           // NHWC fp16, cout channels
    int32_t hin, win, hout, wout;
    int32_t cin0, kc0, nmid_pad;   // expand: input channels (0 = no expand stage), K chunks, packed columns
    int32_t cmid, cmid_pad, kc;    // depthwise channels, padded row, K chunks of the project conv
    int32_t cout, n_pad;
    int32_t stride, pad_t, pad_l;
    int32_t stem, sin_h, sin_w, spad_t, spad_l;   // stem fused in: `in` is the sin_h x sin_w x 4 network input, cin0 = 32 (K 27 padded)
    float* ws;             // fp32 workspace for channel-group partial sums (nullptr: never split)
    uint64_t ws_bytes;
    int32_t M;             // n * hout * wout
    unsigned long long* dbg;   // diagnostics: 16 timestamps (first / last workgroup), or nullptr
    int32_t th, tw, tiles_y, tiles_x, nsplit, cpg, stage, ebufs, nb;   // filled in by the launcher (nb = frames)
    // split-operand blocks (k_mbconv_hp.hip): `in` / `res` are hi + lo pair tensors, `wd` points at float[9][cmid_pad]
    const half_t* we_lo;   // "lo" halves of the expand weights (we = "hi")
    const half_t* wp_lo;   // "lo" halves of the project weights
    int32_t hp, hp_out;    // hp: this block runs on the split-operand kernel; hp_out: 1 = `out` is a hi + lo pair tensor, 2 = `out` has 2 * cout plain
                           // channels holding the fp16 output TWICE ([hi | hi]: the consumer multiplies them with the hi and lo halves of its weights)
    half_t* out2;          // chunk-split kernel: where the expanded tensor is stored as well (hin x win x cmid fp16), or nullptr
    int32_t has_out2;      // the op has such a second output (out2 itself is null while a launcher is only asked to prepare)
    int32_t qenc;          // split-operand kernel: the chunk buffer holds the 16-bit float form of v / 6 (the robust program; wd carries 6 / K, bd the offset)
    unsigned long long* dbg2;   // two-launch form (k_mbconv_hp2.hip): the second launch's stamp block (WZ_LANE_STAMPS builds), else nullptr
    int32_t lone;               // the batch is being launched kernel by kernel because every other lane is idle (run_batch): launch shapes may use the whole chip
                                // (only shapes whose results are bit-identical to the throughput shapes' may depend on it)
};

// Per-camera filter state resident in HBM (see wz_set_camera_filter).
struct WzCamFilter {
    int32_t enabled, width, height, n_zones;   // enabled: bit 0 = filters on, bit 1 = drop mode (failing rows zeroed)
    const int32_t* sat;          // [n_zones][(height+1)][(width+1)] inclusive-prefix sums, row/col 0 = 0
    double conf_thr[WZ_NUM_LABELS];   // NaN = label not configured
    double area_thr[WZ_NUM_LABELS];
    uint8_t allow[WZ_NUM_LABELS][WZ_MAX_ZONES * 4]; // up to 40 zones tracked; allow[l][z]
};
#define WZ_MAX_ZONES_PER_CAM (WZ_MAX_ZONES * 4)


// ---- launchers (each enqueues exactly one kernel on `s`) ------------------------------------
// ... except under wz_profile_stages(), which has the network's launchers enqueue every kernel `wz_launch_repeat` times
// back to back (all of them are pure functions of their inputs): the bracket around a stage then holds N launches and
// their N - 1 in-stream boundaries, and the one-off cost of the event pair is amortised instead of estimated.
extern thread_local int wz_launch_repeat;
// Tuning and A/B knobs (the WZ_* variables named next to the code they steer) exist in the DEVELOPMENT build only
// (`make dev`: -DWZ_DEV_BUILD, libwatsor_hip_dev.so -- what tools/ and the stage-level parity tests load).  The product library
// takes every default and reads exactly four operator settings from the environment: WZ_LANES, WZ_STREAMS, WZ_GRAPH, WZ_SCHEDULE.
static inline const char* wz_dev_getenv(const char* name) {
#ifdef WZ_DEV_BUILD
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// The SCHEDULE of a process (wz_set_schedule() before its first engine, else WZ_SCHEDULE in the environment, else throughput; fixed the
// first time anything asks): "latency" = the launch shapes that make ONE batch finish soonest on an otherwise idle GPU -- eight waves
// per 19x19 tile, the 10x10 blocks on 256 workgroups (+ a reduce launch in the default program), the SSD heads' K slices cut for the
// whole chip, page-locked host frames read in place.  "throughput" (default): the shapes that leave room for the other lanes'
// launches -- 9 % more frames/s with four lanes in flight, 3 % more latency of a lone batch (DESIGN.md section 5,
// profiles/r03_wave_counts_and_cu_footprints.txt).  Defined in wz_engine.hip.
bool wz_latency_schedule();

// ---- lane stamps (`make stamps`: -DWZ_DEV_BUILD -DWZ_LANE_STAMPS=1 -> libwatsor_hip_stamps.so; tools/lane_overlap.py) -----------------
// What rocprofv3 cannot show on this stack: which kernels of DIFFERENT lanes are on the chip at the same time (under its kernel trace the
// four lanes serialise, concurrency 1.18; unprofiled the headline needs > 2).  In this build every kernel of a batch stamps the constant
// 100 MHz clock (s_memrealtime: one clock for all XCDs) into its launch's block of the lane -- PLAIN stores, no atomics (a first version
// with one device-scope atomic min / max per workgroup on one address cost 21 % of the throughput: 2 816 of them per resize launch
// queue up at the memory side): the first 64 workgroups each store their entry time into a word of their own ([16 + id]); every
// workgroup stores its exit time into bucket [80 + id % 256] -- workgroups are dealt round-robin over the 8 XCDs, so all writers of a
// bucket share one L2 and the last one to leave is the value that stays.  The chain's last kernel (wz_k_nms) reduces a launch's words to
// (earliest entry, latest exit), hands the pairs to the lane's page-locked stamp block and resets the words; the host reads that
// block after wz_wait().  Nothing of this exists in the product or the development library.
#ifndef WZ_LANE_STAMPS
#define WZ_LANE_STAMPS 0
#endif
#define WZ_STAMP_SLOTS 128            // launches of a batch that can be stamped (the robust program has 30, the per-layer one ~75)
#define WZ_STAMP_WORDS 384            // 64-bit words per launch block: [0 .. 15] the older per-kernel phase stamps, [16 .. 79] entry times of
                                      // workgroups 0 .. 63, [80 .. 335] exit-time buckets (workgroup id mod 256)
#define WZ_STAMP_ENTRY 16
#define WZ_STAMP_EXIT 80
#define WZ_STAMP_PRE_BYTES (WZ_STAMP_WORDS * 8)   // the resize kernel's block lies this far in front of the descriptors it leaves behind (`keep`)
struct WzLaunchNote { const void* func; unsigned grid[3], block[3]; unsigned lds; };
#if WZ_LANE_STAMPS
void wz_note_launch(const void* func, dim3 grid, dim3 block, size_t lds);
#define WZ_LAUNCH(kern, grid, block, lds, s, ...) do { wz_note_launch(reinterpret_cast<const void*>(kern), grid, block, lds); \
        for (int _wz_r = 0; _wz_r < wz_launch_repeat; ++_wz_r) hipLaunchKernelGGL(kern, grid, block, lds, s, __VA_ARGS__); } while (0)
#ifdef __HIPCC__
struct WzLaneStamp {
    unsigned long long* p;
    __device__ __forceinline__ explicit WzLaneStamp(unsigned long long* q) : p(q) {
        if (p && threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
            const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            if (id < 64u) p[WZ_STAMP_ENTRY + id] = (unsigned long long)wall_clock64();   // (the first workgroups dispatched hold the earliest entry)
        }
    }
    __device__ __forceinline__ ~WzLaneStamp() {
        if (p && threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
            const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            p[WZ_STAMP_EXIT + (id & 255u)] = (unsigned long long)wall_clock64();
        }
    }
};
#define WZ_LANE_STAMP(ptr) WzLaneStamp _wz_lane_stamp(const_cast<unsigned long long*>(ptr))
#endif
#else
#define WZ_LAUNCH(...) do { for (int _wz_r = 0; _wz_r < wz_launch_repeat; ++_wz_r) hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define WZ_LANE_STAMP(ptr) do { } while (0)
#endif
// hp: the input tensor is a hi + lo pair (8 halves per pixel: r g b 0 | r g b 0)
#define WZ_DESC_PACK 16
struct WzDescPack { WzFrameDesc d[WZ_DESC_PACK]; };   // frame descriptors as kernel arguments (512 bytes)
// keep, half_pixel, by_value (host descriptors to pass as kernel arguments when n <= WZ_DESC_PACK): see k_preprocess.hip
void wz_launch_preprocess(const WzFrameDesc* d_frames, int n, int size, half_t* out, hipStream_t s, bool hp = false,
                          WzFrameDesc* keep = nullptr, bool half_pixel = false, const WzFrameDesc* by_value = nullptr, int rows_lds = 0);
// rows_lds > 0: the row-staged form of the kernel (one workgroup per output row, `rows_lds` bytes of LDS = wz_preprocess_rows_lds(widest frame))
size_t wz_preprocess_rows_lds(int max_w);
int wz_preprocess_flags(bool half_pixel, int rows_lds);   // the resize kernels' last argument (k_preprocess.hip)
int wz_preprocess_rows_threads();
const void* wz_preprocess_func(bool hp, bool rows = false);   // the kernel's host-side address (to find its node in a captured graph)
void wz_launch_stem(const half_t* in, const float* w, const float* bias, half_t* out, int n, int hin, int win,
                    int hout, int wout, int pad_t, int pad_l, hipStream_t s);
void wz_launch_dw(const half_t* in, const half_t* w, const float* bias, half_t* out, int n, int hin, int win,
                  int c, int hout, int wout, int stride, int pad_t, int pad_l, int act, hipStream_t s);
void wz_launch_conv(const WzConvArgs& a, hipStream_t s);
void wz_launch_splitk_reduce(const WzConvArgs& a, const float* ws, hipStream_t s);
// several split-K reductions (each with its own epilogue arguments and partial sums) in ONE launch
#define WZ_REDUCE_GROUP_MAX 8
struct WzReduceGroup {
    int32_t n;
    int32_t first[WZ_REDUCE_GROUP_MAX + 1];   // first workgroup of entry i; first[n] = grid size
    WzConvArgs a[WZ_REDUCE_GROUP_MAX];
    const float* ws[WZ_REDUCE_GROUP_MAX];
    // decode != 0: the thread that finishes the four box-encoding columns of an anchor also decodes + clips the box
    // (exactly wz_k_decode's arithmetic) and the launch clears hist / count / band -- wz_k_decode is then not launched
    int32_t decode, n_frames;
    // list != 0: class logits at or above the frame's hint_logit get their bit set in cbits[f] -- the first band of
    // wz_k_nms then needs no scan of the logits at all
    int32_t list, cbits_words;
    unsigned long long* stamp;   // WZ_LANE_STAMPS builds: this launch's block of the lane's stamps, else nullptr
    const float* hint_logit;
    uint32_t* cbits;
    WzPostConsts pc;
    const float* anchors;
    float* boxes;
    uint8_t* valid;
    uint32_t *hist, *count, *band;
};
void wz_reduce_group_add(WzReduceGroup& g, const WzConvArgs& a, const float* ws);
// several independent small 3x3 convolutions (wz_k_conv<3, 2, 2, 4> shapes) in ONE launch
#define WZ_CONV_GROUP_MAX 6
struct WzConvGroup {
    int32_t n;
    int32_t first[WZ_CONV_GROUP_MAX + 1];
    int32_t gx[WZ_CONV_GROUP_MAX], gy[WZ_CONV_GROUP_MAX];
    WzConvArgs a[WZ_CONV_GROUP_MAX];
    int32_t* tickets;           // host side only: the lane's counter block and how much of it the entries added so far use
    int32_t ticket_off;
    unsigned long long* stamp;  // WZ_LANE_STAMPS builds: this launch's block of the lane's stamps, else nullptr
};
// split-K across the waves of a workgroup (no partials in HBM, no reduce launch): the extras chain
bool wz_conv_ws_applies(const WzConvArgs& a);
void wz_launch_conv_ws(const WzConvArgs& a, hipStream_t s);
bool wz_conv_groupable(const WzConvArgs& a);
void wz_conv_group_add(WzConvGroup& g, const WzConvArgs& a);
void wz_launch_conv_group(const WzConvGroup& g, hipStream_t s);
// ... and the convolutions of the register-staged 128 x 128 tile kernel (the two big heads)
bool wz_conv_rs_groupable(const WzConvArgs& a);
int wz_conv_rs_group_add(WzConvGroup& g, const WzConvArgs& a);   // entries added (a head may be split along N), 0 = full
void wz_launch_conv_rs_group(const WzConvGroup& g, hipStream_t s);
// ... and the wide tile kernel (k_conv_wide.hip: 128 pixels x up to 320 channels per workgroup), which serves the big heads by default
bool wz_conv_wide_applies(const WzConvArgs& a);
int wz_conv_wide_ntw();                                                 // channel tiles per wave of the build in use (5, or 3: two workgroups per CU)
void wz_conv_wide_shape(const WzConvArgs& a, int* tiles, int* steps);   // workgroup tiles and K steps (of 64 channels x one tap)
int wz_choose_wide_T(const int* tiles, const int* steps, const long long* tile_bytes, int n, int cus);   // steps per K slice
int wz_conv_wide_group_add(WzConvGroup& g, const WzConvArgs& a);         // a.splitk set by the caller; 0 = the group is full
void wz_launch_conv_wide_group(const WzConvGroup& g, hipStream_t s);
void wz_launch_splitk_reduce_group(const WzReduceGroup& g, hipStream_t s);
int wz_choose_splitk(int M, int n_pad, int kchunks);
bool wz_conv_use_lds(const WzConvArgs& a);               // the LDS-tiled kernel will serve this conv
int wz_choose_splitk_lds(int M, int n_pad, int kchunks);
int wz_lds_nw(int M, int n_pad, int kchunks);             // 16-channel tiles per wave pair: 2 -> 64-channel tile, 4 -> 128
bool wz_conv_f32_use_rs(const WzConvArgs& a);            // fp32 engine: the register-staged tile kernel will serve this conv
int wz_choose_splitk_rs_f32(int M, int n_pad, int kchunks);
bool wz_conv_ws_f32_applies(const WzConvArgs& a);         // fp32 engine: the extras chain on the wave-split kernel
void wz_launch_conv_ws_f32(const WzConvArgs& a, hipStream_t s);
void wz_conv_init();
// a 1x1 convolution and the 3x3 stride-2 convolution behind it on the small maps of the extras chain, in one launch (k_extras_pair.hip)
bool wz_extras_pair_applies(const WzConvArgs& a, const WzConvArgs& b);
void wz_launch_extras_pair(const WzConvArgs& a, const WzConvArgs& b, int n, hipStream_t s);
// `-p 32` engine (k_f32.hip): fp32 activations and weights, exact-fp32 MFMA
void wz_launch_stem_f32(const half_t* in, const float* w, const float* bias, float* out, int n, int hin, int win,
                        int hout, int wout, int pad_t, int pad_l, hipStream_t s, bool pair = false);
void wz_launch_dw_f32(const float* in, const float* w, const float* bias, float* out, int n, int hin, int win, int c,
                      int hout, int wout, int stride, int pad_t, int pad_l, int act, hipStream_t s);
void wz_launch_conv_f32(const WzConvArgs& a, hipStream_t s, bool reduce = true);   // + its split-K reduce when a.splitk > 1 (unless !reduce)
int wz_launch_mbconv(const WzMbArgs& a, int n, hipStream_t s, bool prepare);   // -1: no kernel; else #channel groups
int wz_launch_mbconv_wave(const WzMbArgs& a, int n, hipStream_t s, bool prepare);   // wave-per-tile variant; -2: not applicable
int wz_launch_mbconv_cs(const WzMbArgs& a, int n, hipStream_t s, bool prepare);     // channels split over waves (small maps); -2: n/a
int wz_launch_mbconv_hp(const WzMbArgs& a, int n, hipStream_t s, bool prepare);     // split-operand blocks; -1: no kernel for this shape
// ... the 10x10 ones of the robust program as TWO launches (k_mbconv_hp2.hip: expand + depthwise per band and chunk -> project fragments in the
// workspace; then a plain split-operand GEMM over the whole batch's pixels).  applies: 1 = this block takes that form (its only one).
// phase 0: both launches, 1: the first only, 2: the second only; prepare: kernel attributes (phase ignored).  -1: no kernel for this shape
int wz_mbconv_hp2_applies(const WzMbArgs& a, int n);
int wz_launch_mbconv_hp2(const WzMbArgs& a, int n, hipStream_t s, bool prepare, int phase);

#define WZ_HIST_BINS 1024
#define WZ_CAND_CAP 4096
#define WZ_NMS_KEEP_MAX 128   // capacity of the NMS walk's kept list (k_post.hip); >= max_total (100)
#define WZ_CAND_TARGET 192
struct WzPostBuffers {
    const float* box_enc;     // [n][A][4]
    const float* logits;      // [n][A][C]
    const float* anchors;     // [A][4] (ycenter, xcenter, h, w)
    float* boxes;             // [n][A][4] decoded + clipped
    uint8_t* valid;           // [n][A]   clipped area > 0
    uint32_t* hist;           // [n][WZ_HIST_BINS]
    uint32_t* count;          // [n]  (directly behind hist so one memset clears both)
    uint32_t* band;           // [n][2] threshold bin of band 0 and the frame's candidate total (written by wz_k_compact)
    uint32_t* hint;           // [n] self-scan mode of wz_k_nms: the score bin the first band of this frame slot started at last time
    float* hint_logit;        // [n] wz_logit_floor(hint): what the grouped head reduce compares the finished logits with
    uint32_t* cbits;          // [n][ceil(A*C/32)] one bit per class logit: set by the grouped head reduce where the logit can
                              // reach the frame's first band (fire-and-forget atomicOr), read and cleared by wz_k_nms
    uint2* cand;              // [n][WZ_CAND_CAP] (score bits, tie index c*A + a)
    float* det_boxes;         // [n][100][4]
    float* det_scores;        // [n][100]
    int32_t* det_classes;     // [n][100] 1-based
    int32_t* det_num;         // [n]
    unsigned long long* dbg;  // [n][16] phase timestamps of wz_k_nms (wall_clock64, 100 MHz), diagnostics only
    // WZ_LANE_STAMPS builds (else all null / 0): the lane's launch blocks, the resize kernel's block, the page-locked copy the host reads
    // (pairs: [2k] entry, [2k + 1] exit of launch k; launch 0 = the resize kernel; wz_k_nms's own pairs follow, one per frame) and
    // how many launches of this batch were stamped
    unsigned long long* stamps;
    unsigned long long* stamps_pre;
    unsigned long long* stamps_host;
    int32_t stamps_n, _stamps_pad;
};
void wz_launch_decode(const WzPostBuffers& b, const WzPostConsts& c, int n, hipStream_t s);
void wz_launch_hist(const WzPostBuffers& b, const WzPostConsts& c, int n, hipStream_t s);
void wz_launch_compact(const WzPostBuffers& b, const WzPostConsts& c, int n, hipStream_t s);
// d_frames != nullptr: the kernel also writes the Detection rows + pass bytes (then no wz_launch_rows is needed)
// self_scan: the kernel selects its candidates itself (no wz_k_hist / wz_k_compact in front of it)
void wz_launch_nms(const WzPostBuffers& b, const WzPostConsts& c, int n, hipStream_t s, const WzFrameDesc* d_frames = nullptr,
                   const WzCamFilter* d_cams = nullptr, wz_detection_t* rows = nullptr, uint8_t* pass = nullptr,
                   bool self_scan = false, bool listed = false, uint32_t* status = nullptr);   // status: see wz_k_nms
void wz_launch_rows(const WzPostBuffers& b, const WzFrameDesc* d_frames, const WzCamFilter* d_cams, int n,
                    int max_total, wz_detection_t* rows, uint8_t* pass, hipStream_t s);
int wz_set_error(int code, const char* fmt, ...);   // sets wz_last_error() of the calling thread, returns code
void wz_post_init();   // one-time kernel attributes (must run before any stream capture)
void wz_launch_filter_rows(const WzCamFilter* d_cams, int cam, wz_detection_t* rows, uint8_t* pass, hipStream_t s);
void wz_launch_sat(const uint8_t* fill, int32_t* sat, int width, int height, int n_zones, hipStream_t s);
