// Runtime + C ABI of libwatsor_hip.so (see include/watsor_hip.h for the contract of every entry point).
//
// One engine = one MI355X = one HIP stream.  A batch is one dependent chain of kernels
// (pre-process -> ~70 conv ops -> post-process -> rows) captured once per (slot, batch size) into a
// hipGraph and replayed; frame descriptors travel through a pinned host block that the graph's first
// node copies to HBM, results come back through a pinned row block that its last nodes fill.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <string>
#include <vector>

#include "wz_common.h"

// Lanes are HIP streams; ROCclr multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two
// lanes sharing one queue serialise (measured: 4 lanes 23k frames/s on 4 queues, 34k on 8).  The variable is read
// when the HIP runtime initialises, i.e. at the first HIP call, which cannot precede this constructor.
__attribute__((constructor)) static void wz_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

thread_local int wz_launch_repeat = 1;
static thread_local char g_err[512] = "";
static int wz_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int wz_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return wz_fail(WZ_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                             __FILE__, __LINE__);                                     \
    } while (0)

#define WZ_TICKETS 8192   // tile counters per lane: first half the tile-kernel heads, second half the small ones

struct StageTimer {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipStream_t s = nullptr;
    void mark() {
        if (used == ev.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            ev.push_back(e);
        }
        (void)hipEventRecord(ev[used++], s);
    }
};

struct wz_engine {
    int device = 0;
    hipStream_t stream = nullptr;            // = lanes[0].stream (stage-level entry points, filters)
    std::string name;
    std::vector<uint8_t> blob;
    WzBlobHeader hdr;
    const WzTensorDesc* tensors = nullptr;
    const WzOpDesc* ops = nullptr;
    int max_batch = 0, max_w = 0, max_h = 0;
    bool no_reuse = false, use_graph = true, use_splitk = true;
    bool graph_adaptive = false;   // use_graph && WZ_GRAPH unset: a batch that finds every other lane idle is launched kernel by kernel (wz_create)
    bool wide_frag = true;     // the wide head kernel's partial sums in fragment order (WZ_WIDE_FRAG=0: [slice][pixel][column])
    bool list_cands = true;    // WZ_LIST_CANDS=0: the self-scanning NMS kernel always scans
    bool post_self = true;     // WZ_POST_SELF=0: histogram + compaction kernels in front of the NMS kernel
    bool fuse_decode = true;   // WZ_FUSE_DECODE=0: keep wz_k_decode as its own launch
    bool defer_heads = true;   // the SSD heads' split-K reductions run as one launch after the last head (WZ_DEFER_HEADS=0: one each)
    bool conv_wide = true;     // the big SSD heads on the wide tile kernel (k_conv_wide.hip); WZ_CONV_WIDE=0: on wz_k_conv_rs
    int wide_min_m = 1;        // WZ_WIDE_MIN_M=n: heads with fewer output pixels than this (per batch) stay on wz_k_conv_group
    int wide_T = 0;            // WZ_WIDE_T=n: K steps per slice of that kernel (0: chosen per launch by wz_choose_wide_T)
                                  // 2 (round 5): only the chain behind the 3x3 map -- its last four convolutions, 0.8 MB of weights per frame's workgroup
    bool desc_by_value = true;    // the frame descriptors travel as arguments of the resize kernel (WZ_DESC_ARGS=0: zero-copy / copied)
    bool desc_zero_copy = true;   // the resize kernel reads the frame descriptors from page-locked host memory (WZ_DESC_COPY=1: copied first)
    bool pre_rows = false;        // WZ_PRE_ROWS=1: every batch on the row-staged form of the resize kernel (k_preprocess.hip:
                                  // wz_k_preprocess_rows); default: only batches with a frame that is read in place (the per-pixel form is
                                  // faster out of HBM: 51.5 k against 48.0 k frames/s on the headline workload, profiles/r04_host_read_ab.txt)
    int pre_rows_lds = 0;         // ... and its LDS bytes for the widest frame this engine takes
    int host_read = 2;            // page-locked host frames (WZ_HOST_READ): 0 = staged by one DMA each, 1 = read in place by the resize kernel
                                  // (no copy), 2 (default) = read in place when the resize skips rows (vertical down-scale >= 2: 1080p RGB24
                                  // 8.0 k -> 13.0 k frames/s, NV12 15.6 k -> 18.7 k), staged otherwise (640x480 in place: 26 k against 33 k)
    struct HostRange { const uint8_t* host; uint64_t bytes; const uint8_t* dev; };
    std::vector<HostRange> host_ranges;   // what wz_host_register page-locked, with the address the device sees it at
    int wide_cus = 128;        // CUs the wide head kernel's K slices are sized for when several lanes are in flight (WZ_WIDE_CUS)
    int num_cus = 256;         // compute units of the device (the wide head kernel sizes its K slices for one round over them)
    bool head_inline = false;  // WZ_HEAD_INLINE=1: ... or inside the head convolutions themselves, by each tile's last K slice.
                               // Bit-identical and one launch less, but measured SLOWER (profiles/r02l_*: heads 76 + 18 us against
                               // 33 + 7 + 10 us, 37.4 k against 41.5 k frames/s): the reduction of a tile then runs on ONE workgroup at
                               // the tail of the launch and its epilogue stores from the MFMA fragment layout (16 pixels x 4 columns per
                               // instruction) instead of row-contiguous as the reduce kernel does.  Off by default.

    uint8_t* d_weights = nullptr;
    half_t* d_zeros = nullptr;               // 4 KiB of zeros
    unsigned long long* d_mbdbg = nullptr;   // WZ_MB_DEBUG=1: [n_ops][16] phase timestamps of the fused-block kernels
    std::vector<int> mb_groups;
    float* d_anchors = nullptr;
    uint8_t* d_frames = nullptr;             // staging for host frames [max_batch][max_w*max_h*3]
    size_t frame_stride = 0;
    WzPostConsts pc;
    size_t post_scratch_bytes = 0;
    size_t ws_bytes = 0;       // split-K / channel-group workspace per lane: 32 MiB per frame of max_batch, at least 64 MiB (the heads'
                               // parked partial sums take ~10 MiB per frame, the other ops keep the lower half)

    // A lane is everything one in-flight batch needs: its own stream, activation buffers, head
    // outputs, post-processing scratch, descriptor/result blocks and captured graphs.  Lanes run
    // concurrently on the GPU (a batch-8 layer fills only a fraction of the 256 CUs), slot i of the
    // API is lane i.
    struct Lane {
        hipStream_t stream = nullptr;
        bool owns_stream = false;            // false: the stream belongs to lane (index mod n_streams)
        std::vector<void*> bufs;             // owned activation buffers
        std::vector<half_t*> tptr;           // tensor index -> device pointer
        float* d_box_enc = nullptr;
        float* d_logits = nullptr;
        float* d_ws = nullptr;
        int32_t* d_tickets = nullptr;        // tile counters of the in-launch head reductions (zero between launches)
        WzHeadFinish* d_fin = nullptr;       // what finishing a head output needs (static)
        int launch_failed = 0;               // op index + 1 of a block no kernel took at launch time (enqueue_network), else 0
        bool decode_fused = false;           // set by enqueue_network: the grouped head reduce decoded the boxes
        bool cands_listed = false;           // ... and listed the candidates of the NMS kernel's first band
        uint8_t* d_frames = nullptr;         // staging for host frames of this lane [max_batch][frame_stride] (lazy)
        std::map<int, int> graph_nodes;      // batch size -> nodes of the captured graph (kernels + the descriptor copy)
        WzPostBuffers post;
        void* d_post_scratch = nullptr;      // hist + count (memset per batch)
        WzFrameDesc* h_desc = nullptr;       // pinned
        WzFrameDesc* h_desc_dev = nullptr;   // ... and its address as the device sees it (nullptr: not mapped)
        WzFrameDesc* d_desc = nullptr;
        wz_detection_t* d_rows = nullptr;
        uint8_t* d_pass = nullptr;
        wz_detection_t* h_rows = nullptr;    // pinned, device-mapped: wz_k_rows writes it directly
        uint8_t* h_pass = nullptr;
        wz_detection_t* m_rows = nullptr;    // device addresses of h_rows / h_pass
        uint8_t* m_pass = nullptr;
        uint32_t* h_status = nullptr;        // pinned, device-mapped: one word per frame, non-zero = the frame's rows may be short (wz_k_nms)
        uint32_t* m_status = nullptr;
        std::vector<int32_t> bound_idx;      // frame-table entries of the batch in flight (empty: not a bound batch)
        hipEvent_t done = nullptr;
        int n = 0;
        bool rows = false;                       // the batch being enqueued runs the row-staged resize kernel (a frame of it is read in place)
        int key = 0;                             // graph key of the batch in flight: batch size | rows << 16
        std::map<int, hipGraphExec_t> graphs;    // key = batch size | (row-staged resize kernel) << 16
        std::map<int, hipGraph_t> graph_src;     // the captured graph itself, kept where one of its node handles is
        std::map<int, hipGraphNode_t> pre_nodes; // ... and the resize kernel's node in that graph (its arguments carry the frame
                                                 // descriptors and are rewritten before every replay), if it takes them by value
        // WZ_LANE_STAMPS builds (wz_common.h): the lane's launch blocks [WZ_STAMP_SLOTS][WZ_STAMP_WORDS], the resize kernel's block and the
        // descriptors behind it in ONE allocation; the page-locked pairs the host reads; launches stamped so far in the batch being enqueued
        unsigned char* d_stamp_raw = nullptr;
        unsigned long long* d_stamps = nullptr;
        unsigned long long* d_stamps_pre = nullptr;
        unsigned long long* h_stamps = nullptr;
        unsigned long long* m_stamps = nullptr;
        int stamp_next = 1;
        bool lone = false;                       // the batch being enqueued goes kernel by kernel because the other lanes are idle (run_batch)
        std::map<int, std::vector<WzLaunchNote>> launch_notes;   // graph key -> what was launched, in order (grid, block, LDS, kernel)
        WzDescPack pack;                         // storage behind that node's kernelParams
        const WzFrameDesc* pre_frames = nullptr;
        int pre_size = 0, pre_half_pixel = 0;
        half_t* pre_out = nullptr;
        WzFrameDesc* pre_keep = nullptr;
        void* pre_kp[6];
    };
    Lane lanes[WZ_SLOTS];
    int n_lanes = 4;   // default; WZ_LANES overrides (1..WZ_SLOTS)
    int n_streams = 4; // HIP streams the lanes are spread over (WZ_STREAMS); more than 4 run slower on this stack (HISTORY.md part B)

    // the worker's frame table (wz_bind_frames): one entry per Frame of every FrameBuffer, described once instead of once per batch
    struct BoundFrame {
        const uint8_t* host;   // the frame's pixels (host address)
        const uint8_t* dev;    // the same bytes as the device sees them when the frame lies in a wz_host_register range, else nullptr
        int32_t w, h, fmt, cam;
        uint64_t bytes;
        wz_detection_t* rows;  // Header.detections of that frame (host), written by wz_collect_bound
    };
    std::vector<BoundFrame> bound;

    std::vector<WzCamFilter> h_cams;
    WzCamFilter* d_cams = nullptr;
    std::vector<int32_t*> cam_sat;
    wz_detection_t* d_tmp_rows = nullptr;
    uint8_t* d_tmp_pass = nullptr;

    std::vector<std::string> stage_names;
};

static int input_tensor_index(wz_engine* e) { return e->ops[0].src; }

// lane stamps: the block of the next launch of the batch being enqueued (nullptr in every build but `make stamps`)
#if WZ_LANE_STAMPS
static thread_local std::vector<WzLaunchNote>* g_launch_sink = nullptr;
void wz_note_launch(const void* func, dim3 grid, dim3 block, size_t lds) {
    if (g_launch_sink) g_launch_sink->push_back({func, {grid.x, grid.y, grid.z}, {block.x, block.y, block.z}, (unsigned)lds});
}
static unsigned long long* next_stamp(wz_engine::Lane& L) {
    if (!L.d_stamps || L.stamp_next > WZ_STAMP_SLOTS) return nullptr;
    return L.d_stamps + (size_t)(L.stamp_next++ - 1) * WZ_STAMP_WORDS;
}
#define WZ_STAMP_ARG(a) ((a).dbg = next_stamp(L))
#define WZ_STAMP_ARG2(a) ((a).dbg2 = next_stamp(L))
#define WZ_STAMP_GROUP(g) ((g).stamp = next_stamp(L))
#else
#define WZ_STAMP_ARG(a) ((void)0)
#define WZ_STAMP_ARG2(a) ((void)0)
#define WZ_STAMP_GROUP(g) ((void)0)
#endif
static bool tensor_is_pair(wz_engine* e, int idx) { return (e->tensors[idx].flags & WZ_TENSOR_HP) != 0; }
static bool input_is_pair(wz_engine* e) { return tensor_is_pair(e, input_tensor_index(e)); }
// bytes of one frame of tensor idx (pair tensors hold two halves per value; the input tensor is fp16 in both engines)
static size_t tensor_frame_bytes(wz_engine* e, int idx) {
    const WzTensorDesc& t = e->tensors[idx];
    const size_t es = idx == input_tensor_index(e) ? 2 : e->hdr.precision / 8;
    return (size_t)t.h * t.w * t.c * es * (tensor_is_pair(e, idx) ? 2 : 1);
}
typedef wz_engine::Lane Lane;

// ------------------------------------------------------------------------------------------------
// pipeline
// ------------------------------------------------------------------------------------------------
static WzMbArgs mb_args(wz_engine* e, const Lane& L, const WzOpDesc& op) {
    const uint8_t* wbase = e->d_weights;
    WzMbArgs a;
    memset(&a, 0, sizeof(a));
    a.in = L.tptr.empty() ? nullptr : L.tptr[op.src];
    a.we = (op.cin0 > 0 || op.stem) ? (const half_t*)(wbase + op.we_off) : nullptr;
    a.be = (op.cin0 > 0 || op.stem) ? (const float*)(wbase + op.be_off) : nullptr;
    a.wd = (const half_t*)(wbase + op.wd_off);
    a.bd = (const float*)(wbase + op.bd_off);
    a.wp = (const half_t*)(wbase + op.w_off);
    a.bp = (const float*)(wbase + op.b_off);
    a.res = (op.res >= 0 && !L.tptr.empty()) ? L.tptr[op.res] : nullptr;
    a.out = L.tptr.empty() ? nullptr : L.tptr[op.dst];
    a.hin = op.hin; a.win = op.win; a.hout = op.hout; a.wout = op.wout;
    a.cin0 = op.cin0; a.kc0 = op.kc0; a.nmid_pad = op.nmid_pad;
    a.cmid = op.cmid; a.cmid_pad = op.cmid_pad; a.kc = op.kc;
    a.cout = op.cout; a.n_pad = op.n_pad;
    a.stride = op.stride; a.pad_t = op.pad_t; a.pad_l = op.pad_l;
    a.hp = (op.flags & WZ_OPF_HP) ? 1 : 0;
    a.hp_out = (op.flags & WZ_OPF_HP_OUT) ? 1 : (op.flags & WZ_OPF_DUP_OUT) ? 2 : 0;
    a.qenc = (op.flags & WZ_OPF_QENC) ? 1 : 0;
    a.out2 = (op.dst2 > 0 && !L.tptr.empty()) ? L.tptr[op.dst2 - 1] : nullptr;
    a.has_out2 = op.dst2 > 0 ? 1 : 0;
    if (a.hp) {
        a.we_lo = (const half_t*)(wbase + op.we_lo_off);
        a.wp_lo = (const half_t*)(wbase + op.w_lo_off);
    }
    a.stem = op.stem;
    if (op.stem) {
        const WzTensorDesc& in = e->tensors[op.src];
        a.sin_h = in.h; a.sin_w = in.w;
        a.spad_t = op.stem_pad >> 16; a.spad_l = op.stem_pad & 0xffff;
    }
    return a;
}

// with_post: the post-processing chain follows in the same stream (it consumes -- and resets -- the candidate list)
static void enqueue_network(wz_engine* e, Lane& L, int n, StageTimer* t, bool with_post = true) {
    hipStream_t s = L.stream;
    const bool f32 = e->hdr.precision == 32;
    // The heads write only into the box / logit buffers that the post kernels read at the very end, so their partial
    // sums can wait: each head's slab is parked at the top of the workspace and ONE launch reduces them all after the
    // last op (six launches fewer per batch).  Everything else uses the workspace below `ws_top`.
    size_t ws_top = e->ws_bytes;
    WzReduceGroup heads;
    heads.n = 0;
    heads.first[0] = 0;
    heads.decode = 0;
    int box_ops = 0, box_ops_grouped = 0;   // ops that produce box encodings / how many of them went into `heads`
    int head_ops = 0;                       // ops that write box encodings or class logits
    // ... and the small heads themselves (3x3 ... 1x1 maps: a handful of workgroups each) share one launch as well
    WzConvGroup small, big;
    small.n = big.n = 0;
    small.first[0] = big.first[0] = 0;
    big.tickets = L.d_tickets;
    small.tickets = L.d_tickets ? L.d_tickets + WZ_TICKETS / 2 : nullptr;
    big.ticket_off = small.ticket_off = 0;
    int heads_in_groups = 0;                // entries of `heads` whose convolution sits in `big` or `small`
    // The heads with a long K loop run on the wide tile kernel (k_conv_wide.hip), all in one launch; their K slices are chosen
    // together, for the whole launch (one round over the CUs, slices of equal length), before the first one is enqueued.
    WzConvGroup wide;
    wide.n = 0;
    wide.first[0] = 0;
    wide.tickets = nullptr;
    wide.ticket_off = 0;
    int wide_T = 0;   // K steps per slice (0: no head goes there)
    if (!f32 && e->conv_wide && e->use_splitk && e->defer_heads && !e->head_inline) {
        int tiles[WZ_CONV_GROUP_MAX], steps[WZ_CONV_GROUP_MAX], cnt = 0;
        long long tile_bytes[WZ_CONV_GROUP_MAX];
        for (uint32_t i = 0; i < e->hdr.n_ops && cnt < WZ_CONV_GROUP_MAX; ++i) {
            const WzOpDesc& op = e->ops[i];
            if (op.kind == WZ_OP_STEM || op.kind == WZ_OP_DW || op.kind == WZ_OP_MBCONV) continue;
            WzConvArgs a;
            memset(&a, 0, sizeof(a));
            a.M = n * op.hout * op.wout;
            a.cin = op.cin; a.cout = op.cout; a.n_pad = op.n_pad; a.ksize = op.ksize; a.kc = op.kc;
            a.out_mode = op.out_mode;
            a.kchunks = op.ksize * op.ksize * op.kc;
            a.zeros = e->d_zeros;
            if (!wz_conv_wide_applies(a) || a.M < e->wide_min_m) continue;
            wz_conv_wide_shape(a, &tiles[cnt], &steps[cnt]);
            tile_bytes[cnt] = 128ll * 4 * (((a.cout + 15) & ~15) / (((a.cout + 15) / 16 + 4 * wz_conv_wide_ntw() - 1) / (4 * wz_conv_wide_ntw())));
            ++cnt;
        }
        // (K slices sized for HALF the chip when other lanes are there to use the rest: at batch 8 T = 27 -> 41 steps per slice, 179 -> ~120
        // workgroups of this one-wave-per-SIMD kernel, a third fewer fp32 partial tiles: 49.8 k -> 50.5 k frames/s, p50 +8 us)
        // (the three-tile build puts two workgroups on a CU: twice the slots in the same part of the chip)
        if (cnt > 0) wide_T = e->wide_T > 0 ? e->wide_T : wz_choose_wide_T(tiles, steps, tile_bytes, cnt, (e->n_lanes > 1 ? e->wide_cus : e->num_cus) * (wz_conv_wide_ntw() == 3 ? 2 : 1));
    }
    int big_head[WZ_CONV_GROUP_MAX] = {0}, small_head[WZ_CONV_GROUP_MAX] = {0};   // ... and which entry
    for (uint32_t i = 0; i < e->hdr.n_ops; ++i) {
        const WzOpDesc& op = e->ops[i];
        const uint8_t* wbase = e->d_weights;
        if (f32 && op.kind == WZ_OP_STEM) {
            wz_launch_stem_f32(L.tptr[op.src], (const float*)(wbase + op.w_off), (const float*)(wbase + op.b_off),
                               (float*)L.tptr[op.dst], n, op.hin, op.win, op.hout, op.wout, op.pad_t, op.pad_l, s, input_is_pair(e));
        } else if (f32 && op.kind == WZ_OP_DW) {
            wz_launch_dw_f32((const float*)L.tptr[op.src], (const float*)(wbase + op.w_off), (const float*)(wbase + op.b_off),
                             (float*)L.tptr[op.dst], n, op.hin, op.win, op.cin, op.hout, op.wout, op.stride, op.pad_t,
                             op.pad_l, op.act, s);
        } else if (op.kind == WZ_OP_STEM) {
            wz_launch_stem(L.tptr[op.src], (const float*)(wbase + op.w_off), (const float*)(wbase + op.b_off),
                           L.tptr[op.dst], n, op.hin, op.win, op.hout, op.wout, op.pad_t, op.pad_l, s);
        } else if (op.kind == WZ_OP_DW) {
            wz_launch_dw(L.tptr[op.src], (const half_t*)(wbase + op.w_off), (const float*)(wbase + op.b_off),
                         L.tptr[op.dst], n, op.hin, op.win, op.cin, op.hout, op.wout, op.stride, op.pad_t,
                         op.pad_l, op.act, s);
        } else if (op.kind == WZ_OP_MBCONV) {
            WzMbArgs a = mb_args(e, L, op);
            a.M = n * op.hout * op.wout;
            a.lone = L.lone ? 1 : 0;
            a.ws = (e->use_splitk || a.hp) ? L.d_ws : nullptr;   // (split-K partials of the plain blocks; the two-launch split blocks' project fragments)
            a.ws_bytes = ws_top;
            a.dbg = e->d_mbdbg ? e->d_mbdbg + (size_t)i * 16 : nullptr;
            WZ_STAMP_ARG(a);
            if (a.hp && wz_mbconv_hp2_applies(a, n)) {
                // the 10x10 split blocks of the robust program: two GEMM-shaped launches (k_mbconv_hp2.hip); the stage timer books the second one
                // on the op's (otherwise empty) reduce slot
                WZ_STAMP_ARG2(a);
                int r2 = wz_launch_mbconv_hp2(a, n, s, false, 1);
                if (t) t->mark();
                if (r2 >= 0) r2 = wz_launch_mbconv_hp2(a, n, s, false, 2);
                if (r2 < 0) L.launch_failed = (int)i + 1;
                if (e->d_mbdbg) e->mb_groups[i] = 1;
                if (t) t->mark();
                continue;
            }
            int groups = a.hp ? wz_launch_mbconv_hp(a, n, s, false)   // split-operand blocks (the `-p 16` program's first 13)
                              : wz_launch_mbconv_wave(a, n, s, false);   // large maps: one wavefront per pixel tile
            if (a.hp && groups < 0) L.launch_failed = (int)i + 1;   // nothing was enqueued for a split-operand block: run_batch reports it
            if (groups == -2 && (e->use_splitk || a.has_out2)) groups = wz_launch_mbconv_cs(a, n, s, false);   // small maps: channels over waves
            if (a.has_out2 && groups < 0) L.launch_failed = (int)i + 1;   // only the chunk-split kernel stores a second output
            if (groups == -2) groups = wz_launch_mbconv(a, n, s, false);
            if (e->d_mbdbg) e->mb_groups[i] = groups;
            if (t) t->mark();
            if (groups > 1) {   // sum the channel groups' partials in a fixed order, + bias, + residual, -> fp16
                WzConvArgs r;
                memset(&r, 0, sizeof(r));
                r.bias = a.bp; r.res = a.res; r.out = a.out;
                r.M = a.M; r.hout = op.hout; r.wout = op.wout; r.cout = op.cout; r.n_pad = op.n_pad;
                r.act = WZ_ACT_NONE; r.out_mode = WZ_OUT_ACT; r.splitk = groups;
                WZ_STAMP_ARG(r);
                wz_launch_splitk_reduce(r, L.d_ws, s);
            }
        } else {
            // Two convolutions of the extras chain in one launch (k_extras_pair.hip): a 1x1 whose output only the 3x3 stride-2 convolution behind it
            // reads, on the 5x5 / 3x3 / 2x2 maps -- three launches and three boundaries less per batch
            if (!f32 && op.out_mode == WZ_OUT_ACT && op.ksize == 1 && i + 1 < e->hdr.n_ops) {
                const WzOpDesc& nx = e->ops[i + 1];
                bool only = nx.kind == WZ_OP_CONV && nx.src == op.dst && nx.out_mode == WZ_OUT_ACT && op.res < 0 && nx.res < 0;
                for (uint32_t j = 0; only && j < e->hdr.n_ops; ++j)
                    if (j != i + 1 && (e->ops[j].src == op.dst || e->ops[j].res == op.dst)) only = false;
                if (only) {
                    auto conv_args = [&](const WzOpDesc& o) {
                        WzConvArgs c;
                        memset(&c, 0, sizeof(c));
                        c.in = L.tptr[o.src];
                        c.w = (const half_t*)(wbase + o.w_off);
                        c.bias = (const float*)(wbase + o.b_off);
                        c.out = L.tptr[o.dst];
                        c.M = n * o.hout * o.wout;
                        c.hin = o.hin; c.win = o.win; c.cin = o.cin;
                        c.hout = o.hout; c.wout = o.wout; c.cout = o.cout; c.n_pad = o.n_pad;
                        c.ksize = o.ksize; c.stride = o.stride; c.pad_t = o.pad_t; c.pad_l = o.pad_l; c.kc = o.kc;
                        c.act = o.act; c.out_mode = o.out_mode;
                        c.kchunks = o.ksize * o.ksize * o.kc;
                        c.splitk = 1;
                        return c;
                    };
                    WzConvArgs pa = conv_args(op);
                    const WzConvArgs pb = conv_args(nx);
                    if (wz_extras_pair_applies(pa, pb)) {
                        WZ_STAMP_ARG(pa);
                        wz_launch_extras_pair(pa, pb, n, s);
                        if (t) { t->mark(); t->mark(); t->mark(); t->mark(); }   // (both ops' slots; the launch is booked on the first)
                        ++i;
                        continue;
                    }
                }
            }
            WzConvArgs a;
            memset(&a, 0, sizeof(a));
            a.in = L.tptr[op.src];
            a.w = (const half_t*)(wbase + op.w_off);
            a.bias = (const float*)(wbase + op.b_off);
            a.res = op.res >= 0 ? L.tptr[op.res] : nullptr;
            a.M = n * op.hout * op.wout;
            a.hin = op.hin; a.win = op.win; a.cin = op.cin;
            a.hout = op.hout; a.wout = op.wout; a.cout = op.cout; a.n_pad = op.n_pad;
            a.ksize = op.ksize; a.stride = op.stride; a.pad_t = op.pad_t; a.pad_l = op.pad_l; a.kc = op.kc;
            a.act = op.act; a.out_mode = op.out_mode;
            a.kchunks = op.ksize * op.ksize * op.kc;
            void* final_out;
            if (op.out_mode == WZ_OUT_ACT) {
                final_out = L.tptr[op.dst];
            } else if (op.out_mode == WZ_OUT_BOX) {
                final_out = L.d_box_enc;
                a.out_batch_stride = (int64_t)e->hdr.num_anchors * 4;
                a.out_off = (int64_t)op.anchor_off * 4;
            } else if (op.out_mode == WZ_OUT_CLS) {
                final_out = L.d_logits;
                a.out_batch_stride = (int64_t)e->hdr.num_anchors * e->hdr.num_classes;
                a.out_off = (int64_t)op.anchor_off * e->hdr.num_classes;
            } else {   // WZ_OUT_HEAD: box columns -> d_box_enc, class columns -> d_logits
                final_out = L.d_box_enc;
                a.out_batch_stride = (int64_t)e->hdr.num_anchors * 4;
                a.out_off = (int64_t)op.anchor_off * 4;
                a.out2 = L.d_logits;
                a.out2_batch_stride = (int64_t)e->hdr.num_anchors * e->hdr.num_classes;
                a.out2_off = (int64_t)op.anchor_off * e->hdr.num_classes;
                a.n_box = op.n_box;
            }
            a.zeros = e->d_zeros;
            a.dbg = e->d_mbdbg ? e->d_mbdbg + (size_t)i * 16 : nullptr;
            const bool makes_boxes = op.out_mode == WZ_OUT_HEAD || op.out_mode == WZ_OUT_BOX;
            if (makes_boxes) ++box_ops;
            if (op.out_mode != WZ_OUT_ACT) ++head_ops;
            if (f32) {   // a.kc / a.kchunks count 16-channel chunks here
                int sk = !e->use_splitk ? 1 : wz_conv_f32_use_rs(a) ? wz_choose_splitk_rs_f32(a.M, a.n_pad, a.kchunks) : wz_choose_splitk(a.M, a.n_pad, a.kchunks / 2);
                const size_t slab32 = (((size_t)sk * a.M * a.n_pad * 4) + 255) & ~(size_t)255;
                if (sk > 1 && e->defer_heads && op.out_mode != WZ_OUT_ACT && heads.n < WZ_REDUCE_GROUP_MAX &&
                    slab32 + (e->ws_bytes >> 1) <= ws_top) {
                    // the heads' outputs are fp32 in both engines and their epilogue is the same: the partial sums are
                    // parked and reduced (+ decoded, + candidates marked) by the same grouped launch as in the fp16 engine
                    ws_top -= slab32;
                    float* const park = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(L.d_ws) + ws_top);
                    a.splitk = sk;
                    a.ws = park;
                    a.out = final_out;
                    wz_launch_conv_f32(a, s, false);
                    if (t) { t->mark(); t->mark(); }
                    wz_reduce_group_add(heads, a, park);
                    if (makes_boxes) ++box_ops_grouped;
                    continue;
                }
                if (wz_conv_ws_f32_applies(a)) {   // the extras chain: K split across the waves of a workgroup, no reduce
                    a.splitk = 1;
                    a.out = final_out;
                    wz_launch_conv_ws_f32(a, s);
                    if (t) { t->mark(); t->mark(); }
                    continue;
                }
                while (sk > 1 && (size_t)sk * a.M * a.n_pad * 4 > ws_top) --sk;
                a.splitk = sk;
                a.ws = L.d_ws;
                a.out = final_out;
                wz_launch_conv_f32(a, s);
                if (t) { t->mark(); t->mark(); }
                continue;
            }
            int sk = 1;
            if (e->use_splitk)
                sk = wz_conv_use_lds(a) ? wz_choose_splitk_lds(a.M, a.n_pad, a.kchunks) : wz_choose_splitk(a.M, a.n_pad, a.kchunks);
            bool to_wide = wide_T > 0 && wide.n < WZ_CONV_GROUP_MAX && wz_conv_wide_applies(a) && a.M >= e->wide_min_m;
            const size_t m16 = ((size_t)a.M + 15) & ~(size_t)15;   // (the wide kernel's partials are whole 16-pixel fragments)
            if (to_wide) {   // its partial sums always go through the grouped reduce, also with a single K slice
                const int wsk = ((a.kchunks >> 1) + wide_T - 1) / wide_T;
                if (heads.n < WZ_REDUCE_GROUP_MAX && ((((size_t)wsk * m16 * a.n_pad * 4) + 255) & ~(size_t)255) + (e->ws_bytes >> 1) <= ws_top)
                    sk = wsk;
                else
                    to_wide = false;
            }
            a.frag_ws = (to_wide && e->wide_frag) ? 1 : 0;
            const size_t slab = (((size_t)sk * (to_wide ? m16 : (size_t)a.M) * a.n_pad * 4) + 255) & ~(size_t)255;
            if ((sk > 1 || to_wide) && e->defer_heads && op.out_mode != WZ_OUT_ACT && heads.n < WZ_REDUCE_GROUP_MAX &&
                slab + (e->ws_bytes >> 1) <= ws_top) {   // keep at least half of the workspace for the other ops
                ws_top -= slab;
                float* const park = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(L.d_ws) + ws_top);
                a.splitk = sk;
                a.out = park;
                if (to_wide) {
                    wz_conv_wide_group_add(wide, a);
                } else if (small.n < WZ_CONV_GROUP_MAX && wz_conv_groupable(a)) {
                    small_head[small.n] = heads.n;
                    wz_conv_group_add(small, a);   // launched with the other small heads after the last op
                    ++heads_in_groups;
                } else if (int added = wz_conv_rs_groupable(a) ? wz_conv_rs_group_add(big, a) : 0) {
                    for (int k = 0; k < added; ++k) big_head[big.n - 1 - k] = heads.n;   // the heads on the tile kernel: one launch, too
                    ++heads_in_groups;
                } else {
                    WZ_STAMP_ARG(a);
                    wz_launch_conv(a, s);
                }
                if (t) { t->mark(); t->mark(); }   // its own reduce slot stays empty
                a.out = final_out;
                wz_reduce_group_add(heads, a, park);
                if (makes_boxes) ++box_ops_grouped;
                continue;
            }
            a.frag_ws = 0;   // (only the wide kernel and the grouped reduce behind it know that order)
            if ((sk > 1 || a.M < 128) && wz_conv_ws_applies(a)) {   // the K split happens inside the workgroups: nothing to reduce
                // (on the 2x2 and 1x1 maps also where one wave per tile would walk all of K alone)
                a.splitk = 1;
                a.out = final_out;
                WZ_STAMP_ARG(a);
                wz_launch_conv_ws(a, s);
                if (t) { t->mark(); t->mark(); }
                continue;
            }
            while (sk > 1 && (size_t)sk * a.M * a.n_pad * 4 > ws_top) --sk;
            a.splitk = sk;
            if (sk > 1) {
                a.out = L.d_ws;
                WZ_STAMP_ARG(a);
                wz_launch_conv(a, s);
                if (t) t->mark();
                a.out = final_out;
                WZ_STAMP_ARG(a);
                wz_launch_splitk_reduce(a, L.d_ws, s);
            } else {
                a.out = final_out;
                WZ_STAMP_ARG(a);
                wz_launch_conv(a, s);
                if (t) t->mark();
            }
        }
        if (t) t->mark();
    }
    // every box encoding is finished by the grouped reduce: let it decode the boxes as well (one launch less)
    L.decode_fused = heads.n > 0 && box_ops > 0 && box_ops == box_ops_grouped && e->fuse_decode;
    // ... and, when the NMS kernel selects its own candidates, list the class logits that can reach its first band
    L.cands_listed = with_post && L.decode_fused && e->post_self && head_ops == heads.n && e->list_cands;
    // With WZ_HEAD_INLINE=1 (and all of the above) the reduction does not take a launch of its own: the head convolutions do
    // it themselves, tile by tile, in the workgroup (wave) that publishes a tile's last K slice
    const bool inline_heads = e->head_inline && !f32 && L.cands_listed && L.d_tickets && L.d_fin && heads.n > 0 &&
                              heads_in_groups == heads.n && big.ticket_off <= WZ_TICKETS / 2 && small.ticket_off <= WZ_TICKETS / 2;
    if (inline_heads) {
        for (int i = 0; i < big.n + small.n; ++i) {
            WzConvArgs& a = i < big.n ? big.a[i] : small.a[i - big.n];
            const WzConvArgs& fin = heads.a[i < big.n ? big_head[i] : small_head[i - big.n]];
            a.ws = reinterpret_cast<float*>(a.out);     // the slab of partial tiles ...
            a.out = fin.out;                            // ... and where the finished columns go
            a.inline_reduce = 1;
            a.fin_flags = 3;                            // decode + list
            a.fin = L.d_fin;
        }
    }
    wide.stamp = big.stamp = small.stamp = heads.stamp = nullptr;   // (set per launch in the stamps build only)
    if (wide.n > 0) { WZ_STAMP_GROUP(wide); wz_launch_conv_wide_group(wide, s); }
    if (big.n > 0) { WZ_STAMP_GROUP(big); wz_launch_conv_rs_group(big, s); }
    if (t) t->mark();
    if (small.n > 0) { WZ_STAMP_GROUP(small); wz_launch_conv_group(small, s); }
    if (t) t->mark();
    if (L.decode_fused) {
        heads.decode = 1;
        heads.n_frames = n;
        heads.pc = e->pc;
        heads.anchors = L.post.anchors;
        heads.boxes = L.post.boxes;
        heads.valid = L.post.valid;
        heads.hist = L.post.hist;
        heads.count = L.post.count;
        heads.band = L.post.band;
    }
    heads.list = L.cands_listed ? 1 : 0;
    if (L.cands_listed) {
        heads.hint_logit = L.post.hint_logit;
        heads.cbits = L.post.cbits;
        heads.cbits_words = (e->pc.num_anchors * e->pc.num_classes + 31) >> 5;
    }
    if (heads.n > 0 && !inline_heads) { WZ_STAMP_GROUP(heads); wz_launch_splitk_reduce_group(heads, s); }
    if (t) t->mark();
}

static void enqueue_post(wz_engine* e, Lane& L, bool rows, int n, StageTimer* t, bool decode_done = false, bool listed = false) {
    hipStream_t s = L.stream;
    if (!decode_done) wz_launch_decode(L.post, e->pc, n, s);   // (also clears hist / count / band)
    if (t) t->mark();
    // default: the NMS kernel selects its candidates itself, one workgroup per frame; WZ_POST_SELF=0 puts the two
    // 256-CU scans (histogram of the scores, compaction of the band above its threshold) back in front of it
    if (!e->post_self) wz_launch_hist(L.post, e->pc, n, s);
    if (t) t->mark();
    if (!e->post_self) wz_launch_compact(L.post, e->pc, n, s);
    if (t) t->mark();
    // with `rows` the NMS kernel also fills the Detection rows (straight into the lane's pinned, device-mapped host
    // block: no D2H copy node, no separate row kernel); the "post/rows" stage slot stays empty
#if WZ_LANE_STAMPS
    L.post.stamps_n = rows && L.m_stamps ? L.stamp_next : 0;   // launches stamped in front of the NMS kernel (launch 0 = the resize kernel)
    L.post.stamps_host = rows && L.post.stamps_n ? L.m_stamps : nullptr;
#endif
    if (rows)
        wz_launch_nms(L.post, e->pc, n, s, L.d_desc, e->d_cams, L.m_rows, L.m_pass, e->post_self, listed, L.m_status);
    else
        wz_launch_nms(L.post, e->pc, n, s, nullptr, nullptr, nullptr, nullptr, e->post_self, listed);
    if (t) t->mark();
    if (rows && t) t->mark();
}

// everything between "descriptors are in h_desc[slot]" and "rows are in h_rows[slot]"
// inner > 1 (profiling only): every kernel of the pre-processing and the network is enqueued `inner` times back to back
static void enqueue_batch(wz_engine* e, Lane& L, int n, StageTimer* t, int inner = 1) {
    hipStream_t s = L.stream;
    if (t) t->mark();   // "(empty)": two event records with nothing between them = the bracket's own cost
    // the descriptors: read by the resize kernel straight out of the lane's page-locked host block (and left in d_desc for the
    // kernels behind it), or -- WZ_DESC_COPY=1, and where the host block has no device address -- copied in front of it
    // (by value: the descriptors are the resize kernel's arguments -- no copy, no PCIe read; batches of more than WZ_DESC_PACK frames
    // and WZ_DESC_ARGS=0 take one of the older routes)
    const bool by_value = e->desc_by_value && n <= WZ_DESC_PACK;
    const bool zero_copy = !by_value && e->desc_zero_copy && L.h_desc_dev;
    if (!by_value && !zero_copy) (void)hipMemcpyAsync(L.d_desc, L.h_desc, sizeof(WzFrameDesc) * n, hipMemcpyHostToDevice, s);
    if (t) t->mark();
    wz_launch_repeat = inner;
    L.stamp_next = 1;   // (lane stamps: launch 0 is the resize kernel, whose block lies in front of the descriptors it leaves behind)
    wz_launch_preprocess(zero_copy ? L.h_desc_dev : L.d_desc, n, (int)e->hdr.input_size, L.tptr[input_tensor_index(e)], s,
                         input_is_pair(e), (zero_copy || by_value) ? L.d_desc : nullptr, e->hdr.resize_mode == 1,
                         by_value ? L.h_desc : nullptr, L.rows ? e->pre_rows_lds : 0);
    if (t) t->mark();
    enqueue_network(e, L, n, t);
    wz_launch_repeat = 1;
    enqueue_post(e, L, true, n, t, L.decode_fused, L.cands_listed);
}

static int run_batch(wz_engine* e, int slot, int n) {
    Lane& L = e->lanes[slot];
    auto failed = [&]() -> int {   // a split-operand block no kernel took (the load-time check prepared batch 1 only): nothing may run on stale data
        const int op = L.launch_failed - 1;
        L.launch_failed = 0;
        return wz_fail(WZ_EFORMAT, "no split-operand kernel took op %d (%s) at batch %d", op, e->ops[op].name, n);
    };
    const int key = n | (L.rows ? 1 << 16 : 0);
    L.key = key;
    bool graph = e->use_graph;
    if (graph && e->graph_adaptive) {
        // a LONE batch -- nothing in flight on any other lane -- is launched kernel by kernel: the GPU starts on the first kernel while the host issues the
        // other 29 and the batch is done ~7 us sooner; with another lane busy the captured graph is replayed (10 us of host time instead of 84: what
        // keeps one thread ahead of four lanes).  The same launches either way: the rows do not depend on it (tests/test_gpu_variants.py).
        graph = false;
        for (int li = 0; li < e->n_lanes && !graph; ++li)
            if (li != slot && e->lanes[li].done && hipEventQuery(e->lanes[li].done) == hipErrorNotReady) graph = true;
        (void)hipGetLastError();
    }
    // the captured graph of (lane, batch size, resize form): captured and instantiated the first time the key is seen -- also when that first batch itself
    // goes kernel by kernel (adaptive launch): the several ms of capture + instantiation then land on the first batch of a size, not on the first moment
    // another lane happens to be busy
    auto ensure_graph = [&]() -> int {
        if (L.graphs.count(key)) return WZ_OK;
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(L.stream, hipStreamCaptureModeThreadLocal));
#if WZ_LANE_STAMPS
        L.launch_notes[key].clear();
        g_launch_sink = &L.launch_notes[key];
#endif
        L.lone = false;   // (a captured graph is what runs while the other lanes are busy)
        enqueue_batch(e, L, n, nullptr);
#if WZ_LANE_STAMPS
        g_launch_sink = nullptr;
#endif
        HIPCHK(hipStreamEndCapture(L.stream, &g));
        if (L.launch_failed) {
            (void)hipGraphDestroy(g);
            return failed();
        }
        size_t nodes = 0;
        if (hipGraphGetNodes(g, nullptr, &nodes) == hipSuccess) L.graph_nodes[key] = (int)nodes;
        if (e->desc_by_value && n <= WZ_DESC_PACK && nodes > 0) {   // the resize kernel's node: its arguments are this batch's descriptors
            std::vector<hipGraphNode_t> all(nodes);
            const void* want = wz_preprocess_func(input_is_pair(e), L.rows);
            if (hipGraphGetNodes(g, all.data(), &nodes) == hipSuccess)
                for (size_t k = 0; k < nodes; ++k) {
                    hipGraphNodeType ty;
                    hipKernelNodeParams kp;
                    if (hipGraphNodeGetType(all[k], &ty) == hipSuccess && ty == hipGraphNodeTypeKernel &&
                        hipGraphKernelNodeGetParams(all[k], &kp) == hipSuccess && kp.func == want) {
                        L.pre_nodes[key] = all[k];
                        break;
                    }
                }
            if (!L.pre_nodes.count(key)) {
                (void)hipGraphDestroy(g);
                return wz_fail(WZ_EHIP, "the resize kernel's node was not found in the captured graph");
            }
        }
        hipGraphExec_t ge = nullptr;
        HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        if (L.pre_nodes.count(key)) L.graph_src[key] = g;   // (the node handle lives as long as the graph it belongs to)
        else (void)hipGraphDestroy(g);
        L.graphs.emplace(key, ge);
        return WZ_OK;
    };
    if (graph || (e->use_graph && !L.graphs.count(key))) {
        const int rc = ensure_graph();
        if (rc != WZ_OK) return rc;
    }
    if (graph) {
        auto it = L.graphs.find(key);
        auto pn = L.pre_nodes.find(key);
        if (pn != L.pre_nodes.end()) {   // this batch's descriptors into the node's arguments
            memset(&L.pack, 0, sizeof(L.pack));
            memcpy(L.pack.d, L.h_desc, sizeof(WzFrameDesc) * n);
            L.pre_frames = nullptr;
            L.pre_size = (int)e->hdr.input_size;
            L.pre_out = L.tptr[input_tensor_index(e)];
            L.pre_keep = L.d_desc;
            L.pre_half_pixel = wz_preprocess_flags(e->hdr.resize_mode == 1, L.rows ? e->pre_rows_lds : 0);
            L.pre_kp[0] = &L.pre_frames; L.pre_kp[1] = &L.pack; L.pre_kp[2] = &L.pre_size;
            L.pre_kp[3] = &L.pre_out; L.pre_kp[4] = &L.pre_keep; L.pre_kp[5] = &L.pre_half_pixel;
            hipKernelNodeParams kp;
            memset(&kp, 0, sizeof(kp));
            kp.func = const_cast<void*>(wz_preprocess_func(input_is_pair(e), L.rows));
            if (L.rows) {
                kp.gridDim = dim3((unsigned)L.pre_size, n);
                kp.blockDim = dim3(wz_preprocess_rows_threads());
                kp.sharedMemBytes = (unsigned)e->pre_rows_lds;
            } else {
                kp.gridDim = dim3(((unsigned)L.pre_size * L.pre_size + 255) / 256, n);
                kp.blockDim = dim3(256);
            }
            kp.kernelParams = L.pre_kp;
            HIPCHK(hipGraphExecKernelNodeSetParams(it->second, pn->second, &kp));
        }
        HIPCHK(hipGraphLaunch(it->second, L.stream));
    } else {
#if WZ_LANE_STAMPS
        if (!L.launch_notes.count(key)) {   // (what was launched: also without a captured graph -- WZ_GRAPH=0, one lane)
            L.launch_notes[key].clear();
            g_launch_sink = &L.launch_notes[key];
        }
#endif
        L.lone = true;    // kernel by kernel: the other lanes are idle (adaptive launch), or the process runs without graphs (latency schedule, WZ_GRAPH=0)
        enqueue_batch(e, L, n, nullptr);
        L.lone = false;
#if WZ_LANE_STAMPS
        g_launch_sink = nullptr;
#endif
        HIPCHK(hipGetLastError());
        if (L.launch_failed) {
            (void)hipStreamSynchronize(L.stream);
            return failed();
        }
    }
    HIPCHK(hipEventRecord(L.done, L.stream));
    L.n = n;
    return WZ_OK;
}

// ------------------------------------------------------------------------------------------------
// lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" int wz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- the process's schedule (wz_common.h: wz_latency_schedule; include/watsor_hip.h: wz_set_schedule) --------------------------
static std::atomic<int> g_schedule{-1};   // -1: not fixed yet; WZ_SCHEDULE_THROUGHPUT / WZ_SCHEDULE_LATENCY once anything asked
bool wz_latency_schedule() {
    int s = g_schedule.load();
    if (s < 0) {
        const char* env = getenv("WZ_SCHEDULE");
        int want = (env && (env[0] == 'l' || env[0] == 'L')) ? WZ_SCHEDULE_LATENCY : WZ_SCHEDULE_THROUGHPUT;
        int expect = -1;
        g_schedule.compare_exchange_strong(expect, want);
        s = g_schedule.load();
    }
    return s == WZ_SCHEDULE_LATENCY;
}
extern "C" int wz_set_schedule(int schedule) {
    if (schedule != WZ_SCHEDULE_THROUGHPUT && schedule != WZ_SCHEDULE_LATENCY) return wz_fail(WZ_EINVAL, "wz_set_schedule: unknown schedule %d", schedule);
    int expect = -1;
    if (g_schedule.compare_exchange_strong(expect, schedule) || expect == schedule) return WZ_OK;
    return wz_fail(WZ_EINVAL, "wz_set_schedule: the launch shapes of this process were already fixed for the %s schedule (set it before the first wz_create)",
                   expect == WZ_SCHEDULE_LATENCY ? "latency" : "throughput");
}
extern "C" int wz_get_schedule(void) { return wz_latency_schedule() ? WZ_SCHEDULE_LATENCY : WZ_SCHEDULE_THROUGHPUT; }

// "0000:c1:00.0" of a device: what /sys/bus/pci/devices/<id>/{numa_node,local_cpulist} are keyed by (watsor_amd/numa.py)
extern "C" int wz_device_pci_bus_id(int device, char* buf, int buflen) {
    if (!buf || buflen < 16) return wz_fail(WZ_EINVAL, "wz_device_pci_bus_id: buffer of at least 16 bytes expected");
    if (hipDeviceGetPCIBusId(buf, buflen, device) != hipSuccess) {
        (void)hipGetLastError();
        return wz_fail(WZ_ENODEV, "no HIP device %d", device);
    }
    return WZ_OK;
}

extern "C" int wz_device_name_of(int device, char* buf, int buflen) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return wz_fail(WZ_ENODEV, "no HIP device %d", device);
    snprintf(buf, buflen, "%s (%s)", p.name, p.gcnArchName);
    return WZ_OK;
}

extern "C" const char* wz_last_error(void) { return g_err; }

static int load_blob(wz_engine* e, const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return wz_fail(WZ_ENOENT, "engine file not found: %s", path);
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < (long)sizeof(WzBlobHeader)) {
        fclose(f);
        return wz_fail(WZ_EFORMAT, "engine file %s is truncated", path);
    }
    e->blob.resize(sz);
    size_t rd = fread(e->blob.data(), 1, sz, f);
    fclose(f);
    if (rd != (size_t)sz) return wz_fail(WZ_EFORMAT, "short read on %s", path);
    memcpy(&e->hdr, e->blob.data(), sizeof(WzBlobHeader));
    const WzBlobHeader& h = e->hdr;
    if (h.magic != WZ_MAGIC) return wz_fail(WZ_EFORMAT, "%s is not an mi355x engine (bad magic)", path);
    if (h.version == WZ_FORMAT_VERSION && (h.resize_mode > 1 || h.post_flags > 1))
        return wz_fail(WZ_EFORMAT, "%s: unknown resize mode %u / post-processing flags %u", path, h.resize_mode, h.post_flags);
    if (h.version != WZ_FORMAT_VERSION)
        return wz_fail(WZ_EFORMAT, "%s: engine format %u, runtime expects %u -- rebuild it with watsor_amd.engine",
                       path, h.version, WZ_FORMAT_VERSION);
    if (h.total_bytes != (uint64_t)sz || (h.precision != 16 && h.precision != 32) || h.max_total > WZ_MAX_DETECTIONS || h.max_total < 1 ||
        h.num_classes > 4096 || h.n_ops == 0 || h.weights_off + h.weights_bytes > (uint64_t)sz ||
        h.tensors_off + (uint64_t)h.n_tensors * sizeof(WzTensorDesc) > (uint64_t)sz ||
        h.ops_off + (uint64_t)h.n_ops * sizeof(WzOpDesc) > (uint64_t)sz ||
        h.anchors_off + (uint64_t)h.num_anchors * 16 > (uint64_t)sz)
        return wz_fail(WZ_EFORMAT, "%s: inconsistent engine header", path);
    e->tensors = reinterpret_cast<const WzTensorDesc*>(e->blob.data() + h.tensors_off);
    e->ops = reinterpret_cast<const WzOpDesc*>(e->blob.data() + h.ops_off);
    for (uint32_t i = 0; i < h.n_ops; ++i) {
        const WzOpDesc& op = e->ops[i];
        if (op.src < 0 || op.src >= (int)h.n_tensors || op.dst >= (int)h.n_tensors || op.res >= (int)h.n_tensors ||
            (op.out_mode == WZ_OUT_ACT && op.dst < 0) || op.w_off < 0 || (uint64_t)op.w_off >= h.weights_bytes ||
            (op.kind == WZ_OP_CONV && (op.n_pad % 32 != 0 || op.cin % 8 != 0 || op.n_pad < op.cout ||
                                       (op.out_mode == WZ_OUT_ACT && op.cout % 8 != 0) ||
                                       (op.ksize != 1 && op.ksize != 3) ||
                                       (op.out_mode == WZ_OUT_HEAD && (op.n_box % 4 != 0 || op.n_box <= 0 || op.n_box >= op.cout)))) ||
            (op.kind == WZ_OP_DW && op.cin % 8 != 0) ||
            (op.kind == WZ_OP_MBCONV &&
             (op.dst < 0 || op.cmid < 8 || op.cmid % 8 != 0 || op.cmid_pad != (op.cmid + 31) / 32 * 32 ||
              op.kc != op.cmid_pad / 32 || op.n_pad % 32 != 0 || op.n_pad < op.cout || op.cout % 8 != 0 ||
              op.wd_off < 0 || (uint64_t)op.wd_off >= h.weights_bytes || op.bd_off < 0 ||
              (uint64_t)op.bd_off >= h.weights_bytes || op.stride < 1 || op.stride > 2 ||
              (op.cin0 != 0 && (op.cin0 % 8 != 0 || op.kc0 != (op.cin0 + 31) / 32 || op.nmid_pad % 16 != 0 ||
                                op.nmid_pad < op.cmid || op.we_off < 0 || (uint64_t)op.we_off >= h.weights_bytes ||
                                op.be_off < 0 || (uint64_t)op.be_off >= h.weights_bytes)))))
            return wz_fail(WZ_EFORMAT, "%s: op %u (%s) is malformed", path, i, op.name);
        if (op.dst2 != 0) {
            const WzTensorDesc* t2 = (op.dst2 > 0 && op.dst2 <= (int64_t)h.n_tensors) ? &e->tensors[op.dst2 - 1] : nullptr;
            if (op.kind != WZ_OP_MBCONV || !t2 || op.cin0 <= 0 || t2->h != op.hin || t2->w != op.win ||
                t2->c != op.cmid || (t2->flags & WZ_TENSOR_HP) || op.dst2 - 1 == op.dst || op.dst2 - 1 == op.src)
                return wz_fail(WZ_EFORMAT, "%s: op %u (%s): malformed second output", path, i, op.name);
        }
        if (op.kind == WZ_OP_MBCONV && h.precision != 16)
            return wz_fail(WZ_EFORMAT, "%s: fused blocks exist for the fp16 engine only", path);
        if (h.precision == 32 && ((op.kind == WZ_OP_CONV && op.cin % 4 != 0) || (op.kind == WZ_OP_DW && op.cin % 4 != 0)))
            return wz_fail(WZ_EFORMAT, "%s: op %u (%s) is malformed", path, i, op.name);
        if (op.kind == WZ_OP_MBCONV && (op.flags & WZ_OPF_HP)) {
            if (!(e->tensors[op.src].flags & WZ_TENSOR_HP) || (op.res >= 0 && op.res != op.src) ||
                ((op.flags & WZ_OPF_HP_OUT) != 0) != ((e->tensors[op.dst].flags & WZ_TENSOR_HP) != 0) ||
                ((op.flags & WZ_OPF_DUP_OUT) && ((op.flags & WZ_OPF_HP_OUT) || e->tensors[op.dst].c != 2 * op.cout)) ||
                (!(op.flags & WZ_OPF_DUP_OUT) && e->tensors[op.dst].c != op.cout) ||
                op.we_lo_off <= 0 || (uint64_t)op.we_lo_off >= h.weights_bytes || op.w_lo_off <= 0 ||
                (uint64_t)op.w_lo_off >= h.weights_bytes || (op.stem && e->tensors[op.src].c != 4))
                return wz_fail(WZ_EFORMAT, "%s: op %u (%s): malformed split-operand block", path, i, op.name);
            wz_engine::Lane none;
            if (wz_launch_mbconv_hp(mb_args(e, none, op), 1, nullptr, true) != 0)
                return wz_fail(WZ_EFORMAT, "%s: op %u (%s): no split-operand kernel for this shape", path, i, op.name);
        } else if (op.kind == WZ_OP_STEM && h.precision == 32 && (e->tensors[op.src].flags & WZ_TENSOR_HP) &&
                   !(op.dst >= 0 && (e->tensors[op.dst].flags & WZ_TENSOR_HP))) {
            // the fp32 program's stem reads the network input as a hi + lo pair (wz_k_stem_f32)
        } else if ((e->tensors[op.src].flags & WZ_TENSOR_HP) || (op.dst >= 0 && (e->tensors[op.dst].flags & WZ_TENSOR_HP)) ||
                   (op.res >= 0 && (e->tensors[op.res].flags & WZ_TENSOR_HP))) {
            return wz_fail(WZ_EFORMAT, "%s: op %u (%s) touches a pair tensor but is not a split-operand block", path, i, op.name);
        } else if (op.kind == WZ_OP_MBCONV && op.stem) {
            if (e->tensors[op.src].c != 4 || op.cin0 != 32 || op.kc0 != 1 || (e->tensors[op.src].h + 1) / 2 != op.hin ||
                (e->tensors[op.src].w + 1) / 2 != op.win)
                return wz_fail(WZ_EFORMAT, "%s: op %u (%s): malformed stem fusion", path, i, op.name);
            wz_engine::Lane none;
            if (wz_launch_mbconv_wave(mb_args(e, none, op), 1, nullptr, true) < 0)
                return wz_fail(WZ_EFORMAT, "%s: op %u (%s): no stem-fused kernel for this shape", path, i, op.name);
        } else if (op.kind == WZ_OP_MBCONV) {
            wz_engine::Lane none;
            if (op.dst2 > 0 ? wz_launch_mbconv_cs(mb_args(e, none, op), 1, nullptr, true) != 0
                            : wz_launch_mbconv(mb_args(e, none, op), 1, nullptr, true) != 0)
                return wz_fail(WZ_EFORMAT, "%s: op %u (%s): no fused-block kernel for this shape", path, i, op.name);
        }
    }
    return WZ_OK;
}

extern "C" int wz_create(const char* engine_path, int device, int max_batch, int max_width, int max_height,
                         wz_engine_t** out) {
    if (!engine_path || !out || max_batch < 1 || max_batch > 4095 || max_width < 1 || max_height < 1)
        return wz_fail(WZ_EINVAL, "wz_create: bad argument");
    *out = nullptr;
    wz_engine* e = new wz_engine();
    int rc = load_blob(e, engine_path);
    if (rc != WZ_OK) {
        delete e;
        return rc;
    }
    int ndev = wz_device_count();
    if (device < 0 || device >= ndev) {
        delete e;
        return wz_fail(WZ_ENODEV, "HIP device %d not present (%d visible)", device, ndev);
    }
    e->device = device;
    e->max_batch = max_batch;
    e->max_w = max_width;
    e->max_h = max_height;
    const char* env;
    e->no_reuse = (env = wz_dev_getenv("WZ_NO_BUFFER_REUSE")) && atoi(env) != 0;
    // WZ_GRAPH=0|1 decides when set.  Unset: captured graphs under the throughput schedule WHILE ANOTHER LANE IS BUSY (a batch is handed over in 10.5 us: one
    // thread keeps four lanes fed; a batch that finds the other lanes idle goes kernel by kernel: run_batch), kernel-by-kernel launches under the LATENCY schedule -- the GPU starts on the first kernel while the host still issues the rest, and a lone
    // batch is done 7 us sooner (0.380 -> 0.372 ms at batch 8, 0.2915 -> 0.2865 at batch 1; 84 us of host time per batch instead of 10, and 2 - 4 % of the
    // saturated throughput: profiles/r05_submit_probe.txt)
    env = getenv("WZ_GRAPH");
    if (env && !env[0]) env = nullptr;   // (set but empty = not set)
    e->use_graph = env ? atoi(env) != 0 : !wz_latency_schedule();
    e->graph_adaptive = e->use_graph && !env;   // (throughput schedule, nothing said: graphs while the lanes are busy, kernel by kernel for a lone batch)
    e->use_splitk = !((env = wz_dev_getenv("WZ_SPLITK")) && atoi(env) == 0);
    e->wide_frag = !((env = wz_dev_getenv("WZ_WIDE_FRAG")) && atoi(env) == 0);
    e->defer_heads = !((env = wz_dev_getenv("WZ_DEFER_HEADS")) && atoi(env) == 0);
    e->fuse_decode = !((env = wz_dev_getenv("WZ_FUSE_DECODE")) && atoi(env) == 0);
    e->post_self = !((env = wz_dev_getenv("WZ_POST_SELF")) && atoi(env) == 0);
    e->list_cands = !((env = wz_dev_getenv("WZ_LIST_CANDS")) && atoi(env) == 0);
    e->head_inline = (env = wz_dev_getenv("WZ_HEAD_INLINE")) && atoi(env) != 0;
    e->conv_wide = !((env = wz_dev_getenv("WZ_CONV_WIDE")) && atoi(env) == 0);
    e->desc_zero_copy = !((env = wz_dev_getenv("WZ_DESC_COPY")) && atoi(env) != 0);
    e->desc_by_value = !((env = wz_dev_getenv("WZ_DESC_ARGS")) && atoi(env) == 0);
    e->pre_rows = (env = wz_dev_getenv("WZ_PRE_ROWS")) ? atoi(env) != 0 : e->pre_rows;
    // WZ_SCHEDULE=latency: every page-locked frame is read in place -- a lone batch is done sooner without the staging copy in front of it
    // (640x480, batch 8: 0.505 against 0.575 ms) although the waiting workgroups cost frames/s with four lanes in flight (29.7 k against 34 k)
    if (wz_latency_schedule()) e->host_read = 1;
    e->host_read = (env = wz_dev_getenv("WZ_HOST_READ")) ? atoi(env) : e->host_read;
    e->pre_rows_lds = (int)wz_preprocess_rows_lds(max_width);
    if (e->pre_rows_lds > 60 * 1024) e->pre_rows = false, e->host_read = 0;   // (frames wider than ~6 800 pixels need more LDS than a launch gets by default: the per-pixel form, staged)
    e->wide_T = (env = wz_dev_getenv("WZ_WIDE_T")) ? atoi(env) : 0;
    e->wide_min_m = (env = wz_dev_getenv("WZ_WIDE_MIN_M")) ? atoi(env) : 1;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            e->num_cus = cus;
        e->wide_cus = wz_latency_schedule() ? e->num_cus : e->num_cus / 2;
        if ((env = wz_dev_getenv("WZ_WIDE_CUS")) && atoi(env) > 0) e->wide_cus = atoi(env);
    }
    if ((env = getenv("WZ_LANES")) && atoi(env) >= 1 && atoi(env) <= WZ_SLOTS) e->n_lanes = atoi(env);
    if ((env = getenv("WZ_STREAMS")) && atoi(env) >= 1 && atoi(env) <= WZ_SLOTS) e->n_streams = atoi(env);
    if (e->n_streams > e->n_lanes) e->n_streams = e->n_lanes;

#define CK(expr)                                                                                        \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            wz_fail(WZ_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);       \
            wz_destroy(e);                                                                              \
            return WZ_EHIP;                                                                             \
        }                                                                                               \
    } while (0)

    CK(hipSetDevice(device));
    char nm[256];
    wz_device_name_of(device, nm, sizeof(nm));
    e->name = nm;
    wz_post_init();
    wz_conv_init();
    for (uint32_t i = 0; i < e->hdr.n_ops; ++i)   // kernel attributes of the fused-block kernels, on THIS device
        if (e->ops[i].kind == WZ_OP_MBCONV) {
            wz_engine::Lane none;
            if (e->ops[i].flags & WZ_OPF_HP) {
                (void)wz_launch_mbconv_hp(mb_args(e, none, e->ops[i]), max_batch, nullptr, true);
                continue;
            }
            if (!e->ops[i].stem) (void)wz_launch_mbconv(mb_args(e, none, e->ops[i]), max_batch, nullptr, true);
            (void)wz_launch_mbconv_wave(mb_args(e, none, e->ops[i]), max_batch, nullptr, true);
            (void)wz_launch_mbconv_cs(mb_args(e, none, e->ops[i]), max_batch, nullptr, true);
        }

    const WzBlobHeader& h = e->hdr;
    CK(hipMalloc((void**)&e->d_weights, h.weights_bytes));
    CK(hipMemcpy(e->d_weights, e->blob.data() + h.weights_off, h.weights_bytes, hipMemcpyHostToDevice));
    if ((env = wz_dev_getenv("WZ_MB_DEBUG")) && atoi(env) != 0) {
        CK(hipMalloc((void**)&e->d_mbdbg, (size_t)h.n_ops * 16 * 8));
        CK(hipMemset(e->d_mbdbg, 0, (size_t)h.n_ops * 16 * 8));
        e->mb_groups.assign(h.n_ops, 0);
    }
    CK(hipMalloc((void**)&e->d_zeros, 4096));
    CK(hipMemset(e->d_zeros, 0, 4096));
    CK(hipMalloc((void**)&e->d_anchors, (size_t)h.num_anchors * 16));
    CK(hipMemcpy(e->d_anchors, e->blob.data() + h.anchors_off, (size_t)h.num_anchors * 16, hipMemcpyHostToDevice));
    e->frame_stride = ((size_t)max_width * max_height * 3 + 255) & ~(size_t)255;
    e->ws_bytes = std::max<size_t>((size_t)64 << 20, (size_t)max_batch * ((size_t)32 << 20));
    CK(hipMalloc((void**)&e->d_frames, e->frame_stride * max_batch));

    e->pc.num_anchors = h.num_anchors;
    e->pc.num_classes = h.num_classes;
    e->pc.max_total = h.max_total;
    e->pc.max_per_class = h.max_per_class;
    e->pc.score_thr = h.score_threshold;
    e->pc.iou_thr = h.iou_threshold;
    e->pc.scale_y = h.scale_y; e->pc.scale_x = h.scale_x; e->pc.scale_h = h.scale_h; e->pc.scale_w = h.scale_w;
    e->pc.clip_after = (h.post_flags & WZ_POSTF_CLIP_AFTER) ? 1 : 0;
    e->pc._pad = 0;
    e->post_scratch_bytes = (size_t)max_batch * (WZ_HIST_BINS + 5 + ((h.num_anchors * h.num_classes + 31) >> 5)) * 4;   // hist, count, band[2], hint, hint_logit, cbits

    for (uint32_t i = 0; i < h.n_tensors; ++i)
        if (e->tensors[i].slot < 0 || e->tensors[i].slot >= (int)h.n_slots) {
            wz_fail(WZ_EFORMAT, "tensor %u has a bad slot", i);
            wz_destroy(e);
            return WZ_EFORMAT;
        }

    for (int li = 0; li < e->n_lanes; ++li) {
        Lane& L = e->lanes[li];
        // More lanes than streams (WZ_STREAMS, default 4 = what this stack runs side by side): lane i rides the stream of lane
        // i mod streams, with buffers, graph and host blocks of its own -- the stream always has the next batch queued behind the one
        // it is running, so the host's turnaround between a batch's end and the next submit is not idle time on that queue.
        if (li < e->n_streams) {
            CK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
            L.owns_stream = true;
        } else {
            L.stream = e->lanes[li % e->n_streams].stream;
        }
        // activation buffers: one per slot (tensors with disjoint lifetimes share), or one per tensor
        L.tptr.assign(h.n_tensors, nullptr);
        if (e->no_reuse) {
            for (uint32_t i = 0; i < h.n_tensors; ++i) {
                const WzTensorDesc& t = e->tensors[i];
                void* p = nullptr;
                (void)t;
                CK(hipMalloc(&p, (size_t)max_batch * tensor_frame_bytes(e, i) + 256));
                L.bufs.push_back(p);
                L.tptr[i] = (half_t*)p;
            }
        } else {
            std::vector<size_t> slot_bytes(h.n_slots, 0);
            for (uint32_t i = 0; i < h.n_tensors; ++i) {
                const WzTensorDesc& t = e->tensors[i];
                const size_t b = (size_t)max_batch * tensor_frame_bytes(e, i) + 256;
                if (b > slot_bytes[t.slot]) slot_bytes[t.slot] = b;
            }
            for (uint32_t sidx = 0; sidx < h.n_slots; ++sidx) {
                void* p = nullptr;
                CK(hipMalloc(&p, slot_bytes[sidx]));
                L.bufs.push_back(p);
            }
            for (uint32_t i = 0; i < h.n_tensors; ++i) L.tptr[i] = (half_t*)L.bufs[e->tensors[i].slot];
        }
        CK(hipMalloc((void**)&L.d_box_enc, (size_t)max_batch * h.num_anchors * 4 * 4));
        CK(hipMalloc((void**)&L.d_logits, (size_t)max_batch * h.num_anchors * h.num_classes * 4));
        CK(hipMalloc((void**)&L.d_ws, e->ws_bytes));
        CK(hipMalloc((void**)&L.d_tickets, WZ_TICKETS * 4));
        CK(hipMemset(L.d_tickets, 0, WZ_TICKETS * 4));
        CK(hipMalloc((void**)&L.d_fin, sizeof(WzHeadFinish)));
        CK(hipMalloc((void**)&L.d_frames, e->frame_stride * max_batch));
        WzPostBuffers& pb = L.post;
        pb.box_enc = L.d_box_enc;
        pb.logits = L.d_logits;
        pb.anchors = e->d_anchors;
        CK(hipMalloc((void**)&pb.boxes, (size_t)max_batch * h.num_anchors * 16));
        CK(hipMalloc((void**)&pb.valid, (size_t)max_batch * h.num_anchors));
        CK(hipMalloc(&L.d_post_scratch, e->post_scratch_bytes));
        pb.hist = (uint32_t*)L.d_post_scratch;
        pb.count = pb.hist + (size_t)max_batch * WZ_HIST_BINS;
        pb.band = pb.count + max_batch;
        pb.hint = pb.band + 2 * max_batch;
        {   // first band of the self-scanning NMS kernel: start at score 0.25 until the frame slot has a history
            const uint32_t bin0 = 0x3E800000u >> 20;
            std::vector<uint32_t> hint0((size_t)max_batch, bin0);
            CK(hipMemcpy(pb.hint, hint0.data(), hint0.size() * 4, hipMemcpyHostToDevice));
            pb.hint_logit = reinterpret_cast<float*>(pb.hint + max_batch);
            uint32_t sbits = bin0 << 20;
            float s0;
            memcpy(&s0, &sbits, 4);
            const float l0 = logf(s0 / (1.0f - s0));   // wz_logit_floor(bin0), k_post.hip
            std::vector<float> hl0((size_t)max_batch, l0 - 0.01f - 1e-3f * fabsf(l0));
            CK(hipMemcpy(pb.hint_logit, hl0.data(), hl0.size() * 4, hipMemcpyHostToDevice));
            pb.cbits = reinterpret_cast<uint32_t*>(pb.hint_logit + max_batch);
            CK(hipMemset(pb.cbits, 0, (size_t)max_batch * ((h.num_anchors * h.num_classes + 31) >> 5) * 4));
        }
        {
            WzHeadFinish hf;
            memset(&hf, 0, sizeof(hf));
            hf.hint_logit = pb.hint_logit;
            hf.cbits = pb.cbits;
            hf.cbits_words = (int32_t)((h.num_anchors * h.num_classes + 31) >> 5);
            hf.pc = e->pc;
            hf.anchors = pb.anchors;
            hf.boxes = pb.boxes;
            hf.valid = pb.valid;
            CK(hipMemcpy(L.d_fin, &hf, sizeof(hf), hipMemcpyHostToDevice));
        }
        CK(hipMalloc((void**)&pb.cand, (size_t)max_batch * WZ_CAND_CAP * sizeof(uint2)));
        CK(hipMalloc((void**)&pb.det_boxes, (size_t)max_batch * h.max_total * 16));
        CK(hipMalloc((void**)&pb.det_scores, (size_t)max_batch * h.max_total * 4));
        CK(hipMalloc((void**)&pb.det_classes, (size_t)max_batch * h.max_total * 4));
        CK(hipMalloc((void**)&pb.det_num, (size_t)max_batch * 4));
        CK(hipMalloc((void**)&pb.dbg, (size_t)max_batch * 16 * 8));
        CK(hipMemset(pb.dbg, 0, (size_t)max_batch * 16 * 8));
        CK(hipHostMalloc((void**)&L.h_desc, sizeof(WzFrameDesc) * max_batch, hipHostMallocDefault));
        if (hipHostGetDevicePointer((void**)&L.h_desc_dev, L.h_desc, 0) != hipSuccess) {
            (void)hipGetLastError();
            L.h_desc_dev = nullptr;
        }
#if WZ_LANE_STAMPS
        {   // [launch blocks][the resize kernel's block][descriptors]: every pair starts as (entry = max, exit = 0)
            const size_t blocks = (size_t)WZ_STAMP_SLOTS * WZ_STAMP_WORDS * 8;
            CK(hipMalloc((void**)&L.d_stamp_raw, blocks + WZ_STAMP_PRE_BYTES + sizeof(WzFrameDesc) * max_batch));
            L.d_stamps = reinterpret_cast<unsigned long long*>(L.d_stamp_raw);
            L.d_stamps_pre = reinterpret_cast<unsigned long long*>(L.d_stamp_raw + blocks);
            L.d_desc = reinterpret_cast<WzFrameDesc*>(L.d_stamp_raw + blocks + WZ_STAMP_PRE_BYTES);
            std::vector<unsigned long long> init((blocks + WZ_STAMP_PRE_BYTES) / 8, 0ull);
            for (size_t k = 0; k < init.size(); k += WZ_STAMP_WORDS)
                for (int j = 0; j < 64; ++j) init[k + WZ_STAMP_ENTRY + j] = ~0ull;
            CK(hipMemcpy(L.d_stamp_raw, init.data(), init.size() * 8, hipMemcpyHostToDevice));
            const size_t host_pairs = (size_t)(WZ_STAMP_SLOTS + 1 + max_batch) * 2 * 8;
            CK(hipHostMalloc((void**)&L.h_stamps, host_pairs, hipHostMallocMapped));
            memset(L.h_stamps, 0, host_pairs);
            CK(hipHostGetDevicePointer((void**)&L.m_stamps, L.h_stamps, 0));
        }
#else
        CK(hipMalloc((void**)&L.d_desc, sizeof(WzFrameDesc) * max_batch));
#endif
        pb.stamps = L.d_stamps;
        pb.stamps_pre = L.d_stamps_pre;
        pb.stamps_host = nullptr;
        pb.stamps_n = 0;
        pb._stamps_pad = 0;
        CK(hipMalloc((void**)&L.d_rows, sizeof(wz_detection_t) * WZ_MAX_DETECTIONS * max_batch));
        CK(hipMalloc((void**)&L.d_pass, (size_t)WZ_MAX_DETECTIONS * max_batch));
        CK(hipHostMalloc((void**)&L.h_rows, sizeof(wz_detection_t) * WZ_MAX_DETECTIONS * max_batch, hipHostMallocMapped));
        CK(hipHostMalloc((void**)&L.h_pass, (size_t)WZ_MAX_DETECTIONS * max_batch, hipHostMallocMapped));
        CK(hipHostGetDevicePointer((void**)&L.m_rows, L.h_rows, 0));
        CK(hipHostGetDevicePointer((void**)&L.m_pass, L.h_pass, 0));
        CK(hipHostMalloc((void**)&L.h_status, sizeof(uint32_t) * max_batch, hipHostMallocMapped));
        memset(L.h_status, 0, sizeof(uint32_t) * max_batch);
        CK(hipHostGetDevicePointer((void**)&L.m_status, L.h_status, 0));
        CK(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    }
    e->stream = e->lanes[0].stream;

    e->h_cams.resize(WZ_MAX_CAMS);
    memset(e->h_cams.data(), 0, sizeof(WzCamFilter) * WZ_MAX_CAMS);
    e->cam_sat.assign(WZ_MAX_CAMS, nullptr);
    CK(hipMalloc((void**)&e->d_cams, sizeof(WzCamFilter) * WZ_MAX_CAMS));
    CK(hipMemset(e->d_cams, 0, sizeof(WzCamFilter) * WZ_MAX_CAMS));
    CK(hipMalloc((void**)&e->d_tmp_rows, sizeof(wz_detection_t) * WZ_MAX_DETECTIONS));
    CK(hipMalloc((void**)&e->d_tmp_pass, WZ_MAX_DETECTIONS));

    e->stage_names.push_back("(empty)");
    e->stage_names.push_back("h2d_descriptors");
    e->stage_names.push_back("preprocess");
    for (uint32_t i = 0; i < h.n_ops; ++i) {
        e->stage_names.push_back(e->ops[i].name);
        if (e->ops[i].kind == WZ_OP_CONV || e->ops[i].kind == WZ_OP_MBCONV)
            e->stage_names.push_back(std::string(e->ops[i].name) + "#splitk_reduce");
    }
    e->stage_names.push_back("heads#big_convs");
    e->stage_names.push_back("heads#small_convs");
    e->stage_names.push_back("heads#splitk_reduce");
    e->stage_names.push_back("post/decode");
    e->stage_names.push_back("post/hist");
    e->stage_names.push_back("post/compact");
    e->stage_names.push_back("post/nms");
    e->stage_names.push_back("post/rows");
    CK(hipDeviceSynchronize());
#undef CK
    *out = e;
    return WZ_OK;
}

static int sync_all(wz_engine* e) {
    for (int li = 0; li < e->n_lanes; ++li)
        if (e->lanes[li].stream) HIPCHK(hipStreamSynchronize(e->lanes[li].stream));
    return WZ_OK;
}

extern "C" void wz_destroy(wz_engine_t* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)sync_all(e);
    for (int32_t* p : e->cam_sat)
        if (p) (void)hipFree(p);
    void* devp[] = {e->d_weights, e->d_zeros, e->d_mbdbg, e->d_anchors, e->d_frames, e->d_cams, e->d_tmp_rows, e->d_tmp_pass};
    for (void* p : devp)
        if (p) (void)hipFree(p);
    for (int li = 0; li < WZ_SLOTS; ++li) {
        Lane& L = e->lanes[li];
        for (auto& kv : L.graphs) (void)hipGraphExecDestroy(kv.second);
        for (auto& kv : L.graph_src) (void)hipGraphDestroy(kv.second);
        for (void* p : L.bufs) (void)hipFree(p);
        void* lp[] = {L.d_frames, L.d_box_enc, L.d_logits, L.d_ws, L.d_tickets, L.d_fin, L.post.boxes, L.post.valid, L.d_post_scratch, L.post.cand,
                      L.post.det_boxes, L.post.det_scores, L.post.det_classes, L.post.det_num, L.post.dbg,
                      L.d_stamp_raw ? (void*)L.d_stamp_raw : (void*)L.d_desc, L.d_rows, L.d_pass};
        for (void* p : lp)
            if (p) (void)hipFree(p);
        if (L.h_desc) (void)hipHostFree(L.h_desc);
        if (L.h_rows) (void)hipHostFree(L.h_rows);
        if (L.h_pass) (void)hipHostFree(L.h_pass);
        if (L.h_status) (void)hipHostFree(L.h_status);
        if (L.h_stamps) (void)hipHostFree(L.h_stamps);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.stream && L.owns_stream) (void)hipStreamDestroy(L.stream);
    }
    delete e;
}

extern "C" const char* wz_device_name(wz_engine_t* e) { return e ? e->name.c_str() : ""; }

// ------------------------------------------------------------------------------------------------
// the hot call
// ------------------------------------------------------------------------------------------------
extern "C" uint64_t wz_frame_bytes(int w, int h, int fmt) {
    if (w < 1 || h < 1) return 0;
    if (fmt == WZ_FMT_RGB24) return (uint64_t)w * h * 3;
    if ((fmt == WZ_FMT_NV12 || fmt == WZ_FMT_I420) && !(w & 1) && !(h & 1)) return (uint64_t)w * h * 3 / 2;
    return 0;
}

static int fill_desc(wz_engine* e, int slot, int n, const uint8_t* const* d_rgb, const int* w, const int* h,
                     const int* fmt, const int* cam) {
    if (slot < 0 || slot >= e->n_lanes) return wz_fail(WZ_EINVAL, "slot %d out of range [0,%d)", slot, e->n_lanes);
    if (n < 1 || n > e->max_batch) return wz_fail(WZ_ELIMIT, "batch %d exceeds max_batch %d", n, e->max_batch);
    const float size = (float)e->hdr.input_size;
    e->lanes[slot].bound_idx.clear();
    for (int i = 0; i < n; ++i) {
        if (w[i] < 1 || h[i] < 1 || !d_rgb[i]) return wz_fail(WZ_EINVAL, "frame %d: bad pointer or size", i);
        const int pf = fmt ? fmt[i] : WZ_FMT_RGB24;
        if (!wz_frame_bytes(w[i], h[i], pf))
            return wz_fail(WZ_EINVAL, "frame %d: pixel format %d at %dx%d (NV12 / I420 need even sides)", i, pf, w[i], h[i]);
        const int c = cam ? cam[i] : -1;
        if (c >= WZ_MAX_CAMS) return wz_fail(WZ_ELIMIT, "camera id %d >= %d", c, WZ_MAX_CAMS);
        if (c >= 0 && e->h_cams[c].enabled && (e->h_cams[c].width != w[i] || e->h_cams[c].height != h[i]))
            return wz_fail(WZ_EINVAL, "frame %d is %dx%d but camera %d filter was set for %dx%d", i, w[i], h[i], c,
                           e->h_cams[c].width, e->h_cams[c].height);
        WzFrameDesc& d = e->lanes[slot].h_desc[i];
        d.rgb = d_rgb[i];
        d.w = w[i];
        d.h = h[i];
        d.scale_x = (float)w[i] / size;   // CalculateResizeScale(in, out, align_corners=false)
        d.scale_y = (float)h[i] / size;
        d.cam = c < 0 ? -1 : c;
        d.fmt = pf;
    }
    return WZ_OK;
}

extern "C" int wz_submit_device_fmt(wz_engine_t* e, int slot, int n, const uint8_t* const* d_rgb, const int* w,
                                    const int* h, const int* fmt, const int* cam) {
    if (!e || !d_rgb || !w || !h) return wz_fail(WZ_EINVAL, "wz_submit_device: null argument");
    if (slot < 0 || slot >= e->n_lanes) return wz_fail(WZ_EINVAL, "slot %d out of range [0,%d)", slot, e->n_lanes);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipEventSynchronize(e->lanes[slot].done));   // the lane's previous batch has fully drained
    int rc = fill_desc(e, slot, n, d_rgb, w, h, fmt, cam);
    if (rc != WZ_OK) return rc;
    e->lanes[slot].rows = e->pre_rows;
    return run_batch(e, slot, n);
}
extern "C" int wz_submit_device(wz_engine_t* e, int slot, int n, const uint8_t* const* d_rgb, const int* w,
                                const int* h, const int* cam) {
    return wz_submit_device_fmt(e, slot, n, d_rgb, w, h, nullptr, cam);
}

extern "C" int wz_wait(wz_engine_t* e, int slot) {
    if (!e || slot < 0 || slot >= e->n_lanes) return wz_fail(WZ_EINVAL, "wz_wait: bad argument");
    HIPCHK(hipEventSynchronize(e->lanes[slot].done));
    return WZ_OK;
}

// after a lane's rows were handed over: did the NMS kernel flag one of its frames?  (clip-after-NMS engines only; k_post.hip: wz_k_nms)
static int lane_status(wz_engine* e, int slot) {
    const Lane& L = e->lanes[slot];
    for (int i = 0; i < L.n; ++i)
        if (L.h_status && L.h_status[i])
            return wz_fail(WZ_EINCOMPLETE, "frame %d of the batch on lane %d: more than %d selected boxes lie entirely outside the image, "
                                           "its rows may be incomplete", i, slot, WZ_NMS_KEEP_MAX - (int)e->hdr.max_total);
    return WZ_OK;
}

extern "C" const wz_detection_t* wz_slot_rows(wz_engine_t* e, int slot) {
    if (!e || slot < 0 || slot >= e->n_lanes) return nullptr;
    return e->lanes[slot].h_rows;
}

extern "C" int wz_collect(wz_engine_t* e, int slot, wz_detection_t* const* out, uint8_t* const* pass) {
    int rc = wz_wait(e, slot);
    if (rc != WZ_OK) return rc;
    const Lane& L = e->lanes[slot];
    for (int i = 0; i < L.n; ++i) {
        if (out && out[i])
            memcpy(out[i], L.h_rows + (size_t)i * WZ_MAX_DETECTIONS, sizeof(wz_detection_t) * WZ_MAX_DETECTIONS);
        if (pass && pass[i]) memcpy(pass[i], L.h_pass + (size_t)i * WZ_MAX_DETECTIONS, WZ_MAX_DETECTIONS);
    }
    return lane_status(e, slot);
}

extern "C" int wz_num_slots(wz_engine_t* e) { return e ? e->n_lanes : 0; }

extern "C" int wz_graph_nodes(wz_engine_t* e, int slot) {
    if (!e || slot < 0 || slot >= e->n_lanes) return 0;
    const Lane& L = e->lanes[slot];
    auto it = L.graph_nodes.find(L.key);
    return it == L.graph_nodes.end() ? 0 : it->second;
}

extern "C" int wz_sync(wz_engine_t* e) {
    if (!e) return wz_fail(WZ_EINVAL, "wz_sync: null engine");
    return sync_all(e);
}

extern "C" int wz_detect_batch(wz_engine_t* e, int n, const uint8_t* const* rgb, const int* w, const int* h,
                               const int* cam, wz_detection_t* const* out, uint8_t* const* pass, float* ms) {
    return wz_detect_batch_fmt(e, n, rgb, w, h, nullptr, cam, out, pass, ms);
}
extern "C" int wz_detect_batch_fmt(wz_engine_t* e, int n, const uint8_t* const* rgb, const int* w, const int* h, const int* fmt,
                                   const int* cam, wz_detection_t* const* out, uint8_t* const* pass, float* ms) {
    if (!e || !rgb || !w || !h) return wz_fail(WZ_EINVAL, "wz_detect_batch: null argument");
    if (n < 1 || n > e->max_batch) return wz_fail(WZ_ELIMIT, "batch %d exceeds max_batch %d", n, e->max_batch);
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipEventSynchronize(e->lanes[0].done));
    std::vector<const uint8_t*> dptr(n);
    for (int i = 0; i < n; ++i) {
        if (!rgb[i] || w[i] < 1 || h[i] < 1) return wz_fail(WZ_EINVAL, "frame %d: bad pointer or size", i);
        if (w[i] > e->max_w || h[i] > e->max_h || (size_t)w[i] * h[i] * 3 > e->frame_stride)
            return wz_fail(WZ_ELIMIT, "frame %d is %dx%d, engine was created for at most %dx%d", i, w[i], h[i],
                           e->max_w, e->max_h);
        const uint64_t bytes = wz_frame_bytes(w[i], h[i], fmt ? fmt[i] : WZ_FMT_RGB24);
        if (!bytes) return wz_fail(WZ_EINVAL, "frame %d: pixel format %d at %dx%d (NV12 / I420 need even sides)", i, fmt[i], w[i], h[i]);
        uint8_t* dst = e->d_frames + e->frame_stride * i;
        HIPCHK(hipMemcpyAsync(dst, rgb[i], bytes, hipMemcpyHostToDevice, e->lanes[0].stream));
        dptr[i] = dst;
    }
    int rc = wz_submit_device_fmt(e, 0, n, dptr.data(), w, h, fmt, cam);
    if (rc != WZ_OK) return rc;
    rc = wz_collect(e, 0, out, pass);
    if (rc != WZ_OK && rc != WZ_EINCOMPLETE) return rc;
    const float el = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms)
        for (int i = 0; i < n; ++i) ms[i] = el;
    return rc;   // WZ_OK, or WZ_EINCOMPLETE: the rows are written (and timed), one frame's may be short
}

// ------------------------------------------------------------------------------------------------
// host frames, asynchronously: H2D of batch k+1 (DMA engine) under the kernels of batch k
// ------------------------------------------------------------------------------------------------
// Frame i of the batch going onto lane L: one copy into the lane's staging area, on the lane's stream.  From page-locked memory
// (wz_host_register) that is one SDMA transfer at PCIe rate; pageable memory is staged by the runtime.  (A copy KERNEL reading
// the frames through their device-mapped addresses -- one graph node per batch instead of one call per frame -- reaches
// 55 GB/s on its own, tools/micro/h2d_streams.hip, but only 23 GB/s beside the other lanes' kernels against the copies'
// 29 GB/s; a fifth stream for the copies alone, so that whole batches arrive back to back: 24.8 k against 31.5 k frames/s at
// 640x480 -- this stack runs four streams side by side, HISTORY.md part B: profiles/r03_host_path_*.)
// Where the device sees a page-locked host frame, or nullptr: inside a range wz_host_register locked (WzFrameDesc::rgb may then point
// at the frame itself and the resize kernel reads its tap rows over PCIe, no staging copy -- `host_read`).
static const uint8_t* device_view(wz_engine* e, const uint8_t* host, uint64_t bytes) {
    for (const auto& r : e->host_ranges)
        if (host >= r.host && host + bytes <= r.host + r.bytes) return r.dev ? r.dev + (host - r.host) : nullptr;
    return nullptr;
}
// host_read policy for one frame: 1 = always in place, 2 = only when the vertical down-scale skips source rows (>= 2: at most half
// of the rows are tapped; below that nearly every row is, some twice, and one DMA of the whole frame moves fewer bytes)
static bool read_in_place(const wz_engine* e, int h) {
    return e->host_read == 1 || (e->host_read == 2 && h >= 2 * (int)e->hdr.input_size);
}

static int stage_frame(wz_engine* e, Lane& L, int i, const uint8_t* host, uint64_t bytes, const uint8_t** where) {
    uint8_t* dst = L.d_frames + e->frame_stride * i;
    HIPCHK(hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, L.stream));
    *where = dst;
    return WZ_OK;
}

extern "C" int wz_submit_host(wz_engine_t* e, int slot, int n, const uint8_t* const* rgb, const int* w, const int* h,
                              const int* cam) {
    return wz_submit_host_fmt(e, slot, n, rgb, w, h, nullptr, cam);
}
extern "C" int wz_submit_host_fmt(wz_engine_t* e, int slot, int n, const uint8_t* const* rgb, const int* w, const int* h,
                                  const int* fmt, const int* cam) {
    if (!e || !rgb || !w || !h) return wz_fail(WZ_EINVAL, "wz_submit_host: null argument");
    if (slot < 0 || slot >= e->n_lanes) return wz_fail(WZ_EINVAL, "slot %d out of range [0,%d)", slot, e->n_lanes);
    if (n < 1 || n > e->max_batch) return wz_fail(WZ_ELIMIT, "batch %d exceeds max_batch %d", n, e->max_batch);
    HIPCHK(hipSetDevice(e->device));
    Lane& L = e->lanes[slot];
    HIPCHK(hipEventSynchronize(L.done));   // the lane's previous batch (and its reads of the staging area) drained
    if (!L.d_frames) return wz_fail(WZ_EINVAL, "wz_submit_host: lane %d has no staging area", slot);
    bool in_place = false;
    std::vector<const uint8_t*> dptr(n);
    for (int i = 0; i < n; ++i) {
        if (!rgb[i] || w[i] < 1 || h[i] < 1) return wz_fail(WZ_EINVAL, "frame %d: bad pointer or size", i);
        if (w[i] > e->max_w || h[i] > e->max_h || (size_t)w[i] * h[i] * 3 > e->frame_stride)
            return wz_fail(WZ_ELIMIT, "frame %d is %dx%d, engine was created for at most %dx%d", i, w[i], h[i],
                           e->max_w, e->max_h);
        const uint64_t bytes = wz_frame_bytes(w[i], h[i], fmt ? fmt[i] : WZ_FMT_RGB24);
        if (!bytes) return wz_fail(WZ_EINVAL, "frame %d: pixel format %d at %dx%d (NV12 / I420 need even sides)", i, fmt[i], w[i], h[i]);
        // pageable source: the runtime stages it (slow, synchronous); registered / pinned source: one DMA -- or no copy at all
        const uint8_t* dv = read_in_place(e, h[i]) ? device_view(e, rgb[i], bytes) : nullptr;
        if (dv) {
            dptr[i] = dv;
            in_place = true;
            continue;
        }
        int rc = stage_frame(e, L, i, rgb[i], bytes, &dptr[i]);
        if (rc != WZ_OK) return rc;
    }
    int rc = fill_desc(e, slot, n, dptr.data(), w, h, fmt, cam);
    if (rc != WZ_OK) return rc;
    L.rows = e->pre_rows || in_place;
    return run_batch(e, slot, n);
}

// Page-lock a host range (e.g. a FrameBuffer arena, watsor/stream/share.py:35-41) so that frames inside it go to
// the GPU by DMA at PCIe rate instead of through the runtime's bounce buffer.
extern "C" int wz_host_register(wz_engine_t* e, void* ptr, uint64_t bytes) {
    if (!e || !ptr || !bytes) return wz_fail(WZ_EINVAL, "wz_host_register: bad argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterMapped));
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        dev = nullptr;                        // (not mapped on this platform: frames inside it are staged by DMA as before)
    }
    e->host_ranges.push_back({static_cast<const uint8_t*>(ptr), bytes, static_cast<const uint8_t*>(dev)});
    for (auto& f : e->bound)                  // (a frame table bound before its memory was page-locked)
        if (!f.dev) f.dev = device_view(e, f.host, f.bytes);
    return WZ_OK;
}
extern "C" int wz_host_unregister(wz_engine_t* e, void* ptr) {
    if (!e || !ptr) return wz_fail(WZ_EINVAL, "wz_host_unregister: bad argument");
    HIPCHK(hipSetDevice(e->device));
    (void)sync_all(e);   // a copy (or a resize kernel reading in place) in flight may still be reading the range
    for (size_t k = 0; k < e->host_ranges.size(); ++k)
        if (e->host_ranges[k].host == ptr) {
            e->host_ranges.erase(e->host_ranges.begin() + k);
            break;
        }
    for (auto& f : e->bound)
        if (f.dev && f.host >= static_cast<const uint8_t*>(ptr)) f.dev = device_view(e, f.host, f.bytes);   // (re-resolved: stale views must not survive)
    HIPCHK(hipHostUnregister(ptr));
    return WZ_OK;
}

// ------------------------------------------------------------------------------------------------
// the worker's frame table: every Frame of every FrameBuffer described once (watsor/stream/share.py:27-41,76-81 -- sizes,
// pixels and rows of a frame never change after the buffers are created), a batch is then a list of indices
// (`frame_buffers[payload.sender].frames[payload.frame_index]`, watsor/detection/detector.py:104)
// ------------------------------------------------------------------------------------------------
extern "C" int wz_bind_frames(wz_engine_t* e, int n, const uint8_t* const* pixels, const int* w, const int* h, const int* fmt,
                              const int* cam, wz_detection_t* const* rows) {
    if (!e || n < 0 || (n && (!pixels || !w || !h || !rows))) return wz_fail(WZ_EINVAL, "wz_bind_frames: bad argument");
    HIPCHK(hipSetDevice(e->device));
    int rc = sync_all(e);   // no batch in flight refers to the old table
    if (rc != WZ_OK) return rc;
    std::vector<wz_engine::BoundFrame> tbl((size_t)n);
    for (int i = 0; i < n; ++i) {
        const int pf = fmt ? fmt[i] : WZ_FMT_RGB24;
        const int c = cam ? cam[i] : -1;
        if (!pixels[i] || !rows[i] || w[i] < 1 || h[i] < 1) return wz_fail(WZ_EINVAL, "frame-table entry %d: bad pointer or size", i);
        const uint64_t bytes = wz_frame_bytes(w[i], h[i], pf);
        if (!bytes) return wz_fail(WZ_EINVAL, "frame-table entry %d: pixel format %d at %dx%d (NV12 / I420 need even sides)", i, pf, w[i], h[i]);
        if (w[i] > e->max_w || h[i] > e->max_h || (size_t)w[i] * h[i] * 3 > e->frame_stride)
            return wz_fail(WZ_ELIMIT, "frame-table entry %d is %dx%d, engine was created for at most %dx%d", i, w[i], h[i], e->max_w, e->max_h);
        if (c >= WZ_MAX_CAMS) return wz_fail(WZ_ELIMIT, "camera id %d >= %d", c, WZ_MAX_CAMS);
        tbl[i] = {pixels[i], device_view(e, pixels[i], bytes), w[i], h[i], pf, c < 0 ? -1 : c, bytes, rows[i]};
    }
    for (int li = 0; li < e->n_lanes; ++li) e->lanes[li].bound_idx.clear();
    e->bound.swap(tbl);
    return WZ_OK;
}

extern "C" int wz_submit_bound(wz_engine_t* e, int slot, int n, const int32_t* entries) {
    if (!e || !entries) return wz_fail(WZ_EINVAL, "wz_submit_bound: null argument");
    if (slot < 0 || slot >= e->n_lanes) return wz_fail(WZ_EINVAL, "slot %d out of range [0,%d)", slot, e->n_lanes);
    if (n < 1 || n > e->max_batch) return wz_fail(WZ_ELIMIT, "batch %d exceeds max_batch %d", n, e->max_batch);
    for (int i = 0; i < n; ++i)
        if (entries[i] < 0 || (size_t)entries[i] >= e->bound.size())
            return wz_fail(WZ_EINVAL, "frame-table entry %d out of range [0,%zu)", entries[i], e->bound.size());
    HIPCHK(hipSetDevice(e->device));
    Lane& L = e->lanes[slot];
    HIPCHK(hipEventSynchronize(L.done));   // the lane's previous batch (and its reads of the staging area) drained
    if (!L.d_frames) return wz_fail(WZ_EINVAL, "wz_submit_bound: lane %d has no staging area", slot);
    bool in_place = false;
    std::vector<const uint8_t*> dptr(n);
    std::vector<int> ws(n), hs(n), fmts(n), cams(n);
    for (int i = 0; i < n; ++i) {
        const wz_engine::BoundFrame& f = e->bound[entries[i]];
        // (the filter of the frame's camera may have been set for another size since the table was built: fill_desc checks)
        ws[i] = f.w; hs[i] = f.h; fmts[i] = f.fmt; cams[i] = f.cam;
        if (f.dev && read_in_place(e, f.h)) {
            dptr[i] = f.dev;
            in_place = true;
            continue;
        }
        int rc = stage_frame(e, L, i, f.host, f.bytes, &dptr[i]);
        if (rc != WZ_OK) return rc;
    }
    int rc = fill_desc(e, slot, n, dptr.data(), ws.data(), hs.data(), fmts.data(), cams.data());
    if (rc != WZ_OK) return rc;
    L.rows = e->pre_rows || in_place;
    rc = run_batch(e, slot, n);
    if (rc != WZ_OK) return rc;
    L.bound_idx.assign(entries, entries + n);
    return WZ_OK;
}

extern "C" int wz_collect_bound(wz_engine_t* e, int slot) {
    int rc = wz_wait(e, slot);
    if (rc != WZ_OK) return rc;
    Lane& L = e->lanes[slot];
    if (L.bound_idx.empty() || (int)L.bound_idx.size() != L.n) return wz_fail(WZ_EINVAL, "wz_collect_bound: lane %d holds no bound batch", slot);
    for (int i = 0; i < L.n; ++i)
        memcpy(e->bound[L.bound_idx[i]].rows, L.h_rows + (size_t)i * WZ_MAX_DETECTIONS, sizeof(wz_detection_t) * WZ_MAX_DETECTIONS);
    L.bound_idx.clear();
    return lane_status(e, slot);
}

// ------------------------------------------------------------------------------------------------
// filters
// ------------------------------------------------------------------------------------------------
extern "C" int wz_set_camera_filter(wz_engine_t* e, int cam, int width, int height, const double* conf_thr,
                                    const double* area_thr, int n_zones, const uint8_t* zone_allow,
                                    const uint8_t* zone_fill) {
    if (!e || !conf_thr || !area_thr || width < 1 || height < 1) return wz_fail(WZ_EINVAL, "wz_set_camera_filter: bad argument");
    if (cam < 0 || cam >= WZ_MAX_CAMS) return wz_fail(WZ_ELIMIT, "camera id %d out of range [0,%d)", cam, WZ_MAX_CAMS);
    if (n_zones < 0 || n_zones > WZ_MAX_ZONES_PER_CAM) return wz_fail(WZ_ELIMIT, "%d zones > %d", n_zones, WZ_MAX_ZONES_PER_CAM);
    if (n_zones > 0 && !zone_fill) return wz_fail(WZ_EINVAL, "zone_fill is null");
    HIPCHK(hipSetDevice(e->device));
    { int _rc = sync_all(e); if (_rc != WZ_OK) return _rc; }
    // the new state is built aside and committed only when all of it exists: a failure (e.g. no memory for the
    // summed-area tables of a 1080p multi-zone mask) leaves the camera's previous filter intact
    WzCamFilter c;
    memset(&c, 0, sizeof(c));
    c.enabled = 1 | ((n_zones == 0 && zone_fill) ? 4 : 0);   // bit 2: a mask is configured but has no zone -> every row fails it
    c.width = width;
    c.height = height;
    c.n_zones = n_zones;
    memcpy(c.conf_thr, conf_thr, sizeof(double) * WZ_NUM_LABELS);
    memcpy(c.area_thr, area_thr, sizeof(double) * WZ_NUM_LABELS);
    for (int l = 0; l < WZ_NUM_LABELS; ++l)
        for (int z = 0; z < n_zones; ++z) c.allow[l][z] = zone_allow ? (zone_allow[(size_t)l * n_zones + z] ? 1 : 0) : 1;
    int32_t* sat = nullptr;
    if (n_zones > 0) {
        const size_t plane = (size_t)(width + 1) * (height + 1);
        uint8_t* d_fill = nullptr;
        hipError_t err = hipMalloc((void**)&sat, plane * n_zones * 4);
        if (err == hipSuccess) err = hipMalloc((void**)&d_fill, (size_t)width * height * n_zones);
        if (err == hipSuccess) err = hipMemcpy(d_fill, zone_fill, (size_t)width * height * n_zones, hipMemcpyHostToDevice);
        if (err == hipSuccess) {
            wz_launch_sat(d_fill, sat, width, height, n_zones, e->stream);
            err = hipStreamSynchronize(e->stream);
        }
        if (d_fill) (void)hipFree(d_fill);
        if (err != hipSuccess) {
            if (sat) (void)hipFree(sat);
            return wz_fail(WZ_EHIP, "wz_set_camera_filter: %s (camera %d keeps its previous filter)", hipGetErrorString(err), cam);
        }
        c.sat = sat;
    }
    hipError_t err = hipMemcpy(e->d_cams + cam, &c, sizeof(WzCamFilter), hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        if (sat) (void)hipFree(sat);
        return wz_fail(WZ_EHIP, "wz_set_camera_filter: %s (camera %d keeps its previous filter)", hipGetErrorString(err), cam);
    }
    if (e->cam_sat[cam]) (void)hipFree(e->cam_sat[cam]);   // nothing in flight reads it: every lane was synchronised above
    e->cam_sat[cam] = sat;
    e->h_cams[cam] = c;
    return WZ_OK;
}

extern "C" int wz_set_camera_drop(wz_engine_t* e, int cam, int drop) {
    if (!e || cam < 0 || cam >= WZ_MAX_CAMS) return wz_fail(WZ_EINVAL, "wz_set_camera_drop: bad argument");
    if (!(e->h_cams[cam].enabled & 1)) return wz_fail(WZ_EINVAL, "camera %d has no filter (wz_set_camera_filter first)", cam);
    HIPCHK(hipSetDevice(e->device));
    { int _rc = sync_all(e); if (_rc != WZ_OK) return _rc; }
    e->h_cams[cam].enabled = (e->h_cams[cam].enabled & ~2) | (drop ? 2 : 0);
    HIPCHK(hipMemcpy(e->d_cams + cam, &e->h_cams[cam], sizeof(WzCamFilter), hipMemcpyHostToDevice));
    return WZ_OK;
}

extern "C" int wz_clear_camera_filter(wz_engine_t* e, int cam) {
    if (!e || cam < 0 || cam >= WZ_MAX_CAMS) return wz_fail(WZ_EINVAL, "wz_clear_camera_filter: bad argument");
    HIPCHK(hipSetDevice(e->device));
    { int _rc = sync_all(e); if (_rc != WZ_OK) return _rc; }
    memset(&e->h_cams[cam], 0, sizeof(WzCamFilter));
    HIPCHK(hipMemcpy(e->d_cams + cam, &e->h_cams[cam], sizeof(WzCamFilter), hipMemcpyHostToDevice));
    if (e->cam_sat[cam]) {
        (void)hipFree(e->cam_sat[cam]);
        e->cam_sat[cam] = nullptr;
    }
    return WZ_OK;
}

extern "C" int wz_filter_rows(wz_engine_t* e, int cam, wz_detection_t* rows, uint8_t* pass) {
    if (!e || !rows || !pass || cam < 0 || cam >= WZ_MAX_CAMS) return wz_fail(WZ_EINVAL, "wz_filter_rows: bad argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->d_tmp_rows, rows, sizeof(wz_detection_t) * WZ_MAX_DETECTIONS, hipMemcpyHostToDevice, e->stream));
    wz_launch_filter_rows(e->d_cams, cam, e->d_tmp_rows, e->d_tmp_pass, e->stream);
    HIPCHK(hipMemcpyAsync(rows, e->d_tmp_rows, sizeof(wz_detection_t) * WZ_MAX_DETECTIONS, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(pass, e->d_tmp_pass, WZ_MAX_DETECTIONS, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return WZ_OK;
}

// ------------------------------------------------------------------------------------------------
// introspection / profiling
// ------------------------------------------------------------------------------------------------
extern "C" int wz_input_size(wz_engine_t* e) { return e ? (int)e->hdr.input_size : 0; }
extern "C" int wz_precision(wz_engine_t* e) { return e ? (int)e->hdr.precision : 0; }
extern "C" int wz_num_anchors(wz_engine_t* e) { return e ? (int)e->hdr.num_anchors : 0; }
extern "C" int wz_num_classes(wz_engine_t* e) { return e ? (int)e->hdr.num_classes : 0; }
extern "C" int wz_hp_blocks(wz_engine_t* e) { return e ? (int)e->hdr.hp_blocks : 0; }

#ifdef WZ_DEV_BUILD   // ---- everything below this line exists in libwatsor_hip_dev.so only (include/watsor_hip.h, last section)
extern "C" int wz_num_tensors(wz_engine_t* e) { return e ? (int)e->hdr.n_tensors : 0; }
extern "C" int wz_num_ops(wz_engine_t* e) { return e ? (int)e->hdr.n_ops : 0; }

extern "C" int wz_tensor_info(wz_engine_t* e, int idx, char* name, int namelen, int* h, int* w, int* c) {
    if (!e || idx < 0 || idx >= (int)e->hdr.n_tensors) return wz_fail(WZ_EINVAL, "tensor index %d", idx);
    const WzTensorDesc& t = e->tensors[idx];
    if (name && namelen > 0) snprintf(name, namelen, "%s", t.name);
    if (h) *h = t.h;
    if (w) *w = t.w;
    if (c) *c = t.c;
    return WZ_OK;
}

extern "C" int wz_tensor_flags(wz_engine_t* e, int idx) {
    if (!e || idx < 0 || idx >= (int)e->hdr.n_tensors) return wz_fail(WZ_EINVAL, "tensor index %d", idx);
    return e->tensors[idx].flags;
}


extern "C" int wz_op_info(wz_engine_t* e, int idx, char* name, int namelen, int* dims) {
    if (!e || idx < 0 || idx >= (int)e->hdr.n_ops) return wz_fail(WZ_EINVAL, "op index %d", idx);
    const WzOpDesc& o = e->ops[idx];
    if (name && namelen > 0) snprintf(name, namelen, "%s", o.name);
    if (dims) {
        // WZ_OP_MBCONV: cin = block input channels, slot 11 = depthwise (expanded) channels
        // (a stem-fused block reports cin = 3, the image channels)
        const int v[12] = {o.kind, o.kind == WZ_OP_MBCONV ? (o.stem ? 3 : o.cin0 ? o.cin0 : o.cmid) : o.cin, o.cout, o.ksize, o.stride,
                           o.hin, o.win, o.hout, o.wout, o.n_pad, o.kc, o.kind == WZ_OP_MBCONV ? o.cmid : 0};
        memcpy(dims, v, sizeof(v));
    }
    return WZ_OK;
}

extern "C" int wz_debug_nms(wz_engine_t* e, int n, uint64_t* out) {
    if (!e || !out || n < 1 || n > e->max_batch) return wz_fail(WZ_EINVAL, "wz_debug_nms: bad argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, e->lanes[0].post.dbg, (size_t)n * 16 * 8, hipMemcpyDeviceToHost));
    return WZ_OK;
}

extern "C" int wz_debug_mbconv(wz_engine_t* e, uint64_t* out, int32_t* groups) {
    if (!e || !out || !e->d_mbdbg) return wz_fail(WZ_EINVAL, "wz_debug_mbconv: create the engine with WZ_MB_DEBUG=1");
    HIPCHK(hipSetDevice(e->device));
    { int _rc = sync_all(e); if (_rc != WZ_OK) return _rc; }
    HIPCHK(hipMemcpy(out, e->d_mbdbg, (size_t)e->hdr.n_ops * 16 * 8, hipMemcpyDeviceToHost));
    if (groups) memcpy(groups, e->mb_groups.data(), sizeof(int) * e->hdr.n_ops);
    return WZ_OK;
}

// ---- lane stamps (`make stamps`; tools/lane_overlap.py): entry / exit of every launch of the batch last run on `slot`, 100 MHz ticks
extern "C" int wz_debug_lane_stamps(wz_engine_t* e, int slot, uint64_t* out, int cap) {
#if WZ_LANE_STAMPS
    if (!e || !out || slot < 0 || slot >= e->n_lanes) return wz_fail(WZ_EINVAL, "wz_debug_lane_stamps: bad argument");
    const Lane& L = e->lanes[slot];
    const int k = L.post.stamps_n;          // launches in front of the NMS kernel; its own pairs (one per frame) follow and are folded here
    if (!L.h_stamps || k <= 0 || cap < k + 1) return wz_fail(WZ_EINVAL, "wz_debug_lane_stamps: no stamped batch on lane %d (or cap %d < %d)", slot, cap, k + 1);
    for (int i = 0; i < 2 * k; ++i) out[i] = L.h_stamps[i];
    uint64_t t0 = ~0ull, t1 = 0;
    for (int f = 0; f < L.n; ++f) {
        t0 = std::min<uint64_t>(t0, L.h_stamps[2 * (k + f)]);
        t1 = std::max<uint64_t>(t1, L.h_stamps[2 * (k + f) + 1]);
    }
    out[2 * k] = t0;
    out[2 * k + 1] = t1;
    return k + 1;
#else
    (void)e; (void)slot; (void)out; (void)cap;
    return wz_fail(WZ_EINVAL, "wz_debug_lane_stamps: this library was not built with -DWZ_LANE_STAMPS=1 (`make stamps`)");
#endif
}
// launch `idx` of that batch: kernel name, dims = {workgroups, threads per workgroup, LDS bytes (static + dynamic), workgroups a CU holds}
extern "C" int wz_debug_lane_launch(wz_engine_t* e, int slot, int idx, char* name, int namelen, int* dims) {
#if WZ_LANE_STAMPS
    if (!e || slot < 0 || slot >= e->n_lanes || !dims) return wz_fail(WZ_EINVAL, "wz_debug_lane_launch: bad argument");
    Lane& L = e->lanes[slot];
    auto it = L.launch_notes.find(L.key);
    if (it == L.launch_notes.end() || idx < 0 || idx >= (int)it->second.size()) return wz_fail(WZ_EINVAL, "wz_debug_lane_launch: no launch %d", idx);
    const WzLaunchNote& ln = it->second[idx];
    HIPCHK(hipSetDevice(e->device));
    const char* nm = hipKernelNameRefByPtr(ln.func, L.stream);
    if (name && namelen > 0) snprintf(name, namelen, "%s", nm ? nm : "?");
    hipFuncAttributes fa;
    memset(&fa, 0, sizeof(fa));
    (void)hipFuncGetAttributes(&fa, ln.func);
    const int threads = (int)(ln.block[0] * ln.block[1] * ln.block[2]);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ln.func, threads, ln.lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    dims[0] = (int)(ln.grid[0] * ln.grid[1] * ln.grid[2]);
    dims[1] = threads;
    dims[2] = (int)fa.sharedSizeBytes + (int)ln.lds;
    dims[3] = per_cu;
    dims[4] = fa.numRegs;
    return (int)it->second.size();
#else
    (void)e; (void)slot; (void)idx; (void)name; (void)namelen; (void)dims;
    return wz_fail(WZ_EINVAL, "wz_debug_lane_launch: this library was not built with -DWZ_LANE_STAMPS=1 (`make stamps`)");
#endif
}

extern "C" int wz_num_stages(wz_engine_t* e) { return e ? (int)e->stage_names.size() : 0; }
extern "C" int wz_stage_name(wz_engine_t* e, int stage, char* name, int namelen) {
    if (!e || stage < 0 || stage >= (int)e->stage_names.size()) return wz_fail(WZ_EINVAL, "stage index %d", stage);
    snprintf(name, namelen, "%s", e->stage_names[stage].c_str());
    return WZ_OK;
}

extern "C" int wz_profile_device(wz_engine_t* e, int n, const uint8_t* const* d_rgb, const int* w, const int* h,
                                 int reps, float* stage_ms) {
    return wz_profile_stages(e, n, d_rgb, w, h, reps, 1, stage_ms);
}

extern "C" int wz_profile_stages(wz_engine_t* e, int n, const uint8_t* const* d_rgb, const int* w, const int* h,
                                 int reps, int inner, float* stage_ms) {
    if (!e || !stage_ms || reps < 1 || inner < 1 || inner > 64) return wz_fail(WZ_EINVAL, "wz_profile_stages: bad argument");
    HIPCHK(hipSetDevice(e->device));
    { int _rc = sync_all(e); if (_rc != WZ_OK) return _rc; }
    int rc = fill_desc(e, 0, n, d_rgb, w, h, nullptr, nullptr);
    if (rc != WZ_OK) return rc;
    e->lanes[0].rows = e->pre_rows;
    const size_t ns = e->stage_names.size();
    std::vector<double> acc(ns, 0.0);
    StageTimer t;
    t.s = e->stream;
    for (int r = 0; r < reps + 1; ++r) {   // first repetition is an untimed warm-up
        t.used = 0;
        t.mark();
        enqueue_batch(e, e->lanes[0], n, &t, inner);
        HIPCHK(hipStreamSynchronize(e->stream));
        if (t.used != ns + 1) return wz_fail(WZ_EINVAL, "profile: %zu marks for %zu stages", t.used, ns);
        if (r == 0) continue;
        for (size_t i = 0; i < ns; ++i) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]));
            acc[i] += ms;
        }
    }
    for (size_t i = 0; i < ns; ++i) stage_ms[i] = (float)(acc[i] / reps);
    for (hipEvent_t ev : t.ev) (void)hipEventDestroy(ev);
    return WZ_OK;
}

#endif   // WZ_DEV_BUILD

// ------------------------------------------------------------------------------------------------
// device memory helpers
// ------------------------------------------------------------------------------------------------
extern "C" int wz_dev_alloc(wz_engine_t* e, uint64_t bytes, void** d_ptr) {
    if (!e || !d_ptr) return wz_fail(WZ_EINVAL, "wz_dev_alloc: null argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMalloc(d_ptr, bytes));
    return WZ_OK;
}
extern "C" int wz_dev_free(wz_engine_t* e, void* d_ptr) {
    if (!e) return wz_fail(WZ_EINVAL, "wz_dev_free: null engine");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipFree(d_ptr));
    return WZ_OK;
}
extern "C" int wz_dev_upload(wz_engine_t* e, void* d_dst, const void* h_src, uint64_t bytes) {
    if (!e) return wz_fail(WZ_EINVAL, "wz_dev_upload: null engine");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return WZ_OK;
}
extern "C" int wz_dev_download(wz_engine_t* e, void* h_dst, const void* d_src, uint64_t bytes) {
    if (!e) return wz_fail(WZ_EINVAL, "wz_dev_download: null engine");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return WZ_OK;
}

#ifdef WZ_DEV_BUILD
// ------------------------------------------------------------------------------------------------
// stage-level entry points (parity tests)
// ------------------------------------------------------------------------------------------------
extern "C" int wz_stage_preprocess(wz_engine_t* e, const uint8_t* rgb, int w, int h, uint16_t* out_half) {
    return wz_stage_preprocess_fmt(e, rgb, w, h, WZ_FMT_RGB24, out_half);
}
extern "C" int wz_stage_preprocess_fmt(wz_engine_t* e, const uint8_t* rgb, int w, int h, int fmt, uint16_t* out_half) {
    if (!e || !rgb || !out_half) return wz_fail(WZ_EINVAL, "wz_stage_preprocess: null argument");
    if (w > e->max_w || h > e->max_h || (size_t)w * h * 3 > e->frame_stride) return wz_fail(WZ_ELIMIT, "frame too large");
    const uint64_t bytes = wz_frame_bytes(w, h, fmt);
    if (!bytes) return wz_fail(WZ_EINVAL, "pixel format %d at %dx%d (NV12 / I420 need even sides)", fmt, w, h);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->d_frames, rgb, bytes, hipMemcpyHostToDevice));
    const uint8_t* p = e->d_frames;
    int rc = fill_desc(e, 0, 1, &p, &w, &h, &fmt, nullptr);
    if (rc != WZ_OK) return rc;
    HIPCHK(hipMemcpyAsync(e->lanes[0].d_desc, e->lanes[0].h_desc, sizeof(WzFrameDesc), hipMemcpyHostToDevice, e->stream));
    const int S = (int)e->hdr.input_size;
    wz_launch_preprocess(e->lanes[0].d_desc, 1, S, e->lanes[0].tptr[input_tensor_index(e)], e->stream, input_is_pair(e), nullptr,
                         e->hdr.resize_mode == 1, nullptr, e->pre_rows ? e->pre_rows_lds : 0);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out_half, e->lanes[0].tptr[input_tensor_index(e)], tensor_frame_bytes(e, input_tensor_index(e)),
                     hipMemcpyDeviceToHost));
    return WZ_OK;
}

extern "C" int wz_stage_forward(wz_engine_t* e, int n, const uint16_t* in_half, float* box_enc, float* logits) {
    if (!e || !in_half) return wz_fail(WZ_EINVAL, "wz_stage_forward: null argument");
    if (n < 1 || n > e->max_batch) return wz_fail(WZ_ELIMIT, "batch %d exceeds max_batch %d", n, e->max_batch);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const int S = (int)e->hdr.input_size;
    if (input_is_pair(e)) {   // the given halves are the "hi" parts, the "lo" parts are zero (the input is exactly these halves)
        std::vector<uint16_t> pair((size_t)n * S * S * 8, 0);
        for (size_t px = 0; px < (size_t)n * S * S; ++px) memcpy(&pair[px * 8], in_half + px * 4, 8);
        HIPCHK(hipMemcpy(e->lanes[0].tptr[input_tensor_index(e)], pair.data(), pair.size() * 2, hipMemcpyHostToDevice));
    } else {
        HIPCHK(hipMemcpy(e->lanes[0].tptr[input_tensor_index(e)], in_half, (size_t)n * S * S * 4 * 2, hipMemcpyHostToDevice));
    }
    enqueue_network(e, e->lanes[0], n, nullptr, false);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    if (box_enc) HIPCHK(hipMemcpy(box_enc, e->lanes[0].d_box_enc, (size_t)n * e->hdr.num_anchors * 16, hipMemcpyDeviceToHost));
    if (logits)
        HIPCHK(hipMemcpy(logits, e->lanes[0].d_logits, (size_t)n * e->hdr.num_anchors * e->hdr.num_classes * 4, hipMemcpyDeviceToHost));
    return WZ_OK;
}

extern "C" int wz_stage_read_tensor(wz_engine_t* e, int idx, int frame, uint16_t* out_half) {
    if (!e || !out_half || idx < 0 || idx >= (int)e->hdr.n_tensors || frame < 0 || frame >= e->max_batch)
        return wz_fail(WZ_EINVAL, "wz_stage_read_tensor: bad argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t per = tensor_frame_bytes(e, idx);   // a pair tensor: 2c halves per pixel, hi then lo
    HIPCHK(hipMemcpy(out_half, (const uint8_t*)e->lanes[0].tptr[idx] + per * frame, per, hipMemcpyDeviceToHost));
    return WZ_OK;
}

extern "C" int wz_stage_postprocess(wz_engine_t* e, int n, const float* box_enc, const float* logits, float* boxes,
                                    float* scores, int32_t* classes, int32_t* num) {
    if (!e || !box_enc || !logits) return wz_fail(WZ_EINVAL, "wz_stage_postprocess: null argument");
    if (n < 1 || n > e->max_batch) return wz_fail(WZ_ELIMIT, "batch %d exceeds max_batch %d", n, e->max_batch);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t A = e->hdr.num_anchors, C = e->hdr.num_classes, T = e->hdr.max_total;
    HIPCHK(hipMemcpy(e->lanes[0].d_box_enc, box_enc, n * A * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->lanes[0].d_logits, logits, n * A * C * 4, hipMemcpyHostToDevice));
    enqueue_post(e, e->lanes[0], false, n, nullptr);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    if (boxes) HIPCHK(hipMemcpy(boxes, e->lanes[0].post.det_boxes, n * T * 16, hipMemcpyDeviceToHost));
    if (scores) HIPCHK(hipMemcpy(scores, e->lanes[0].post.det_scores, n * T * 4, hipMemcpyDeviceToHost));
    if (classes) HIPCHK(hipMemcpy(classes, e->lanes[0].post.det_classes, n * T * 4, hipMemcpyDeviceToHost));
    if (num) HIPCHK(hipMemcpy(num, e->lanes[0].post.det_num, n * 4, hipMemcpyDeviceToHost));
    return WZ_OK;
}

extern "C" int wz_stage_rows(wz_engine_t* e, int w, int h, const float* boxes, const float* scores,
                             const int32_t* classes, wz_detection_t* rows) {
    if (!e || !boxes || !scores || !classes || !rows) return wz_fail(WZ_EINVAL, "wz_stage_rows: null argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t T = e->hdr.max_total;
    HIPCHK(hipMemcpy(e->lanes[0].post.det_boxes, boxes, T * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->lanes[0].post.det_scores, scores, T * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->lanes[0].post.det_classes, classes, T * 4, hipMemcpyHostToDevice));
    WzFrameDesc& d = e->lanes[0].h_desc[0];
    memset(&d, 0, sizeof(d));
    d.w = w;
    d.h = h;
    d.cam = -1;
    HIPCHK(hipMemcpyAsync(e->lanes[0].d_desc, e->lanes[0].h_desc, sizeof(WzFrameDesc), hipMemcpyHostToDevice, e->stream));
    wz_launch_rows(e->lanes[0].post, e->lanes[0].d_desc, e->d_cams, 1, e->pc.max_total, e->lanes[0].d_rows, e->lanes[0].d_pass, e->stream);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(rows, e->lanes[0].d_rows, sizeof(wz_detection_t) * WZ_MAX_DETECTIONS, hipMemcpyDeviceToHost));
    return WZ_OK;
}
#endif   // WZ_DEV_BUILD
