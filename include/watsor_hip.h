/* libwatsor_hip.so -- C ABI of the MI355X detection backend for Watsor.
 *
 * The reference has no FFI on this path: the boundary is a duck-typed Python plugin
 * (`watsor/detection/tensorflow_cpu.py:13,64-92`, `watsor/detection/tensorrt_gpu.py:20,55-91`)
 * driven by `ObjectDetector._next_frame` (`watsor/detection/detector.py:102-112`).  This header is
 * what sits *behind* the Python class `watsor_amd.detection.hip_gpu.HipObjectDetector`: plain
 * pointers and sizes, no torch / numpy types.  Every entry point names the reference code it
 * stands in for.  INTEGRATION.md shows the ctypes binding and the two-line change in
 * `watsor/detection/detector.py` that enables it.
 *
 * Conventions: all functions return 0 on success or a negative WZ_E* code; `wz_last_error()`
 * returns a human readable message for the calling thread's last failure.  One engine = one GPU
 * with up to WZ_SLOTS lanes (a lane = one in-flight batch: its own HIP stream, activation buffers
 * and captured graph; wz_num_slots() tells how many, 4 unless WZ_LANES says otherwise); calls on
 * one engine must be serialised by the caller (the reference worker is sequential per detector
 * instance, detector.py:84-112) -- the lanes overlap on the GPU, not on the host.  Host frames are packed RGB24, HWC,
 * C-contiguous (`Frame.get_numpy_image`, watsor/stream/share.py:68-73); they are read only and
 * not retained past the call.
 *
 * Environment (read when the first engine of a process is created): WZ_LANES=1..8 (batches in flight, default 4), WZ_STREAMS (HIP
 * streams they ride, default = lanes, at most 4 pay), WZ_GRAPH=0|1 (launch kernel by kernel / replay a captured graph; unset: under the throughput schedule a captured graph while another lane is busy and
 * kernel by kernel for a batch that finds the other lanes idle, under the latency schedule always kernel by kernel),
 * WZ_SCHEDULE=latency (launch shapes for the shortest lone batch instead of the most frames per second with every lane busy).
 */
#ifndef WATSOR_HIP_H
#define WATSOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WZ_MAX_DETECTIONS 100 /* Header.detections length, watsor/stream/share.py:31 */
#define WZ_MAX_ZONES 10       /* Detection.zones length,   watsor/stream/share.py:22 */
#define WZ_NUM_LABELS 91      /* len(COCO_CLASSES),        watsor/config/coco.py:14-106 */
#define WZ_MAX_CAMS 256       /* camera ids with a GPU filter (wz_set_camera_filter) per engine: 0 .. WZ_MAX_CAMS - 1 */

#define WZ_OK 0
#define WZ_EINVAL (-1)   /* bad argument */
#define WZ_ENOENT (-2)   /* engine file missing  -> Python raises FileNotFoundError (detector.py:97-98) */
#define WZ_EFORMAT (-3)  /* engine file corrupt / wrong version */
#define WZ_EHIP (-4)     /* HIP runtime error (message has the hipError string) */
#define WZ_ENODEV (-5)   /* no such GPU */
#define WZ_ELIMIT (-6)   /* batch / resolution / camera id beyond what wz_create() reserved */
#define WZ_EINCOMPLETE (-7) /* the call DID its work -- every row was written -- but a frame's rows may be short (clip-after-NMS engines: more
                             * selected boxes lay entirely outside the image than the walk keeps in reserve; wz_collect / wz_collect_bound /
                             * wz_detect_batch).  Every other code means the rows were NOT written.  -> Python raises RowsIncomplete */

/* Byte-for-byte `Detection` of watsor/stream/share.py:11-25 (72 bytes; label@0 zones@4
 * confidence@48 bounding_box@56).  Rows are written in place. */
typedef struct wz_detection {
    int32_t label;
    int32_t zones[WZ_MAX_ZONES];
    int32_t _pad;
    double confidence;
    int32_t x_min, y_min, x_max, y_max;
} wz_detection_t;

typedef struct wz_engine wz_engine_t;

/* ---- device enumeration: what `cuda_gpus()` does with pycuda (watsor/detection/devices.py:28-77) */
int wz_device_count(void);
int wz_device_name_of(int device, char* buf, int buflen);
/* PCI address of a device ("0000:c1:00.0"; buflen >= 16): the key of /sys/bus/pci/devices/<id>/numa_node and local_cpulist, by which
 * a detector process pins itself -- and with it the first-touch placement of the frame memory it page-locks -- to the GPU's NUMA node
 * (the reference starts one detector process per device, watsor/detection/detector.py:34-50, watsor/main.py:414-418). */
int wz_device_pci_bus_id(int device, char* buf, int buflen);

/* ---- the schedule of this process: which of two sets of launch shapes its engines use.  THROUGHPUT (default): most frames per second
 * with every lane busy.  LATENCY: the shortest lone batch -- the reference's normal load is one frame at a time per detector
 * (`_next_frame`, detector.py:102-112).  Process-wide; call before the first wz_create (the shapes are fixed when first asked for;
 * afterwards only the value already in force is accepted).  Unset: WZ_SCHEDULE in the environment decides.  Plugin option:
 * hip_options={"schedule": "latency" | "throughput" | "auto"} (watsor_amd/detection/hip_gpu.py). */
#define WZ_SCHEDULE_THROUGHPUT 0
#define WZ_SCHEDULE_LATENCY 1
int wz_set_schedule(int schedule);
int wz_get_schedule(void);

/* ---- lifecycle: TensorRTObjectDetector.__init__/__exit__ (tensorrt_gpu.py:20-53,62-63)
 * engine_path: file written by `python -m watsor_amd.engine` (the analogue of gpu.trt).
 * max_batch frames per call, each at most max_width x max_height. */
int wz_create(const char* engine_path, int device, int max_batch, int max_width, int max_height,
              wz_engine_t** out);
void wz_destroy(wz_engine_t* e);
const char* wz_device_name(wz_engine_t* e);          /* plugin property `device_name` */
const char* wz_last_error(void);

/* ---- the hot call: `detect(image_shape, image_np, detections) -> ms`
 * (tensorflow_cpu.py:74-92 / tensorrt_gpu.py:65-91) for n frames at once.
 *   rgb[i]   host pointer to frame i (h[i] x w[i] x 3 uint8)
 *   cam[i]   camera id whose filter (wz_set_camera_filter) is applied to frame i, or -1; cam may be NULL
 *   out[i]   100 Detection rows of frame i, ALL rewritten (rows past the detections carry label 1,
 *            confidence 0, box 0 exactly like the TF graph's zero padding + label offset)
 *   pass[i]  optional (may be NULL / pass may be NULL): 100 bytes, 1 where the row survives
 *            `label > 0 and Confidence and Area and Mask` (watsor/filter/track.py:26)
 *   ms[i]    wall time of the batch in milliseconds (every frame of a batch reports the batch time)
 */
int wz_detect_batch(wz_engine_t* e, int n, const uint8_t* const* rgb, const int* w, const int* h,
                    const int* cam, wz_detection_t* const* out, uint8_t* const* pass, float* ms);

/* Same, frames already resident in this GPU's HBM (device pointers).  Split into an asynchronous
 * submit (everything enqueued on the engine's stream, results land in pinned slot `slot`) and a
 * collect that waits for that slot -- so a caller can keep WZ_SLOTS batches in flight. */
#define WZ_SLOTS 8
int wz_submit_device(wz_engine_t* e, int slot, int n, const uint8_t* const* d_rgb, const int* w,
                     const int* h, const int* cam);
/* The same with HOST frame pointers (as wz_detect_batch takes them): each lane has its own staging area, the copies
 * ride the lane's stream, so the H2D of one batch overlaps the kernels of the batches on the other lanes. */
int wz_submit_host(wz_engine_t* e, int slot, int n, const uint8_t* const* rgb, const int* w,
                   const int* h, const int* cam);
/* ---- other pixel formats (the decoder side).  The reference's decoders write rawvideo RGB24 because its schema
 * says so (`watsor/config/schema.py:161`, `watsor/stream/ffmpeg.py:78-88` reads whatever the decoder writes into the
 * FrameBuffer); a decoder told `-pix_fmt nv12` / `yuv420p` writes half the bytes, and the colour conversion ffmpeg would have
 * done on the host happens in the resize kernel (8-bit BT.601 limited range, nearest chroma; csrc/k_preprocess.hip).
 *   fmt[i]  WZ_FMT_* of frame i; fmt == NULL: all RGB24.  NV12 / I420 frames are w*h*3/2 bytes and need even w and h.
 * The three calls are the ones above with that one argument more. */
#define WZ_FMT_RGB24 0
#define WZ_FMT_NV12 1   /* h x w luma, then h/2 x w/2 interleaved (U, V) */
#define WZ_FMT_I420 2   /* h x w luma, then the h/2 x w/2 U plane, then the V plane (ffmpeg's yuv420p) */
int wz_detect_batch_fmt(wz_engine_t* e, int n, const uint8_t* const* frames, const int* w, const int* h, const int* fmt,
                        const int* cam, wz_detection_t* const* out, uint8_t* const* pass, float* ms);
int wz_submit_device_fmt(wz_engine_t* e, int slot, int n, const uint8_t* const* d_frames, const int* w, const int* h,
                         const int* fmt, const int* cam);
int wz_submit_host_fmt(wz_engine_t* e, int slot, int n, const uint8_t* const* frames, const int* w, const int* h,
                       const int* fmt, const int* cam);
uint64_t wz_frame_bytes(int w, int h, int fmt);   /* bytes of one frame; 0 for a format / size the engine does not take */
/* Page-lock / release a host range that frames are handed over from (the reference's FrameBuffer arenas,
 * watsor/stream/share.py:35-41): copies out of it become DMA transfers at PCIe rate, and a frame whose height is at least twice the
 * network's input (the bilinear resize then skips rows) is not copied at all -- the resize kernel reads its tap rows IN PLACE through
 * the range's device mapping.  Either way a submitted frame must stay unchanged until its batch is collected. */
int wz_host_register(wz_engine_t* e, void* ptr, uint64_t bytes);
int wz_host_unregister(wz_engine_t* e, void* ptr);
int wz_collect(wz_engine_t* e, int slot, wz_detection_t* const* out, uint8_t* const* pass);
/* ---- the worker's frame table.  The reference worker resolves every payload to `frame_buffers[sender].frames[index]`, asks
 * the frame for a numpy view of its pixels and hands the frame header's `Detection[100]` array to the plugin
 * (watsor/detection/detector.py:102-109); pixels, size and rows of a Frame never move after the FrameBuffers are created
 * (watsor/stream/share.py:27-41,76-81).  So they are described ONCE (entry i = one Frame: pixels, w, h, WZ_FMT_*, camera id or
 * -1, its rows), a batch is then n table indices, and the rows land in the frames' own headers.  fmt / cam may be NULL (RGB24 /
 * no camera).  The pixels travel as wz_submit_host moves them (one copy per frame on the lane's stream: DMA from
 * wz_host_register()ed memory).  wz_bind_frames replaces the whole table (waits for the lanes first). */
int wz_bind_frames(wz_engine_t* e, int n, const uint8_t* const* pixels, const int* w, const int* h, const int* fmt,
                   const int* cam, wz_detection_t* const* rows);
int wz_submit_bound(wz_engine_t* e, int slot, int n, const int32_t* entries);
int wz_collect_bound(wz_engine_t* e, int slot);   /* waits for `slot`, writes its rows into the bound frames' rows */
/* Wait for `slot` without copying rows out (rows stay readable via wz_slot_rows). */
int wz_wait(wz_engine_t* e, int slot);
const wz_detection_t* wz_slot_rows(wz_engine_t* e, int slot); /* pinned host, [n][100] */
int wz_sync(wz_engine_t* e);
int wz_graph_nodes(wz_engine_t* e, int slot);   /* nodes of the hipGraph last replayed on `slot` (kernel launches; no copy node unless the lane's descriptor block has no device address); 0 without graphs */
int wz_num_slots(wz_engine_t* e);   /* lanes actually created (WZ_SLOTS unless WZ_LANES in the environment says fewer) */

/* ---- per-camera filters on the GPU: ConfidenceFilter / AreaFilter / MaskFilter
 * (watsor/filter/confidence.py:10-19, area.py:10-26, mask.py:17-59).
 *   conf_thr[l]  confidence/100 for label l, NaN when the label is not configured (-> reject)
 *   area_thr[l]  area/100 * (width*height) as the reference computes it, NaN when not configured
 *   n_zones      number of zones of the camera's mask (0 = no mask -> Mask filter absent)
 *   zone_allow   [WZ_NUM_LABELS][n_zones] bytes: 1 if label l may report zone z (mask.py:29-42);
 *                NULL = every label sees every zone
 *   zone_fill    [n_zones][height][width] bytes: 1 on the lattice points of zone z's polygon
 *                (8-connected alpha==255 component with holes filled, ordered as mask.py:78-88)
 * n_zones == 0 with zone_fill != NULL: a mask IS configured but holds no zone -- every row then fails the mask test,
 * as the reference's MaskFilter with an empty polygon list does (mask.py:44-59); n_zones == 0 with zone_fill == NULL:
 * no mask.  On failure the camera keeps the filter it had.
 */
int wz_set_camera_filter(wz_engine_t* e, int cam, int width, int height, const double* conf_thr,
                         const double* area_thr, int n_zones, const uint8_t* zone_allow,
                         const uint8_t* zone_fill);
int wz_clear_camera_filter(wz_engine_t* e, int cam);
/* Drop mode for a camera that has a filter: rows that fail `label > 0 and Confidence and Area and Mask`
 * (watsor/filter/track.py:26) are written as all-zero rows, so that whatever reads the frame's detections next
 * (the sieve's TrackFilter) rejects them by its own `label > 0` test and need not run the filters again. */
int wz_set_camera_drop(wz_engine_t* e, int cam, int drop);
/* Run only the filter stage on caller-provided rows (host), in place: zones are written into the
 * rows exactly where the reference's MaskFilter would have been invoked; pass[100] receives the verdict. */
int wz_filter_rows(wz_engine_t* e, int cam, wz_detection_t* rows, uint8_t* pass);

/* Host-side zone extraction from an alpha plane: MaskFilter.__init__ / find_contours / contours_key
 * (mask.py:17-27,78-88) without OpenCV.  Writes up to max_zones filled-zone bitmaps ([z][h][w] bytes)
 * ordered by the reference's centroid key and returns the zone count (or a negative error). */
int wz_zones_from_alpha(const uint8_t* alpha, int width, int height, int max_zones, uint8_t* zone_fill,
                        int32_t* centroid_xy /* [max_zones][2], may be NULL */);

/* ---- the sieve's tracker, host code, one instance per camera, no engine needed.
 * Replaces TrackFilter(filters, sensitivity, history) of watsor/filter/track.py:19-149 for rows whose per-detection
 * filters already ran on the GPU; row order, new-track order (CPython set iteration) and zone order of the combined
 * rows are the reference's; equally-near tracks are visited in index order (np.argsort's tie order is not defined). */
typedef struct wz_tracker wz_tracker_t;
int wz_tracker_create(int sensitivity, int history, wz_tracker_t** out);   /* track.py:19-23; defaults 5, 10 */
void wz_tracker_destroy(wz_tracker_t* t);
int wz_tracker_reset(wz_tracker_t* t);
int wz_tracker_count(wz_tracker_t* t);                                     /* live tracks, all labels */
/* TrackFilter.__call__ (track.py:25-110) on n rows; a row takes part iff label > 0 and (pass == NULL or pass[i]).
 * Writes min(*n_out, cap) combined rows (track.py:118-149) to out, *n_out = rows the reference would return,
 * *suspicious = its second return value (track.py:38). */
int wz_tracker_update(wz_tracker_t* t, const wz_detection_t* rows, int n, const uint8_t* pass, wz_detection_t* out,
                      int cap, int* n_out, int* suspicious);
/* DetectionSieve._incoming_frame (watsor/filter/sieve.py:21-33,44-56) with one TrackFilter, in place on the
 * frame header's rows: results first, the remaining rows zeroed. */
int wz_tracker_sieve(wz_tracker_t* t, wz_detection_t* rows, int n, const uint8_t* pass, int* suspicious);
/* ---- what the engine file holds (read once by HipEngine.__init__) */
int wz_input_size(wz_engine_t* e);
int wz_num_anchors(wz_engine_t* e);
int wz_num_classes(wz_engine_t* e);
int wz_precision(wz_engine_t* e);   /* 16: fp16 storage / fp16 MFMA; 32: fp32 storage / exact-fp32 MFMA (engine built with -p 32) */
/* leading inverted-residual blocks that run with split (hi + lo) matrix operands; 0 = plain fp16 program
 * (`python -m watsor_amd.engine --plain-fp16`), whose scores miss the 1e-3 tolerance */
int wz_hp_blocks(wz_engine_t* e);

/* ---- device memory helpers so callers need no other GPU runtime binding */
int wz_dev_alloc(wz_engine_t* e, uint64_t bytes, void** d_ptr);
int wz_dev_free(wz_engine_t* e, void* d_ptr);
int wz_dev_upload(wz_engine_t* e, void* d_dst, const void* h_src, uint64_t bytes);
int wz_dev_download(wz_engine_t* e, void* h_dst, const void* d_src, uint64_t bytes);

/* ======================================================================================================================
 * Development library only (`make dev` -> libwatsor_hip_dev.so, compiled with -DWZ_DEV_BUILD): stage-level entry points of the
 * parity tests, per-kernel profiling for bench.py's roofline, diagnostics, and -- inside the library -- the WZ_* tuning knobs and
 * the kernel variants that lost their A/B (HISTORY.md part B).  libwatsor_hip.so exports nothing below this line and reads
 * only WZ_LANES, WZ_STREAMS, WZ_GRAPH and WZ_SCHEDULE from the environment.
 * ====================================================================================================================== */
#ifdef WZ_DEV_BUILD
/* Test hooks for the CPython-set emulation: iteration order after adding keys[0..n) / of
 * `set(range(n)).difference(used)`; both return the number of values written to out. */
int wz_debug_pyset_order(const int32_t* keys, int n, int32_t* out);
int wz_debug_unused_order(int n, const uint8_t* used, int32_t* out);

/* ---- introspection used by bench.py (roofline) and the parity tests */
int wz_num_tensors(wz_engine_t* e);
int wz_tensor_info(wz_engine_t* e, int idx, char* name, int namelen, int* h, int* w, int* c);
/* bit 0: the tensor is stored as a hi + lo pair of halves per value (2c halves per pixel: c hi, then c lo) --
 * the tensors between the split-operand blocks of the `-p 16` program (csrc/k_mbconv_hp.hip) */
int wz_tensor_flags(wz_engine_t* e, int idx);
int wz_num_ops(wz_engine_t* e);
/* dims[12] = kind,cin,cout,ksize,stride,hin,win,hout,wout,n_pad,kc,splitk */
int wz_op_info(wz_engine_t* e, int idx, char* name, int namelen, int* dims);
int wz_num_stages(wz_engine_t* e);                              /* kernels per batch, pre + ops + post */
int wz_stage_name(wz_engine_t* e, int stage, char* name, int namelen);
/* Run the pipeline `reps` times on device frames with a hipEvent pair around every kernel launch
 * (events on the engine's own stream); stage_ms[stage] = mean milliseconds of that launch. */
int wz_profile_device(wz_engine_t* e, int n, const uint8_t* const* d_rgb, const int* w, const int* h,
                      int reps, float* stage_ms);
/* The same with every kernel of the pre-processing and the network enqueued `inner` times back to back inside its
 * bracket (they are pure functions of their inputs): stage_ms[stage] = mean milliseconds of the whole bracket, so
 * (stage_ms - empty bracket) / inner is a launch INCLUDING its in-stream boundary, with the cost of the event pair
 * amortised rather than estimated.  The post-processing stages run once. */
int wz_profile_stages(wz_engine_t* e, int n, const uint8_t* const* d_rgb, const int* w, const int* h,
                      int reps, int inner, float* stage_ms);

/* diagnostics: 16 words per frame written by the NMS kernel of lane 0 (phase timestamps at 100 MHz, counts) */
int wz_debug_nms(wz_engine_t* e, int n, uint64_t* out);
/* Engines created with WZ_MB_DEBUG=1 in the environment: phase timestamps (100 MHz) of the first and the last
 * workgroup of every fused inverted-residual block, out[n_ops][16]; groups[n_ops] = channel groups launched. */
int wz_debug_mbconv(wz_engine_t* e, uint64_t* out, int32_t* groups);

/* Lane stamps -- only in a library built with -DWZ_LANE_STAMPS=1 (`make stamps` -> libwatsor_hip_stamps.so; elsewhere both return
 * WZ_EINVAL): every kernel of a batch records when its first workgroup entered and its last one left (100 MHz ticks of the
 * device's constant clock).  After wz_wait(slot): out[2k], out[2k+1] = entry / exit of launch k of that batch (launch 0 = the resize
 * kernel, the last one = the NMS kernel); returns the number of launches.  wz_debug_lane_launch: kernel name and
 * dims[5] = {workgroups, threads per workgroup, LDS bytes, workgroups one CU holds, registers} of launch idx; returns the number of
 * launches.  What tools/lane_overlap.py turns into kernels-in-flight and CU-slot-time per step without a profiler in the way. */
int wz_debug_lane_stamps(wz_engine_t* e, int slot, uint64_t* out, int cap);
int wz_debug_lane_launch(wz_engine_t* e, int slot, int idx, char* name, int namelen, int* dims);

/* ---- stage-level entry points for the parity tests (host in, host out, synchronous) */
/* resize + normalise of one frame -> half[size*size*4] (x,y,z,0 per pixel); when the input tensor is a pair
 * (wz_tensor_flags) half[size*size*8]: (x,y,z,0) hi then (x,y,z,0) lo per pixel */
int wz_stage_preprocess(wz_engine_t* e, const uint8_t* rgb, int w, int h, uint16_t* out_half);
int wz_stage_preprocess_fmt(wz_engine_t* e, const uint8_t* frame, int w, int h, int fmt, uint16_t* out_half);
/* network only: half input [n][size][size][4] -> float box encodings [n][A][4], logits [n][A][C]
 * (a pair input tensor gets these halves as its hi parts and zeros as its lo parts) */
int wz_stage_forward(wz_engine_t* e, int n, const uint16_t* in_half, float* box_enc, float* logits);
/* read activation tensor `idx` of frame `frame` left behind by the last forward (half, NHWC; a pair tensor
 * comes back as stored, 2c halves per pixel).
 * Only meaningful when the engine was created with WZ_NO_BUFFER_REUSE=1 in the environment. */
int wz_stage_read_tensor(wz_engine_t* e, int idx, int frame, uint16_t* out_half);
/* decode + sigmoid + NMS + top-k on caller-provided head outputs:
 * boxes [n][100][4] (ymin,xmin,ymax,xmax), scores [n][100], classes [n][100] (1-based), num [n] */
int wz_stage_postprocess(wz_engine_t* e, int n, const float* box_enc, const float* logits, float* boxes,
                         float* scores, int32_t* classes, int32_t* num);
/* row fill (tensorflow_cpu.py:79-90) for caller-provided detections of one frame of size w x h */
int wz_stage_rows(wz_engine_t* e, int w, int h, const float* boxes, const float* scores,
                  const int32_t* classes, wz_detection_t* rows);

#endif /* WZ_DEV_BUILD */

#ifdef __cplusplus
}
#endif
#endif /* WATSOR_HIP_H */
