// Engine file format ("mi355x.bin") shared by the packer (watsor_amd/engine.py) and the runtime.
//
// The reference's GPU plugin loads an opaque serialized TensorRT engine (`watsor/engine.py:54-65`,
// `watsor/detection/tensorrt_gpu.py:28-34`); this is the MI355X analogue: a flat, little-endian,
// position-independent image holding the op program, the anchors, the post-processing constants and
// the weights already BatchNorm-folded, fp16-rounded and laid out for the HIP kernels.
#pragma once
#include <stdint.h>

#define WZ_MAGIC 0x35335A57u /* "WZ35" */
#define WZ_FORMAT_VERSION 11u   // 11: the float-form chunk buffer's scaling changed (k_hp_ops.h, round 6): depthwise taps of WZ_OPF_QENC blocks carry 6 * 2^60 / (2 - 2^-13)

enum WzOpKind { WZ_OP_STEM = 1, WZ_OP_DW = 2, WZ_OP_CONV = 3, WZ_OP_MBCONV = 4 };
enum WzOutMode { WZ_OUT_ACT = 0, WZ_OUT_BOX = 1, WZ_OUT_CLS = 2, WZ_OUT_HEAD = 3 };
enum WzAct { WZ_ACT_NONE = 0, WZ_ACT_RELU6 = 1 };
enum WzTensorFlags { WZ_TENSOR_HP = 1 };
enum WzOpFlags {
    WZ_OPF_HP = 1, WZ_OPF_HP_OUT = 2,   // split-operand block (k_mbconv_hp.hip) / its output tensor is a hi + lo pair
    WZ_OPF_QENC = 4,                    // ... whose chunk buffer holds the 16-bit float form of relu6(x) / 6 (the robust program; k_hp_ops.h:
                                        // depthwise weights carry 6 * 2^60 / (2 - 2^-13), the depthwise bias is the plain one)
    WZ_OPF_DUP_OUT = 8                  // ... whose output tensor has 2 * cout PLAIN channels holding the fp16 output twice: the 1x1 conv behind it
                                        // (Conv_1 of the robust program) has K = 2 * cout with the hi halves of its weights over the first copy and
                                        // the lo halves over the second, i.e. split WEIGHTS on the plain convolution kernels
};

#pragma pack(push, 1)
struct WzBlobHeader {  // 160 bytes
    uint32_t magic, version, precision, input_size;
    uint32_t num_classes;     // class-head columns per anchor, background included (91)
    uint32_t num_anchors;     // 1917
    uint32_t n_tensors, n_ops;
    uint32_t max_total, max_per_class;
    float score_threshold, iou_threshold;
    float scale_y, scale_x, scale_h, scale_w;   // box coder scale factors (10,10,5,5)
    uint64_t tensors_off, ops_off, anchors_off, weights_off, weights_bytes, total_bytes;
    uint32_t n_slots;         // activation buffer slots after liveness packing
    uint32_t hp_blocks;       // leading inverted-residual blocks on the split-operand kernel (0 = plain fp16 program)
    uint32_t resize_mode;     // 0: TF1 legacy ResizeBilinear (src = dst * scale); 1: half_pixel_centers ((dst + 0.5) * scale - 0.5)
    uint32_t post_flags;      // bit 0: per-class NMS on the unclipped boxes, clip afterwards (else: clip, drop zero-area, NMS)
    uint32_t reserved[8];
};
#define WZ_POSTF_CLIP_AFTER 1u

struct WzTensorDesc {  // 64 bytes
    int32_t h, w, c;          // per frame, NHWC; c is the stored channel count (input: 4)
    int32_t slot;             // buffer slot (tensors with disjoint lifetimes share one)
    int32_t flags;            // WZ_TENSOR_HP: a pixel holds c "hi" halves followed by c "lo" halves (value = hi + lo)
    char name[44];
};

struct WzOpDesc {  // 256 bytes
    int32_t kind, src, dst, res;            // tensor indices, res = -1 when absent
    int32_t cin, cout, ksize, stride;
    int32_t hin, win, hout, wout;
    int32_t pad_t, pad_l, act, out_mode;
    int32_t anchor_off, anchors_per_loc;    // head ops: first anchor of this feature map
    int32_t n_pad;                          // packed output columns (multiple of 32) for WZ_OP_CONV
    int32_t kc;                             // 32-channel K chunks per filter tap (ceil(cin/32))
    int64_t w_off, b_off;                   // byte offsets from weights_off
    int32_t n_box;                          // WZ_OUT_HEAD: leading columns that go to the box-encoding buffer
    // WZ_OP_MBCONV (one inverted-residual block = [1x1 expand ->] depthwise 3x3 -> 1x1 project [+ residual]):
    // cin/cout/n_pad/kc/w_off/b_off describe the PROJECT conv (cin = cmid); the fields below the rest.
    int32_t cmid;                           // depthwise channels (= expanded channels)
    int32_t cin0;                           // block input channels (expand K); 0 = no expand stage
    int32_t kc0;                            // 32-channel K chunks of the expand conv
    int32_t cmid_pad;                       // cmid rounded up to 32 (row length of the packed depthwise weights)
    int32_t nmid_pad;                       // packed output columns of the expand conv
    int32_t stem;                           // 1: the expand stage IS the stem conv (3x3 s2 on the 4-channel input tensor `src`,
                                            //    K = 27 padded to 32); hin/win are the stem's OUTPUT map
    int32_t stem_pad;                       // stem padding, pad_t << 16 | pad_l
    int64_t we_off, be_off;                 // expand weights (WZ_OP_CONV layout, 1 tap) / float bias[nmid_pad]
    int64_t wd_off, bd_off;                 // depthwise: half w[9][cmid_pad] / float bias[cmid_pad]
    // WZ_OPF_HP: we_off / w_off hold the "hi" fragments, these the "lo" ones (same layout); the expand weights and bias
    // carry a factor 1/6 and the depthwise weights are float w[9][cmid_pad] carrying 6/65535 (the expanded tensor lives
    // in LDS as unorm16 of relu6(x)/6)
    int64_t we_lo_off, w_lo_off;
    int64_t flags;                          // WzOpFlags
    int64_t dst2;                           // WZ_OP_MBCONV: 1 + index of a second output tensor, or 0 -- the block's EXPANDED tensor (hin x win x cmid,
                                            // plain fp16), stored by the block itself (block 13: the first SSD feature map)
    char name[64];
};
#pragma pack(pop)

// Weight layouts (all offsets 256-byte aligned):
//  WZ_OP_STEM : float  w[27][32]  (k = (ky*3+kx)*3 + c), float bias[32]
//  WZ_OP_DW   : half   w[9][C]    (tap-major, channel innermost), float bias[C]
//  WZ_OP_CONV : half   w[n_pad/16][taps][kc][64 lanes][8]  where lane l of N-tile t holds
//               W[k = chunk*32 + (l>>4)*8 + j][n = t*16 + (l&15)], zero beyond cin / cout
//               (exactly the A-operand fragment of v_mfma_f32_16x16x32_f16), float bias[n_pad]
//  WZ_OP_MBCONV: expand + project in the WZ_OP_CONV layout, depthwise as WZ_OP_DW with rows padded to cmid_pad
//               (WZ_OPF_HP: every GEMM weight twice, hi and lo halves; depthwise weights fp32)
