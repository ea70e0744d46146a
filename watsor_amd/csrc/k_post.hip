// SSD post-processing on gfx950: anchor-box decode, sigmoid, per-class greedy NMS, top-100, row fill,
// and the per-camera confidence / area / zone filters.
//
// Decode .. top-100 live inside the TF graph the reference executes (`tensorflow_cpu.py:94-121`,
// SURVEY.md D7 / Appendix B.3-B.5); the row fill is `tensorflow_cpu.py:79-90`; the filters are
// `watsor/filter/confidence.py:17-19`, `area.py:19-26`, `mask.py:44-59`.
//
// NMS formulation: TF runs greedy NMS class by class (<= max_per_class each), concatenates, sorts by
// score and keeps max_total.  A candidate's fate depends only on higher-scored boxes of its own class,
// so walking ALL (anchor, class) candidates once in global order (score desc, class asc, anchor asc),
// suppressing only against kept boxes of the same class and stopping at max_total kept, yields exactly
// the same rows.  The walk needs the head of that order only: a 1024-bin histogram of the score bits
// picks a threshold that admits >= WZ_CAND_TARGET candidates, they are compacted, bitonic-sorted in LDS
// and walked by one wavefront (lanes = already kept boxes, `__any` = "suppressed").  If that head is
// exhausted before max_total rows are kept (or the threshold bin overflows WZ_CAND_CAP) the kernel
// continues with an exact, slower "next best candidate below the bound" scan, so the result never
// depends on the tuning constants.
//
// All comparisons the result depends on (score order, IoU > thr, area > 0, truncation) use fp32/fp64
// operations rounded once each in the oracle's order: contraction is disabled for this file.
#pragma clang fp contract(off)
#include "wz_common.h"

__device__ __forceinline__ float wz_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------
// decode + clip (FasterRcnnBoxCoder._decode, clip_to_window [0,0,1,1], area > 0)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wz_k_decode(WzPostBuffers b, WzPostConsts k, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    // first kernel of the post-processing chain: it also clears the per-frame histogram, candidate count and band
    // record (num_anchors > WZ_HIST_BINS threads per frame exist) -- one launch less than a memset node
    if (i < n * WZ_HIST_BINS) b.hist[i] = 0u;
    if (i < n) b.count[i] = 0u;
    if (i < 2 * n) b.band[i] = 0u;
    if (i >= n * k.num_anchors) return;
    const int a = i % k.num_anchors;
    const float4_t e = *reinterpret_cast<const float4_t*>(b.box_enc + (size_t)i * 4);
    const float4_t an = *reinterpret_cast<const float4_t*>(b.anchors + (size_t)a * 4);  // yc, xc, h, w
    const float ty = e[0] / k.scale_y, tx = e[1] / k.scale_x, th = e[2] / k.scale_h, tw = e[3] / k.scale_w;
    const float w = expf(tw) * an[3];
    const float h = expf(th) * an[2];
    const float yc = ty * an[2] + an[0];
    const float xc = tx * an[3] + an[1];
    const float hh = h / 2.0f, hw = w / 2.0f;
    float ymin = yc - hh, xmin = xc - hw, ymax = yc + hh, xmax = xc + hw;
    if (k.clip_after) {   // the NMS takes the boxes as decoded; every anchor is a candidate (wz_k_nms clips what it keeps)
        *reinterpret_cast<float4_t*>(b.boxes + (size_t)i * 4) = (float4_t){ymin, xmin, ymax, xmax};
        b.valid[i] = 1;
        return;
    }
    ymin = fminf(fmaxf(ymin, 0.0f), 1.0f);
    xmin = fminf(fmaxf(xmin, 0.0f), 1.0f);
    ymax = fminf(fmaxf(ymax, 0.0f), 1.0f);
    xmax = fminf(fmaxf(xmax, 0.0f), 1.0f);
    const float area = (ymax - ymin) * (xmax - xmin);
    *reinterpret_cast<float4_t*>(b.boxes + (size_t)i * 4) = (float4_t){ymin, xmin, ymax, xmax};
    b.valid[i] = area > 0.0f ? 1 : 0;
}

// candidate j of a frame = entry j of logits[f] (anchor-major, class innermost, column 0 = background)
__device__ __forceinline__ bool wz_candidate(const WzPostBuffers& b, const WzPostConsts& k, int f, int j,
                                             uint32_t& key, uint32_t& tie) {
    const int a = j / k.num_classes, col = j - a * k.num_classes;
    if (col == 0) return false;
    if (!b.valid[(size_t)f * k.num_anchors + a]) return false;
    const float s = wz_sigmoid(b.logits[(size_t)f * k.num_anchors * k.num_classes + j]);
    if (!(s > k.score_thr)) return false;
    key = __float_as_uint(s);
    tie = (uint32_t)(col - 1) * (uint32_t)k.num_anchors + (uint32_t)a;   // class asc, then anchor asc
    return true;
}

// Entries per thread of the two scans over the 1917 x 91 class logits (measured: 32 per thread, i.e. 22 fat
// workgroups per frame, triples both kernels' time -- the scans want every CU).
#ifndef POST_ITEMS
#define POST_ITEMS 8
#endif
#ifndef COMPACT_ITEMS
#define COMPACT_ITEMS 8    // (16 per thread measured slower: 15 vs 11 us)
#endif
#define POST_LIST 1024   // candidates one workgroup stages in LDS before publishing them (more go straight to HBM)
__global__ __launch_bounds__(256) void wz_k_hist(WzPostBuffers b, WzPostConsts k) {
    __shared__ uint32_t h[WZ_HIST_BINS];
    for (int i = threadIdx.x; i < WZ_HIST_BINS; i += 256) h[i] = 0;
    __syncthreads();
    const int f = blockIdx.y, total = k.num_anchors * k.num_classes;
    const int base = blockIdx.x * 256 * POST_ITEMS;
#pragma unroll 4
    for (int it = 0; it < POST_ITEMS; ++it) {
        const int j = base + it * 256 + threadIdx.x;
        uint32_t key, tie;
        if (j < total && wz_candidate(b, k, f, j, key, tie)) atomicAdd(&h[key >> 20], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < WZ_HIST_BINS; i += 256)
        if (h[i]) atomicAdd(&b.hist[(size_t)f * WZ_HIST_BINS + i], h[i]);
}

// threshold bin: the largest b < hi with (number of candidates in bins [b, hi)) >= target, else 0;
// *total_out = candidates in bins [0, hi).
// Called by every thread of the block (blockDim.x in {256, 1024}); `sh` = 64 uint32 of LDS scratch.
// Thread t owns bins [t*per, (t+1)*per); suffix sums run across lanes (shuffles) and waves (LDS).
__device__ int wz_threshold_bin(const uint32_t* __restrict__ ghist, uint32_t* sh, uint32_t target,
                                uint32_t* total_out, int hi = WZ_HIST_BINS) {
    const int nth = blockDim.x, per = WZ_HIST_BINS / nth;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
    uint32_t v[4] = {0, 0, 0, 0};
    uint32_t sum = 0;
    for (int i = 0; i < per; ++i) {
        v[i] = (tid * per + i < hi) ? ghist[tid * per + i] : 0u;   // bins >= hi are already consumed
        sum += v[i];
    }
    uint32_t s = sum;   // -> sum over lanes >= lane of this wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_down(s, o);
        if (lane + o < 64) s += x;
    }
    if (lane == 0) sh[wave] = s;
    if (tid == 0) sh[32] = 0;
    __syncthreads();
    uint32_t higher = 0, total = 0;
    for (int w = 0; w < nw; ++w) {
        total += sh[w];
        if (w > wave) higher += sh[w];
    }
    uint32_t run = s + higher - sum;   // candidates in bins above this thread's bins
    int best = -1;
    for (int i = per - 1; i >= 0; --i) {
        run += v[i];
        if (run >= target) {
            best = tid * per + i;
            break;
        }
    }
    if (best >= 0) atomicMax(&sh[32], (uint32_t)(best + 1));
    __syncthreads();
    if (total_out) *total_out = total;
    const uint32_t r = sh[32];
    __syncthreads();
    return r ? (int)r - 1 : 0;
}

__global__ __launch_bounds__(256) void wz_k_compact(WzPostBuffers b, WzPostConsts k) {
    __shared__ uint32_t sh[64];
    __shared__ uint32_t s_cnt, s_base;
    const int f = blockIdx.y, total = k.num_anchors * k.num_classes;
    if (threadIdx.x == 0) s_cnt = 0;
    // the scan's loads go out first: they land while the threshold bin is being derived from the histogram (two
    // barriers and a dependent chain of its own)
    const int base = blockIdx.x * 256 * COMPACT_ITEMS;
    float lg[COMPACT_ITEMS];
    bool live[COMPACT_ITEMS];
#pragma unroll
    for (int it = 0; it < COMPACT_ITEMS; ++it) {
        const int j = base + it * 256 + threadIdx.x;
        const int a = j / k.num_classes, col = j - a * k.num_classes;
        live[it] = j < total && col != 0 && b.valid[(size_t)f * k.num_anchors + a];
        lg[it] = live[it] ? b.logits[(size_t)f * total + j] : 0.0f;
    }
    uint32_t all = 0;
    const uint32_t thr = (uint32_t)wz_threshold_bin(b.hist + (size_t)f * WZ_HIST_BINS, sh, WZ_CAND_TARGET, &all);
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // every block computes the same band; one publishes it for wz_k_nms
        b.band[2 * f] = thr;
        b.band[2 * f + 1] = all;
    }
    __shared__ uint2 s_list[POST_LIST];
#pragma unroll
    for (int it = 0; it < COMPACT_ITEMS; ++it) {
        const int j = base + it * 256 + threadIdx.x;
        const float sc = wz_sigmoid(lg[it]);                  // exactly wz_candidate()'s arithmetic
        const uint32_t key = __float_as_uint(sc);
        if (live[it] && sc > k.score_thr && (key >> 20) >= thr) {
            const int a = j / k.num_classes, col = j - a * k.num_classes;
            const uint32_t tie = (uint32_t)(col - 1) * (uint32_t)k.num_anchors + (uint32_t)a;
            const uint32_t p = atomicAdd(&s_cnt, 1u);     // LDS atomic: order is irrelevant, the list gets sorted
            if (p < POST_LIST) {
                s_list[p] = make_uint2(key, tie);
            } else {                                      // a dense band: publish this one directly
                const uint32_t q = atomicAdd(&b.count[f], 1u);
                if (q < WZ_CAND_CAP) b.cand[(size_t)f * WZ_CAND_CAP + q] = make_uint2(key, tie);
            }
        }
    }
    __syncthreads();
    const uint32_t n_list = min(s_cnt, (uint32_t)POST_LIST);
    if (threadIdx.x == 0 && n_list) s_base = atomicAdd(&b.count[f], n_list);   // one global atomic per block
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_list; i += 256) {
        const uint32_t p = s_base + i;
        if (p < WZ_CAND_CAP) b.cand[(size_t)f * WZ_CAND_CAP + p] = s_list[i];
    }
}

// ---------------------------------------------------------------------------------------------
// greedy NMS walk: one workgroup per frame
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wz_iou(const float4_t a, const float4_t c) {
    // tensorflow/core/kernels/non_max_suppression_op.cc IOU(): boxes are (y1,x1,y2,x2)
    const float ymin_i = fminf(a[0], a[2]), xmin_i = fminf(a[1], a[3]);
    const float ymax_i = fmaxf(a[0], a[2]), xmax_i = fmaxf(a[1], a[3]);
    const float ymin_j = fminf(c[0], c[2]), xmin_j = fminf(c[1], c[3]);
    const float ymax_j = fmaxf(c[0], c[2]), xmax_j = fmaxf(c[1], c[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    const float iy0 = fmaxf(ymin_i, ymin_j), ix0 = fmaxf(xmin_i, xmin_j);
    const float iy1 = fminf(ymax_i, ymax_j), ix1 = fminf(xmax_i, xmax_j);
    const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
    return inter / (area_i + area_j - inter);
}

// Exact, division-free form of `IOU(i, j) > thr` for boxes already normalised to (ymin,xmin,ymax,xmax)
// with their areas precomputed.  TF evaluates q = RN_f32(inter / uni) > thr.  With m = the midpoint
// between thr and the next float above it, q > thr  <=>  inter/uni > m, or inter/uni == m when
// round-to-nearest-even resolves the tie upward (thr's mantissa odd).  inter, uni are floats and m has
// 25 significant bits, so m * uni is exact in double and the comparison is exact.
struct WzIouThr {
    double mid;
    float lo, hi;   // thr * (1 -/+ 1e-3)
    float cl, ch;   // lo / (1 + lo), hi / (1 + hi): inter > lo * uni  <=>  inter > cl * (area_a + area_c)
    bool tie_up;
};
__device__ __forceinline__ WzIouThr wz_iou_thr(float thr) {
    WzIouThr t;
    const float up = __uint_as_float(__float_as_uint(thr) + 1u);   // thr >= 0
    t.mid = ((double)thr + (double)up) * 0.5;
    t.tie_up = (__float_as_uint(thr) & 1u) != 0u;
    t.lo = thr * 0.999f;
    t.hi = thr * 1.001f;
    t.cl = (float)((double)t.lo / (1.0 + (double)t.lo));
    t.ch = (float)((double)t.hi / (1.0 + (double)t.hi));
    return t;
}
__device__ __forceinline__ bool wz_iou_exceeds(const float4_t a, float area_a, const float4_t c, float area_c,
                                               const WzIouThr t) {
    if (area_a <= 0.0f || area_c <= 0.0f) return false;   // IOU() returns 0 and thr >= 0
    const float iy0 = fmaxf(a[0], c[0]), ix0 = fmaxf(a[1], c[1]);
    const float iy1 = fminf(a[2], c[2]), ix1 = fminf(a[3], c[3]);
    const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
    const float uni = area_a + area_c - inter;
    // clear cases in fp32 (margins 1e-3 >> the 2^-24 roundings), the exact double test only near the threshold
    if (!(inter > t.lo * uni)) return false;
    if (inter > t.hi * uni) return true;
    const double lhs = (double)inter, rhs = t.mid * (double)uni;
    return lhs > rhs || (t.tie_up && lhs == rhs);
}
// The pair loop of the walk runs on ONE CU and is VALU-bound, so the per-box half of the clear-case test is hoisted:
// pre = {cl * area, ch * area, area, class bits}, with +inf in the first two when the area is not positive (IOU() is 0
// for such a box).  inter > lo * uni  <=>  inter * (1 + lo) > lo * (area_a + area_c)  <=>  inter > pre_a[0] + pre_c[0];
// the roundings of the products and of the sum (< 3e-7 relative) are far inside the 1e-3 margins, and everything
// between the two margins still goes through the exact test.
__device__ __forceinline__ float4_t wz_pair_pre(float area, int cls, const WzIouThr t) {
    const float inf = __builtin_inff();
    const bool ok = area > 0.0f;
    return (float4_t){ok ? t.cl * area : inf, ok ? t.ch * area : inf, area, __int_as_float(cls)};
}
// max / min of two NON-NEGATIVE floats (no -0.0, see wz_norm_box) as unsigned integers: one instruction each, where
// `fmaxf` / `fminf` on values that come out of LDS cost an extra canonicalising v_max_f32 per operand
__device__ __forceinline__ float wz_max_nn(float a, float b) { return __uint_as_float(max(__float_as_uint(a), __float_as_uint(b))); }
__device__ __forceinline__ float wz_min_nn(float a, float b) { return __uint_as_float(min(__float_as_uint(a), __float_as_uint(b))); }
// SIGNED: the boxes may have negative coordinates (clip-after-NMS programs walk the boxes as decoded): plain fmaxf / fminf
template <bool SIGNED>
__device__ __forceinline__ bool wz_pair_suppresses(const float4_t a, const float4_t pa, const float4_t c,
                                                   const float4_t pc, const WzIouThr t) {
    // straight-line except for the rare near-threshold case: the pair loop is issue-bound (one CU, ~30 instructions
    // per pair), so every instruction and every divergent branch counts
    const bool same = __float_as_int(pa[3]) == __float_as_int(pc[3]);
    const float iy0 = SIGNED ? fmaxf(a[0], c[0]) : wz_max_nn(a[0], c[0]), ix0 = SIGNED ? fmaxf(a[1], c[1]) : wz_max_nn(a[1], c[1]);
    const float iy1 = SIGNED ? fminf(a[2], c[2]) : wz_min_nn(a[2], c[2]), ix1 = SIGNED ? fminf(a[3], c[3]) : wz_min_nn(a[3], c[3]);
    const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
    const bool above_lo = inter > pa[0] + pc[0], above_hi = inter > pa[1] + pc[1];
    bool r = same & above_hi;
    if (same & above_lo & !above_hi) {
        const float uni = pa[2] + pc[2] - inter;   // as IOU() computes it
        const double lhs = (double)inter, rhs = t.mid * (double)uni;
        r = lhs > rhs || (t.tie_up && lhs == rhs);
    }
    return r;
}
__device__ __forceinline__ float4_t wz_norm_box(const float4_t b, float& area) {
    // + 0.0f turns a -0.0 into +0.0 (the boxes are clipped to [0, 1]): the pair test compares the components as
    // unsigned integers.  Same values, same area.
    const float4_t n = {fminf(b[0], b[2]) + 0.0f, fminf(b[1], b[3]) + 0.0f, fmaxf(b[0], b[2]) + 0.0f, fmaxf(b[1], b[3]) + 0.0f};
    area = (n[2] - n[0]) * (n[3] - n[1]);
    return n;
}

#define NMS_THREADS 1024
#define NMS_KEEP_MAX WZ_NMS_KEEP_MAX   // >= max_total (100); wz_common.h (the engine's overflow message quotes it)
#define NMS_CHUNK 256     // candidates per parallel suppression pass (4 per lane of the scanning wave)
#define NMS_RANK_MAX 1536  // up to here an O(n^2/threads) rank sort beats the barrier-bound bitonic network

struct NmsShared {   // carved from dynamic LDS, every member 16-byte aligned
    unsigned long long keys[WZ_CAND_CAP];   // 32 KiB
    unsigned long long keys2[WZ_CAND_CAP];  // 32 KiB (rank-sort destination)
    float4_t sbox[WZ_CAND_CAP];             // 64 KiB
    float4_t kbox[NMS_KEEP_MAX];
    float kscore[NMS_KEEP_MAX];
    int32_t kcls[NMS_KEEP_MAX];
    unsigned long long red[NMS_THREADS / 64];
    // chunk state of the parallel suppression pass (NMS_CHUNK sorted candidates at a time)
    unsigned long long supp[NMS_CHUNK][NMS_CHUNK / 64];   // supp[j] = mask of earlier chunk members i < j that suppress j
    float4_t cnorm[NMS_CHUNK];                            // candidate boxes normalised to (ymin,xmin,ymax,xmax)
    float4_t knorm[NMS_KEEP_MAX];                         // the same for the kept list
    float4_t cpre[NMS_CHUNK];                             // wz_pair_pre of the chunk members
    float4_t kpre[NMS_KEEP_MAX];                          // ... and of the kept list
    float carea[NMS_CHUNK];
    float karea[NMS_KEEP_MAX];
    int32_t ccls[NMS_CHUNK];
    uint32_t cdead[NMS_CHUNK];                            // 1 = suppressed by a box kept in an earlier chunk / band
    int32_t cnext[NMS_CHUNK];                             // next chunk member of the same class, or 0x7fffffff
    unsigned long long keptmask[NMS_CHUNK / 64];          // chunk members kept by the per-class resolution
    uint32_t hist[64];
    int32_t kept;
    uint32_t ncand;
    int32_t pad[2];
};

// wave 0: try to keep candidate (box, cls, score); returns new kept count (uniform across the wave)
__device__ __forceinline__ int wz_try_keep(NmsShared* S, int kept, const float4_t box, int cls, float score,
                                           const WzPostConsts& k, int lane) {
    bool sup = false;
    int same = 0;
    for (int j = lane; j < kept; j += 64) {
        if (S->kcls[j] == cls) {
            ++same;
            if (wz_iou(box, S->kbox[j]) > k.iou_thr) sup = true;
        }
    }
    const bool any_sup = __any(sup);
    int tot = same;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (!any_sup && tot < k.max_per_class) {
        if (lane == 0) {
            float ar;
            S->kbox[kept] = box;
            S->knorm[kept] = wz_norm_box(box, ar);
            S->karea[kept] = ar;
            S->kpre[kept] = wz_pair_pre(ar, cls, wz_iou_thr(k.iou_thr));
            S->kcls[kept] = cls;
            S->kscore[kept] = score;
        }
        ++kept;
    }
    return kept;
}

// One band = all candidates whose score bits fall in histogram bins [lo_bin, hi_bin), `cnt` of them in
// S->keys (unsorted composites).  Sort, gather boxes, walk.  Returns the new kept count (block-uniform).
template <bool SIGNED, bool COUNT>
__device__ int wz_nms_band(NmsShared* S, const WzPostBuffers& b, const WzPostConsts& k, int f, int cnt, int kept) {
    const int tid = threadIdx.x;
    const int A = k.num_anchors;
    unsigned long long* sorted = S->keys;
    if (cnt <= NMS_RANK_MAX) {
        // rank sort: keys are unique (the tie index is), so rank = #larger keys is a permutation.  A key's rank is counted by P threads
        // (a power of two: the P threads of a key are neighbours in one wavefront), each over its share of the list -- 8 keys per step
        // from LDS, same address = broadcast; 64-bit compares run at a quarter of the rate, and with one thread per key a few hundred
        // candidates left 3/4 of the workgroup idle through 270 of them -- and the partial ranks are summed with shuffles.
        const int cnt8 = (cnt + 7) & ~7;
        int P = 1;
        while (P < 16 && cnt * (P * 2) <= NMS_THREADS) P *= 2;
        const int share = (((cnt8 + P - 1) / P) + 7) & ~7;    // keys per part, a multiple of 8
        for (int i = cnt + tid; i < cnt8; i += NMS_THREADS) S->keys[i] = 0ull;   // 0 is never "larger"
        __syncthreads();
        for (int t0 = 0; t0 < cnt * P; t0 += NMS_THREADS) {   // (one trip unless the list is longer than the workgroup)
            const int t = t0 + tid;
            const int i = min(t / P, cnt - 1), part = t & (P - 1);
            const unsigned long long mine = S->keys[i];
            const int j0 = part * share, j1 = min(j0 + share, cnt8);
            int r = 0;
            for (int j = j0; j < j1; j += 8) {
                unsigned long long kk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kk[u] = S->keys[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) r += (kk[u] > mine) ? 1 : 0;
            }
            for (int o = 1; o < P; o <<= 1) r += __shfl_xor(r, o);
            if (part == 0 && t < cnt * P) S->keys2[r] = mine;
        }
        __syncthreads();
        sorted = S->keys2;
    } else {
        int npow = 64;
        while (npow < cnt) npow <<= 1;
        for (int i = cnt + tid; i < npow; i += NMS_THREADS) S->keys[i] = 0ull;
        __syncthreads();
        for (int size = 2; size <= npow; size <<= 1) {   // bitonic network, descending
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = tid; i < (npow >> 1); i += NMS_THREADS) {
                    const int lo = 2 * i - (i & (stride - 1));
                    const int hi = lo + stride;
                    const bool desc = ((lo & size) == 0);
                    const unsigned long long x = S->keys[lo], y = S->keys[hi];
                    if ((x < y) == desc) { S->keys[lo] = y; S->keys[hi] = x; }
                }
                __syncthreads();
            }
        }
    }
    if (tid == 0) b.dbg[(size_t)f * 16 + 5] = wall_clock64();
    for (int i = tid; i < cnt; i += NMS_THREADS) {
        const uint32_t tie = 0xFFFFFFFFu - (uint32_t)(sorted[i] & 0xFFFFFFFFull);
        const int a = (int)(tie % (uint32_t)A);
        S->sbox[i] = *reinterpret_cast<const float4_t*>(b.boxes + ((size_t)f * A + a) * 4);
    }
    __syncthreads();
    if (tid == 0) b.dbg[(size_t)f * 16 + 6] = wall_clock64();

    // Greedy NMS over the sorted band, NMS_CHUNK candidates at a time, with no serial pass over the band:
    //  (1) in parallel: which chunk members are suppressed by boxes kept earlier (cdead) and the pairwise
    //      "i suppresses j" relation inside the chunk (same class, i before j, IoU > thr) as bit masks
    //      supp[j], each 64-bit word built in one thread's register;
    //  (2) "j is kept iff it is alive and no KEPT earlier member is in supp[j]" is a recurrence in band
    //      order with a unique solution; one wavefront iterates it to its fixed point with ballots (round
    //      r finalises the first r members; real detections need a handful of rounds).  When the per-class
    //      cap can bind (max_per_class < max_total) one thread per class walks its chain instead;
    //  (3) the kept bits, read in band order, are the rows; the first max_total of them are exactly what a
    //      one-by-one walk keeps (a candidate's fate depends only on higher-scored boxes of its own class,
    //      so the members behind the cut cannot change the rows before it).
    const WzIouThr ithr = wz_iou_thr(k.iou_thr);
    constexpr bool count_classes = COUNT;   // k.max_per_class < k.max_total: the per-class cap can bind (a template parameter of the kernel)
    const int ncls = k.num_classes - 1;
    // the sort buffer that does not hold the sorted band is free: per-class tables live there
    int32_t* const first = reinterpret_cast<int32_t*>(sorted == S->keys2 ? S->keys : S->keys2);   // [ncls]
    int32_t* const ccount = first + 4096;                                                          // [ncls]
    // (a FIRST chunk of 128 candidates was tried: a quarter of the pair work when it is the only one -- but the benchmark's scenes keep
    // 100 of ~270 candidates, the second chunk then costs its fixed part again: 27 -> 37 us; profiles/r04_nms_pair_list.txt)
    for (int base = 0, m = 0; base < cnt && kept < k.max_total; base += m) {
        m = min(NMS_CHUNK, cnt - base);
        if constexpr (!count_classes) {
            // ---- the chunk as a list of same-class PAIRS (round 4).  Only members of one class can suppress each other, and the blocks
            // of the 256 x 256 pair matrix hold all classes mixed: walked block by block (the COUNT path below, rounds 1 .. 3) every
            // lane tests 128 earlier members on average, ~30 instructions each, whatever their class -- 10.6 us of a 28 us kernel,
            // dealt out statically over the SIMDs.  Here the members are grouped by class (a counting sort with LDS atomics: the order
            // INSIDE a class is whatever the atomics made it -- which of two members is the earlier one is read from their band indices),
            // the same-class pairs are numbered class after class, and the pair numbers are dealt out evenly: every wavefront takes a
            // run of them, its lanes consecutive pairs (neighbouring lanes read neighbouring members: no bank conflicts, the later
            // member of a pair mostly a broadcast).  A trained detector's scene (dozens of classes in a band): 10.5 -> 2 - 3 us; one class
            // holding everything: the same pairs as before.  Classes are bucketed modulo 128 (two classes in one bucket only cost
            // pair tests: the test compares classes).
            int32_t* const posn = first + 256;      // [256]  rank inside its bucket, then its position in class order
            int32_t* const bcnt = first + 512;      // [128]  members per bucket
            int32_t* const bstart = first + 640;    // [128]  first position of a bucket
            int32_t* const pstart = first + 768;    // [129]  pairs in front of a bucket's own (exclusive prefix), [128] = all
            int32_t* const orig = S->cnext;         // [256]  band index of the member at a position
            uint32_t* const supp32 = reinterpret_cast<uint32_t*>(&S->supp[0][0]);   // [256][8]: bit a of row b = "a suppresses b"
            if (tid < 128) bcnt[tid] = 0;
            S->supp[tid >> 2][tid & 3] = 0ull;      // (NMS_THREADS == NMS_CHUNK * NMS_CHUNK / 64: one word each)
            if (tid < NMS_CHUNK / 64) S->keptmask[tid] = 0ull;
            if (tid < 8) S->hist[tid] = 0u;
            if (tid < NMS_CHUNK) S->cdead[tid] = 0u;
            __syncthreads();
            int my_cls = -1;
            if (tid < m) {
                const uint32_t tie = 0xFFFFFFFFu - (uint32_t)(sorted[base + tid] & 0xFFFFFFFFull);
                my_cls = (int)(tie / (uint32_t)A);
                posn[tid] = atomicAdd(&bcnt[my_cls & 127], 1);
            }
            __syncthreads();
            if (tid < 64) {   // buckets 2 l, 2 l + 1: starts (exclusive prefix of the counts) and pair prefix, one wavefront
                const int n0 = bcnt[2 * tid], n1 = bcnt[2 * tid + 1];
                const int ms = n0 + n1, ps = n0 * (n0 - 1) / 2 + n1 * (n1 - 1) / 2;
                int mi = ms, pi = ps;                                     // inclusive scans over the lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int um = __shfl_up(mi, o), up = __shfl_up(pi, o);
                    if (tid >= o) { mi += um; pi += up; }
                }
                bstart[2 * tid] = mi - ms;
                bstart[2 * tid + 1] = mi - ms + n0;
                pstart[2 * tid] = pi - ps;
                pstart[2 * tid + 1] = pi - ps + n0 * (n0 - 1) / 2;
                if (tid == 63) pstart[128] = pi;
            }
            __syncthreads();
            if (tid < NMS_CHUNK) {
                if (tid < m) {
                    const int p = bstart[my_cls & 127] + posn[tid];
                    posn[tid] = p;
                    float ar;
                    S->ccls[p] = my_cls;
                    S->cnorm[p] = wz_norm_box(S->sbox[base + tid], ar);
                    S->carea[p] = ar;
                    S->cpre[p] = wz_pair_pre(ar, my_cls, ithr);
                    orig[p] = tid;
                } else {
                    S->ccls[tid] = -1;
                    S->cpre[tid] = wz_pair_pre(0.0f, -1, ithr);
                }
            }
            __syncthreads();
            if (tid == 0 && base == 0) b.dbg[(size_t)f * 16 + 11] = wall_clock64();
            for (int p = tid; p < m * kept; p += NMS_THREADS) {          // vs boxes kept before this chunk
                const int i = p / kept, j = p - i * kept;
                if (wz_pair_suppresses<SIGNED>(S->cnorm[i], S->cpre[i], S->knorm[j], S->kpre[j], ithr))
                    S->cdead[i] = 1u;                                    // benign race: every writer stores 1
            }
            {   // pair numbers: wave w takes [w * run, (w + 1) * run), lane l of it the numbers w * run + l, + 64, + 128, ...
                const int total = pstart[128];
                const int run = (((total + NMS_THREADS / 64 - 1) / (NMS_THREADS / 64)) + 63) & ~63;
                int Q = (tid >> 6) * run + (tid & 63);
                const int Q1 = min(((tid >> 6) + 1) * run, total);
                if (Q < Q1) {
                    auto bucket_of = [&](int Qx) {                        // the bucket holding pair Qx: pstart[c] <= Qx < pstart[c + 1]
                        int lo = 0, hi = 128;                             // (seven dependent LDS reads; walking the buckets one by one
                        while (hi - lo > 1) {                             //  across the empty ones of a 90-class table cost up to 70)
                            const int mid = (lo + hi) >> 1;
                            if (pstart[mid] <= Qx) lo = mid; else hi = mid;
                        }
                        return lo;
                    };
                    int c = bucket_of(Q);
                    int pb = bstart[c], pend = pstart[c + 1];
                    const int q = Q - pstart[c];                          // pair (a < b) number q of this bucket: q = b (b - 1) / 2 + a
                    int bb = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)q)) * 0.5f);
                    while (bb * (bb - 1) / 2 > q) --bb;
                    while ((bb + 1) * bb / 2 <= q) ++bb;
                    int aa = q - bb * (bb - 1) / 2;
                    for (;;) {
                        const int pa = pb + aa, pbb = pb + bb;
                        if (wz_pair_suppresses<SIGNED>(S->cnorm[pbb], S->cpre[pbb], S->cnorm[pa], S->cpre[pa], ithr)) {
                            const bool a_first = orig[pa] < orig[pbb];    // the earlier one in the band suppresses the later one
                            const int pe = a_first ? pa : pbb, pl = a_first ? pbb : pa;
                            atomicOr(&supp32[pl * 8 + (pe >> 5)], 1u << (pe & 31));
                        }
                        Q += 64;
                        if (Q >= Q1) break;
                        aa += 64;
                        if (Q >= pend) {                                  // into a later bucket
                            c = bucket_of(Q);
                            pb = bstart[c];
                            aa = Q - pstart[c];
                            bb = 1;
                            pend = pstart[c + 1];
                        }
                        while (aa >= bb) {                                // row b of the triangle holds b pairs
                            aa -= bb;
                            ++bb;
                        }
                    }
                }
            }
            __syncthreads();
            if (tid == 0 && base == 0) b.dbg[(size_t)f * 16 + 12] = wall_clock64();
            // Member p (class order) is kept iff it is not dead and no KEPT earlier member is in supp[p]: the recurrence of the COUNT
            // path below, in class order -- a member only depends on earlier members of its own class, which precede it here as well.
            if (tid < 64) {
                unsigned long long row[NMS_CHUNK / 64][NMS_CHUNK / 64];
                bool alive[NMS_CHUNK / 64];
                unsigned long long km[NMS_CHUNK / 64];
#pragma unroll
                for (int q = 0; q < NMS_CHUNK / 64; ++q) {
                    const int j = tid + 64 * q;
#pragma unroll
                    for (int w = 0; w < NMS_CHUNK / 64; ++w) row[q][w] = S->supp[j][w];
                    alive[q] = j < m && !S->cdead[j];
                    km[q] = __ballot(alive[q]);
                }
                for (int round = 0; round <= NMS_CHUNK; ++round) {
                    unsigned long long nk[NMS_CHUNK / 64];
                    bool same = true;
#pragma unroll
                    for (int q = 0; q < NMS_CHUNK / 64; ++q) {
                        unsigned long long hit = 0ull;
#pragma unroll
                        for (int w = 0; w < NMS_CHUNK / 64; ++w) hit |= row[q][w] & km[w];
                        nk[q] = __ballot(alive[q] && hit == 0ull);
                        same = same && nk[q] == km[q];
                    }
#pragma unroll
                    for (int q = 0; q < NMS_CHUNK / 64; ++q) km[q] = nk[q];
                    if (same) break;
                }
                // the kept bits in BAND order (the rows come out by score): member p sets the bit of its band index
#pragma unroll
                for (int q = 0; q < NMS_CHUNK / 64; ++q) {
                    const int pp = tid + 64 * q;
                    if ((km[q] >> tid) & 1ull) atomicOr(&S->hist[orig[pp] >> 5], 1u << (orig[pp] & 31));
                }
            }
            __syncthreads();
            if (tid == 0 && base == 0) b.dbg[(size_t)f * 16 + 13] = wall_clock64();
            unsigned long long bandmask[NMS_CHUNK / 64];
            int total_new = 0;
#pragma unroll
            for (int w = 0; w < NMS_CHUNK / 64; ++w) {
                bandmask[w] = (unsigned long long)S->hist[2 * w] | ((unsigned long long)S->hist[2 * w + 1] << 32);
                total_new += __popcll(bandmask[w]);
            }
            for (int i = tid; i < m; i += NMS_THREADS) {                 // materialise the newly kept rows, band order
                unsigned long long mine = 0ull;
                int before = 0;
#pragma unroll
                for (int w = 0; w < NMS_CHUNK / 64; ++w) {
                    if (w == (i >> 6)) mine = bandmask[w];
                    if (w < (i >> 6)) before += __popcll(bandmask[w]);
                }
                if (!((mine >> (i & 63)) & 1ull)) continue;
                before += __popcll(mine & ((1ull << (i & 63)) - 1ull));
                const int j = kept + before;
                if (j >= k.max_total) continue;
                const int p = posn[i];
                const unsigned long long comp = sorted[base + i];
                S->kbox[j] = S->sbox[base + i];
                S->knorm[j] = S->cnorm[p];
                S->karea[j] = S->carea[p];
                S->kpre[j] = S->cpre[p];
                S->kcls[j] = S->ccls[p];
                S->kscore[j] = __uint_as_float((uint32_t)(comp >> 32));
            }
            __syncthreads();
            kept = min(kept + total_new, k.max_total);
            continue;
        }
        for (int i = tid; i < NMS_CHUNK; i += NMS_THREADS) {
            S->cdead[i] = 0u;
            if (i < m) {
                const uint32_t tie = 0xFFFFFFFFu - (uint32_t)(sorted[base + i] & 0xFFFFFFFFull);
                S->ccls[i] = (int)(tie / (uint32_t)A);
                float ar;
                S->cnorm[i] = wz_norm_box(S->sbox[base + i], ar);
                S->carea[i] = ar;
                S->cpre[i] = wz_pair_pre(ar, S->ccls[i], ithr);
            } else {
                S->ccls[i] = -1;
                S->cpre[i] = wz_pair_pre(0.0f, -1, ithr);
            }
        }
        if constexpr (count_classes)
            for (int c = tid; c < ncls; c += NMS_THREADS) {
                first[c] = 0x7fffffff;
                ccount[c] = 0;
            }
        if (tid < NMS_CHUNK / 64) S->keptmask[tid] = 0ull;
        __syncthreads();
        if (tid == 0 && base == 0) b.dbg[(size_t)f * 16 + 11] = wall_clock64();
        for (int p = tid; p < m * kept; p += NMS_THREADS) {          // vs boxes kept before this chunk
            const int i = p / kept, j = p - i * kept;
            if (wz_pair_suppresses<SIGNED>(S->cnorm[i], S->cpre[i], S->knorm[j], S->kpre[j], ithr))
                S->cdead[i] = 1u;                                    // benign race: every writer stores 1
        }
        {   // supp[j][w]: bit i of word w = "member 64w + i (before j) suppresses j", built in registers.
            // The (64 members j) x (64 members i) blocks below the diagonal cost 64 pair tests per lane, the diagonal
            // ones half of that, the ones above nothing: 8 units of work in all.  The loop is VALU-bound and a wave
            // stays on SIMD (wave & 3), so blocks are dealt out by hand, 2 units per SIMD: each full block as two
            // 32-member halves on two waves of one SIMD, the four diagonal blocks on the fourth SIMD.
            //                                   wave:  0   1   2   3   4   5   6   7   8   9  10  11  12  13  14  15
            constexpr unsigned char TASK_JB[16] = {1,  2,  3,  0,  1,  2,  3,  1,  2,  3,  3,  2,  2,  3,  3,  3};
            constexpr unsigned char TASK_W[16]  = {0,  1,  1,  0,  0,  1,  1,  1,  0,  0,  2,  2,  0,  0,  2,  3};
            constexpr unsigned char TASK_H[16]  = {0,  0,  0,  2,  1,  1,  1,  2,  0,  0,  0,  2,  1,  1,  1,  2};   // 2 = whole word
            const int wv = tid >> 6;
            const int j = TASK_JB[wv] * 64 + (tid & 63), w = TASK_W[wv], half = TASK_H[wv];
            uint32_t* const word = reinterpret_cast<uint32_t*>(&S->supp[j][w]);
            float4_t bj = {0.f, 0.f, 0.f, 0.f}, pj = {0.f, 0.f, 0.f, 0.f};
            if (j < m) {
                bj = S->cnorm[j];
                pj = S->cpre[j];
            }
            for (int h = (half == 2 ? 0 : half); h <= (half == 2 ? 1 : half); ++h) {   // 32 earlier members at a time
                const int i0 = w * 64 + h * 32;
                const int i_end = j < m ? min(i0 + 32, j) : i0;      // i before j
                uint32_t bits = 0u;
                for (int i = i0; i < i_end; ++i)                      // i is wave-uniform: LDS broadcasts
                    bits |= wz_pair_suppresses<SIGNED>(bj, pj, S->cnorm[i], S->cpre[i], ithr) ? 1u << (i - i0) : 0u;
                word[h] = bits;
            }
        }
        if (tid < NMS_CHUNK) {   // words above the diagonal (members behind j) are empty
            for (int w = (tid >> 6) + 1; w < NMS_CHUNK / 64; ++w) S->supp[tid][w] = 0ull;
        }
        if constexpr (count_classes) {   // per-class chains in band order (only needed when the per-class cap can bind)
            for (int j = tid; j < kept; j += NMS_THREADS) atomicAdd(&ccount[S->kcls[j]], 1);
            for (int i = tid; i < m; i += NMS_THREADS) {
                atomicMin(&first[S->ccls[i]], i);
                int nx = 0x7fffffff;
                for (int j = i + 1; j < m; ++j)
                    if (S->ccls[j] == S->ccls[i]) { nx = j; break; }
                S->cnext[i] = nx;
            }
        }
        __syncthreads();
        if (tid == 0 && base == 0) b.dbg[(size_t)f * 16 + 12] = wall_clock64();
        if constexpr (!count_classes) {
            // Member j is kept iff it is not dead and no KEPT earlier member is in supp[j].  That is a
            // recurrence in band order with a unique solution; iterate "kept = alive & no kept suppressor"
            // from "everything alive is kept": after r rounds the first r members are final, and the
            // typical dependency depth is a handful.  One wavefront, 4 members per lane, ballots only.
            if (tid < 64) {
                unsigned long long row[NMS_CHUNK / 64][NMS_CHUNK / 64];
                bool alive[NMS_CHUNK / 64];
                unsigned long long km[NMS_CHUNK / 64];
#pragma unroll
                for (int q = 0; q < NMS_CHUNK / 64; ++q) {
                    const int j = tid + 64 * q;
#pragma unroll
                    for (int w = 0; w < NMS_CHUNK / 64; ++w) row[q][w] = S->supp[j][w];
                    alive[q] = j < m && !S->cdead[j];
                    km[q] = __ballot(alive[q]);
                }
                for (int round = 0; round <= NMS_CHUNK; ++round) {
                    unsigned long long nk[NMS_CHUNK / 64];
                    bool same = true;
#pragma unroll
                    for (int q = 0; q < NMS_CHUNK / 64; ++q) {
                        unsigned long long hit = 0ull;
#pragma unroll
                        for (int w = 0; w < NMS_CHUNK / 64; ++w) hit |= row[q][w] & km[w];
                        nk[q] = __ballot(alive[q] && hit == 0ull);
                        same = same && nk[q] == km[q];
                    }
#pragma unroll
                    for (int q = 0; q < NMS_CHUNK / 64; ++q) km[q] = nk[q];
                    if (same) break;
                }
                if (tid < NMS_CHUNK / 64) {
#pragma unroll
                    for (int q = 0; q < NMS_CHUNK / 64; ++q)
                        if (tid == q) S->keptmask[q] = km[q];
                }
            }
        } else
        for (int c = tid; c < ncls; c += NMS_THREADS) {              // one thread per class walks its chain
            int j = first[c];
            if (j == 0x7fffffff) continue;
            unsigned long long km[NMS_CHUNK / 64];
#pragma unroll
            for (int w = 0; w < NMS_CHUNK / 64; ++w) km[w] = 0ull;
            int have = ccount[c];
            while (j != 0x7fffffff) {
                unsigned long long hit = 0ull;
#pragma unroll
                for (int w = 0; w < NMS_CHUNK / 64; ++w) hit |= S->supp[j][w] & km[w];
                if (!S->cdead[j] && hit == 0ull && have < k.max_per_class) {
#pragma unroll
                    for (int w = 0; w < NMS_CHUNK / 64; ++w)
                        if (w == (j >> 6)) km[w] |= 1ull << (j & 63);
                    ++have;
                }
                j = S->cnext[j];
            }
#pragma unroll
            for (int w = 0; w < NMS_CHUNK / 64; ++w)
                if (km[w]) atomicOr(&S->keptmask[w], km[w]);
        }
        __syncthreads();
        if (tid == 0 && base == 0) b.dbg[(size_t)f * 16 + 13] = wall_clock64();
        int total_new = 0;
#pragma unroll
        for (int w = 0; w < NMS_CHUNK / 64; ++w) total_new += __popcll(S->keptmask[w]);
        for (int i = tid; i < m; i += NMS_THREADS) {                 // materialise the newly kept rows, band order
            if (!((S->keptmask[i >> 6] >> (i & 63)) & 1ull)) continue;
            int before = __popcll(S->keptmask[i >> 6] & ((1ull << (i & 63)) - 1ull));
            for (int w = 0; w < (i >> 6); ++w) before += __popcll(S->keptmask[w]);
            const int j = kept + before;
            if (j >= k.max_total) continue;
            const unsigned long long comp = sorted[base + i];
            S->kbox[j] = S->sbox[base + i];
            S->knorm[j] = S->cnorm[i];
            S->karea[j] = S->carea[i];
            S->kpre[j] = S->cpre[i];
            S->kcls[j] = S->ccls[i];
            S->kscore[j] = __uint_as_float((uint32_t)(comp >> 32));
        }
        __syncthreads();
        kept = min(kept + total_new, k.max_total);
    }
    if (tid == 0) b.dbg[(size_t)f * 16 + 7] = wall_clock64();
    return kept;
}

// A band too crowded for the LDS list (massive score ties): exact one-candidate-per-scan walk over
// composites in [lower, upper).  Slow, only reachable with pathological inputs.
__device__ int wz_nms_band_serial(NmsShared* S, const WzPostBuffers& b, const WzPostConsts& k, int f, int kept,
                                  unsigned long long lower, unsigned long long upper) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A = k.num_anchors, n_entries = A * k.num_classes;
    unsigned long long bound = upper;
    while (kept < k.max_total) {
        unsigned long long best = 0ull;
        for (int j = tid; j < n_entries; j += NMS_THREADS) {
            uint32_t key, tie;
            if (wz_candidate(b, k, f, j, key, tie)) {
                const unsigned long long comp = ((unsigned long long)key << 32) | (0xFFFFFFFFu - tie);
                if (comp < bound && comp >= lower && comp > best) best = comp;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            if (other > best) best = other;
        }
        if (lane == 0) S->red[wave] = best;
        __syncthreads();
        best = 0ull;
        for (int w = 0; w < NMS_THREADS / 64; ++w)
            if (S->red[w] > best) best = S->red[w];
        __syncthreads();
        if (best == 0ull) break;
        bound = best;
        if (wave == 0) {
            const uint32_t tie = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull);
            const int cls = (int)(tie / (uint32_t)A), a = (int)(tie % (uint32_t)A);
            const float4_t box = *reinterpret_cast<const float4_t*>(b.boxes + ((size_t)f * A + a) * 4);
            kept = wz_try_keep(S, kept, box, cls, __uint_as_float((uint32_t)(best >> 32)), k, lane);
            if (lane == 0) S->kept = kept;
        }
        __syncthreads();
        kept = S->kept;
    }
    return kept;
}

// row fill + per-camera filters of one detection (defined with wz_k_rows below)
__device__ void wz_make_row(const WzFrameDesc& fd, const WzCamFilter* __restrict__ cams, bool on, const float4_t bx,
                            float score, int label, wz_detection_t* __restrict__ row, uint8_t* __restrict__ pass);

// Self-scan mode: one pass of the frame's workgroup over its 1917 x 91 logits collects the candidates of a band of score
// bins [lo, hi) into the LDS list -- no histogram, no compaction kernel, no 256-CU launches in front of the walk.
//   * A logit below `logit_floor(lo)` cannot reach bin lo (sigmoid is monotone; the floor is lowered by a margin four
//     orders of magnitude above its rounding error), so all but a few hundred entries cost one load and one compare;
//     the survivors go through exactly wz_candidate()'s arithmetic.
//   * The walk is exact for ANY sequence of bands that partitions the score range from the top down (a candidate's
//     fate depends only on higher-scored boxes of its class), so the band edges need not come from a histogram: the
//     first band starts at a per-slot hint (where it started last time, nudged so that it holds a few hundred
//     candidates); a band that overflows the list is retried with its lower edge raised (down to a single bin, which
//     then takes the serial path); when a band runs dry before max_total rows are kept the next one reaches further
//     down, twice as far each time, until bin 0.
__device__ __forceinline__ float wz_logit_floor(int bin) {
    if (bin <= 0) return -__builtin_inff();
    const float s = __uint_as_float((uint32_t)bin << 20);
    if (!(s < 1.0f)) return 15.0f;                       // sigmoid(x) rounds to 1.0f only for x > 16.6
    const float l = logf(s / (1.0f - s));
    return l - 0.01f - 1e-3f * fabsf(l);
}
__device__ uint32_t wz_nms_scan_band(NmsShared* S, const WzPostBuffers& b, const WzPostConsts& k, int f, int lo_bin,
                                     int hi_bin) {
    const int tid = threadIdx.x;
    const int C = k.num_classes, n_entries = k.num_anchors * C;
    const float* __restrict__ lg = b.logits + (size_t)f * n_entries;
    const float lf = wz_logit_floor(lo_bin);
    if (tid == 0) S->ncand = 0;
    __syncthreads();
    auto consider = [&](int j, float x) {
        if (!(x >= lf)) return;
        const int a = j / C, col = j - a * C;
        if (col == 0 || !b.valid[(size_t)f * k.num_anchors + a]) return;
        const float sc = wz_sigmoid(x);                   // wz_candidate()'s arithmetic
        if (!(sc > k.score_thr)) return;
        const uint32_t key = __float_as_uint(sc);
        const int bin = (int)(key >> 20);
        if (bin < lo_bin || bin >= hi_bin) return;
        const uint32_t tie = (uint32_t)(col - 1) * (uint32_t)k.num_anchors + (uint32_t)a;
        const uint32_t pos = atomicAdd(&S->ncand, 1u);
        if (pos < WZ_CAND_CAP) S->keys[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - tie);
    };
    // 16-byte loads where the frame's logits allow it (n_entries is odd for 1917 x 91: the frame base is only 4-byte
    // aligned in general), scalar loads for the ragged head and tail
    const uintptr_t addr = reinterpret_cast<uintptr_t>(lg);
    const int head = (int)(((16 - (addr & 15)) & 15) >> 2);           // entries before the first 16-byte boundary
    const int n4 = (n_entries - min(head, n_entries)) >> 2;
    // eight loads in flight per thread: taken one at a time the scan is a chain of ~43 memory latencies (29 us)
    constexpr int UN = 8;
    for (int q0 = tid; q0 < n4; q0 += NMS_THREADS * UN) {
        float4_t v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int q = q0 + u * NMS_THREADS;
            v[u] = q < n4 ? *reinterpret_cast<const float4_t*>(lg + head + 4 * q) : (float4_t){-1e30f, -1e30f, -1e30f, -1e30f};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float m = fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3]));
            if (m >= lf) {
                const int q = q0 + u * NMS_THREADS;
#pragma unroll
                for (int r = 0; r < 4; ++r) consider(head + 4 * q + r, v[u][r]);
            }
        }
    }
    for (int j = tid; j < head && j < n_entries; j += NMS_THREADS) consider(j, lg[j]);
    for (int j = head + 4 * n4 + tid; j < n_entries; j += NMS_THREADS) consider(j, lg[j]);
    __syncthreads();
    const uint32_t cnt = S->ncand;
    __syncthreads();
    return cnt;
}

// frames != nullptr: the kernel also writes the frame's 100 Detection rows (what wz_k_rows does from the det_* arrays) --
// one launch less per batch
// Clip-after-NMS programs (WzPostConsts::clip_after, oracle/postprocess.py: multiclass_nms_clip_after): the walk runs on the boxes as
// decoded and keeps up to NMS_KEEP_MAX of them -- a kept box that lies outside the image still suppresses its neighbours and
// uses a slot of its class, but is no row --, then the kept boxes are clipped and the first max_total that still have an area are
// the rows.  `status[f]` = 1 if the kept list filled up before max_total rows with an area were found (more than
// NMS_KEEP_MAX - max_total selected boxes entirely outside the image: the frame's rows may be short; wz_collect reports it).
// SELF / CLIP: the two run-time modes as template parameters (`self_scan`, WzPostConsts::clip_after): one copy of the band walk per
// kernel instead of four inlined ones -- the kernel lives at 128 registers (1024 threads) and spills; what is not there cannot.
template <bool SELF, bool CLIP, bool COUNT>
__global__ __launch_bounds__(NMS_THREADS) void wz_k_nms(WzPostBuffers b, WzPostConsts kc,
                                                        const WzFrameDesc* __restrict__ frames,
                                                        const WzCamFilter* __restrict__ cams,
                                                        wz_detection_t* __restrict__ rows, uint8_t* __restrict__ pass,
                                                        int self_scan, int listed, uint32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    NmsShared* S = reinterpret_cast<NmsShared*>(smem);
    const int f = blockIdx.x, tid = threadIdx.x;
    const int out_total = kc.max_total;
    WzPostConsts k = kc;
    if (CLIP) k.max_total = NMS_KEEP_MAX;                    // the walk's "enough" (and the kept list's capacity)
    const int A = k.num_anchors;
#define NMS_STAMP(i) do { if (tid == 0) b.dbg[(size_t)f * 16 + (i)] = wall_clock64(); } while (0)
    NMS_STAMP(0);
#if WZ_LANE_STAMPS
    // lane stamps (wz_common.h): every kernel in front of this one has finished -- frame 0's workgroup hands their entry / exit pairs
    // to the page-locked block the host reads and resets them for the lane's next batch; this kernel's own pairs (one per frame)
    // go straight to that block
    if (b.stamps_host) {
        if (tid == 0) b.stamps_host[2 * (b.stamps_n + f)] = wall_clock64();
        if (f == 0) {   // a wavefront per launch: earliest of the 64 entry words, latest of the 256 exit buckets
            const int sw = tid >> 6, sl = tid & 63;
            for (int q = sw; q < b.stamps_n; q += NMS_THREADS / 64) {
                unsigned long long* src = q == 0 ? b.stamps_pre : b.stamps + (size_t)(q - 1) * WZ_STAMP_WORDS;
                unsigned long long t0 = src[WZ_STAMP_ENTRY + sl], t1 = 0ull;
                src[WZ_STAMP_ENTRY + sl] = ~0ull;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long v = src[WZ_STAMP_EXIT + sl + 64 * j];
                    t1 = v > t1 ? v : t1;
                    src[WZ_STAMP_EXIT + sl + 64 * j] = 0ull;
                }
                for (int off = 32; off; off >>= 1) {
                    const unsigned long long o0 = __shfl_xor(t0, off), o1 = __shfl_xor(t1, off);
                    t0 = o0 < t0 ? o0 : t0;
                    t1 = o1 > t1 ? o1 : t1;
                }
                if (sl == 0) {
                    b.stamps_host[2 * q] = t0;
                    b.stamps_host[2 * q + 1] = t1;
                }
            }
        }
    }
#endif
    uint32_t processed = 0;
    int kept = 0;
    if (tid == 0) S->kept = 0;
    if constexpr (SELF) {
        // the bit map of the listed candidates is requested before anything else: with the hint, the logits and the
        // validity bytes behind it the first band is a chain of global-memory latencies (~2 us each on a busy chip)
        constexpr int WPT = 8;                               // words per thread (the launcher checks that this covers the map)
        uint32_t mw[WPT];
        {
            const int words = (k.num_anchors * k.num_classes + 31) >> 5;
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const int w = tid + u * NMS_THREADS;
                mw[u] = (listed && w < words) ? b.cbits[(size_t)f * words + w] : 0u;
            }
        }
        NMS_STAMP(1);
        int hi_bin = WZ_HIST_BINS;
        int lo_bin = min((int)b.hint[f], WZ_HIST_BINS - 1);
        int reach = 4;                                     // bins the next band extends below the current one
        uint32_t first_cnt = 0;
        int first_lo = lo_bin;
        bool first = true;
        for (;;) {
            uint32_t cnt;
            if (first && listed) {
                // the grouped head reduce marked every class logit >= wz_logit_floor(hint) of this frame while it still
                // had it in registers: the first band [hint, 1024) needs no scan of the logits, only of the bit map
                // (22 KiB per frame; read and cleared here).  Same tests as wz_nms_scan_band.
                if (tid == 0) { S->ncand = 0; S->pad[0] = 0; }
                __syncthreads();
                const int C = k.num_classes, n_entries = k.num_anchors * C, words = (n_entries + 31) >> 5;
                uint32_t* const bits = b.cbits + (size_t)f * words;   // (words <= WPT * NMS_THREADS: checked by the launcher)
                const float* __restrict__ lg = b.logits + (size_t)f * n_entries;
                // one listed entry: the tests of wz_nms_scan_band, the key into the band's list
                auto take = [&](int j) {
                    const int a = j / C, col = j - a * C;
                    const float x = lg[j];                      // both loads issued before either is used
                    const uint8_t ok = b.valid[(size_t)f * k.num_anchors + a];
                    if (col == 0 || !ok) return;
                    const float sc = wz_sigmoid(x);
                    if (!(sc > k.score_thr)) return;
                    const uint32_t key = __float_as_uint(sc);
                    const int bin = (int)(key >> 20);
                    if (bin < lo_bin || bin >= hi_bin) return;
                    const uint32_t tie = (uint32_t)(col - 1) * (uint32_t)k.num_anchors + (uint32_t)a;
                    const uint32_t pos = atomicAdd(&S->ncand, 1u);
                    if (pos < WZ_CAND_CAP)
                        S->keys[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - tie);
                };
                // Two passes since late round 6.  A thread that found several bits in its words used to walk them one after the other, a global round trip
                // (logit + validity byte) each -- the slowest thread's chain was 2 - 4 of them (6 - 8 us).  Now the set bits are first written as entry numbers
                // into LDS (the rank sort's destination buffer is idle here: 8 192 entries), then dealt out one per thread: ONE round trip for all of them.
                // Which thread takes which entry changes nothing: the band's list is sorted on unique keys before anything reads it.
                uint32_t* const elist = reinterpret_cast<uint32_t*>(S->keys2);
                constexpr uint32_t ELIST_CAP = 2 * WZ_CAND_CAP;
#pragma unroll
                for (int u = 0; u < WPT; ++u) {
                    const int w = tid + u * NMS_THREADS;
                    uint32_t m = mw[u];
                    if (m) bits[w] = 0u;
                    while (m) {
                        const int j = w * 32 + __builtin_ctz(m);
                        m &= m - 1;
                        const uint32_t q = atomicAdd(reinterpret_cast<uint32_t*>(&S->pad[0]), 1u);
                        if (q < ELIST_CAP) elist[q] = (uint32_t)j;
                        else take(j);                               // (more listed entries than the buffer holds: the old way, nothing is lost)
                    }
                }
                __syncthreads();
                const uint32_t nl = min((uint32_t)S->pad[0], ELIST_CAP);
                for (uint32_t q = tid; q < nl; q += NMS_THREADS) take((int)elist[q]);
                __syncthreads();
                cnt = S->ncand;
                __syncthreads();
            } else {
                cnt = wz_nms_scan_band(S, b, k, f, lo_bin, hi_bin);
            }
            while (cnt > WZ_CAND_CAP && lo_bin + 1 < hi_bin) {   // too many for the list: raise the band's lower edge
                lo_bin += (hi_bin - lo_bin + 1) >> 1;
                cnt = wz_nms_scan_band(S, b, k, f, lo_bin, hi_bin);
            }
            if (first) {
                first_cnt = cnt;
                first_lo = lo_bin;
                NMS_STAMP(2);
            }
            if (cnt > WZ_CAND_CAP)                           // one bin holds more than the list: exact serial walk
                kept = wz_nms_band_serial(S, b, k, f, kept, (unsigned long long)((uint32_t)lo_bin << 20) << 32,
                                          (unsigned long long)((uint32_t)hi_bin << 20) << 32);
            else if (cnt > 0)
                kept = wz_nms_band<CLIP, COUNT>(S, b, k, f, (int)cnt, kept);
            processed += cnt;
            if (first) { NMS_STAMP(3); if (tid == 0) { b.dbg[(size_t)f * 16 + 8] = cnt; b.dbg[(size_t)f * 16 + 9] = kept; } }
            first = false;
            if (kept >= k.max_total || lo_bin == 0) break;
            hi_bin = lo_bin;
            lo_bin = max(lo_bin - reach, 0);
            reach *= 2;
        }
        if (tid == 0) {   // where to start next time: a first band of roughly 200 .. 800 candidates
            int h = first_lo;
            if (first_cnt < WZ_CAND_TARGET) h = max(first_lo - 2, 1);
            else if (first_cnt > 4 * WZ_CAND_TARGET) h = min(first_lo + 1, WZ_HIST_BINS - 1);
            b.hint[f] = (uint32_t)h;
            b.hint_logit[f] = wz_logit_floor(h);
        }
    } else {
    // Bands of the score histogram, highest first.  Band 0 = bins [thr, 1024) was compacted by
    // wz_k_compact; further bands (needed only when NMS suppresses so much that band 0 runs dry
    // before max_total rows are kept) are collected here by one scan over the frame's candidates.
    const uint32_t total = b.band[2 * f + 1];
    int lo_bin = (int)b.band[2 * f];
    NMS_STAMP(1);
    int hi_bin = WZ_HIST_BINS;
    bool first = true;
    while (total > 0) {
        uint32_t cnt_raw;
        if (first) {
            cnt_raw = b.count[f];
            for (int i = tid; i < (int)min(cnt_raw, (uint32_t)WZ_CAND_CAP); i += NMS_THREADS) {
                const uint2 c = b.cand[(size_t)f * WZ_CAND_CAP + i];
                S->keys[i] = ((unsigned long long)c.x << 32) | (unsigned long long)(0xFFFFFFFFu - c.y);
            }
        } else {
            if (tid == 0) S->ncand = 0;
            __syncthreads();
            const int n_entries = A * k.num_classes;
            for (int j = tid; j < n_entries; j += NMS_THREADS) {
                uint32_t key, tie;
                if (wz_candidate(b, k, f, j, key, tie)) {
                    const int bin = (int)(key >> 20);
                    if (bin >= lo_bin && bin < hi_bin) {
                        const uint32_t pos = atomicAdd(&S->ncand, 1u);
                        if (pos < WZ_CAND_CAP)
                            S->keys[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - tie);
                    }
                }
            }
            __syncthreads();
            cnt_raw = S->ncand;
        }
        __syncthreads();
        if (first) NMS_STAMP(2);
        if (cnt_raw > WZ_CAND_CAP)
            kept = wz_nms_band_serial(S, b, k, f, kept, (unsigned long long)((uint32_t)lo_bin << 20) << 32,
                                      (unsigned long long)((uint32_t)hi_bin << 20) << 32);
        else if (cnt_raw > 0)
            kept = wz_nms_band<CLIP, COUNT>(S, b, k, f, (int)cnt_raw, kept);
        processed += cnt_raw;
        if (first) { NMS_STAMP(3); if (tid == 0) { b.dbg[(size_t)f * 16 + 8] = cnt_raw; b.dbg[(size_t)f * 16 + 9] = kept; } }
        if (kept >= k.max_total || processed >= total || lo_bin == 0) break;
        hi_bin = lo_bin;
        lo_bin = wz_threshold_bin(b.hist + (size_t)f * WZ_HIST_BINS, S->hist, WZ_CAND_TARGET, nullptr, hi_bin);
        first = false;
    }
    }
    __syncthreads();

    // clip-after programs: clip what the walk kept, drop what has no area left, close the ranks (band order is kept)
    const float4_t* obox = S->kbox;
    const float* oscore = S->kscore;
    const int32_t* ocls = S->kcls;
    uint32_t overflow = 0;
    if constexpr (CLIP) {
        for (int i = tid; i < NMS_KEEP_MAX; i += NMS_THREADS) {
            float4_t c = {0.f, 0.f, 0.f, 0.f};
            bool ok = false;
            if (i < kept) {
                const float4_t u = S->kbox[i];
                c = (float4_t){fminf(fmaxf(u[0], 0.0f), 1.0f), fminf(fmaxf(u[1], 0.0f), 1.0f), fminf(fmaxf(u[2], 0.0f), 1.0f),
                               fminf(fmaxf(u[3], 0.0f), 1.0f)};
                ok = (c[2] - c[0]) * (c[3] - c[1]) > 0.0f;
            }
            S->knorm[i] = c;
            S->cdead[i] = ok ? 0u : 1u;
        }
        __syncthreads();
        int total_ok = 0;
        for (int i = 0; i < kept; ++i) total_ok += S->cdead[i] ? 0 : 1;   // (<= 128 LDS broadcasts per thread)
        for (int i = tid; i < kept; i += NMS_THREADS) {
            if (S->cdead[i]) continue;
            int pos = 0;
            for (int j = 0; j < i; ++j) pos += S->cdead[j] ? 0 : 1;
            if (pos >= out_total) continue;
            S->cnorm[pos] = S->knorm[i];
            S->carea[pos] = S->kscore[i];
            S->ccls[pos] = S->kcls[i];
        }
        __syncthreads();
        overflow = (kept >= NMS_KEEP_MAX && total_ok < out_total) ? 1u : 0u;
        kept = min(total_ok, out_total);
        obox = S->cnorm;
        oscore = S->carea;
        ocls = S->ccls;
    }
    if (status && tid == 0) status[f] = overflow;
    // detection_boxes / scores / classes (+1 label offset on every row, zero padding included)
    for (int i = tid; i < out_total; i += NMS_THREADS) {
        const bool on = i < kept;
        const float4_t bx = on ? obox[i] : (float4_t){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<float4_t*>(b.det_boxes + ((size_t)f * out_total + i) * 4) = bx;
        b.det_scores[(size_t)f * out_total + i] = on ? oscore[i] : 0.0f;
        b.det_classes[(size_t)f * out_total + i] = (on ? ocls[i] : 0) + 1;
    }
    if (tid == 0) b.det_num[f] = kept;
    if (frames && tid < WZ_MAX_DETECTIONS) {
        const bool on = tid < kept;
        wz_make_row(frames[f], cams, tid < out_total, on ? obox[tid] : (float4_t){0.f, 0.f, 0.f, 0.f},
                    on ? oscore[tid] : 0.0f, (on ? ocls[tid] : 0) + 1,
                    rows + (size_t)f * WZ_MAX_DETECTIONS + tid, pass + (size_t)f * WZ_MAX_DETECTIONS + tid);
    }
    NMS_STAMP(4);
    if (tid == 0) b.dbg[(size_t)f * 16 + 10] = processed;
#if WZ_LANE_STAMPS
    if (b.stamps_host && tid == 0) b.stamps_host[2 * (b.stamps_n + f) + 1] = wall_clock64();
#endif
}

// ---------------------------------------------------------------------------------------------
// row fill + per-camera filters
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wz_zone_sum(const int32_t* sat, int W, int x0, int y0, int x1, int y1) {
    // inclusive pixel rectangle [x0..x1] x [y0..y1] on a (H+1) x (W+1) summed-area table
    const int s = W + 1;
    return sat[(y1 + 1) * s + (x1 + 1)] - sat[y0 * s + (x1 + 1)] - sat[(y1 + 1) * s + x0] + sat[y0 * s + x0];
}

__device__ void wz_apply_filter(const WzCamFilter& cf, wz_detection_t& d, uint8_t* pass) {
    // `label > 0 and Confidence and Area and Mask`, short-circuit left to right (track.py:26)
    bool ok = d.label > 0 && d.label < WZ_NUM_LABELS;
    if (ok) {
        const double ct = cf.conf_thr[d.label];            // confidence.py:17-19
        ok = (ct == ct) && d.confidence >= ct;
    }
    if (ok) {
        const double at = cf.area_thr[d.label];            // area.py:19-26
        long long ar = (long long)(d.x_max - d.x_min + 1) * (long long)(d.y_max - d.y_min + 1);
        if (ar < 0) ar = -ar;
        ok = (at == at) && (double)ar >= at;
    }
    if (ok && (cf.n_zones > 0 || (cf.enabled & 4))) {      // mask.py:44-59 (bit 2: a mask WITHOUT zones -- nothing can hit)
        // lattice rule: hit <=> a filled-zone pixel lies inside the closed box
        const int x0 = max(min(d.x_min, d.x_max), 0), x1 = min(max(d.x_min, d.x_max), cf.width - 1);
        const int y0 = max(min(d.y_min, d.y_max), 0), y1 = min(max(d.y_min, d.y_max), cf.height - 1);
        bool hit = false;
        int z = 0;
        const size_t plane = (size_t)(cf.width + 1) * (cf.height + 1);
        for (int p = 0; p < cf.n_zones && z < WZ_MAX_ZONES; ++p) {
            if (!cf.allow[d.label][p]) continue;
            if (x0 <= x1 && y0 <= y1 && wz_zone_sum(cf.sat + p * plane, cf.width, x0, y0, x1, y1) > 0) {
                d.zones[z++] = p + 1;
                hit = true;
            }
        }
        ok = hit;
    }
    if (!ok && (cf.enabled & 2)) {
        // drop mode (wz_set_camera_drop): the row leaves the GPU as an all-zero row, which `label > 0`
        // (track.py:26) rejects -- the sieve needs neither the pass byte nor the Python filters
        d.label = 0;
#pragma unroll
        for (int z = 0; z < WZ_MAX_ZONES; ++z) d.zones[z] = 0;
        d.confidence = 0.0;
        d.x_min = d.y_min = d.x_max = d.y_max = 0;
    }
    if (pass) *pass = ok ? 1 : 0;
}

__device__ void wz_make_row(const WzFrameDesc& fd, const WzCamFilter* __restrict__ cams, bool on, const float4_t bx,
                            float score, int label, wz_detection_t* __restrict__ row, uint8_t* __restrict__ pass) {
    wz_detection_t d;
    d.label = 0;
#pragma unroll
    for (int z = 0; z < WZ_MAX_ZONES; ++z) d.zones[z] = 0;
    d._pad = 0;
    d.confidence = 0.0;
    d.x_min = d.y_min = d.x_max = d.y_max = 0;
    if (on) {
        // tensorflow_cpu.py:79-90: label=int(class), confidence=score (float32 widened to double),
        // int(box * (dim-1)) with the product exact in double, truncation toward zero, no clamp
        const double mh = (double)(fd.h - 1), mw = (double)(fd.w - 1);
        d.label = label;
        d.confidence = (double)score;
        d.y_min = (int)((double)bx[0] * mh);
        d.x_min = (int)((double)bx[1] * mw);
        d.y_max = (int)((double)bx[2] * mh);
        d.x_max = (int)((double)bx[3] * mw);
    }
    uint8_t p = (d.label > 0) ? 1 : 0;
    if (fd.cam >= 0 && cams[fd.cam].enabled) wz_apply_filter(cams[fd.cam], d, &p);
    *row = d;
    *pass = p;
}

__global__ __launch_bounds__(128) void wz_k_rows(WzPostBuffers b, const WzFrameDesc* __restrict__ frames,
                                                 const WzCamFilter* __restrict__ cams, int max_total,
                                                 wz_detection_t* __restrict__ rows, uint8_t* __restrict__ pass) {
    const int f = blockIdx.x, i = threadIdx.x;
    if (i >= WZ_MAX_DETECTIONS) return;
    const bool on = i < max_total;
    const size_t src = (size_t)f * max_total + (on ? i : 0);
    wz_make_row(frames[f], cams, on, *reinterpret_cast<const float4_t*>(b.det_boxes + src * 4), b.det_scores[src],
                b.det_classes[src], rows + (size_t)f * WZ_MAX_DETECTIONS + i, pass + (size_t)f * WZ_MAX_DETECTIONS + i);
}

__global__ __launch_bounds__(128) void wz_k_filter_rows(const WzCamFilter* __restrict__ cams, int cam,
                                                        wz_detection_t* __restrict__ rows,
                                                        uint8_t* __restrict__ pass) {
    const int i = threadIdx.x;
    if (i >= WZ_MAX_DETECTIONS) return;
    wz_detection_t d = rows[i];
    uint8_t p = (d.label > 0) ? 1 : 0;
    if (cams[cam].enabled) wz_apply_filter(cams[cam], d, &p);
    rows[i] = d;
    pass[i] = p;
}

// summed-area tables of the filled-zone bitmaps: one workgroup per zone, two passes (rows, then columns)
__global__ __launch_bounds__(256) void wz_k_sat_rows(const uint8_t* __restrict__ fill, int32_t* __restrict__ sat,
                                                     int W, int H) {
    const int z = blockIdx.y;
    const int y = blockIdx.x * 256 + threadIdx.x;   // one thread per image row
    const size_t plane = (size_t)(W + 1) * (H + 1);
    int32_t* o = sat + z * plane;
    if (y == 0)
        for (int x = 0; x <= W; ++x) o[x] = 0;
    if (y >= H) return;
    const uint8_t* src = fill + ((size_t)z * H + y) * W;
    int32_t run = 0;
    o[(size_t)(y + 1) * (W + 1)] = 0;
    for (int x = 0; x < W; ++x) {
        run += src[x] ? 1 : 0;
        o[(size_t)(y + 1) * (W + 1) + x + 1] = run;
    }
}
__global__ __launch_bounds__(256) void wz_k_sat_cols(int32_t* __restrict__ sat, int W, int H) {
    const int z = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;   // one thread per column (coalesced across threads)
    if (x > W) return;
    int32_t* o = sat + z * (size_t)(W + 1) * (H + 1);
    int32_t run = 0;
    for (int y = 1; y <= H; ++y) {
        run += o[(size_t)y * (W + 1) + x];
        o[(size_t)y * (W + 1) + x] = run;
    }
}

// ---------------------------------------------------------------------------------------------
void wz_launch_decode(const WzPostBuffers& b, const WzPostConsts& c, int n, hipStream_t s) {
    const int total = n * c.num_anchors;
    hipLaunchKernelGGL(wz_k_decode, dim3((total + 255) / 256), dim3(256), 0, s, b, c, n);
}
void wz_launch_hist(const WzPostBuffers& b, const WzPostConsts& c, int n, hipStream_t s) {
    const int total = c.num_anchors * c.num_classes;
    dim3 grid((total + 256 * POST_ITEMS - 1) / (256 * POST_ITEMS), n);
    hipLaunchKernelGGL(wz_k_hist, grid, dim3(256), 0, s, b, c);
}
void wz_launch_compact(const WzPostBuffers& b, const WzPostConsts& c, int n, hipStream_t s) {
    const int total = c.num_anchors * c.num_classes;
    dim3 grid((total + 256 * COMPACT_ITEMS - 1) / (256 * COMPACT_ITEMS), n);
    hipLaunchKernelGGL(wz_k_compact, grid, dim3(256), 0, s, b, c);
}
void wz_post_init() {
#define WZ_NMS_ATTR(SELF, CLIP, COUNT) \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wz_k_nms<SELF, CLIP, COUNT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(NmsShared))
    WZ_NMS_ATTR(true, false, false); WZ_NMS_ATTR(true, false, true); WZ_NMS_ATTR(true, true, false); WZ_NMS_ATTR(true, true, true);
    WZ_NMS_ATTR(false, false, false); WZ_NMS_ATTR(false, false, true); WZ_NMS_ATTR(false, true, false); WZ_NMS_ATTR(false, true, true);
#undef WZ_NMS_ATTR
}
void wz_launch_nms(const WzPostBuffers& b, const WzPostConsts& c, int n, hipStream_t s, const WzFrameDesc* d_frames,
                   const WzCamFilter* d_cams, wz_detection_t* rows, uint8_t* pass, bool self_scan, bool listed, uint32_t* status) {
    if (((c.num_anchors * c.num_classes + 31) >> 5) > 8 * NMS_THREADS) listed = false;   // bit map larger than one pass: scan instead
    const int ls = (self_scan && listed) ? 1 : 0;
    const bool count = c.max_per_class < c.max_total;   // the per-class cap can bind: one thread per class walks its chain (wz_nms_band)
#define WZ_NMS_GO(SELF, CLIP, COUNT) \
    WZ_LAUNCH((wz_k_nms<SELF, CLIP, COUNT>), dim3(n), dim3(NMS_THREADS), sizeof(NmsShared), s, b, c, d_frames, d_cams, rows, pass, SELF ? 1 : 0, SELF ? ls : 0, status)
    if (self_scan) {
        if (!c.clip_after) { if (!count) WZ_NMS_GO(true, false, false); else WZ_NMS_GO(true, false, true); }
        else { if (!count) WZ_NMS_GO(true, true, false); else WZ_NMS_GO(true, true, true); }
    } else {
        if (!c.clip_after) { if (!count) WZ_NMS_GO(false, false, false); else WZ_NMS_GO(false, false, true); }
        else { if (!count) WZ_NMS_GO(false, true, false); else WZ_NMS_GO(false, true, true); }
    }
#undef WZ_NMS_GO
}
void wz_launch_rows(const WzPostBuffers& b, const WzFrameDesc* d_frames, const WzCamFilter* d_cams, int n,
                    int max_total, wz_detection_t* rows, uint8_t* pass, hipStream_t s) {
    hipLaunchKernelGGL(wz_k_rows, dim3(n), dim3(128), 0, s, b, d_frames, d_cams, max_total, rows, pass);
}
void wz_launch_filter_rows(const WzCamFilter* d_cams, int cam, wz_detection_t* rows, uint8_t* pass, hipStream_t s) {
    hipLaunchKernelGGL(wz_k_filter_rows, dim3(1), dim3(128), 0, s, d_cams, cam, rows, pass);
}
void wz_launch_sat(const uint8_t* fill, int32_t* sat, int width, int height, int n_zones, hipStream_t s) {
    hipLaunchKernelGGL(wz_k_sat_rows, dim3((height + 255) / 256, n_zones), dim3(256), 0, s, fill, sat, width, height);
    hipLaunchKernelGGL(wz_k_sat_cols, dim3((width + 1 + 255) / 256, n_zones), dim3(256), 0, s, sat, width, height);
}
