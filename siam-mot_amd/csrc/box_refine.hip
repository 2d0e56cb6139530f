// Box-head post-processing of the propagated tracks — the device half of CombinedROIHeads._refine_tracks.
//
// Replaces, for proposals that are ALL tracks (ids >= 0, labels given — what _refine_tracks passes,
// reference siammot/modelling/roi_heads.py:60-84): PostProcessor.forward + filter_results
// (siammot/modelling/box_head/inference.py:46-185: soft-max, BoxCoder.decode [UPSTREAM box_coder.py], "a track row
// keeps only its own label's probability + 1", clip_to_image, per-class threshold, grouping by label) and the score
// average of _refine_tracks (roi_heads.py:66-76), in ONE launch and without a host synchronisation.  The reference
// (and siammot_amd.box_refine.PostProcessor, the general path) runs ~25 tensor ops with a .nonzero() per class here.
//
// Why no row can be dropped on this path: a track row's score is p_label + 1 > 1 > score_thresh (the host wrapper
// refuses score_thresh >= 1), rows with ids >= 0 bypass the NMS (inference.py:165-174), clip_to_image is called with
// remove_empty=False.  The output therefore has exactly N rows: the rows grouped by label in ascending order, input
// order inside a label (inference.py:152-185) — rank = #(label_j < label_i) + #(j < i, label_j == label_i).
// The reference's quirk is kept: the box head's scores arrive in OUTPUT order, the matching scores are taken in
// INPUT order (roi_heads.py:66,70,75), so out_scores[p] = (det[p] + (track_conf[p] + 1)) / 2.
//
// One workgroup (N <= 512 rows, one thread per row); arithmetic in the reference's op order, fp32, separately rounded.
#include "smot_common.h"

namespace smot {

constexpr int BR_MAXN = 512;
constexpr int BR_MAXK = 16;       // classes incl. background held in registers per row

struct BoxRefineArgs {
    const float* logits;     // [N, ldo] : columns [0, K) class logits, columns [K, K + 4*KR) box deltas
    int ldo, K, KR;          // KR = K (per-class regression) or 2 (class-agnostic: the LAST four columns are used)
    const float* boxes;      // [N,4] proposals (the propagated boxes)
    const long long* labels; // [N]
    const long long* ids;    // [N]
    const float* track_conf; // [N] matching scores in [0,1] (the +1 band is applied here)
    float wx, wy, ww, wh, xform_clip;
    float clip_w, clip_h;    // image size, or 0: amodal (no clipping)
    int tracktor;
    float* out_boxes;        // [N,4]
    float* out_scores;       // [N]
    long long* out_ids;      // [N]
    long long* out_labels;   // [N]
    // the head's output still as the K-slice partial sums of linear_rows_partial_kernel (logits == nullptr): S slices of
    // [rows_pad][64] (one neuron block: ldo <= 64), summed here in slice order + bias — the arithmetic of
    // linear_rows_reduce_kernel, one launch fewer
    const float* part;
    int S, rows_pad;
    const float* cls_b;      // [K] or nullptr
    const float* reg_b;      // [4*KR] or nullptr
};
constexpr int BR_PART_ROWS = 128, BR_PART_COLS = 64;

template <bool PART>
__global__ void __launch_bounds__(BR_MAXN) box_refine_post_kernel(BoxRefineArgs A, int N) {
    __shared__ int slab[BR_MAXN];
    __shared__ float sdet[BR_MAXN];
    __shared__ float head[PART ? BR_PART_ROWS * BR_PART_COLS : 1];
    const int i = threadIdx.x;
    if constexpr (PART) {
        // thread = (row, column) of the head's output: S independent loads (one round trip), ordered adds, bias
        const size_t stride = (size_t)A.rows_pad * 64;                       // between K slices (one neuron block)
        for (int e = threadIdx.x; e < N * A.ldo; e += BR_MAXN) {
            const int m = e / A.ldo, n = e - m * A.ldo;
            const float* __restrict__ p = A.part + (size_t)m * 64 + n;
            float v = 0.0f;
            for (int s0 = 0; s0 < A.S; s0 += 32) {
                float t[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) t[q] = p[(size_t)min(s0 + q, A.S - 1) * stride];      // unconditional (clamped)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 32; ++q)
                    if (s0 + q < A.S) v = add_rn(v, t[q]);
            }
            const float* bsel = n >= A.K ? (A.reg_b != nullptr ? A.reg_b + (n - A.K) : nullptr)
                                         : (A.cls_b != nullptr ? A.cls_b + n : nullptr);
            if (bsel != nullptr) v = add_rn(v, *bsel);
            head[e] = v;
        }
        __syncthreads();
    }
    int lab = 0;
    float det = 0.0f;
    float bx1 = 0.f, by1 = 0.f, bx2 = 0.f, by2 = 0.f;
    long long id = -1;
    if (i < N) {
        lab = min(max((int)A.labels[i], 0), A.K - 1);        // (memory safety: a label outside [0, K) would index past the row)
        id = A.ids[i];
        const float* row = PART ? head + i * A.ldo : A.logits + (size_t)i * A.ldo;
        // F.softmax(class_logits, -1): max, exp of the difference, sum in class order, divide
        float m = row[0];
        for (int k = 1; k < A.K; ++k) m = fmaxf(m, row[k]);
        float s = 0.0f, el = 0.0f;
        for (int k = 0; k < A.K; ++k) {
            const float e = expf(sub_rn(row[k], m));
            s = add_rn(s, e);
            if (k == lab) el = e;
        }
        det = add_rn(div_rn(el, s), 1.0f);                                   // inference.py:102: prob + 1
        // BoxCoder.decode on the label's four deltas (class-agnostic: the last four columns, inference.py:67-68)
        const int r4 = (A.KR == A.K) ? 4 * lab : 4 * (A.KR - 1);
        const float* d = row + A.K + r4;
        const float x1 = A.boxes[i * 4 + 0], y1 = A.boxes[i * 4 + 1], x2 = A.boxes[i * 4 + 2], y2 = A.boxes[i * 4 + 3];
        const float w = add_rn(sub_rn(x2, x1), 1.0f), h = add_rn(sub_rn(y2, y1), 1.0f);
        const float cx = add_rn(x1, mul_rn(0.5f, w)), cy = add_rn(y1, mul_rn(0.5f, h));
        const float dx = div_rn(d[0], A.wx), dy = div_rn(d[1], A.wy);
        const float dw = fminf(div_rn(d[2], A.ww), A.xform_clip), dh = fminf(div_rn(d[3], A.wh), A.xform_clip);
        const float pcx = add_rn(mul_rn(dx, w), cx), pcy = add_rn(mul_rn(dy, h), cy);
        const float pw = mul_rn(expf(dw), w), ph = mul_rn(expf(dh), h);
        bx1 = sub_rn(pcx, mul_rn(0.5f, pw));
        by1 = sub_rn(pcy, mul_rn(0.5f, ph));
        bx2 = sub_rn(add_rn(pcx, mul_rn(0.5f, pw)), 1.0f);
        by2 = sub_rn(add_rn(pcy, mul_rn(0.5f, ph)), 1.0f);
        if (A.clip_w > 0.0f) {                                               // clip_to_image(remove_empty=False)
            bx1 = clamp_nan(bx1, 0.0f, A.clip_w - 1.0f);
            by1 = clamp_nan(by1, 0.0f, A.clip_h - 1.0f);
            bx2 = clamp_nan(bx2, 0.0f, A.clip_w - 1.0f);
            by2 = clamp_nan(by2, 0.0f, A.clip_h - 1.0f);
        }
        slab[i] = lab;
    }
    __syncthreads();
    int pos = 0;
    if (i < N) {
        for (int j = 0; j < N; ++j) {
            const int lj = slab[j];
            pos += (lj < lab || (lj == lab && j < i)) ? 1 : 0;
        }
        sdet[pos] = det;
        A.out_boxes[pos * 4 + 0] = bx1;
        A.out_boxes[pos * 4 + 1] = by1;
        A.out_boxes[pos * 4 + 2] = bx2;
        A.out_boxes[pos * 4 + 3] = by2;
        A.out_ids[pos] = id;
        A.out_labels[pos] = (long long)lab;
    }
    __syncthreads();
    if (i < N) {
        const float d = sdet[i];                                             // the box head's score of OUTPUT row i
        // roi_heads.py:66,75: (det_scores + (track_scores + 1)) / 2 — the matching score of INPUT row i
        A.out_scores[i] = A.tracktor ? d : div_rn(add_rn(d, add_rn(A.track_conf[i], 1.0f)), 2.0f);
    }
}

}  // namespace smot

extern "C" int smot_box_refine_post_max_rows(void) { return smot::BR_MAXN; }

extern "C" int smot_box_refine_post_fwd(const float* head_out, int ld, int num_classes, int reg_classes,
                                        const float* boxes, const int64_t* labels, const int64_t* ids,
                                        const float* track_conf, int N, float wx, float wy, float ww, float wh,
                                        float xform_clip, float clip_w, float clip_h, int tracktor, float* out_boxes,
                                        float* out_scores, int64_t* out_ids, int64_t* out_labels, smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(N >= 0 && N <= BR_MAXN, "box_refine_post: N=%d not in [0,%d]", N, BR_MAXN);
    SMOT_REQUIRE(num_classes >= 2 && (reg_classes == num_classes || reg_classes == 2) &&
                     ld >= num_classes + 4 * reg_classes,
                 "box_refine_post: bad head shape K=%d KR=%d ld=%d", num_classes, reg_classes, ld);
    SMOT_REQUIRE(wx > 0.f && wy > 0.f && ww > 0.f && wh > 0.f, "box_refine_post: regression weights must be positive");
    if (N == 0) return SMOT_OK;
    SMOT_REQUIRE(head_out && boxes && labels && ids && track_conf && out_boxes && out_scores && out_ids && out_labels,
                 "box_refine_post: null pointer");
    BoxRefineArgs A;
    A.logits = head_out;
    A.ldo = ld;
    A.K = num_classes;
    A.KR = reg_classes;
    A.boxes = boxes;
    A.labels = (const long long*)labels;
    A.ids = (const long long*)ids;
    A.track_conf = track_conf;
    A.wx = wx;
    A.wy = wy;
    A.ww = ww;
    A.wh = wh;
    A.xform_clip = xform_clip;
    A.clip_w = clip_w;
    A.clip_h = clip_h;
    A.tracktor = tracktor;
    A.out_boxes = out_boxes;
    A.out_scores = out_scores;
    A.out_ids = (long long*)out_ids;
    A.out_labels = (long long*)out_labels;
    A.part = nullptr;
    A.S = A.rows_pad = 0;
    A.cls_b = A.reg_b = nullptr;
    hipLaunchKernelGGL(box_refine_post_kernel<false>, dim3(1), dim3(BR_MAXN), 0, (hipStream_t)stream, A, N);
    return check_launch("box_refine_post");
}


// ---- the whole refinement of the propagated tracks behind ONE call ------------------------------------------------------
// HIP 7x7 pooler -> fc6 + ReLU -> fc7 + ReLU -> cls_score | bbox_pred -> post-processing: eight launches enqueued by one
// C-ABI crossing (the Python layer paid one ctypes call + one tensor allocation per stage).
#include "roi_common.h"
namespace smot {
int launch_roi_pool_separable(const LevelParams& P, int C, const float* rois, const float* level_boxes, int R,
                              int out_size, float* out, int32_t* levels_out, hipStream_t st);          // sr_xcorr.hip
int launch_linear_rows(const float* x, int M, int K, const float* W, const float* bias, int N, int relu, float* ws,
                       float* y, int ldy, hipStream_t st);                                              // linear_rows.hip
int launch_linear_rows2(const float* x, int M, int K, const float* W, const float* bias, int N1, const float* W2,
                        const float* bias2, int N2, int relu, float* ws, float* y, int ldy, hipStream_t st);
void linear_rows_layout(int M, int K, int N, int* S, int* nblk, int* rows_pad);
int launch_linear_rows_chain(const float* x, int M, int K, const float* WA, const float* bA, int NA, int reluA, float* ws_a,
                             const float* WB, int N1, const float* WB2, int N2, float* ws_b, hipStream_t st, int* rc);
}
extern "C" long long smot_linear_rows_ws_floats(int M, int K, int N);

static inline size_t br_align4(size_t n) { return (n + 3) & ~(size_t)3; }

extern "C" long long smot_box_refine_ws_floats(int N, int C, int pooled, int dim6, int dim7, int num_classes,
                                               int reg_classes) {
    if (N <= 0 || N > 128) return 0;
    const int K0 = C * pooled * pooled, NH = num_classes + 4 * reg_classes;
    long long g = smot_linear_rows_ws_floats(N, K0, dim6);
    const long long g7 = smot_linear_rows_ws_floats(N, dim6, dim7), gh = smot_linear_rows_ws_floats(N, dim7, num_classes + 4 * reg_classes);
    if (g7 > g) g = g7;
    // (the head's K-slice sums have a region of their own: the head reads fc7's slices while it writes its own)
    return (long long)(br_align4((size_t)N * K0) + br_align4((size_t)N * dim6) + br_align4((size_t)N * dim7) +
                       br_align4((size_t)N * NH)) + (long long)br_align4((size_t)g) + gh;
}

extern "C" int smot_box_refine_fwd(const float* const* feats, const int* heights, const int* widths, const float* scales,
                                   int num_levels, int C, int pooled, int sampling_ratio, const float* boxes,
                                   const int64_t* labels, const int64_t* ids, const float* track_conf, int N,
                                   const float* fc6_w, const float* fc6_b, int dim6, const float* fc7_w,
                                   const float* fc7_b, int dim7, const float* cls_w, const float* cls_b, int num_classes,
                                   const float* reg_w, const float* reg_b, int reg_classes, float wx, float wy, float ww,
                                   float wh, float xform_clip, float clip_w, float clip_h, int tracktor, float* ws,
                                   float* out_boxes, float* out_scores, int64_t* out_ids, int64_t* out_labels,
                                   smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(N >= 0 && N <= 128, "box_refine: N=%d not in [0,128] (use the stage-wise entries)", N);
    const int K0 = C * pooled * pooled;
    if (!((pooled == 7 || pooled == 15 || pooled == 30) && sampling_ratio == 2 && (K0 & 3) == 0 && (dim6 & 3) == 0 &&
          (dim7 & 3) == 0)) {
        set_error("box_refine: pooler %dx%d / sampling %d / layer widths %d,%d,%d not covered by the one-call path", pooled,
                  pooled, sampling_ratio, K0, dim6, dim7);
        return SMOT_ERR_UNSUPPORTED;
    }
    if (N == 0) return SMOT_OK;
    SMOT_REQUIRE(boxes && ws && fc6_w && fc7_w && cls_w && reg_w, "box_refine: null pointer");
    SMOT_REQUIRE(((uintptr_t)ws & 15) == 0, "box_refine: ws must be 16-byte aligned");
    LevelParams P;
    int rc = fill_level_params(&P, feats, heights, widths, nullptr, scales, num_levels, "box_refine");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int NH = num_classes + 4 * reg_classes;
    float* x0 = ws;
    float* h6 = x0 + br_align4((size_t)N * K0);
    float* h7 = h6 + br_align4((size_t)N * dim6);
    float* ho = h7 + br_align4((size_t)N * dim7);
    float* gw = ho + br_align4((size_t)N * NH);
    long long g67 = smot_linear_rows_ws_floats(N, K0, dim6);
    const long long g7 = smot_linear_rows_ws_floats(N, dim6, dim7);
    if (g7 > g67) g67 = g7;
    float* gh = gw + br_align4((size_t)g67);                                                  // the head's K-slice sums
    rc = launch_roi_pool_separable(P, C, boxes, boxes, N, pooled, x0, nullptr, st);          // Pooler: level of the roi itself
    if (rc) return rc;
    if ((rc = launch_linear_rows(x0, N, K0, fc6_w, fc6_b, dim6, 1, gw, h6, dim6, st))) return rc;
    // cls_score | bbox_pred side by side in one launch; their K-slice sums are added by the post-processing kernel while
    // it loads (heads of up to 64 columns: one neuron block), otherwise by the usual reduction launch — and fc7's sums
    // by the head's launch while IT loads (launch_linear_rows_chain): 6 launches instead of 8
    int S = 0, nblk = 0, rows_pad = 0;
    linear_rows_layout(N, dim7, NH, &S, &nblk, &rows_pad);
    const bool in_post = nblk == 1 && NH <= BR_PART_COLS && N <= BR_PART_ROWS;
    int chained = 0;
    if (in_post) {
        chained = launch_linear_rows_chain(h6, N, dim6, fc7_w, fc7_b, dim7, 1, gw, cls_w, num_classes, reg_w, 4 * reg_classes,
                                           gh, st, &rc);
        if (rc) return rc;
    }
    if (!chained) {
        if ((rc = launch_linear_rows(h6, N, dim6, fc7_w, fc7_b, dim7, 1, gw, h7, dim7, st))) return rc;
        if ((rc = launch_linear_rows2(h7, N, dim7, cls_w, cls_b, num_classes, reg_w, reg_b, 4 * reg_classes, 0, gh,
                                      in_post ? nullptr : ho, NH, st)))
            return rc;
    }
    if (!in_post)
        return smot_box_refine_post_fwd(ho, NH, num_classes, reg_classes, boxes, labels, ids, track_conf, N, wx, wy, ww,
                                        wh, xform_clip, clip_w, clip_h, tracktor, out_boxes, out_scores, out_ids,
                                        out_labels, stream);
    SMOT_REQUIRE(num_classes >= 2 && (reg_classes == num_classes || reg_classes == 2), "box_refine: bad head shape K=%d KR=%d",
                 num_classes, reg_classes);
    SMOT_REQUIRE(wx > 0.f && wy > 0.f && ww > 0.f && wh > 0.f, "box_refine: regression weights must be positive");
    SMOT_REQUIRE(labels && ids && track_conf && out_boxes && out_scores && out_ids && out_labels, "box_refine: null pointer");
    BoxRefineArgs A;
    A.logits = nullptr;
    A.ldo = NH;
    A.K = num_classes;
    A.KR = reg_classes;
    A.boxes = boxes;
    A.labels = (const long long*)labels;
    A.ids = (const long long*)ids;
    A.track_conf = track_conf;
    A.wx = wx;
    A.wy = wy;
    A.ww = ww;
    A.wh = wh;
    A.xform_clip = xform_clip;
    A.clip_w = clip_w;
    A.clip_h = clip_h;
    A.tracktor = tracktor;
    A.out_boxes = out_boxes;
    A.out_scores = out_scores;
    A.out_ids = (long long*)out_ids;
    A.out_labels = (long long*)out_labels;
    A.part = gh;
    A.S = S;
    A.rows_pad = rows_pad;
    A.cls_b = cls_b;
    A.reg_b = reg_b;
    hipLaunchKernelGGL(box_refine_post_kernel<true>, dim3(1), dim3(BR_MAXN), 0, st, A, N);
    return check_launch("box_refine_post");
}
