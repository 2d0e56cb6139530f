// One-call entry points for the two halves of a frame pair (host-side fusion: one FFI crossing and
// one argument validation per call instead of five; the kernels are the ones in the other files and
// run back-to-back on `stream`).
//
//   smot_emm_track_fwd         == the inference branch of EMM.forward          (reference EMM/track_core.py:28-79)
//   smot_emm_extract_cache_fwd == EMM.extract_cache                             (reference EMM/track_core.py:81-98)
#include "smot_common.h"
#include "logit_src.h"
#include "knobs.h"

namespace smot {
int predictor_impl(const float* resp, int N, int C, int Ho, const float* cls_tower_w, const float* cls_gn_w,
                   const float* cls_gn_b, const float* reg_tower_w, const float* reg_gn_w, const float* reg_gn_b,
                   const float* cls_w, const float* cls_b, const float* center_w, const float* center_b,
                   const float* reg_w, const float* reg_b, int gn_groups, float gn_eps,
                   const float* tower_packed, float* tower_ws, float* logits, smot_stream_t stream, int* tiles_out,
                   unsigned* zero_words, bool* zeroed, const float* plane_max);
int decode_impl(LogitSrc L, const float* sr, const float* boxes, const float* hann, int N, int Ho, int up, int rx,
                int rz, float pad_pixels, float one_minus_sigma, float sigma, int use_centerness, float clip_w,
                float clip_h, float* cand_ws, float* bb, float* conf, int64_t* idx, bool tickets_zeroed,
                hipStream_t st, const int* poison);
unsigned* decode_tickets(float* cand_ws, int N, int Ho);
int launch_extract_cache(const float* const* feats, const int* heights, const int* widths, const float* scales,
                         int num_levels, int C, const float* boxes, int N, int rz, float pad_pixels, float half_e,
                         float two_e, float min_wh, float* templates, float* sr, const int* n_valid, float* order_hint,
                         hipStream_t st, int hint_extra_rows);
int sr_xcorr_fused_impl(const float* const* feats, const int* heights, const int* widths, const int* pad_cells,
                        const float* scales, int num_levels, int C, const float* boxes, const float* sr,
                        const float* templates, int N, float* resp, float* x_debug, const float* order_hint,
                        hipStream_t st, const int** hint_status, float* plane_max);
int sr_xcorr_gather_impl(const float* const* feats, const int* heights, const int* widths, const int* pad_cells,
                         const float* scales, int num_levels, int C, const float* boxes, const float* sr,
                         const float* templates, int N, int rx, int rz, int sampling_ratio, float* resp, hipStream_t st,
                         float* plane_max);
}  // namespace smot

static inline bool p12_form3(int N, int C, int ho) { return smot_emm_tower_form(N, C, ho) == 3; }

extern "C" long long smot_emm_track_ws_floats(int N, int C, int rx, int rz) {
    if (N < 0 || C <= 0 || rz <= 0 || rx < rz) return -1;
    const long long ho = rx - rz + 1;
    const long long n = N > 0 ? N : 1;
    // x | response | tower workspace | logits | decode candidates (8-byte aligned: all terms even)
    return n * C * rx * rx + n * C * ho * ho + n * 2 * C * ho * ho + n * 8 * ho * ho +
           n * smot_emm_decode_ws_floats((int)ho, 16);
}

extern "C" int smot_emm_track_fwd(const float* const* feats, const int* heights, const int* widths,
                                  const int* pad_cells, const float* scales, int num_levels, int C,
                                  const float* boxes, const float* sr, const float* templates, int N, int rx,
                                  int rz, int sampling_ratio, const float* const* predictor_params, int gn_groups,
                                  float gn_eps, const float* hann, int up, float pad_pixels,
                                  float one_minus_sigma, float sigma, int use_centerness, float clip_w,
                                  float clip_h, float* ws, float* bb, float* conf, int64_t* idx,
                                  const float* order_hint, smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(N >= 0 && C > 0 && rz > 0 && rx >= rz, "emm_track: bad sizes N=%d C=%d rx=%d rz=%d", N, C, rx, rz);
    if (N == 0) return SMOT_OK;
    SMOT_REQUIRE(predictor_params && ws && boxes && sr && templates, "emm_track: null pointer");
    SMOT_REQUIRE(((uintptr_t)ws & 15) == 0, "emm_track: workspace must be 16-byte aligned");
    const int ho = rx - rz + 1;
    float* x = ws;
    float* resp = x + (size_t)N * C * rx * rx;
    float* tower = resp + (size_t)N * C * ho * ho;
    float* logits = tower + (size_t)N * 2 * C * ho * ho;
    float* cand = logits + (size_t)N * 8 * ho * ho;       // 7 planes used; 8 keeps the 8-byte alignment
    int rc;
    // the response's plane maxima [N][C] for the tower kernel's split form: in the head of the logits buffer, which nothing
    // reads before the towers and the decode kernel writes after them; nullptr = the predictor makes them itself
    const float* plane_max = nullptr;
    const int* poison = nullptr;
    const bool no_fuse = knobs().no_fuse;             // A/B: measurement library only (constant false otherwise)
    if (rx == 30 && rz == 15 && sampling_ratio == 2 && !no_fuse) {
        // pooling feeds the correlation inside one kernel: the search-region tensor never reaches HBM
        // (a hint the kernel honours is VERIFIED against `boxes` / `sr` by it; `poison` = the list's status word)
        float* pm = (C <= 7 * ho * ho && p12_form3(N, C, ho)) ? logits : nullptr;
        rc = sr_xcorr_fused_impl(feats, heights, widths, pad_cells, scales, num_levels, C, boxes, sr, templates, N, resp,
                                 nullptr, order_hint, (hipStream_t)stream, &poison, pm);
        if (rc) return rc;
        plane_max = pm;
    } else if (rx == 35 && rz == 7 && sampling_ratio == 2 && !no_fuse) {
        // the second yaml family's shape: gathers + correlation in one kernel (sr_xcorr_small.hip), same arithmetic
        float* pm = (C <= 7 * ho * ho && p12_form3(N, C, ho)) ? logits : nullptr;
        rc = sr_xcorr_gather_impl(feats, heights, widths, pad_cells, scales, num_levels, C, boxes, sr, templates, N, rx, rz,
                                  sampling_ratio, resp, (hipStream_t)stream, pm);
        if (rc) return rc;
        plane_max = pm;
    } else {
        rc = smot_roi_align_levels_fwd(feats, heights, widths, pad_cells, scales, num_levels, C, sr, boxes, N, rx, rx,
                                       sampling_ratio, x, nullptr, stream);
        if (rc) return rc;
        rc = smot_xcorr_dw_fwd(x, templates, resp, N, C, rx, rz, stream);
        if (rc) return rc;
    }
    const float* const* p = predictor_params;
    // towers only: when the matrix-core path applies the per-tile partial head sums stay in `tower` and the
    // decode kernels sum them while loading (no combine launch, no logits round trip)
    int tiles = 0;
    bool tickets_zeroed = false;           // the Winograd tower kernel zeroes the decode kernel's tickets on its way
    rc = predictor_impl(resp, N, C, ho, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11],
                        gn_groups, gn_eps, p[12], tower, logits, stream, &tiles, decode_tickets(cand, N, ho),
                        &tickets_zeroed, plane_max);
    if (rc) return rc;
    LogitSrc L;
    L.logits = tiles > 0 ? nullptr : logits;
    L.part = tower;
    L.tpt = tiles;
    L.cls_b = p[7];
    L.center_b = p[9];
    L.reg_b = p[11];
    L.logits_out = logits;
    return decode_impl(L, sr, boxes, hann, N, ho, up, rx, rz, pad_pixels, one_minus_sigma, sigma, use_centerness,
                       clip_w, clip_h, cand, bb, conf, idx, tickets_zeroed, (hipStream_t)stream, poison);
}

extern "C" int smot_emm_extract_cache_fwd(const float* const* feats, const int* heights, const int* widths,
                                          const float* scales, int num_levels, int C, const float* boxes, int N,
                                          int rz, int sampling_ratio, float pad_pixels, float search_expansion,
                                          float min_search_wh, float* templates, float* sr, float* order_hint,
                                          smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(N >= 0 && num_levels >= 1 && num_levels <= SMOT_MAX_LEVELS, "emm_extract_cache: bad sizes");
    if (N == 0) return SMOT_OK;
    if ((rz == 15 || rz == 7) && sampling_ratio == 2 && !knobs().roi_generic) {
        // one launch: separable template pooling, with the search regions written by the same kernel
        const float half_e = (float)((double)search_expansion / 2.0);
        const float two_e = (float)((double)search_expansion * 2.0);
        return launch_extract_cache(feats, heights, widths, scales, num_levels, C, boxes, N, rz, pad_pixels, half_e,
                                    two_e, min_search_wh, templates, sr, nullptr, order_hint, (hipStream_t)stream, 0);
    }
    int zero_pad[SMOT_MAX_LEVELS] = {0};
    int rc = smot_roi_align_levels_fwd(feats, heights, widths, zero_pad, scales, num_levels, C, boxes, boxes, N, rz, rz,
                                       sampling_ratio, templates, nullptr, stream);
    if (rc) return rc;
    return smot_search_region_fwd(boxes, N, pad_pixels, search_expansion, min_search_wh, sr, stream);
}

// EMM.extract_cache over a CAPACITY of boxes of which only the first *n_valid (a device-resident count, e.g. word 1
// of smot_track_solve_fwd's record) are real: the launch is enqueued before the host knows the count, workgroups
// of the other rows return at once and their output rows stay unwritten.  The tracker's frame then needs no host
// synchronisation between the solver and the template extraction.
extern "C" int smot_emm_extract_cache_masked_fwd(const float* const* feats, const int* heights, const int* widths,
                                                 const float* scales, int num_levels, int C, const float* boxes,
                                                 int capacity, const int* n_valid, int rz, int sampling_ratio,
                                                 float pad_pixels, float search_expansion, float min_search_wh,
                                                 float* templates, float* sr, float* order_hint,
                                                 smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(capacity >= 0 && num_levels >= 1 && num_levels <= SMOT_MAX_LEVELS, "emm_extract_cache_masked: bad sizes");
    SMOT_REQUIRE(n_valid != nullptr, "emm_extract_cache_masked: null count pointer");
    if (capacity == 0) return SMOT_OK;
    if (!((rz == 15 || rz == 7) && sampling_ratio == 2)) {
        set_error("emm_extract_cache_masked: only Rz=15 or 7, sampling_ratio=2 (got %d, %d); use smot_emm_extract_cache_fwd "
                  "with the count on the host", rz, sampling_ratio);
        return SMOT_ERR_UNSUPPORTED;
    }
    const float half_e = (float)((double)search_expansion / 2.0);
    const float two_e = (float)((double)search_expansion * 2.0);
    return launch_extract_cache(feats, heights, widths, scales, num_levels, C, boxes, capacity, rz, pad_pixels, half_e,
                                two_e, min_search_wh, templates, sr, n_valid, order_hint, (hipStream_t)stream, 0);
}


// One tracking frame behind one call (include/smot_emm.h, smot_frame_args): the four stages' existing entry points in
// sequence on one stream — no new arithmetic, one FFI crossing instead of four and no host work between the launches.
extern "C" int smot_track_frame_fwd(const smot_frame_args* a, smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(a != nullptr, "track_frame: null argument block");
    const int stages = a->stages == 0 ? (SMOT_STAGE_HEAD | SMOT_STAGE_REFINE | SMOT_STAGE_SOLVE | SMOT_STAGE_EXTRACT) : a->stages;
    SMOT_REQUIRE(a->n_trk >= 0, "track_frame: n_trk=%d", a->n_trk);
    int rc;
    if ((stages & SMOT_STAGE_HEAD) && a->n_trk > 0) {
        rc = smot_emm_track_fwd(a->feats, a->heights, a->widths, a->pad_cells, a->scales, a->num_levels, a->C, a->tpl_boxes,
                                a->sr, a->templates, a->n_trk, a->rx, a->rz, a->sampling_ratio, a->predictor_params,
                                a->gn_groups, a->gn_eps, a->hann, a->up, a->pad_pixels, a->one_minus_sigma, a->sigma,
                                a->use_centerness, a->clip_w, a->clip_h, a->head_ws, a->trk_boxes, a->trk_conf, nullptr,
                                a->order_hint, stream);
        if (rc) return rc;
    }
    if (!(stages & (SMOT_STAGE_REFINE | SMOT_STAGE_SOLVE | SMOT_STAGE_EXTRACT))) return SMOT_OK;
    SMOT_REQUIRE(a->n_det >= 0 && a->n_trk + a->n_det > 0, "track_frame: n_trk=%d n_det=%d", a->n_trk, a->n_det);
    // what the solver reads as its track segment: the head's output, or the refinement's (refine != 0)
    const float* trk_boxes = nullptr;
    float* trk_scores = nullptr;
    const int64_t* trk_ids = nullptr;
    const int64_t* trk_labels = nullptr;
    float bias = 1.0f;
    if (a->n_trk > 0) {
        trk_boxes = a->trk_boxes;
        trk_scores = a->trk_conf;
        trk_ids = a->trk_ids;
        trk_labels = a->trk_labels;
        if (a->refine) {
            if (stages & SMOT_STAGE_REFINE) {
                rc = smot_box_refine_fwd(a->feats, a->heights, a->widths, a->scales, a->num_levels, a->C, a->box_pooled,
                                         a->box_sampling_ratio, a->trk_boxes, a->trk_labels, a->trk_ids, a->trk_conf,
                                         a->n_trk, a->fc6_w, a->fc6_b, a->dim6, a->fc7_w, a->fc7_b, a->dim7, a->cls_w,
                                         a->cls_b, a->num_classes, a->reg_w, a->reg_b, a->reg_classes, a->box_wx,
                                         a->box_wy, a->box_ww, a->box_wh, a->box_xform_clip, a->clip_w, a->clip_h,
                                         a->tracktor, a->refine_ws, a->ref_boxes, a->ref_scores, a->ref_ids,
                                         a->ref_labels, stream);
                if (rc) return rc;
            }
            trk_boxes = a->ref_boxes;
            trk_scores = a->ref_scores;
            trk_ids = a->ref_ids;
            trk_labels = a->ref_labels;
            bias = 0.0f;                       // the refined scores are in the (1, 2] band already
        }
    }
    if (stages & SMOT_STAGE_SOLVE) {
        const int carry = (stages & SMOT_STAGE_CARRY) ? a->carry_rows : 0;
        rc = smot_track_solve_carry_fwd(a->det_boxes, a->det_scores, a->det_ids, a->det_labels, a->n_det, trk_boxes, trk_scores,
                                        trk_ids, trk_labels, a->n_trk, bias, a->track_thresh, a->start_thresh,
                                        a->resume_thresh, a->nms_thresh, a->max_dormant_frames, a->pool_state,
                                        a->pool_capacity, a->out_boxes, a->out_scores, a->out_ids, a->out_labels,
                                        a->act_boxes, a->act_ids, a->act_labels, a->act_scores, a->record,
                                        a->carry_templates, a->carry_boxes, a->carry_sr, a->carry_ids, a->carry_labels,
                                        a->carry_scores, a->carry_src_row0, carry, a->carry_dst_row0, a->next_templates,
                                        a->next_sr, a->C * a->rz * a->rz, stream);
        if (rc) return rc;
    }
    if (!(stages & SMOT_STAGE_EXTRACT)) return SMOT_OK;
    // = smot_emm_extract_cache_masked_fwd, whose order hint also ranks the dormant rows the solver's launch has just put
    // behind the active ones (SMOT_STAGE_CARRY: their boxes stand in act_boxes behind the count): a memory with dormant
    // tracks — most frames of a MOT sequence — keeps its hint
    SMOT_REQUIRE(a->n_det + a->n_trk >= 0 && a->num_levels >= 1 && a->num_levels <= SMOT_MAX_LEVELS, "track_frame: bad sizes");
    if (a->n_det + a->n_trk == 0) return SMOT_OK;
    if (!((a->rz == 15 || a->rz == 7) && a->sampling_ratio == 2)) {
        set_error("track_frame: the masked extraction needs Rz=15 or 7, sampling_ratio=2 (got %d, %d)", a->rz, a->sampling_ratio);
        return SMOT_ERR_UNSUPPORTED;
    }
    return launch_extract_cache(a->feats, a->heights, a->widths, a->scales, a->num_levels, a->C, a->act_boxes,
                                a->n_det + a->n_trk, a->rz, a->pad_pixels, (float)((double)a->search_expansion / 2.0),
                                (float)((double)a->search_expansion * 2.0), a->min_search_wh, a->next_templates, a->next_sr,
                                a->pool_state + 4, a->next_order_hint, (hipStream_t)stream,
                                (stages & SMOT_STAGE_CARRY) ? a->carry_rows : 0);
}
