/*
 * smot_emm.h — C ABI of libsmot_emm.so, the MI355X (gfx950) implementation of the SiamMOT
 * EMM tracker-head hot path.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * amazon-science/siam-mot root; [UPSTREAM] = facebookresearch/maskrcnn-benchmark, the
 * reference's un-vendored native substrate).
 *
 * Conventions:
 *   - all tensors are contiguous fp32, NCHW; boxes are fp32 xyxy pixels; device pointers unless
 *     the parameter is documented as a HOST array;
 *   - the caller owns every buffer (outputs and workspaces included); nothing is allocated,
 *     nothing persists between calls, no host synchronisation happens inside;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: 0 on success, a negative SMOT_ERR_* for argument/shape errors, or a positive
 *     hipError_t if the launch failed; smot_last_error() gives a message for the calling thread.
 *     The Python host layer turns non-zero into RuntimeError (the reference's native ops raise
 *     RuntimeError through TORCH_CHECK).
 */
#ifndef SMOT_EMM_H
#define SMOT_EMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* smot_stream_t; /* hipStream_t */

#define SMOT_OK 0
#define SMOT_ERR_BAD_ARG (-1)      /* null pointer, non-positive size, inconsistent shapes */
#define SMOT_ERR_UNSUPPORTED (-2)  /* legal in the reference but not implemented here (documented per call) */

#define SMOT_MAX_LEVELS 8
/* ABI history.
 * 10: the image of smot_emm_tower_pack grew (fp32 image + three-part bf16 image: ask smot_emm_tower_pack_floats), an image
 *     packed by a version-9 library is too short for this one; smot_emm_tower_form added.  (9: order-hint entries of 528 floats)
 * 11: smot_frame_args grew by the six carry_* pointers and carry_src_row0 / carry_rows / carry_dst_row0 (SMOT_STAGE_CARRY);
 *     smot_track_solve_carry_fwd and smot_memory_carry_fwd added.
 * 12: order-hint entries of 536 floats — behind the tables every entry carries the by-roi record the consumer VERIFIES against
 *     its own boxes / search regions, and entry 0 the list's status word; the geometry stamp includes the level's scale;
 *     smot_emm_track_fwd writes NaN rows when the verification fails; record word 6 of the solver became a bit field;
 *     smot_emm_order_hint_status added.
 * 13: the second image of smot_emm_tower_pack is the TWO-part fp16 image (C/32 + 1 rotated blocks of 8192 floats per
 *     16-channel tile) followed by a four-word header {largest |w| bits, 2^-ku, 0, 0}: ask smot_emm_tower_pack_floats — an
 *     image packed by a version-12 library has another size and layout; smot_emm_tower_form returns 3 at every track count
 *     (one form); smot_emm_predictor_fwd may use the head of its `logits` output as scratch before it writes the logits;
 *     smot_sr_xcorr_fused_fwd / smot_emm_track_fwd correlate on the matrix cores (no signature change: responses equal
 *     smot_xcorr_dw_fwd's to rounding, not bit for bit).
 */
#define SMOT_ABI_VERSION 13

/* ABI version of the loaded library (checked by the host layer at load time). */
int smot_abi_version(void);

/* Build flavour of the loaded library.  bit 0 = measurement build (libsmot_emm_debug.so, compiled with
 * -DSMOT_DEBUG): kernel A/B switches and timing ablations exist and smot_debug_set_knob() is exported.  The
 * product library (libsmot_emm.so) returns 0: it reads no environment variable and contains no switch that
 * changes which kernel runs or what it computes. */
int smot_build_info(void);

#ifdef SMOT_DEBUG
/* Measurement library only: set one A/B or ablation switch by its SMOT_* name (csrc/knobs.h), value spelled as
 * the environment variable would be.  The environment itself is read once, when the library is loaded. */
int smot_debug_set_knob(const char* name, const char* value);
/* Measurement library only: smot_sr_xcorr_fused_fwd (Rx = 30, Rz = 15, sampling_ratio = 2) WITH an order hint (see
 * smot_emm_track_fwd below), for phase traces and A/B runs of that kernel alone. */
int smot_debug_sr_xcorr_fused_hint_fwd(const float* const* feats, const int* heights, const int* widths,
                                       const int* pad_cells, const float* scales, int num_levels, int C,
                                       const float* boxes, const float* sr, const float* templates, int N,
                                       float* resp, const float* order_hint, smot_stream_t stream);
#endif

/* Message describing the last non-zero return on the calling thread ("" if none). */
const char* smot_last_error(void);

/*
 * FPN-level-routed ROIAlign with VIRTUAL zero padding.
 *
 * Replaces: SRPooler.forward (siammot/modelling/track_head/EMM/sr_pool.py:53-91) =
 *   LevelMapper [UPSTREAM modeling/poolers.py] + 4x torch.nonzero + per-level
 *   ROIAlign.forward -> _C.roi_align_forward(input, rois, spatial_scale, pooled_h, pooled_w,
 *   sampling_ratio) [UPSTREAM csrc/cuda/ROIAlign_cuda.cu, layers/roi_align.py], AND the
 *   TrackUtils.pad_feature zero-padding (track_head/track_utils.py:87-107) that precedes it in
 *   EMM.forward (EMM/track_core.py:49): level l is treated as if padded by pad_cells[l] zero
 *   cells on every side, without materialising the padded map.
 *
 *   feats[l]      device [1, C, heights[l], widths[l]]   (HOST array of num_levels device pointers)
 *   heights, widths, pad_cells, scales : HOST arrays [num_levels]
 *   rois          device [R,4]  boxes that are pooled, in PADDED-image pixels (xyxy)
 *   level_boxes   device [R,4]  boxes whose area picks the level (the TEMPLATE boxes in the
 *                 reference; pass rois again for the template pooler / a plain Pooler)
 *   out           device [R, C, out_h, out_w]
 *   levels_out    device [R] int32 or NULL: the level chosen per roi (diagnostics/tests)
 *
 * num_levels == 1 skips the level mapping (sr_pool.py:70-71).
 * Level rule: floor(4 + log2(sqrt(area)/224 + 1e-6)) clamped to
 *   [-log2(scales[0]), -log2(scales[L-1])], area with the upstream +1 convention.
 * SMOT_ERR_UNSUPPORTED: sampling_ratio <= 0 (adaptive grid) or > 4.
 */
int smot_roi_align_levels_fwd(const float* const* feats, const int* heights, const int* widths,
                              const int* pad_cells, const float* scales, int num_levels, int C,
                              const float* rois, const float* level_boxes, int R,
                              int out_h, int out_w, int sampling_ratio,
                              float* out, int32_t* levels_out, smot_stream_t stream);

/*
 * Single-level ROIAlign with upstream's native signature (SURVEY.md §8(b) "Native boundary").
 *
 * Replaces: [UPSTREAM] maskrcnn_benchmark._C.roi_align_forward(Tensor input, Tensor rois, float spatial_scale,
 *   int pooled_h, int pooled_w, int sampling_ratio) -> Tensor, the call inside layers/roi_align.py::ROIAlign.forward
 *   that the reference reaches through SRPooler (EMM/sr_pool.py:28-31,89) and the box head's Pooler
 *   (box_head/box_head.py:46): legacy (non-"aligned") ROIAlign, sampling_ratio x sampling_ratio samples per bin.
 *
 *   input   device [num_images, C, H, W] contiguous
 *   rois5   device [R,5] = (image index, x1, y1, x2, y2), the tensor convert_to_roi_format builds (sr_pool.py:40-51);
 *           the image index is honoured (a row whose index is outside [0, num_images) pools to zeros — upstream reads
 *           out of bounds there)
 *   out     device [R, C, pooled_h, pooled_w], caller-allocated
 *   pad_cells  0 for upstream's behaviour; > 0 treats the map as zero-padded by that many cells on every side
 *           (TrackUtils.pad_feature, track_utils.py:87-107) with rois in padded coordinates, without the padded copy.
 * SMOT_ERR_UNSUPPORTED: sampling_ratio <= 0 (adaptive grid) or > 4.  The reference-side monkey patch is in
 * INTEGRATION.md §2.
 */
int smot_roi_align_fwd(const float* input, int num_images, int C, int H, int W, int pad_cells,
                       const float* rois5, int R, float spatial_scale, int pooled_h, int pooled_w,
                       int sampling_ratio, float* out, smot_stream_t stream);

/*
 * Search-region boxes from template boxes.
 *
 * Replaces: TrackUtils.update_boxes_in_pad_images + TrackUtils.extend_bbox
 *   (track_head/track_utils.py:109-135, :62-85) as called by EMM.extract_cache
 *   (EMM/track_core.py:94-95).  search_expansion = SEARCH_REGION - 1 (track_utils.py:260).
 *   boxes device [N,4] -> sr device [N,4] (padded-image coordinates).
 */
int smot_search_region_fwd(const float* boxes, int N, float pad_pixels, float search_expansion,
                           float min_search_wh, float* sr, smot_stream_t stream);

/*
 * Depthwise (per-track x per-channel) valid cross-correlation.
 *
 * Replaces: xcorr_depthwise(x, kernel) (EMM/xcorr.py:37-46; F.conv2d with groups = N*C).
 *   x [N,C,Rx,Rx], z [N,C,Rz,Rz] -> out [N,C,Ho,Ho], Ho = Rx-Rz+1,
 *   out[n,c,i,j] = sum_{u,v} x[n,c,i+u,j+v] * z[n,c,u,v]   (u-major, v-minor fp32 FMA chain).
 * Algorithmic HBM bytes: 4*N*C*(Rx^2 + Rz^2 + Ho^2).
 */
int smot_xcorr_dw_fwd(const float* x, const float* z, float* out, int N, int C, int Rx, int Rz,
                      smot_stream_t stream);

/*
 * Search-region pooling fused with the cross-correlation (the [N,C,Rx,Rx] tensor stays on chip).
 *
 * Replaces, in EMM.forward (EMM/track_core.py:49-53): pad_feature -> SRPooler(sr) -> xcorr_depthwise.
 *   Arguments as smot_roi_align_levels_fwd (rois = sr, level_boxes = boxes) and smot_xcorr_dw_fwd
 *   (z = templates [N,C,rz,rz]); resp [N,C,Ho,Ho].  x_debug: NULL, or a [N,C,rx,rx] buffer that
 *   receives the pooled planes (tests).  The pooling is separable (fp32-rounding-level differences to ROIAlign).  Since
 *   ABI 13 the correlation runs on the matrix cores (two-part fp16 operands of power-of-two-scaled planes, fp32
 *   accumulation): responses equal smot_xcorr_dw_fwd on the pooled planes to ROUNDING — within 6e-7 * sum |x||z| of an
 *   fp64 correlation on everything measured (worst 5.8e-7 over 1,950 random launches; the fp32 FMA chain of
 *   smot_xcorr_dw_fwd itself: 1.3e-6 on the same planes) — not bit for bit any more.  A pooled plane or a
 *   template that holds an inf / NaN gives an ALL-NaN response plane (smot_xcorr_dw_fwd: NaN / inf in the outputs whose
 *   window covers the value; behind the towers' GroupNorm both are a NaN track).
 * SMOT_ERR_UNSUPPORTED unless rx == 30, rz == 15, sampling_ratio == 2 (35 / 7: smot_sr_xcorr_gather_fwd; otherwise the two
 * unfused calls).
 */
int smot_sr_xcorr_fused_fwd(const float* const* feats, const int* heights, const int* widths,
                            const int* pad_cells, const float* scales, int num_levels, int C,
                            const float* boxes, const float* sr, const float* templates, int N,
                            int rx, int rz, int sampling_ratio,
                            float* resp, float* x_debug, smot_stream_t stream);

/*
 * The same for the reference's second yaml family (DLA_34_FPN_EMM_AOT.yaml: rx == 35, rz == 7, sampling_ratio == 2): the
 * generic ROIAlign kernel's gathers and the row-patch correlation in one kernel — responses bit-identical to
 * smot_roi_align_levels_fwd + smot_xcorr_dw_fwd; the [N,C,35,35] tensor stays on chip.  SMOT_ERR_UNSUPPORTED otherwise.
 */
int smot_sr_xcorr_gather_fwd(const float* const* feats, const int* heights, const int* widths,
                             const int* pad_cells, const float* scales, int num_levels, int C,
                             const float* boxes, const float* sr, const float* templates, int N, int rx, int rz,
                             int sampling_ratio, float* resp, smot_stream_t stream);

/*
 * EMM prediction tower + heads.
 *
 * Replaces: EMMPredictor.forward (EMM/feature_extractor.py:62-69): cls_tower / reg_tower =
 *   conv3x3(C->C, pad 1, no bias) + GroupNorm(gn_groups, eps, affine) + ReLU
 *   [UPSTREAM make_conv3x3(use_gn=True, use_relu=True)], then cls (C->2), center (C->1) on the
 *   cls tower and reg (C->4, followed by ReLU) on the reg tower, each conv3x3 + bias.
 *   Weight pointers are the reference state_dict tensors as they are (OIHW, contiguous):
 *   predictor.{cls_tower.0.weight, cls_tower.1.weight, cls_tower.1.bias, reg_tower.0.weight,
 *   reg_tower.1.weight, reg_tower.1.bias, cls.weight, cls.bias, center.weight, center.bias,
 *   reg.weight, reg.bias}.
 *
 *   resp      [N, C, Ho, Ho]
 *   tower_ws  [N, 2C, Ho, Ho] caller-owned workspace (cls tower in channels [0,C), reg in [C,2C))
 *   logits    [N, 7, Ho, Ho]  channel order: cls0, cls1, center, reg_l, reg_t, reg_r, reg_b
 * Requires C % gn_groups == 0.
 *
 *   tower_packed: optional (may be NULL) output of smot_emm_tower_pack for the SAME tower weights: the
 *   Winograd F(2x2,3x3)-transformed tower filters in the order the matrix-core kernel consumes them.
 *   With it (and Ho == 16) the towers run as a Winograd convolution on the matrix cores — 2.25x fewer
 *   multiplies, results equal to the direct convolution up to fp32 rounding order; without it the direct
 *   fp32 kernel runs.  For C % 32 == 0 (a power of two, <= 512) the transform-domain GEMMs run on fp16 matrix
 *   instructions with every fp32 operand as TWO fp16 parts of a power-of-two-scaled value (weights: scaled and
 *   split once into the second image; a track's response: scaled by 2^kv chosen from its planes' largest
 *   |response| and split in registers; three of the four part products, fp32 accumulation, the scales taken
 *   out exactly afterwards): the logits' error against an fp64 evaluation equals the fp32 form's, at every
 *   track count (one form: a track's logits do not depend on how many tracks the call has).  Other channel
 *   counts with C % 16 == 0 run the fp32 matrix instructions on the fp32 image.
 *   The plane maxima come from the pooling + correlation kernel inside smot_emm_track_fwd; a stand-alone call
 *   computes them with one small launch into the head of `logits` (N*C floats) before the logits are written.
 *   It is a pure function of the two tower weight tensors: recompute it whenever they change (the Python
 *   layer keys it on the tensors' versions).
 */
/* floats of the packed image: 2*C*C*16 (fp32 image; C % 16 == 0, else 0 = no packed path), plus for C % 32 == 0 the
 * two-part fp16 image, (2*C/16) * (C/32 + 1) * 8192, and its four-word header */
long long smot_emm_tower_pack_floats(int C);
/* which form of the packed towers N tracks get (reporting: bench.py, tools/): 0 = none (the direct kernel: C not a power
 * of two, C % 32 != 0, C > 512 or a response map other than 16 x 16 / 29 x 29), 1 = one 16-channel tile per workgroup
 * (fp32; measurement library only since ABI 13), 2 = two tiles (fp32), 3 = two tiles, two-part fp16 operands */
int smot_emm_tower_form(int N, int C, int Ho);
int smot_emm_tower_pack(const float* cls_tower_w, const float* reg_tower_w, int C, float* packed,
                        smot_stream_t stream);

int smot_emm_predictor_fwd(const float* resp, int N, int C, int Ho,
                           const float* cls_tower_w, const float* cls_gn_w, const float* cls_gn_b,
                           const float* reg_tower_w, const float* reg_gn_w, const float* reg_gn_b,
                           const float* cls_w, const float* cls_b,
                           const float* center_w, const float* center_b,
                           const float* reg_w, const float* reg_b,
                           int gn_groups, float gn_eps, const float* tower_packed,
                           float* tower_ws, float* logits, smot_stream_t stream);

/*
 * Fused bicubic x`up` up-sampling + location grid + response decode + argmax.
 *
 * Replaces, in EMM.forward (EMM/track_core.py:69-77): 3x F.interpolate(scale_factor=16,
 *   mode='bicubic'), get_locations (:184-225) and decode_response (:101-135, with
 *   get_scale_penalty :138-152 and get_cosine_window_penalty :155-162).  The up-sampled planes
 *   and the [N, G*G, 2] location tensor are never materialised.
 *
 *   logits  [N,7,Ho,Ho] (layout of smot_emm_predictor_fwd)
 *   sr      [N,4] search regions (padded-image coords),  boxes [N,4] template boxes
 *   hann    [G] the 1-D window, G = up*Ho (host layer passes torch.hann_window(G), periodic)
 *   cand_ws [N * smot_emm_decode_ws_floats(Ho, up)] fp32 workspace, 8-byte aligned
 *   bb [N,4], conf [N]; idx [N] int64 flat argmax index (y*G+x) or NULL
 *   rx, rz : search / template pooler resolutions (Ho must equal rx-rz+1; rz odd)
 *   one_minus_sigma and sigma are passed separately so the host can form 1-sigma in double as
 *   the reference does.
 *   clip_w, clip_h : image size for the clamp of wrap_results_to_boxlist (track_core.py:177-178:
 *   BoxList.clip_to_image, x in [0,w-1], y in [0,h-1]; the reference discards the filtered copy, so
 *   boxes are clamped, never removed); pass clip_w <= 0 for INPUT.AMODAL (no clamp).
 * NaN scores win the argmax and ties go to the lowest index (torch.argmax on CPU).
 */
int smot_emm_decode_fwd(const float* logits, const float* sr, const float* boxes, const float* hann,
                        int N, int Ho, int up, int rx, int rz, float pad_pixels,
                        float one_minus_sigma, float sigma, int use_centerness,
                        float clip_w, float clip_h,
                        float* cand_ws, float* bb, float* conf, int64_t* idx, smot_stream_t stream);

/* fp32 elements of decode workspace needed PER TRACK. */
int smot_emm_decode_ws_floats(int Ho, int up);

/*
 * Greedy IoU non-maximum suppression.
 *
 * Replaces [UPSTREAM] _C.nms(dets, scores, thresh) as reached through boxlist_nms from the solver
 *   (track_head/track_solver.py:22), the RPN post-processor (operator_patch/rpn_patch.py:53) and the box
 *   post-processor (box_head/inference.py:174).
 *   boxes_sorted [n,4] xyxy, ALREADY sorted by descending score (the host layer sorts, as upstream's wrapper
 *   does); IoU with the upstream +1 convention; box j is dropped when a kept earlier box i has
 *   IoU(i,j) > thresh.  keep [n] bytes (1 = kept), in the sorted order.  mask_ws: smot_nms_ws_bytes(n)
 *   bytes, 8-byte aligned.  n <= 8192.
 */
long long smot_nms_ws_bytes(int n);
int smot_nms_fwd(const float* boxes_sorted, int n, float thresh, void* mask_ws, unsigned char* keep,
                 smot_stream_t stream);

/*
 * Frame pre-processing: uint8 RGB HWC frame -> fp32 CHW network input in one launch.
 *
 * Replaces the CPU chain of demos/demo_inference.py:74-82 and build_augmentation.py:52-66 / image_augmentation.py:21-50:
 *   PIL.Image.resize((OW, OH), BILINEAR) -> ToTensor -> Normalize(mean, std, to_bgr255) [UPSTREAM transforms.py].
 *   Bit-exact with Pillow's 8-bit resampler: the caller supplies Pillow's per-axis tables (device memory),
 *   bounds [O,2] int32 = (first input index, tap count) and coeffs [O,k] int32 = weights quantised to 22
 *   fractional bits (an axis that keeps its size: (i,1) and 1<<22).  max_tile_rows = the largest number of
 *   input rows any 8 consecutive output rows touch.  mean3 / std3: HOST arrays, indexed by OUTPUT channel
 *   (after the optional BGR swap, as upstream applies them).  out [3, OH, OW].
 */
int smot_preprocess_fwd(const unsigned char* frame, int H, int W,
                        const int* xbounds, const int* xcoeffs, int kx,
                        const int* ybounds, const int* ycoeffs, int ky,
                        int OH, int OW, int max_tile_rows,
                        const float* mean3, const float* std3, int to_bgr255,
                        float* out, smot_stream_t stream);

/*
 * Instrumentation (bench.py roofline leg): between _begin and _end every smot_xcorr_dw_fwd and
 * smot_sr_xcorr_fused_fwd launch — direct or inside smot_emm_track_fwd — is issued through hipExtLaunchKernel with a
 * start / stop event pair of its own (events are created in _begin, outside any timed region; at most max_launches
 * pairs): the elapsed time between the two is the kernel's duration as rocprofv3 reports it (round 1 recorded two
 * marker events AROUND the launch: that span ran 2.5-3.5 us longer).  _end synchronises the events and returns the
 * summed kernel durations and the number of launches timed.  Not thread-safe; one timing session at a time.
 */
int smot_xcorr_timer_begin(int max_launches);
int smot_xcorr_timer_end(double* total_ms, int* launches);
/* Same mechanism per slot, timing every `stride`-th launch only: 0 = the cross-correlation kernels (what the two
 * calls above use, stride 1), 1 = the tower kernel of smot_emm_predictor_fwd / smot_emm_track_fwd. */
/* Phase trace: while buf != NULL every workgroup of the tower kernels, of the pooling / correlation kernels, of the
 * decode and of the solver kernel writes s_memtime stamps to buf[item*8 + 0..7] (device memory, 8 int64 per
 * workgroup of the launch grid; tower: start, main loop begin/end, output exchange done, GroupNorm done, end;
 * pooling: start, tables done, templates staged, pooling done, end, after the workgroup assignment).  NULL switches
 * it off. */
void smot_debug_trace(long long* buf);
int smot_kernel_timer_begin(int slot, int max_launches, int stride);
int smot_kernel_timer_end(int slot, double* total_ms, int* launches);
/* Median span (microseconds) of `reps` EMPTY marker-event brackets on `stream` (kept for tools that still time with
 * markers; bench.py no longer needs it). */
int smot_kernel_timer_bracket_overhead(smot_stream_t stream, int reps, double* median_us);
/* An empty kernel of workgroups x threads inside timer slot 0's event bracket (bench.py's
 * `roofline.empty_launch_event_bracket_us`): 4.1 us on MI355X whatever the grid — the BRACKET's floor, an upper bound on the
 * fixed cost of a dispatch (the packet's own timestamps under rocprofv3 give 0.8-1.5 us for the same empty kernel). */
int smot_dispatch_floor_fwd(int workgroups, int threads, smot_stream_t stream);

/*
 * One-call halves of a frame pair (same kernels, one FFI crossing each).
 *
 * smot_emm_track_fwd replaces the inference branch of EMM.forward (EMM/track_core.py:28-79):
 *   pad_feature (virtual) + SRPooler on the search regions -> xcorr_depthwise -> EMMPredictor ->
 *   bicubic x`up` + get_locations + decode_response -> clip of wrap_results_to_boxlist.
 *   boxes     [N,4] template boxes (pick the FPN level, scale penalty), sr [N,4] search regions
 *   templates [N,C,rz,rz] (track memory), predictor_params: HOST array of 13 device pointers: the 12 parameters in
 *   the order cls_tower.0.weight, cls_tower.1.weight, cls_tower.1.bias, reg_tower.0.weight,
 *   reg_tower.1.weight, reg_tower.1.bias, cls.weight, cls.bias, center.weight, center.bias,
 *   reg.weight, reg.bias, followed by a 13th entry: the smot_emm_tower_pack image of the tower weights
 *   or NULL (see smot_emm_predictor_fwd).   ws: smot_emm_track_ws_floats(N,C,rx,rz) floats, 16-byte aligned.
 *   Remaining arguments as in the per-operator calls above.
 *
 * smot_emm_extract_cache_fwd replaces EMM.extract_cache (EMM/track_core.py:81-98): template pooling
 *   on the un-padded maps (boxes pick level and roi) + search regions for the next frame.
 *
 * order_hint (optional, may be NULL everywhere): a scheduling side channel between the two halves, no part of the
 *   reference's interface.  A hint made from exactly the boxes / search regions it is passed with changes no result
 *   (bit-identical outputs with and without, tests/test_hip_parity.py); a hint made from OTHER boxes gives wrong
 *   results — see the last sentences.  The pooling + correlation kernel of
 *   smot_emm_track_fwd balances the chip by handing its workgroups the rois in cost order (wide search windows
 *   first); ranking them costs every workgroup ~3 us of start-up latency.  The extraction that CREATES those rois
 *   can rank them once: given a buffer of smot_emm_order_hint_floats(N, rz, sampling_ratio) floats (32-byte
 *   aligned; 0 = this shape / count writes no hint: pass NULL), smot_emm_extract_cache[_masked]_fwd writes
 *   SMOT_HINT_FLOATS dwords per roi — {search region x1,y1,x2,y2, FPN level (int32 bits), roi index (int32 bits),
 *   0, 0 | ymin, ymax, xmin, xmax of the touched window, pad cells / H / W / scale of the level the tables stand for |
 *   y sample table 64 x {row byte offset lo, hi, weight lo, hi} | x sample table 64 x {window column lo, hi, weight lo,
 *   hi} | by-roi record (8 dwords, see below)}, entry k = the roi of rank k; the tables (ABI 9) are the FINISHED sample tables of the roi's 30x30 search-region
 *   pooling in the next frame (zero-pad int(pad_pixels * scale) cells, same map sizes), so that a consumer workgroup
 *   copies them instead of building them (a geometry stamp that does not match the consumer's level makes it rebuild
 *   them) — and smot_emm_track_fwd given that buffer TOGETHER WITH exactly the `boxes`
 *   and `sr` of the same extraction (same N, same row order, same maps geometry) reads one entry per workgroup
 *   instead of ranking.  For the masked form the hint covers the first *n_valid rows.
 *   VERIFIED, not trusted (ABI 12): behind the tables, entry x carries the by-roi record of roi x — {its search region,
 *   its FPN level, the number of rois the list ranks, a status word, 0} — and the consuming kernel compares every record with
 *   ITS sr[x] (bit for bit), the level it derives from ITS boxes[x] and ITS N (one workgroup per roi, loads that travel
 *   beside the entry's own: no cost on the kernel's chain).  The list is a permutation of the rois it was made from by
 *   construction, so a list that passes describes exactly these rois.  One that does not (other boxes, another row
 *   order, another count or level geometry) raises the status word of entry 0 (SMOT_HINT_STATUS_WORD; the buffer is
 *   therefore IN/OUT for smot_emm_track_fwd), and the same call's decode kernel then writes NaN into every row of `bb`
 *   and `conf`: a stale hint is reported (NaN rows; smot_emm_order_hint_status; bit 1 of the solver's record word 6),
 *   never silently used.  Indices are clamped, so nothing is read out of range either way.
 */
#define SMOT_HINT_FLOATS 536
#define SMOT_HINT_STATUS_WORD 534   /* dword index (in the buffer, i.e. of entry 0) of the list's status word */
/* Copies the status word of an order hint to *status_host (pinned or pageable host memory) on `stream`: 0 = every head that
 * was given this hint found it describing its rois; non-zero = one did not (and wrote NaN rows).  Asynchronous: valid once
 * the stream has been synchronised. */
int smot_emm_order_hint_status(const float* order_hint, int* status_host, smot_stream_t stream);
long long smot_emm_order_hint_floats(int N, int rz, int sampling_ratio);

long long smot_emm_track_ws_floats(int N, int C, int rx, int rz);

int smot_emm_track_fwd(const float* const* feats, const int* heights, const int* widths,
                       const int* pad_cells, const float* scales, int num_levels, int C,
                       const float* boxes, const float* sr, const float* templates, int N,
                       int rx, int rz, int sampling_ratio,
                       const float* const* predictor_params, int gn_groups, float gn_eps,
                       const float* hann, int up, float pad_pixels,
                       float one_minus_sigma, float sigma, int use_centerness,
                       float clip_w, float clip_h,
                       float* ws, float* bb, float* conf, int64_t* idx, const float* order_hint,
                       smot_stream_t stream);

int smot_emm_extract_cache_fwd(const float* const* feats, const int* heights, const int* widths,
                               const float* scales, int num_levels, int C,
                               const float* boxes, int N, int rz, int sampling_ratio,
                               float pad_pixels, float search_expansion, float min_search_wh,
                               float* templates, float* sr, float* order_hint, smot_stream_t stream);

/* EMM.extract_cache over a CAPACITY of boxes of which the first *n_valid (device int32, e.g. &record[1] of
 * smot_track_solve_fwd) are real: rows >= *n_valid are skipped on the device, their outputs stay unwritten.  Lets
 * the tracker enqueue the template extraction before the host has read the solver's counts (Rz = 15 or 7,
 * sampling_ratio = 2 only). */
int smot_emm_extract_cache_masked_fwd(const float* const* feats, const int* heights, const int* widths,
                                      const float* scales, int num_levels, int C,
                                      const float* boxes, int capacity, const int* n_valid, int rz,
                                      int sampling_ratio, float pad_pixels, float search_expansion,
                                      float min_search_wh, float* templates, float* sr, float* order_hint,
                                      smot_stream_t stream);

/*
 * Box-head post-processing of the propagated tracks + the score average of _refine_tracks, one launch, no host sync.
 *
 * Replaces, when every proposal is a track (ids >= 0 and labels given — what CombinedROIHeads._refine_tracks passes to
 * the box head, siammot/modelling/roi_heads.py:60-84): PostProcessor.forward / filter_results
 * (siammot/modelling/box_head/inference.py:46-185: soft-max, BoxCoder.decode [UPSTREAM modeling/box_coder.py], track
 * rows keep their label's probability + 1, clip_to_image(remove_empty=False), per-class grouping) and
 * roi_heads.py:66-82 (score = (box-head score + matching score + 1) / 2, or the box-head score alone for TRACKTOR).
 *
 *   head_out   device [N, ld]: columns [0, K) = FPNPredictor.cls_score logits, [K, K + 4*KR) = bbox_pred deltas
 *              (KR = K, or 2 with CLS_AGNOSTIC_BBOX_REG: the last four columns are used, inference.py:67-68)
 *   boxes      device [N,4] xyxy proposals (the boxes EMM.forward propagated); labels / ids device [N] int64;
 *   track_conf device [N] matching scores in [0, 1] (the + 1 band is applied inside)
 *   wx..wh     BBOX_REG_WEIGHTS; xform_clip = log(1000/16); clip_w / clip_h = image size, 0 = INPUT.AMODAL
 *   out_*      device [N,4] / [N] / [N] int64 / [N] int64: exactly N rows, grouped by label in ascending order, input
 *              order inside a label (the order the reference's per-class loop produces); like the reference the
 *              matching scores are taken in INPUT order and the box-head scores in OUTPUT order (roi_heads.py:66,70).
 * Preconditions (checked by the host wrapper, not here): SCORE_THRESH < 1 (a track row scores p + 1 and is never
 * dropped), labels in [1, K).  N <= smot_box_refine_post_max_rows().
 */
int smot_box_refine_post_max_rows(void);
int smot_box_refine_post_fwd(const float* head_out, int ld, int num_classes, int reg_classes,
                             const float* boxes, const int64_t* labels, const int64_t* ids, const float* track_conf,
                             int N, float wx, float wy, float ww, float wh, float xform_clip,
                             float clip_w, float clip_h, int tracktor,
                             float* out_boxes, float* out_scores, int64_t* out_ids, int64_t* out_labels,
                             smot_stream_t stream);

/*
 * The whole box-head refinement of the propagated tracks behind one call (six launches, no host synchronisation: the
 * K-slice sums of fc7 and of cls_score | bbox_pred are added by their consumers while they load; eight launches when the
 * head is wider than 64 columns or fc7's width is not a multiple of 64).
 *
 * Replaces: CombinedROIHeads._refine_tracks (siammot/modelling/roi_heads.py:60-84) = ROIBoxHead.forward on the
 *   propagated boxes as proposals (box_head/box_head.py:46-50: [UPSTREAM] Pooler 7x7 -> fc6 -> ReLU -> fc7 -> ReLU ->
 *   cls_score / bbox_pred -> PostProcessor, box_head/inference.py:46-185) + the score average (roi_heads.py:66-82).
 *   feats / heights / widths / scales: the FPN levels as in smot_roi_align_levels_fwd (the level of a roi is picked by
 *   the roi itself, no padding); pooled = POOLER_RESOLUTION (7; 15 and 30 also run), sampling_ratio 2;
 *   fc6_w [dim6, C*pooled^2], fc7_w [dim7, dim6], cls_w [num_classes, dim7], reg_w [4*reg_classes, dim7] (+ biases) as
 *   the state_dict holds them; the remaining arguments as smot_box_refine_post_fwd.
 *   ws: device fp32 [smot_box_refine_ws_floats(...)], 16-byte aligned.   N <= 128.
 * SMOT_ERR_UNSUPPORTED: another pooler shape or layer widths that are not multiples of 4 (use the stage-wise entries).
 */
long long smot_box_refine_ws_floats(int N, int C, int pooled, int dim6, int dim7, int num_classes, int reg_classes);
int smot_box_refine_fwd(const float* const* feats, const int* heights, const int* widths, const float* scales,
                        int num_levels, int C, int pooled, int sampling_ratio, const float* boxes,
                        const int64_t* labels, const int64_t* ids, const float* track_conf, int N,
                        const float* fc6_w, const float* fc6_b, int dim6, const float* fc7_w, const float* fc7_b, int dim7,
                        const float* cls_w, const float* cls_b, int num_classes,
                        const float* reg_w, const float* reg_b, int reg_classes,
                        float wx, float wy, float ww, float wh, float xform_clip, float clip_w, float clip_h, int tracktor,
                        float* ws, float* out_boxes, float* out_scores, int64_t* out_ids, int64_t* out_labels,
                        smot_stream_t stream);

/*
 * y = act(x W^T + b) on a handful of rows: the box head's linear layers for the propagated tracks.
 *
 * Replaces: [UPSTREAM] FPN2MLPFeatureExtractor.fc6 / fc7 (+ ReLU) and FPNPredictor.cls_score / bbox_pred as
 *   ROIBoxHead.forward calls them (siammot/modelling/box_head/box_head.py:46-50) from
 *   CombinedROIHeads._refine_tracks (roi_heads.py:60-84), where x has one row per propagated track.
 *   x device [M, K] (M <= smot_linear_rows_max_rows(), K % 4 == 0, 16-byte aligned), W device [N, K] as
 *   torch.nn.Linear stores it, bias device [N] or NULL, relu 0/1, ws device fp32 [smot_linear_rows_ws_floats(M, K, N)]
 *   (16-byte aligned scratch for the split-K partial sums), y device [M, ldy] (ldy >= N: several layers can write
 *   column blocks of one row-major buffer).  Two launches (split-K partial products on the fp32 matrix cores, then the
 *   slice-ordered sum + bias + ReLU); deterministic.  SMOT_ERR_UNSUPPORTED: K % 4 != 0.
 */
int smot_linear_rows_max_rows(void);
long long smot_linear_rows_ws_floats(int M, int K, int N);
int smot_linear_rows_fwd(const float* x, int M, int K, const float* W, const float* bias, int N, int relu,
                         float* ws, float* y, int ldy, smot_stream_t stream);

/*
 * Track solver: one launch for a frame's TrackSolver.forward + pool transitions + active-row filter.
 *
 * Replaces TrackSolver.forward (siammot/modelling/track_head/track_solver.py:36-108: score banding of active
 * tracks :63-69, class-agnostic NMS at IoU 0.5 over detections + dormant + active boxes :22/:71 [UPSTREAM
 * boxlist_nms -> _C.nms], band removal :30-31, start / inactive / resume decisions :78-92, id writes :94-103),
 * the TrackPool transitions they trigger (track_head/track_utils.py:157-236: resume_track, start_track,
 * suspend_track, expire_tracks, increment_frame) and TrackHead._get_track_targets (track_head/track_head.py:99-110).
 * The reference synchronises with the host once per BOX here; this call never does.
 *
 *   det_* / trk_*  the frame's boxes in two segments (the detector's boxes; the boxes the tracker propagated —
 *                  either may be empty): boxes [n,4] xyxy, scores [n] (IN/OUT: banded in place as the reference
 *                  does), ids [n] int64 (-1 = no track), labels [n] int64 (may be NULL: 1).
 *                  trk_score_bias is added to the propagated scores first (the +1 of roi_heads.py:67 when no box
 *                  head refines them; 0 if they are already in the (1,2] band).
 *   pool_state     device int32 [8 + 3*pool_capacity], PERSISTENT between frames: word 0 max_id (-1 for a fresh
 *                  pool), word 1 frame_idx, word 2 n_active, word 3 n_dormant, word 4 the number of act_* rows the
 *                  last call wrote (A: a device-resident count, e.g. the n_valid of
 *                  smot_emm_extract_cache_masked_fwd), words 5-7 reserved (0); then the
 *                  active ids [pool_capacity], the dormant ids [pool_capacity] and the frame each dormant id was
 *                  last active in [pool_capacity].  Read and rewritten by every call.
 *   out_*          kept rows in ascending original order (boxes [M,4], scores back in [0,1], ids with new ids
 *                  started and inactive ones set to -1, labels); M = n_det + n_trk rows of capacity each.
 *   act_*          the rows of the output whose id is active after the update (the next frame's track targets).
 *   record         int32 [8 + 4*M + 3*pool_capacity], DEVICE memory or device-accessible pinned HOST memory (the
 *                  kernel's stores then land in host memory directly and the caller needs no copy, only an event
 *                  behind this launch): K (kept), A (active rows), max_id, frame_idx,
 *                  n_active, n_dormant, flags (bit 0: id table overflow; bit 1, ABI 12: a propagated track came in with a NaN
 *                  score — what a head writes whose order hint failed its verification, see order_hint; bit 2, ABI 12: no id
 *                  started, resumed, was suspended or expired in this frame — the three tables stand), M; kept
 *                  original row [M]; kept id [M]; active-row id
 *                  [M]; snapshot of the three pool tables.  The only thing the host has to read back.  frame_idx
 *                  (word 3, always >= 1) is stored last, behind a system-scope fence: a host polling a pinned
 *                  record it zeroed before the launch finds the record complete once word 3 is non-zero.
 * At most smot_track_solve_max_boxes() boxes / ids per call (one workgroup); more -> SMOT_ERR_UNSUPPORTED.
 */
int smot_track_solve_max_boxes(void);
int smot_track_solve_fwd(const float* det_boxes, float* det_scores, const int64_t* det_ids,
                         const int64_t* det_labels, int n_det,
                         const float* trk_boxes, float* trk_scores, const int64_t* trk_ids,
                         const int64_t* trk_labels, int n_trk, float trk_score_bias,
                         float track_thresh, float start_thresh, float resume_thresh, float nms_thresh,
                         int max_dormant_frames, int* pool_state, int pool_capacity,
                         float* out_boxes, float* out_scores, int64_t* out_ids, int64_t* out_labels,
                         float* act_boxes, int64_t* act_ids, int64_t* act_labels, float* act_scores,
                         int* record, smot_stream_t stream);

/*
 * The same launch carrying the dormant tracks' rows of the track memory (track_head/track_head.py:77-97; the stand-alone
 * form is smot_memory_carry_fwd, below): rows carry_src_row0 .. carry_src_row0 + carry_rows - 1 of the memory the frame's
 * head ran on (carry_* arrays: templates [*, row_floats], boxes / sr [*,4], ids / labels int64, scores) go behind the rows
 * this launch leaves active.  Boxes, ids, labels and scores are appended to act_* behind the count the launch determines
 * (always the right place).  Templates and search regions are copied by extra workgroups beside the solver's one — they
 * add nothing to the launch's duration but cannot know that count: they write to rows carry_dst_row0 .. of next_templates /
 * next_sr (capacity n_det + n_trk rows; the masked template extraction fills rows 0 .. count-1 of the same buffers
 * afterwards), the CALLER'S GUESS of the count; a caller whose guess turns out wrong (record word 1 != carry_dst_row0)
 * copies again with smot_memory_carry_fwd.  row_floats % 4 == 0, 16-byte aligned arrays.  carry_rows == 0: plain
 * smot_track_solve_fwd.
 */
int smot_track_solve_carry_fwd(const float* det_boxes, float* det_scores, const int64_t* det_ids,
                               const int64_t* det_labels, int n_det,
                               const float* trk_boxes, float* trk_scores, const int64_t* trk_ids,
                               const int64_t* trk_labels, int n_trk, float trk_score_bias,
                               float track_thresh, float start_thresh, float resume_thresh, float nms_thresh,
                               int max_dormant_frames, int* pool_state, int pool_capacity,
                               float* out_boxes, float* out_scores, int64_t* out_ids, int64_t* out_labels,
                               float* act_boxes, int64_t* act_ids, int64_t* act_labels, float* act_scores,
                               int* record,
                               const float* carry_templates, const float* carry_boxes, const float* carry_sr,
                               const int64_t* carry_ids, const int64_t* carry_labels, const float* carry_scores,
                               int carry_src_row0, int carry_rows, int carry_dst_row0,
                               float* next_templates, float* next_sr, int row_floats, smot_stream_t stream);

/*
 * One tracking frame behind ONE call (or two): the launches of smot_emm_track_fwd (3) [+ smot_box_refine_fwd (8)] +
 * smot_track_solve_fwd (1) + smot_emm_extract_cache_masked_fwd (1) enqueued back to back on `stream`.
 *
 * Replaces the inference branch of CombinedROIHeads.forward from `self.track(...)` on
 * (siammot/modelling/roi_heads.py:38-50): TrackHead.forward_inference -> EMM.forward (track_head.py:37-46),
 * _refine_tracks (roi_heads.py:60-84), cat_boxlist + TrackSolver.forward (track_solver.py:36-108),
 * TrackHead.get_track_memory -> EMM.extract_cache on the rows the solver leaves active (track_head.py:54-75,99-110).
 * Every field is an argument of one of the four entry points above and means what it means there; the outputs of a
 * stage are the inputs of the next (the propagated boxes / scores feed the refinement or the solver, the solver's
 * act_boxes and pool_state[4] feed the masked template extraction).  n_trk == 0 skips the head and the refinement
 * (first frame, or an empty memory); refine == 0 skips the refinement (the solver then applies trk_score_bias = 1).
 * SMOT_STAGE_CARRY (with SMOT_STAGE_SOLVE; never part of stages == 0) makes the solver's launch carry the dormant rows
 * of the track memory the head ran on (the carry_* fields: smot_track_solve_carry_fwd; next_templates / next_sr are the
 * destination buffers, row_floats = C * rz * rz).  ORDERING CONTRACT of the carried templates / search regions: they are
 * written to rows [carry_dst_row0, carry_dst_row0 + carry_rows) of next_templates / next_sr BEFORE the active count A is
 * known; when carry_dst_row0 < A the masked extraction of the same frame (SMOT_STAGE_EXTRACT, enqueued behind the solver
 * on the same stream) overwrites rows 0 .. A-1 afterwards and never writes rows >= A — so a wrong guess leaves garbage
 * only in rows the caller re-copies (record word 1 != carry_dst_row0 -> smot_memory_carry_fwd).  A caller that runs the
 * two launches on different streams, or the extraction first, must not use SMOT_STAGE_CARRY.
 * `stages` selects what this call enqueues (0 = everything): a frame is a serial chain — host work before the first
 * launch, the kernels, the record, host bookkeeping — so a caller whose argument preparation is not free calls once
 * with SMOT_STAGE_HEAD as soon as the head's arguments stand (fields of later stages are not read) and a second time
 * with the remaining stages while the head runs (siammot_amd.track_head.TrackingLoop does).
 * The host reads `record` (pinned host memory) as with smot_track_solve_fwd.  Plain C struct: pointers first, then
 * 32-bit fields, no padding; inside each group the fields that change from frame to frame are contiguous (a binding
 * that keeps the block between frames rewrites two short ranges per call).
 */
#define SMOT_STAGE_HEAD 1
#define SMOT_STAGE_REFINE 2
#define SMOT_STAGE_SOLVE 4
#define SMOT_STAGE_EXTRACT 8
#define SMOT_STAGE_CARRY 16     /* with SMOT_STAGE_SOLVE: smot_track_solve_carry_fwd on the carry_* fields */
typedef struct smot_frame_args {
    /* ---- fixed while the video's geometry and the model stand ---- */
    /* FPN levels (HOST arrays of num_levels entries, as in smot_roi_align_levels_fwd; the entries of `feats` are
     * this frame's maps — the array itself can stay where it is) */
    const float* const* feats;
    const int* heights;
    const int* widths;
    const int* pad_cells;        /* search-region pooling (virtual padding) */
    const float* scales;
    const float* const* predictor_params;   /* HOST array of 13 device pointers (smot_emm_track_fwd) */
    const float* hann;
    const float* fc6_w; const float* fc6_b; const float* fc7_w; const float* fc7_b;   /* box head (refine != 0) */
    const float* cls_w; const float* cls_b; const float* reg_w; const float* reg_b;
    int* pool_state;
    /* ---- this frame: the head (SMOT_STAGE_HEAD reads up to trk_conf) ---- */
    float* head_ws;              /* smot_emm_track_ws_floats(n_trk, C, rx, rz) */
    const float* tpl_boxes;      /* [n_trk,4] track memory of the previous frame */
    const float* sr;             /* [n_trk,4] */
    const float* templates;      /* [n_trk,C,rz,rz] */
    const float* order_hint;     /* order hint of exactly these rows (smot_emm_track_fwd) or NULL */
    const int64_t* trk_ids;      /* [n_trk] */
    const int64_t* trk_labels;   /* [n_trk] */
    float* trk_boxes;            /* [n_trk,4] out: propagated boxes */
    float* trk_conf;             /* [n_trk]   out: matching scores */
    /* ---- this frame: refinement, detections, solver, next memory ---- */
    float* refine_ws;            /* smot_box_refine_ws_floats(...) */
    float* ref_boxes;            /* [n_trk,4] out */
    float* ref_scores;           /* [n_trk]   out */
    int64_t* ref_ids;            /* [n_trk]   out */
    int64_t* ref_labels;         /* [n_trk]   out */
    const float* det_boxes; float* det_scores; const int64_t* det_ids; const int64_t* det_labels;
    float* out_boxes; float* out_scores; int64_t* out_ids; int64_t* out_labels;
    float* act_boxes; int64_t* act_ids; int64_t* act_labels; float* act_scores;
    int* record;
    /* next frame's memory (capacity n_det + n_trk rows; the first pool_state[4] are written) */
    float* next_templates;
    float* next_sr;
    float* next_order_hint;      /* smot_emm_order_hint_floats(n_det + n_trk, rz, sampling_ratio) floats or NULL */
    /* dormant rows carried by the solver's launch (SMOT_STAGE_CARRY; smot_track_solve_carry_fwd) */
    const float* carry_templates; const float* carry_boxes; const float* carry_sr;
    const int64_t* carry_ids; const int64_t* carry_labels; const float* carry_scores;
    /* ---- sizes and scalars: per frame first ---- */
    int n_trk, stages, n_det;
    int num_levels, C;
    int rx, rz, sampling_ratio, gn_groups, up, use_centerness;
    int refine, box_pooled, box_sampling_ratio, dim6, dim7, num_classes, reg_classes, tracktor;
    int max_dormant_frames, pool_capacity;
    int carry_src_row0, carry_rows, carry_dst_row0;
    float track_thresh, start_thresh, resume_thresh;
    float gn_eps, pad_pixels, one_minus_sigma, sigma, clip_w, clip_h;
    float box_wx, box_wy, box_ww, box_wh, box_xform_clip;
    float nms_thresh, search_expansion, min_search_wh;
} smot_frame_args;

int smot_track_frame_fwd(const smot_frame_args* args, smot_stream_t stream);

/*
 * Dormant rows of the track memory, carried on the device.
 *
 * Replaces: TrackHead._update_memory_with_dormant_track (track_head/track_head.py:77-97: torch.cat of the templates +
 *   two cat_boxlist calls per frame).  A dormant track's (template, search region, box, id, label, score) never changes
 *   while it is dormant and is a row of the memory the frame's head just ran on: D rows `rows[j]` (host array, indices
 *   into the source memory of `src_rows_total` rows) are copied to rows dst_row0 + j of the destination buffers
 *   (`dst_capacity` rows each; templates [*, row_floats], boxes / sr [*,4], ids / labels int64 [*], scores [*]).
 *   dst_row0_dev != NULL: the first destination row is read from that device word instead (the solver's pool state
 *   word 4 = the number of active rows: a launch enqueued before the host has read the frame's record); rows at or
 *   beyond dst_capacity are then dropped.  D <= smot_memory_carry_max_rows().  Pure copies (bit-exact).
 */
int smot_memory_carry_max_rows(void);
int smot_memory_carry_fwd(const float* src_templates, const float* src_boxes, const float* src_sr,
                          const int64_t* src_ids, const int64_t* src_labels, const float* src_scores,
                          int src_rows_total, float* dst_templates, float* dst_boxes, float* dst_sr,
                          int64_t* dst_ids, int64_t* dst_labels, float* dst_scores, int dst_capacity,
                          const int* rows, int D, int dst_row0, const int* dst_row0_dev, int row_floats,
                          smot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SMOT_EMM_H */
