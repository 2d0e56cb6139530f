// K1 — FPN-level-routed ROIAlign with virtual zero padding, and the search-region geometry.
//
// Replaces (see include/smot_emm.h): SRPooler.forward (reference EMM/sr_pool.py:53-91) incl.
// LevelMapper and ROIAlign [UPSTREAM maskrcnn_benchmark csrc/cuda/ROIAlign_cuda.cu], the
// TrackUtils.pad_feature pass (track_head/track_utils.py:87-107) and
// update_boxes_in_pad_images + extend_bbox (track_utils.py:62-85,109-135).
//
// Work decomposition: one workgroup per (roi, group of CH_PER_BLOCK channels).  The per-axis
// sample bookkeeping of the legacy ROIAlign (validity, clamp, low/high cell, bilinear weights —
// all channel-independent) is computed once per workgroup into LDS tables; cells that fall in
// the virtual zero border get weight 0 and a safe index, so the inner loop is branch-free.
// The bounding window of the cells a roi really touches (<= ~58x58 cells when the level matches
// the box; found with two LDS atomics per table entry) is staged per channel into LDS with
// coalesced row reads — every feature cell is fetched from HBM/L2 once instead of being gathered
// 16x per output; co-resident workgroups (2-3 per CU) overlap one's staging with another's pooling.
// Lanes walk the flattened (ph,pw) bin index (tables of one bin in registers, channels inner), so
// stores are fully coalesced.  Windows that exceed the LDS budget
// (degenerate aspect ratios) fall back to direct gathers from the map, same arithmetic.
#include "roi_common.h"
#include "knobs.h"

namespace smot {

constexpr int RA_WIN_FLOATS = 4096;   // LDS window budget per channel (16 KiB)
constexpr int RA_CH = 4;              // channels per workgroup (windows staged together: 64 KiB)

template <int G>
__global__ void __launch_bounds__(256)
roi_align_levels_kernel(LevelParams P, int C, const float* __restrict__ rois,
                        const float* __restrict__ level_boxes, int PH, int PW, int ch_per_block,
                        float* __restrict__ out, int32_t* __restrict__ levels_out, int num_images) {
    // num_images == 0: rois are [R,4] on the one image of the call (the level-routed pooler).
    // num_images >= 1: rois are [R,5] = (image index, x1, y1, x2, y2) as upstream's _C.roi_align_forward takes them,
    //                  P.feat[0] is [num_images, C, H, W]; a row whose index is out of range pools to zeros.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int wbound[4];              // ymin, ymax, xmin, xmax of the cells with non-zero weight
    const int ny = PH * G, nx = PW * G;
    float* win = reinterpret_cast<float*>(smem);
    int* y_lo = reinterpret_cast<int*>(win + RA_CH * RA_WIN_FLOATS);
    int* y_hi = y_lo + ny;
    float* wy_lo = reinterpret_cast<float*>(y_hi + ny);
    float* wy_hi = wy_lo + ny;
    int* x_lo = reinterpret_cast<int*>(wy_hi + ny);
    int* x_hi = x_lo + nx;
    float* wx_lo = reinterpret_cast<float*>(x_hi + nx);
    float* wx_hi = wx_lo + nx;

    const int r = blockIdx.x;
    const float* roi = num_images ? rois + (size_t)r * 5 + 1 : rois + (size_t)r * 4;
    const int image = num_images ? (int)rois[(size_t)r * 5] : 0;
    int lvl = 0;
    if (P.num_levels > 1) lvl = map_level(level_boxes + (size_t)r * 4, P.k_min, P.k_max);
    if (levels_out != nullptr && blockIdx.y == 0 && threadIdx.x == 0) levels_out[r] = lvl;

    const int H = P.H[lvl], W = P.W[lvl], pad = P.pad[lvl];
    const float scale = P.scale[lvl];
    const float x1 = mul_rn(roi[0], scale), y1 = mul_rn(roi[1], scale);
    const float x2 = mul_rn(roi[2], scale), y2 = mul_rn(roi[3], scale);
    const float roi_w = fmaxf(sub_rn(x2, x1), 1.0f);
    const float roi_h = fmaxf(sub_rn(y2, y1), 1.0f);
    const float bin_h = div_rn(roi_h, (float)PH);
    const float bin_w = div_rn(roi_w, (float)PW);

    if (threadIdx.x == 0) {
        wbound[0] = 0x7fffffff;
        wbound[1] = -1;
        wbound[2] = 0x7fffffff;
        wbound[3] = -1;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < ny + nx; s += blockDim.x) {
        int lo, hi;
        float wl, wh;
        if (s < ny) {
            axis_sample(y1, bin_h, G, s, H, pad, &lo, &hi, &wl, &wh);
            y_lo[s] = lo;
            y_hi[s] = hi;
            wy_lo[s] = wl;
            wy_hi[s] = wh;
        } else {
            const int sx = s - ny;
            axis_sample(x1, bin_w, G, sx, W, pad, &lo, &hi, &wl, &wh);
            x_lo[sx] = lo;
            x_hi[sx] = hi;
            wx_lo[sx] = wl;
            wx_hi[sx] = wh;
        }
        const int b = (s < ny) ? 0 : 2;
        if (wl != 0.0f) {
            atomicMin(&wbound[b], lo);
            atomicMax(&wbound[b + 1], lo);
        }
        if (wh != 0.0f) {
            atomicMin(&wbound[b], hi);
            atomicMax(&wbound[b + 1], hi);
        }
    }
    __syncthreads();
    const int ymin = wbound[0], ymax = wbound[1], xmin = wbound[2], xmax = wbound[3];
    const int c0 = blockIdx.y * ch_per_block;
    const int c1 = min(C, c0 + ch_per_block);
    const int bins = PH * PW;
    const float* __restrict__ f = P.feat[lvl] + (size_t)image * C * H * W;
    if (ymax < ymin || xmax < xmin || image < 0 || image >= max(num_images, 1)) {
        // every sample lies in the virtual zero border (or outside the padded map): exact zeros
        for (int c = c0; c < c1; ++c)
            for (int t = threadIdx.x; t < bins; t += blockDim.x) out[((size_t)r * C + c) * bins + t] = 0.0f;
        return;
    }
    const int wh_ = ymax - ymin + 1, ww = xmax - xmin + 1;
    const bool staged = (wh_ * ww <= RA_WIN_FLOATS);     // workgroup-uniform
    // re-base the tables: window-relative when staged, map-relative row offsets otherwise
    __syncthreads();
    for (int s = threadIdx.x; s < ny + nx; s += blockDim.x) {
        if (s < ny) {
            const int lo = (wy_lo[s] != 0.0f) ? y_lo[s] : ymin;
            const int hi = (wy_hi[s] != 0.0f) ? y_hi[s] : ymin;
            y_lo[s] = staged ? (lo - ymin) * ww : lo * W;
            y_hi[s] = staged ? (hi - ymin) * ww : hi * W;
        } else {
            const int sx = s - ny;
            const int lo = (wx_lo[sx] != 0.0f) ? x_lo[sx] : xmin;
            const int hi = (wx_hi[sx] != 0.0f) ? x_hi[sx] : xmin;
            x_lo[sx] = staged ? lo - xmin : lo;
            x_hi[sx] = staged ? hi - xmin : hi;
        }
    }
    __syncthreads();

    const int nch = c1 - c0;
    if (staged) {
        // ---- stage the (wh x ww) windows of all channels of this workgroup.  Wave w takes rows
        // w, w+4, ...; lanes take columns (contiguous, coalesced row segments).  All loads of a pass
        // (up to 16 rows x RA_CH channels per lane) are issued before the first LDS store, so one
        // memory round trip covers the whole pass; co-resident workgroups cover the rest. ----
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
        for (int col0 = 0; col0 < ww; col0 += 64) {
            const int col = col0 + tx;
            for (int row0 = 0; row0 < wh_; row0 += 64) {
                float tmp[RA_CH][16];
#pragma unroll
                for (int cl = 0; cl < RA_CH; ++cl) {
                    const float* __restrict__ fc = f + (size_t)(c0 + min(cl, nch - 1)) * H * W;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int row = row0 + ty + 4 * k;
                        tmp[cl][k] = (row < wh_ && col < ww) ? fc[(ymin + row) * W + xmin + col] : 0.0f;
                    }
                }
#pragma unroll
                for (int cl = 0; cl < RA_CH; ++cl) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int row = row0 + ty + 4 * k;
                        if (row < wh_ && col < ww) win[cl * RA_WIN_FLOATS + row * ww + col] = tmp[cl][k];
                    }
                }
            }
        }
        __syncthreads();
    }
    // bins outer (tables of one bin in registers), channels inner
    for (int t = threadIdx.x; t < bins; t += 256) {
        const int ph = t / PW;
        const int pw = t - ph * PW;
        int ylo[G], yhi[G], xlo[G], xhi[G];
        float wyl[G], wyh[G], wxl[G], wxh[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            ylo[i] = y_lo[ph * G + i];
            yhi[i] = y_hi[ph * G + i];
            wyl[i] = wy_lo[ph * G + i];
            wyh[i] = wy_hi[ph * G + i];
            xlo[i] = x_lo[pw * G + i];
            xhi[i] = x_hi[pw * G + i];
            wxl[i] = wx_lo[pw * G + i];
            wxh[i] = wx_hi[pw * G + i];
        }
        for (int cl = 0; cl < nch; ++cl) {
            // window larger than the LDS budget (degenerate aspect ratios): gather straight from the map
            const float* __restrict__ src = staged ? (const float*)(win + cl * RA_WIN_FLOATS)
                                                   : f + (size_t)(c0 + cl) * H * W;
            float acc = 0.0f;
#pragma unroll
            for (int iy = 0; iy < G; ++iy) {
#pragma unroll
                for (int ix = 0; ix < G; ++ix) {
                    const float v1 = src[ylo[iy] + xlo[ix]];
                    const float v2 = src[ylo[iy] + xhi[ix]];
                    const float v3 = src[yhi[iy] + xlo[ix]];
                    const float v4 = src[yhi[iy] + xhi[ix]];
                    const float w1 = wyl[iy] * wxl[ix], w2 = wyl[iy] * wxh[ix];
                    const float w3 = wyh[iy] * wxl[ix], w4 = wyh[iy] * wxh[ix];
                    acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                }
            }
            out[((size_t)r * C + c0 + cl) * bins + t] = acc / (float)(G * G);
        }
    }
}

__global__ void search_region_kernel(const float* __restrict__ boxes, int N, float pad, float half_e,
                                     float two_e, float min_wh, float* __restrict__ sr) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float x1 = add_rn(boxes[n * 4 + 0], pad), y1 = add_rn(boxes[n * 4 + 1], pad);
    const float x2 = add_rn(boxes[n * 4 + 2], pad), y2 = add_rn(boxes[n * 4 + 3], pad);
    const float w = add_rn(sub_rn(x2, x1), 1.0f);
    const float h = add_rn(sub_rn(y2, y1), 1.0f);
    const float w_ext = max_nan(div_rn(sub_rn(min_wh, w), two_e), mul_rn(w, half_e));
    const float h_ext = max_nan(div_rn(sub_rn(min_wh, h), two_e), mul_rn(h, half_e));
    sr[n * 4 + 0] = sub_rn(x1, w_ext);
    sr[n * 4 + 1] = sub_rn(y1, h_ext);
    sr[n * 4 + 2] = add_rn(x2, w_ext);
    sr[n * 4 + 3] = add_rn(y2, h_ext);
}

}  // namespace smot

namespace smot {
int launch_roi_pool_separable(const LevelParams& P, int C, const float* rois, const float* level_boxes, int R,
                              int out_size, float* out, int32_t* levels_out, hipStream_t st);   // sr_xcorr.hip
}

extern "C" int smot_roi_align_levels_fwd(const float* const* feats, const int* heights,
                                         const int* widths, const int* pad_cells,
                                         const float* scales, int num_levels, int C,
                                         const float* rois, const float* level_boxes, int R,
                                         int out_h, int out_w, int sampling_ratio, float* out,
                                         int32_t* levels_out, smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(C > 0 && out_h > 0 && out_w > 0 && R >= 0, "roi_align: bad sizes C=%d out=%dx%d R=%d", C,
                 out_h, out_w, R);
    if (sampling_ratio <= 0 || sampling_ratio > 4) {
        set_error("roi_align: sampling_ratio=%d unsupported (need 1..4; adaptive grid not implemented)",
                  sampling_ratio);
        return SMOT_ERR_UNSUPPORTED;
    }
    if (R == 0) return SMOT_OK;
    SMOT_REQUIRE(rois && out, "roi_align: null rois/out");
    SMOT_REQUIRE(num_levels == 1 || level_boxes, "roi_align: level_boxes required for num_levels>1");
    LevelParams P;
    {
        const int rc = fill_level_params(&P, feats, heights, widths, pad_cells, scales, num_levels, "roi_align");
        if (rc) return rc;
    }

    // the EMM pooler shapes (15x15 templates, 30x30 search regions) and the box head's 7x7, 2x2 samples, take the separable
    // wave-per-two-planes kernel of sr_xcorr.hip; everything else the generic kernel below
    if (out_h == out_w && (out_h == 7 || out_h == 15 || out_h == 30) && sampling_ratio == 2 && !knobs().roi_generic)
        return launch_roi_pool_separable(P, C, rois, num_levels > 1 ? level_boxes : rois, R, out_h, out, levels_out,
                                         (hipStream_t)stream);
    const int ch_per_block = RA_CH;
    dim3 grid(R, (C + ch_per_block - 1) / ch_per_block);
    const size_t smem = (size_t)(out_h + out_w) * sampling_ratio * 16 + (size_t)RA_CH * RA_WIN_FLOATS * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(G)                                                                                        \
    {                                                                                                    \
        const int rco = ensure_lds_optin((const void*)roi_align_levels_kernel<G>, 96 * 1024, "roi_align");   \
        if (rco) return rco;                                                                             \
        hipLaunchKernelGGL(roi_align_levels_kernel<G>, grid, dim3(256), smem, st, P, C, rois, level_boxes, \
                           out_h, out_w, ch_per_block, out, levels_out, 0);                              \
    }
    SMOT_REQUIRE(smem <= 96 * 1024, "roi_align: pooled size %dx%d needs too much LDS", out_h, out_w);
    switch (sampling_ratio) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    return check_launch("roi_align");
}

extern "C" int smot_roi_align_fwd(const float* input, int num_images, int C, int H, int W, int pad_cells,
                                  const float* rois5, int R, float spatial_scale, int pooled_h, int pooled_w,
                                  int sampling_ratio, float* out, smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(num_images > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0 && R >= 0 && pad_cells >= 0 &&
                     spatial_scale > 0.f,
                 "roi_align: bad sizes B=%d C=%d H=%d W=%d pooled=%dx%d R=%d pad=%d scale=%g", num_images, C, H, W,
                 pooled_h, pooled_w, R, pad_cells, (double)spatial_scale);
    if (sampling_ratio <= 0 || sampling_ratio > 4) {
        set_error("roi_align: sampling_ratio=%d unsupported (need 1..4; adaptive grid not implemented)",
                  sampling_ratio);
        return SMOT_ERR_UNSUPPORTED;
    }
    if (R == 0) return SMOT_OK;
    SMOT_REQUIRE(input && rois5 && out, "roi_align: null pointer");
    LevelParams P;
    {
        const int rc = fill_level_params(&P, &input, &H, &W, &pad_cells, &spatial_scale, 1, "roi_align");
        if (rc) return rc;
    }
    dim3 grid(R, (C + RA_CH - 1) / RA_CH);
    const size_t smem = (size_t)(pooled_h + pooled_w) * sampling_ratio * 16 + (size_t)RA_CH * RA_WIN_FLOATS * sizeof(float);
    SMOT_REQUIRE(smem <= 96 * 1024, "roi_align: pooled size %dx%d needs too much LDS", pooled_h, pooled_w);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(G)                                                                                        \
    {                                                                                                    \
        const int rco = ensure_lds_optin((const void*)roi_align_levels_kernel<G>, 96 * 1024, "roi_align");   \
        if (rco) return rco;                                                                             \
        hipLaunchKernelGGL(roi_align_levels_kernel<G>, grid, dim3(256), smem, st, P, C, rois5,           \
                           (const float*)nullptr, pooled_h, pooled_w, RA_CH, out, (int32_t*)nullptr, num_images); \
    }
    switch (sampling_ratio) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    return check_launch("roi_align");
}

extern "C" int smot_search_region_fwd(const float* boxes, int N, float pad_pixels, float search_expansion,
                                      float min_search_wh, float* sr, smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(N >= 0, "search_region: N=%d", N);
    if (N == 0) return SMOT_OK;
    SMOT_REQUIRE(boxes && sr, "search_region: null pointer");
    // the reference forms e/2 and e*2 in Python double before they meet the fp32 tensors
    const float half_e = (float)((double)search_expansion / 2.0);
    const float two_e = (float)((double)search_expansion * 2.0);
    hipLaunchKernelGGL(search_region_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, boxes, N,
                       pad_pixels, half_e, two_e, min_search_wh, sr);
    return check_launch("search_region");
}
