// Shared by roi_align.hip and sr_xcorr.hip: level parameters, [UPSTREAM] LevelMapper and the per-axis
// sample bookkeeping of the legacy ROIAlign with virtual zero padding.
#pragma once
#include "smot_common.h"

namespace smot {

struct LevelParams {
    const float* feat[SMOT_MAX_LEVELS];
    int H[SMOT_MAX_LEVELS];
    int W[SMOT_MAX_LEVELS];
    int pad[SMOT_MAX_LEVELS];
    float scale[SMOT_MAX_LEVELS];
    int num_levels;
    float k_min, k_max;
};

// [UPSTREAM] LevelMapper: floor(4 + log2(sqrt(area)/224 + 1e-6)), clamped, 0-based.
__device__ __forceinline__ int map_level(const float* b, float k_min, float k_max) {
    const float w = add_rn(sub_rn(b[2], b[0]), 1.0f);
    const float h = add_rn(sub_rn(b[3], b[1]), 1.0f);
    const float s = sqrtf(mul_rn(w, h));
    float lvl = floorf(add_rn(4.0f, log2f(add_rn(div_rn(s, 224.0f), 1e-6f))));
    lvl = fminf(fmaxf(lvl, k_min), k_max);
    return (int)lvl - (int)k_min;
}

// One axis sample of the legacy ROIAlign, evaluated against the PADDED extent `size_p`
// (= real + 2*pad) and re-expressed as indices into the REAL map.
__device__ __forceinline__ void axis_sample(float start, float bin, int G, int s, int size_real,
                                            int pad, int* lo, int* hi, float* w_lo, float* w_hi) {
    const int p = s / G;
    const int i = s - p * G;
    const int size_p = size_real + 2 * pad;
    // roi_start + p*bin + (i+.5f)*bin/G, each op rounded separately as in the reference
    float c = add_rn(add_rn(start, mul_rn((float)p, bin)),
                     div_rn(mul_rn((float)i + 0.5f, bin), (float)G));
    const bool valid = !(c < -1.0f || c > (float)size_p);
    if (c <= 0.0f) c = 0.0f;
    int l = (int)c;
    int h;
    if (l >= size_p - 1) {
        h = l = size_p - 1;
        c = (float)l;
    } else {
        h = l + 1;
    }
    const float fl = sub_rn(c, (float)l);   // weight of the high cell
    const float fh = sub_rn(1.0f, fl);      // weight of the low cell
    const int lr = l - pad, hr = h - pad;
    const bool lo_in = valid && lr >= 0 && lr < size_real;
    const bool hi_in = valid && hr >= 0 && hr < size_real;
    *lo = lo_in ? lr : 0;
    *hi = hi_in ? hr : 0;
    *w_lo = lo_in ? fh : 0.0f;
    *w_hi = hi_in ? fl : 0.0f;
}

// Host side: validate the per-level HOST arrays of the C ABI and pack them into kernel parameters.
inline int fill_level_params(LevelParams* P, const float* const* feats, const int* heights, const int* widths,
                             const int* pad_cells, const float* scales, int num_levels, const char* who) {
    if (!(feats && heights && widths && scales)) {
        set_error("%s: null level array", who);
        return SMOT_ERR_BAD_ARG;
    }
    if (num_levels < 1 || num_levels > SMOT_MAX_LEVELS) {
        set_error("%s: num_levels=%d not in [1,%d]", who, num_levels, SMOT_MAX_LEVELS);
        return SMOT_ERR_BAD_ARG;
    }
    for (int l = 0; l < num_levels; ++l) {
        const int pad = pad_cells ? pad_cells[l] : 0;
        if (!(feats[l] && heights[l] > 0 && widths[l] > 0 && pad >= 0 && scales[l] > 0.f)) {
            set_error("%s: bad level %d", who, l);
            return SMOT_ERR_BAD_ARG;
        }
        P->feat[l] = feats[l];
        P->H[l] = heights[l];
        P->W[l] = widths[l];
        P->pad[l] = pad;
        P->scale[l] = scales[l];
    }
    P->num_levels = num_levels;
    P->k_min = -log2f(scales[0]);
    P->k_max = -log2f(scales[num_levels - 1]);
    return SMOT_OK;
}

}  // namespace smot
