// Shared by decode.hip (device side) and emm_fused.hip (host side fills it).
#pragma once
#include "smot_common.h"

namespace smot {

// Where the 7 x Ho x Ho logits of a track come from: the [N,7,Ho,Ho] tensor of smot_emm_predictor_fwd, or
// (one-call path, Ho == 16) the per-tile partial head sums the tower kernel leaves behind — summed in
// fixed tile order + bias, ReLU on the reg channels, exactly what heads_combine_kernel computes — so the
// combine launch and the logits round trip disappear.
struct LogitSrc {
    const float* logits;     // [N,7,HW], or nullptr when `part` is the source
    const float* part;       // [N, 2*tpt, 4, 256] per-tile partial head sums (Ho == 16)
    int tpt;                 // tiles per tower
    const float* cls_b;
    const float* center_b;
    const float* reg_b;
    float* logits_out;       // with `part`: where the band-0 workgroup leaves the combined [N,7,256] logits
                             // for pass 2 (the finalize kernel reads plain logits)

    __device__ __forceinline__ float get(int n, int ch, int pos, int HW) const {
        return logits[((size_t)n * 7 + ch) * HW + pos];
    }

    // Combined logit of channel `ch` at position `pos` from the partials: all tile loads are issued before the
    // first add (one memory round trip), then summed in tile order + bias (+ReLU on reg) — the arithmetic of
    // heads_combine_kernel.
    __device__ __forceinline__ float combine(int n, int ch, int pos) const {
        const int side = ch >= 3;
        const int o = side ? ch - 3 : ch;
        const float* p = part + ((size_t)n * 2 * tpt + side * tpt) * 4 * 256 + (size_t)o * 256 + pos;
        float s = 0.0f;
        for (int t0 = 0; t0 < tpt; t0 += 8) {
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = (t0 + t < tpt) ? p[(size_t)(t0 + t) * 4 * 256] : 0.0f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t0 + t < tpt) s += v[t];
        }
        s += (ch < 2) ? cls_b[ch] : ((ch == 2) ? center_b[0] : reg_b[ch - 3]);
        return side ? relu_nan(s) : s;
    }
};

}  // namespace smot
