// Shared by decode.hip (device side) and emm_fused.hip (host side fills it).
#pragma once
#include "smot_common.h"

namespace smot {

// Where the 7 x Ho x Ho logits of a track come from: the [N,7,Ho,Ho] tensor of smot_emm_predictor_fwd, or
// (one-call path, Ho == 16) the per-tile partial head sums the tower kernel leaves behind — summed in
// fixed tile order + bias, ReLU on the reg channels, exactly what heads_combine_kernel computes — so the
// combine launch and the logits round trip disappear.
struct LogitSrc {
    const float* logits;     // [N,7,HW] or nullptr
    const float* part;       // [N, 2*tpt, 4, 256]
    int tpt;                 // tiles per tower
    const float* cls_b;
    const float* center_b;
    const float* reg_b;
    __device__ __forceinline__ float get(int n, int ch, int pos, int HW) const {
        if (logits != nullptr) return logits[((size_t)n * 7 + ch) * HW + pos];
        const int side = ch >= 3;
        const int o = side ? ch - 3 : ch;
        const float* p = part + ((size_t)n * 2 * tpt + side * tpt) * 4 * 256 + (size_t)o * 256 + pos;
        float s = 0.0f;
        for (int t = 0; t < tpt; ++t) s += p[(size_t)t * 4 * 256];
        s += (ch < 2) ? cls_b[ch] : ((ch == 2) ? center_b[0] : reg_b[ch - 3]);
        return side ? fmaxf(s, 0.0f) : s;
    }
};

}  // namespace smot
