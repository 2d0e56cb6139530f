// Shared by the tower kernels (predictor.hip, tower_wino.hip).
#pragma once
#include "smot_common.h"

namespace smot {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct TowerParams {
    const float* w[2];      // [C, C, 3, 3] cls_tower.0.weight / reg_tower.0.weight
    const float* gamma[2];  // [C]
    const float* beta[2];   // [C]
    // head filters (fused partial heads): cls [2,C,3,3], center [1,C,3,3], reg [4,C,3,3]
    const float* cls_w;
    const float* center_w;
    const float* reg_w;
};

__device__ __forceinline__ float group16_sum(float v) {
    // sum over the 16 lanes that share lane>>4 (= one DPP row): four row rotations, no LDS crossbar round trips
    // (__shfl_xor compiles to ds_bpermute: four dependent ~100-cycle trips per sum)
#define SMOT_ROR(N) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (N), 0xf, 0xf, false))
    v += SMOT_ROR(8);
    v += SMOT_ROR(4);
    v += SMOT_ROR(2);
    v += SMOT_ROR(1);
#undef SMOT_ROR
    return v;
}

// |x| maxima of a response plane (the split form of the tower kernel scales a track's response by a power of two chosen from
// them): NaN is ignored (fmaxf), an infinite value stays; the wave form returns the maximum over the 64 lanes, wave-uniform
__device__ __forceinline__ float plane_max_step(float m, float v) { return fmaxf(m, fabsf(v)); }
__device__ __forceinline__ float plane_max_wave(float v) {
#define SMOT_ROR(N) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (N), 0xf, 0xf, false))
    v = fmaxf(v, SMOT_ROR(8));
    v = fmaxf(v, SMOT_ROR(4));
    v = fmaxf(v, SMOT_ROR(2));
    v = fmaxf(v, SMOT_ROR(1));
#undef SMOT_ROR
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

constexpr int T_PLANE = 336;                      // 18*18 = 324 padded to 336

// tower_wino.hip
// zero_words: N words the kernel sets to zero (the decode kernel's per-track tickets: zeroed here, one launch
// earlier in the same stream, instead of by a memset node of their own), or nullptr
// plane_max: |resp| maximum of every (track, channel) plane [N][C] — read by the split (fp16 x 2) form only
int launch_tower_wino(const float* resp, const float* packed, const TowerParams& P, int N, int C, int cpg, float eps,
                      float* part, unsigned* zero_words, hipStream_t st, const float* plane_max);
// tower_wino.hip: the plane maxima of a response that did not come with them (one wave per plane)
int launch_plane_absmax(const float* resp, int planes, int hw, float* pm, hipStream_t st);

// tower_conv.hip: towers + heads of a response map other than 16x16 on the matrix cores (instantiated for Ho = 29,
// the reference's second yaml family): logits complete on return; SMOT_ERR_UNSUPPORTED when no instantiation fits
int launch_tower_conv(const float* resp, const TowerParams& P, int N, int C, int Ho, int cpg, float eps,
                      const float* cls_b, const float* center_b, const float* reg_b, float* tower_ws, float* logits,
                      unsigned* zero_words, hipStream_t st);
// tower_conv.hip: the same through the blocked Winograd kernel (needs the packed filters); SMOT_ERR_UNSUPPORTED otherwise
int launch_tower_conv_wino(const float* resp, const float* packed, const TowerParams& P, int N, int C, int Ho, int cpg,
                           float eps, const float* cls_b, const float* center_b, const float* reg_b, float* tower_ws,
                           float* logits, unsigned* zero_words, hipStream_t st, const float* plane_max);

}  // namespace smot
