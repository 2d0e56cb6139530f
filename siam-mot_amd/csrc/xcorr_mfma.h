// Depthwise cross-correlation on the 4x4x1 matrix instruction (v_mfma_f32_4x4x1_16B_f32), shared by the
// stand-alone kernel (xcorr.hip) and the fused pooling + correlation kernel (sr_xcorr.hip).
//
// One instruction = 16 independent 4x4 rank-1 updates (256 MACs in 8 cycles: twice the v_fmac rate).  The 16
// blocks are the sixteen 4x4 tiles of one 16x16 response plane, block b = (bi, bj) = (b % 4, b / 4).  For a
// template row u and a search-plane column offset cc in [0, 18):
//     A (lane = 4b + i') = x[4bi + i' + u][4bj + cc]          -- four rows of the search plane, one column
//     B (lane = 4b + j') = z[u][cc - j']  (0 outside [0,15))  -- the template row, shifted per output column
//     D[i'][j'] += A[i'] * B[j']   ==>   out[4bi+i'][4bj+j'] += x[4bi+i'+u][4bj+j'+v] * z[u][v],  v = cc - j'
// 15 x 18 = 270 instructions per plane, 83 % of their MACs useful (the 3+3 zero taps at the ends of the shifted
// template row).  Every output is one fp32 fmaf chain in (u, v) order — the reference's order — so the result is
// bit-identical to the VALU kernels, with one caveat: a zero-weight tap still multiplies a real x value
// (0 * x), so an inf / NaN inside the plane also poisons the neighbouring outputs of its 4-column tile.
//
// LDS image: search planes at row stride XM_XS = 40 floats (with the column-major block order this makes every
// 16-lane group of the A ds_read_b128 hit 16 distinct 16-byte slots: conflict-free); templates as four
// pre-shifted, zero-padded copies zs4[j'][u][20] so that B is a broadcast ds_read_b128 as well.
// Per template row: 5 + 5 ds_read_b128 feed 18 MFMAs.  Two planes per wave, interleaved (independent
// accumulators back to back).
#pragma once
#include "smot_common.h"

namespace smot {

typedef float xm_f32x4 __attribute__((ext_vector_type(4)));

constexpr int XM_XS = 40;                    // search-plane row stride
constexpr int XM_ZC = 20;                    // columns of a shifted template row (18 used)

template <int RX>
constexpr int xm_xplane() { return RX * XM_XS; }
template <int RZ>
constexpr int xm_zplane() { return 4 * RZ * XM_ZC; }

// park one template (row-major RZ x RZ, value v = z[u][c] held by the caller) into the four shifted copies
template <int RZ>
__device__ __forceinline__ void xm_store_template(float* zs4, int u, int c, float v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) zs4[(j * RZ + u) * XM_ZC + c + j] = v;      // zs4[j][u][cc] = z[u][cc - j]
}

// zero the padding of the shifted copies (entries cc - j outside [0, RZ)); `lane` strides over one template
template <int RZ>
__device__ __forceinline__ void xm_zero_template_pad(float* zs4, int lane) {
    for (int e = lane; e < 4 * RZ * XM_ZC; e += 64) {
        const int j = e / (RZ * XM_ZC);
        const int cc = e % XM_ZC;
        if (cc - j < 0 || cc - j >= RZ) zs4[e] = 0.0f;
    }
}

// xs: two search planes (plane stride xm_xplane<RX>()), zs: two templates (stride xm_zplane<RZ>()).
template <int RX, int RZ>
__device__ __forceinline__ void xcorr_mfma_compute(const float* xs, const float* zs, int lane,
                                                   float* __restrict__ out, int plane0, int planes) {
    static_assert(RX - RZ + 1 == 16 && RZ == 15, "tiles a 16x16 response of a 15x15 template");
    constexpr int XP = RX * XM_XS, ZP = 4 * RZ * XM_ZC;
    const int b = lane >> 2, q = lane & 3;
    const int bi = b & 3, bj = b >> 2;
    const float* xa = xs + (4 * bi + q) * XM_XS + 4 * bj;        // A: row 4bi + i' (+u), columns 4bj + cc
    const float* zb = zs + q * (RZ * XM_ZC);                      // B: shifted copy j'
    xm_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int u = 0; u < RZ; ++u) {
        xm_f32x4 a0[5], a1[5], b0[5], b1[5];
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            a0[g] = *reinterpret_cast<const xm_f32x4*>(xa + u * XM_XS + 4 * g);
            a1[g] = *reinterpret_cast<const xm_f32x4*>(xa + XP + u * XM_XS + 4 * g);
            b0[g] = *reinterpret_cast<const xm_f32x4*>(zb + u * XM_ZC + 4 * g);
            b1[g] = *reinterpret_cast<const xm_f32x4*>(zb + ZP + u * XM_ZC + 4 * g);
        }
#pragma unroll
        for (int cc = 0; cc < RZ + 3; ++cc) {
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[cc >> 2][cc & 3], b0[cc >> 2][cc & 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[cc >> 2][cc & 3], b1[cc >> 2][cc & 3], acc1, 0, 0, 0);
        }
    }
    // D: lane (b, j') holds out[4bi + i'][4bj + j'] in register i'
    const int col = 4 * bj + q;
    if (plane0 < planes) {
        float* o = out + (size_t)plane0 * 256 + (4 * bi) * 16 + col;
        o[0] = acc0[0];
        o[16] = acc0[1];
        o[32] = acc0[2];
        o[48] = acc0[3];
    }
    if (plane0 + 1 < planes) {
        float* o = out + (size_t)(plane0 + 1) * 256 + (4 * bi) * 16 + col;
        o[0] = acc1[0];
        o[16] = acc1[1];
        o[32] = acc1[2];
        o[48] = acc1[3];
    }
}

}  // namespace smot
