// Packed-fp32 FMA phase of the depthwise cross-correlation inside the fused pooling + correlation kernel
// (sr_xcorr.hip; replaces xcorr_depthwise, reference EMM/xcorr.py:37-46): v_pk_fma_f32 with NO register shuffles.
//
// A packed FMA needs its two multiplicands in one aligned VGPR pair.  Pairing neighbouring outputs (columns c, c+1 or
// rows r, r+1) makes every second tap straddle two pairs (gfx950 wants 64-bit operands even-aligned), which is what
// the SLP vectoriser pays 204 v_mov per plane for.  Pairing output rows r and r+8 does not: the LDS image stores
// pair-row i = (x[i][:], x[i+8][:]) interleaved as float2 for i = 0..21 (written that way by the pooling phase), the
// outputs (r, c) and (r+8, c) need (x[r+u][c+v], x[r+8+u][c+v]) = element c+v of pair-row r+u for EVERY tap — always
// one aligned float2 — and the template tap is the op_sel-broadcast half of a VGPR pair.
//   900 v_fma_f32 + 128 ds_read_b64 per plane  ->  450 v_pk_fma_f32 + 64 ds_read_b128 per plane.
// Each output is still ONE fmaf chain in (u, v) order: results are bit-identical to xcorr_patch1/2 and to the
// stand-alone operator.
//
// Two planes per wave (lanes 0..31 / 32..63).  Lane (rq, g): output rows 2rq + k (+8 in the high half), columns 2g + j,
// k, j in {0, 1}: four accumulator pairs.  The wave walks the pair-rows t = 0..15 in lock-step: pair-row 2rq + t is
// read once (8 x ds_read_b128: columns 2g .. 2g+15) and feeds (k = 0, u = t) and (k = 1, u = t - 1).
// Image row stride 80 floats: 2*80 = 32 mod 64, so the four 16-lane groups a ds_read_b128 is serviced in
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... MI355X_MICROARCH.md, LDS) each cover the 64 banks exactly once.
// The 20 spare floats of pair-row u hold template row u (all lanes of a plane read the same address: broadcast).
#pragma once
#include "smot_common.h"

namespace smot {
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int XPP_RS = 80;        // floats per pair-row of the interleaved image (40 float2; 30 used + template row)
constexpr int XPP_ROWS = 22;      // pair-rows i = 0..21: (x[i], x[i+8])
constexpr int XPP_XP = XPP_ROWS * XPP_RS;     // floats per plane (7,040 B)
constexpr int XPP_ZOFF = 60;      // template row u lives at floats 60..74 of pair-row u

__device__ __forceinline__ v2f pk_fma_lo(v2f a, v2f b, v2f c) {   // c + a * b.x
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ v2f pk_fma_hi(v2f a, v2f b, v2f c) {   // c + a * b.y
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(c) : "v"(a), "v"(b));
    return c;
}

// `img`: this LANE's plane image (lanes 0..31: the wave's first plane, lanes 32..63: its second).
// PREFETCH: the next pair-row is read while the current one is multiplied (109 instead of 76 VGPRs).
template <int RX, int RZ, bool PREFETCH>
__device__ __forceinline__ void xcorr_pairs_compute(const float* img, int lane, float* __restrict__ out, int plane,
                                                    bool store) {
    constexpr int HO = RX - RZ + 1;
    static_assert(HO == 16 && RZ == 15 && RX == 30, "16x16 response of a 15x15 template in a 30x30 search plane");
    constexpr int RS = XPP_RS, ZS = XPP_RS;
    const int rq = (lane >> 3) & 3, g = lane & 7;
    const float* zs = img + XPP_ZOFF;
    const float* wrow = img + (2 * rq) * RS + 4 * g;          // float2 column 2g
    v2f acc[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
    v2f wa[16], wb[16];
    v2f zr[2][8];
#define LOAD_W(T, DST)                                                                   \
    {                                                                                    \
        _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                  \
            const float4 v4 = *reinterpret_cast<const float4*>(wrow + (T) * RS + 4 * m); \
            DST[2 * m] = (v2f){v4.x, v4.y};                                              \
            DST[2 * m + 1] = (v2f){v4.z, v4.w};                                          \
        }                                                                                \
    }
#define LOAD_Z(U)                                                                        \
    {                                                                                    \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                  \
            const float4 v4 = *reinterpret_cast<const float4*>(zs + (U) * ZS + 4 * m);   \
            zr[(U) & 1][2 * m] = (v2f){v4.x, v4.y};                                      \
            zr[(U) & 1][2 * m + 1] = (v2f){v4.z, v4.w};                                  \
        }                                                                                \
    }
#define STEP(T, CUR, NXT)                                                                \
    {                                                                                    \
        if (PREFETCH && (T) + 1 < 16) LOAD_W((T) + 1, NXT)                               \
        if (!PREFETCH && (T) > 0) LOAD_W((T), CUR)                                       \
        if ((T) > 0 && (T) < RZ) LOAD_Z((T))                                             \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                  \
            const int u = (T) - k;                                                       \
            if (u >= 0 && u < RZ) {                                                      \
                _Pragma("unroll") for (int v = 0; v < RZ; ++v) {                         \
                    if (v & 1) {                                                         \
                        acc[k][0] = pk_fma_hi(CUR[v], zr[u & 1][v >> 1], acc[k][0]);     \
                        acc[k][1] = pk_fma_hi(CUR[v + 1], zr[u & 1][v >> 1], acc[k][1]); \
                    } else {                                                             \
                        acc[k][0] = pk_fma_lo(CUR[v], zr[u & 1][v >> 1], acc[k][0]);     \
                        acc[k][1] = pk_fma_lo(CUR[v + 1], zr[u & 1][v >> 1], acc[k][1]); \
                    }                                                                    \
                }                                                                        \
            }                                                                            \
        }                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                               \
    }
    LOAD_W(0, wa)
    LOAD_Z(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t2 = 0; t2 < 16; t2 += 2) {
        if (PREFETCH) {
            STEP(t2, wa, wb)
            STEP(t2 + 1, wb, wa)
        } else {
            STEP(t2, wa, wa)
            STEP(t2 + 1, wa, wa)
        }
    }
#undef STEP
#undef LOAD_Z
#undef LOAD_W
    if (store) {
        float* o = out + (size_t)plane * (HO * HO) + (2 * rq) * HO + 2 * g;
        *reinterpret_cast<float2*>(o) = make_float2(acc[0][0].x, acc[0][1].x);
        *reinterpret_cast<float2*>(o + HO) = make_float2(acc[1][0].x, acc[1][1].x);
        *reinterpret_cast<float2*>(o + 8 * HO) = make_float2(acc[0][0].y, acc[0][1].y);
        *reinterpret_cast<float2*>(o + 9 * HO) = make_float2(acc[1][0].y, acc[1][1].y);
    }
}
}  // namespace smot
