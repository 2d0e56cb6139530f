// The FMA phase of the two-planes-per-wave xcorr kernel (see xcorr.hip), shared with the fused
// search-region-pool + xcorr kernel (sr_xcorr.hip).  `xs`: two search planes in LDS at plane stride
// XP2_XP floats / row stride XP2_XS; `zs`: two templates at plane stride RZ*XP2_ZS / row stride XP2_ZS.
// MODE 1 = skip the FMAs (staging ablation), otherwise full.
#pragma once
#include "smot_common.h"

namespace smot {

constexpr int XP2_XS = 36;      // 4*XS mod 64 == 16 -> the four row-quads land on disjoint LDS bank windows
constexpr int XP2_XP = 1088;
constexpr int XP2_ZS = 16;

template <int RX, int RZ, int MODE>
__device__ __forceinline__ void xcorr_patch2_compute(const float* xs, const float* zs, int lane,
                                                     float* __restrict__ out, int plane0, int planes) {
    constexpr int HO = RX - RZ + 1;
    constexpr int XS = XP2_XS, XP = XP2_XP, ZS = XP2_ZS, ZP = RZ * XP2_ZS;
    constexpr int WIN = RZ + 1;
    const int p = lane >> 5, q = (lane >> 3) & 3, g = lane & 7;
    const float* xrow = xs + p * XP + (4 * q) * XS + 2 * g;
    const float* zrow = zs + p * ZP;
    float acc[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k][0] = acc[k][1] = 0.0f;
    float zr[RZ][RZ];
    float wa[WIN], wb[WIN];
#define SMOT_LOAD_X(T, DST)                                                                 \
    {                                                                                       \
        _Pragma("unroll") for (int m = 0; m < WIN / 2; ++m) {                               \
            const float2 v2 = *reinterpret_cast<const float2*>(xrow + (T) * XS + 2 * m);    \
            DST[2 * m + 0] = v2.x;                                                          \
            DST[2 * m + 1] = v2.y;                                                          \
        }                                                                                   \
    }
#define SMOT_LOAD_Z(T)                                                                      \
    {                                                                                       \
        _Pragma("unroll") for (int m = 0; m < 3; ++m) {                                     \
            const float4 v4 = *reinterpret_cast<const float4*>(zrow + (T) * ZS + 4 * m);    \
            zr[T][4 * m + 0] = v4.x;                                                        \
            zr[T][4 * m + 1] = v4.y;                                                        \
            zr[T][4 * m + 2] = v4.z;                                                        \
            zr[T][4 * m + 3] = v4.w;                                                        \
        }                                                                                   \
        zr[T][12] = zrow[(T) * ZS + 12];                                                    \
        zr[T][13] = zrow[(T) * ZS + 13];                                                    \
        zr[T][14] = zrow[(T) * ZS + 14];                                                    \
    }
#define SMOT_PIN_ACC()                                                                      \
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]),   \
                      "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
#define SMOT_STEP(T, CUR, NXT)                                                              \
    {                                                                                       \
        if ((T) + 1 < RZ + 3) SMOT_LOAD_X((T) + 1, NXT)                                     \
        if ((T) + 1 < RZ) SMOT_LOAD_Z((T) + 1)                                              \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                     \
            const int u = (T) - k;                                                          \
            if (MODE == 1) {                                                                \
                if (u >= 0 && u < RZ) acc[k][0] += CUR[k] + zr[u][k];                       \
            } else if (u >= 0 && u < RZ) {                                                  \
                _Pragma("unroll") for (int v = 0; v < RZ; ++v) {                            \
                    acc[k][0] = fmaf(CUR[v], zr[u][v], acc[k][0]);                          \
                    acc[k][1] = fmaf(CUR[v + 1], zr[u][v], acc[k][1]);                      \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        SMOT_PIN_ACC()                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    }
    SMOT_LOAD_X(0, wa)
    SMOT_LOAD_Z(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t2 = 0; t2 < RZ + 3; t2 += 2) {
        SMOT_STEP(t2, wa, wb)
        if (t2 + 1 < RZ + 3) SMOT_STEP(t2 + 1, wb, wa)
    }
#undef SMOT_STEP
#undef SMOT_PIN_ACC
#undef SMOT_LOAD_Z
#undef SMOT_LOAD_X

    const int plane = plane0 + p;
    if (plane < planes) {
        float* o = out + (size_t)plane * (HO * HO) + (4 * q) * HO + 2 * g;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<float2*>(o + k * HO) = make_float2(acc[k][0], acc[k][1]);
    }
}

}  // namespace smot
