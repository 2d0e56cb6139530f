// One-plane-per-wave FMA phase of the depthwise cross-correlation (sixth generation, xcorr.hip).
// Same scheme as xcorr_patch2.h — the wave walks the window rows in lock-step, a row segment is read from LDS once
// and feeds every (output row, template row) pair that needs it, one fmaf chain per output in (u, v) order — but
// all 64 lanes work on ONE plane with 2x2 output patches (lane (q, g): rows 2q..2q+1, columns 2g..2g+1).
// Half the FMAs per lane and twice the waves per plane set: at the metric's 30 tracks (3,840 planes, 1,024 SIMDs)
// that is ~3.75 resident waves per SIMD instead of ~1.9, which is what the fp32 FMA issue rate needs (measured:
// 5.6 cycles per v_fmac with one wave per SIMD, 2.8 with two or more).  The price is 1.8x the LDS read volume per
// FMA (a row segment now feeds two output rows instead of four).
// `xs`: one search plane at row stride XP1_XS = 40 floats (2*40 = 16 mod 64: the four q of a 32-lane ds_read_b64
// group land on disjoint 16-bank windows); `zs`: the template at row stride 16.
#pragma once
#include "smot_common.h"

namespace smot {

constexpr int XP1_XS = 40;
constexpr int XP1_ZS = 16;

// LEAN: no software prefetch of the next window / template row (one row buffer instead of two, two live template
// rows instead of three): ~65 instead of ~90 VGPRs.  For callers that run six waves per SIMD (the fused pooling +
// correlation kernel at three workgroups per CU), where other waves cover the LDS latency.  Same FMA order.
// pmax != nullptr: the plane's largest |response| goes to pmax[plane] (NaN ignored, an infinite value kept) — the tower
// kernel's split form scales a track's response by a power of two chosen from these (tower_wino.hip)
template <int RX, int RZ, bool LEAN = false>
__device__ __forceinline__ void xcorr_patch1_compute(const float* xs, const float* zs, int lane,
                                                     float* __restrict__ out, int plane, float* __restrict__ pmax = nullptr) {
    constexpr int HO = RX - RZ + 1;
    static_assert(HO == 16 && RZ == 15, "tiles a 16x16 response of a 15x15 template");
    constexpr int XS = XP1_XS, ZS = XP1_ZS;
    constexpr int WIN = RZ + 1;                       // 16 floats: columns 2g .. 2g+15
    const int q = lane >> 3, g = lane & 7;
    const float* xrow = xs + (2 * q) * XS + 2 * g;
    float acc[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
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
            const float4 v4 = *reinterpret_cast<const float4*>(zs + (T) * ZS + 4 * m);      \
            zr[T][4 * m + 0] = v4.x;                                                        \
            zr[T][4 * m + 1] = v4.y;                                                        \
            zr[T][4 * m + 2] = v4.z;                                                        \
            zr[T][4 * m + 3] = v4.w;                                                        \
        }                                                                                   \
        zr[T][12] = zs[(T) * ZS + 12];                                                      \
        zr[T][13] = zs[(T) * ZS + 13];                                                      \
        zr[T][14] = zs[(T) * ZS + 14];                                                      \
    }
#define SMOT_STEP(T, CUR, NXT)                                                              \
    {                                                                                       \
        if (!LEAN && (T) + 1 < RZ + 1) SMOT_LOAD_X((T) + 1, NXT)                            \
        if (!LEAN && (T) + 1 < RZ) SMOT_LOAD_Z((T) + 1)                                     \
        if (LEAN && (T) > 0) SMOT_LOAD_X((T), CUR)                                          \
        if (LEAN && (T) > 0 && (T) < RZ) SMOT_LOAD_Z((T))                                   \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                     \
            const int u = (T) - k;                                                          \
            if (u >= 0 && u < RZ) {                                                         \
                _Pragma("unroll") for (int v = 0; v < RZ; ++v) {                            \
                    acc[k][0] = fmaf(CUR[v], zr[u][v], acc[k][0]);                          \
                    acc[k][1] = fmaf(CUR[v + 1], zr[u][v], acc[k][1]);                      \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1])); \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    }
    SMOT_LOAD_X(0, wa)
    SMOT_LOAD_Z(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t2 = 0; t2 < RZ + 1; t2 += 2) {
        if (LEAN) {
            SMOT_STEP(t2, wa, wa)
            SMOT_STEP(t2 + 1, wa, wa)
        } else {
            SMOT_STEP(t2, wa, wb)
            SMOT_STEP(t2 + 1, wb, wa)
        }
    }
#undef SMOT_STEP
#undef SMOT_LOAD_Z
#undef SMOT_LOAD_X
    float* o = out + (size_t)plane * (HO * HO) + (2 * q) * HO + 2 * g;
    *reinterpret_cast<float2*>(o) = make_float2(acc[0][0], acc[0][1]);
    *reinterpret_cast<float2*>(o + HO) = make_float2(acc[1][0], acc[1][1]);
    if (pmax != nullptr) {
        // four DPP row rotations + four lane reads (no LDS crossbar trip): ~12 vector instructions per plane of ~1,300
        float m = fmaxf(fmaxf(fabsf(acc[0][0]), fabsf(acc[0][1])), fmaxf(fabsf(acc[1][0]), fabsf(acc[1][1])));
#define SMOT_ROR(N) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x120 + (N), 0xf, 0xf, false))
        m = fmaxf(m, SMOT_ROR(8));
        m = fmaxf(m, SMOT_ROR(4));
        m = fmaxf(m, SMOT_ROR(2));
        m = fmaxf(m, SMOT_ROR(1));
#undef SMOT_ROR
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 16));
        const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 48));
        if (lane == 0) pmax[plane] = fmaxf(fmaxf(a, b), fmaxf(c, d));
    }
}

}  // namespace smot
