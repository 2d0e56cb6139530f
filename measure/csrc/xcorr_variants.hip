// Earlier generations of the depthwise cross-correlation kernel (xcorr.hip) — A/B and ablation material for
// tools/kernel_bench.py and tests/test_hip_parity.py::test_xcorr_kernel_generations_are_bitwise_equal.
// Compiled into libsmot_emm_debug.so only (build.py: -DSMOT_DEBUG); the product library does not contain them.
//
// Wave kernel (first generation, Ho == 16): ONE WAVEFRONT PER PLANE.
//   * the 64 lanes tile the 16x16 output as 16 rows x 4 column-quads: lane = 4*i + g owns
//     out[i][4g..4g+3] -> 4 independent accumulator chains per lane, one float4 store per lane,
//     1 KiB fully coalesced per wave;
//   * the search plane is staged once into an LDS slab private to the wave (row stride 48 floats,
//     which makes the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots: conflict-free);
//     per template row u a lane reads its 18-float window with 4 x ds_read_b128 + 1 x ds_read_b64
//     and feeds 4*Rz FMAs from it (60 FMA per 18 LDS dwords);
//   * template taps are wave-uniform: they are fetched with scalar loads and enter v_fma as the
//     SGPR operand, costing neither VGPRs, LDS bandwidth nor VALU issue slots;
//   * taps are accumulated u-major / v-minor in one fp32 FMA chain per output (deterministic;
//     oracle/emm_oracle.py:xcorr_depthwise uses the same order).
// No workgroup barrier is needed: each wave only touches its own LDS slab.
#include "smot_common.h"
#include "smot_common.h"
#include "knobs.h"
#include "xcorr_patch2.h"
#include "xcorr_mfma.h"
#include "xcorr_patch1.h"

namespace smot {

template <int RX, int RZ>
__global__ void __launch_bounds__(256)
xcorr_dw_wave_kernel(const float* __restrict__ x, const float* __restrict__ z,
                     float* __restrict__ out, int planes) {
    constexpr int HO = RX - RZ + 1;
    static_assert(HO == 16, "wave-per-plane kernel tiles a 16x16 response");
    constexpr int XS = 48;                 // LDS row stride in floats (see header comment)
    constexpr int WIN = RZ + 3;            // floats of one row a lane consumes
    constexpr int NV4 = WIN / 4;           // full float4 reads
    constexpr int REM = WIN - NV4 * 4;     // remainder (0..3 floats)
    static_assert(4 * 3 + WIN <= XS, "window exceeds padded row");
    __shared__ __attribute__((aligned(16))) float xs[4][RX * XS];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int plane = blockIdx.x * 4 + wave;   // wave-uniform
    if (plane >= planes) return;

    float* xw = xs[wave];
    const float* __restrict__ xg = x + (size_t)plane * (RX * RX);
    const float* __restrict__ zg = z + (size_t)plane * (RZ * RZ);

    // stage the search plane: coalesced dword loads, conflict-free LDS stores
    constexpr int NLOAD = (RX * RX + 63) / 64;
    float stage[NLOAD];
#pragma unroll
    for (int t = 0; t < NLOAD; ++t) {
        const int e = lane + 64 * t;
        stage[t] = (e < RX * RX) ? xg[e] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < NLOAD; ++t) {
        const int e = lane + 64 * t;
        if (e < RX * RX) {
            const int r = e / RX;
            xw[r * XS + (e - r * RX)] = stage[t];
        }
    }
    __builtin_amdgcn_wave_barrier();

    const int i = lane >> 2;
    const int g = lane & 3;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
    const float* rowp = xw + i * XS + 4 * g;
#pragma unroll
    for (int u = 0; u < RZ; ++u) {
        float w[NV4 * 4 + 4];
#pragma unroll
        for (int q = 0; q < NV4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(rowp + u * XS + 4 * q);
            w[4 * q + 0] = v.x;
            w[4 * q + 1] = v.y;
            w[4 * q + 2] = v.z;
            w[4 * q + 3] = v.w;
        }
        if (REM == 1) {
            w[4 * NV4] = rowp[u * XS + 4 * NV4];
        } else if (REM == 2) {
            const float2 v = *reinterpret_cast<const float2*>(rowp + u * XS + 4 * NV4);
            w[4 * NV4] = v.x;
            w[4 * NV4 + 1] = v.y;
        } else if (REM == 3) {
            const float2 v = *reinterpret_cast<const float2*>(rowp + u * XS + 4 * NV4);
            w[4 * NV4] = v.x;
            w[4 * NV4 + 1] = v.y;
            w[4 * NV4 + 2] = rowp[u * XS + 4 * NV4 + 2];
        }
#pragma unroll
        for (int v = 0; v < RZ; ++v) {
            const float zt = zg[u * RZ + v];   // wave-uniform address -> scalar load
            acc0 = fmaf(w[v + 0], zt, acc0);
            acc1 = fmaf(w[v + 1], zt, acc1);
            acc2 = fmaf(w[v + 2], zt, acc2);
            acc3 = fmaf(w[v + 3], zt, acc3);
        }
    }
    float4 o;
    o.x = acc0;
    o.y = acc1;
    o.z = acc2;
    o.w = acc3;
    *reinterpret_cast<float4*>(out + (size_t)plane * (HO * HO) + i * HO + 4 * g) = o;
}

// Patch kernel (Ho == 16, default fast path): ONE WAVEFRONT PER FOUR PLANES, no scalar memory.
//   * a 16-lane group owns a plane; lane (q, g) of the group owns the 4x4 output patch rows 4q..4q+3,
//     cols 4g..4g+3 (16 accumulators);
//   * the wave walks the 18 window rows t = 0..17 of its patches in lock-step: row 4q+t of the search
//     plane is read ONCE from LDS (18 floats: 4 x ds_read_b128 + ds_read_b64) and feeds up to four
//     (output row k, template row u = t-k) combinations = up to 240 FMAs — 4x fewer LDS bytes per FMA
//     than the wave-per-plane kernel;
//   * the template row needed at step t is read once per step (4 x ds_read_b128, one address per
//     16-lane group) and stays in registers for the four steps that use it; taps are ordinary VGPR
//     operands, so nothing waits on the scalar cache (in the wave-per-plane kernel every template row
//     was an s_load whose lgkmcnt(0) wait was exposed 15 times per plane);
//   * LDS image: plane stride 1088 floats, row stride 36 floats -> the four 16-lane groups of a
//     ds_read_b128 touch 16 distinct 16-byte slots (conflict-free, derivation in DESIGN.md);
//   * per output the taps are accumulated u-major / v-minor in one fp32 FMA chain: bit-identical to
//     the wave-per-plane kernel and to the order of oracle/emm_oracle.py:xcorr_depthwise.
// MODE 0 = the kernel; 1 = staging + stores only (no FMAs); 2 = FMAs + stores only (no global loads):
// ablation builds selected with SMOT_XCORR_VARIANT=fill|compute for phase timing (tools/kernel_bench.py).
template <int RX, int RZ, int MODE>
__global__ void __launch_bounds__(64)
xcorr_dw_patch_kernel(const float* __restrict__ x, const float* __restrict__ z,
                      float* __restrict__ out, int planes) {
    constexpr int HO = RX - RZ + 1;
    static_assert(HO == 16, "patch kernel tiles a 16x16 response");
    constexpr int XS = 36;                  // search-plane row stride (floats)
    constexpr int XP = 1088;                // search-plane stride (>= RX*XS = 1080, multiple of 64)
    constexpr int ZS = 16;                  // template row stride
    constexpr int ZP = RZ * ZS;             // template plane stride (240)
    constexpr int WIN = RZ + 3;             // 18 floats of a window row feed a 4-wide patch
    static_assert(RX * XS <= XP && RZ <= ZS && 4 * 3 + WIN <= XS, "LDS image too small");
    __shared__ __attribute__((aligned(16))) float sm[4 * XP + 4 * ZP];
    float* xs = sm;
    float* zs = sm + 4 * XP;

    const int lane = threadIdx.x;
    const int plane0 = blockIdx.x * 4;

    // ---- stage 4 search planes + 4 templates ------------------------------------------------
    // The 4 search planes of a set are contiguous in HBM (3600 floats = 900 float4, 16-B aligned):
    // 15 x global_load_dwordx4 per lane; a float4 never straddles a plane (900 % 4 == 0) and splits
    // into two aligned float2 that land in one or two LDS rows (30 % 2 == 0): 2 x ds_write_b64.
    // Templates (225 floats per plane, odd) use dword loads.
    constexpr int NX4 = (4 * RX * RX / 4 + 63) / 64;     // 15
    constexpr int NZ = (RZ * RZ + 63) / 64;              // 4
    if (MODE != 2) {
        const long long last4 = (long long)planes * (RX * RX / 4) - 1;     // last valid float4 of x
        const float4* __restrict__ xg4 = reinterpret_cast<const float4*>(x);
        float4 sx[NX4];
#pragma unroll
        for (int t = 0; t < NX4; ++t) {
            const int k = lane + 64 * t;
            long long gk = (long long)plane0 * (RX * RX / 4) + k;
            gk = gk < last4 ? gk : last4;                  // tail sets re-read valid memory
            sx[t] = xg4[gk];
        }
        int zoff[NZ];
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            const int e = min(lane + 64 * t, RZ * RZ - 1);
            const int u = e / RZ;
            zoff[t] = u * ZS + (e - u * RZ);
        }
        float sz[4][NZ];
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            const int plane = min(plane0 + pl, planes - 1);
            const float* __restrict__ zg = z + (size_t)plane * (RZ * RZ);
#pragma unroll
            for (int t = 0; t < NZ; ++t) sz[pl][t] = zg[min(lane + 64 * t, RZ * RZ - 1)];
        }
#pragma unroll
        for (int t = 0; t < NX4; ++t) {
            const int k = lane + 64 * t;
            if (k < 4 * RX * RX / 4) {
                const int e0 = 4 * k;
                const int pl = e0 / (RX * RX);
                const int el = e0 - pl * (RX * RX);
                const int r = el / RX;
                const int c0 = el - r * RX;
                const int o0 = pl * XP + r * XS + c0;
                const int o1 = (c0 + 2 < RX) ? o0 + 2 : o0 + XS - c0;      // (r, c0+2) or (r+1, 0)
                *reinterpret_cast<float2*>(xs + o0) = make_float2(sx[t].x, sx[t].y);
                *reinterpret_cast<float2*>(xs + o1) = make_float2(sx[t].z, sx[t].w);
            }
        }
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
#pragma unroll
            for (int t = 0; t < NZ; ++t) zs[pl * ZP + zoff[t]] = sz[pl][t];
    }
    __builtin_amdgcn_wave_barrier();

    const int p = lane >> 4, q = (lane >> 2) & 3, g = lane & 3;
    const float* xrow = xs + p * XP + (4 * q) * XS + 4 * g;
    const float* zrow = zs + p * ZP;
    float acc[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.0f;
    // Software pipeline: the LDS reads of step t+1 are issued before the FMAs of step t (two named
    // register windows wa/wb alternate); the sched_barrier at the end of every step keeps hipcc from
    // hoisting ALL reads of the fully unrolled loop to the top (it otherwise does: scratch spills).
    float zr[RZ][ZS];
    float wa[20], wb[20];
#define SMOT_LOAD_X(T, DST)                                                                 \
    {                                                                                       \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                     \
            const float4 v4 = *reinterpret_cast<const float4*>(xrow + (T) * XS + 4 * m);    \
            DST[4 * m + 0] = v4.x;                                                          \
            DST[4 * m + 1] = v4.y;                                                          \
            DST[4 * m + 2] = v4.z;                                                          \
            DST[4 * m + 3] = v4.w;                                                          \
        }                                                                                   \
        const float2 v2 = *reinterpret_cast<const float2*>(xrow + (T) * XS + 16);           \
        DST[16] = v2.x;                                                                     \
        DST[17] = v2.y;                                                                     \
    }
#define SMOT_LOAD_Z(T)                                                                      \
    {                                                                                       \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                     \
            const float4 v4 = *reinterpret_cast<const float4*>(zrow + (T) * ZS + 4 * m);    \
            zr[T][4 * m + 0] = v4.x;                                                        \
            zr[T][4 * m + 1] = v4.y;                                                        \
            zr[T][4 * m + 2] = v4.z;                                                        \
            zr[T][4 * m + 3] = v4.w;                                                        \
        }                                                                                   \
    }
/* register-only FMAs carry no chain: without this pin SelectionDAG linearises ALL of them after the   \
   last sched_barrier.  An empty volatile asm that "modifies" the accumulators orders them per step. */ \
#define SMOT_PIN_ACC()                                                                      \
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),   \
                      "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]),   \
                      "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]),   \
                      "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]));
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
                    _Pragma("unroll") for (int j = 0; j < 4; ++j)                           \
                        acc[k][j] = fmaf(CUR[j + v], zr[u][v], acc[k][j]);                  \
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
        float* o = out + (size_t)plane * (HO * HO) + (4 * q) * HO + 4 * g;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 v;
            v.x = acc[k][0];
            v.y = acc[k][1];
            v.z = acc[k][2];
            v.w = acc[k][3];
            *reinterpret_cast<float4*>(o + k * HO) = v;
        }
    }
}

// Packed kernel (default fast path, Ho == 16): the patch decomposition above on v_pk_fma_f32.
// gfx950 retires fp32 FMAs at full rate only as packed pairs (measured here: the plain-FMA patch
// kernel needs ~3.6 SIMD cycles per wave64 v_fmac; v_pk_fma_f32 does two FMAs per lane in the same
// slot).  A lane still owns a 4x4 patch, but its accumulators are the row pairs (0,2) and (1,3):
//     A0[j] = (out[4q+0][j], out[4q+2][j])      A1[j] = (out[4q+1][j], out[4q+3][j])
// At window row t both halves of A0 consume the SAME search value w[j+v] (broadcast through op_sel)
// and the tap pair ZZ[t][v] = (z[t][v], z[t-2][v]); A1 uses ZZ[t-1].  The template is therefore kept
// in LDS as 17 rows of such pairs (rows -2,-1,15,16 of z are zero), built while staging: every tap
// is written twice.  Per output the taps are still added u-major / v-minor in one fp32 FMA chain, so
// results are bit-identical to the other kernels and to the oracle's order.
typedef float v2f __attribute__((ext_vector_type(2)));

template <int RX, int RZ, int MODE>
__global__ void __launch_bounds__(64)
xcorr_dw_pk_kernel(const float* __restrict__ x, const float* __restrict__ z,
                   float* __restrict__ out, int planes) {
    constexpr int HO = RX - RZ + 1;
    static_assert(HO == 16 && RZ == 15, "packed kernel is specialised for the 30/15/16 geometry");
    constexpr int XS = 36;                  // search-plane row stride (floats)
    constexpr int XP = 1088;                // search-plane stride
    constexpr int ZR = RZ + 2;              // 17 rows of tap pairs
    constexpr int ZRS = 32;                 // floats per pair row (16 pairs, 15 used)
    constexpr int ZPP = ZR * ZRS + 16;      // 560: staggers the four planes over distinct 16-B slots
    __shared__ __attribute__((aligned(16))) float sm[4 * XP + 4 * ZPP];
    float* xs = sm;
    float* zz = sm + 4 * XP;

    const int lane = threadIdx.x;
    const int plane0 = blockIdx.x * 4;

    // ---- staging --------------------------------------------------------------------------
    constexpr int NX4 = (RX * RX + 63) / 64;             // 15 float4 per lane cover 4 planes
    constexpr int NZ = (RZ * RZ + 63) / 64;              // 4 dwords per lane per plane
    if (MODE != 2) {
        const long long last4 = (long long)planes * (RX * RX / 4) - 1;
        const float4* __restrict__ xg4 = reinterpret_cast<const float4*>(x);
        float4 sx[NX4];
#pragma unroll
        for (int t = 0; t < NX4; ++t) {
            long long gk = (long long)plane0 * (RX * RX / 4) + lane + 64 * t;
            gk = gk < last4 ? gk : last4;
            sx[t] = xg4[gk];
        }
        float sz[4][NZ];
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            const int plane = min(plane0 + pl, planes - 1);
            const float* __restrict__ zg = z + (size_t)plane * (RZ * RZ);
#pragma unroll
            for (int t = 0; t < NZ; ++t) sz[pl][t] = zg[min(lane + 64 * t, RZ * RZ - 1)];
        }
        // zero the pair image (rows/halves that have no tap stay zero)
        for (int e = lane; e < 4 * ZPP / 4; e += 64)
            reinterpret_cast<float4*>(zz)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < NX4; ++t) {
            const int k = lane + 64 * t;
            if (k < RX * RX) {
                const int e0 = 4 * k;
                const int pl = e0 / (RX * RX);
                const int el = e0 - pl * (RX * RX);
                const int r = el / RX;
                const int c0 = el - r * RX;
                const int o0 = pl * XP + r * XS + c0;
                const int o1 = (c0 + 2 < RX) ? o0 + 2 : o0 + XS - c0;
                *reinterpret_cast<float2*>(xs + o0) = make_float2(sx[t].x, sx[t].y);
                *reinterpret_cast<float2*>(xs + o1) = make_float2(sx[t].z, sx[t].w);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            const int e = lane + 64 * t;
            if (e < RZ * RZ) {
                const int u = e / RZ;
                const int v = e - u * RZ;
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) {
                    zz[pl * ZPP + u * ZRS + 2 * v] = sz[pl][t];                 // .x of row u
                    zz[pl * ZPP + (u + 2) * ZRS + 2 * v + 1] = sz[pl][t];       // .y of row u+2
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    const int p = lane >> 4, q = (lane >> 2) & 3, g = lane & 3;
    const float* xrow = xs + p * XP + (4 * q) * XS + 4 * g;
    const float* zrow = zz + p * ZPP;
    v2f a0[4], a1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a0[j] = (v2f){0.f, 0.f};
        a1[j] = (v2f){0.f, 0.f};
    }
    float wa[20], wb[20];
    v2f za[16], zb[16], zc[16];       // three pair rows rotate: (cur, prev, next)
#define SMOT_LOAD_X(T, DST)                                                                 \
    {                                                                                       \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                     \
            const float4 v4 = *reinterpret_cast<const float4*>(xrow + (T) * XS + 4 * m);    \
            DST[4 * m + 0] = v4.x;                                                          \
            DST[4 * m + 1] = v4.y;                                                          \
            DST[4 * m + 2] = v4.z;                                                          \
            DST[4 * m + 3] = v4.w;                                                          \
        }                                                                                   \
        const float2 v2 = *reinterpret_cast<const float2*>(xrow + (T) * XS + 16);           \
        DST[16] = v2.x;                                                                     \
        DST[17] = v2.y;                                                                     \
    }
#define SMOT_LOAD_ZZ(U, DST)                                                                \
    {                                                                                       \
        _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                     \
            const float4 v4 = *reinterpret_cast<const float4*>(zrow + (U) * ZRS + 4 * m);   \
            DST[2 * m] = (v2f){v4.x, v4.y};                                                 \
            DST[2 * m + 1] = (v2f){v4.z, v4.w};                                             \
        }                                                                                   \
    }
#define SMOT_PIN_ACC()                                                                      \
    asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]),                   \
                      "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]));
    // step T: window row CUR (prefetch NXT); A0 uses pair row ZT = ZZ[T], A1 uses ZP = ZZ[T-1];
    // ZN receives ZZ[T+1]
#define SMOT_STEP(T, CUR, NXT, ZT, ZP, ZN)                                                  \
    {                                                                                       \
        if ((T) + 1 < RZ + 3) SMOT_LOAD_X((T) + 1, NXT)                                     \
        if ((T) + 1 < ZR) SMOT_LOAD_ZZ((T) + 1, ZN)                                         \
        if (MODE == 1) {                                                                    \
            a0[0] += (v2f){CUR[0], CUR[17]} + ZT[0] + ZP[14];                               \
        } else {                                                                            \
            if ((T) < ZR) {                                                                 \
                _Pragma("unroll") for (int v = 0; v < RZ; ++v) {                            \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j)                           \
                        a0[j] = __builtin_elementwise_fma((v2f){CUR[j + v], CUR[j + v]}, ZT[v], a0[j]); \
                }                                                                           \
            }                                                                               \
            if ((T) >= 1) {                                                                 \
                _Pragma("unroll") for (int v = 0; v < RZ; ++v) {                            \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j)                           \
                        a1[j] = __builtin_elementwise_fma((v2f){CUR[j + v], CUR[j + v]}, ZP[v], a1[j]); \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        SMOT_PIN_ACC()                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    }
    SMOT_LOAD_X(0, wa)
    SMOT_LOAD_ZZ(0, za)
    __builtin_amdgcn_sched_barrier(0);
    // 18 steps; window registers alternate (wa, wb); pair rows rotate with period 3
    SMOT_STEP(0, wa, wb, za, zc, zb)
    SMOT_STEP(1, wb, wa, zb, za, zc)
    SMOT_STEP(2, wa, wb, zc, zb, za)
    SMOT_STEP(3, wb, wa, za, zc, zb)
    SMOT_STEP(4, wa, wb, zb, za, zc)
    SMOT_STEP(5, wb, wa, zc, zb, za)
    SMOT_STEP(6, wa, wb, za, zc, zb)
    SMOT_STEP(7, wb, wa, zb, za, zc)
    SMOT_STEP(8, wa, wb, zc, zb, za)
    SMOT_STEP(9, wb, wa, za, zc, zb)
    SMOT_STEP(10, wa, wb, zb, za, zc)
    SMOT_STEP(11, wb, wa, zc, zb, za)
    SMOT_STEP(12, wa, wb, za, zc, zb)
    SMOT_STEP(13, wb, wa, zb, za, zc)
    SMOT_STEP(14, wa, wb, zc, zb, za)
    SMOT_STEP(15, wb, wa, za, zc, zb)
    SMOT_STEP(16, wa, wb, zb, za, zc)
    SMOT_STEP(17, wb, wa, zc, zb, za)
#undef SMOT_STEP
#undef SMOT_PIN_ACC
#undef SMOT_LOAD_ZZ
#undef SMOT_LOAD_X

    const int plane = plane0 + p;
    if (plane < planes) {
        float* o = out + (size_t)plane * (HO * HO) + (4 * q) * HO + 4 * g;
        *reinterpret_cast<float4*>(o + 0 * HO) = make_float4(a0[0].x, a0[1].x, a0[2].x, a0[3].x);
        *reinterpret_cast<float4*>(o + 1 * HO) = make_float4(a1[0].x, a1[1].x, a1[2].x, a1[3].x);
        *reinterpret_cast<float4*>(o + 2 * HO) = make_float4(a0[0].y, a0[1].y, a0[2].y, a0[3].y);
        *reinterpret_cast<float4*>(o + 3 * HO) = make_float4(a1[0].y, a1[1].y, a1[2].y, a1[3].y);
    }
}

// Sixth generation: one plane per wave, 2x2 output patches (xcorr_patch1.h) — twice the waves of the kernel above.
template <int RX, int RZ>
__global__ void __launch_bounds__(64, 4)
xcorr_dw_patch1_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ out, int planes) {
    constexpr int XS = XP1_XS, ZS = XP1_ZS;
    __shared__ __attribute__((aligned(16))) float sm[RX * XS + RZ * ZS];
    float* xs = sm;
    float* zs = sm + RX * XS;
    const int lane = threadIdx.x;
    const int plane = blockIdx.x;
    (void)planes;
    constexpr int NX2 = (RX * RX / 2 + 63) / 64;         // float2 per lane (a row of 30 is 15 float2: no straddling)
    constexpr int NZ = (RZ * RZ + 63) / 64;
    const float2* __restrict__ xg2 = reinterpret_cast<const float2*>(x + (size_t)plane * RX * RX);
    const float* __restrict__ zg = z + (size_t)plane * RZ * RZ;
    float2 sx[NX2];
    float sz[NZ];
#pragma unroll
    for (int t = 0; t < NX2; ++t) sx[t] = xg2[min(lane + 64 * t, RX * RX / 2 - 1)];
#pragma unroll
    for (int t = 0; t < NZ; ++t) sz[t] = zg[min(lane + 64 * t, RZ * RZ - 1)];
#pragma unroll
    for (int t = 0; t < NX2; ++t) {
        const int k = lane + 64 * t;
        if (k < RX * RX / 2) {
            const int r = (2 * k) / RX;
            *reinterpret_cast<float2*>(xs + r * XS + (2 * k - r * RX)) = sx[t];
        }
    }
#pragma unroll
    for (int t = 0; t < NZ; ++t) {
        const int e = lane + 64 * t;
        if (e < RZ * RZ) {
            const int u = e / RZ;
            zs[u * ZS + (e - u * RZ)] = sz[t];
        }
    }
    __builtin_amdgcn_wave_barrier();
    xcorr_patch1_compute<RX, RZ>(xs, zs, lane, out, plane);
}

// Fifth generation: the correlation on v_mfma_f32_4x4x1 (xcorr_mfma.h) — same staging as above into the MFMA
// path's LDS image (row stride 40, four shifted template copies).
template <int RX, int RZ>
__global__ void __launch_bounds__(64, 2)
xcorr_dw_mfma_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ out, int planes) {
    constexpr int XP = RX * XM_XS, ZP = 4 * RZ * XM_ZC;
    __shared__ __attribute__((aligned(16))) float sm[2 * XP + 2 * ZP];
    float* xs = sm;
    float* zs = sm + 2 * XP;
    const int lane = threadIdx.x;
    const int plane0 = blockIdx.x * 2;
    constexpr int NX2 = (2 * RX * RX / 2 + 63) / 64;     // float2 per lane: a row of 30 is 15 float2
    constexpr int NZ = (2 * RZ * RZ + 63) / 64;
    const long long last2 = (long long)planes * (RX * RX / 2) - 1;
    const float2* __restrict__ xg2 = reinterpret_cast<const float2*>(x);
    float2 sx[NX2];
#pragma unroll
    for (int t = 0; t < NX2; ++t) {
        long long gk = (long long)plane0 * (RX * RX / 2) + lane + 64 * t;
        gk = gk < last2 ? gk : last2;                      // odd plane counts: re-read valid memory
        sx[t] = xg2[gk];
    }
    const long long lastz = (long long)planes * (RZ * RZ) - 1;
    float sz[NZ];
#pragma unroll
    for (int t = 0; t < NZ; ++t) {
        long long ge = (long long)plane0 * (RZ * RZ) + lane + 64 * t;
        ge = ge < lastz ? ge : lastz;
        sz[t] = z[ge];
    }
    xm_zero_template_pad<RZ>(zs, lane);
    xm_zero_template_pad<RZ>(zs + ZP, lane);
#pragma unroll
    for (int t = 0; t < NX2; ++t) {
        const int k = lane + 64 * t;
        if (k < 2 * RX * RX / 2) {
            const int e0 = 2 * k;
            const int pl = e0 / (RX * RX);
            const int el = e0 - pl * (RX * RX);
            const int r = el / RX;
            *reinterpret_cast<float2*>(xs + pl * XP + r * XM_XS + (el - r * RX)) = sx[t];
        }
    }
#pragma unroll
    for (int t = 0; t < NZ; ++t) {
        const int e = lane + 64 * t;
        if (e < 2 * RZ * RZ) {
            const int pl = e / (RZ * RZ);
            const int el = e - pl * (RZ * RZ);
            const int u = el / RZ;
            xm_store_template<RZ>(zs + pl * ZP, u, el - u * RZ, sz[t]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    xcorr_mfma_compute<RX, RZ>(xs, zs, lane, out, plane0, planes);
}

// variant: see knobs.h (XV_*).  Returns false when `variant` names no kernel of this file.
bool launch_xcorr_variant(int variant, const float* x, const float* z, float* out, int planes, hipStream_t st) {
    const dim3 g4((planes + 3) / 4), g2((planes + 1) / 2), b64(64);
    switch (variant) {
        case XV_WAVE: hipLaunchKernelGGL((xcorr_dw_wave_kernel<30, 15>), g4, dim3(256), 0, st, x, z, out, planes); return true;
        case XV_PATCH: hipLaunchKernelGGL((xcorr_dw_patch_kernel<30, 15, 0>), g4, b64, 0, st, x, z, out, planes); return true;
        case XV_PK: hipLaunchKernelGGL((xcorr_dw_pk_kernel<30, 15, 0>), g4, b64, 0, st, x, z, out, planes); return true;
        case XV_ONE: hipLaunchKernelGGL((xcorr_dw_patch1_kernel<30, 15>), dim3(planes), b64, 0, st, x, z, out, planes); return true;
        case XV_MFMA: hipLaunchKernelGGL((xcorr_dw_mfma_kernel<30, 15>), g2, b64, 0, st, x, z, out, planes); return true;
        default: return false;
    }
}

}  // namespace smot
