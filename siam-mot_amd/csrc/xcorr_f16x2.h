// The 30x30 * 15x15 -> 16x16 depthwise cross-correlation of one plane by one wave ON THE MATRIX PIPE (round 6).
//
// The fp32 FMA phase (xcorr_patch1.h) is bound by vector issue: 900 v_fmac_f32 per lane and plane at 4 cycles, 3,600 cycles
// per plane and SIMD.  Here the correlation is a chain of 15 small GEMMs, one per template row i,
//   out[y][x] = sum_i sum_c X[y+i][c] * T_i[c][x],     T_i[c][x] = Z[i][c - x]  (0 <= c - x < 15, else 0),
// A_i = rows i .. i+15 of the search plane (16 x 32, columns 30, 31 zero), B_i = the Toeplitz matrix of template row i
// (32 x 16), on v_mfma_f32_16x16x32_f16 with TWO-PART fp16 operands of power-of-two-scaled values (a = a1 + a2, a1 =
// RNE11(a), a2 = RNE11(a - a1): error <= 2^-23 |a| or 2^-37 of the plane's largest value) and three part products
// (x1 z1 + x1 z2 + x2 z1; x2 z2 <= 2^-22 of a term), accumulated in fp32: 45 matrix instructions (720 cycles) per plane.
// Error against an fp64 evaluation 1.0e-7 * sum |x z| — the fp32 FMA chain's is 2.3e-7 (tools/ubench/xcorr_f16x2.hip,
// profiles/r06_ubench_xcorr_f16x2.jsonl).
//
// What the operands cost is decided in LDS (MI355X: a vector read that is not aligned to its size is served lane by
// lane — ~64 LDS cycles per ds_read_b128 instead of 4; the first prototype lost to the FMA phase on exactly that):
//   * A: the two half images overwrite the plane's fp32 image IN PLACE, row pitch unchanged (40 dwords: part 1 in dwords
//     0..15 of a row, part 2 in 16..31) — lane (y, kq) reads 16 aligned bytes, and a pitch of 40 dwords is conflict-free
//     for the lane groups of ds_read_b128;
//   * B: a lane's window is eight consecutive halves of the zero-padded template row at a LANE-DEPENDENT half-word
//     offset 8 kq - x.  Every row is therefore kept twice — as dwords of halves (2p, 2p+1) and of halves (2p+1, 2p+2) —
//     and a lane reads FOUR ALIGNED DWORDS from the copy its window's parity selects (two ds_read2_b32).  Rows are
//     clamped to the 32 halves in which a window can meet a template value (windows further out are all zeros either
//     way); the odd copies sit 16 banks from the even ones, so the two kinds of lane never meet on a bank.
//     2 copies x 2 parts x 15 rows x 64 B = 3,840 B (+ 64 B of offset) per plane.
// The template's rows are built while the search plane is still being pooled (xh_template_*), the plane's half images
// after the pooling barrier (xh_correlate).
#pragma once
#include "smot_common.h"
#include "xcorr_patch1.h"

namespace smot {

typedef float xh_f32x4 __attribute__((ext_vector_type(4)));
typedef float xh_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 xh_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 xh_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 xh_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned xh_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned xh_u32x2 __attribute__((ext_vector_type(2)));

constexpr int XH_TZ_ROW = 64;                              // bytes of a clamped Toeplitz row (32 halves)
constexpr int XH_TZ_ODD = (2 * 15 * 16 + 16) * 4;          // byte offset of the odd copies: 16 dwords mod 32 from the even ones
constexpr int XH_TZ_BYTES = XH_TZ_ODD + 2 * 15 * XH_TZ_ROW;
constexpr int XH_TZ_FLOATS = XH_TZ_BYTES / 4;              // 976
static_assert(XH_TZ_BYTES % 16 == 0, "zero fill by 16-byte stores");

__device__ __forceinline__ float xh_wave_absmax(float v) {           // (NaN ignored: the response's plane maximum)
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
// the operands' maximum is taken over MAGNITUDE BITS (v & 0x7fffffff as unsigned): NaN > inf > every finite value, so one
// word says both how to scale the plane and whether it holds a non-finite value at all
__device__ __forceinline__ unsigned xh_mag(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned xh_wave_umax(unsigned v) {
#define SMOT_ROR(N) (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + (N), 0xf, 0xf, false)
    v = max(v, SMOT_ROR(8));
    v = max(v, SMOT_ROR(4));
    v = max(v, SMOT_ROR(2));
    v = max(v, SMOT_ROR(1));
#undef SMOT_ROR
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
// power of two s with m * s in [2^13, 2^14) (exponent clamped to +-100: magnitudes 2^-87 .. 2^113 keep full precision,
// denormal planes are lifted by 2^100) and its inverse; m = 0: 1 / 1; m = inf or NaN: 1 / NaN — a plane (or template) with a
// non-finite value gives an all-NaN response plane (the fp16 split has no meaning for it; the reference leaves NaN / inf in
// the outputs whose window covers the value, and the towers' GroupNorm makes a NaN track of either)
__device__ __forceinline__ void xh_pow2_scale(unsigned mbits, float* s, float* inv) {
    const int e = (int)(mbits >> 23);
    *s = 1.0f;
    *inv = 1.0f;
    if (e == 255) {
        *inv = __uint_as_float(0x7FC00000u);
    } else if (mbits != 0u) {
        int k = 140 - e;
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
        *s = __uint_as_float((unsigned)(127 + k) << 23);
        *inv = __uint_as_float((unsigned)(127 - k) << 23);
    }
}
#define SMOT_XH_CVT(A, B) __builtin_bit_cast(unsigned, __builtin_convertvector((xh_f32x2){A, B}, xh_f16x2))
// -I for v_mfma_f32_4x4x4_16b_f16: D = C - B for the lane's own four values (exact: the residual of the split)
__device__ __forceinline__ xh_f16x4 xh_neg_identity(int lane) {
    return __builtin_bit_cast(xh_f16x4, (xh_u32x2){(lane & 3) == 0 ? 0x0000BC00u : ((lane & 3) == 1 ? 0xBC000000u : 0u),
                                                   (lane & 3) == 2 ? 0x0000BC00u : ((lane & 3) == 3 ? 0xBC000000u : 0u)});
}

// The wave's template (15 x 15 floats, contiguous in global memory): lane (r0 = lane / 16, j = lane % 16) takes Z[r0 + 4 t][j],
// t = 0..3 (row 15 and column 15 do not exist: 0).  Loads only — call early.
__device__ __forceinline__ void xh_template_load(const float* __restrict__ zg, int lane, float zq[4]) {
    const int j = lane & 15, r0 = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool live = j < 15 && r0 + 4 * t < 15;
        zq[t] = zg[live ? (r0 + 4 * t) * 15 + j : 0];          // (the lanes without a value are zeroed by xh_template_store: no
    }                                                          // use of the loaded registers here, no wait before the caller's next phase)
}
// ... scaled, split and written as the even / odd Toeplitz rows of both parts into `tz` (XH_TZ_BYTES, 16-byte aligned; the
// wave's own area: no barrier).  Returns the inverse of the template's scale (wave-uniform).
__device__ __forceinline__ float xh_template_store(const float zq[4], unsigned char* tz, int lane) {
    const int j = lane & 15, r0 = lane >> 4;
    float zv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) zv[t] = (j < 15 && r0 + 4 * t < 15) ? zq[t] : 0.0f;
    const unsigned mz = max(max(xh_mag(zv[0]), xh_mag(zv[1])), max(xh_mag(zv[2]), xh_mag(zv[3])));
    float sz, isz;
    xh_pow2_scale(xh_wave_umax(mz), &sz, &isz);
    {   // zeros under the rows (their pads are what windows beside the template read)
        xh_u32x4* z4 = reinterpret_cast<xh_u32x4*>(tz);
#pragma unroll
        for (int e = 0; e < (XH_TZ_BYTES / 16 + 63) / 64; ++e)
            if (lane + 64 * e < XH_TZ_BYTES / 16) z4[lane + 64 * e] = (xh_u32x4){0u, 0u, 0u, 0u};
    }
    xh_f32x4 v = {zv[0] * sz, zv[1] * sz, zv[2] * sz, zv[3] * sz};
    const unsigned a0 = SMOT_XH_CVT(v[0], v[1]), a1 = SMOT_XH_CVT(v[2], v[3]);
    v = __builtin_amdgcn_mfma_f32_4x4x4f16(xh_neg_identity(lane), __builtin_bit_cast(xh_f16x4, (xh_u32x2){a0, a1}), v, 0, 0, 0);
    const unsigned b0 = SMOT_XH_CVT(v[0], v[1]), b1 = SMOT_XH_CVT(v[2], v[3]);
    const unsigned h1[4] = {a0 & 0xffffu, a0 >> 16, a1 & 0xffffu, a1 >> 16};
    const unsigned h2[4] = {b0 & 0xffffu, b0 >> 16, b1 & 0xffffu, b1 >> 16};
    // Template value j of a row sits at local half 8 + j of the even copy and 7 + j of the odd one.  A lane packs its value
    // with its right neighbour's (row rotation by 15 = the value of lane j + 1, lane 15 — which holds 0 — gets lane 0's):
    // even j -> dword (8 + j) / 2 of the even copy = (Z[j], Z[j+1]); odd j -> dword (7 + j) / 2 of the odd copy = (Z[j],
    // Z[j+1]); lane 15 -> dword 3 of the odd copy = (0, Z[0]).  One aligned dword store per lane, part and row.
    const int dw = (j & 1) ? (j == 15 ? 3 : (7 + j) >> 1) : (8 + j) >> 1;
    unsigned* dst = reinterpret_cast<unsigned*>(tz + ((j & 1) ? XH_TZ_ODD : 0) + r0 * XH_TZ_ROW + dw * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned n1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)h1[t], 0x12F, 0xf, 0xf, false);
        const unsigned n2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)h2[t], 0x12F, 0xf, 0xf, false);
        if (r0 + 4 * t < 15) {
            dst[t * 4 * (XH_TZ_ROW / 4)] = h1[t] | (n1 << 16);
            dst[(15 + t * 4) * (XH_TZ_ROW / 4)] = h2[t] | (n2 << 16);
        }
    }
    return isz;
}

// The correlation proper.  `xs`: the plane's fp32 image at row pitch XP1_XS (overwritten by its half images); `tz`: the rows
// xh_template_store wrote (same wave); `isz`: what it returned.  Output layout and the plane maximum as xcorr_patch1_compute.
template <int RX, int RZ>
__device__ __forceinline__ void xh_correlate(float* xs, const unsigned char* tz, float isz, int lane, float* __restrict__ out,
                                             int plane, float* __restrict__ pmax) {
    static_assert(RX == 30 && RZ == 15 && XP1_XS == 40, "the 30 / 15 / 16 geometry at a row pitch of 40 dwords");
    constexpr int XS = XP1_XS;
    const xh_f16x4 negI = xh_neg_identity(lane);
    // ---- the search plane: lane (r0 = lane / 16, c2 = lane % 16) takes the column pair (2 c2, 2 c2 + 1) of rows r0 + 4 t
    // (columns 30, 31 and rows 30, 31 are not part of the plane: 0 — the image's words there are whatever pooling left)
    const int c2 = lane & 15, r0 = lane >> 4;
    float* xrow = xs + r0 * XS + 2 * c2;
    xh_f32x4 xv[4];
    unsigned m = 0u;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const bool live = c2 < 15 && (t < 7 || r0 < 2);
        const float2 v = *reinterpret_cast<const float2*>(xrow + t * 4 * XS);
        xv[t >> 1][(t & 1) * 2] = live ? v.x : 0.0f;
        xv[t >> 1][(t & 1) * 2 + 1] = live ? v.y : 0.0f;
        m = max(m, live ? max(xh_mag(v.x), xh_mag(v.y)) : 0u);
    }
    float sx, isx;
    xh_pow2_scale(xh_wave_umax(m), &sx, &isx);
    unsigned* xh = reinterpret_cast<unsigned*>(xs) + r0 * XS + c2;          // row r: dwords 0..15 part 1, 16..31 part 2
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        xh_f32x4 v = xv[g] * sx;
        const unsigned a0 = SMOT_XH_CVT(v[0], v[1]), a1 = SMOT_XH_CVT(v[2], v[3]);
        v = __builtin_amdgcn_mfma_f32_4x4x4f16(negI, __builtin_bit_cast(xh_f16x4, (xh_u32x2){a0, a1}), v, 0, 0, 0);
        const unsigned b0 = SMOT_XH_CVT(v[0], v[1]), b1 = SMOT_XH_CVT(v[2], v[3]);
        // (LDS operations of a wave execute in order and the scale depends on every read above: reads before writes)
        xh[(2 * g) * 4 * XS] = a0;
        xh[(2 * g) * 4 * XS + 16] = b0;
        if (g < 3 || r0 < 2) {
            xh[(2 * g + 1) * 4 * XS] = a1;
            xh[(2 * g + 1) * 4 * XS + 16] = b1;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 15 template rows x 3 part products
    const int x = lane & 15, kq = lane >> 4;
    const unsigned a_addr = (unsigned)(size_t)xs + (unsigned)(x * XS * 4 + kq * 16);          // row y = x of A; + i * 160 (+ 64: part 2)
    int sw = 8 * kq - x + 16;                      // first half of the lane's window in the 48-half padded row
    sw = sw < 8 ? 8 : (sw > 31 ? 31 : sw);         // (windows clamped here hold zeros only, like the ones they stand for)
    const int loc = sw - 8;
    const unsigned b_addr = (unsigned)(size_t)tz + (unsigned)((loc & 1) ? XH_TZ_ODD + ((loc - 1) >> 1) * 4 : (loc >> 1) * 4);
    const unsigned b_addr2 = b_addr + 15 * XH_TZ_ROW;
    xh_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
    // operands of row i + 1 are requested before the three instructions of row i are issued (two register sets)
    xh_u32x4 A1[2], A2[2];
    xh_u32x2 B1[2][2], B2[2][2];
#define SMOT_XH_RD(I, S)                                                                                           \
    asm volatile("ds_read_b128 %0, %6 offset:%9\n\tds_read_b128 %1, %6 offset:%10\n\t"                             \
                 "ds_read2_b32 %2, %7 offset0:%11 offset1:%12\n\tds_read2_b32 %3, %7 offset0:%13 offset1:%14\n\t"  \
                 "ds_read2_b32 %4, %8 offset0:%11 offset1:%12\n\tds_read2_b32 %5, %8 offset0:%13 offset1:%14"      \
                 : "=&v"(A1[S]), "=&v"(A2[S]), "=&v"(B1[S][0]), "=&v"(B1[S][1]), "=&v"(B2[S][0]), "=&v"(B2[S][1])  \
                 : "v"(a_addr), "v"(b_addr), "v"(b_addr2), "n"((I) * XS * 4), "n"((I) * XS * 4 + 64),              \
                   "n"((I) * 16), "n"((I) * 16 + 1), "n"((I) * 16 + 2), "n"((I) * 16 + 3) : "memory");
#define SMOT_XH_B(X, S) __builtin_bit_cast(xh_f16x8, (xh_u32x4){X[S][0][0], X[S][0][1], X[S][1][0], X[S][1][1]})
#define SMOT_XH_MM(S)                                                                                              \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xh_f16x8, A2[S]), SMOT_XH_B(B1, S), acc0, 0, 0, 0); \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xh_f16x8, A1[S]), SMOT_XH_B(B2, S), acc1, 0, 0, 0); \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xh_f16x8, A1[S]), SMOT_XH_B(B1, S), acc2, 0, 0, 0);
#define SMOT_XH_STEP(I)                                                                                            \
    if ((I) + 1 < 15) { SMOT_XH_RD((I) + 1, ((I) + 1) & 1) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); }    \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    SMOT_XH_MM((I) & 1)                                                                                            \
    __builtin_amdgcn_sched_barrier(0);
    SMOT_XH_RD(0, 0)
    SMOT_XH_STEP(0) SMOT_XH_STEP(1) SMOT_XH_STEP(2) SMOT_XH_STEP(3) SMOT_XH_STEP(4) SMOT_XH_STEP(5) SMOT_XH_STEP(6) SMOT_XH_STEP(7)
    SMOT_XH_STEP(8) SMOT_XH_STEP(9) SMOT_XH_STEP(10) SMOT_XH_STEP(11) SMOT_XH_STEP(12) SMOT_XH_STEP(13) SMOT_XH_STEP(14)
#undef SMOT_XH_STEP
#undef SMOT_XH_MM
#undef SMOT_XH_B
#undef SMOT_XH_RD
    // D: lane (x = lane % 16, g = lane / 16) holds rows y = 4 g .. 4 g + 3; the small part products first
    // (the two inverse scales one after the other: their product may leave fp32's range when neither result does)
    float* o = out + (size_t)plane * 256 + kq * 4 * 16 + x;
    float mo = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = (((acc0[r] + acc1[r]) + acc2[r]) * isx) * isz;
        o[r * 16] = v;
        mo = fmaxf(mo, fabsf(v));
    }
    if (pmax != nullptr) {
        mo = xh_wave_absmax(mo);
        if (lane == 0) pmax[plane] = mo;
    }
}

// ---- the LEAN layout: the same correlation in 5,760 B of LDS per plane instead of 9,024 — three workgroups per CU ----------
// (at more than two workgroups per CU and launch — 100 tracks, the 256-channel configuration — the third resident
// workgroup is worth more than the reads it costs)
//   * image rows of 32 dwords (128 B, the fp32 row's own size): a row holds the 8 sixteen-byte chunks (part, kq) of both
//     half images ROTATED by the row index — chunk c of row r at position (c + r) mod 8 — which makes the A reads
//     conflict-free at this pitch (searched: tools notes in DESIGN.md);
//   * ONE copy of the Toeplitz rows (dwords of halves (2p, 2p+1)): a lane reads FIVE aligned dwords and, when its window
//     starts at an odd half, funnel-shifts neighbouring dwords by 16 bits (v_alignbit_b32 with a per-lane shift of 0 or 16).
constexpr int XL_XS = 32;
constexpr int XL_TZ_BYTES = 2 * 15 * XH_TZ_ROW;            // 1,920
constexpr int XL_TZ_FLOATS = XL_TZ_BYTES / 4;

__device__ __forceinline__ float xl_template_store(const float zq[4], unsigned char* tz, int lane) {
    const int j = lane & 15, r0 = lane >> 4;
    float zv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) zv[t] = (j < 15 && r0 + 4 * t < 15) ? zq[t] : 0.0f;
    const unsigned mz = max(max(xh_mag(zv[0]), xh_mag(zv[1])), max(xh_mag(zv[2]), xh_mag(zv[3])));
    float sz, isz;
    xh_pow2_scale(xh_wave_umax(mz), &sz, &isz);
    {
        xh_u32x4* z4 = reinterpret_cast<xh_u32x4*>(tz);
#pragma unroll
        for (int e = 0; e < (XL_TZ_BYTES / 16 + 63) / 64; ++e)
            if (lane + 64 * e < XL_TZ_BYTES / 16) z4[lane + 64 * e] = (xh_u32x4){0u, 0u, 0u, 0u};
    }
    xh_f32x4 v = {zv[0] * sz, zv[1] * sz, zv[2] * sz, zv[3] * sz};
    const unsigned a0 = SMOT_XH_CVT(v[0], v[1]), a1 = SMOT_XH_CVT(v[2], v[3]);
    v = __builtin_amdgcn_mfma_f32_4x4x4f16(xh_neg_identity(lane), __builtin_bit_cast(xh_f16x4, (xh_u32x2){a0, a1}), v, 0, 0, 0);
    const unsigned b0 = SMOT_XH_CVT(v[0], v[1]), b1 = SMOT_XH_CVT(v[2], v[3]);
    const unsigned h1[4] = {a0 & 0xffffu, a0 >> 16, a1 & 0xffffu, a1 >> 16};
    const unsigned h2[4] = {b0 & 0xffffu, b0 >> 16, b1 & 0xffffu, b1 >> 16};
    // even j: dword (8 + j) / 2 = (Z[j], Z[j+1]) — the right neighbour's value by a row rotation; odd lanes store nothing
    unsigned* dst = reinterpret_cast<unsigned*>(tz + r0 * XH_TZ_ROW + ((8 + j) >> 1) * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned n1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)h1[t], 0x12F, 0xf, 0xf, false);
        const unsigned n2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)h2[t], 0x12F, 0xf, 0xf, false);
        if ((j & 1) == 0 && r0 + 4 * t < 15) {
            dst[t * 4 * (XH_TZ_ROW / 4)] = h1[t] | (n1 << 16);
            dst[(15 + t * 4) * (XH_TZ_ROW / 4)] = h2[t] | (n2 << 16);
        }
    }
    return isz;
}

// `xs`: the plane's fp32 image at row pitch XL_XS = 32 floats, 128-BYTE ALIGNED (30 rows; overwritten by its half images)
template <int RX, int RZ>
__device__ __forceinline__ void xl_correlate(float* xs, const unsigned char* tz, float isz, int lane, float* __restrict__ out,
                                             int plane, float* __restrict__ pmax) {
    static_assert(RX == 30 && RZ == 15, "the 30 / 15 / 16 geometry");
    constexpr int XS = XL_XS;
    const xh_f16x4 negI = xh_neg_identity(lane);
    const int c2 = lane & 15, r0 = lane >> 4;
    const bool tail = r0 < 2;                                   // rows 28 + r0 exist
    float* xrow = xs + r0 * XS + 2 * c2;
    xh_f32x4 xv[4];
    unsigned m = 0u;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const bool live = c2 < 15 && (t < 7 || tail);
        const float2 v = *reinterpret_cast<const float2*>((t < 7 || tail) ? xrow + t * 4 * XS : xrow);
        xv[t >> 1][(t & 1) * 2] = live ? v.x : 0.0f;
        xv[t >> 1][(t & 1) * 2 + 1] = live ? v.y : 0.0f;
        m = max(m, live ? max(xh_mag(v.x), xh_mag(v.y)) : 0u);
    }
    float sx, isx;
    xh_pow2_scale(xh_wave_umax(m), &sx, &isx);
    // pair c2 of row r = r0 + 4 t: part 1 is dword c2 & 3 of chunk c2 >> 2, part 2 of chunk 4 + (c2 >> 2); chunk c of row r
    // stands at position (c + r) & 7.  (c2 >> 2) + r0 + 4 t: even t -> cb, odd t -> cb ^ 4; part 2 is always the other one.
    const int cb = ((c2 >> 2) + r0) & 7;
    unsigned* xe = reinterpret_cast<unsigned*>(xs) + r0 * XS + cb * 4 + (c2 & 3);           // part 1 of even t, part 2 of odd t
    unsigned* xo = reinterpret_cast<unsigned*>(xs) + r0 * XS + (cb ^ 4) * 4 + (c2 & 3);     // part 2 of even t, part 1 of odd t
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        xh_f32x4 v = xv[g] * sx;
        const unsigned a0 = SMOT_XH_CVT(v[0], v[1]), a1 = SMOT_XH_CVT(v[2], v[3]);
        v = __builtin_amdgcn_mfma_f32_4x4x4f16(negI, __builtin_bit_cast(xh_f16x4, (xh_u32x2){a0, a1}), v, 0, 0, 0);
        const unsigned b0 = SMOT_XH_CVT(v[0], v[1]), b1 = SMOT_XH_CVT(v[2], v[3]);
        xe[(2 * g) * 4 * XS] = a0;
        xo[(2 * g) * 4 * XS] = b0;
        if (g < 3 || tail) {
            xo[(2 * g + 1) * 4 * XS] = a1;
            xe[(2 * g + 1) * 4 * XS] = b1;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int x = lane & 15, kq = lane >> 4;
    const unsigned a_base = (unsigned)(size_t)xs + (unsigned)(x * XS * 4);        // row y = x of A; + i * 128 by the offset field
    const int c0 = kq + x;                                                       // chunk kq of row x + i stands at (c0 + i) & 7
    int sw = 8 * kq - x + 16;
    sw = sw < 8 ? 8 : (sw > 31 ? 31 : sw);
    const int loc = sw - 8;
    const unsigned sh = (unsigned)(loc & 1) * 16u;
    const unsigned b_addr = (unsigned)(size_t)tz + (unsigned)((loc >> 1) * 4);
    const unsigned b_addr2 = b_addr + 15 * XH_TZ_ROW;
    xh_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
    xh_u32x4 A1[2], A2[2];
    xh_u32x2 B1[2][2], B2[2][2];
    unsigned B1t[2], B2t[2];
#define SMOT_XL_RD(I, S)                                                                                           \
    {                                                                                                              \
        const unsigned o1 = (unsigned)((c0 + (I)) & 7) << 4;                                                       \
        const unsigned a1_ = a_base + o1, a2_ = a_base + (o1 ^ 64u);                                               \
        asm volatile("ds_read_b128 %0, %8 offset:%12\n\tds_read_b128 %1, %9 offset:%12\n\t"                        \
                     "ds_read2_b32 %2, %10 offset0:%13 offset1:%14\n\tds_read2_b32 %3, %10 offset0:%15 offset1:%16\n\t" \
                     "ds_read_b32 %4, %10 offset:%17\n\t"                                                          \
                     "ds_read2_b32 %5, %11 offset0:%13 offset1:%14\n\tds_read2_b32 %6, %11 offset0:%15 offset1:%16\n\t" \
                     "ds_read_b32 %7, %11 offset:%17"                                                              \
                     : "=&v"(A1[S]), "=&v"(A2[S]), "=&v"(B1[S][0]), "=&v"(B1[S][1]), "=&v"(B1t[S]), "=&v"(B2[S][0]),   \
                       "=&v"(B2[S][1]), "=&v"(B2t[S])                                                              \
                     : "v"(a1_), "v"(a2_), "v"(b_addr), "v"(b_addr2), "n"((I) * XS * 4), "n"((I) * 16), "n"((I) * 16 + 1), \
                       "n"((I) * 16 + 2), "n"((I) * 16 + 3), "n"((I) * 64 + 16) : "memory");                       \
    }
#define SMOT_XL_AL(HI, LO) __builtin_amdgcn_alignbit(HI, LO, sh)
#define SMOT_XL_B(X, XT, S) __builtin_bit_cast(xh_f16x8, (xh_u32x4){SMOT_XL_AL(X[S][0][1], X[S][0][0]), SMOT_XL_AL(X[S][1][0], X[S][0][1]), \
                                                                     SMOT_XL_AL(X[S][1][1], X[S][1][0]), SMOT_XL_AL(XT[S], X[S][1][1])})
#define SMOT_XL_MM(S)                                                                                              \
    {                                                                                                              \
        const xh_f16x8 b1_ = SMOT_XL_B(B1, B1t, S), b2_ = SMOT_XL_B(B2, B2t, S);                                   \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xh_f16x8, A2[S]), b1_, acc0, 0, 0, 0);    \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xh_f16x8, A1[S]), b2_, acc1, 0, 0, 0);    \
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xh_f16x8, A1[S]), b1_, acc2, 0, 0, 0);    \
    }
#define SMOT_XL_STEP(I)                                                                                            \
    if ((I) + 1 < 15) { SMOT_XL_RD((I) + 1, ((I) + 1) & 1) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); }    \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    SMOT_XL_MM((I) & 1)                                                                                            \
    __builtin_amdgcn_sched_barrier(0);
    SMOT_XL_RD(0, 0)
    SMOT_XL_STEP(0) SMOT_XL_STEP(1) SMOT_XL_STEP(2) SMOT_XL_STEP(3) SMOT_XL_STEP(4) SMOT_XL_STEP(5) SMOT_XL_STEP(6) SMOT_XL_STEP(7)
    SMOT_XL_STEP(8) SMOT_XL_STEP(9) SMOT_XL_STEP(10) SMOT_XL_STEP(11) SMOT_XL_STEP(12) SMOT_XL_STEP(13) SMOT_XL_STEP(14)
#undef SMOT_XL_STEP
#undef SMOT_XL_MM
#undef SMOT_XL_B
#undef SMOT_XL_AL
#undef SMOT_XL_RD
    float* o = out + (size_t)plane * 256 + kq * 4 * 16 + x;
    float mo = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = (((acc0[r] + acc1[r]) + acc2[r]) * isx) * isz;
        o[r * 16] = v;
        mo = fmaxf(mo, fabsf(v));
    }
    if (pmax != nullptr) {
        mo = xh_wave_absmax(mo);
        if (lane == 0) pmax[plane] = mo;
    }
}
#undef SMOT_XH_CVT

}  // namespace smot
