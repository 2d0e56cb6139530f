// K1+K2 fused, fourth generation — plan-driven, barrier-free: one wave = one PAIR of planes, from row loads to response.
//
// Replaces the same reference code as sr_xcorr.hip (EMM/track_core.py:49-53: TrackUtils.pad_feature,
// track_utils.py:87-107 -> SRPooler on the search regions, EMM/sr_pool.py:53-91 -> xcorr_depthwise, EMM/xcorr.py:37-46)
// with the same arithmetic, term by term (bit-identical responses; tests/test_hip_parity.py A/Bs the generations).
//
// What the third generation (sr_xcorr_fused9_kernel) spent its time on, measured in round 4 by the dispatch's own
// timestamps (measure/fused_ab2.py, 30 tracks, order hint on): 15.7 us, of which 8.8 us before the first FMA of the
// correlation — a chain  roi ranking / hint -> sample tables (two IEEE divisions per axis, LDS image, barrier) -> 60
// row loads per wave, every offset and weight through v_readlane -> barrier -> correlation  that all sixteen waves of a
// CU walk in lock-step, so that nothing overlaps anything.  This generation removes the chain's links instead of
// shortening them:
//   * PLANS.  Everything a workgroup needs to know about its roi — rank in the cost-sorted order, FPN level, window,
//     and BOTH finished sample tables — is written once per roi by roi_plan_kernel (by the extraction launch that
//     creates the search regions, one frame earlier; or by a stand-alone launch) as 2 KB in the constant address
//     space.  A consumer wave reads header + row offsets with scalar loads: ONE scalar-cache round trip before its
//     first feature load, no ranking, no division, no LDS table, no barrier.
//   * ROW OFFSETS AND VERTICAL WEIGHTS ARRIVE IN SGPRs (s_load_dwordx16 of the plan's y table): the 120 v_readlane
//     per batch of generation 3 are gone; a row load is `buffer_load voffset=lane column, soffset=SGPR`, a vertical
//     tap `v_fmac v, s, v`.
//   * ONE WAVE OWNS ITS TWO PLANES END TO END (lanes 0..31 plane A, 32..63 plane B while pooling; all 64 lanes on
//     one plane at a time in the correlation).  No __syncthreads anywhere: LDS operations of one wave execute in
//     order.  Waves drift apart, so one wave's correlation FMAs fill the issue slots another wave's load wait leaves.
//   * 256 VGPRs per wave (two waves per SIMD is what the fp32 FMA rate needs, xcorr_patch1.h): ALL row loads of a
//     plane pair are in flight at once (two batches of 60), and the correlation runs its prefetching form.
#include "roi_common.h"
#include "xcorr_patch1.h"
#include "knobs.h"
#include <type_traits>

#define SMOT_PLAN_FLOATS 1024      // one roi plan: header, band list, both sample tables

namespace smot {

// ---- roi plan: 1024 dwords per roi, entry k = the roi of rank k in the cost-sorted order ---------------------------
constexpr int PLAN_DW = SMOT_PLAN_FLOATS;
// header (16 dwords)
constexpr int PL_N = 0, PL_LVL = 1, PL_XMIN = 2, PL_WW = 3, PL_EMPTY = 4, PL_PLANE_BYTES = 5, PL_MODE = 6, PL_GW = 7;
constexpr int PL_RPI = 8, PL_WS_BYTES = 9, PL_ROW_BYTES = 10, PL_RPI_ROW_BYTES = 11, PL_RECIP = 12;
constexpr int PL_BANDS = 16;    // [8]{global byte offset of the band's first window row inside a plane, row-block loads}
constexpr int PL_YT = 32;       // [64][4] {LDS byte offset of the low row, of the high row (inside the band's window image), weight lo, hi}
constexpr int PL_YG = 288;      // [64][2] global byte offsets of the low / high row inside a plane (the per-row path)
constexpr int PL_X = 416;       // [60][4] {window column lo, hi, weight lo, weight hi}
static_assert(PL_X + 60 * 4 <= PLAN_DW, "plan layout");
// how a wave pools its roi (chosen by the plan writer from the window's shape)
constexpr int PM_NARROW = 0;    // window <= 32 columns and it fits the LDS image whole: lane = (half of the pooled rows, column)
constexpr int PM_WIDE = 1;      // window <= 64 columns in two bands of 15 pooled rows: lane = column
constexpr int PM_ROWS = 2;      // anything else: per-row dword loads in 64-column chunks (generation 3's rare path)

constexpr int FX10_WAVES = 8;                       // waves per workgroup = planes of an 8-channel group
constexpr int FX10_WBUF = 1936;                     // floats of window image per wave (one band; lies over the pooled plane)
constexpr int FX10_IBM = 6, FX10_IBM1 = 5;          // row-block loads of the first / second band, at most

typedef int v16i_t __attribute__((ext_vector_type(16)));
typedef int v4i10_t __attribute__((ext_vector_type(4)));
typedef int v2i10_t __attribute__((ext_vector_type(2)));

// 16 / 2 dwords at a wave-uniform address through the constant address space: one s_load
__device__ __forceinline__ v16i_t sload16(const int* base, int dw) {
    typedef v16i_t __attribute__((aligned(4))) v16i_a4;
    return *reinterpret_cast<const __attribute__((address_space(4))) v16i_a4*>(
        reinterpret_cast<unsigned long long>(base) + 4ull * (unsigned)dw);
}

// One wave per roi.  Ranks the rois by window-width class (the cost order of generation 3: class descending, roi
// ascending — any bijection gives the same responses), builds both sample tables in the reference's rounding
// sequence (axis_sample, roi_common.h: the legacy ROIAlign against the PADDED extent, cells re-expressed in the real
// map, zero weights in the virtual border), decides how the roi's window is brought on chip (whole, or in two bands of
// window rows, fetched by 16-byte loads into an LDS image) and writes plan[rank].
template <int RX, int G>
__global__ void __launch_bounds__(64) roi_plan_kernel(LevelParams P, const float* __restrict__ sr,
                                                      const float* __restrict__ boxes, int NT, int* __restrict__ plans) {
    constexpr int NS = RX * G;
    static_assert(NS <= 64, "one lane per sample");
    const int lane = threadIdx.x;
    const int x = blockIdx.x;
    int rank = x;
    if (NT >= 2 && NT <= 256) {
        unsigned long long mask[4][3];
        int cnt[3] = {0, 0, 0};
        int cl[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int t = lane + 64 * p;
            cl[p] = -1;
            if (64 * p < NT && t < NT) {
                int lvl = 0;
                if (P.num_levels > 1) lvl = map_level(boxes + (size_t)t * 4, P.k_min, P.k_max);
                float scale = P.scale[0];
#pragma unroll
                for (int l = 1; l < SMOT_MAX_LEVELS; ++l) scale = (lvl == l) ? P.scale[l] : scale;
                const float4 b4 = *reinterpret_cast<const float4*>(sr + (size_t)t * 4);
                const float ww = (b4.z - b4.x) * scale;                              // window width in cells
                cl[p] = ww <= 30.0f ? 0 : (ww <= 62.0f ? 1 : 2);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mask[p][c] = __ballot(cl[p] == c);
                cnt[c] += __popcll(mask[p][c]);
            }
        }
        const int px = x >> 6, lx = x & 63;
        int cx = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p == px) cx = __builtin_amdgcn_readlane(cl[p], lx);
        rank = cx == 2 ? 0 : (cx == 1 ? cnt[2] : cnt[2] + cnt[1]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned long long mk = cx == 2 ? mask[p][2] : (cx == 1 ? mask[p][1] : mask[p][0]);
            if (p < px) rank += __popcll(mk);
            if (p == px) rank += __popcll(mk & ((1ull << lx) - 1ull));
        }
    }
    const float* roi = sr + (size_t)x * 4;
    int lvl = 0;
    if (P.num_levels > 1) lvl = map_level(boxes + (size_t)x * 4, P.k_min, P.k_max);
    lvl = __builtin_amdgcn_readfirstlane(lvl);
    const int H = P.H[lvl], W = P.W[lvl], pad = P.pad[lvl];
    const float scale = P.scale[lvl];
    const float x1 = mul_rn(roi[0], scale), y1 = mul_rn(roi[1], scale);
    const float x2 = mul_rn(roi[2], scale), y2 = mul_rn(roi[3], scale);
    const float bin_h = div_rn(fmaxf(sub_rn(y2, y1), 1.0f), (float)RX);
    const float bin_w = div_rn(fmaxf(sub_rn(x2, x1), 1.0f), (float)RX);
    int* plan = plans + (size_t)rank * PLAN_DW;

    // ---- x axis -----------------------------------------------------------------------------------------------------
    int xmn = 0x7fffffff, xmx = -1;
    {
        int lo = 0, hi = 0;
        float wl = 0.0f, wh = 0.0f;
        if (lane < NS) axis_sample(x1, bin_w, G, lane, W, pad, &lo, &hi, &wl, &wh);
        // bounding window of the touched real cells: cell indices are non-decreasing in the sample index and a zero low
        // weight means "outside" (1 - frac is never 0): first touched entry = minimum, last = maximum
        const unsigned long long m = __ballot(wl != 0.0f || wh != 0.0f);
        if (m != 0ull) {
            xmn = __builtin_amdgcn_readlane((wl != 0.0f) ? lo : hi, __ffsll((long long)m) - 1);
            xmx = __builtin_amdgcn_readlane((wh != 0.0f) ? hi : lo, 63 - __clzll((long long)m));
        }
        if (lane < NS) {
            v4i10_t e;
            e.x = (wl != 0.0f) ? lo - xmn : 0;                // window-relative columns
            e.y = (wh != 0.0f) ? hi - xmn : 0;
            e.z = __float_as_int(wl);
            e.w = __float_as_int(wh);
            *reinterpret_cast<v4i10_t*>(plan + PL_X + 4 * lane) = e;
        }
    }
    // ---- y axis -----------------------------------------------------------------------------------------------------
    int lo = 0, hi = 0;
    float wl = 0.0f, wh = 0.0f;
    if (lane < NS) axis_sample(y1, bin_h, G, lane, H, pad, &lo, &hi, &wl, &wh);
    const bool touched = wl != 0.0f || wh != 0.0f;
    const int first_row = (wl != 0.0f) ? lo : hi, last_row = (wh != 0.0f) ? hi : lo;      // of a touched entry
    int ymn = 0x7fffffff, ymx = -1;
    const unsigned long long my = __ballot(touched);
    if (my != 0ull) {
        ymn = __builtin_amdgcn_readlane(first_row, __ffsll((long long)my) - 1);
        ymx = __builtin_amdgcn_readlane(last_row, 63 - __clzll((long long)my));
    }
    const bool empty = ymx < ymn || xmx < xmn;
    const int ww = empty ? 1 : xmx - xmn + 1;
    // ---- how the window comes on chip: 16-byte loads, lane = (row of a block, 4-column group) -------------------------
    const int gw = (ww + 3) >> 2;                       // 4-column groups per window row
    const int rpi = gw <= 16 ? 64 / gw : 1;             // window rows per load instruction
    const int ws = 4 * gw;                              // floats per row of the LDS window image
    // a band = [first touched row .. last touched row] of a range of entries
    int band_r0[2] = {0, 0}, band_ib[2] = {0, 0};
    auto band_fits = [&](int slot, unsigned long long members) __attribute__((always_inline)) -> bool {
        const unsigned long long mb = my & members;
        int r0 = empty ? 0 : ymn, r1 = r0;
        if (mb != 0ull) {
            r0 = __builtin_amdgcn_readlane(first_row, __ffsll((long long)mb) - 1);
            r1 = __builtin_amdgcn_readlane(last_row, 63 - __clzll((long long)mb));
        }
        const int ib = (r1 - r0 + rpi) / rpi;
        band_r0[slot] = r0;
        band_ib[slot] = ib;
        // (the second band's image sits behind the first band's pooled rows: 15 rows of the pooled plane)
        return ib <= (slot ? FX10_IBM1 : FX10_IBM) && ib * rpi * ws <= FX10_WBUF - slot * (15 * XP1_XS);
    };
    int mode = PM_ROWS;
    if (!empty && ww <= 32 && band_fits(0, ~0ull)) {
        mode = PM_NARROW;
    } else if (!empty && ww <= 64) {
        const bool f0 = band_fits(0, (1ull << (NS / 2)) - 1ull);
        const bool f1 = band_fits(1, ((1ull << (NS / 2)) - 1ull) << (NS / 2));
        if (f0 && f1) mode = PM_WIDE;
    }
    {
        // this entry's rows inside its band's LDS image (zero-weight taps point at the band's first row: loaded, finite)
        const int r0 = (mode == PM_WIDE && lane >= NS / 2) ? band_r0[1] : band_r0[0];
        const int ll = (wl != 0.0f) ? lo - r0 : 0, lh = (wh != 0.0f) ? hi - r0 : 0;
        const int gl = (wl != 0.0f) ? lo : ymn, gh = (wh != 0.0f) ? hi : ymn;
        const bool in = lane < NS && mode != PM_ROWS;
        v4i10_t e;
        e.x = in ? ll * ws * 4 : 0;
        e.y = in ? lh * ws * 4 : 0;
        e.z = __float_as_int(wl);                                  // (lanes >= NS: weight 0)
        e.w = __float_as_int(wh);
        *reinterpret_cast<v4i10_t*>(plan + PL_YT + 4 * lane) = e;
        plan[PL_YG + 2 * lane + 0] = (lane < NS && !empty) ? (int)((unsigned)(gl * W) * 4u) : 0;
        plan[PL_YG + 2 * lane + 1] = (lane < NS && !empty) ? (int)((unsigned)(gh * W) * 4u) : 0;
    }
    if (lane == 0) {
        plan[PL_N] = x;
        plan[PL_LVL] = lvl;
        plan[PL_XMIN] = empty ? 0 : xmn;
        plan[PL_WW] = ww;
        plan[PL_EMPTY] = empty ? 1 : 0;
        plan[PL_PLANE_BYTES] = (int)((unsigned)(H * W) * 4u);
        plan[PL_MODE] = mode;
        plan[PL_GW] = gw;
        plan[PL_RPI] = rpi;
        plan[PL_WS_BYTES] = ws * 4;
        plan[PL_ROW_BYTES] = W * 4;
        plan[PL_RPI_ROW_BYTES] = rpi * W * 4;
        plan[PL_RECIP] = (65536 + gw - 1) / gw;                    // lane / gw == (lane * recip) >> 16 for lane < 64
        plan[13] = plan[14] = plan[15] = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            plan[PL_BANDS + 2 * b + 0] = b < 2 ? (int)((unsigned)(band_r0[b] * W) * 4u) : 0;
            plan[PL_BANDS + 2 * b + 1] = b < 2 ? band_ib[b] : 0;
        }
    }
}

// ---- the consumer ------------------------------------------------------------------------------------------------------
// LDS per wave (floats): template 240 | y table 256 | pooled plane 1200 | extension 864.  The window image of a step
// lies over the pooled plane + extension (the pooled rows stay in registers until the last window read).
constexpr int FX10_ZP = 15 * XP1_ZS, FX10_XPL = 30 * XP1_XS;
constexpr int FX10_WV = FX10_ZP + 256 + FX10_WBUF;
static_assert(FX10_WBUF >= FX10_XPL, "the window image covers the pooled plane");

template <int RX, int RZ, int G>
__global__ void __launch_bounds__(64 * FX10_WAVES, 4)      // four waves per SIMD: 128 VGPRs each
sr_xcorr_fused10_kernel(LevelParams P, int C, const int* __restrict__ plans, const float* __restrict__ z,
                        float* __restrict__ resp, float* __restrict__ x_debug, long long* trace, int abl) {
    constexpr int HO = RX - RZ + 1;
    constexpr int XS = XP1_XS, ZS = XP1_ZS;
    constexpr int RB = RX / 2;                                // pooled rows per step
    static_assert(HO == 16 && RX == 30 && RZ == 15 && G == 2, "the 30/15/16 correlation geometry, two samples per bin and axis");
    extern __shared__ __attribute__((aligned(16))) float sm[];       // FX10_WAVES * FX10_WV floats

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long t_start = trace ? (long long)__builtin_amdgcn_s_memtime() : 0ll;
    if (abl & 16) return;          // ablation: the bare dispatch (480 workgroups x 512 threads x 78 KB of LDS at 30 tracks)
    // experiment (abl >= 256): the second half of the workgroup's waves (the SIMDs' second waves) start (abl >> 8) * 512
    // cycles late, so that their load / pooling phase runs beside the first half's correlation
    if ((abl >> 8) && wave >= FX10_WAVES / 2) {
        for (int d = 0; d < (abl >> 8); ++d) __builtin_amdgcn_s_sleep(8);
    }
    // item L of the cost-sorted list: (rank k, channel group) — expensive rois are dispatched first
    const int ny = (C + FX10_WAVES - 1) / FX10_WAVES;
    const int L = blockIdx.x;
    const int k = L / ny;
    const int cgrp = L - k * ny;
    const int* plan = plans + (size_t)k * PLAN_DW;
    const v16i_t hd = sload16(plan, 0);
    const v4i10_t bd = *reinterpret_cast<const __attribute__((address_space(4))) v4i10_t*>(
        reinterpret_cast<unsigned long long>(plan) + 4ull * PL_BANDS);
    const int n = hd[PL_N], lvl = hd[PL_LVL], xmin = hd[PL_XMIN], ww = hd[PL_WW], mode = hd[PL_MODE];
    const unsigned plane_bytes = (unsigned)hd[PL_PLANE_BYTES];
    const int ch = cgrp * FX10_WAVES + wave;                        // this wave's channel
    if (ch >= C) return;
    const int plane = n * C + ch;
    float* wsm = sm + wave * FX10_WV;
    float* zs = wsm;                                                // template
    int* ytab = reinterpret_cast<int*>(wsm + FX10_ZP);              // y table [64][4]
    float* xs = wsm + FX10_ZP + 256;                                // pooled plane; the window image / staging rows lie over it
    float* wbuf = xs;
#define FX10_TRACE(SLOT)                                                                                      \
    if (trace && tid == 0) trace[((size_t)n * ny + cgrp) * 8 + (SLOT)] = (long long)__builtin_amdgcn_s_memtime();
    if (trace && tid == 0) trace[((size_t)n * ny + cgrp) * 8 + 0] = t_start;
    FX10_TRACE(5)

    // lane roles in the pooling passes.  NARROW: lane = (half, column): the halves pool rows 0..14 and 15..29 of the plane
    // side by side; WIDE: lane = column, the plane's two bands of 15 pooled rows one after the other.
    const bool narrow = mode == PM_NARROW;
    const int half = narrow ? (lane >> 5) : 0;
    const int col = narrow ? (lane & 31) : lane;
    const int pw = col < RX ? col : 0;
    // horizontal taps of this lane's pooled column (entries 2*pw, 2*pw+1 of the x table), this lane's y-table entry and the
    // template: requested AFTER the window's row blocks (registers), parked in LDS before the first window read
    int sxl[G], sxh[G];
    float hxw[G], lxw[G];
    v4i10_t yt;
    constexpr int NZ = (RZ * RZ + 63) / 64;
    float zreg[NZ];
    auto side_loads = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ix = 0; ix < G; ++ix) {
            const v4i10_t e = *reinterpret_cast<const v4i10_t*>(plan + PL_X + 4 * (pw * G + ix));
            sxl[ix] = e.x + half * 32;                       // (positions in a 64-float staging row: halves side by side)
            sxh[ix] = e.y + half * 32;
            hxw[ix] = __int_as_float(e.z);
            lxw[ix] = __int_as_float(e.w);
        }
        yt = *reinterpret_cast<const v4i10_t*>(plan + PL_YT + 4 * lane);
        const float* __restrict__ zg = z + (size_t)plane * (RZ * RZ);
#pragma unroll
        for (int t = 0; t < NZ; ++t) zreg[t] = (abl & 2) ? 1.0f : zg[min(lane + 64 * t, RZ * RZ - 1)];
    };
    if (hd[PL_EMPTY]) {
        // every sample in the virtual zero border: the pooled plane is exact zeros -> zero response
        for (int e = lane; e < HO * HO; e += 64) resp[(size_t)plane * HO * HO + e] = 0.0f;
        if (x_debug != nullptr)
            for (int e = lane; e < RX * RX; e += 64) x_debug[(size_t)plane * RX * RX + e] = 0.0f;
        return;
    }
    // buffer resource of the wave's plane: wave-uniform base, 32-bit offsets; it ENDS with the level's tensor, so that the
    // 16-byte loads' over-read past the last row returns zeros
    const float* fbase = P.feat[lvl];
    const unsigned long long pa = reinterpret_cast<unsigned long long>(fbase) + (unsigned long long)ch * plane_bytes;
    const unsigned pa_lo = __builtin_amdgcn_readfirstlane((unsigned)pa);
    const unsigned pa_hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
    const unsigned long long left = (unsigned long long)(C - ch) * plane_bytes;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)pa_hi << 32) | pa_lo), 0,
        (int)(left > 0x7fffffffull ? 0x7fffffffull : left), 0x00020000);

    auto park = [&]() __attribute__((always_inline)) {          // template and y table into LDS
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            const int e = lane + 64 * t;
            if (e < RZ * RZ) {
                const int u = e / RZ;
                zs[u * ZS + (e - u * RZ)] = zreg[t];
            }
        }
        *reinterpret_cast<v4i10_t*>(ytab + 4 * lane) = yt;
    };

    if (mode != PM_ROWS) {
        // ---- pooling from an LDS image of the window ------------------------------------------------------------------
        // The third generation loaded, for every pooled row, the four window rows of its two samples with one dword load
        // per lane and row: 60 load instructions per plane although the window has 18..66 distinct rows — and a
        // vector-memory instruction occupies the CU's address unit for 16 cycles whatever its width (measured in round 4:
        // a wave that issued 120 row loads needed 9.7 k cycles for that alone; 900 per CU = 14 k cycles).  Here the window
        // (or a band of it) is fetched ONCE by 16-byte loads — lane = (row within a block, 4-column group), 64 / groups
        // rows per instruction: 3..10 instructions per plane — parked in LDS row-major, and the vertical taps read it at
        // `column + row offset` with the offsets and weights of the y table (a broadcast ds_read_b128 per sample).
        const int gw = hd[PL_GW], rpi = hd[PL_RPI];
        const int jrow = (lane * hd[PL_RECIP]) >> 16;        // lane / gw
        const int grp = lane - jrow * gw;
        const bool slot = jrow < rpi;                        // lanes past the last whole row of a block idle
        const unsigned gl_lane = slot ? (unsigned)(jrow * hd[PL_ROW_BYTES]) + (unsigned)(xmin + 4 * grp) * 4u : 0u;
        const unsigned lds_lane = (unsigned)(jrow * hd[PL_WS_BYTES]) + 16u * (unsigned)grp;
        const unsigned blk_lds = (unsigned)(rpi * hd[PL_WS_BYTES]);            // bytes of LDS image per row block
        const unsigned blk_gl = (unsigned)hd[PL_RPI_ROW_BYTES];
        const int wcol = min(col, ww - 1);
        const int spos = half * 32 + wcol;                                    // ... and in a staging row
        const int nsteps = narrow ? 1 : 2;
        typedef unsigned v4u_t __attribute__((__vector_size__(4 * sizeof(unsigned))));
        v4u_t R0[FX10_IBM], R1[FX10_IBM1];
        // all row blocks of the plane are requested up front: ONE memory round trip
        // (branch-free: a block past the band's last repeats the last one — a conditional load costs hipcc a branch, a
        // wait and a spill per load)
        if (!(abl & 1)) {
#pragma unroll
        for (int i = 0; i < FX10_IBM; ++i)
            R0[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, gl_lane, (unsigned)bd.x + (unsigned)min(i, bd.y - 1) * blk_gl, 0);
#pragma unroll
        for (int i = 0; i < FX10_IBM1; ++i)
            R1[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, gl_lane, (unsigned)bd.z + (unsigned)min(i, max(bd.w, 1) - 1) * blk_gl, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        side_loads();
        park();
        FX10_TRACE(2)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (b < nsteps && !(abl & 4)) {
                // the step's window image: band 0 over the pooled plane's place, band 1 behind its first 15 rows (which band 0
                // has written by then); after the previous step's staging reads — LDS operations of a wave run in order
                float* wb = wbuf + b * (RB * XS);
                const char* wimg = reinterpret_cast<const char*>(wb) + wcol * 4;    // this lane's column in the image
                if (slot) {
                    char* dst = reinterpret_cast<char*>(wb) + lds_lane;
                    if (b == 0) {
#pragma unroll
                        for (int i = 0; i < FX10_IBM; ++i) *reinterpret_cast<v4u_t*>(dst + min(i, bd.y - 1) * blk_lds) = R0[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < FX10_IBM1; ++i) *reinterpret_cast<v4u_t*>(dst + min(i, bd.w - 1) * blk_lds) = R1[i];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // vertical taps: this lane's 15 pooled rows = entries tb .. tb+29 of the y table, in groups of five rows:
                // ten table entries, then their twenty window cells, then the FMAs (three LDS round trips per group)
                const char* tb = reinterpret_cast<const char*>(ytab) + (narrow ? half : b) * (RB * G * 16);
                float cs[RB];
#pragma unroll
                for (int r0 = 0; r0 < RB; r0 += 5) {
                    v4i10_t e[5 * G];
#pragma unroll
                    for (int j = 0; j < 5 * G; ++j) e[j] = *reinterpret_cast<const v4i10_t*>(tb + (r0 * G + j) * 16);
                    __builtin_amdgcn_sched_barrier(0);
                    float vl[5 * G], vh[5 * G];
#pragma unroll
                    for (int j = 0; j < 5 * G; ++j) {
                        vl[j] = *reinterpret_cast<const float*>(wimg + e[j].x);
                        vh[j] = *reinterpret_cast<const float*>(wimg + e[j].y);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        float c_ = 0.0f;
#pragma unroll
                        for (int iy = 0; iy < G; ++iy) {
                            c_ = fmaf(__int_as_float(e[r * G + iy].z), vl[r * G + iy], c_);
                            c_ = fmaf(__int_as_float(e[r * G + iy].w), vh[r * G + iy], c_);
                        }
                        cs[r0 + r] = c_;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // horizontal taps through LDS staging rows (over the window image the wave has just finished with)
                constexpr int GRS = 5;
                float acc[RB];
#pragma unroll
                for (int g0 = 0; g0 < RB; g0 += GRS) {
#pragma unroll
                    for (int r = g0; r < g0 + GRS && r < RB; ++r) wb[(r - g0) * 64 + spos] = cs[r];
                    float p_[GRS][G][2];
#pragma unroll
                    for (int r = g0; r < g0 + GRS && r < RB; ++r)
#pragma unroll
                        for (int ix = 0; ix < G; ++ix) {
                            p_[r - g0][ix][0] = wb[(r - g0) * 64 + sxl[ix]];
                            p_[r - g0][ix][1] = wb[(r - g0) * 64 + sxh[ix]];
                        }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = g0; r < g0 + GRS && r < RB; ++r) {
                        float a_ = 0.0f;
#pragma unroll
                        for (int ix = 0; ix < G; ++ix) {
                            a_ = fmaf(hxw[ix], p_[r - g0][ix][0], a_);
                            a_ = fmaf(lxw[ix], p_[r - g0][ix][1], a_);
                        }
                        acc[r] = a_ * (1.0f / (float)(G * G));       // exact: /4
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the step's pooled rows over the window image (every window / staging read of the step has been issued)
                if (col < RX) {
                    float* xrow = xs + ((narrow ? half : b) * RB) * XS + col;
#pragma unroll
                    for (int r = 0; r < RB; ++r) xrow[r * XS] = acc[r];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    } else {
        // ---- pooling by per-row dword loads in 64-column chunks: windows wider than 64 columns or taller than the LDS
        // image (degenerate aspect ratios, rare).  All 64 lanes = columns of a chunk, cross-lane gathers per chunk.
        // Generation 3's arithmetic and term order. ---------------------------------------------------------------------
        constexpr int ROWS = 5;
        side_loads();
        park();
        FX10_TRACE(2)
        const int pwl = lane < RX ? lane : 0;
        int cxl[G], cxh[G];
        float chw[G], clw[G];
#pragma unroll
        for (int ix = 0; ix < G; ++ix) {
            const v4i10_t e = *reinterpret_cast<const v4i10_t*>(plan + PL_X + 4 * (pwl * G + ix));
            cxl[ix] = e.x;
            cxh[ix] = e.y;
            chw[ix] = __int_as_float(e.z);
            clw[ix] = __int_as_float(e.w);
        }
        const int nchunk = (ww + 63) >> 6;
#pragma unroll 1
        for (int r0 = 0; r0 < RX; r0 += ROWS) {
            // this block's y entries: entry (b, iy) into lane b*G + iy (constant-lane readlanes below)
            const int ei = min(lane + r0 * G, 63);
            const unsigned ol = (unsigned)plan[PL_YG + 2 * ei], oh = (unsigned)plan[PL_YG + 2 * ei + 1];
            const float wl = __int_as_float(plan[PL_YT + 4 * ei + 2]), wh = __int_as_float(plan[PL_YT + 4 * ei + 3]);
            float ac[ROWS];
#pragma unroll
            for (int b = 0; b < ROWS; ++b) ac[b] = 0.0f;
#pragma unroll 1
            for (int cc = 0; cc < nchunk; ++cc) {
                const int cbase = cc << 6;
                const int wc = min(cbase + lane, ww - 1);
                const unsigned voff = (unsigned)(xmin + wc) * 4u;
                float v[ROWS][G][2];
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int iy = 0; iy < G; ++iy) {
                        const int e = b * G + iy;
                        v[b][iy][0] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            rsrc, voff, __builtin_amdgcn_readlane((int)ol, e), 0));
                        v[b][iy][1] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            rsrc, voff, __builtin_amdgcn_readlane((int)oh, e), 0));
                    }
                __builtin_amdgcn_sched_barrier(0);
                float cs[ROWS];
#pragma unroll
                for (int b = 0; b < ROWS; ++b) {
                    float c_ = 0.0f;
#pragma unroll
                    for (int iy = 0; iy < G; ++iy) {
                        const int e = b * G + iy;
                        c_ = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), e)), v[b][iy][0], c_);
                        c_ = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wh), e)), v[b][iy][1], c_);
                    }
                    cs[b] = c_;
                }
                __builtin_amdgcn_sched_barrier(0);
                int al[G], ah[G];
                bool inl[G], inh[G];
#pragma unroll
                for (int ix = 0; ix < G; ++ix) {
                    const int tl = cxl[ix] - cbase, th = cxh[ix] - cbase;
                    inl[ix] = (unsigned)tl < 64u;
                    inh[ix] = (unsigned)th < 64u;
                    al[ix] = (tl & 63) << 2;                                  // ds_bpermute takes byte addresses
                    ah[ix] = (th & 63) << 2;
                }
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int ix = 0; ix < G; ++ix) {
                        const float pl_ = __int_as_float(__builtin_amdgcn_ds_bpermute(al[ix], __float_as_int(cs[b])));
                        const float ph_ = __int_as_float(__builtin_amdgcn_ds_bpermute(ah[ix], __float_as_int(cs[b])));
                        // a tap outside the chunk adds nothing (not even 0 * garbage)
                        ac[b] = inl[ix] ? fmaf(chw[ix], pl_, ac[b]) : ac[b];
                        ac[b] = inh[ix] ? fmaf(clw[ix], ph_, ac[b]) : ac[b];
                    }
            }
#pragma unroll
            for (int b = 0; b < ROWS; ++b)
                if (lane < RX) xs[(r0 + b) * XS + lane] = ac[b] * (1.0f / (float)(G * G));
        }
    }
    FX10_TRACE(3)
    if (x_debug != nullptr) {
        for (int e = lane; e < RX * RX; e += 64) {
            const int r = e / RX;
            x_debug[(size_t)plane * RX * RX + e] = xs[r * XS + (e - r * RX)];
        }
    }
    // ---- correlation: 2x2 output patches per lane ----------------------------------------------------------------------
    if (!(abl & 8)) xcorr_patch1_compute<RX, RZ, true>(xs, zs, lane, resp, plane);
    FX10_TRACE(4)
#undef FX10_TRACE
}

// ---- launches ----------------------------------------------------------------------------------------------------------
int fused10_plan_floats() { return SMOT_PLAN_FLOATS; }
int launch_roi_plans(const LevelParams& P, const float* sr, const float* boxes, int N, float* plans, hipStream_t st) {
    hipLaunchKernelGGL((roi_plan_kernel<30, 2>), dim3(N), dim3(64), 0, st, P, sr, boxes, N, reinterpret_cast<int*>(plans));
    return check_launch("roi_plans");
}

int launch_fused10(const LevelParams& P, int C, const float* plans, const float* z, int N, float* resp, float* x_debug,
                   hipStream_t st) {
    const int ny = (C + FX10_WAVES - 1) / FX10_WAVES;
    constexpr size_t smem = (size_t)FX10_WAVES * FX10_WV * sizeof(float);
    const int rco = ensure_lds_optin(reinterpret_cast<const void*>(&sr_xcorr_fused10_kernel<30, 15, 2>), smem, "sr_xcorr_fused10");
    if (rco) return rco;
#ifdef SMOT_DEBUG
    {
        static bool once = false;
        if (!once) {
            once = true;
            int nb = -1;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sr_xcorr_fused10_kernel<30, 15, 2>, 64 * FX10_WAVES, smem);
            fprintf(stderr, "[fused10] occupancy query: %d workgroups per CU (err %d), %zu bytes of LDS per workgroup\n", nb, (int)e, smem);
        }
    }
#endif
    SMOT_LAUNCH((sr_xcorr_fused10_kernel<30, 15, 2>), dim3(N * ny), dim3(64 * FX10_WAVES), smem, st, P, C,
                reinterpret_cast<const int*>(plans), z, resp, x_debug, g_trace, knobs().fused_abl);
    return check_launch("sr_xcorr_fused10");
}

}  // namespace smot
