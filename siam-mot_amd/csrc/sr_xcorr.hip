// K1+K2 fused — search-region pooling straight into the cross-correlation.
//
// Replaces, inside EMM.forward (reference EMM/track_core.py:49-53): TrackUtils.pad_feature
// (track_head/track_utils.py:87-107) -> SRPooler on the search regions (EMM/sr_pool.py:53-91, legacy
// ROIAlign [UPSTREAM]) -> xcorr_depthwise (EMM/xcorr.py:37-46).  The [N,C,30,30] search-region tensor
// (13.8 MB @30 tracks, written once and read once by the unfused path) never leaves the CU.
//
// Workgroup = (track, 8 channels); 4 waves, each owning 2 channels (planes).
//   1. The per-axis sample tables of the roi are built once per workgroup in LDS exactly as in
//      roi_align.hip (reference rounding sequence, padded extent, cells in the virtual border get
//      weight 0) together with the bounding window [ymin..ymax] x [xmin..xmax] of touched real cells.
//   2. Pooling is SEPARABLE: bin(ph,pw) = 1/g^2 * sum_iy sum_ix [ hx*col_iy(xlo) + lx*col_iy(xhi) ] with
//      col_iy(wx) = hy*V[ylo][wx] + ly*V[yhi][wx].  A lane owns window column wx: for every pooled row it
//      loads the (up to) four feature rows of the two y-samples — contiguous row segments, coalesced,
//      no LDS window — forms the two column values, and lane pw gathers its four column values with
//      ds_bpermute (cross-lane, no LDS storage).  Terms are added in the reference's (iy, ix) order; only
//      the factorisation (hy*hx)*v -> hx*(hy*v) differs (fp32 rounding level, tested to 1e-5).
//   3. The pooled 30x30 planes land in the same LDS image the xcorr kernel stages from HBM
//      (xcorr_patch2.h), the templates are staged next to them, and the FMA phase is the shared
//      xcorr_patch2_compute: responses are bit-identical to smot_xcorr_dw_fwd on the pooled planes.
// Windows wider than 64 columns (search regions far larger than their FPN level suggests: degenerate
// aspect ratios) take a workgroup-uniform slow path: per-bin gathers, same arithmetic as roi_align.hip.
#include "roi_common.h"
#include "xcorr_patch2.h"
#include "xcorr_patch1.h"
#include "xcorr_f16x2.h"
#include "knobs.h"
#include <type_traits>
namespace smot { int launch_plane_absmax(const float* resp, int planes, int hw, float* pm, hipStream_t st); }   // tower_wino.hip

namespace smot {

constexpr int FX_CH = 8;          // channels per workgroup (2 per wave)

// EMM.extract_cache in one launch: the pool-only kernel also writes the next frame's search regions
// (update_boxes_in_pad_images + extend_bbox, reference track_utils.py:62-85,109-135; same arithmetic as
// search_region_kernel in roi_align.hip) from its (roi, channel-group 0) workgroup.
struct SrOut {
    float* sr;            // [R,4] or nullptr
    float pad, half_e, two_e, min_wh;
    long long* trace;     // phase trace (smot_debug_trace) or nullptr
    int abl;              // timing ablation of the generation-2 kernel (wrong results): 2 = no correlation phase
                          // (SMOT_FUSED_ABL=2).  A run-time switch around the row loads is NOT an option: the
                          // branch makes hipcc wait for every pair of loads (measured +2 us).
    const int* n_valid;   // device count of valid rois, or nullptr: rois >= *n_valid are skipped (the launch covers
                          // a CAPACITY when the count is still on the device — smot_emm_extract_cache_masked_fwd)
    int order;            // workgroup -> (roi, channel group) assignment: 0 = grid order, 1..3 = cost-sorted forms
                          // (fx_assign below; the measurement library can select any, SMOT_FUSED_ORDER)
    float* hint_out;      // pool-only kernel: write the cost-sorted roi list of the NEXT frame's search regions here
                          // (fx_write_hint below; [R] entries of SMOT_HINT_FLOATS floats) or nullptr
    const float* hint_in; // pooling + correlation kernel: such a list for ITS rois (written by the extraction that made
                          // them), read instead of ranking the rois again in every workgroup; or nullptr
    int hint_extra;       // hint writer with n_valid: the list ranks *n_valid + hint_extra rows — the rows the caller appended behind
                          // the valid ones before this launch (the tracking frame's carried dormant tracks: their boxes stand
                          // in `boxes` behind the active rows, and a dormant row's search region is search_region_of(its box),
                          // bit for bit what the extraction that created it wrote)
    int plan_pad[SMOT_MAX_LEVELS];   // hint writer: zero-pad cells per level of the NEXT frame's search-region pooling
                          // (the extraction itself pools un-padded maps); the finished sample tables in a hint entry are
                          // built against them
    float* plane_max;     // pooling + correlation kernel: the largest |response| of every plane [R][C], or nullptr — the tower
                          // kernel's split form scales a track's response by a power of two chosen from them (tower_wino.hip)
};
// One hint entry = SMOT_HINT_FLOATS dwords: {search region x1,y1,x2,y2, FPN level, roi index, 0, 0 | ymin, ymax, xmin, xmax of the
// touched window, pad cells / H / W of the level the tables were built against, 0 | y table [64] x int4 | x table [64] x int4}
// — since round 4 the FINISHED sample tables of the roi's 30x30 pooling ride along (exactly the entries the consumer's
// waves 0 / 1 would compute: reference rounding sequence, re-based rows / columns), so that a consumer workgroup loads 2 KB
// instead of ranking, dividing and building them.
constexpr int HINT_FLOATS = SMOT_HINT_FLOATS;
constexpr int HINT_BOUNDS = 8, HINT_GEOM = 12, HINT_YTAB = 16, HINT_XTAB = 16 + 256;
// ABI 12: behind the tables, entry x also carries the BY-ROI record of roi x (entries are sorted by rank, this record is
// not): {search region x1,y1,x2,y2 of roi x, its FPN level, the number of rois the list ranks, status word (entry 0's is
// THE status word of the list, zeroed by the writer), 0}.  The consumer verifies it against ITS `sr[x]` / `boxes[x]` / roi
// count (fx_verify_hint) — the list is a permutation of the rois it was made from by construction, so "every by-roi
// record equals the consumer's roi" is "the list describes exactly these rois".
constexpr int HINT_BYROI = HINT_XTAB + 256, HINT_STATUS = HINT_BYROI + 6;
static_assert(HINT_BYROI + 8 == HINT_FLOATS && (HINT_FLOATS * 4) % 32 == 0 && HINT_STATUS == SMOT_HINT_STATUS_WORD,
              "hint entry layout");

// base (wave-uniform, SGPR pair) + 32-bit unsigned BYTE offset: selects the `global_load v, v_off, s[base]`
// addressing form (one address VGPR per load instead of a 64-bit pair — 120 loads are in flight).
typedef const __attribute__((address_space(1))) char* gptr_t;
__device__ __forceinline__ float ld_off(gptr_t base, unsigned byte_off) {
    return *reinterpret_cast<const __attribute__((address_space(1))) float*>(base + byte_off);
}
// Launder a wave-uniform global pointer through an SGPR pair: stops loop-strength-reduction from turning
// "base(pl) + offset(s)" into one 64-bit per-lane pointer induction variable per load (60 of them spill).
__device__ __forceinline__ gptr_t uniform_base(const float* p) {
    unsigned long long a = reinterpret_cast<unsigned long long>(p);
    asm volatile("" : "+s"(a));
    return reinterpret_cast<gptr_t>(a);
}

#ifdef SMOT_DEBUG
#include "../../measure/csrc/sr_xcorr_gen2.inc"      // measurement library only (not product source)
#endif

// ---- third generation (default): sample tables in registers, wave-uniform addressing, gathers in bulk ------------
// What the phase traces of generation 2 showed (round-1 traces, ticks per workgroup @30 tracks):
// tables 4.3 k, pooling 14 k on average but 26 k for the 33..64-column windows, correlation 12 k.  The pooling was
// neither bandwidth- nor LDS-bound; it was a chain of dependent round trips: per pooled row one LDS read of the
// weights, four loads' waits, four ds_bpermute gathers waited for one by one — 15 rows x 2 batches x ~400 cycles —
// behind a load-issue phase that spent a quarter-rate v_mul_lo_u32 + add per load, behind an LDS table + barrier.
// This generation removes the chains:
//   * every wave builds BOTH sample tables itself, lane = sample, in registers (no LDS image, no barrier); the
//     window bounds come from one ballot per axis (the tables are monotone: first / last touched entry);
//   * row offsets and vertical weights are read with v_readlane (constant lane) into SGPRs: the 60 row loads of a
//     batch are buffer loads `voffset = lane column (+ plane), soffset = row` — no per-load VALU at all — and the
//     vertical taps are FMAs with an SGPR weight;
//   * all column sums of a batch are formed first, then all 60 gathers are issued back to back and waited for
//     ONCE, then the horizontal taps run — two LDS round trips per batch instead of thirty;
//   * windows <= 32 columns: a wave pools TWO planes side by side (plane = lane / 32, same rows and weights for
//     both halves, so everything stays wave-uniform) and the two waves of a plane pair split the pooled rows:
//     one batch of 60 loads per wave covers the pair; wider windows: one plane per wave, two batches, the second
//     batch's loads in flight while the first is gathered; windows wider than a wave are walked in 64-column
//     chunks by the same code (there is no separate slow path any more).
// Arithmetic per output is generation 2's, term by term (same fmaf chains), so results are bit-identical to it.
typedef int v4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rl_f(float v, int lane_const) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_const));
}

// update_boxes_in_pad_images + extend_bbox (track_utils.py:62-85,109-135) for one box: the next frame's search region.
__device__ __forceinline__ float4 search_region_of(float b0, float b1, float b2, float b3, const SrOut& S) {
    const float bx1 = add_rn(b0, S.pad), by1 = add_rn(b1, S.pad);
    const float bx2 = add_rn(b2, S.pad), by2 = add_rn(b3, S.pad);
    const float bw = add_rn(sub_rn(bx2, bx1), 1.0f), bh = add_rn(sub_rn(by2, by1), 1.0f);
    const float w_ext = max_nan(div_rn(sub_rn(S.min_wh, bw), S.two_e), mul_rn(bw, S.half_e));
    const float h_ext = max_nan(div_rn(sub_rn(S.min_wh, bh), S.two_e), mul_rn(bh, S.half_e));
    return make_float4(sub_rn(bx1, w_ext), sub_rn(by1, h_ext), add_rn(bx2, w_ext), add_rn(by2, h_ext));
}

// The pooling + correlation kernel of the NEXT frame ranks its rois by cost class in every one of its workgroups
// (fx_assign below: ~6.5 k cycles at the head of each, two vector-memory round trips on lines all CUs want at once).
// The extraction that creates those rois knows everything the ranking needs one frame earlier: one wave of one extra
// workgroup of the template-pooling launch ranks them ONCE and writes the sorted list — entry k = the roi of rank k
// with its search region and FPN level — so that a consumer workgroup needs one scalar load of its own entry.
// Same classes and the same order as fx_assign order 1 (class descending, roi ascending); the consumer's results do
// not depend on it (any permutation of the rois is a valid assignment).
// Workgroup x of the extraction launch's extra row writes the entry of roi x: both of its first two waves rank the rois
// (lane = roi, three ballots per 64 rois: the rank of roi x is read out of its lane), wave 0 builds the y table and wave 1
// the x table of the roi's NEXT-frame search region exactly as the consumer's waves 0 / 1 would (same code, same rounding
// sequence), against the pad cells the next frame's pooling uses.
template <int RXN, int G>
__device__ __forceinline__ void fx_write_hint(const LevelParams& P, const float* __restrict__ boxes, const SrOut& S,
                                              int NT, int x, int wave, int lane) {
    if (NT > 256 || wave >= 2) return;                         // (the consumer keeps grid order beyond 256 rois)
    const int nv = (S.n_valid != nullptr) ? min(*S.n_valid + S.hint_extra, NT) : NT;
    if (x >= nv) return;
    unsigned long long mask[4][3];
    int cnt[3] = {0, 0, 0};
    int cl[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int t = lane + 64 * p;
        cl[p] = -1;
        if (64 * p < nv && t < nv) {
            const float4 b4 = *reinterpret_cast<const float4*>(boxes + (size_t)t * 4);
            int lvl = 0;
            if (P.num_levels > 1) lvl = map_level(boxes + (size_t)t * 4, P.k_min, P.k_max);
            float scale = P.scale[0];
#pragma unroll
            for (int l = 1; l < SMOT_MAX_LEVELS; ++l) scale = (lvl == l) ? P.scale[l] : scale;
            const float4 srb = search_region_of(b4.x, b4.y, b4.z, b4.w, S);
            const float ww = (srb.z - srb.x) * scale;                                    // window width in cells
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
    int rank = cx == 2 ? 0 : (cx == 1 ? cnt[2] : cnt[2] + cnt[1]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned long long mk = cx == 2 ? mask[p][2] : (cx == 1 ? mask[p][1] : mask[p][0]);
        if (p < px) rank += __popcll(mk);
        if (p == px) rank += __popcll(mk & ((1ull << lx) - 1ull));
    }
    // the roi itself (wave-uniform scalar loads)
    const float* bx = boxes + (size_t)x * 4;
    const float4 roi = search_region_of(bx[0], bx[1], bx[2], bx[3], S);
    int lvl = 0;
    if (P.num_levels > 1) lvl = map_level(bx, P.k_min, P.k_max);
    lvl = __builtin_amdgcn_readfirstlane(lvl);
    const int H = P.H[lvl], W = P.W[lvl], pad = S.plan_pad[lvl];
    const float scale = P.scale[lvl];
    const float x1 = mul_rn(roi.x, scale), y1 = mul_rn(roi.y, scale);
    const float x2 = mul_rn(roi.z, scale), y2 = mul_rn(roi.w, scale);
    const float bin_h = div_rn(fmaxf(sub_rn(y2, y1), 1.0f), (float)RXN);
    const float bin_w = div_rn(fmaxf(sub_rn(x2, x1), 1.0f), (float)RXN);
    int* ent = reinterpret_cast<int*>(S.hint_out) + (size_t)rank * HINT_FLOATS;
    int lo = 0, hi = 0;
    float wl = 0.0f, wh = 0.0f;
    if (lane < RXN * G) {
        if (wave == 0) {
            axis_sample(y1, bin_h, G, lane, H, pad, &lo, &hi, &wl, &wh);
        } else {
            axis_sample(x1, bin_w, G, lane, W, pad, &lo, &hi, &wl, &wh);
        }
    }
    int mn = 0x7fffffff, mx = -1;
    const unsigned long long m = __ballot(wl != 0.0f || wh != 0.0f);
    if (m != 0ull) {
        mn = __builtin_amdgcn_readlane((wl != 0.0f) ? lo : hi, __ffsll((long long)m) - 1);
        mx = __builtin_amdgcn_readlane((wh != 0.0f) ? hi : lo, 63 - __clzll((long long)m));
    }
    const int rl = (wl != 0.0f) ? lo : mn, rh = (wh != 0.0f) ? hi : mn;
    int4 e;
    if (wave == 0) {
        e.x = (int)((unsigned)(rl * W) * 4u);
        e.y = (int)((unsigned)(rh * W) * 4u);
    } else {
        e.x = (wl != 0.0f) ? lo - mn : 0;
        e.y = (wh != 0.0f) ? hi - mn : 0;
    }
    e.z = __float_as_int(wl);
    e.w = __float_as_int(wh);
    *reinterpret_cast<int4*>(ent + (wave == 0 ? HINT_YTAB : HINT_XTAB) + 4 * lane) = e;
    if (lane == 0) {
        ent[HINT_BOUNDS + 2 * wave] = mn;
        ent[HINT_BOUNDS + 2 * wave + 1] = mx;
        if (wave == 0) {
            float4* o = reinterpret_cast<float4*>(ent);
            o[0] = roi;
            o[1] = make_float4(__int_as_float(lvl), __int_as_float(x), 0.f, 0.f);
            ent[HINT_GEOM + 0] = pad;
            ent[HINT_GEOM + 1] = H;
            ent[HINT_GEOM + 2] = W;
            ent[HINT_GEOM + 3] = __float_as_int(scale);           // (the tables were built with this scale: part of the stamp)
            // the by-roi record of roi x, in entry x (not entry `rank`): what the consumer verifies against its own tensors
            int* rec = reinterpret_cast<int*>(S.hint_out) + (size_t)x * HINT_FLOATS + HINT_BYROI;
            float4* r4 = reinterpret_cast<float4*>(rec);
            r4[0] = roi;
            rec[4] = lvl;
            rec[5] = nv;
            rec[6] = 0;                                            // status word (entry 0's is the list's)
            rec[7] = 0;
        }
    }
}

// The consumer's check of an order hint (VERDICT r4 "next" #2: detect, not trust).  A workgroup of grid row 0 verifies the
// by-roi record of ITS grid column x against the launch's own tensors: sr[x] bit for bit, the FPN level recomputed from
// boxes[x], and the number of rois.  Every address is known at kernel start, so the loads travel beside the entry's own
// (no dependent round trip), one dword per lane, and the comparison runs in a wave that would otherwise wait at the table
// barrier: no cost on the workgroup's chain.  A mismatch raises the list's status word (entry 0): the decode kernel of the
// same head then writes NaN boxes and scores for every row (smot_emm_track_fwd), the solver's record counts them, and the
// tracking loop raises — a stale hint is reported, never silently used.
__device__ __forceinline__ void fx_verify_hint(const LevelParams& P, const float* __restrict__ sr,
                                               const float* __restrict__ boxes, const float* hint, int x, int NT, int lane) {
    const int* rec = reinterpret_cast<const int*>(hint) + (size_t)x * HINT_FLOATS + HINT_BYROI;
    int v = 0;
    if (lane < 4) {
        v = __float_as_int(sr[(size_t)x * 4 + lane]);
    } else if (lane < 8) {
        v = __float_as_int(boxes[(size_t)x * 4 + lane - 4]);
    } else if (lane < 14) {
        v = rec[lane - 8];
    }
    const int up = __shfl_up(v, 8);                              // lanes 8..11: the consumer's sr[x] beside the record's
    float b4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b4[i] = __int_as_float(__builtin_amdgcn_readlane(v, 4 + i));
    int lvl = 0;
    if (P.num_levels > 1) lvl = map_level(b4, P.k_min, P.k_max);
    const bool bad = (lane >= 8 && lane < 12 && up != v) || (lane == 12 && v != lvl) || (lane == 13 && v != NT);
    if (__ballot(bad) != 0ull && lane == 0)
        atomicOr(reinterpret_cast<int*>(const_cast<float*>(hint)) + HINT_STATUS, 1);
}

// Which (roi, channel group) a workgroup takes.  The grid is (rois, channel groups) and the hardware dispatches
// workgroups in linear order (x fastest), L and L + 256 onto the same CU while the launch fits the chip; a roi whose
// window is wider than 32 columns costs about twice a narrow one (two row blocks per plane), so the order in which
// rois appear in the caller's tensor decided how evenly the CUs were loaded (measure/debug/pairing_probe.py: the same 30
// boxes, 18.9 us in the benchmark's size-cycling order, 17.6 us sorted by width; 49.2 vs 42.5 us at 100).  Every
// workgroup therefore ranks the rois by a cost class itself — lane = roi, three ballots per 64 rois, all wave-uniform
// — and takes its item from the cost-sorted list; results do not depend on the assignment (it is a bijection).
//   order 1: item i = L of the list sorted by (class descending, roi, channel group)
//   order 2: the first 256 workgroups take the list from the front, the next 256 from the back (expensive with cheap
//            on a CU), the rest the middle in descending order
//   order 3: grid position x -> the roi of rank x (channel group = grid y, as in grid order)
// Rois at or past *n_valid keep their own index (they return at once).  More than 256 rois: grid order.
// Returns true when the roi's box and FPN level are handed back as well (read out of the lane that ranked it: the
// workgroup's own roi loads — a second dependent memory round trip — are then not needed).
__device__ __forceinline__ bool fx_assign(const LevelParams& P, const float* __restrict__ sr,
                                          const float* __restrict__ boxes, const int* __restrict__ n_valid, int order,
                                          int by, int ny, const float* __restrict__ hint, int lane, int* n_out,
                                          int* cg_out, float4* roi_out, int* lvl_out, int* k_out) {
    // (by, ny): the workgroup's row and the number of rows of the (roi, channel group) grid — blockIdx.y / gridDim.y
    // less the hint row of an extraction launch
    const int NT = gridDim.x;
    int n = blockIdx.x, cg = by;
    *n_out = n;
    *cg_out = cg;
    if (order == 0 || NT > 256 || NT < 2) return false;
    const int T = NT * ny, L = by * NT + blockIdx.x;
    if (hint != nullptr && order == 1 && n_valid == nullptr) {
        // the list was made when the rois were (fx_write_hint): one scalar load of this workgroup's entry
        const int k = L / ny;
        *cg_out = L - k * ny;
        *k_out = k;
        // through the constant address space: a wave-uniform address there is a scalar load (one s_load_dwordx8 per
        // wave through the scalar cache; as a plain global pointer hipcc issues vector loads — it cannot see that
        // nothing writes the list during this launch)
        typedef int v8i_t __attribute__((ext_vector_type(8)));
        const v8i_t e = *reinterpret_cast<const __attribute__((address_space(4))) v8i_t*>(
            reinterpret_cast<unsigned long long>(hint) + (unsigned long long)k * (HINT_FLOATS * 4));
        roi_out->x = __int_as_float(e[0]);
        roi_out->y = __int_as_float(e[1]);
        roi_out->z = __int_as_float(e[2]);
        roi_out->w = __int_as_float(e[3]);
        *lvl_out = min(max(e[4], 0), P.num_levels - 1);        // (a stale list must not index out of range)
        *n_out = min(max(e[5], 0), NT - 1);
        return true;
    }
    const int nv = (n_valid != nullptr) ? min(*n_valid, NT) : NT;
    constexpr int SLOTS = 256;                       // CUs: one workgroup each per dispatch round
    int i = L;
    if (order == 2) {
        const int r = L / SLOTS, q = L - r * SLOTS;
        if (r == 1) {
            i = T - 1 - q;
        } else if (r >= 2) {
            i = SLOTS + (L - 2 * SLOTS);
        }
    }
    int k;
    if (order == 3) {
        k = blockIdx.x;
    } else {
        k = i / ny;
        cg = i - k * ny;
    }
    *cg_out = cg;
    if (k >= nv) {
        *n_out = k;
        return false;
    }
    unsigned long long mask[4][3];
    int cnt[3] = {0, 0, 0};
    float rbx[4], rby[4], rbz[4], rbw[4];
    int rl[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int cls = -1;
        const int t = lane + 64 * p;
        rbx[p] = rby[p] = rbz[p] = rbw[p] = 0.0f;
        rl[p] = 0;
        if (64 * p < nv && t < nv) {
            int lvl = 0;
            if (P.num_levels > 1) lvl = map_level(boxes + (size_t)t * 4, P.k_min, P.k_max);
            float scale = P.scale[0];
#pragma unroll
            for (int l = 1; l < SMOT_MAX_LEVELS; ++l) scale = (lvl == l) ? P.scale[l] : scale;
            const float4 b4 = *reinterpret_cast<const float4*>(sr + (size_t)t * 4);
            rbx[p] = b4.x;
            rby[p] = b4.y;
            rbz[p] = b4.z;
            rbw[p] = b4.w;
            rl[p] = lvl;
            const float ww = (b4.z - b4.x) * scale;                                         // window width in cells
            cls = ww <= 30.0f ? 0 : (ww <= 62.0f ? 1 : 2);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mask[p][c] = __ballot(cls == c);
            cnt[c] += __popcll(mask[p][c]);
        }
    }
    int m = k, csel = 2;
    if (m >= cnt[2]) {
        m -= cnt[2];
        csel = 1;
        if (m >= cnt[1]) {
            m -= cnt[1];
            csel = 0;
        }
    }
    int psel = 0;
    bool done = false;
    unsigned long long msel = 0ull;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned long long mk = csel == 2 ? mask[p][2] : (csel == 1 ? mask[p][1] : mask[p][0]);
        const int pc = __popcll(mk);
        if (!done) {
            if (m < pc) {
                psel = p;
                msel = mk;
                done = true;
            } else {
                m -= pc;
            }
        }
    }
    const unsigned long long below = msel & ((1ull << lane) - 1ull);
    const bool hit = ((msel >> lane) & 1ull) != 0ull && __popcll(below) == m;
    const unsigned long long b = __ballot(hit);
    if (b == 0ull) return false;                                  // (cannot happen: k < nv)
    const int ln = __ffsll((long long)b) - 1;
    *n_out = ln + 64 * psel;
    float sx = rbx[0], sy = rby[0], sz = rbz[0], sw = rbw[0];
    int lsel = rl[0];
#pragma unroll
    for (int p = 1; p < 4; ++p)
        if (psel == p) {                                         // wave-uniform
            sx = rbx[p];
            sy = rby[p];
            sz = rbz[p];
            sw = rbw[p];
            lsel = rl[p];
        }
    roi_out->x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), ln));
    roi_out->y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), ln));
    roi_out->z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sz), ln));
    roi_out->w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sw), ln));
    *lvl_out = __builtin_amdgcn_readlane(lsel, ln);
    return true;
}

// MM: the correlation on the matrix pipe (xcorr_f16x2.h) — the product's form since round 6.  A plane's slot then holds its
// image (fp32, later the two fp16 half images in place) and the even / odd Toeplitz rows of its template instead of the fp32
// template: 9,024 B instead of 6,080 per plane, TWO workgroups per CU instead of three (76 KB of LDS each).
// MM == 2: the LEAN layout of the same (xl_*: 5,760 B per plane) — three workgroups per CU again.
template <int RX, int RZ, int G, bool XCORR, int NCH = FX_CH, bool P2 = false, int MM = 0>
__global__ void __launch_bounds__(64 * NCH, (P2 || MM == 1) ? 4 : 6) // <= 80 VGPRs: THREE workgroups (24 waves) per CU (LDS allows three)
sr_xcorr_fused9_kernel(LevelParams P, int C, const float* __restrict__ sr, const float* __restrict__ boxes,
                       const float* __restrict__ z, float* __restrict__ resp, float* __restrict__ x_debug,
                       int32_t* __restrict__ levels_out, SrOut S) {
    constexpr int HO = XCORR ? RX - RZ + 1 : 16;
    constexpr int NS = RX * G;                   // samples per axis
    // LDS image of the one-plane-per-wave correlation (xcorr_patch1.h): row stride 40, one plane per slot
    // P2: the correlation runs on plane PAIRS with 4x2 output patches per lane (xcorr_patch2.h: half the LDS read volume
    // per FMA of the one-plane form) by waves 0..NCH/2-1; the image then has that phase's strides
    constexpr int XS = P2 ? XP2_XS : (MM == 2 ? XL_XS : XP1_XS), XP = P2 ? XP2_XP : (MM == 2 ? 30 * XL_XS : 32 * XP1_XS), ZS = XP1_ZS,
                  ZP = MM == 1 ? XH_TZ_FLOATS : (MM == 2 ? XL_TZ_FLOATS : RZ * XP1_ZS);
    static_assert(MM == 0 || (XCORR && !P2 && RX == 30 && RZ == 15), "the matrix-pipe correlation is the 30 / 15 head's");
    static_assert(MM != 2 || ((XP * 4) % 128 == 0 && ((2 * XP + 2 * ZP) * 4) % 128 == 0), "lean layout: 128-byte aligned plane images");
    constexpr int RH = (RX + 1) / 2;             // pooled rows per batch
    static_assert((!XCORR || RX - RZ + 1 == 16) && RX <= 32 && G == 2 && RX * XS <= XP && 2 * RH * G <= 64,
                  "specialised for pooled sizes <= 32, g = 2 (and the 30/15/16 correlation geometry)");
    __shared__ __attribute__((aligned(128))) float sm[(NCH / 2) * (2 * XP + 2 * ZP)];
    __shared__ __attribute__((aligned(16))) int4 tab[2][64 + 2 * RH * G];   // y / x sample tables (+ zero pad)
    __shared__ int wbound[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7
    const long long t_start = S.trace ? (long long)__builtin_amdgcn_s_memtime() : 0ll;
    // The level's map size, padding, scale and base pointer are read from the kernel-argument segment with a DYNAMIC index
    // once the level is known: scalar loads that DEPEND on the workgroup's first memory round trip and — at a kernel's start,
    // with the scalar cache and the XCD's L2 freshly invalidated — may miss all the way to memory.  Their three 64-byte lines
    // (pointers; H, W; pad, scale) are requested here, with the first round trip; the values are "used" behind it (SMOT_KA_USE).
    const int ka_h = P.H[0], ka_w = P.W[0], ka_p = P.pad[0];
    const float ka_s = P.scale[0];
    const unsigned long long ka_f = reinterpret_cast<unsigned long long>(P.feat[0]);
#define SMOT_KA_USE() asm volatile("" ::"s"(ka_h), "s"(ka_w), "s"(ka_p), "s"(ka_s), "s"(ka_f))
#ifdef SMOT_DEBUG
    // experiment (measurement library, SMOT_FUSED_ABL = 100 + k): workgroups of the second dispatch wave start 512*k
    // cycles late, so that the two workgroups of a CU pool (LDS crossbar) and correlate (VALU) in anti-phase.
    // Measured: 17.0 us at no delay, 17.05-17.2 us for delays of 2 k .. 8 k cycles, 18.0 at 16 k — the late workgroup
    // catches up exactly: a CU's time is the SUM of its workgroups' instruction issue (1,500 vector instructions per
    // wave, 900 of them the correlation's FMAs), not a chain of latencies that a phase shift could overlap.
    if (S.abl >= 100 && (int)(blockIdx.y * gridDim.x + blockIdx.x) >= 256) {
        for (int d = 0; d < S.abl - 100; ++d) __builtin_amdgcn_s_sleep(8);
    }
#endif
    int n_assigned, cg_assigned, lvl_assigned = 0;
    float4 roi_assigned = make_float4(0.f, 0.f, 0.f, 0.f);
    // The ranking costs ~3 k cycles per box tensor at the head of every workgroup: one vector-memory round trip on
    // lines that all 256 CUs request at the same moment (the workgroup's own roi used to arrive through the scalar
    // cache in half that).  The 30x30 kernels earn it back (fused: 18.7 -> 17.4 us at 30 rois, 49 -> 42 us at 100);
    // the small template pooler does not (7.1 -> 7.3 us) and keeps grid order.  Measured and dropped: ranking in wave
    // 0 only with an LDS broadcast (same time: the cost is latency, not VALU contention); ranking from the search
    // regions alone with the level estimated (one tensor fewer, but the workgroup's exact level then costs a
    // dependent scalar load: 17.8 -> 18.4 us); the boxes through the scalar cache (sixteen s_load_dwordx4 per wave,
    // parked in LDS for the lanes: 106 SGPRs cost the kernel its occupancy target, 18.4 us).
    // Also measured and dropped for the 33..64-column windows that set the makespan: three 10-row blocks per plane with
    // the next block's row loads issued before the current block's gathers (96 VGPRs, two workgroups per CU):
    // bit-identical, 17.0 instead of 16.75 us at 30 rois, 12.9 instead of 12.3 at 16 (profiles/r02_fused_pipelined_wide.jsonl).
    int grid_row = blockIdx.y, grid_rows = gridDim.y;
    if constexpr (!XCORR) {
        if (S.hint_out != nullptr) {                 // extraction launch with one extra row in front: the hint writer
            if (grid_row == 0) {
                fx_write_hint<30, G>(P, boxes, S, gridDim.x, blockIdx.x, wave, lane);     // (the consumer's shape: 30x30 bins)
                return;
            }
            grid_row -= 1;
            grid_rows -= 1;
        }
    }
    // With a hint the entry's finished tables and its geometry stamp are requested NOW, beside fx_assign's own scalar load
    // of the entry: everything a hinted workgroup needs before its first feature load is ONE memory round trip.
    int4 tab_early = make_int4(0, 0, 0, 0);
    int4 ye_early = make_int4(0, 0, 0, 0), xe_early[2] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
    typedef int v8h_t __attribute__((ext_vector_type(8)));
    v8h_t g8 = {0, 0, 0, 0, -1, -1, -1, 0};
    if constexpr (XCORR && RX == 30) {
        if (S.hint_in != nullptr && S.order == 1 && S.n_valid == nullptr && gridDim.x >= 2 && gridDim.x <= 256) {
            const int k0 = (int)(blockIdx.y * gridDim.x + blockIdx.x) / (int)gridDim.y;        // = fx_assign's rank k
            const int* ent = reinterpret_cast<const int*>(S.hint_in) + (size_t)k0 * HINT_FLOATS;
            g8 = *reinterpret_cast<const __attribute__((address_space(4))) v8h_t*>(
                reinterpret_cast<unsigned long long>(ent) + 4ull * HINT_BOUNDS);
            if (wave < 2) tab_early = *reinterpret_cast<const int4*>(ent + (wave == 0 ? HINT_YTAB : HINT_XTAB) + 4 * lane);
            if constexpr (MM == 1) {
                // (round 6) ... and EVERY wave requests the entries it will use itself — the y entries of its own 15 pooled
                // rows (plane-pair form: the two waves of a pair split the rows), the two x entries of its lane's pooled column —
                // so that a hinted workgroup with a window of <= 64 columns needs neither the tables' image in LDS nor the
                // barrier behind it (`fast` below): the row loads go out ~1.2 k cycles earlier.
                ye_early = *reinterpret_cast<const int4*>(ent + HINT_YTAB + 4 * min(lane + (wave & 1) * RH * G, 63));
                const int pwl = (lane & 31) < RX ? (lane & 31) : 0;
                xe_early[0] = *reinterpret_cast<const int4*>(ent + HINT_XTAB + 4 * (pwl * G));
                xe_early[1] = *reinterpret_cast<const int4*>(ent + HINT_XTAB + 4 * (pwl * G + 1));
            }
        }
    }
    if constexpr (XCORR && RX == 30) {
        if (S.hint_in != nullptr && S.order == 1 && S.n_valid == nullptr && gridDim.x >= 2 && gridDim.x <= 256 &&
            blockIdx.y == 0 && wave == 2
#ifdef SMOT_DEBUG
            && S.abl != 6                 // A/B (measurement library, SMOT_FUSED_ABL=6): the round-4 kernel that trusted the hint
#endif
        )
            fx_verify_hint(P, sr, boxes, S.hint_in, blockIdx.x, gridDim.x, lane);
    }
    int k_assigned = -1;
    const bool have_roi = fx_assign(P, sr, boxes, S.n_valid, RX > 15 ? S.order : 0, grid_row, grid_rows,
                                    XCORR ? S.hint_in : nullptr, lane, &n_assigned, &cg_assigned, &roi_assigned,
                                    &lvl_assigned, &k_assigned);
    SMOT_KA_USE();
#undef SMOT_KA_USE
    const int n = __builtin_amdgcn_readfirstlane(n_assigned);
    const int cgrp = __builtin_amdgcn_readfirstlane(cg_assigned);
    if (S.n_valid != nullptr && n >= *S.n_valid) return;         // workgroup-uniform (scalar load)
    // (trace rows are indexed by the item, not by the workgroup that happened to take it)
#define FX_TRACE(SLOT)                                                                      \
    if (S.trace && tid == 0)                                                                \
        S.trace[((size_t)n * grid_rows + cgrp) * 8 + (SLOT)] = (long long)__builtin_amdgcn_s_memtime();
    if (S.trace && tid == 0) S.trace[((size_t)n * grid_rows + cgrp) * 8 + 0] = t_start;
    FX_TRACE(5)                                   // after the assignment

    float roi0 = roi_assigned.x, roi1 = roi_assigned.y, roi2 = roi_assigned.z, roi3 = roi_assigned.w;
    int lvl = lvl_assigned;
    if (!have_roi) {                             // workgroup-uniform
        roi0 = sr[(size_t)n * 4 + 0];
        roi1 = sr[(size_t)n * 4 + 1];
        roi2 = sr[(size_t)n * 4 + 2];
        roi3 = sr[(size_t)n * 4 + 3];
        lvl = 0;
        if (P.num_levels > 1) lvl = map_level(boxes + (size_t)n * 4, P.k_min, P.k_max);
    }
    const float roi[4] = {roi0, roi1, roi2, roi3};
    lvl = __builtin_amdgcn_readfirstlane(lvl);
    if (levels_out != nullptr && cgrp == 0 && tid == 0) levels_out[n] = lvl;
    if (!XCORR && S.sr != nullptr && cgrp == 0 && tid == 0) {
        const float4 o = search_region_of(roi[0], roi[1], roi[2], roi[3], S);
        S.sr[n * 4 + 0] = o.x;
        S.sr[n * 4 + 1] = o.y;
        S.sr[n * 4 + 2] = o.z;
        S.sr[n * 4 + 3] = o.w;
    }
    const int H = P.H[lvl], W = P.W[lvl], pad = P.pad[lvl];
    const float scale = P.scale[lvl];
    const float x1 = mul_rn(roi[0], scale), y1 = mul_rn(roi[1], scale);
    const float x2 = mul_rn(roi[2], scale), y2 = mul_rn(roi[3], scale);
    const float bin_h = div_rn(fmaxf(sub_rn(y2, y1), 1.0f), (float)RX);
    const float bin_w = div_rn(fmaxf(sub_rn(x2, x1), 1.0f), (float)RX);
    FX_TRACE(6)                                   // (level parameters and bin sizes known)
    // (Measured and dropped, profiles/r02x: one workgroup per (roi, FOUR channels) for rois whose window is wider
    // than 32 columns — two waves per plane, so that they do not set the makespan — with narrow rois using every
    // second workgroup: per-workgroup spans became equal (25-32 k cycles instead of 26 k / 48 k) but 608 working
    // workgroups no longer fit the chip's 512 resident slots (two rounds: 18.7 -> 24 us), and at three workgroups
    // per CU — 74 VGPRs with the one-plane correlation below — the kernel still took 23.0 us.)
    constexpr int nplanes = NCH;
    const int c0 = cgrp * NCH;
    // template of this wave's plane: issue the loads now, park them in LDS after the tables
    const bool owns = (wave < nplanes && c0 + wave < C);  // channel tails / four-plane workgroups: no plane here
    const int plane = n * C + c0 + wave;
    constexpr int NZ = XCORR ? (RZ * RZ + 63) / 64 : 1;
    float zreg[NZ];
    float zq[4] = {0.0f, 0.0f, 0.0f, 0.0f}, isz = 1.0f;
    if constexpr (MM != 0) {
        if (owns) xh_template_load(z + (size_t)plane * (RZ * RZ), lane, zq);
    } else if (XCORR && owns) {
        const float* __restrict__ zg = z + (size_t)plane * (RZ * RZ);
#pragma unroll
        for (int t = 0; t < NZ; ++t) zreg[t] = zg[min(lane + 64 * t, RZ * RZ - 1)];
    }

    // ---- sample tables: wave 0 builds the y axis, wave 1 the x axis (lane = sample), one barrier ------------------
    // (Every wave building both tables itself was measured: eight waves x two tables of ~150 VALU instructions each
    // cost more issue slots on the CU than the barrier they saved: 5.1 k instead of 4.3 k ticks.)
    // Entries are stored re-based and packed (16 bytes): y = {row byte offset lo, hi, weight lo, hi}, x = {window
    // column lo, hi, weight lo, hi}; consumers fetch an entry with one ds_read_b128.
    // With a hint entry the tables arrive FINISHED (fx_write_hint built them when the roi was made, one frame earlier):
    // waves 0 / 1 copy 1 KB each instead of ~150 vector instructions and two IEEE divisions; the entry's geometry stamp
    // must match this launch's level (another pad / map size: the tables are rebuilt here, the assignment stands).
    const bool hent = XCORR && RX == 30 && k_assigned >= 0 && g8[4] == pad && g8[5] == H && g8[6] == W &&
                      g8[7] == __float_as_int(scale);
    const int hb[4] = {g8[0], g8[1], g8[2], g8[3]};
    // the entries every wave fetched for itself are enough (no LDS tables, no barrier): a verified-geometry hint and a window
    // the plane-pair forms take (workgroup-uniform)
    const bool fast = MM == 1 && hent && hb[3] - hb[2] + 1 <= 64
#ifdef SMOT_DEBUG
                      && S.abl != 3 && S.abl != 12        // (3: the one-plane-per-wave A/B form reads the LDS tables; 12: A/B of this path)
#endif
        ;
    if (fast) {
    } else if (wave < 2 && hent) {
        tab[wave][lane] = tab_early;
        if (lane < 2 * RH * G) tab[wave][64 + lane] = make_int4(0, 0, 0, 0);      // the pad behind the table
    } else if (wave < 2) {
        int lo = 0, hi = 0;
        float wl = 0.0f, wh = 0.0f;
        if (lane < NS) {
            if (wave == 0) {
                axis_sample(y1, bin_h, G, lane, H, pad, &lo, &hi, &wl, &wh);
            } else {
                axis_sample(x1, bin_w, G, lane, W, pad, &lo, &hi, &wl, &wh);
            }
        }
        // Bounding window of the touched real cells.  Cell indices are non-decreasing in the sample index and a
        // zero low weight means "outside" (1 - frac is never 0), so the first touched entry holds the minimum and
        // the last the maximum: one ballot + two readlanes instead of a 6-step wave reduction.
        int mn = 0x7fffffff, mx = -1;
        const unsigned long long m = __ballot(wl != 0.0f || wh != 0.0f);
        if (m != 0ull) {
            mn = __builtin_amdgcn_readlane((wl != 0.0f) ? lo : hi, __ffsll((long long)m) - 1);
            mx = __builtin_amdgcn_readlane((wh != 0.0f) ? hi : lo, 63 - __clzll((long long)m));
        }
        // re-base: zero-weight entries point at a safe cell; rows become byte offsets inside a plane, columns
        // become window-relative
        const int rl = (wl != 0.0f) ? lo : mn, rh = (wh != 0.0f) ? hi : mn;
        int4 e;
        if (wave == 0) {
            e.x = (int)((unsigned)(rl * W) * 4u);
            e.y = (int)((unsigned)(rh * W) * 4u);
        } else {
            e.x = (wl != 0.0f) ? lo - mn : 0;
            e.y = (wh != 0.0f) ? hi - mn : 0;
        }
        e.z = __float_as_int(wl);
        e.w = __float_as_int(wh);
        tab[wave][lane] = e;
        if (lane < 2 * RH * G) tab[wave][64 + lane] = make_int4(0, 0, 0, 0);      // the pad behind the table
        if (lane == 0) {
            wbound[2 * wave] = mn;
            wbound[2 * wave + 1] = mx;
        }
    }
    FX_TRACE(7)                                   // (wave 0 at the table barrier)
    if (!fast) __syncthreads();
    const int ymin = hent ? hb[0] : wbound[0], ymax = hent ? hb[1] : wbound[1];
    const int xmin = hent ? hb[2] : wbound[2], xmax = hent ? hb[3] : wbound[3];
    if (ymax < ymin || xmax < xmin) {
        // every sample in the virtual zero border: pooled planes are exact zeros -> zero response
        if (owns) {
            if (XCORR) {
                for (int e = lane; e < HO * HO; e += 64) resp[(size_t)plane * HO * HO + e] = 0.0f;
                if (S.plane_max != nullptr && lane == 0) S.plane_max[plane] = 0.0f;
            }
            if (x_debug != nullptr)
                for (int e = lane; e < RX * RX; e += 64) x_debug[(size_t)plane * RX * RX + e] = 0.0f;
        }
        return;
    }
    const int ww = xmax - xmin + 1;
    // (Measured and dropped, round 4: s_setprio 1..3 for the waves of wide-window workgroups — twice a narrow one's pooling,
    // they set the kernel's makespan: 15.3-15.6 us at every priority against 15.3-15.4 without, 36.9 vs 36.9 at 100 tracks,
    // measure/gpu_r04_prio.sh.  Issue arbitration is not what holds them back.)
    FX_TRACE(1)

    if constexpr (MM != 0) {
        // (the template's Toeplitz rows are built inside the pooling, behind the issue of its first batch of row loads)
    } else if (XCORR && owns) {
        float* zs = sm + (wave >> 1) * (2 * XP + 2 * ZP) + 2 * XP + (wave & 1) * ZP;    // this wave's template
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            const int e = lane + 64 * t;
            if (e < RZ * RZ) {
                const int u = e / RZ;
                zs[u * ZS + (e - u * RZ)] = zreg[t];
            }
        }
    }
    FX_TRACE(2)

    // ---- pooling ----------------------------------------------------------------------------------------------
    const float* __restrict__ fbase = P.feat[lvl];
    const unsigned plane_bytes = (unsigned)(H * W) * 4u;
    // One batch = ROWS pooled rows of one plane (or of a plane pair side by side).  `PAIR`: lanes 32..63 pool the
    // wave's second plane and the two waves of the pair split the pooled rows; `CHUNKED`: windows wider than 64
    // columns (rare: small batches keep its loop-carried accumulators out of the register peak).
    auto pool = [&](auto pair_tag, auto chunk_tag, auto x2_tag, auto rows_tag) {
        constexpr bool PAIR = decltype(pair_tag)::value;
        constexpr bool CHUNKED = decltype(chunk_tag)::value;
        constexpr bool X2 = decltype(x2_tag)::value;           // a lane loads TWO adjacent window columns (33..64-column windows)
        constexpr int ROWS = decltype(rows_tag)::value;
        constexpr int NV = X2 ? 2 : 1;
        static_assert(!X2 || (PAIR && !CHUNKED), "two columns per lane is a form of the plane-pair mode");
        const int half = PAIR ? (lane >> 5) : 0;
        const int col = PAIR ? (lane & 31) : lane;
        const int pw = col < RX ? col : 0;
        // this wave's plane(s) and destination(s) in the LDS image of the correlation
        const int pl0 = PAIR ? 2 * (wave >> 1) : wave;                 // first plane within the workgroup
        const bool has0 = (c0 + pl0 < C);
        const bool has1 = PAIR && (c0 + pl0 + 1 < C);
        if (!has0) return;
        const bool mine = PAIR ? (half == 0 || has1) : true;           // this lane's plane exists
        const unsigned lane_plane = (PAIR && half == 1 && has1) ? plane_bytes : 0u;
        float* xdst = sm + ((pl0 + half) >> 1) * (2 * XP + 2 * ZP) + ((pl0 + half) & 1) * XP;
        // buffer resource of the (first) plane: wave-uniform base, offsets are 32-bit
        const float* pbase = fbase + (size_t)(c0 + pl0) * H * W;
        unsigned long long pa = reinterpret_cast<unsigned long long>(pbase);
        const unsigned pa_lo = __builtin_amdgcn_readfirstlane((unsigned)pa);
        const unsigned pa_hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void*>(((unsigned long long)pa_hi << 32) | pa_lo), 0, 0x7fffffff, 0x00020000);
        // horizontal taps of this lane's pooled column: entries 2*pw, 2*pw+1 of the x table
        int sxl[G], sxh[G];
        float hxw[G], lxw[G];
#pragma unroll
        for (int ix = 0; ix < G; ++ix) {
            const int4 e = fast ? xe_early[ix] : tab[1][pw * G + ix];
            sxl[ix] = e.x;
            sxh[ix] = e.y;
            hxw[ix] = __int_as_float(e.z);
            lxw[ix] = __int_as_float(e.w);
        }
        const int row0 = PAIR ? (wave & 1) * RH : 0;           // first pooled row of this wave (wave-uniform)
        const int nrows = PAIR ? RH : RX;
        const int nchunk = CHUNKED ? (ww + 63) >> 6 : 1;
        if constexpr (X2 && MM == 1) {
            // 33..64-column windows, matrix form (two workgroups per CU: 128 registers): the batches of a wave in a TWO-STAGE
            // PIPELINE — the next batch's row loads are issued as soon as this batch's vertical taps have consumed its load
            // registers, so that their memory round trip runs beside this batch's staging and horizontal taps instead of
            // behind them: ONE exposed round trip per wave instead of one per batch.  Three batches of five rows (40 load
            // registers in flight; two of eight rows spilled 30 registers).  Same loads, same FMA chains: bit-identical.
            typedef int v2i_t __attribute__((ext_vector_type(2)));
            constexpr int GR = (ROWS + 1) / 2, SW = 64;
            static_assert(RH % ROWS == 0 && GR * SW <= ROWS * XS, "full batches; staging fits a batch's own rows");
            const int wcol = min(2 * col, ww - 2);
            const unsigned voff = (unsigned)(xmin + wcol) * 4u + lane_plane;
            v2i_t vl[ROWS][G], vh[ROWS][G];
            // (fast: the wave's own 30 y entries sit in ye_early, batch b's at lanes 10 b ..; else a batch's entries from LDS)
            int4 ye = fast ? ye_early : tab[0][min(lane + row0 * G, 63 + 2 * RH * G)];
            int yoff = 0;
#define SMOT_FX_ISSUE()                                                                                             \
            _Pragma("unroll") for (int b = 0; b < ROWS; ++b)                                                        \
                _Pragma("unroll") for (int iy = 0; iy < G; ++iy) {                                                  \
                    vl[b][iy] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, __builtin_amdgcn_readlane(ye.x, yoff + b * G + iy), 0); \
                    vh[b][iy] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, __builtin_amdgcn_readlane(ye.y, yoff + b * G + iy), 0); \
                }
            SMOT_FX_ISSUE()
            __builtin_amdgcn_sched_barrier(0);
            if (owns) {
                unsigned char* tzw = reinterpret_cast<unsigned char*>(sm + (wave >> 1) * (2 * XP + 2 * ZP) + 2 * XP + (wave & 1) * ZP);
                isz = xh_template_store(zq, tzw, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int r0 = row0; r0 < row0 + nrows; r0 += ROWS) {
                const float wl = __int_as_float(ye.z), wh = __int_as_float(ye.w);
                float cs[ROWS][NV];
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        float c_ = 0.0f;
#pragma unroll
                        for (int iy = 0; iy < G; ++iy) {
                            const int e = yoff + b * G + iy;
                            c_ = fmaf(rl_f(wl, e), __int_as_float(vl[b][iy][k]), c_);
                            c_ = fmaf(rl_f(wh, e), __int_as_float(vh[b][iy][k]), c_);
                        }
                        cs[b][k] = c_;
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (r0 + ROWS < row0 + nrows) {                      // (wave-uniform) the next batch's entries and row loads
                    if (fast) yoff += ROWS * G;
                    else ye = tab[0][min(lane + (r0 + ROWS) * G, 63 + 2 * RH * G)];
                    SMOT_FX_ISSUE()
                }
                __builtin_amdgcn_sched_barrier(0);
                float acc[ROWS];
                float* stage = xdst + r0 * XS;
#pragma unroll
                for (int g0 = 0; g0 < ROWS; g0 += GR) {
#pragma unroll
                    for (int b = g0; b < g0 + GR && b < ROWS; ++b)
#pragma unroll
                        for (int k = 0; k < NV; ++k) stage[(b - g0) * SW + wcol + k] = cs[b][k];
                    float p[GR][G][2];
#pragma unroll
                    for (int b = g0; b < g0 + GR && b < ROWS; ++b)
#pragma unroll
                        for (int ix = 0; ix < G; ++ix) {
                            p[b - g0][ix][0] = stage[(b - g0) * SW + sxl[ix]];
                            p[b - g0][ix][1] = stage[(b - g0) * SW + sxh[ix]];
                        }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = g0; b < g0 + GR && b < ROWS; ++b)
#pragma unroll
                        for (int ix = 0; ix < G; ++ix) {
                            acc[b] = fmaf(hxw[ix], p[b - g0][ix][0], ix == 0 ? 0.0f : acc[b]);
                            acc[b] = fmaf(lxw[ix], p[b - g0][ix][1], acc[b]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int b = 0; b < ROWS; ++b) {
                    const int ph = r0 + b;
                    if (col < RX && ph < row0 + nrows && ph < RX && mine) xdst[ph * XS + col] = acc[b] * (1.0f / (float)(G * G));
                }
            }
#undef SMOT_FX_ISSUE
            return;
        }
#pragma unroll 1
        for (int r0 = row0; r0 < row0 + nrows; r0 += ROWS) {
            // this block's y entries: entry (b, iy) into lane b*G + iy (constant-lane readlanes below); entries past
            // the table (masked tail rows) read the zero pad: weight 0, offset 0
            // (fast: only the narrow plane-pair form comes here — ONE batch at r0 == row0, the entries ye_early holds)
            const int4 ye = (fast && PAIR && !CHUNKED) ? ye_early : tab[0][min(lane + r0 * G, 63 + 2 * RH * G)];
            const unsigned ol = (unsigned)ye.x, oh = (unsigned)ye.y;
            const float wl = __int_as_float(ye.z), wh = __int_as_float(ye.w);
            float acc[ROWS];
            if (CHUNKED) {
#pragma unroll
                for (int b = 0; b < ROWS; ++b) acc[b] = 0.0f;
            }
#pragma unroll 1
            for (int ch = 0; ch < nchunk; ++ch) {
                const int cbase = ch << 6;                                   // first window column of the chunk
                // first window column this lane loads.  X2: columns (wcol, wcol + 1); the last pair of an odd-width
                // window is (ww-2, ww-1) — lanes past the window repeat it (same values to the same staging slots)
                const int wcol = CHUNKED ? min(cbase + col, ww - 1) : (X2 ? min(2 * col, ww - 2) : min(col, ww - 1));
                const unsigned voff = (unsigned)(xmin + wcol) * 4u + lane_plane;
                float v[ROWS][G][2][NV];
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int iy = 0; iy < G; ++iy) {
                        const int e = b * G + iy;
                        if constexpr (X2) {
                            typedef int v2i_t __attribute__((ext_vector_type(2)));
                            const v2i_t l2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, __builtin_amdgcn_readlane((int)ol, e), 0);
                            const v2i_t h2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, __builtin_amdgcn_readlane((int)oh, e), 0);
                            v[b][iy][0][0] = __int_as_float(l2.x);
                            v[b][iy][0][NV - 1] = __int_as_float(l2.y);
                            v[b][iy][1][0] = __int_as_float(h2.x);
                            v[b][iy][1][NV - 1] = __int_as_float(h2.y);
                        } else {
                            v[b][iy][0][0] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                rsrc, voff, __builtin_amdgcn_readlane((int)ol, e), 0));
                            v[b][iy][1][0] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                rsrc, voff, __builtin_amdgcn_readlane((int)oh, e), 0));
                        }
                    }
                // fences: hipcc otherwise sinks the loads to their first use (4 loads, wait, use, next 4 loads ...)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MM != 0) {
                    // the template's Toeplitz rows, while the first batch of row loads is in flight (the template's own loads
                    // were issued at the kernel's start: ~1 k cycles of conversion and LDS stores off the workgroup's chain)
                    if (r0 == row0 && ch == 0 && owns) {
                        unsigned char* tzw = reinterpret_cast<unsigned char*>(sm + (wave >> 1) * (2 * XP + 2 * ZP) + 2 * XP + (wave & 1) * ZP);
                        isz = MM == 2 ? xl_template_store(zq, tzw, lane) : xh_template_store(zq, tzw, lane);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                float cs[ROWS][NV];
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        float c_ = 0.0f;
#pragma unroll
                        for (int iy = 0; iy < G; ++iy) {
                            const int e = b * G + iy;
                            c_ = fmaf(rl_f(wl, e), v[b][iy][0][k], c_);
                            c_ = fmaf(rl_f(wh, e), v[b][iy][1][k], c_);
                        }
                        cs[b][k] = c_;
                    }
                __builtin_amdgcn_sched_barrier(0);
                // Horizontal taps.  A lane (= pooled column) needs the column sums of up to four window columns (other
                // lanes' values).  ds_bpermute_b32 cost 10.6 LDS cycles per wave instruction here (SQ_LDS_IDX_ACTIVE,
                // profiles/r02aa_pmc_counters.md: two thirds of the LDS pipe's busy time) — so the sums of a group of
                // rows are staged in LDS instead (one ds_write per row, 2 array cycles) and every tap is a plain
                // ds_read_b32 (2 cycles): 10 instead of 42 LDS cycles per pooled row.  The staging rows are this
                // wave's own not-yet-written rows of the plane image (the batch's results are stored after its last
                // gather; LDS operations of one wave execute in order, so neither a wait nor a barrier is needed).
                // Rows are processed in two groups to keep the register peak (column sums + gathered taps + tables)
                // where the correlation phase is scheduled for.  Same values, same FMA order: bit-identical.
                constexpr int GR = ((P2 || MM == 2) && X2) ? 3 : (ROWS + 1) / 2;
                if constexpr (!CHUNKED) {
                    constexpr int SW = (X2 || !PAIR) ? 64 : 32;                      // staged floats per row (and plane)
                    static_assert(GR * SW <= (RH - (RH / ROWS) * ROWS == 0 ? ROWS : RH - (RH / ROWS) * ROWS) * XS,
                                  "staging fits the image rows of the wave's LAST batch");
                    float* stage = xdst + r0 * XS;
#pragma unroll
                    for (int g0 = 0; g0 < ROWS; g0 += GR) {
#pragma unroll
                        for (int b = g0; b < g0 + GR && b < ROWS; ++b)
#pragma unroll
                            for (int k = 0; k < NV; ++k) stage[(b - g0) * SW + wcol + k] = cs[b][k];
                        float p[GR][G][2];
#pragma unroll
                        for (int b = g0; b < g0 + GR && b < ROWS; ++b)
#pragma unroll
                            for (int ix = 0; ix < G; ++ix) {
                                p[b - g0][ix][0] = stage[(b - g0) * SW + sxl[ix]];
                                p[b - g0][ix][1] = stage[(b - g0) * SW + sxh[ix]];
                            }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int b = g0; b < g0 + GR && b < ROWS; ++b)
#pragma unroll
                            for (int ix = 0; ix < G; ++ix) {
                                // (acc[b] is not live across the loads here: first written in this group)
                                acc[b] = fmaf(hxw[ix], p[b - g0][ix][0], ix == 0 ? 0.0f : acc[b]);
                                acc[b] = fmaf(lxw[ix], p[b - g0][ix][1], acc[b]);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    // windows wider than 64 columns (rare): cross-lane gathers per 64-column chunk, two groups of rows
                    int al[G], ah[G];
                    bool inl[G], inh[G];
#pragma unroll
                    for (int ix = 0; ix < G; ++ix) {
                        const int tl = sxl[ix] - cbase, th = sxh[ix] - cbase;
                        inl[ix] = (unsigned)tl < 64u;
                        inh[ix] = (unsigned)th < 64u;
                        al[ix] = (tl & 63) << 2;                                  // ds_bpermute takes byte addresses
                        ah[ix] = (th & 63) << 2;
                    }
#pragma unroll
                    for (int g0 = 0; g0 < ROWS; g0 += GR) {
                        float p[GR][G][2];
#pragma unroll
                        for (int b = g0; b < g0 + GR && b < ROWS; ++b)
#pragma unroll
                            for (int ix = 0; ix < G; ++ix) {
                                p[b - g0][ix][0] = __int_as_float(__builtin_amdgcn_ds_bpermute(al[ix], __float_as_int(cs[b][0])));
                                p[b - g0][ix][1] = __int_as_float(__builtin_amdgcn_ds_bpermute(ah[ix], __float_as_int(cs[b][0])));
                            }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int b = g0; b < g0 + GR && b < ROWS; ++b)
#pragma unroll
                            for (int ix = 0; ix < G; ++ix) {
                                // a tap outside the chunk adds nothing (not even 0 * garbage)
                                acc[b] = inl[ix] ? fmaf(hxw[ix], p[b - g0][ix][0], acc[b]) : acc[b];
                                acc[b] = inh[ix] ? fmaf(lxw[ix], p[b - g0][ix][1], acc[b]) : acc[b];
                            }
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < ROWS; ++b) {
                const int ph = r0 + b;
                if (col < RX && ph < row0 + nrows && ph < RX && mine) xdst[ph * XS + col] = acc[b] * (1.0f / (float)(G * G));   // exact: /4
            }
        }
    };
    // <= 32 columns: a wave pools two planes side by side (lane = plane half x column), the two waves of a plane
    // pair split the pooled rows: ONE batch of 60 row loads per wave.  33..64 columns: the same plane-pair form with
    // TWO adjacent columns per lane (8-byte loads), two batches of 8 rows: a wide window now costs the load
    // instructions of a narrow one (it was one plane per wave in two batches of 15 rows — twice the load, FMA and
    // tap instructions per plane, and those workgroups set the kernel's makespan: 25 k vs 12 k cycles of pooling).
    // Wider than 64 (degenerate aspect ratios): one plane per wave in 64-column chunks.
    if (ww <= 32) {
        pool(std::true_type{}, std::false_type{}, std::false_type{}, std::integral_constant<int, RH>{});
#ifdef SMOT_DEBUG
    } else if (ww <= 64 && S.abl == 3 && MM != 2) {      // A/B (measurement library, SMOT_FUSED_ABL=3): one plane per wave, two batches
        if constexpr (MM != 2) pool(std::false_type{}, std::false_type{}, std::false_type{}, std::integral_constant<int, RH>{});
#endif
    } else if (ww <= 64) {
        if constexpr (MM == 1) pool(std::true_type{}, std::false_type{}, std::true_type{}, std::integral_constant<int, RH / 3>{});
        else pool(std::true_type{}, std::false_type{}, std::true_type{}, std::integral_constant<int, (RH + 1) / 2>{});
    } else {
        pool(std::false_type{}, std::true_type{}, std::false_type{}, std::integral_constant<int, (RH + 2) / 3>{});
    }
    FX_TRACE(3)
    __syncthreads();                                      // every plane of the workgroup pooled (pairs share rows)
    if (x_debug != nullptr && owns) {
        const float* xs = sm + (wave >> 1) * (2 * XP + 2 * ZP) + (wave & 1) * XP;
        for (int e = lane; e < RX * RX; e += 64) {
            const int r = e / RX;
            x_debug[(size_t)plane * RX * RX + e] = xs[r * XS + (e - r * RX)];
        }
    }
    if constexpr (XCORR) {
#ifdef SMOT_DEBUG
        if (S.abl == 2) return;                           // timing ablation: measurement library only
#endif
        // every wave correlates its own plane (2x2 output patches per lane): all eight waves work, and the phase
        // needs few enough registers for three workgroups per CU
        if constexpr (P2) {
            if (wave < nplanes / 2 && c0 + 2 * wave < C) {
                const float* xs2 = sm + wave * (2 * XP + 2 * ZP);
                xcorr_patch2_compute<RX, RZ, 0>(xs2, xs2 + 2 * XP, lane, resp, n * C + c0 + 2 * wave, n * C + min(C, c0 + NCH));
            }
        } else if (owns) {
            float* xs1 = sm + (wave >> 1) * (2 * XP + 2 * ZP) + (wave & 1) * XP;
            const float* zs1 = sm + (wave >> 1) * (2 * XP + 2 * ZP) + 2 * XP + (wave & 1) * ZP;
            if constexpr (MM == 1) xh_correlate<RX, RZ>(xs1, reinterpret_cast<const unsigned char*>(zs1), isz, lane, resp, plane, S.plane_max);
            else if constexpr (MM == 2) xl_correlate<RX, RZ>(xs1, reinterpret_cast<const unsigned char*>(zs1), isz, lane, resp, plane, S.plane_max);
            else xcorr_patch1_compute<RX, RZ, true>(xs1, zs1, lane, resp, plane, S.plane_max);
        }
    }
    FX_TRACE(4)
#undef FX_TRACE
}

}  // namespace smot


// Launch the pooling(+correlation) kernel: generation 3 in the product library; the measurement library can
// select generation 2 (SMOT_FUSED_GEN=2) for A/B runs.
namespace smot {
constexpr int FX_ORDER_DEFAULT = 1;
static inline int fused_order() {          // SMOT_FUSED_ORDER: 0 = default, 1..3 = a form of fx_assign, 4 = grid order
    const int k = knobs().fused_order;
    return k == 0 ? FX_ORDER_DEFAULT : (k == 4 ? 0 : k);
}
// An order hint is written / honoured for this many rois (fx_write_hint / fx_assign: one wave ranks up to 256 rois; the
// list has the default order's form; the measurement library's generation-2 kernel knows nothing of it).
static inline bool order_hint_rois(int N, bool consumer) {
    if (knobs().fused_gen == 2 || knobs().no_hint >= (consumer ? 1 : 2)) return false;      // (constant false: product)
    return N >= 2 && N <= 256 && fused_order() == 1;
}
template <int RX, bool XCORR>
static void launch_fused(dim3 grid, hipStream_t st, const LevelParams& P, int C, const float* rois, const float* boxes,
                         const float* z, float* resp, float* out, int32_t* levels_out, const SrOut& S) {
#ifdef SMOT_DEBUG
    if (knobs().fused_gen == 2 && S.n_valid == nullptr) {      // (generation 2 has no masked form)
        hipLaunchKernelGGL((sr_xcorr_fused8_kernel<RX, 15, 2, XCORR>), grid, dim3(512), 0, st, P, C, rois, boxes, z, resp,
                           out, levels_out, S);
        return;
    }
#endif
    if constexpr (RX == 30 && XCORR) {
#ifdef SMOT_DEBUG
        if (knobs().fused_abl == 8 || knobs().fused_abl == 9) {
            // A/B (measurement library, SMOT_FUSED_ABL=8): the fp32 FMA correlation of rounds 2-5; 9: the same with 26 KB of
            // unused dynamic LDS, i.e. at TWO workgroups per CU like the matrix form (what the third workgroup is worth)
            SMOT_LAUNCH((sr_xcorr_fused9_kernel<RX, 15, 2, XCORR>), grid, dim3(512), knobs().fused_abl == 9 ? 26000 : 0, st, P, C, rois,
                        boxes, z, resp, out, levels_out, S);
            return;
        }
#endif
#ifdef SMOT_DEBUG
        if (knobs().fused_abl == 10) {
            // A/B (SMOT_FUSED_ABL=10): the matrix form's LEAN layout (5,760 B per plane, three workgroups per CU).  Measured,
            // same session, hinted: 15.66 vs 15.35 us at 30 tracks, 38.8 vs 39.65 at 100 — the third workgroup buys back what
            // the extra reads and funnel shifts cost (correlation phase 7.5 k vs 5.8 k cycles per workgroup) and no more.
            SMOT_LAUNCH((sr_xcorr_fused9_kernel<RX, 15, 2, XCORR, FX_CH, false, 2>), grid, dim3(512), 0, st, P, C, rois, boxes, z,
                        resp, out, levels_out, S);
            return;
        }
#endif
        SMOT_LAUNCH((sr_xcorr_fused9_kernel<RX, 15, 2, XCORR, FX_CH, false, 1>), grid, dim3(512), 0, st, P, C, rois, boxes, z,
                    resp, out, levels_out, S);
    } else {
        SMOT_LAUNCH((sr_xcorr_fused9_kernel<RX, 15, 2, XCORR>), grid, dim3(512), 0, st, P, C, rois, boxes, z, resp, out,
                    levels_out, S);
    }
}

// Separable stand-alone pooling for the two EMM pooler shapes (called by smot_roi_align_levels_fwd).
int launch_roi_pool_separable(const LevelParams& P, int C, const float* rois, const float* level_boxes, int R,
                              int out_size, float* out, int32_t* levels_out, hipStream_t st) {
    dim3 grid(R, (C + FX_CH - 1) / FX_CH);
    SrOut none = {nullptr, 0.f, 0.f, 0.f, 0.f, g_trace, 0, nullptr, fused_order(), nullptr, nullptr, 0};
    if (out_size == 30) {
        launch_fused<30, false>(grid, st, P, C, rois, level_boxes, nullptr, nullptr, out, levels_out, none);
    } else if (out_size == 7) {        // the box head's 7x7 pooler (box_head.py:46, roi_heads.py:60-84): same kernel
        SMOT_LAUNCH((sr_xcorr_fused9_kernel<7, 15, 2, false>), grid, dim3(512), 0, st, P, C, rois, level_boxes,
                    (const float*)nullptr, (float*)nullptr, out, levels_out, none);
    } else {
        launch_fused<15, false>(grid, st, P, C, rois, level_boxes, nullptr, nullptr, out, levels_out, none);
    }
    return check_launch("roi_pool_separable");
}

int launch_extract_cache(const float* const* feats, const int* heights, const int* widths, const float* scales,
                         int num_levels, int C, const float* boxes, int N, int rz, float pad_pixels, float half_e,
                         float two_e, float min_wh, float* templates, float* sr, const int* n_valid, float* order_hint,
                         hipStream_t st, int hint_extra_rows) {
    LevelParams P;
    const int rc = fill_level_params(&P, feats, heights, widths, nullptr, scales, num_levels, "emm_extract_cache");
    if (rc) return rc;
    SMOT_REQUIRE(boxes && templates && sr, "emm_extract_cache: null pointer");
    if (!order_hint_rois(N, false) || rz != 15) order_hint = nullptr;     // (the hint's consumer is the 30/15 head)
    SMOT_REQUIRE(order_hint == nullptr || ((((uintptr_t)order_hint) & 31) == 0 && (((uintptr_t)boxes) & 15) == 0),
                 "emm_extract_cache: the order hint must be 32-byte aligned (and the boxes 16-byte aligned)");
    // with a hint to write: one extra row of workgroups in front, of which the first ranks the rois (fx_write_hint)
    dim3 grid(N, (C + FX_CH - 1) / FX_CH + (order_hint != nullptr ? 1 : 0));
    SrOut S = {sr, pad_pixels, half_e, two_e, min_wh, g_trace, 0, n_valid, fused_order(), order_hint, nullptr,
               n_valid != nullptr ? hint_extra_rows : 0};
    // the next frame's search-region pooling runs on maps zero-padded by int(pad_pixels / stride) cells per level
    // (track_utils.py:94-96); the hint's finished tables are built against exactly that
    for (int l = 0; l < num_levels && l < SMOT_MAX_LEVELS; ++l) S.plan_pad[l] = (int)(pad_pixels * scales[l]);
    if (rz == 7) {                 // the second yaml family's template (DLA_34_FPN_EMM_AOT.yaml:52-63): same kernel, 7x7 bins
        SMOT_LAUNCH((sr_xcorr_fused9_kernel<7, 15, 2, false>), grid, dim3(512), 0, st, P, C, boxes, boxes, (const float*)nullptr,
                    (float*)nullptr, templates, (int32_t*)nullptr, S);
    } else {
        launch_fused<15, false>(grid, st, P, C, boxes, boxes, nullptr, nullptr, templates, nullptr, S);
    }
    return check_launch("emm_extract_cache");
}

#ifdef SMOT_DEBUG
// measure/csrc/sr_xcorr_plan.hip (measurement library only): the round-4 plan-driven, barrier-free generation
int launch_roi_plans(const LevelParams& P, const float* sr, const float* boxes, int N, float* plans, hipStream_t st);
int launch_fused10(const LevelParams& P, int C, const float* plans, const float* z, int N, float* resp, float* x_debug,
                   hipStream_t st);
int fused10_plan_floats();
#endif

// Pooling + correlation with an optional order hint for its rois (smot_emm_track_fwd; the stand-alone operator
// smot_sr_xcorr_fused_fwd passes none).
int sr_xcorr_fused_impl(const float* const* feats, const int* heights, const int* widths, const int* pad_cells,
                        const float* scales, int num_levels, int C, const float* boxes, const float* sr,
                        const float* templates, int N, float* resp, float* x_debug, const float* order_hint,
                        hipStream_t st, const int** hint_status, float* plane_max) {
    LevelParams P;
    const int rc = fill_level_params(&P, feats, heights, widths, pad_cells, scales, num_levels, "sr_xcorr_fused");
    if (rc) return rc;
    if (!order_hint_rois(N, true)) order_hint = nullptr;
    // the status word of a hint this launch honours (and verifies: fx_verify_hint) — for the head's decode kernel
    if (hint_status != nullptr)
        *hint_status = order_hint != nullptr ? reinterpret_cast<const int*>(order_hint) + HINT_STATUS : nullptr;
    SMOT_REQUIRE(order_hint == nullptr || (((uintptr_t)order_hint) & 31) == 0,
                 "sr_xcorr_fused: the order hint must be 32-byte aligned");
    dim3 grid(N, (C + FX_CH - 1) / FX_CH);
#ifdef SMOT_DEBUG
    if (knobs().fused_gen == 10) {             // stage 1 (measurement library): generation 4 with a stand-alone plan launch
        static float* plans = nullptr;
        if (plans == nullptr && hipMalloc(&plans, (size_t)4096 * fused10_plan_floats() * 4) != hipSuccess) return SMOT_ERR_BAD_ARG;
        SMOT_REQUIRE(N <= 4096, "fused10 debug: at most 4096 rois");
        int rcp = launch_roi_plans(P, sr, boxes, N, plans, st);
        if (rcp) return rcp;
        timer_mark(0, 0, st);
        rcp = launch_fused10(P, C, plans, templates, N, resp, x_debug, st);
        timer_mark(0, 1, st);
        if (rcp == SMOT_OK && plane_max != nullptr) rcp = launch_plane_absmax(resp, N * C, 256, plane_max, st);
        return rcp;
    }
#endif
    timer_mark(0, 0, st);
    SrOut none = {nullptr, 0.f, 0.f, 0.f, 0.f, g_trace, knobs().fused_abl, nullptr, fused_order(), nullptr, order_hint, 0};
    none.plane_max = plane_max;
#ifdef SMOT_DEBUG
    if (plane_max != nullptr && (knobs().fused_gen == 2 || knobs().fused_abl == 5)) {      // kernels that do not write them
        none.plane_max = nullptr;
        launch_fused<30, true>(grid, st, P, C, sr, boxes, templates, resp, x_debug, nullptr, none);
        timer_mark(0, 1, st);
        const int rcl = check_launch("sr_xcorr_fused");
        return rcl ? rcl : launch_plane_absmax(resp, N * C, 256, plane_max, st);
    }
    // A/B (measurement library, SMOT_FUSED_ABL=4): four channels per workgroup, twice the workgroups (960 of four waves at
    // 30 tracks: finer balance, twice the table builds).  Bit-identical; 17.7 vs 17.7 us at 30 tracks, 39.0 vs 39.7 at 100
    // (measure/fused_ab.py): not worth a second configuration.
    if (knobs().fused_abl == 5) {       // A/B: the correlation on plane pairs with 4x2 patches (half the LDS read volume)
        hipEvent_t e0_, e1_;
        if (timer_take(&e0_, &e1_)) {
            hipExtLaunchKernelGGL((sr_xcorr_fused9_kernel<30, 15, 2, true, 8, true>), grid, dim3(512), 0, st, e0_, e1_, 0, P, C, sr, boxes,
                                  templates, resp, x_debug, (int32_t*)nullptr, none);
        } else {
            hipLaunchKernelGGL((sr_xcorr_fused9_kernel<30, 15, 2, true, 8, true>), grid, dim3(512), 0, st, P, C, sr, boxes,
                               templates, resp, x_debug, (int32_t*)nullptr, none);
        }
        timer_mark(0, 1, st);
        return check_launch("sr_xcorr_fused");
    }
    if (knobs().fused_abl == 4) {
        hipLaunchKernelGGL((sr_xcorr_fused9_kernel<30, 15, 2, true, 4>), dim3(N, (C + 3) / 4), dim3(256), 0, st, P, C, sr, boxes,
                           templates, resp, x_debug, (int32_t*)nullptr, none);
        timer_mark(0, 1, st);
        return check_launch("sr_xcorr_fused");
    }
#endif
    launch_fused<30, true>(grid, st, P, C, sr, boxes, templates, resp, x_debug, nullptr, none);
    timer_mark(0, 1, st);
    return check_launch("sr_xcorr_fused");
}
}  // namespace smot

#ifdef SMOT_DEBUG
// measurement library only: the stand-alone pooling + correlation launch WITH an order hint (phase traces / A/B runs of
// the kernel alone; the product passes hints through smot_emm_track_fwd)
extern "C" int smot_debug_sr_xcorr_fused_hint_fwd(const float* const* feats, const int* heights, const int* widths,
                                                  const int* pad_cells, const float* scales, int num_levels, int C,
                                                  const float* boxes, const float* sr, const float* templates, int N,
                                                  float* resp, const float* order_hint, smot_stream_t stream) {
    return smot::sr_xcorr_fused_impl(feats, heights, widths, pad_cells, scales, num_levels, C, boxes, sr, templates, N,
                                     resp, nullptr, order_hint, (hipStream_t)stream, nullptr, nullptr);
}
#endif

extern "C" int smot_emm_order_hint_status(const float* order_hint, int* status_host, smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(order_hint != nullptr && status_host != nullptr, "order_hint_status: null pointer");
    const hipError_t e = hipMemcpyAsync(status_host, reinterpret_cast<const int*>(order_hint) + HINT_STATUS, sizeof(int),
                                        hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e != hipSuccess) {
        set_error("order_hint_status: hipMemcpyAsync: %s", hipGetErrorString(e));
        return (int)e;
    }
    return SMOT_OK;
}

extern "C" long long smot_emm_order_hint_floats(int N, int rz, int sampling_ratio) {
    using namespace smot;
    if (!(rz == 15 && sampling_ratio == 2) || knobs().roi_generic || !order_hint_rois(N, false)) return 0;
    return (long long)N * HINT_FLOATS;
}

extern "C" int smot_sr_xcorr_fused_fwd(const float* const* feats, const int* heights, const int* widths,
                                       const int* pad_cells, const float* scales, int num_levels, int C,
                                       const float* boxes, const float* sr, const float* templates, int N, int rx,
                                       int rz, int sampling_ratio, float* resp, float* x_debug,
                                       smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(N >= 0 && C > 0, "sr_xcorr_fused: bad sizes N=%d C=%d", N, C);
    if (!(rx == 30 && rz == 15 && sampling_ratio == 2)) {
        set_error("sr_xcorr_fused: only Rx=30, Rz=15, sampling_ratio=2 is fused (got %d, %d, %d); use "
                  "smot_roi_align_levels_fwd + smot_xcorr_dw_fwd", rx, rz, sampling_ratio);
        return SMOT_ERR_UNSUPPORTED;
    }
    if (N == 0) return SMOT_OK;
    SMOT_REQUIRE(boxes && sr && templates && resp, "sr_xcorr_fused: null pointer");
    return sr_xcorr_fused_impl(feats, heights, widths, pad_cells, scales, num_levels, C, boxes, sr, templates, N, resp,
                               x_debug, nullptr, (hipStream_t)stream, nullptr, nullptr);
}
