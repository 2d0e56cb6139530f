// K1+K2 for the reference's SECOND yaml family (configs/dla/DLA_34_FPN_EMM_AOT.yaml:52-63: 7x7 templates, search regions
// x5 -> 35x35 bins, 29x29 responses): search-region pooling feeding the depthwise correlation inside one kernel.
//
// Replaces, for this shape, the pair roi_align_levels_kernel<2> (SRPooler.forward, reference EMM/sr_pool.py:53-91 ->
// ROIAlign, with the virtual zero padding of track_utils.py:87-107) + xcorr_dw_rowpatch_kernel<35, 7> (xcorr_depthwise,
// EMM/xcorr.py:37-46): the [N, C, 35, 35] search-region tensor (18.8 MB at 30 tracks, written and read back) never
// reaches HBM and one launch disappears.
//
// The arithmetic is the two kernels' own, operation for operation, so the responses are bit-identical to the pair's
// (tests/test_hip_parity.py): a bin is the sum of its 2 x 2 samples' four weighted corners in the generic kernel's order
// (w1 v1 + w2 v2 + w3 v3 + w4 v4 per sample, accumulated sample by sample, / 4), a response is one fmaf chain over the 49
// taps, u-major / v-minor.  The search windows of this family are 50 x 100 .. 80 x 160 cells — beyond what a workgroup
// can stage — so the corners are gathered straight from the map, as the generic kernel does for such windows (a separable
// / staged form was measured slower there: measure/r03_generic_roi_separable.patch).
//
//   workgroup = (roi, 4 channels), 256 threads:
//     tables   the per-axis sample bookkeeping (70 + 70 samples: cell pair, weight pair), once for the four channels;
//     pooling  thread = bins t, t + 256, ...: the bin's tables in registers, channels inner; the pooled planes go to LDS
//              rows padded to 40 floats (the correlation's row segments are read as float4);
//     xcorr    thread = (response row, group of four response columns): the template's 49 taps in registers, per window row
//              one 10-float segment feeds 28 FMAs; plane after plane.
#include "roi_common.h"
#include "knobs.h"
#include "tower_common.h"      // plane_max_wave

namespace smot {

constexpr int SX_CH = 4;             // channels per workgroup

template <int RX, int RZ, int G>
__global__ void __launch_bounds__(256)
sr_xcorr_gather_kernel(LevelParams P, int C, const float* __restrict__ rois, const float* __restrict__ level_boxes,
                       const float* __restrict__ z, float* __restrict__ resp, float* __restrict__ plane_max) {
    constexpr int HO = RX - RZ + 1;
    constexpr int NQ = (HO + 3) / 4;                     // column groups per response row
    constexpr int XS = ((4 * NQ + RZ - 1 + 3) / 4) * 4;  // padded LDS row of a pooled plane (floats)
    constexpr int SEG = 4 + RZ - 1;                      // floats a thread needs of a window row
    constexpr int NS = RX * G;                           // samples per axis
    static_assert(HO * NQ <= 256 && XS >= RX, "one pass of 256 threads per plane");
    __shared__ __attribute__((aligned(16))) float xs[SX_CH][RX * XS];
    __shared__ float zs[SX_CH][RZ * RZ];
    __shared__ int y_lo[NS], y_hi[NS], x_lo[NS], x_hi[NS];
    __shared__ float wy_lo[NS], wy_hi[NS], wx_lo[NS], wx_hi[NS];
    __shared__ int wbound[4];              // ymin, ymax, xmin, xmax of the cells with non-zero weight
    __shared__ unsigned pmax[SX_CH];       // largest |response| of the workgroup's planes (bits of a non-negative float): the
                                           // tower kernel's split form scales a track's response by a power of two chosen
                                           // from them (tower_wino.hip) — written here, no launch of its own
    if (threadIdx.x < SX_CH) pmax[threadIdx.x] = 0u;

    const int r = blockIdx.x;
    const float* roi = rois + (size_t)r * 4;
    int lvl = 0;
    if (P.num_levels > 1) lvl = map_level(level_boxes + (size_t)r * 4, P.k_min, P.k_max);
    const int H = P.H[lvl], W = P.W[lvl], pad = P.pad[lvl];
    const float scale = P.scale[lvl];
    const float x1 = mul_rn(roi[0], scale), y1 = mul_rn(roi[1], scale);
    const float x2 = mul_rn(roi[2], scale), y2 = mul_rn(roi[3], scale);
    const float roi_w = fmaxf(sub_rn(x2, x1), 1.0f);
    const float roi_h = fmaxf(sub_rn(y2, y1), 1.0f);
    const float bin_h = div_rn(roi_h, (float)RX);
    const float bin_w = div_rn(roi_w, (float)RX);
    const int c0 = blockIdx.y * SX_CH;
    const int nch = min(C, c0 + SX_CH) - c0;

    // templates of the workgroup's planes: requested now, needed after the pooling
    for (int e = threadIdx.x; e < nch * RZ * RZ; e += 256) {
        const int cl = e / (RZ * RZ);
        zs[cl][e - cl * (RZ * RZ)] = z[((size_t)r * C + c0 + cl) * (RZ * RZ) + (e - cl * (RZ * RZ))];
    }
    if (threadIdx.x == 0) {
        wbound[0] = 0x7fffffff;
        wbound[1] = -1;
        wbound[2] = 0x7fffffff;
        wbound[3] = -1;
    }
    // the pad columns of the plane images (read by the correlation's float4 segments, weightless: they only meet taps of
    // outputs past the 29th column, which are not stored — but must be finite)
    for (int e = threadIdx.x; e < SX_CH * RX * (XS - RX); e += 256) {
        const int cl = e / (RX * (XS - RX)), q = e - cl * (RX * (XS - RX));
        xs[cl][(q / (XS - RX)) * XS + RX + q % (XS - RX)] = 0.0f;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < 2 * NS; s += 256) {
        int lo, hi;
        float wl, wh;
        if (s < NS) {
            axis_sample(y1, bin_h, G, s, H, pad, &lo, &hi, &wl, &wh);
            y_lo[s] = lo;
            y_hi[s] = hi;
            wy_lo[s] = wl;
            wy_hi[s] = wh;
        } else {
            const int sx = s - NS;
            axis_sample(x1, bin_w, G, sx, W, pad, &lo, &hi, &wl, &wh);
            x_lo[sx] = lo;
            x_hi[sx] = hi;
            wx_lo[sx] = wl;
            wx_hi[sx] = wh;
        }
        const int b = (s < NS) ? 0 : 2;
        if (wl != 0.0f) {
            atomicMin(&wbound[b], lo);
            atomicMax(&wbound[b + 1], lo);
        }
        if (wh != 0.0f) {
            atomicMin(&wbound[b], hi);
            atomicMax(&wbound[b + 1], hi);
        }
    }
    __syncthreads();
    const int ymin = wbound[0], ymax = wbound[1], xmin = wbound[2], xmax = wbound[3];
    if (ymax < ymin || xmax < xmin) {
        // every sample lies in the virtual zero border: pooled planes are exact zeros -> zero responses
        for (int e = threadIdx.x; e < nch * HO * HO; e += 256) resp[((size_t)r * C + c0) * (HO * HO) + e] = 0.0f;
        if (plane_max != nullptr && (int)threadIdx.x < nch) plane_max[(size_t)r * C + c0 + threadIdx.x] = 0.0f;
        return;
    }
    // re-base the tables: map-relative row offsets; a weightless entry points at a cell inside the window
    for (int s = threadIdx.x; s < 2 * NS; s += 256) {
        if (s < NS) {
            const int lo = (wy_lo[s] != 0.0f) ? y_lo[s] : ymin;
            const int hi = (wy_hi[s] != 0.0f) ? y_hi[s] : ymin;
            y_lo[s] = lo * W;
            y_hi[s] = hi * W;
        } else {
            const int sx = s - NS;
            x_lo[sx] = (wx_lo[sx] != 0.0f) ? x_lo[sx] : xmin;
            x_hi[sx] = (wx_hi[sx] != 0.0f) ? x_hi[sx] : xmin;
        }
    }
    __syncthreads();

    // ---- pooling: bins outer (the bin's tables in registers), channels inner; roi_align_levels_kernel's arithmetic -----
    // The gathers are what this phase costs (the address path takes a wave instruction's 64 addresses at a fixed rate,
    // whatever their width: 16 four-byte gathers per bin and channel 205 us per frame pair, 8 eight-byte ones 192), so the
    // cells are fetched in the widest pieces that hold them; the values, products and sums are the generic kernel's.
    const float* __restrict__ f = P.feat[lvl];
    // one buffer resource per PLANE (its size is the bound the hardware checks the lane offsets against — a scalar offset
    // is not part of that check): the dwords of a 16-byte piece that reach past the plane's last cell read as zeros
    // instead of the next plane or, behind the last channel, unmapped memory
    auto plane_rsrc = [&](int c) __attribute__((always_inline)) {
        const unsigned long long fa = reinterpret_cast<unsigned long long>(f + (size_t)c * H * W);
        // (readfirstlane returns a signed int: through `unsigned`, or a low word with bit 31 set sign-extends into the
        // high word and the resource points into unmapped space)
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)fa);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(fa >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 (int)((unsigned)(H * W) * 4u), 0x00020000);
    };
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    for (int t0 = 0; t0 < RX * RX; t0 += 256) {
        const int t = t0 + threadIdx.x;
        const bool live = t < RX * RX;
        const int ph = live ? t / RX : 0;
        const int pw = live ? t - ph * RX : 0;
        int ylo[G], yhi[G], xlo[G], xhi[G];
        float wyl[G], wyh[G], wxl[G], wxh[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            ylo[i] = y_lo[ph * G + i];
            yhi[i] = y_hi[ph * G + i];
            wyl[i] = wy_lo[ph * G + i];
            wyh[i] = wy_hi[ph * G + i];
            xlo[i] = x_lo[pw * G + i];
            xhi[i] = x_hi[pw * G + i];
            wxl[i] = wx_lo[pw * G + i];
            wxh[i] = wx_hi[pw * G + i];
        }
        // The four cells a sample row needs — (low, low + 1) of the bin's first and of its second sample — lie within four
        // consecutive cells whenever the second sample's low cell is at most two cells to the right of the first's (bins
        // narrower than ~5 cells: the rule, the level mapper sizes the map to the template) and both pairs are (low, low + 1):
        // then ONE 16-byte gather per row brings all four.  Lanes for which that does not hold (map border, weightless
        // entries, very wide bins) take their values from 4-byte gathers of exactly their cells; whether any lane of the
        // wave needs those is decided once per bin (wave-uniform: no divergent memory code).
        static_assert(G == 2, "two samples per bin and axis");
        const int d1 = xlo[1] - xlo[0];
        const bool quad = (xhi[0] == xlo[0] + 1) && (xhi[1] == xlo[1] + 1) && d1 >= 0 && d1 <= 2;
        const bool any_odd = __any(!quad);
        typedef int v4i_t __attribute__((ext_vector_type(4)));
        for (int cl = 0; cl < nch; ++cl) {
            const auto rsrc = plane_rsrc(c0 + cl);
            const int soff = 0;
            float vl[G][G][2], vh[G][G][2];        // [iy][ix][low / high column] of the low / high row
#pragma unroll
            for (int iy = 0; iy < G; ++iy) {
                const v4i_t a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(ylo[iy] + xlo[0]) * 4u, soff, 0);
                const v4i_t b4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(yhi[iy] + xlo[0]) * 4u, soff, 0);
                vl[iy][0][0] = __int_as_float(a[0]);
                vl[iy][0][1] = __int_as_float(a[1]);
                vh[iy][0][0] = __int_as_float(b4[0]);
                vh[iy][0][1] = __int_as_float(b4[1]);
                vl[iy][1][0] = __int_as_float(d1 == 0 ? a[0] : (d1 == 1 ? a[1] : a[2]));
                vl[iy][1][1] = __int_as_float(d1 == 0 ? a[1] : (d1 == 1 ? a[2] : a[3]));
                vh[iy][1][0] = __int_as_float(d1 == 0 ? b4[0] : (d1 == 1 ? b4[1] : b4[2]));
                vh[iy][1][1] = __int_as_float(d1 == 0 ? b4[1] : (d1 == 1 ? b4[2] : b4[3]));
            }
            if (any_odd) {
#pragma unroll
                for (int iy = 0; iy < G; ++iy)
#pragma unroll
                    for (int ix = 0; ix < G; ++ix) {
                        const float o1 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(ylo[iy] + xlo[ix]) * 4u, soff, 0));
                        const float o2 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(ylo[iy] + xhi[ix]) * 4u, soff, 0));
                        const float o3 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(yhi[iy] + xlo[ix]) * 4u, soff, 0));
                        const float o4 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(yhi[iy] + xhi[ix]) * 4u, soff, 0));
                        if (!quad) {
                            vl[iy][ix][0] = o1;
                            vl[iy][ix][1] = o2;
                            vh[iy][ix][0] = o3;
                            vh[iy][ix][1] = o4;
                        }
                    }
            }
            float acc = 0.0f;
#pragma unroll
            for (int iy = 0; iy < G; ++iy) {
#pragma unroll
                for (int ix = 0; ix < G; ++ix) {
                    const float v1 = vl[iy][ix][0], v2 = vl[iy][ix][1], v3 = vh[iy][ix][0], v4 = vh[iy][ix][1];
                    const float w1 = wyl[iy] * wxl[ix], w2 = wyl[iy] * wxh[ix];
                    const float w3 = wyh[iy] * wxl[ix], w4 = wyh[iy] * wxh[ix];
                    acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                }
            }
            if (live) xs[cl][ph * XS + pw] = acc / (float)(G * G);
        }
    }
    __syncthreads();

    // ---- correlation: xcorr_dw_rowpatch_kernel's arithmetic, plane after plane ------------------------------------------
    const int i = threadIdx.x / NQ, jq = threadIdx.x - i * NQ;
    const bool row_live = i < HO;
    for (int cl = 0; cl < nch; ++cl) {
        float pm = 0.0f;
        if (row_live) {
            float tap[RZ * RZ];
#pragma unroll
            for (int t = 0; t < RZ * RZ; ++t) tap[t] = zs[cl][t];
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < RZ; ++u) {
                const float* row = xs[cl] + (i + u) * XS + 4 * jq;
                float seg[((SEG + 3) / 4) * 4];
#pragma unroll
                for (int q = 0; q < (SEG + 3) / 4; ++q) {
                    const float4 v4 = *reinterpret_cast<const float4*>(row + 4 * q);
                    seg[4 * q + 0] = v4.x;
                    seg[4 * q + 1] = v4.y;
                    seg[4 * q + 2] = v4.z;
                    seg[4 * q + 3] = v4.w;
                }
#pragma unroll
                for (int v = 0; v < RZ; ++v)
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[o] = fmaf(seg[o + v], tap[u * RZ + v], acc[o]);
            }
            float* dst = resp + ((size_t)r * C + c0 + cl) * (HO * HO) + i * HO + 4 * jq;
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (4 * jq + o < HO) {
                    dst[o] = acc[o];
                    pm = plane_max_step(pm, acc[o]);
                }
        }
        if (plane_max != nullptr) {            // (all 64 lanes of every wave arrive here: the wave reduction needs them)
            pm = plane_max_wave(pm);
            if ((threadIdx.x & 63) == 0) atomicMax(&pmax[cl], __float_as_uint(pm));
        }
    }
    if (plane_max != nullptr) {
        __syncthreads();
        if ((int)threadIdx.x < nch) plane_max[(size_t)r * C + c0 + threadIdx.x] = __uint_as_float(pmax[threadIdx.x]);
    }
}

// (35, 7, sampling ratio 2) only; returns SMOT_ERR_UNSUPPORTED for anything else (the caller runs the two-kernel form).
int launch_sr_xcorr_gather(const LevelParams& P, int C, const float* sr, const float* boxes, const float* templates, int N,
                           int rx, int rz, int sampling_ratio, float* resp, hipStream_t st, float* plane_max) {
    if (!(rx == 35 && rz == 7 && sampling_ratio == 2)) return SMOT_ERR_UNSUPPORTED;
    if (N == 0) return SMOT_OK;
    dim3 grid(N, (C + SX_CH - 1) / SX_CH);
    timer_mark(0, 0, st);
    SMOT_LAUNCH((sr_xcorr_gather_kernel<35, 7, 2>), grid, dim3(256), 0, st, P, C, sr, boxes, templates, resp, plane_max);
    timer_mark(0, 1, st);
    return check_launch("sr_xcorr_gather");
}

int sr_xcorr_gather_impl(const float* const* feats, const int* heights, const int* widths, const int* pad_cells,
                         const float* scales, int num_levels, int C, const float* boxes, const float* sr,
                         const float* templates, int N, int rx, int rz, int sampling_ratio, float* resp, hipStream_t st,
                         float* plane_max) {
    LevelParams P;
    const int rc = fill_level_params(&P, feats, heights, widths, pad_cells, scales, num_levels, "sr_xcorr_gather");
    if (rc) return rc;
    return launch_sr_xcorr_gather(P, C, sr, boxes, templates, N, rx, rz, sampling_ratio, resp, st, plane_max);
}

}  // namespace smot

// The C entry of the 35 / 7 shape (smot_sr_xcorr_fused_fwd covers 30 / 15): arguments as there, no pooled-plane output.
extern "C" int smot_sr_xcorr_gather_fwd(const float* const* feats, const int* heights, const int* widths,
                                        const int* pad_cells, const float* scales, int num_levels, int C,
                                        const float* boxes, const float* sr, const float* templates, int N, int rx, int rz,
                                        int sampling_ratio, float* resp, smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(N >= 0 && C > 0, "sr_xcorr_gather: bad sizes N=%d C=%d", N, C);
    if (!(rx == 35 && rz == 7 && sampling_ratio == 2)) {
        set_error("sr_xcorr_gather: only Rx=35, Rz=7, sampling_ratio=2 (got %d, %d, %d); use smot_sr_xcorr_fused_fwd (30 / 15) or "
                  "smot_roi_align_levels_fwd + smot_xcorr_dw_fwd", rx, rz, sampling_ratio);
        return SMOT_ERR_UNSUPPORTED;
    }
    if (N == 0) return SMOT_OK;
    SMOT_REQUIRE(boxes && sr && templates && resp, "sr_xcorr_gather: null pointer");
    return sr_xcorr_gather_impl(feats, heights, widths, pad_cells, scales, num_levels, C, boxes, sr, templates, N, rx, rz,
                                sampling_ratio, resp, (hipStream_t)stream, nullptr);
}
