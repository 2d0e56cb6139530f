// K3w — EMM prediction towers as a Winograd F(2x2, 3x3) convolution on the fp32 matrix cores.
//
// Replaces the tower half of EMMPredictor.forward (reference EMM/feature_extractor.py:62-66: conv3x3 128->128
// without bias -> GroupNorm(32) -> ReLU for cls_tower and reg_tower; make_conv3x3 / group_norm [UPSTREAM
// maskrcnn_benchmark modeling/make_layers.py]) for the 16x16 response map, and feeds the same fused partial
// heads as predictor.hip's direct kernel.
//
// Why Winograd here: the direct implicit GEMM is MFMA-bound (4.53 GFLOP @ 30 tracks on a 157 TFLOP/s fp32
// pipe, two workgroups per CU => >= 32.6 us); F(2x2,3x3) needs 16 multiplies per 2x2 output tile and input
// channel instead of 36, i.e. 2.25x fewer MFMAs, and everything stays fp32 (inputs, products, accumulation):
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A,   16 "xi" positions (i,j) of the 4x4 transformed tile,
//     M_xi[oc][tile] = sum_ic U_xi[oc][ic] * V_xi[ic][tile]   -> 16 GEMMs of 16 x 64 x C per workgroup.
// Transform constants are 0, +-1, +-1/2: the result differs from the direct fp32 sum only in rounding order.
//
//   workgroup = (track, 16 output channels) x all 64 2x2 tiles; 4 waves; XCD-aware block order: the 16
//               workgroups of a track share an XCD (one L2 fetch of the track's response map).
//   wave w    = xi row i = w (4 xi) x 4 N-tiles of 16 tiles  -> 16 accumulator tiles (64 AGPRs).
//   A operand = U, transformed ONCE per parameter set by tower_pack_kernel into the exact per-lane order the
//               waves consume (one coalesced 16-byte load per lane and 4-channel k-step, straight to VGPRs).
//   B operand = V, built in registers: lane (ic = lane/16, tile = 16t + lane%16) reads two rows of its 4x4
//               input patch from LDS (only the two rows the wave's xi-row needs: 8 ds_read_b64 per k-step),
//               5 adds per operand.  The transformed input never exists in memory.
//   LDS       = raw response planes only (16 channels per stage, zero-haloed 18 x 24 rows, plane stride 480
//               = 32 mod 64 banks: conflict-free 64-bit patch reads), double-buffered, one barrier per 16
//               input channels; the stage after next is in flight in registers.
//   epilogue  = output transform (column half in registers, row half across the waves through LDS), two-pass
//               GroupNorm + affine + ReLU, fused partial heads exactly as in predictor.hip.
#include "tower_common.h"

namespace smot {

constexpr int W_ROW = 24;                 // floats per LDS row (18 used)
constexpr int W_PLANE = 480;              // 18 rows * 24 = 432, padded: plane stride = 32 (mod 64) banks
constexpr int W_STAGE_IC = 16;            // input channels per LDS stage
constexpr int W_BUF = W_STAGE_IC * W_PLANE;      // 7680 floats
constexpr int W_XOC = 68;                 // oc stride of the exchange image (4*68 = 16 mod 64 banks)
constexpr int W_X_FLOATS = 4 * 2 * 16 * W_XOC;   // 8704
constexpr int W_PL_OFF = W_X_FLOATS;             // head planes [16][336]
constexpr int W_HW_OFF = W_PL_OFF + 16 * T_PLANE;   // head taps [16*9][4]
constexpr int W_ST_OFF = W_HW_OFF + 16 * 36;        // channel sums [16], [16]
static_assert(W_ST_OFF + 32 <= 2 * W_BUF, "epilogue overlays the stage buffers");

// packed[tile][k = ic/4][wave][lane][q] = (G g G^T)[i = wave][j = q] of g = W[oc = 16*tile + lane%16][ic = 4k + lane/16]
// (oc counts cls_tower channels first, then reg_tower).  One thread per (oc, ic).
__global__ void __launch_bounds__(256)
tower_pack_kernel(const float* __restrict__ wc, const float* __restrict__ wr, int C, float* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2 * C * C) return;
    const int oc = idx / C, ic = idx - oc * C;
    const float* g = (oc < C ? wc + (size_t)oc * C * 9 : wr + (size_t)(oc - C) * C * 9) + (size_t)ic * 9;
    float gg[4][3];                       // G g
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const float g0 = g[v], g1 = g[3 + v], g2 = g[6 + v];
        gg[0][v] = g0;
        gg[1][v] = 0.5f * ((g0 + g1) + g2);
        gg[2][v] = 0.5f * ((g0 - g1) + g2);
        gg[3][v] = g2;
    }
    const int tile = oc >> 4, k = ic >> 2, lane = (oc & 15) + 16 * (ic & 3);
    const int nk = C >> 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 u;
        u.x = gg[i][0];
        u.y = 0.5f * ((gg[i][0] + gg[i][1]) + gg[i][2]);
        u.z = 0.5f * ((gg[i][0] - gg[i][1]) + gg[i][2]);
        u.w = gg[i][2];
        *reinterpret_cast<float4*>(packed + ((((size_t)tile * nk + k) * 4 + i) * 64 + lane) * 4) = u;
    }
}

__global__ void __launch_bounds__(256)
tower_wino_kernel(const float* __restrict__ resp, const float* __restrict__ packed, TowerParams P, int N, int C,
                  int cpg, float eps, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_per_tower = C >> 4;
    const int tiles = 2 * tiles_per_tower;
    // consecutive workgroup ids go round-robin over the 8 XCDs: give all tiles of a track the same XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int n = (slot / tiles) * 8 + xcd;
    const int tile = slot % tiles;
    if (n >= N) return;
    const int tower = tile / tiles_per_tower;
    const int oc0 = (tile - tower * tiles_per_tower) * 16;
    const float* __restrict__ in = resp + (size_t)n * C * 256;
    const int nk = C >> 2;
    const int nstages = C / W_STAGE_IC;

    {   // zero both stage buffers once: the halos stay zero for the whole main loop
        float4* z = reinterpret_cast<float4*>(sm);
        for (int e = tid; e < 2 * W_BUF / 4; e += 256) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // head taps of this tile's channels, fetched now, used in the epilogue: hw[(ocl*9+tap)*4 + o]
    float hwreg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int idx = tid + 256 * j;
        float v = 0.0f;
        if (idx < 16 * 36) {
            const int o = idx / (16 * 9);
            const int rem = idx - o * (16 * 9);
            const size_t src = (size_t)oc0 * 9 + rem;
            if (tower == 1) {
                v = P.reg_w[(size_t)o * C * 9 + src];
            } else if (o < 2) {
                v = P.cls_w[(size_t)o * C * 9 + src];
            } else if (o == 2) {
                v = P.center_w[src];
            }
        }
        hwreg[j] = v;
    }

    float4 pr[4];
    auto load_raw = [&](int g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
            pr[j] = *reinterpret_cast<const float4*>(in + (size_t)(W_STAGE_IC * g + (f >> 6)) * 256 + (f & 63) * 4);
        }
    };
    auto store_raw = [&](float* buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
            const int f4 = f & 63;
            float* d = buf + (f >> 6) * W_PLANE + ((f4 >> 2) + 1) * W_ROW + 1 + (f4 & 3) * 4;
            d[0] = pr[j].x;
            d[1] = pr[j].y;
            d[2] = pr[j].z;
            d[3] = pr[j].w;
        }
    };
    const float* __restrict__ ua = packed + (((size_t)tile * nk) * 4 + wave) * 256 + lane * 4;   // + k*1024
    f32x4 ac[4], an[4];
    auto load_a = [&](int g, f32x4* dst) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f32x4*>(ua + (size_t)(4 * g + q) * 1024);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_raw(0);
    load_a(0, ac);
    __syncthreads();                       // zero fill complete before interior writes
    store_raw(sm);
    if (nstages > 1) load_raw(1);
    __syncthreads();

    // rows of the 4x4 patch this wave's xi-row combines:  i=0: d0-d2, 1: d1+d2, 2: d2-d1, 3: d1-d3
    const int row_a = (wave == 0) ? 0 : ((wave == 2) ? 2 : 1);
    const int row_b = (wave == 2) ? 1 : ((wave == 3) ? 3 : 2);
    const float sgn = (wave == 1) ? 1.0f : -1.0f;
    const int kq = lane >> 4, xl = lane & 15;
    // tile (ty, tx) = (2t + xl/8, xl%8): patch origin (haloed coordinates) row 2*ty, column 2*tx
    const int patch0 = kq * W_PLANE + (2 * (xl >> 3)) * W_ROW + 2 * (xl & 7);
    const int off_a = patch0 + row_a * W_ROW, off_b = patch0 + row_b * W_ROW;

    for (int g = 0; g < nstages; ++g) {
        const float* buf = sm + (g & 1) * W_BUF;
        if (g + 1 < nstages) {
            store_raw(sm + ((g + 1) & 1) * W_BUF);      // last read in stage g-1, barrier passed
            load_a(g + 1, an);
        }
        if (g + 2 < nstages) load_raw(g + 2);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float* R = buf + q4 * 4 * W_PLANE;
            float bop[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float2 a01 = *reinterpret_cast<const float2*>(R + off_a + t * 4 * W_ROW);
                const float2 a23 = *reinterpret_cast<const float2*>(R + off_a + t * 4 * W_ROW + 2);
                const float2 b01 = *reinterpret_cast<const float2*>(R + off_b + t * 4 * W_ROW);
                const float2 b23 = *reinterpret_cast<const float2*>(R + off_b + t * 4 * W_ROW + 2);
                const float w0 = fmaf(sgn, b01.x, a01.x), w1 = fmaf(sgn, b01.y, a01.y);
                const float w2 = fmaf(sgn, b23.x, a23.x), w3 = fmaf(sgn, b23.y, a23.y);
                bop[0][t] = w0 - w2;
                bop[1][t] = w1 + w2;
                bop[2][t] = w2 - w1;
                bop[3][t] = w1 - w3;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q4][0], bop[0][t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q4][1], bop[1][t], acc[1][t], 0, 0, 0);
                acc[2][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q4][2], bop[2][t], acc[2][t], 0, 0, 0);
                acc[3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q4][3], bop[3][t], acc[3][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ac[q] = an[q];
        __syncthreads();
    }

    // ---- output transform --------------------------------------------------------------------------
    // acc[q][t][r] = M_(i=wave, j=q)[oc = 4*kq + r][tile = 16t + xl].  Column half (j) in registers,
    // row half (i) across the four waves through the exchange image X[i][b][oc][tile].
    float* X = sm;
    float* planes = sm + W_PL_OFF;
    float* hw = sm + W_HW_OFF;
    float* chs = sm + W_ST_OFF;            // [16] channel sums, [16] channel squared deviations
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p0 = (acc[0][t][r] + acc[1][t][r]) + acc[2][t][r];
            const float p1 = (acc[1][t][r] - acc[2][t][r]) - acc[3][t][r];
            float* dst = X + ((wave * 2) * 16 + (kq * 4 + r)) * W_XOC + 16 * t + xl;
            dst[0] = p0;
            dst[16 * W_XOC] = p1;
        }
    for (int e = tid; e < 16 * T_PLANE; e += 256) planes[e] = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int idx = tid + 256 * j;
        if (idx < 16 * 36) {
            const int o = idx / (16 * 9);
            const int rem = idx - o * (16 * 9);
            hw[rem * 4 + o] = hwreg[j];
        }
    }
    __syncthreads();

    // thread = (output channel ocl = tid/16, tiles 16t + x16): 4 tiles x 2x2 outputs
    const int ocl = tid >> 4, x16 = tid & 15;
    float y[4][2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float* src = X + (b * 16 + ocl) * W_XOC + 16 * t + x16;
            const float x0 = src[0 * 32 * W_XOC], x1 = src[1 * 32 * W_XOC];
            const float x2 = src[2 * 32 * W_XOC], x3 = src[3 * 32 * W_XOC];
            y[t][0][b] = (x0 + x1) + x2;
            y[t][1][b] = (x1 - x2) - x3;
        }

    // ---- GroupNorm (two-pass, fp32) + affine + ReLU --------------------------------------------------
    const float inv_cnt = 1.0f / (float)(cpg * 256);
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += (y[t][0][0] + y[t][0][1]) + (y[t][1][0] + y[t][1][1]);
    s = group16_sum(s);
    if (x16 == 0) chs[ocl] = s;
    __syncthreads();
    const int g0 = (ocl / cpg) * cpg;
    float mean = 0.0f;
    for (int ch = g0; ch < g0 + cpg; ++ch) mean += chs[ch];
    mean *= inv_cnt;
    float sq = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float d = y[t][a][b] - mean;
                sq += d * d;
            }
    sq = group16_sum(sq);
    if (x16 == 0) chs[16 + ocl] = sq;
    __syncthreads();
    float var = 0.0f;
    for (int ch = g0; ch < g0 + cpg; ++ch) var += chs[16 + ch];
    const float rstd = 1.0f / sqrtf(var * inv_cnt + eps);
    const float ga = P.gamma[tower][oc0 + ocl], be = P.beta[tower][oc0 + ocl];
    {
        float* pl = planes + ocl * T_PLANE;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ty = 2 * t + (x16 >> 3), tx = x16 & 7;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float v = (y[t][a][b] - mean) * rstd * ga + be;
                    pl[(2 * ty + a + 1) * 18 + 2 * tx + b + 1] = fmaxf(v, 0.0f);
                }
        }
    }
    __syncthreads();

    // ---- fused partial heads: this tile's 16 channels x 9 taps -> 4 head outputs per position ---------
    {
        const int py = tid >> 4, px = tid & 15;
        float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;
        const float* pl0 = planes + py * 18 + px;
#pragma unroll 4
        for (int cl = 0; cl < 16; ++cl) {
            const float* pl = pl0 + cl * T_PLANE;
            const float4* wrow = reinterpret_cast<const float4*>(hw + cl * 36);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float a = pl[(tap / 3) * 18 + (tap % 3)];
                const float4 w = wrow[tap];
                h0 = fmaf(a, w.x, h0);
                h1 = fmaf(a, w.y, h1);
                h2 = fmaf(a, w.z, h2);
                h3 = fmaf(a, w.w, h3);
            }
        }
        float* __restrict__ dst = part + ((size_t)n * tiles + tile) * 4 * 256 + tid;
        dst[0 * 256] = h0;
        dst[1 * 256] = h1;
        dst[2 * 256] = h2;
        dst[3 * 256] = h3;
    }
}

int launch_tower_wino(const float* resp, const float* packed, const TowerParams& P, int N, int C, int cpg, float eps,
                      float* part, hipStream_t st) {
    const size_t smem = (size_t)2 * W_BUF * sizeof(float);     // 61,440 B: two workgroups per CU
    const int tiles = 2 * (C / 16);
    const int grid = ((N + 7) / 8) * 8 * tiles;
    hipLaunchKernelGGL(tower_wino_kernel, dim3(grid), dim3(256), smem, st, resp, packed, P, N, C, cpg, eps, part);
    return check_launch("predictor towers (winograd)");
}

}  // namespace smot

extern "C" long long smot_emm_tower_pack_floats(int C) {
    if (C <= 0 || C % 16 != 0) return 0;          // the packed path needs 16-channel tiles
    return (long long)2 * C * C * 16;
}

extern "C" int smot_emm_tower_pack(const float* cls_tower_w, const float* reg_tower_w, int C, float* packed,
                                   smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(C > 0 && C % 16 == 0, "tower_pack: C=%d must be a multiple of 16", C);
    SMOT_REQUIRE(cls_tower_w && reg_tower_w && packed, "tower_pack: null pointer");
    SMOT_REQUIRE(((uintptr_t)packed & 15) == 0, "tower_pack: output must be 16-byte aligned");
    hipLaunchKernelGGL(tower_pack_kernel, dim3((2 * C * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls_tower_w,
                       reg_tower_w, C, packed);
    return check_launch("tower_pack");
}
