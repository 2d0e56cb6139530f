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
//   B operand = V, built in registers: a lane owns one input channel of the k-step and two horizontally
//               adjacent tiles per pair of N-tiles; it reads only the two patch rows the wave's xi-row
//               combines (one ds_read_b128 + one ds_read_b64 per row and tile pair) and spends 14 adds per
//               8 operands.  The transformed input never exists in memory.
//   LDS       = raw response planes only (8 channels per stage, zero-haloed 18 x 24 rows, plane stride 448),
//               ring of four stages, one barrier per stage; two further stages are in flight in registers.
//   epilogue  = output transform (column half in registers, row half across the waves through LDS), two-pass
//               GroupNorm + affine + ReLU, fused partial heads exactly as in predictor.hip.
#include "tower_common.h"
#include "knobs.h"

namespace smot {

constexpr int W_ROW = 24;                 // floats per LDS row (18 used)
constexpr int W_PLANE = 448;              // 18 rows * 24 = 432, padded: plane stride = 0 (mod 64) banks, see W_READ
constexpr int W_STAGE_IC = 8;             // input channels per LDS stage (2 k-steps)
constexpr int W_RING = 4;                 // stage buffers
constexpr int W_BUF = W_STAGE_IC * W_PLANE;      // 3840 floats
constexpr int W_XOC = 68;                 // oc stride of the exchange image (4*68 = 16 mod 64 banks)
constexpr int W_X_FLOATS = 4 * 2 * 16 * W_XOC;   // 8704
constexpr int W_PL_OFF = W_X_FLOATS;             // head planes [16][336]
constexpr int W_HW_OFF = W_PL_OFF + 16 * T_PLANE;   // head taps [16*9][4]
constexpr int W_ST_OFF = W_HW_OFF + 16 * 36;        // channel sums [16], [16]
constexpr int W_SMEM_FLOATS = (W_ST_OFF + 32 > W_RING * W_BUF) ? W_ST_OFF + 32 : W_RING * W_BUF;   // epilogue overlays the ring
static_assert(W_SMEM_FLOATS * 4 <= 64 * 1024, "two workgroups per CU, no opt-in needed");

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

// ABL (timing ablations for profiles/, wrong results): 1 = no raw staging inside the loop, 2 = no A-operand
// loads inside the loop, 3 = no LDS reads / operand transform, 4 = no MFMAs, 5 = MFMAs + barriers only,
// 6 = MFMAs only.  Instantiated in the measurement library only (knobs.h: SMOT_WINO_ABL).
template <int ABL>
__global__ void __launch_bounds__(256, 2)
tower_wino_kernel(const float* __restrict__ resp, const float* __restrict__ packed, TowerParams P, int N, int C,
                  int cpg, float eps, float* __restrict__ part, unsigned* __restrict__ zero_words,
                  long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
#define W_TRACE(SLOT) \
    if (trace && tid == 0) trace[(size_t)blockIdx.x * 8 + (SLOT)] = (long long)__builtin_amdgcn_s_memtime();
    W_TRACE(0)
    if (trace && tid == 0) {      // where this workgroup runs: HW_ID (cu/sh/se) and XCC_ID
        trace[(size_t)blockIdx.x * 8 + 6] = (long long)__builtin_amdgcn_s_getreg(0xF804);
        trace[(size_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_getreg(0xF814);
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_per_tower = C >> 4;
    const int tiles = 2 * tiles_per_tower;
    // consecutive workgroup ids go round-robin over the 8 XCDs: give all tiles of a track the same XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int n = (slot / tiles) * 8 + xcd;
    const int tile = slot % tiles;
    if (n >= N) return;
    if (zero_words != nullptr && tile == 0 && tid == 0) zero_words[n] = 0u;     // visible at the kernel boundary
    // (Workgroups b and b + 256 share a CU — HW_ID trace in tools/debug/tower_bench.py.  Delaying the second
    // dispatch round so that one workgroup's epilogue overlaps the other's main loop was measured: every 4 k
    // cycles of stagger cost 1 us — the CU is throughput-bound in every phase, not latency-bound.)
    const int tower = tile / tiles_per_tower;
    const int oc0 = (tile - tower * tiles_per_tower) * 16;
    const float* __restrict__ in = resp + (size_t)n * C * 256;
    const int nk = C >> 2;
    const int nstages = C / W_STAGE_IC;

    {   // zero the stage buffers once: the halos stay zero for the whole main loop (a halo-only fill was measured
        // slower: its scattered ds_write_b32 and index arithmetic cost more than 14 ds_write_b128 per thread)
        float4* z = reinterpret_cast<float4*>(sm);
        for (int e = tid; e < W_RING * W_BUF / 4; e += 256) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
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

    // ---- staging: ring of four 8-channel stages of raw response planes --------------------------------
    float4 prs[2][2];                         // two stages in flight in registers
    auto load_raw = [&](int st, float4* pr) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = tid + 256 * j;
            pr[j] = *reinterpret_cast<const float4*>(in + (size_t)(W_STAGE_IC * st + (f >> 6)) * 256 + (f & 63) * 4);
        }
    };
    auto store_raw = [&](float* buf, const float4* pr) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
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
    f32x4 aset[2][2];                         // A operands: the stage in use and the next one
    auto load_a = [&](int st, f32x4* dst) {
        dst[0] = *reinterpret_cast<const f32x4*>(ua + (size_t)(2 * st) * 1024);
        dst[1] = *reinterpret_cast<const f32x4*>(ua + (size_t)(2 * st + 1) * 1024);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_raw(0, prs[0]);
    load_raw(1, prs[1]);
    load_a(0, aset[0]);
    if (ABL == 2 || ABL >= 5) {
        aset[1][0] = aset[0][0]; aset[1][1] = aset[0][1];
    }
    __syncthreads();                       // zero fill complete before interior writes
    store_raw(sm, prs[0]);
    store_raw(sm + W_BUF, prs[1]);
    load_raw(min(2, nstages - 1), prs[0]);
    load_raw(min(3, nstages - 1), prs[1]);
    __syncthreads();

    // rows of the 4x4 patch this wave's xi-row combines:  i=0: d0-d2, 1: d1+d2, 2: d2-d1, 3: d1-d3
    const int row_a = (wave == 0) ? 0 : ((wave == 2) ? 2 : 1);
    const int row_b = (wave == 2) ? 1 : ((wave == 3) ? 3 : 2);
    const int kq = lane >> 4, xl = lane & 15;
    // Lane (kq, xl) supplies input channel kq of the k-step and, for N-tile t, the 2x2-output tile
    //     ty = xl/4 + 4*(t/2),  tx = 2*(xl%4) + (t%2):
    // t = 2p and 2p+1 are horizontal neighbours, so one 6-float row segment (one ds_read_b128 + one
    // ds_read_b64, both naturally aligned: haloed column 4*(xl%4)) serves both.  Banks: row stride 24 and
    // plane stride 448 put the four (plane, tile-row) classes of every 16-lane b128 group on disjoint
    // 16-bank windows; the b64 halves of planes 0/1 collide 2-way (4 instead of 2 LDS cycles).
    const int patch0 = kq * W_PLANE + (2 * (xl >> 2)) * W_ROW + 4 * (xl & 3);
    const int ia = patch0 + row_a * W_ROW, ib = patch0 + row_b * W_ROW;

    // Main loop.  Stage = 8 input channels = 2 k-steps; ring of four stage buffers; one barrier per stage.
    // At the start of stage s: store the raw planes of stage s+2 (fetched two stages ago), fetch stage s+4 and
    // the A operands of stage s+1.  A k-step is: issue the LDS reads of the NEXT k-step (they may already touch
    // stage s+1, visible since this stage's barrier), the 16 MFMAs of this k-step (which cover the LDS latency),
    // then the operand transform of the next k-step.  MFMA and VALU work are serialised on purpose: fp32 MFMAs
    // and fp32 VALU ops do not overlap on gfx950 (measured with the ablations: loop time = MFMA time + VALU time;
    // the fp32 matrix rate equals the packed-fp32 vector rate), and one operand/read set instead of two keeps
    // the kernel inside the 256-register budget of two workgroups per CU.
    // Everything is branch-free (prefetch indices are clamped; the tail re-fetches the last stage): with guards
    // the waitcnt pass drains vmcnt at the top of every stage; the fences keep hipcc from sinking the prefetch
    // loads to their first use.
    float rd[2][6][2];
    float bop[4][4];
    if (ABL >= 3) {
#pragma unroll
        for (int e = 0; e < 24; ++e) {
            (&bop[0][0])[e & 15] = (float)(lane + e);
            (&rd[0][0][0])[e] = (float)(lane - e);
        }
    }
#define W_READ(BUFI, Q, RD)                                                                      \
    if (ABL != 3 && ABL < 5) {                                                                   \
        /* one laundered base per k-step so the pair offsets fold into the ds_read immediates */ \
        int oa = ia + (BUFI) * W_BUF + (Q) * 4 * W_PLANE;                                        \
        int ob = ib + (BUFI) * W_BUF + (Q) * 4 * W_PLANE;                                        \
        asm volatile("" : "+v"(oa), "+v"(ob));                                                   \
        __builtin_assume((oa & 3) == 0);                                                         \
        __builtin_assume((ob & 3) == 0);                                                         \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {          /* tile rows ty and ty + 4 */    \
            const float4 a4 = *reinterpret_cast<const float4*>(sm + oa + p * 8 * W_ROW);         \
            const float2 a2 = *reinterpret_cast<const float2*>(sm + oa + p * 8 * W_ROW + 4);     \
            const float4 b4 = *reinterpret_cast<const float4*>(sm + ob + p * 8 * W_ROW);         \
            const float2 b2 = *reinterpret_cast<const float2*>(sm + ob + p * 8 * W_ROW + 4);     \
            RD[p][0][0] = a4.x; RD[p][0][1] = b4.x;                                              \
            RD[p][1][0] = a4.y; RD[p][1][1] = b4.y;                                              \
            RD[p][2][0] = a4.z; RD[p][2][1] = b4.z;                                              \
            RD[p][3][0] = a4.w; RD[p][3][1] = b4.w;                                              \
            RD[p][4][0] = a2.x; RD[p][4][1] = b2.x;                                              \
            RD[p][5][0] = a2.y; RD[p][5][1] = b2.y;                                              \
        }                                                                                        \
    }
#define W_XFORM(RD, BOP)                                                                         \
    if (ABL != 3 && ABL < 5) _Pragma("unroll") for (int p = 0; p < 2; ++p) {                     \
        /* six columns of the row combination d_a +- d_b, shared by the tile pair */             \
        float w[6];                                                                              \
        _Pragma("unroll") for (int c = 0; c < 6; ++c)                                            \
            w[c] = PLUS ? RD[p][c][0] + RD[p][c][1] : RD[p][c][0] - RD[p][c][1];                 \
        BOP[0][2 * p] = w[0] - w[2];                                                             \
        BOP[1][2 * p] = w[1] + w[2];                                                             \
        BOP[2][2 * p] = w[2] - w[1];                                                             \
        BOP[3][2 * p] = w[1] - w[3];                                                             \
        BOP[0][2 * p + 1] = w[2] - w[4];                                                         \
        BOP[1][2 * p + 1] = w[3] + w[4];                                                         \
        BOP[2][2 * p + 1] = w[4] - w[3];                                                         \
        BOP[3][2 * p + 1] = w[3] - w[5];                                                         \
    }
#define W_MFMA(AV, BOP)                                                                          \
    if (ABL != 4) _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                \
        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[0], BOP[0][t], acc[0][t], 0, 0, 0);   \
        acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[1], BOP[1][t], acc[1][t], 0, 0, 0);   \
        acc[2][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[2], BOP[2][t], acc[2][t], 0, 0, 0);   \
        acc[3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[3], BOP[3][t], acc[3][t], 0, 0, 0);   \
    }
#define W_FENCE __builtin_amdgcn_sched_barrier(0);
#define W_STAGE(J)                                                                               \
    {                                                                                            \
        if (ABL != 6) __syncthreads();                                                           \
        if (ABL != 1 && ABL < 5) store_raw(sm + (((J) + 2) & 3) * W_BUF, prs[(J) & 1]);          \
        if (ABL != 1 && ABL < 5) load_raw(min(s0 + (J) + 4, nstages - 1), prs[(J) & 1]);         \
        if (ABL != 2 && ABL < 5) load_a(min(s0 + (J) + 1, nstages - 1), aset[((J) + 1) & 1]);    \
        W_FENCE                                                                                  \
        W_READ((J), 1, rd)                             /* k-step 2s+1 */                         \
        W_MFMA(aset[(J) & 1][0], bop)                                                            \
        W_FENCE                                                                                  \
        W_XFORM(rd, bop)                                                                         \
        W_FENCE                                                                                  \
        W_READ(((J) + 1) & 3, 0, rd)                   /* k-step 2(s+1): next stage's buffer */  \
        W_MFMA(aset[(J) & 1][1], bop)                                                            \
        W_FENCE                                                                                  \
        W_XFORM(rd, bop)                                                                         \
        W_FENCE                                                                                  \
    }
#define W_LOOP                                                                                   \
    W_READ(0, 0, rd)                                                                             \
    W_XFORM(rd, bop)                                                                             \
    W_FENCE                                                                                      \
    for (int s0 = 0; s0 < nstages; s0 += 4) {       /* nstages is a multiple of 4 (C % 32 == 0) */ \
        W_STAGE(0)                                                                               \
        W_STAGE(1)                                                                               \
        W_STAGE(2)                                                                               \
        W_STAGE(3)                                                                               \
    }
    W_TRACE(1)
    if (wave == 1) {             // the xi-row 1 combination adds its two patch rows, the others subtract
        constexpr bool PLUS = true;
        W_LOOP
    } else {
        constexpr bool PLUS = false;
        W_LOOP
    }
    __syncthreads();
#undef W_LOOP
#undef W_FENCE
#undef W_STAGE
#undef W_MFMA
#undef W_XFORM
#undef W_READ
    W_TRACE(2)

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
    for (int e = tid; e < 16 * 68; e += 256) {          // halo of the head planes (interiors are written below)
        const int pl = e / 68, c = e - pl * 68;
        const int row = c < 18 ? 0 : (c < 36 ? 17 : 1 + ((c - 36) >> 1));
        const int col = c < 18 ? c : (c < 36 ? c - 18 : (((c - 36) & 1) ? 17 : 0));
        planes[pl * T_PLANE + row * 18 + col] = 0.0f;
    }
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

    W_TRACE(3)
    // thread = (output channel ocl = tid/16, tiles 4*x16 + t): 4 tiles x 2x2 outputs, one ds_read_b128 per (i, b)
    const int ocl = tid >> 4, x16 = tid & 15;
    float y[4][2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float* src = X + (b * 16 + ocl) * W_XOC + 4 * x16;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(src + 0 * 32 * W_XOC);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(src + 1 * 32 * W_XOC);
        const f32x4 x2 = *reinterpret_cast<const f32x4*>(src + 2 * 32 * W_XOC);
        const f32x4 x3 = *reinterpret_cast<const f32x4*>(src + 3 * 32 * W_XOC);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            y[t][0][b] = (x0[t] + x1[t]) + x2[t];
            y[t][1][b] = (x1[t] - x2[t]) - x3[t];
        }
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
            // tile 4*x16 + t  ->  MFMA N-tile t' = x16/4, lane xl = 4*(x16%4) + t  ->  (ty, tx) as in the main loop
            const int ty = (x16 & 3) + 4 * (x16 >> 3), tx = 2 * t + ((x16 >> 2) & 1);
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

    W_TRACE(4)
    // ---- fused partial heads: this tile's 16 channels x 9 taps -> 4 head outputs per position ---------
    // v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products per instruction, block = 4 neighbouring
    // positions: A = the 4 head weights of one (channel, tap) (lane%4 = output), B = the lane's own activation,
    // D[output][position] accumulates in 4 VGPRs per lane — the layout the stores below need.  256 MACs per
    // 8-cycle instruction: twice the v_fmac rate with no padding waste, and half the instructions
    // (2 LDS reads + 1 MFMA instead of 2 LDS reads + 4 FMAs per tap).  One fmaf chain per output, as before.
    {
        const int py = tid >> 4, px = tid & 15;
        f32x4 hacc = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* pl0 = planes + py * 18 + px;
        const float* hwl = hw + (lane & 3);
#pragma unroll 4
        for (int cl = 0; cl < 16; ++cl) {
            const float* pl = pl0 + cl * T_PLANE;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                hacc = __builtin_amdgcn_mfma_f32_4x4x1f32(hwl[(cl * 9 + tap) * 4], pl[(tap / 3) * 18 + (tap % 3)], hacc, 0,
                                                          0, 0);
        }
        float* __restrict__ dst = part + ((size_t)n * tiles + tile) * 4 * 256 + tid;
        dst[0 * 256] = hacc[0];
        dst[1 * 256] = hacc[1];
        dst[2 * 256] = hacc[2];
        dst[3 * 256] = hacc[3];
    }
    W_TRACE(5)
#undef W_TRACE
}

int launch_tower_wino(const float* resp, const float* packed, const TowerParams& P, int N, int C, int cpg, float eps,
                      float* part, unsigned* zero_words, hipStream_t st) {
    const size_t smem = (size_t)W_SMEM_FLOATS * sizeof(float);     // 58,752 B: two workgroups per CU
    const int tiles = 2 * (C / 16);
    const int grid = ((N + 7) / 8) * 8 * tiles;
#define W_LAUNCH(A)                                                                                             \
    hipLaunchKernelGGL(tower_wino_kernel<A>, dim3(grid), dim3(256), smem, st, resp, packed, P, N, C, cpg, eps, part, \
                       zero_words, g_trace)
#ifdef SMOT_DEBUG
    switch (knobs().wino_abl) {          // timing ablations (wrong results): measurement library only
        case 1: W_LAUNCH(1); break;
        case 2: W_LAUNCH(2); break;
        case 3: W_LAUNCH(3); break;
        case 4: W_LAUNCH(4); break;
        case 5: W_LAUNCH(5); break;
        case 6: W_LAUNCH(6); break;
        default: W_LAUNCH(0); break;
    }
#else
    W_LAUNCH(0);
#endif
#undef W_LAUNCH
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
