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
//   workgroup = (track, OCT x 16 output channels) x all 64 2x2 tiles; 4*OCT waves; XCD-aware block order: the
//               workgroups of a track share an XCD (one L2 fetch of the track's response map).
//   OCT = 1   : wave w = xi row i = w (4 xi) x 4 N-tiles of 16 tiles -> 16 accumulator tiles.
//   OCT = 2   : wave w = xi row i = w%4 x the N-tile pair h = w/4 x TWO 16-channel tiles -> 16 accumulator tiles
//               again, but every B operand a lane builds now feeds two MFMAs: half the operand-transform VALU
//               work, LDS reads and raw staging per MFMA, one zero fill and one response fetch per 32 channels.
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
#include <utility>

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
constexpr int W_EPI_FLOATS = W_ST_OFF + 32;         // epilogue image of ONE 16-channel tile
constexpr int w_smem_floats(int oct) {              // the epilogue images (one per tile) overlay the ring
    return (oct * W_EPI_FLOATS > W_RING * W_BUF) ? oct * W_EPI_FLOATS : W_RING * W_BUF;
}
constexpr int W_A_FLOATS = 4 * 4 * 2 * 2 * 256;     // BF3: fp16 A parts of one 32-channel block [q][xi][o][part][1 KB]
constexpr int W_A_SLOT = 4 * 2 * 2 * 256;           // floats per slot (one xi column q): [xi][o][part][256]
constexpr int W_AB_ROW = 4 * 2 * 1024;              // bytes of one xi row of a (tile, block): [q][part][lane][16 B]
static_assert((W_RING * W_BUF + W_A_FLOATS) * 4 <= 160 * 1024, "BF3: ring + A image in one CU's LDS");
static_assert(w_smem_floats(1) * 4 <= 64 * 1024, "OCT = 1: two workgroups per CU, no opt-in needed");
static_assert(w_smem_floats(2) * 4 <= 160 * 1024, "OCT = 2: one workgroup per CU");

// packed[tile][k = ic/4][wave][lane][q] = (G g G^T)[i = wave][j = q] of g = W[oc = 16*tile + lane%16][ic = 4k + lane/16]
// (oc counts cls_tower channels first, then reg_tower).  One thread per (oc, ic).
// scale 2^ku of the fp16 weight image from the largest |w| (bits): |u| <= 2.25 max|w| (rows of G have absolute sums <=
// 1.5), so 2^ku * 2.25 * max|w| < 2^15; 1 for a zero, infinite or NaN maximum
__device__ __forceinline__ float tower_pack_scale(unsigned wmax_bits) {
    const int e = (int)((wmax_bits >> 23) & 0xffu);              // max|w| < 2^(e - 126)
    if (e == 0 || e == 255) return 1.0f;
    int k = 266 - e;                                             // |u| < 2^(e - 124): 2^ku = 2^(139 - e), biased exponent 266 - e
    k = k < 1 ? 1 : (k > 254 ? 254 : k);
    return __uint_as_float((unsigned)k << 23);
}

__global__ void __launch_bounds__(256)
tower_pack_kernel(const float* __restrict__ wc, const float* __restrict__ wr, int C, float* __restrict__ packed,
                  unsigned* __restrict__ hdr) {
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
        if ((C & 31) == 0) {
            // fp16 image (tower_wino_kernel<.., BF3>, round 6): every transformed weight, scaled by the power of two 2^ku the
            // header names (so that the largest |u| lies in [2^13, 2^15): the second part stays a normal fp16 number), as TWO
            // fp16 parts u 2^ku = u1 + u2 (u1 = round-to-nearest fp16, u2 = round-to-nearest fp16 of the exact residual:
            // |u 2^ku - u1 - u2| <= 2^-23 |u| 2^ku), in the A-operand order of v_mfma_f32_16x16x32_f16 (lane = oc%16 +
            // 16*(ic%4) holds the eight channels 32 kb + 4 s + ic%4, s = 0..7), with the 32-channel blocks of xi column q
            // ROTATED by q + 1 stages (a stage = 8 channels = one dword d = s/2 of the lane's operand): block kb' of column q
            // holds the dwords d <= q of channel block kb' and the dwords d > q of channel block kb' - 1 (zero outside
            // 0..C/32-1; the image is zero-filled before this kernel), so that the kernel can consume one column per stage:
            //   hf[tile][kb' = 0..C/32][i][q][part][lane][s]   (16 B per lane)
            unsigned short* hf = reinterpret_cast<unsigned short*>(packed + (size_t)2 * C * C * 16);
            const int kb = ic >> 5, sidx = (ic & 31) >> 2, nkb1 = (C >> 5) + 1;
            const float su = tower_pack_scale(hdr[0]);
            const float uq[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kbp = ((sidx >> 1) <= q) ? kb : kb + 1;
                float r = uq[q] * su;                 // exact (power of two, no underflow of anything that matters)
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const _Float16 h = (_Float16)r;   // round to nearest even
                    r -= (float)h;                    // exact
                    hf[((((((size_t)tile * nkb1 + kbp) * 4 + i) * 4 + q) * 2 + part) * 64 + lane) * 8 + sidx] =
                        __builtin_bit_cast(unsigned short, h);
                }
            }
            if (idx == 0 && i == 0) reinterpret_cast<float*>(hdr)[1] = 1.0f / su;      // what the kernel multiplies back
        }
    }
}

// largest |w| of both tower filters -> hdr[0] (bits of a non-negative float order as unsigned integers; NaN ignored)
__global__ void __launch_bounds__(256)
tower_wmax_kernel(const float* __restrict__ wc, const float* __restrict__ wr, int count, unsigned* __restrict__ hdr) {
    float m = 0.0f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < count; e += gridDim.x * 256) m = fmaxf(m, fmaxf(fabsf(wc[e]), fabsf(wr[e])));
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(hdr, __float_as_uint(m));
}

// |resp| maximum of every (track, channel) plane -> pm[n*C + c] (one wave per plane).  The split form of the tower kernel
// scales a track's response by a power of two chosen from these (fp16 operand parts); the pooling + correlation kernel
// writes them itself, this launch serves responses that come from elsewhere (stand-alone operator, other shapes).
__global__ void __launch_bounds__(256)
plane_absmax_kernel(const float* __restrict__ resp, int planes, int hw, float* __restrict__ pm) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const float* __restrict__ src = resp + (size_t)plane * hw;
    float m = 0.0f;
    for (int e = lane; e < hw; e += 64) m = plane_max_step(m, src[e]);
    m = plane_max_wave(m);
    if (lane == 0) pm[plane] = m;
}

int launch_plane_absmax(const float* resp, int planes, int hw, float* pm, hipStream_t st) {
    hipLaunchKernelGGL(plane_absmax_kernel, dim3((planes + 3) / 4), dim3(256), 0, st, resp, planes, hw, pm);
    return check_launch("predictor towers (plane maxima)");
}

// 36 head taps (four channels) of channel group G on two alternating accumulators: tap E of the group is tap G * 36 + E of
// the tile, its four weights = block (G * 36 + E) % 16 of weight register (G * 36 + E) / 16 (the instruction's A-broadcast
// field must be a constant: hence the index sequence)
template <int G, int... E>
__device__ __forceinline__ void heads_group(const float (&hw)[9], const float (&a)[36], f32x4& h0, f32x4& h1,
                                            std::integer_sequence<int, E...>) {
    (((E % 2 == 0 ? h0 : h1) = __builtin_amdgcn_mfma_f32_4x4x1f32(hw[(G * 36 + E) / 16], a[E], (E % 2 == 0 ? h0 : h1), 4,
                                                                  (G * 36 + E) % 16, 0)), ...);
}

// ABL (timing ablations for profiles/, wrong results): 1 = no raw staging inside the loop, 2 = no A-operand
// loads inside the loop, 3 = no LDS reads / operand transform, 4 = no MFMAs, 5 = MFMAs + barriers only,
// 6 = MFMAs only.  Instantiated in the measurement library only (knobs.h: SMOT_WINO_ABL).
// BHO > 0: BLOCKED mode for a response map of BHO x BHO (16 < BHO <= 32; the second yaml family's 29 x 29): the map is
// covered by four overlapping 16 x 16 output blocks with origins {0, BHO - 16}^2 and a workgroup convolves ONE block of
// one track — N counts tracks, the grid 4 N "block tracks".  The stage buffers then hold 18 x 18 windows of the map with
// their REAL halo (zero only outside the map: buffer loads past the resource's size return 0), the main loop is the
// same, and the epilogue stops after the output transform: the block's outputs (those no block before it owns) go to
// `part` = the convolution output [N][2C][BHO * BHO]; GroupNorm needs the whole map and runs in tower_gn_heads_kernel
// (tower_conv.hip).  17 % more MFMAs than a 15 x 15-tile Winograd of the whole map would need (overlap + the 32 x 32
// cover), 54 % of the direct convolution's.
// BF3: the 16 transform-domain GEMMs on the bf16 matrix pipe at fp32 accuracy — every fp32 operand is the sum of three
// bf16 parts (weights: split once by tower_pack_kernel; activations: split in registers with v_cvt_pk_bf16_f32 as they
// leave the operand transform), and a product keeps the six part products down to 2^-24 of its size
//     a b ~ a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),         dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-25 |a b|,
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (K = 32 input channels per instruction, 16.6 cycles, against 8 x 32
// cycles of v_mfma_f32_16x16x4_f32: 6 x 16.6 against 256 cycles per 32 channels and tile —
// profiles/r04_ubench_bf16x3_rate.jsonl).  The transformed operands of a K = 32 block (four stages) are collected as
// packed bf16 pairs in 96 registers, then one burst of 96 instructions consumes them against A parts fetched as needed.
// OCT = 2 only (also in BLOCKED mode: the same loop over 18 x 18 windows).
template <int ABL, int OCT, int BHO = 0, bool BF3 = false>
__global__ void __launch_bounds__(256 * OCT, OCT == 1 ? 2 : 1)
tower_wino_kernel(const float* __restrict__ resp, const float* __restrict__ packed, TowerParams P, int N, int C,
                  int cpg, float eps, float* __restrict__ part, unsigned* __restrict__ zero_words,
                  long long* __restrict__ trace, const float* __restrict__ plane_max) {
    constexpr int NT = 256 * OCT;             // threads
    constexpr int NP = 2 / OCT;               // tile-row halves (pairs of N-tiles) per wave
    constexpr bool BLOCKED = BHO > 0;
    constexpr int MAP = BLOCKED ? BHO * BHO : 256;                    // positions of a response plane
    constexpr int RAW4 = 512 / NT;            // float4 per thread and stage of raw response
    constexpr int RAWB = (W_STAGE_IC * 324 + NT - 1) / NT;            // BLOCKED: dwords per thread and stage (18 x 18 windows)
    static_assert(!BLOCKED || (BHO > 16 && BHO <= 32), "four 16 x 16 blocks cover maps of 17 .. 32");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
#define W_TRACE(SLOT) \
    if (ABL != 15 && trace && tid == 0) trace[(size_t)blockIdx.x * 8 + (SLOT)] = (long long)__builtin_amdgcn_s_memtime();
    W_TRACE(0)
    if (ABL != 15 && trace && tid == 0) {      // where this workgroup runs: HW_ID (cu/sh/se) and XCC_ID
        trace[(size_t)blockIdx.x * 8 + 6] = (long long)__builtin_amdgcn_s_getreg(0xF804);
        trace[(size_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_getreg(0xF814);
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave & 3;                  // xi row of this wave
    const int nh = wave >> 2;                 // OCT = 2: which pair of N-tiles (0 for OCT = 1)
    const int tiles_per_tower = C >> 4;
    const int tiles = 2 * tiles_per_tower;
    const int wg_per_track = tiles / OCT;
    // consecutive workgroup ids go round-robin over the 8 XCDs: give all workgroups of a track the same XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    // BLOCKED: the four blocks of a track are four consecutive "tracks" of the slot arithmetic, all on the track's XCD
    // (the map is fetched into one L2 and the convolution output lands where tower_gn_heads_kernel reads it)
    const int wg_per_n = BLOCKED ? 4 * wg_per_track : wg_per_track;
    const int n = (slot / wg_per_n) * 8 + xcd;
    const int within = slot % wg_per_n;
    const int blk = BLOCKED ? within / wg_per_track : 0;
    const int oy = BLOCKED ? (blk >> 1) * (BHO - 16) : 0, ox = BLOCKED ? (blk & 1) * (BHO - 16) : 0;   // block origin
    const int tile0 = (within % wg_per_track) * OCT;    // first 16-channel tile (OCT = 2: tile0, tile0 + 1: same tower)
    if (n >= N) return;
    if (zero_words != nullptr && tile0 == 0 && blk == 0 && tid == 0) zero_words[n] = 0u;     // visible at the kernel boundary
    // (Workgroups b and b + 256 share a CU — HW_ID trace in measure/debug/tower_bench.py.  Delaying the second
    // dispatch round so that one workgroup's epilogue overlaps the other's main loop was measured: every 4 k
    // cycles of stagger cost 1 us — the CU is throughput-bound in every phase, not latency-bound.)
    const float* __restrict__ in = resp + (size_t)n * C * MAP;
    const int nk = C >> 2;
    const int nstages = C / W_STAGE_IC;
    // epilogue roles: thread (sub, ltid) works on tile tile0 + sub exactly as a 256-thread workgroup would
    const int sub = tid >> 8, ltid = tid & 255;
    const int etile = tile0 + sub;
    const int tower = etile / tiles_per_tower;
    const int oc0 = (etile - tower * tiles_per_tower) * 16;

    // ---- staging: ring of four 8-channel stages of raw response planes --------------------------------
    // Global addresses are buffer resource + per-thread byte offset (set once) + wave-uniform SGPR offset: no
    // vector instruction is spent on addressing inside the loop (every VALU instruction occupies the fp32 matrix
    // pipe on gfx950 — profiles/r02_ubench_mfma_valu_overlap.jsonl).
    auto make_rsrc = [](const float* base) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 0x7fffffff, 0x00020000);
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // BLOCKED: the resource ends with the track's response, so that an offset past it (an element outside the map)
    // loads 0 — the zero padding of the convolution, with no instruction spent on it
    auto make_rsrc_sized = [](const float* base, unsigned bytes) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 (int)bytes, 0x00020000);
    };
    const auto rs_in = BLOCKED ? make_rsrc_sized(in, (unsigned)C * MAP * 4u) : make_rsrc(in);
    const auto rs_a = make_rsrc(packed);
    constexpr int NRAW = BLOCKED ? (RAWB + 3) / 4 : RAW4;             // float4 registers per thread and stage
    float4 prs[2][NRAW];                      // two stages in flight in registers
    unsigned raw_voff[BLOCKED ? RAWB : RAW4];
    int raw_lds[BLOCKED ? RAWB : RAW4];
    if constexpr (!BLOCKED) {
#pragma unroll
        for (int j = 0; j < RAW4; ++j) {
            const int f = tid + NT * j;
            const int f4 = f & 63;
            raw_voff[j] = (unsigned)((f >> 6) * 256 + f4 * 4) * 4u;
            raw_lds[j] = (f >> 6) * W_PLANE + ((f4 >> 2) + 1) * W_ROW + 1 + (f4 & 3) * 4;
        }
    } else {
        // element e = tid + NT j of a stage: channel e / 324, window row (e % 324) / 18, column e % 18; the window starts
        // one cell above / left of the block.  Elements past the stage (the last j of some threads) go to the unused
        // tail of the stage's last plane (floats 432 .. 447: no read touches them).
#pragma unroll
        for (int j = 0; j < RAWB; ++j) {
            const int e = tid + NT * j;
            const int ch = e / 324, rem = e - ch * 324;
            const int r = rem / 18, c = rem - r * 18;
            const int sy = oy - 1 + r, sx = ox - 1 + c;
            const bool inside = e < W_STAGE_IC * 324 && sy >= 0 && sy < BHO && sx >= 0 && sx < BHO;
            raw_voff[j] = inside ? (unsigned)(ch * MAP + sy * BHO + sx) * 4u : 0x7FFF0000u;
            raw_lds[j] = e < W_STAGE_IC * 324 ? ch * W_PLANE + r * W_ROW + c : (W_STAGE_IC - 1) * W_PLANE + 432 + (tid & 15);
        }
    }
    auto load_raw = [&](int st, float4* pr) {
        const int soff = __builtin_amdgcn_readfirstlane(st * (W_STAGE_IC * MAP * 4));
        if constexpr (!BLOCKED) {
#pragma unroll
            for (int j = 0; j < RAW4; ++j) {
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, raw_voff[j], soff, 0));
                pr[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        } else {
            float* f = reinterpret_cast<float*>(pr);
#pragma unroll
            for (int j = 0; j < RAWB; ++j)
                f[j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, raw_voff[j], soff, 0));
        }
    };
    // the same fetch as inline asm: hipcc's waitcnt pass does not count it, so the loop places its own s_waitcnt vmcnt(N)
    // (round 6: a stage's fetches stay in flight across the stage's end; see WB_STAGE)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long pin = reinterpret_cast<unsigned long long>(in);
    const i32x4 rs_in_words = {__builtin_amdgcn_readfirstlane((int)(unsigned)pin),
                               __builtin_amdgcn_readfirstlane((int)((unsigned)(pin >> 32) & 0xffffu)),
                               BLOCKED ? (int)((unsigned)C * MAP * 4u) : 0x7fffffff, 0x00020000};
    auto load_raw_asm = [&](int st, float4* pr) __attribute__((always_inline)) {
        const int soff = __builtin_amdgcn_readfirstlane(st * (W_STAGE_IC * MAP * 4));
        if constexpr (!BLOCKED) {
#pragma unroll
            for (int j = 0; j < RAW4; ++j) {
                f32x4 v;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(raw_voff[j]), "s"(rs_in_words), "s"(soff) : "memory");
                pr[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
            float* f = reinterpret_cast<float*>(pr);
#pragma unroll
            for (int j = 0; j < RAWB; ++j)
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(f[j]) : "v"(raw_voff[j]), "s"(rs_in_words), "s"(soff) : "memory");
        }
    };
    // BF3: the track's response scaled by the power of two 2^kv (see the BF3 form of the loop): 4 |resp| 2^kv < 2^15, kv
    // clamped to +-100; 1 for an all-zero track and for one with an infinite maximum (its outputs are NaN / inf either way).
    // Set behind the first fetches (the loads of the maxima travel beside them).
    float sv = 1.0f, inv_sv = 1.0f;
    auto store_raw = [&](float* buf, const float4* pr) {
        if constexpr (!BLOCKED) {
#pragma unroll
            for (int j = 0; j < RAW4; ++j) {
                float* d = buf + raw_lds[j];
                d[0] = BF3 ? pr[j].x * sv : pr[j].x;
                d[1] = BF3 ? pr[j].y * sv : pr[j].y;
                d[2] = BF3 ? pr[j].z * sv : pr[j].z;
                d[3] = BF3 ? pr[j].w * sv : pr[j].w;
            }
        } else {
            const float* f = reinterpret_cast<const float*>(pr);
#pragma unroll
            for (int j = 0; j < RAWB; ++j) buf[raw_lds[j]] = BF3 ? f[j] * sv : f[j];
        }
    };
    // A operands: packed[tile][k][xi][lane][4]
    const unsigned a_voff = (unsigned)lane * 16u;
    const int a_base = __builtin_amdgcn_readfirstlane((int)((((size_t)tile0 * nk) * 4 + xi) * 1024));     // bytes
    const int a_tile = nk * 4096;                                                                          // bytes
    f32x4 aset[2][2][OCT];                    // A operands: [the stage in use / the next one][k-step][tile]
    auto load_a = [&](int st, f32x4 (*dst)[OCT]) {
#pragma unroll
        for (int o = 0; o < OCT; ++o)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int soff = __builtin_amdgcn_readfirstlane(a_base + o * a_tile + (2 * st + k) * 4096);
                dst[k][o] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff, soff, 0));
            }
    };

    // BF3: direct-to-LDS fetch of the fp16 A parts (see the BF3 form of the loop below)
    const int nkb = C >> 5;
    const int ab_block = 4 * W_AB_ROW;          // bytes per (tile, kb'): [xi][q][part][lane][8 fp16]
    const int ab_base = __builtin_amdgcn_readfirstlane(2 * C * C * 64 + ((tile0 + nh) * (nkb + 1) * 4 + xi) * W_AB_ROW);
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float* const a_lds = (lds_float*)(sm + W_RING * W_BUF);
    const int al_wave = __builtin_amdgcn_readfirstlane((xi * 2 + nh) * 512);            // floats: this wave's DMA slots
    const int al_row = __builtin_amdgcn_readfirstlane(xi * 2 * 512);                    // floats: the row's slots
    // (inline asm: through the builtin, hipcc drains vmcnt before the next LDS read of ANY address — right after the issue;
    // here the loop places its own s_waitcnt vmcnt(N), see WB_STAGE)
    const unsigned long long pa = reinterpret_cast<unsigned long long>(packed);
    const i32x4 rs_a_words = {__builtin_amdgcn_readfirstlane((int)(unsigned)pa),
                              __builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu)), 0x7fffffff, 0x00020000};
    const unsigned a_lds_byte = (unsigned)(size_t)a_lds;
    auto dma_part = [&](int q, int kbp, int part) __attribute__((always_inline)) {     // column q of rotated block kbp, tile tile0 + nh
        const int soff = __builtin_amdgcn_readfirstlane(ab_base + (ABL == 12 ? 0 : kbp) * ab_block + (q * 2 + part) * 1024);
        const unsigned dst = __builtin_amdgcn_readfirstlane(a_lds_byte + (unsigned)(q * W_A_SLOT + al_wave + part * 256) * 4u);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(a_voff), "s"(rs_a_words), "s"(soff) : "memory");
    };
    auto dma_slot = [&](int q, int kbp) __attribute__((always_inline)) {
        dma_part(q, kbp, 0);
        dma_part(q, kbp, 1);
    };
    auto load_al = [&](int q, u32x4 (*dst)[2]) __attribute__((always_inline)) {
        const float* src = sm + W_RING * W_BUF + q * W_A_SLOT + al_row + lane * 4;
#pragma unroll
        for (int o = 0; o < OCT; ++o)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                dst[o][part] = *reinterpret_cast<const u32x4*>(src + (o * 2 + part) * 256);
    };
    f32x4 acc[OCT][4][2 * NP];
#pragma unroll
    for (int o = 0; o < OCT; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 2 * NP; ++t) acc[o][q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_raw(0, prs[0]);
    load_raw(1, prs[1]);
    if constexpr (BF3) {                       // columns 0 and 1 of the first block: consumed in the stages 0 and 1
        dma_slot(0, 0);
        dma_slot(1, 0);
    }
    if constexpr (!BF3) load_a(0, aset[0]);
    if (ABL == 2 || ABL >= 5) {
#pragma unroll
        for (int o = 0; o < OCT; ++o) {
            aset[1][0][o] = aset[0][0][o];
            aset[1][1][o] = aset[0][1][o];
        }
    }
    // BF3: the plane maxima are requested here, beside the first fetches, and reduced BEHIND the zero fill (reduced here,
    // their wait would also wait for the fetches above and push the fill behind a cold memory round trip: +1.3 us)
    float pmv[2] = {0.0f, 0.0f};
    if constexpr (BF3 && ABL != 10) {
        const float* __restrict__ pmsrc = plane_max + (size_t)n * C;
        if (lane < C) pmv[0] = pmsrc[lane];
        if (lane + 64 < C) pmv[1] = pmsrc[lane + 64];
        __builtin_amdgcn_sched_barrier(0);
    }
    // (first global loads are in flight: their latency covers the fill and the weight fetch)
    {   // zero the stage buffers once: the halos stay zero for the whole main loop (a halo-only fill was measured
        // slower: its scattered ds_write_b32 and index arithmetic cost more than 14 ds_write_b128 per thread)
        float4* z = reinterpret_cast<float4*>(sm);
        for (int e = tid; e < W_RING * W_BUF / 4; e += NT) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Head taps of this tile's channels, fetched now, used in the epilogue — kept in NINE registers per lane: the
    // 4x4x1 matrix instruction can broadcast the A operand of ONE of its 16 blocks to all blocks (CBSZ = 4, ABID =
    // block), so block b of register r holds the four head weights (lane % 4 = output) of tap r*16 + b, and the
    // epilogue's 144 instructions name their tap through ABID.  No LDS image of the taps, no LDS read per tap.
    float hwv[9];
    auto load_hwv = [&]() __attribute__((always_inline)) {
        const int o = lane & 3, blk = lane >> 2;
        const float* wsrc = (tower == 1) ? P.reg_w + (size_t)o * C * 9
                                         : (o < 2 ? P.cls_w + (size_t)o * C * 9 : P.center_w);
        const bool live = (tower == 1) || (o < 3);
        wsrc += (size_t)oc0 * 9 + blk;
#pragma unroll
        for (int r = 0; r < 9; ++r) hwv[r] = live ? wsrc[r * 16] : 0.0f;
    };
    if constexpr (!BF3) load_hwv();          // BF3: after the main loop (nine registers the loop needs)
    if constexpr (BF3 && ABL != 10) {
        float m = fmaxf(pmv[0], pmv[1]);
        for (int c = lane + 128; c < C; c += 64) m = fmaxf(m, plane_max[(size_t)n * C + c]);
        m = plane_max_wave(m);
        const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);      // m < 2^(e - 126)
        if (e != 0 && e != 255) {
            int k = 139 - e;
            k = k < -100 ? -100 : (k > 100 ? 100 : k);
            sv = __uint_as_float((unsigned)(127 + k) << 23);
            inv_sv = __uint_as_float((unsigned)(127 - k) << 23);
        }
    }

    __syncthreads();                       // zero fill complete before interior writes
    // BF3: the first A parts are in LDS (the first raw planes are needed here anyway; waiting BEFORE the next fetches are
    // issued keeps their latency out of this wait)
    if constexpr (BF3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_raw(sm, prs[0]);
    store_raw(sm + W_BUF, prs[1]);
    load_raw(min(2, nstages - 1), prs[0]);
    load_raw(min(3, nstages - 1), prs[1]);
    // GroupNorm's affine parameters of the thread's epilogue channel: requested here (two registers through the loop);
    // fetched where they are used, their cold round trip stood in the epilogue (the barrier behind the exchange drains vmcnt)
    float ga = 0.0f, be = 0.0f;
    if constexpr (!BLOCKED) {
        ga = P.gamma[tower][oc0 + (ltid >> 4)];
        be = P.beta[tower][oc0 + (ltid >> 4)];
    }
    __syncthreads();

    // rows of the 4x4 patch this wave's xi-row combines:  i=0: d0-d2, 1: d1+d2, 2: d2-d1, 3: d1-d3
    const int row_a = (xi == 0) ? 0 : ((xi == 2) ? 2 : 1);
    const int row_b = (xi == 2) ? 1 : ((xi == 3) ? 3 : 2);
    const int kq = lane >> 4, xl = lane & 15;
    // Lane (kq, xl) supplies input channel kq of the k-step and, for N-tile t, the 2x2-output tile
    //     ty = xl/4 + 4*(t/2),  tx = 2*(xl%4) + (t%2):
    // t = 2p and 2p+1 are horizontal neighbours, so one 6-float row segment (one ds_read_b128 + one
    // ds_read_b64, both naturally aligned: haloed column 4*(xl%4)) serves both.  Banks: row stride 24 and
    // plane stride 448 put the four (plane, tile-row) classes of every 16-lane b128 group on disjoint
    // 16-bank windows; the b64 halves of planes 0/1 collide 2-way (4 instead of 2 LDS cycles).
    // OCT = 1: the wave walks p = 0, 1;  OCT = 2: the wave owns p = nh.
    const int patch0 = kq * W_PLANE + (2 * (xl >> 2) + (OCT == 2 ? 8 * nh : 0)) * W_ROW + 4 * (xl & 3);
    int ia = patch0 + row_a * W_ROW, ib = patch0 + row_b * W_ROW;
    asm volatile("" : "+v"(ia), "+v"(ib));   // two base registers; every (stage, k-step, pair) offset is an immediate

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
    float rd[NP][6][2];
    float bop[4][2 * NP];
    if (ABL >= 3) {
#pragma unroll
        for (int e = 0; e < 12 * NP; ++e) {
            (&bop[0][0])[e % (8 * NP)] = (float)(lane + e);
            (&rd[0][0][0])[e] = (float)(lane - e);
        }
    }
#define W_READ(BUFI, Q, RD)                                                                      \
    if (ABL != 3 && (ABL < 5 || ABL >= 7)) {                                                     \
        const float* oa = sm + ia + ((BUFI) * W_BUF + (Q) * 4 * W_PLANE);                        \
        const float* ob = sm + ib + ((BUFI) * W_BUF + (Q) * 4 * W_PLANE);                        \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {         /* tile rows ty and ty + 4 */    \
            f32x4 a4 = *reinterpret_cast<const f32x4*>(oa + p * 8 * W_ROW);                      \
            f32x2 a2 = *reinterpret_cast<const f32x2*>(oa + p * 8 * W_ROW + 4);                  \
            f32x4 b4 = *reinterpret_cast<const f32x4*>(ob + p * 8 * W_ROW);                      \
            f32x2 b2 = *reinterpret_cast<const f32x2*>(ob + p * 8 * W_ROW + 4);                  \
            if (BF3) asm volatile("" : "+v"(a4), "+v"(a2), "+v"(b4), "+v"(b2));                  \
            RD[p][0][0] = a4[0]; RD[p][0][1] = b4[0];                                            \
            RD[p][1][0] = a4[1]; RD[p][1][1] = b4[1];                                            \
            RD[p][2][0] = a4[2]; RD[p][2][1] = b4[2];                                            \
            RD[p][3][0] = a4[3]; RD[p][3][1] = b4[3];                                            \
            RD[p][4][0] = a2[0]; RD[p][4][1] = b2[0];                                            \
            RD[p][5][0] = a2[1]; RD[p][5][1] = b2[1];                                            \
        }                                                                                        \
    }
#define W_XFORM(RD, BOP)                                                                         \
    if (ABL != 3 && (ABL < 5 || ABL >= 7)) _Pragma("unroll") for (int p = 0; p < NP; ++p) {      \
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
    if (ABL != 4) _Pragma("unroll") for (int t = 0; t < 2 * NP; ++t)                             \
        _Pragma("unroll") for (int o = 0; o < OCT; ++o) {                                        \
            acc[o][0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[o][0], BOP[0][t], acc[o][0][t], 0, 0, 0);   \
            acc[o][1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[o][1], BOP[1][t], acc[o][1][t], 0, 0, 0);   \
            acc[o][2][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[o][2], BOP[2][t], acc[o][2][t], 0, 0, 0);   \
            acc[o][3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[o][3], BOP[3][t], acc[o][3][t], 0, 0, 0);   \
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
    // ---- BF3 form of the loop: the 16 GEMMs on the fp16 matrix pipe, two-part operands (round 6) --------------------
    // Every fp32 operand is the sum of two fp16 parts a = a1 + a2 (a1 = RNE(a), a2 = RNE(a - a1): |a - a1 - a2| <= 2^-23 |a|),
    // of a value scaled by a power of two so that fp16's exponent range is not an issue: the weights by 2^ku (chosen from
    // the largest |w| by tower_pack_kernel: largest |u| 2^ku in [2^12, 2^15)), the track's response by 2^kv (chosen here
    // from the track's largest |response|, plane_max: |V| 2^kv < 2^15 — V sums four responses); the accumulators are
    // multiplied back by 2^-kv 2^-ku after the output transform — exact, so the result is that of the unscaled operands.  A
    // product keeps three of the four part products,
    //     a b ~ a1 b1 + a1 b2 + a2 b1,          dropped: a2 b2 <= 2^-22 |a b|, typically 2^-24,
    // accumulated in fp32 by v_mfma_f32_16x16x32_f16.  Error against fp64 = the fp32 form's (tests/test_hip_parity.py holds
    // it to 1.25 x; measure/debug/tower_bf3_check.py), at HALF the matrix instructions of the three-part bf16 form of
    // rounds 4-5 (12 instead of 24 per stage) and a third less operand traffic.
    // K index of the instruction: lane group kq = lane/16 holds K = 8 kq + s, s = 0..7 <-> input channel 32 kb + 4 s +
    // kq: s is the k-step inside a 32-channel block, so the lane's operand of k-step s is element s of its 8-vector and a
    // stage (two k-steps) fills dword (stage % 4).  Bp[part][q][t] = the two fp16x8 B operands of xi column q, N-tile t.
    // One xi column per stage: the K blocks of column q end with the stages = q (mod 4) (the weights are stored rotated to
    // match, tower_pack_kernel), so every stage ends with the 12 instructions of ONE column: the operand registers are
    // single-buffered and the A parts of a column are needed once per four stages.
    // A parts: the two waves of an xi row need the same 4 KB per column and block; every wave brings HALF of them (those
    // of tile tile0 + nh) into LDS with direct-to-LDS loads (no registers, 1 KB per instruction), two stages before the
    // column is consumed, and both read them from there.
    //   LDS: A image behind the ring: slot q = [xi][o][part][lane][8 fp16] = 16 KB, four slots.
    //
    // The split.  The exact residual v - h of four operands is "C - B" on values that already sit in registers in a matrix
    // layout: v_mfma_f32_4x4x4_16b_f16 computes per lane D[i] = C[i] + sum_k A[lane's block][i][k] * B[k] with B = the
    // lane's OWN four fp16 values, so with A = -I (lane L holds -1.0 at element L % 4) one 8-cycle matrix instruction
    // returns the four residuals — exact, because the difference is representable and the products are +-h
    // (tools/ubench/mfma_residual.hip checks 2^28 values per exponent pattern bit for bit against the vector form, bf16
    // there).  A column's four operands (two tiles x two k-steps) are one such group: 4 conversions + 1 matrix instruction.
    //
    // The schedule: the two waves of a SIMD in ANTI-PHASE.  A stage is a vector phase P1 (raw store, operand transform, LDS
    // reads of the next stage, split) and a matrix phase P2 (the column's 12 instructions, the stage's fetches in their
    // shadow).  tools/ubench/mfma_residual.hip ("pingpong"): a wave that issues only matrix instructions is not slowed by a
    // partner issuing only vector instructions, and the partner keeps ~55 % of its rate.  So the waves 4-7 (nh = 1; wave w
    // and w + 4 share a SIMD) run ONE INTERVAL behind the waves 0-3: a barrier between P1 and P2, one extra barrier for the
    // late half before the loop and one for the early half after it — the k-th s_barrier of every wave pairs up, whatever
    // its address — and at any time one wave of a SIMD is in P1 while its partner is in P2.  One code stream, no branch.
    // Ring / slot hazards with the half-stage lag (X = waves 0-3, Y = 4-7; X runs P1(s), P2(s) in the intervals 2s, 2s + 1, Y
    // in 2s + 1, 2s + 2): raw planes of stage s + 2 are stored at the start of P1(s) [X 2s, Y 2s + 1] and first read in P1(s
    // + 1) [X 2s + 2]; the slot's last readers were in P1(s - 3).  A parts of column (s + 2) % 4 are requested in P2(s) [X 2s
    // + 1, Y 2s + 2], confirmed by the issuing wave's vmcnt(0) at the end of P1(s + 1) [Y: end of 2s + 3], read in P2(s + 2)
    // [X 2s + 5]; the slot's last readers were in P2(s - 2) [Y 2s - 2].  The tail keeps the cadence (three pseudo-stages).
    // Fetches are inline asm (hipcc's waitcnt pass does not count them; through the builtins it drains vmcnt at the next
    // LDS read): issued at the start of a matrix phase, waited for with ONE s_waitcnt vmcnt(0) at the end of the next vector
    // phase — a full stage later, and with no assumption on the order in which direct-to-LDS and register loads retire (a
    // vmcnt(N) form that relied on it produced a wrong A part in 2 of 300 workgroups of the placement test).
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x4 Bp[2][4][2];
    float rdb[NP][6][2];                        // second read set: both k-steps of the next stage are in flight
    if constexpr (BF3) {
#pragma unroll
        for (int e = 0; e < 16; ++e) (&Bp[0][0][0])[e] = (u32x4){0u, 0u, 0u, 0u};
    }
    const f16x4 negI = __builtin_bit_cast(f16x4, (u32x2){(lane & 3) == 0 ? 0x0000BC00u : ((lane & 3) == 1 ? 0xBC000000u : 0u),
                                                         (lane & 3) == 2 ? 0x0000BC00u : ((lane & 3) == 3 ? 0xBC000000u : 0u)});
    // (v_cvt_pk_f16_f32 through the vector conversion, NOT inline asm: the results are matrix-instruction operands a few
    // instructions later, and hipcc's hazard recognizer does not see what an asm statement writes)
#define WB_CVT(D, A, B) D = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){A, B}, f16x2));
    // the four columns' chains level by level, so that a dependent instruction is four matrix instructions behind the one it
    // waits for; fences keep hipcc from re-serialising the chains
#define WB_SPLIT(J, BA, BB)                                                                      \
    {                                                                                            \
        f32x4 v4[4];                                                                             \
        unsigned hh[4][2], mm[4][2];                                                             \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                          \
            v4[q] = (f32x4){BA[q][0], BB[q][0], BA[q][1], BB[q][1]};                             \
            WB_CVT(hh[q][0], v4[q][0], v4[q][1]) WB_CVT(hh[q][1], v4[q][2], v4[q][3])            \
        }                                                                                        \
        W_FENCE                                                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                            \
            v4[q] = __builtin_amdgcn_mfma_f32_4x4x4f16(negI, __builtin_bit_cast(f16x4, (u32x2){hh[q][0], hh[q][1]}), v4[q], 0, 0, 0); \
        W_FENCE                                                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                          \
            WB_CVT(mm[q][0], v4[q][0], v4[q][1]) WB_CVT(mm[q][1], v4[q][2], v4[q][3])            \
        }                                                                                        \
        W_FENCE                                                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                          \
            Bp[0][q][0][J] = hh[q][0]; Bp[0][q][1][J] = hh[q][1];                                \
            Bp[1][q][0][J] = mm[q][0]; Bp[1][q][1][J] = mm[q][1];                                \
        }                                                                                        \
    }
#define WB_MM(O, Q, T, PA, PB)                                                                   \
    if (ABL != 7) acc[O][Q][T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, AB[O][PA]), \
                                                          __builtin_bit_cast(f16x8, Bp[PB][Q][T]), acc[O][Q][T], 0, 0, 0);
#define WB_TERM(Q, PA, PB) WB_MM(0, Q, 0, PA, PB) WB_MM(1, Q, 0, PA, PB) WB_MM(0, Q, 1, PA, PB) WB_MM(1, Q, 1, PA, PB)
    // one column: smallest terms first; four independent accumulators per term
#define WB_Q(Q)                                                                                  \
    {                                                                                            \
        u32x4 AB[OCT][2];                                                                        \
        load_al(Q, AB);                                                                          \
        WB_QM(Q)                                                                                 \
    }
#define WB_QM(Q)                                                                                 \
    {                                                                                            \
        if (ABL == 7) { _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int e = 0; e < 4; ++e)  \
            acc[0][Q][t][e] += __uint_as_float(Bp[0][Q][t][e] ^ Bp[1][Q][t][e] ^ AB[0][0][e] ^ AB[1][1][e]); } \
        WB_TERM(Q, 1, 0) WB_TERM(Q, 0, 1) WB_TERM(Q, 0, 0)                                       \
    }
    // the row combination d_a +- d_b as one fused multiply-add with the wave's sign in a scalar register (exact: the
    // factor is +-1): ONE copy of the loop for all four xi rows, which leaves the instruction cache room for unrolling it
    const float sgn = __uint_as_float(__builtin_amdgcn_readfirstlane(xi == 1 ? 0x3f800000u : 0xbf800000u));
#define WB_XFORM(RD, BOP)                                                                        \
    _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                             \
        float w[6];                                                                              \
        _Pragma("unroll") for (int c = 0; c < 6; ++c) w[c] = __builtin_fmaf(sgn, RD[p][c][1], RD[p][c][0]); \
        BOP[0][2 * p] = w[0] - w[2];                                                             \
        BOP[1][2 * p] = w[1] + w[2];                                                             \
        BOP[2][2 * p] = w[2] - w[1];                                                             \
        BOP[3][2 * p] = w[1] - w[3];                                                             \
        BOP[0][2 * p + 1] = w[2] - w[4];                                                         \
        BOP[1][2 * p + 1] = w[3] + w[4];                                                         \
        BOP[2][2 * p + 1] = w[4] - w[3];                                                         \
        BOP[3][2 * p + 1] = w[3] - w[5];                                                         \
    }
    // ABL 15: timeline of ONE stage (KB 2, J 1) per wave: trace[(workgroup * 8 + wave) * 8 + slot], slots: 0 before the stage's
    // first barrier, 1 behind it, 2 raw planes stored, 3 transform + reads + split done, 4 behind the second barrier, 5 matrix
    // instructions and fetches issued, 6 behind vmcnt(N)
#define WB_STAMP(J, KB, SLOT) if (ABL == 15 && (KB) == 2 && (J) == 1 && trace && lane == 0) \
        trace[((size_t)blockIdx.x * 8 + wave) * 8 + (SLOT)] = (long long)__builtin_amdgcn_s_memtime();
#define WB_STAGE(J, KB)                                                                          \
    {                                                                                            \
        /* P1, vector phase: raw store, operand transform, next stage's LDS reads, split, this column's A parts requested */ \
        WB_STAMP(J, KB, 0)                                                                       \
        asm volatile("s_barrier" ::: "memory");                                                  \
        WB_STAMP(J, KB, 1)                                                                       \
        store_raw(sm + (((J) + 2) & 3) * W_BUF, prs[(J) & 1]);                                   \
        W_FENCE                                                                                  \
        WB_STAMP(J, KB, 2)                                                                       \
        float ba[4][2 * NP], bb[4][2 * NP];                                                      \
        u32x4 AB[OCT][2];                                                                        \
        WB_XFORM(rd, ba)                               /* k-step 2s */                           \
        W_READ(((J) + 1) & 3, 0, rd)                   /* k-step 2(s+1): next stage's buffer */  \
        WB_XFORM(rdb, bb)                              /* k-step 2s+1 */                         \
        W_READ(((J) + 1) & 3, 1, rdb)                                                            \
        WB_SPLIT(J, ba, bb)                                                                      \
        load_al(J, AB);                                                                          \
        W_FENCE                                                                                  \
        WB_STAMP(J, KB, 3)                                                                       \
        /* everything this wave fetched in the last matrix phase has arrived (no assumption on the order in which direct-to- \
           LDS and register loads retire): the A parts of column (J + 1) % 4, read after the next stage's second barrier, and  \
           the raw planes stored at the start of the next stage.  The raw stores above are behind at least the 12 LDS reads    \
           (LDS operations of a wave complete in order). */                                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(12)\n\ts_barrier" ::: "memory");                \
        W_FENCE                                                                                  \
        WB_STAMP(J, KB, 4)                                                                       \
        /* (the next stage's LDS reads issued HERE instead of in the vector phase were measured: loop 25.2 k against 23.3 k    \
           cycles — the vector phase is the shorter one of the pair, reads included)                                           \
           P2, matrix phase: the column's 12 instructions, this stage's fetches in their shadow (a full stage ahead of the    \
           wait above); slot (J + 2) % 4 was consumed two stages ago, its next use is two stages ahead */ \
        WB_TERM(J, 1, 0)                                                                         \
        W_FENCE                                                                                  \
        if (ABL != 9) load_raw_asm(min(4 * (KB) + (J) + 4, nstages - 1), prs[(J) & 1]);          \
        if (ABL != 9) dma_part(((J) + 2) & 3, (KB) + ((J) >= 2 ? 1 : 0), 0);                     \
        W_FENCE                                                                                  \
        WB_TERM(J, 0, 1)                                                                         \
        W_FENCE                                                                                  \
        if (ABL != 9) dma_part(((J) + 2) & 3, (KB) + ((J) >= 2 ? 1 : 0), 1);                     \
        W_FENCE                                                                                  \
        WB_TERM(J, 0, 0)                                                                         \
        W_FENCE                                                                                  \
        WB_STAMP(J, KB, 5)                                                                       \
    }
#define WB_BLOCK(KB) WB_STAGE(0, KB) WB_STAGE(1, KB) WB_STAGE(2, KB) WB_STAGE(3, KB)
#define WB_LOOP                                                                                  \
    W_READ(0, 0, rd)                                                                             \
    W_READ(0, 1, rdb)                                                                            \
    if (nh == 1) asm volatile("s_barrier" ::: "memory");            /* the late half: one interval behind */ \
    if (nkb == 4) {        /* C = 128 unrolled: no loop-carried register shuffle */              \
        WB_BLOCK(0) WB_BLOCK(1) WB_BLOCK(2) WB_BLOCK(3)                                          \
    } else if ((nkb & 1) == 0) {                     /* C = 64, 256, 512: two blocks per trip (half the shuffle) */ \
        for (int kb = 0; kb < nkb; kb += 2) { WB_BLOCK(kb) WB_BLOCK(kb + 1) }                    \
    } else {                                                                                     \
        for (int kb = 0; kb < nkb; ++kb) { WB_BLOCK(kb) }                                        \
    }                                                                                            \
    /* the last, partial blocks of the columns 0..2 (zero weights where their stages do not exist): three pseudo-stages  \
       in the loop's cadence */                                                                  \
    asm volatile("s_barrier" ::: "memory");                                                      \
    dma_slot(2, nkb);                                                                            \
    asm volatile("s_barrier" ::: "memory");                                                      \
    WB_Q(0)                                                                                      \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
    /* the loop's last raw fetches (re-fetches of the last stage, never stored) have landed: their registers may die */ \
    _Pragma("unroll") for (int j = 0; j < NRAW; ++j) asm volatile("" :: "v"(prs[0][j].x), "v"(prs[0][j].y), "v"(prs[0][j].z), "v"(prs[0][j].w), \
                                                                    "v"(prs[1][j].x), "v"(prs[1][j].y), "v"(prs[1][j].z), "v"(prs[1][j].w)); \
    asm volatile("s_barrier\n\ts_barrier" ::: "memory");                                         \
    WB_Q(1)                                                                                      \
    asm volatile("s_barrier\n\ts_barrier" ::: "memory");                                         \
    WB_Q(2)                                                                                      \
    if (nh == 0) asm volatile("s_barrier" ::: "memory");            /* the early half waits for the late one */
    W_TRACE(1)
    if constexpr (BF3) {
        static_assert(!BF3 || (OCT == 2 && (ABL == 0 || ABL == 7 || ABL == 9 || ABL == 10 || ABL == 12 || ABL == 15)), "BF3: two-tile workgroups only");
        {
            WB_LOOP
        }
        load_hwv();
    } else if (xi == 1) {        // the xi-row 1 combination adds its two patch rows, the others subtract
        constexpr bool PLUS = true;
        W_LOOP
    } else {
        constexpr bool PLUS = false;
        W_LOOP
    }
    __syncthreads();
#undef WB_LOOP
#undef WB_STAGE
#undef WB_STAMP
#undef WB_BLOCK
#undef WB_XFORM
#undef WB_Q
#undef WB_QM
#undef WB_TERM
#undef WB_MM
#undef WB_SPLIT
#undef WB_CVT
#undef W_LOOP
#undef W_FENCE
#undef W_STAGE
#undef W_MFMA
#undef W_XFORM
#undef W_READ
    W_TRACE(2)

    // ---- output transform --------------------------------------------------------------------------
    // acc[o][q][t][r] = M_(i=xi, j=q)[oc = 4*kq + r of tile o][tile = 16*(t + 2*NP*nh) + xl].  Column half (j) in
    // registers, row half (i) across the four xi-waves through the exchange image X[i][b][oc][tile] of tile o.
#pragma unroll
    for (int o = 0; o < OCT; ++o)
#pragma unroll
        for (int t = 0; t < 2 * NP; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p0 = (acc[o][0][t][r] + acc[o][1][t][r]) + acc[o][2][t][r];
                const float p1 = (acc[o][1][t][r] - acc[o][2][t][r]) - acc[o][3][t][r];
                float* dst = sm + o * W_EPI_FLOATS + ((xi * 2) * 16 + (kq * 4 + r)) * W_XOC +
                             16 * (t + (OCT == 2 ? 2 * nh : 0)) + xl;
                dst[0] = p0;
                dst[16 * W_XOC] = p1;
            }
    float* X = sm + sub * W_EPI_FLOATS;
    float* planes = X + W_PL_OFF;
    float* chs = X + W_ST_OFF;             // [16] channel sums, [16] channel squared deviations
    for (int e = ltid; e < 16 * 68; e += 256) {         // halo of the head planes (interiors are written below)
        const int pl = e / 68, c = e - pl * 68;
        const int row = c < 18 ? 0 : (c < 36 ? 17 : 1 + ((c - 36) >> 1));
        const int col = c < 18 ? c : (c < 36 ? c - 18 : (((c - 36) & 1) ? 17 : 0));
        planes[pl * T_PLANE + row * 18 + col] = 0.0f;
    }
    __syncthreads();

    W_TRACE(3)
    // thread = (output channel ocl = ltid/16, tiles 4*x16 + t): 4 tiles x 2x2 outputs, one ds_read_b128 per (i, b)
    const int ocl = ltid >> 4, x16 = ltid & 15;
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
    if constexpr (BF3) {        // back from the scaled operands (two exact multiplications by powers of two)
        const float inv_su = packed[(size_t)2 * C * C * 16 + (size_t)tiles * (nkb + 1) * 8192 + 1];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) y[t][a][b] = (y[t][a][b] * inv_sv) * inv_su;
    }

    if constexpr (BLOCKED) {
        // the block's convolution outputs that no earlier block owns (blocks overlap by 32 - BHO rows / columns) -> part
        // = conv [N][2C][BHO*BHO]; tile 4*x16 + t -> (ty, tx) as below
        constexpr int OV = 32 - BHO;
        float* __restrict__ cdst = part + ((size_t)n * 2 * C + (size_t)tower * C + oc0 + ocl) * MAP;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ty = (x16 & 3) + 4 * (x16 >> 3), tx = 2 * t + ((x16 >> 2) & 1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int lr = 2 * ty + a, lc = 2 * tx + b;
                    if ((oy == 0 || lr >= OV) && (ox == 0 || lc >= OV)) cdst[(oy + lr) * BHO + ox + lc] = y[t][a][b];
                }
        }
        return;
    }
    // ---- GroupNorm (two-pass, fp32) + affine + ReLU --------------------------------------------------
    const float inv_cnt = 1.0f / (float)(cpg * 256);
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += (y[t][0][0] + y[t][0][1]) + (y[t][1][0] + y[t][1][1]);
    s = group16_sum(s);
    float mean = 0.0f, var = 0.0f;
    const int g0 = (ocl / cpg) * cpg;
    if (cpg == 4) {
        // a group = 4 channels = the four 16-lane rows of THIS wave: no LDS, no barrier; rows are added in channel
        // order, as the general path below does
        // (v_readlane at wave-uniform lanes, not __shfl: that is ds_bpermute — four dependent LDS round trips per sum)
#define W_RL(V, L) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(V), (L)))
        mean = (((0.0f + W_RL(s, 0)) + W_RL(s, 16)) + W_RL(s, 32)) + W_RL(s, 48);
    } else {
        if (x16 == 0) chs[ocl] = s;
        __syncthreads();
        for (int ch = g0; ch < g0 + cpg; ++ch) mean += chs[ch];
    }
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
    if (cpg == 4) {
        var = (((0.0f + W_RL(sq, 0)) + W_RL(sq, 16)) + W_RL(sq, 32)) + W_RL(sq, 48);
#undef W_RL
    } else {
        if (x16 == 0) chs[16 + ocl] = sq;
        __syncthreads();
        for (int ch = g0; ch < g0 + cpg; ++ch) var += chs[16 + ch];
    }
    const float rstd = 1.0f / sqrtf(var * inv_cnt + eps);
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
                    pl[(2 * ty + a + 1) * 18 + 2 * tx + b + 1] = relu_nan(v);
                }
        }
    }
    __syncthreads();

    W_TRACE(4)
    // ---- fused partial heads: this tile's 16 channels x 9 taps -> 4 head outputs per position ---------
    // v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products per instruction, block = 4 neighbouring
    // positions: A = the 4 head weights of one (channel, tap) — broadcast from block ABID of a weight register
    // (see hwv above) — B = the lane's own activation, D[output][position] accumulates in 4 VGPRs per lane, the
    // layout the stores below need.  256 MACs per 8-cycle instruction, one LDS read (the activation) per tap, all
    // 144 offsets immediates.  One fmaf chain per output in (channel, tap) order, as in predictor.hip.
    {
        // position of a thread: the two 16-lane halves of a ds_read_b32 lane group read rows r and r + 8 — 144 floats = 16
        // banks apart, conflict-free; rows r and r + 1 (the linear order) are 18 floats apart and collide on two banks, which
        // doubled the LDS cycles of all 1,152 reads of this phase (heads 4.7 k -> 3.x k cycles per workgroup)
        const int py = ((ltid >> 6) << 1) + ((ltid >> 5) & 1) + (((ltid >> 4) & 1) << 3), px = ltid & 15;
        const float* pl0 = planes + py * 18 + px;
        // Activations of FOUR channels (36 taps) are requested one group ahead of the 36 instructions that consume them
        // (hipcc otherwise issues a read a few instructions before its use and the phase waits on LDS latency), and the
        // taps alternate between two accumulators (even / odd tap index: a dependent 4x4x1 instruction needs two wait states
        // behind its predecessor), added at the end.
        f32x4 hacc = {0.0f, 0.0f, 0.0f, 0.0f}, hacc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        float av[2][36];
#define W_HLOAD(G, BUF) _Pragma("unroll") for (int e = 0; e < 36; ++e)                                        \
        av[BUF][e] = pl0[((G) * 4 + e / 9) * T_PLANE + ((e % 9) / 3) * 18 + (e % 9) % 3];
#define W_HMM(G, BUF) heads_group<G>(hwv, av[BUF], hacc, hacc1, std::make_integer_sequence<int, 36>{});
        W_HLOAD(0, 0)
        __builtin_amdgcn_sched_barrier(0);
        W_HLOAD(1, 1)
        __builtin_amdgcn_sched_barrier(0);
        W_HMM(0, 0)
        __builtin_amdgcn_sched_barrier(0);
        W_HLOAD(2, 0)
        __builtin_amdgcn_sched_barrier(0);
        W_HMM(1, 1)
        __builtin_amdgcn_sched_barrier(0);
        W_HLOAD(3, 1)
        __builtin_amdgcn_sched_barrier(0);
        W_HMM(2, 0)
        __builtin_amdgcn_sched_barrier(0);
        W_HMM(3, 1)
#undef W_HMM
#undef W_HLOAD
        hacc[0] += hacc1[0];
        hacc[1] += hacc1[1];
        hacc[2] += hacc1[2];
        hacc[3] += hacc1[3];
        float* __restrict__ dst = part + ((size_t)n * tiles + etile) * 4 * 256 + py * 16 + px;
        dst[0 * 256] = hacc[0];
        dst[1 * 256] = hacc[1];
        dst[2 * 256] = hacc[2];
        dst[3 * 256] = hacc[3];
    }
    W_TRACE(5)
#undef W_TRACE
}

// 16-channel tiles per workgroup of the 16 x 16 towers for N tracks (1 or 2)
static int tower_tiles_per_workgroup(int N, int C) {
    // Round 6: the two-tile split form (fp16 x 2 operands on the matrix pipe) for EVERY track count — it is the faster form
    // everywhere (kernel durations by start / stop events at C = 128, measure/debug/tower_forms_by_tracks.py,
    // profiles/r06_tower_forms_by_tracks.jsonl: 1 track 17.3 vs 18.5 us for the one-tile fp32 form, 16 tracks 18.9 vs 20.1,
    // 30: 20.0 vs 29.5, 64: 36.7 vs 55.5, 100: 68.0 vs 92.9), and one form means that a track's logits no longer depend on how
    // many other tracks the frame has (rounds 4-5: fp32 up to 16 tracks, three-part bf16 above).
    // SMOT_TOWER_OCT = 1 / 2 forces a form, SMOT_TOWER_BF3 = 0 the fp32 form of two tiles, in the measurement library.
    (void)N;
    (void)C;
    int oct = 2;
    if (knobs().tower_oct == 1 || knobs().tower_oct == 2) oct = knobs().tower_oct;
    return oct;
}

int launch_tower_wino(const float* resp, const float* packed, const TowerParams& P, int N, int C, int cpg, float eps,
                      float* part, unsigned* zero_words, hipStream_t st, const float* plane_max) {
    const int tiles = 2 * (C / 16);
    const int oct = tower_tiles_per_workgroup(N, C);
    const bool bf3 = oct == 2 && knobs().tower_bf3 != 0;
    const size_t smem = (size_t)(bf3 ? W_RING * W_BUF + W_A_FLOATS : w_smem_floats(oct)) * sizeof(float);
    const int grid = ((N + 7) / 8) * 8 * (tiles / oct);
    if (oct == 2) {                            // 115 KB of dynamic LDS
        const void* fn = bf3 ? reinterpret_cast<const void*>(&tower_wino_kernel<0, 2, 0, true>)
                             : reinterpret_cast<const void*>(&tower_wino_kernel<0, 2>);
#ifdef SMOT_DEBUG
        if (bf3 && knobs().wino_abl == 7) fn = reinterpret_cast<const void*>(&tower_wino_kernel<7, 2, 0, true>);
        if (bf3 && knobs().wino_abl == 9) fn = reinterpret_cast<const void*>(&tower_wino_kernel<9, 2, 0, true>);
        if (bf3 && knobs().wino_abl == 10) fn = reinterpret_cast<const void*>(&tower_wino_kernel<10, 2, 0, true>);
        if (bf3 && knobs().wino_abl == 12) fn = reinterpret_cast<const void*>(&tower_wino_kernel<12, 2, 0, true>);
        if (bf3 && knobs().wino_abl == 15) fn = reinterpret_cast<const void*>(&tower_wino_kernel<15, 2, 0, true>);
#endif
        const int rco = ensure_lds_optin(fn, smem, "predictor towers (winograd)");
        if (rco) return rco;
    }
    if (bf3) {
        SMOT_REQUIRE(plane_max != nullptr, "predictor towers: the split form needs the response's plane maxima");
#define WB_LAUNCH(A)                                                                                              \
    SMOT_LAUNCH((tower_wino_kernel<A, 2, 0, true>), dim3(grid), dim3(512), smem, st, resp, packed, P, N, C, cpg, eps, part, \
                zero_words, g_trace, plane_max)
#ifdef SMOT_DEBUG
        switch (knobs().wino_abl) {
            case 7: WB_LAUNCH(7); break;        // no matrix instructions (timing, WRONG results)
            case 9: WB_LAUNCH(9); break;
            case 10: WB_LAUNCH(10); break;
            case 12: WB_LAUNCH(12); break;      // every stage fetches A block 0 (L2-hot; timing, WRONG results)
            case 15: WB_LAUNCH(15); break;      // one stage's timeline per wave (trace layout [workgroup][wave][8])
            default: WB_LAUNCH(0); break;
        }
#else
        WB_LAUNCH(0);
#endif
#undef WB_LAUNCH
        return check_launch("predictor towers (winograd, fp16 x 2)");
    }
#define W_LAUNCH(A, O)                                                                                            \
    SMOT_LAUNCH((tower_wino_kernel<A, O>), dim3(grid), dim3(256 * O), smem, st, resp, packed, P, N, C, cpg, eps, part, \
                zero_words, g_trace, plane_max)
#ifdef SMOT_DEBUG
    if (oct == 1) {
        switch (knobs().wino_abl) {          // timing ablations (wrong results): measurement library only
            case 1: W_LAUNCH(1, 1); break;
            case 2: W_LAUNCH(2, 1); break;
            case 3: W_LAUNCH(3, 1); break;
            case 4: W_LAUNCH(4, 1); break;
            case 5: W_LAUNCH(5, 1); break;
            case 6: W_LAUNCH(6, 1); break;
            default: W_LAUNCH(0, 1); break;
        }
    } else {
        switch (knobs().wino_abl) {
            case 3: W_LAUNCH(3, 2); break;
            case 4: W_LAUNCH(4, 2); break;
            case 6: W_LAUNCH(6, 2); break;
            default: W_LAUNCH(0, 2); break;
        }
    }
#else
    if (oct == 2) {
        W_LAUNCH(0, 2);
    } else {
        W_LAUNCH(0, 1);
    }
#endif
#undef W_LAUNCH
    return check_launch("predictor towers (winograd)");
}

// 16-channel tiles per workgroup of the blocked 29 x 29 convolution for N tracks (1 or 2)
static int tower_blocks_tiles_per_workgroup(int N, int C) {
    const int tiles = 2 * (C / 16);
    const int np8 = ((N + 7) / 8) * 8 * 4;                 // block tracks: tracks padded to the XCD count, four blocks each
    // same dispatch-round arithmetic as tower_tiles_per_workgroup, on four block tracks per track (the two-tile form on
    // three-part bf16 operands: 108.7 -> 97.0 us at 30 tracks, measure/debug/tower_blocked_bf3.py: 0.9 of the fp32 rounds)
    const int w1 = np8 * tiles, w2 = np8 * (tiles / 2);
    const float c1 = (w1 <= 256) ? 20.0f
                                 : 27.0f * (float)(w1 / 512) + ((w1 % 512) == 0 ? 0.0f : ((w1 % 512) <= 256 ? 15.0f : 27.0f));
    const float c2 = (knobs().tower_bf3 != 0 ? 0.9f : 1.0f) * (24.5f * (float)(w2 / 256) + ((w2 % 256) == 0 ? 0.0f : 22.5f));
    int oct = (c2 < c1) ? 2 : 1;
    if (knobs().tower_oct == 1 || knobs().tower_oct == 2) oct = knobs().tower_oct;
    return oct;
}

// Convolution output of the two towers for a 29 x 29 response (blocked mode above): conv [N][2C][841].
int launch_tower_wino_blocks(const float* resp, const float* packed, const TowerParams& P, int N, int C, int cpg,
                             float* conv, unsigned* zero_words, hipStream_t st, const float* plane_max) {
    const int tiles = 2 * (C / 16);
    const int np8 = ((N + 7) / 8) * 8 * 4;
    const int oct = tower_blocks_tiles_per_workgroup(N, C);
    const size_t smem = (size_t)w_smem_floats(oct) * sizeof(float);
    const int grid = np8 * (tiles / oct);
    if (oct == 2 && knobs().tower_bf3 != 0) {      // the three-part bf16 form of the main loop (BF3 above)
        const size_t smem3 = (size_t)(W_RING * W_BUF + W_A_FLOATS) * sizeof(float);
        const int rco = ensure_lds_optin(reinterpret_cast<const void*>(&tower_wino_kernel<0, 2, 29, true>), smem3,
                                         "predictor towers (winograd, 29x29 in blocks, fp16 x 2)");
        if (rco) return rco;
        SMOT_REQUIRE(plane_max != nullptr, "predictor towers: the split form needs the response's plane maxima");
        SMOT_LAUNCH((tower_wino_kernel<0, 2, 29, true>), dim3(grid), dim3(512), smem3, st, resp, packed, P, N, C, cpg, 0.0f, conv,
                    zero_words, (long long*)nullptr, plane_max);
    } else if (oct == 2) {
        const int rco = ensure_lds_optin(reinterpret_cast<const void*>(&tower_wino_kernel<0, 2, 29>), smem,
                                         "predictor towers (winograd, 29x29 in blocks)");
        if (rco) return rco;
        SMOT_LAUNCH((tower_wino_kernel<0, 2, 29>), dim3(grid), dim3(512), smem, st, resp, packed, P, N, C, cpg, 0.0f, conv,
                    zero_words, (long long*)nullptr, plane_max);
    } else {
        SMOT_LAUNCH((tower_wino_kernel<0, 1, 29>), dim3(grid), dim3(256), smem, st, resp, packed, P, N, C, cpg, 0.0f, conv,
                    zero_words, (long long*)nullptr, plane_max);
    }
    return check_launch("predictor towers (winograd, 29x29 in blocks)");
}

}  // namespace smot

extern "C" int smot_emm_tower_form(int N, int C, int Ho) {
    using namespace smot;
    const bool pow2 = C > 0 && (C & (C - 1)) == 0;
    if (N <= 0 || !pow2 || C % 32 != 0 || C > 512 || (Ho != 16 && Ho != 29)) return 0;     // (predictor.hip: mfma_ok)
    const int oct = Ho == 29 ? tower_blocks_tiles_per_workgroup(N, C) : tower_tiles_per_workgroup(N, C);
    return oct == 1 ? 1 : (knobs().tower_bf3 != 0 ? 3 : 2);
}

extern "C" long long smot_emm_tower_pack_floats(int C) {
    if (C <= 0 || C % 16 != 0) return 0;          // the packed path needs 16-channel tiles
    // fp32 image 2 C^2 x 16, then (C % 32 == 0) the two-part fp16 image: C/32 + 1 rotated blocks of 8192 floats per
    // 16-channel tile, and a header of four words {largest |w| (bits), 2^-ku, 0, 0}
    return (long long)2 * C * C * 16 + ((C % 32 == 0) ? (long long)(2 * C / 16) * (C / 32 + 1) * 8192 + 4 : 0);
}

extern "C" int smot_emm_tower_pack(const float* cls_tower_w, const float* reg_tower_w, int C, float* packed,
                                   smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(C > 0 && C % 16 == 0, "tower_pack: C=%d must be a multiple of 16", C);
    SMOT_REQUIRE(cls_tower_w && reg_tower_w && packed, "tower_pack: null pointer");
    SMOT_REQUIRE(((uintptr_t)packed & 15) == 0, "tower_pack: output must be 16-byte aligned");
    unsigned* hdr = nullptr;
    if (C % 32 == 0) {
        const size_t half_floats = (size_t)(2 * C / 16) * (C / 32 + 1) * 8192;
        const hipError_t e = hipMemsetAsync(packed + (size_t)2 * C * C * 16, 0, (half_floats + 4) * 4, (hipStream_t)stream);
        SMOT_REQUIRE(e == hipSuccess, "tower_pack: memset failed: %s", hipGetErrorString(e));
        hdr = reinterpret_cast<unsigned*>(packed + (size_t)2 * C * C * 16 + half_floats);
        hipLaunchKernelGGL(tower_wmax_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, cls_tower_w, reg_tower_w, C * C * 9, hdr);
    }
    hipLaunchKernelGGL(tower_pack_kernel, dim3((2 * C * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls_tower_w,
                       reg_tower_w, C, packed, hdr);
    return check_launch("tower_pack");
}
