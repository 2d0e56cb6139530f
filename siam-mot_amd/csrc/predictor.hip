// K3 — EMM prediction towers (conv3x3 + GroupNorm + ReLU) and heads (conv3x3 + bias).
//
// Replaces EMMPredictor.forward (reference EMM/feature_extractor.py:62-69; make_conv3x3 /
// group_norm [UPSTREAM maskrcnn_benchmark modeling/make_layers.py]).
//
// Towers (the only dense contraction of the EMM head): both towers read the same response map,
// so they are one implicit GEMM per track,  D[2C x Ho*Ho] = W[2C x 9C] * im2col(resp)[9C x Ho*Ho],
// run on the fp32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, bitwise a k-ordered fmaf
// chain — parity with the fp32 reference is kept; no bf16/xf32 shortcut).
//   workgroup  = (track, 32 output channels) x all 256 positions  -> every GroupNorm group of the
//                tile is complete inside the workgroup, so GN + ReLU are fused into the epilogue;
//   wave w     = the 32 channels x positions of rows 4w..4w+3 (2 M-tiles x 4 N-tiles, 32 acc VGPRs);
//   K order    = input-channel chunk (16) > tap (9) > 4 channels per MFMA;
//   LDS        = per chunk an A image [36 k-steps][2][64 lanes] (lane-linear, conflict-free
//                ds_read_b32) and a B image of 16 zero-haloed 18x18 planes at plane stride 336
//                (336 mod 32 = 16: the two k-rows of a 32-lane group land on disjoint banks);
//                both double-buffered (79,872 B -> two workgroups per CU), next chunk prefetched
//                into registers while the current one feeds the matrix cores.
// Heads: 7 output channels, 4 MFLOP/track — plain VALU from LDS-staged tower planes, head weights
// as wave-uniform scalar loads.
// A generic (any Ho / channel count) tower kernel covers shapes the MFMA tiling does not.
#include "tower_common.h"
#include "knobs.h"

namespace smot {

constexpr int T_IC = 16;                          // input channels per K chunk
constexpr int T_STEPS = 9 * (T_IC / 4);           // MFMA k-steps per chunk (36)
constexpr int T_B_FLOATS = T_IC * T_PLANE;        // 5376
constexpr int T_B_PER_THREAD = T_IC;              // one position of each plane per thread

// MT = number of 16-channel M tiles per workgroup (output-channel tile T_OC = 16*MT).
// MT = 2 halves the B-operand traffic per MFMA; MT = 1 doubles the number of workgroups, which is
// what fills the chip (and puts two waves on every SIMD) at small track counts.
// ABL: 0 = the kernel; 1 = per-chunk staging removed (every chunk recomputes buffer 0: wrong results,
// timing only); 2 = fused partial heads removed.  Instantiated in the measurement library only (knobs.h:
// SMOT_TOWER_ABL).
template <int MT, int ABL>
__global__ void __launch_bounds__(256)
tower_mfma_kernel(const float* __restrict__ resp, TowerParams P, int C, int cpg, float eps,
                  float* __restrict__ part) {
    constexpr int T_OC = 16 * MT;
    constexpr int A_FLOATS = T_STEPS * MT * 64;
    constexpr int BUF_FLOATS = A_FLOATS + T_B_FLOATS;
    constexpr int A_PER_THREAD = T_OC * T_IC * 9 / 256;       // 9 * MT
    constexpr int HW_PER_THREAD = (T_OC * 36 + 255) / 256;    // head taps staged per thread
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_per_tower = C / T_OC;
    const int tiles = 2 * tiles_per_tower;
    const int n = blockIdx.x / tiles;
    const int tile = blockIdx.x - n * tiles;
    const int tower = tile / tiles_per_tower;
    const int oc0 = (tile - tower * tiles_per_tower) * T_OC;
    const float* __restrict__ W = P.w[tower];
    const float* __restrict__ in = resp + (size_t)n * C * 256;

    // zero both buffers once: the B halos stay zero for the whole kernel
    for (int e = tid; e < 2 * BUF_FLOATS; e += 256) sm[e] = 0.0f;

    // head taps of this tile's channels, fetched now, used in the epilogue: hw[(ocl*9+tap)*4 + o]
    float hwreg[HW_PER_THREAD];
#pragma unroll
    for (int j = 0; j < HW_PER_THREAD; ++j) {
        const int idx = tid + 256 * j;
        float v = 0.0f;
        if (idx < T_OC * 36) {
            const int o = idx / (T_OC * 9);
            const int rem = idx - o * (T_OC * 9);      // ocl*9 + tap
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

    // A image element d = ((s*MT + m)*4 + kq)*16 + i  <->  W[oc0 + m*16 + i][ic0 + 4*(s%4) + kq][tap = s/4].
    // Threads walk the DESTINATION index, so LDS stores are lane-linear (conflict-free); the global
    // reads are 16 rows x 576-byte runs per chunk that the L1/L2 serve across the 9*MT passes.
    float pa[A_PER_THREAD], pb[T_B_PER_THREAD];
    int asrc[A_PER_THREAD];
#pragma unroll
    for (int j = 0; j < A_PER_THREAD; ++j) {
        const int d = tid + 256 * j;
        const int i = d & 15, kq_ = (d >> 4) & 3;
        const int sm_ = d >> 6;                 // s*MT + m
        const int s = sm_ / MT, m = sm_ - s * MT;
        const int tap = s >> 2, icl = 4 * (s & 3) + kq_;
        asrc[j] = ((oc0 + m * 16 + i) * C + icl) * 9 + tap;
    }
    auto load_chunk = [&](int ic0) {
#pragma unroll
        for (int j = 0; j < A_PER_THREAD; ++j) pa[j] = W[asrc[j] + ic0 * 9];
#pragma unroll
        for (int j = 0; j < T_B_PER_THREAD; ++j) pb[j] = in[(size_t)(ic0 + j) * 256 + tid];
    };
    auto store_chunk = [&](float* buf) {
        float* A = buf;
        float* B = buf + A_FLOATS;
#pragma unroll
        for (int j = 0; j < A_PER_THREAD; ++j) A[tid + 256 * j] = pa[j];
        const int y = tid >> 4, x = tid & 15;
#pragma unroll
        for (int j = 0; j < T_B_PER_THREAD; ++j) B[j * T_PLANE + (y + 1) * 18 + (x + 1)] = pb[j];
    };

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    __syncthreads();                 // zero-fill complete before interior writes
    store_chunk(sm);
    __syncthreads();

    const int nchunks = C / T_IC;
    const int kq = lane >> 4, xl = lane & 15;
    for (int c = 0; c < nchunks; ++c) {
        float* buf = sm + ((ABL == 1) ? 0 : (c & 1)) * BUF_FLOATS;
        if (ABL != 1 && c + 1 < nchunks) load_chunk((c + 1) * T_IC);
        const float* A = buf + lane;
        const float* B = buf + A_FLOATS + kq * T_PLANE + (4 * wave) * 18 + xl;
        // k-step S = tap*4 + icq.  Operands of step S+1 are read from LDS before the MFMAs of step S are
        // issued (two named register sets); the accumulator pin + sched_barrier per step keep hipcc from
        // re-clustering the reads next to their use (measured before: 51 % MFMA-pipe utilisation with
        // two waves per SIMD, waves stalling on lgkmcnt right in front of every MFMA group).
        float a_0[MT], b_0[4], a_1[MT], b_1[4], a_2[MT], b_2[4];
#define SMOT_LD(S, A_, B_)                                                                   \
    {                                                                                        \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) A_[m] = A[((S) * MT + m) * 64];        \
        const float* Bp = B + (4 * ((S) & 3)) * T_PLANE + (((S) >> 2) / 3) * 18 + (((S) >> 2) % 3); \
        B_[0] = Bp[0 * 18];                                                                  \
        B_[1] = Bp[1 * 18];                                                                  \
        B_[2] = Bp[2 * 18];                                                                  \
        B_[3] = Bp[3 * 18];                                                                  \
    }
#define SMOT_MM(A_, B_)                                                                      \
    {                                                                                        \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                      \
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[m], B_[0], acc[m][0], 0, 0, 0); \
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[m], B_[1], acc[m][1], 0, 0, 0); \
            acc[m][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[m], B_[2], acc[m][2], 0, 0, 0); \
            acc[m][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[m], B_[3], acc[m][3], 0, 0, 0); \
        }                                                                                    \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                        \
            asm volatile("" : "+a"(acc[m][0]), "+a"(acc[m][1]), "+a"(acc[m][2]), "+a"(acc[m][3])); \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    }
        // operands are fetched TWO k-steps ahead (three rotating register sets): one step (~136 cycles of
        // MFMA issue) does not cover the LDS latency with 8 waves per CU reading
        SMOT_LD(0, a_0, b_0)
        SMOT_LD(1, a_1, b_1)
        __builtin_amdgcn_sched_barrier(0);
        static_assert(T_STEPS % 3 == 0, "rotation period");
#pragma unroll
        for (int s3 = 0; s3 < T_STEPS; s3 += 3) {
            // the next chunk's operands (global loads issued at the top of this chunk) go to the OTHER LDS
            // buffer half-way through this chunk's MFMAs, so the end of the chunk is only the barrier
            if (ABL != 1 && s3 == T_STEPS / 2 && c + 1 < nchunks) store_chunk(sm + ((c + 1) & 1) * BUF_FLOATS);
            SMOT_LD(s3 + 2, a_2, b_2)
            SMOT_MM(a_0, b_0)
            if (s3 + 3 < T_STEPS) SMOT_LD(s3 + 3, a_0, b_0)
            SMOT_MM(a_1, b_1)
            if (s3 + 4 < T_STEPS) SMOT_LD(s3 + 4, a_1, b_1)
            SMOT_MM(a_2, b_2)
        }
#undef SMOT_MM
#undef SMOT_LD
        if (ABL != 1) __syncthreads();
    }
    if (ABL == 1) __syncthreads();

    // ---- fused GroupNorm (two-pass, fp32) + affine + ReLU -----------------------------------
    // acc[m][t][r] = conv[oc = m*16 + kq*4 + r][pos = (4*wave + t)*16 + xl]
    // scratch in the (now free) A image of buffer 0; head taps go to the A image of buffer 1
    float* red = sm;                       // [4 waves][T_OC]
    float* stat = sm + 4 * T_OC;           // [T_OC] group mean, then rstd, per channel
    float* hw = sm + BUF_FLOATS;           // [T_OC*9][4]
#pragma unroll
    for (int j = 0; j < HW_PER_THREAD; ++j) {
        const int idx = tid + 256 * j;
        if (idx < T_OC * 36) {
            const int o = idx / (T_OC * 9);
            const int rem = idx - o * (T_OC * 9);
            hw[rem * 4 + o] = hwreg[j];
        }
    }
    const float inv_cnt = 1.0f / (float)(cpg * 256);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = acc[m][0][r] + acc[m][1][r] + acc[m][2][r] + acc[m][3][r];
            s = group16_sum(s);
            if (xl == 0) red[wave * T_OC + m * 16 + kq * 4 + r] = s;
        }
    __syncthreads();
    if (tid < T_OC) {
        const int g0 = (tid / cpg) * cpg;
        float s = 0.0f;
        for (int ch = g0; ch < g0 + cpg; ++ch)
            s += red[0 * T_OC + ch] + red[1 * T_OC + ch] + red[2 * T_OC + ch] + red[3 * T_OC + ch];
        stat[tid] = s * inv_cnt;
    }
    __syncthreads();
    float mean[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) mean[m][r] = stat[m * 16 + kq * 4 + r];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float d = acc[m][t][r] - mean[m][r];
                s += d * d;
            }
            s = group16_sum(s);
            if (xl == 0) red[wave * T_OC + m * 16 + kq * 4 + r] = s;
        }
    __syncthreads();
    if (tid < T_OC) {
        const int g0 = (tid / cpg) * cpg;
        float s = 0.0f;
        for (int ch = g0; ch < g0 + cpg; ++ch)
            s += red[0 * T_OC + ch] + red[1 * T_OC + ch] + red[2 * T_OC + ch] + red[3 * T_OC + ch];
        stat[tid] = 1.0f / sqrtf(s * inv_cnt + eps);
    }
    __syncthreads();
    // normalised + ReLU'd activations -> zero-haloed LDS planes (interiors of the B images: plane ocl
    // lives in buffer ocl>>4, slot ocl&15; the halos were zeroed at kernel start and never written)
    const float* __restrict__ gamma = P.gamma[tower];
    const float* __restrict__ beta = P.beta[tower];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ocl = m * 16 + kq * 4 + r;
            const float rstd = stat[ocl];
            const float ga = gamma[oc0 + ocl], be = beta[oc0 + ocl];
            float* pl = sm + m * BUF_FLOATS + A_FLOATS + (kq * 4 + r) * T_PLANE;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = (acc[m][t][r] - mean[m][r]) * rstd * ga + be;
                v = relu_nan(v);
                pl[(4 * wave + t + 1) * 18 + xl + 1] = v;
            }
        }
    __syncthreads();

    // ---- fused partial heads: this tile's T_OC channels x 9 taps -> 4 head outputs per position ----
    // thread = output position; hw rows are read as one broadcast ds_read_b128 per (channel, tap)
    {
        const int y = tid >> 4, x = tid & 15;
        float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;
#pragma unroll
        for (int m = 0; m < (ABL == 2 ? 0 : MT); ++m) {
            const float* pl0 = sm + m * BUF_FLOATS + A_FLOATS + y * 18 + x;
#pragma unroll 4
            for (int cl = 0; cl < 16; ++cl) {
                const float* pl = pl0 + cl * T_PLANE;
                const float4* wrow = reinterpret_cast<const float4*>(hw + (m * 16 + cl) * 36);
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
        }
        float* __restrict__ dst = part + ((size_t)n * tiles + tile) * 4 * 256 + tid;
        dst[0 * 256] = h0;
        dst[1 * 256] = h1;
        dst[2 * 256] = h2;
        dst[3 * 256] = h3;
    }
}

// logits[n][ch][pos] = bias[ch] + sum over the tiles of ch's tower of part[n][tile][o][pos]
// (tile order, fixed), ReLU on the four reg channels.  Ho == 16 only (MFMA path).  grid (N, 7).
template <int TPT>
__global__ void __launch_bounds__(256)
heads_combine_kernel(const float* __restrict__ part, const float* __restrict__ cls_b,
                     const float* __restrict__ center_b, const float* __restrict__ reg_b,
                     float* __restrict__ logits) {
    const int n = blockIdx.x;
    const int ch = blockIdx.y;
    const int pos = threadIdx.x;
    const int side = ch >= 3;
    const int o = side ? ch - 3 : ch;
    const float* __restrict__ p = part + ((size_t)n * 2 * TPT + side * TPT) * 4 * 256 + (size_t)o * 256 + pos;
    float v[TPT];
#pragma unroll
    for (int t = 0; t < TPT; ++t) v[t] = p[(size_t)t * 4 * 256];      // all loads in flight at once
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < TPT; ++t) s += v[t];
    s += (ch < 2) ? cls_b[ch] : ((ch == 2) ? center_b[0] : reg_b[ch - 3]);
    if (side) s = relu_nan(s);
    logits[((size_t)n * 7 + ch) * 256 + pos] = s;
}

// Any Ho / C: one workgroup per (track, tower, GroupNorm group); direct convolution, outputs kept
// in LDS for the two-pass GroupNorm.
__global__ void __launch_bounds__(256)
tower_generic_kernel(const float* __restrict__ resp, TowerParams P, int C, int Ho, int cpg, float eps,
                     float* __restrict__ tower_ws) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[256];
    const int HW = Ho * Ho;
    const int groups = C / cpg;
    const int n = blockIdx.x / (2 * groups);
    const int rem = blockIdx.x - n * 2 * groups;
    const int tower = rem / groups;
    const int oc0 = (rem - tower * groups) * cpg;
    const float* __restrict__ W = P.w[tower];
    const float* __restrict__ in = resp + (size_t)n * C * HW;
    const int total = cpg * HW;
    float lsum = 0.0f;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int ocl = e / HW;
        const int pos = e - ocl * HW;
        const int y = pos / Ho, x = pos - y * Ho;
        const float* __restrict__ w = W + (size_t)(oc0 + ocl) * C * 9;
        float acc = 0.0f;
        for (int ic = 0; ic < C; ++ic) {
            const float* __restrict__ pl = in + (size_t)ic * HW;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = y + dy - 1;
                if (yy < 0 || yy >= Ho) continue;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int xx = x + dx - 1;
                    if (xx < 0 || xx >= Ho) continue;
                    acc = fmaf(pl[yy * Ho + xx], w[ic * 9 + dy * 3 + dx], acc);
                }
            }
        }
        sm[e] = acc;
        lsum += acc;
    }
    auto block_sum = [&](float v) -> float {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        const float r = red[0];
        __syncthreads();
        return r;
    };
    const float mean = block_sum(lsum) / (float)total;
    float lvar = 0.0f;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const float d = sm[e] - mean;
        lvar += d * d;
    }
    const float rstd = 1.0f / sqrtf(block_sum(lvar) / (float)total + eps);
    const float* __restrict__ gamma = P.gamma[tower];
    const float* __restrict__ beta = P.beta[tower];
    float* __restrict__ dst = tower_ws + ((size_t)n * 2 * C + (size_t)tower * C + oc0) * HW;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int ocl = e / HW;
        const float v = (sm[e] - mean) * rstd * gamma[oc0 + ocl] + beta[oc0 + ocl];
        dst[e] = relu_nan(v);
    }
}

struct HeadsParams {
    const float* cls_w;      // [2, C, 3, 3]
    const float* cls_b;      // [2]
    const float* center_w;   // [1, C, 3, 3]
    const float* center_b;   // [1]
    const float* reg_w;      // [4, C, 3, 3]
    const float* reg_b;      // [4]
};

// Heads: grid (N, 2): y = 0 -> cls0, cls1, center from the cls tower; y = 1 -> reg l/t/r/b (+ReLU) from
// the reg tower.  Tower planes are staged H_IC at a time into zero-haloed LDS planes; a thread owns
// NPOS output positions and accumulates 4 output channels per position (the cls side computes the
// center filter twice and drops the copy).  Filter taps are wave-uniform: addressed only through
// kernel arguments and loop counters, so they are fetched with scalar loads and used as the SGPR
// operand of v_fmac — no LDS traffic, no VGPRs, no per-lane global loads for weights.
constexpr int H_IC = 16;
constexpr int H_MAXPOS = 4;     // position chunks of 256 (grid z) for Ho*Ho up to 1024

template <int NPOS>
__global__ void __launch_bounds__(256)
heads_kernel(const float* __restrict__ tower_ws, HeadsParams H, int C, int Ho, float* __restrict__ logits) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const bool reg_side = (blockIdx.y == 1);
    const int n = blockIdx.x;
    const int HW = Ho * Ho;
    const int PW = Ho + 2;
    const int plane = PW * PW;
    const int C9 = C * 9;
    const float* __restrict__ w0 = reg_side ? H.reg_w : H.cls_w;
    const float* __restrict__ w1 = w0 + C9;
    const float* __restrict__ w2 = reg_side ? H.reg_w + 2 * C9 : H.center_w;
    const float* __restrict__ w3 = reg_side ? H.reg_w + 3 * C9 : H.center_w;
    const float* __restrict__ in = tower_ws + ((size_t)n * 2 * C + (reg_side ? C : 0)) * HW;

    const int pos0 = blockIdx.z * 256 * NPOS;          // grid z = position chunk (maps larger than 256 cells)
    int off[NPOS];
#pragma unroll
    for (int p = 0; p < NPOS; ++p) {
        const int pos = min(pos0 + (int)threadIdx.x + p * 256, HW - 1);
        const int y = pos / Ho, x = pos - y * Ho;
        off[p] = y * PW + x;
    }
    float acc[NPOS][4];
#pragma unroll
    for (int p = 0; p < NPOS; ++p)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[p][o] = 0.0f;

    for (int e = threadIdx.x; e < H_IC * plane; e += 256) sm[e] = 0.0f;   // halos stay zero
    __syncthreads();
    for (int ic0 = 0; ic0 < C; ic0 += H_IC) {
        const int icn = min(H_IC, C - ic0);
        for (int e = threadIdx.x; e < icn * HW; e += 256) {
            const int ic = e / HW;
            const int pos = e - ic * HW;
            const int y = pos / Ho, x = pos - y * Ho;
            sm[ic * plane + (y + 1) * PW + (x + 1)] = in[(size_t)(ic0 + ic) * HW + pos];
        }
        __syncthreads();
        // two input channels per trip: the 8 scalar tap loads of both are issued before the single
        // lgkmcnt(0) wait, halving the exposed scalar-cache latency
#pragma unroll 2
        for (int ic = 0; ic < icn; ++ic) {
            const int wb = (ic0 + ic) * 9;          // wave-uniform
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float t0 = w0[wb + k], t1 = w1[wb + k], t2 = w2[wb + k], t3 = w3[wb + k];
#pragma unroll
                for (int p = 0; p < NPOS; ++p) {
                    const float v = sm[ic * plane + off[p] + (k / 3) * PW + (k % 3)];
                    acc[p][0] = fmaf(v, t0, acc[p][0]);
                    acc[p][1] = fmaf(v, t1, acc[p][1]);
                    acc[p][2] = fmaf(v, t2, acc[p][2]);
                    acc[p][3] = fmaf(v, t3, acc[p][3]);
                }
            }
        }
        __syncthreads();
    }
    const int n_out = reg_side ? 4 : 3;
    const int out_ch0 = reg_side ? 3 : 0;
    float bias[4];
    bias[0] = reg_side ? H.reg_b[0] : H.cls_b[0];
    bias[1] = reg_side ? H.reg_b[1] : H.cls_b[1];
    bias[2] = reg_side ? H.reg_b[2] : H.center_b[0];
    bias[3] = reg_side ? H.reg_b[3] : 0.0f;
#pragma unroll
    for (int p = 0; p < NPOS; ++p) {
        const int pos = pos0 + threadIdx.x + p * 256;
        if (pos < HW) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < n_out) {
                    float v = acc[p][o] + bias[o];
                    if (reg_side) v = relu_nan(v);
                    logits[((size_t)n * 7 + out_ch0 + o) * HW + pos] = v;
                }
            }
        }
    }
}

}  // namespace smot

namespace smot {
// tiles_out != nullptr: "towers only" — when the MFMA path applies, stop after the tower kernel, leave
// the per-tile partial head sums in tower_ws and return the tile count (the caller's decode sums them);
// *tiles_out = 0 means the generic path ran and `logits` is complete.
int predictor_impl(const float* resp, int N, int C, int Ho, const float* cls_tower_w, const float* cls_gn_w,
                   const float* cls_gn_b, const float* reg_tower_w, const float* reg_gn_w, const float* reg_gn_b,
                   const float* cls_w, const float* cls_b, const float* center_w, const float* center_b,
                   const float* reg_w, const float* reg_b, int gn_groups, float gn_eps,
                   const float* tower_packed, float* tower_ws, float* logits, smot_stream_t stream, int* tiles_out,
                   unsigned* zero_words, bool* zeroed, const float* plane_max) {
    if (zeroed) *zeroed = false;
    if (tiles_out) *tiles_out = 0;
    SMOT_REQUIRE(N >= 0 && C > 0 && Ho > 0 && gn_groups > 0, "predictor: bad sizes N=%d C=%d Ho=%d groups=%d", N, C,
                 Ho, gn_groups);
    SMOT_REQUIRE(C % gn_groups == 0, "predictor: C=%d not divisible by gn_groups=%d", C, gn_groups);
    SMOT_REQUIRE(Ho * Ho <= 256 * H_MAXPOS, "predictor: Ho=%d too large (max %d positions)", Ho, 256 * H_MAXPOS);
    if (N == 0) return SMOT_OK;
    SMOT_REQUIRE(resp && cls_tower_w && cls_gn_w && cls_gn_b && reg_tower_w && reg_gn_w && reg_gn_b && cls_w &&
                     cls_b && center_w && center_b && reg_w && reg_b && tower_ws && logits,
                 "predictor: null pointer");
    hipStream_t st = (hipStream_t)stream;
    TowerParams T;
    T.w[0] = cls_tower_w;
    T.gamma[0] = cls_gn_w;
    T.beta[0] = cls_gn_b;
    T.w[1] = reg_tower_w;
    T.gamma[1] = reg_gn_w;
    T.beta[1] = reg_gn_b;
    const int cpg = C / gn_groups;
    T.cls_w = cls_w;
    T.center_w = center_w;
    T.reg_w = reg_w;
    const bool pow2 = (C & (C - 1)) == 0;      // tile counts 1,2,4,...: what heads_combine is built for
    const bool mfma_ok = (Ho == 16) && (C % 32 == 0) && pow2 && (C <= 512) && (cpg <= 16) && (16 % cpg == 0);
    const bool wino = mfma_ok && tower_packed != nullptr && !knobs().tower_direct;
    // The split (fp16 x 2) form of the Winograd towers scales every track's response by a power of two chosen from the
    // largest |response| of its planes (tower_wino.hip).  The pooling + correlation kernel hands the plane maxima over
    // (`plane_max`); for a response that came without them one small launch makes them, in the head of the `logits`
    // buffer — N C <= N 7 Ho^2 floats that nothing reads before the tower kernel has and the logits overwrite afterwards.
    if (plane_max == nullptr && tower_packed != nullptr && !knobs().tower_direct && smot_emm_tower_form(N, C, Ho) == 3 &&
        (mfma_ok || Ho == 29)) {
        SMOT_REQUIRE(C <= 7 * Ho * Ho, "predictor: C=%d too large for the plane maxima's scratch", C);
        const int rcm = launch_plane_absmax(resp, N * C, Ho * Ho, logits, st);
        if (rcm) return rcm;
        plane_max = logits;
    }
    if (mfma_ok) {
        // 16-channel tiles double the workgroup count: use them while 32-channel tiles would leave
        // CUs idle or single-wave (256 CUs; two workgroups per CU fit either way)
        const int blocks32 = N * 2 * (C / 32);
        // measured in round 1 (profiles/r01_n_kernel_bench_tower_ablations.md): 16-channel tiles 1.79 us/track @N=30; 32-channel tiles 2.06 us/track @N=100
        const bool narrow = wino || blocks32 < 2 * 256 || !knobs().tower_wide;
        const int mt = narrow ? 1 : 2;
        const int tiles_per_tower = C / (16 * mt);
        const size_t smem = (size_t)2 * (T_STEPS * mt * 64 + T_B_FLOATS) * sizeof(float);
#ifdef SMOT_DEBUG
        const int abl = knobs().tower_abl;       // timing ablations (wrong results): measurement library only
        const void* fn = narrow ? (abl == 1 ? (const void*)tower_mfma_kernel<1, 1>
                                             : abl == 2 ? (const void*)tower_mfma_kernel<1, 2>
                                                        : (const void*)tower_mfma_kernel<1, 0>)
                                : (const void*)tower_mfma_kernel<2, 0>;
#else
        const void* fn = narrow ? (const void*)tower_mfma_kernel<1, 0> : (const void*)tower_mfma_kernel<2, 0>;
#endif
        if (!wino) {
            const int rco = ensure_lds_optin(fn, smem, "predictor towers");
            if (rco) return rco;
        }
        // tower_ws holds the per-tile partial head sums [N][2*tiles_per_tower][4][256] (<= N*2C*256 floats)
        const dim3 tg(N * 2 * tiles_per_tower);
        timer_mark(1, 0, st);
        if (wino) {
            SMOT_REQUIRE(((uintptr_t)tower_packed & 15) == 0, "predictor: tower_packed must be 16-byte aligned");
            int rcw = launch_tower_wino(resp, tower_packed, T, N, C, cpg, gn_eps, tower_ws, zero_words, st, plane_max);
            if (rcw) return rcw;
            if (zeroed) *zeroed = (zero_words != nullptr);
        } else if (!narrow) {
            SMOT_LAUNCH((tower_mfma_kernel<2, 0>), tg, dim3(256), smem, st, resp, T, C, cpg, gn_eps, tower_ws);
#ifdef SMOT_DEBUG
        } else if (abl == 1) {
            hipLaunchKernelGGL((tower_mfma_kernel<1, 1>), tg, dim3(256), smem, st, resp, T, C, cpg, gn_eps, tower_ws);
        } else if (abl == 2) {
            hipLaunchKernelGGL((tower_mfma_kernel<1, 2>), tg, dim3(256), smem, st, resp, T, C, cpg, gn_eps, tower_ws);
#endif
        } else {
            SMOT_LAUNCH((tower_mfma_kernel<1, 0>), tg, dim3(256), smem, st, resp, T, C, cpg, gn_eps, tower_ws);
        }
        timer_mark(1, 1, st);
        int rc = check_launch("predictor towers");
        if (rc) return rc;
        if (tiles_out) {
            *tiles_out = tiles_per_tower;
            return SMOT_OK;
        }
#define SMOT_COMBINE(TPT)                                                                                \
    hipLaunchKernelGGL(heads_combine_kernel<TPT>, dim3(N, 7), dim3(256), 0, st, (const float*)tower_ws, cls_b, \
                       center_b, reg_b, logits)
        switch (tiles_per_tower) {
            case 1: SMOT_COMBINE(1); break;
            case 2: SMOT_COMBINE(2); break;
            case 4: SMOT_COMBINE(4); break;
            case 8: SMOT_COMBINE(8); break;
            case 16: SMOT_COMBINE(16); break;
            case 32: SMOT_COMBINE(32); break;
            default:
                set_error("predictor: unsupported tile count %d (C=%d)", tiles_per_tower, C);
                return SMOT_ERR_UNSUPPORTED;
        }
#undef SMOT_COMBINE
        return check_launch("predictor heads combine");
    }
    // the reference's second shape family (Ho = 29): Winograd in 16 x 16 blocks when the packed filters are given, else
    // a direct matrix-core kernel of its own; anything else: scalar
    int rc = SMOT_ERR_UNSUPPORTED;
    if (tower_packed != nullptr && !knobs().tower_direct) {
        SMOT_REQUIRE(((uintptr_t)tower_packed & 15) == 0, "predictor: tower_packed must be 16-byte aligned");
        timer_mark(1, 0, st);
        rc = launch_tower_conv_wino(resp, tower_packed, T, N, C, Ho, cpg, gn_eps, cls_b, center_b, reg_b, tower_ws, logits,
                                    zero_words, st, plane_max);
        timer_mark(1, 1, st);
    }
    if (rc == SMOT_ERR_UNSUPPORTED)
        rc = launch_tower_conv(resp, T, N, C, Ho, cpg, gn_eps, cls_b, center_b, reg_b, tower_ws, logits, zero_words, st);
    if (rc == SMOT_OK) {
        if (zeroed) *zeroed = (zero_words != nullptr);
        return SMOT_OK;
    }
    if (rc == SMOT_ERR_UNSUPPORTED) {
        const size_t smem = (size_t)cpg * Ho * Ho * sizeof(float);
        SMOT_REQUIRE(smem <= 64 * 1024, "predictor: GroupNorm group too large for the generic tower kernel");
        hipLaunchKernelGGL(tower_generic_kernel, dim3(N * 2 * gn_groups), dim3(256), smem, st, resp, T, C, Ho, cpg,
                           gn_eps, tower_ws);
        rc = check_launch("predictor towers");
    }
    if (rc) return rc;

    HeadsParams H;
    H.cls_w = cls_w;
    H.cls_b = cls_b;
    H.center_w = center_w;
    H.center_b = center_b;
    H.reg_w = reg_w;
    H.reg_b = reg_b;
    const size_t hsmem = (size_t)H_IC * (Ho + 2) * (Ho + 2) * sizeof(float);
    SMOT_REQUIRE(hsmem <= 64 * 1024, "predictor: Ho=%d too large for the heads kernel", Ho);
    // one position per thread, position chunks of 256 on grid z: 4x the workgroups of the former 4-positions-per-
    // thread form for a 29x29 map (60 workgroups left three quarters of the chip idle: 177 us)
    hipLaunchKernelGGL(heads_kernel<1>, dim3(N, 2, (Ho * Ho + 255) / 256), dim3(256), hsmem, st,
                       (const float*)tower_ws, H, C, Ho, logits);
    return check_launch("predictor heads");
}
}  // namespace smot

extern "C" int smot_emm_predictor_fwd(const float* resp, int N, int C, int Ho, const float* cls_tower_w,
                                      const float* cls_gn_w, const float* cls_gn_b, const float* reg_tower_w,
                                      const float* reg_gn_w, const float* reg_gn_b, const float* cls_w,
                                      const float* cls_b, const float* center_w, const float* center_b,
                                      const float* reg_w, const float* reg_b, int gn_groups, float gn_eps,
                                      const float* tower_packed, float* tower_ws, float* logits,
                                      smot_stream_t stream) {
    return smot::predictor_impl(resp, N, C, Ho, cls_tower_w, cls_gn_w, cls_gn_b, reg_tower_w, reg_gn_w, reg_gn_b, cls_w,
                                cls_b, center_w, center_b, reg_w, reg_b, gn_groups, gn_eps, tower_packed, tower_ws,
                                logits, stream, nullptr, nullptr, nullptr, nullptr);
}

