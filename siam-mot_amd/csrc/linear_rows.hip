// Linear layers on a handful of rows — the box head's fc6 / fc7 / cls_score|bbox_pred for the propagated tracks.
//
// Replaces, inside CombinedROIHeads._refine_tracks (reference siammot/modelling/roi_heads.py:60-84 ->
// box_head/box_head.py:46-50 -> [UPSTREAM] FPN2MLPFeatureExtractor / FPNPredictor): y = act(x W^T + b) with
// x [M, K], M = number of propagated tracks (tens), W [N, K] as torch.nn.Linear stores it.  At M = 30 the library GEMM
// (hipBLASLt through torch) runs fc6 (K = 6272, N = 1024: 25.7 MB of weights) on 128 workgroups in 20 us; the layer is
// WEIGHT STREAMING — every weight is used M times — so the job is to pull W through all 256 CUs at HBM speed.
//
//   * split-K: workgroup = (64 output neurons, one K slice), grid chosen for about two workgroups per CU; four waves,
//     each a 16-neuron tile; per step of 64 k a lane loads four float4 of its neuron's weights (256 contiguous bytes
//     per neuron per step: whole lines) and the workgroup stages the x slice [rows][64 k] in LDS once for all four
//     waves; the loads of up to four steps (the usual slice) are all issued before the first use;
//   * the products run on v_mfma_f32_16x16x4_f32 (exact fp32 multiply-add, D[neuron][row]): A = weights, B = x^T read
//     back from LDS as float4 (element j of a lane's float4 is the operand of the j-th MFMA, for A and B alike, so no
//     shuffles);
//   * the partial sums of the K slices are written out and a second, tiny launch adds them in slice order (+ bias,
//     ReLU): deterministic, and the kernel boundary is the cross-XCD visibility point (no atomics).
// Rows: up to 128 (1..8 row tiles of 16; round 4: 64 -> 128, so that 100 propagated tracks — BASELINE.json configs[2] —
// stay on this path instead of the library GEMM); K must be a multiple of 4 (the wrapper falls back to the library otherwise).
#include "smot_common.h"
#include "tower_common.h"

namespace smot {

constexpr int LR_NB = 64;        // neurons per workgroup (4 waves x 16)
constexpr int LR_KS = 64;        // k per step (a lane holds four float4 of weights: 4 KB per wave per step in flight)
constexpr int LR_XS = 68;        // LDS row stride of the staged x slice (floats)

template <int MT>                // row tiles of 16
__global__ void __launch_bounds__(256)
linear_rows_partial_kernel(const float* __restrict__ x, int M, int K, const float* __restrict__ W, int N,
                           const float* __restrict__ W2, int N1, int kslice, float* __restrict__ part) {
    // (W2 != nullptr: two layers on the same input side by side — neurons [0, N1) are rows of W, [N1, N) rows of W2:
    // cls_score | bbox_pred in one launch)
    // A workgroup lives for a few microseconds, so its loads must not be a chain of round trips: the slice is walked
    // in "quads" of QS steps and ALL loads of a quad — QS x 4 float4 of weights per lane, the x chunks of its steps —
    // are issued before the first use (one HBM latency per quad; the usual slice is one quad).
    constexpr int QS = MT <= 2 ? 4 : (MT <= 4 ? 2 : 1);       // steps per quad (LDS: QS x MT*16 x 68 floats <= 35 KB)
    __shared__ __attribute__((aligned(16))) float xs[QS][MT * 16][LR_XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = blockIdx.x, s = blockIdx.y;
    const int k_begin = s * kslice, k_end = min(K, k_begin + kslice);
    const int n = nb * LR_NB + wave * 16 + (lane & 15);
    const int kq = lane >> 4;
    const bool n_ok = n < N;
    const int nc = min(n, N - 1);
    const float* __restrict__ wrow = (W2 != nullptr && nc >= N1) ? W2 + (size_t)(nc - N1) * K : W + (size_t)nc * K;
    // x staging: thread t loads float4 (row t / 16 (+16 q), k4 = (t % 16) * 4)
    const int xr = tid >> 4, xk = (tid & 15) * 4;
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int k0 = k_begin; k0 < k_end; k0 += QS * LR_KS) {
        // every load is UNCONDITIONAL at a clamped in-bounds address (a predicated load is a branch with its own wait:
        // eight serial round trips); out-of-range elements are zeroed in registers afterwards
        float4 xreg[QS][MT], wreg[QS][4];
#pragma unroll
        for (int u = 0; u < QS; ++u)
#pragma unroll
            for (int q = 0; q < MT; ++q) {
                const int r = min(xr + 16 * q, M - 1), k = min(k0 + u * LR_KS + xk, K - 4);
                xreg[u][q] = *reinterpret_cast<const float4*>(x + (size_t)r * K + k);
            }
#pragma unroll
        for (int u = 0; u < QS; ++u)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int k = min(k0 + u * LR_KS + 16 * h + 4 * kq, K - 4);
                wreg[u][h] = *reinterpret_cast<const float4*>(wrow + k);
            }
        __builtin_amdgcn_sched_barrier(0);               // all loads of the quad are in flight before anything waits
#pragma unroll
        for (int u = 0; u < QS; ++u) {
#pragma unroll
            for (int q = 0; q < MT; ++q)
                if (!(xr + 16 * q < M && k0 + u * LR_KS + xk < k_end)) xreg[u][q] = zero4;
#pragma unroll
            for (int h = 0; h < 4; ++h)
                if (!(n_ok && k0 + u * LR_KS + 16 * h + 4 * kq < k_end)) wreg[u][h] = zero4;
        }
        if (k0 != k_begin) __syncthreads();              // the previous quad's readers are done with xs
#pragma unroll
        for (int u = 0; u < QS; ++u)
#pragma unroll
            for (int q = 0; q < MT; ++q) *reinterpret_cast<float4*>(&xs[u][xr + 16 * q][xk]) = xreg[u][q];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < QS; ++u) {
            if (k0 + u * LR_KS < k_end) {                    // workgroup-uniform
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const float a4[4] = {wreg[u][h].x, wreg[u][h].y, wreg[u][h].z, wreg[u][h].w};
                    float4 b4[MT];
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        b4[t] = *reinterpret_cast<const float4*>(&xs[u][t * 16 + (lane & 15)][16 * h + 4 * kq]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < MT; ++t) {
                            const float b = j == 0 ? b4[t].x : (j == 1 ? b4[t].y : (j == 2 ? b4[t].z : b4[t].w));
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b, acc[t], 0, 0, 0);
                        }
                }
            }
        }
    }
    // D[neuron = 4 * (lane >> 4) + r][row = lane & 15 (+ 16 t)] -> part[s][nb][row][neuron]
    float* __restrict__ p = part + ((size_t)(s * gridDim.x + nb) * (MT * 16)) * LR_NB;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int row = t * 16 + (lane & 15);
        *reinterpret_cast<float4*>(p + (size_t)row * LR_NB + wave * 16 + 4 * (lane >> 4)) =
            make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
}

// The same layer with its input still in the K-slice partial sums of the PREVIOUS layer (xp: SX slices of
// [nblk_x][rows_pad][64], that layer's bias bx and ReLU applied here, in linear_rows_reduce_kernel's order: slices in
// order, then the bias, then the ReLU): the previous layer's reduction launch disappears.  For layers whose own K slice is
// one 64-k step (kslice == 64: the head behind fc7), so that a workgroup's x slice is exactly one neuron block of the
// previous layer and the extra loads (SX per element instead of one) stay a single round trip.
template <int MT>
__global__ void __launch_bounds__(256)
linear_rows_partial_xpart_kernel(const float* __restrict__ xp, int SX, int nblk_x, const float* __restrict__ bx, int relu_x,
                                 int M, int K, const float* __restrict__ W, int N, const float* __restrict__ W2, int N1,
                                 float* __restrict__ part) {
    constexpr int SMAX = MT <= 4 ? 16 : 4;         // slices of the previous layer summed per load batch (registers: MT x SMAX float4)
    __shared__ __attribute__((aligned(16))) float xs[MT * 16][LR_XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = blockIdx.x, s = blockIdx.y;
    const int k0 = s * LR_KS;                      // this workgroup's 64 k = neuron block s of the previous layer
    const int n = nb * LR_NB + wave * 16 + (lane & 15);
    const int kq = lane >> 4;
    const bool n_ok = n < N;
    const int nc = min(n, N - 1);
    const float* __restrict__ wrow = (W2 != nullptr && nc >= N1) ? W2 + (size_t)(nc - N1) * K : W + (size_t)nc * K;
    const int xr = tid >> 4, xk = (tid & 15) * 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 wreg[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) wreg[h] = *reinterpret_cast<const float4*>(wrow + min(k0 + 16 * h + 4 * kq, K - 4));
    const size_t xstride = (size_t)nblk_x * (MT * 16) * LR_NB;      // between the previous layer's K slices
    float4 xsum[MT];
#pragma unroll
    for (int q = 0; q < MT; ++q) xsum[q] = zero4;
    for (int s0 = 0; s0 < SX; s0 += SMAX) {
        float4 t[MT][SMAX];
#pragma unroll
        for (int q = 0; q < MT; ++q)
#pragma unroll
            for (int j = 0; j < SMAX; ++j)
                t[q][j] = *reinterpret_cast<const float4*>(xp + (size_t)min(s0 + j, SX - 1) * xstride +
                                                           ((size_t)s * (MT * 16) + xr + 16 * q) * LR_NB + xk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < MT; ++q)
#pragma unroll
            for (int j = 0; j < SMAX; ++j)
                if (s0 + j < SX) {
                    xsum[q].x = add_rn(xsum[q].x, t[q][j].x);
                    xsum[q].y = add_rn(xsum[q].y, t[q][j].y);
                    xsum[q].z = add_rn(xsum[q].z, t[q][j].z);
                    xsum[q].w = add_rn(xsum[q].w, t[q][j].w);
                }
    }
    float4 b4 = zero4;
    if (bx != nullptr) b4 = *reinterpret_cast<const float4*>(bx + k0 + xk);
#pragma unroll
    for (int q = 0; q < MT; ++q) {
        float4 v = xsum[q];
        if (bx != nullptr) {
            v.x = add_rn(v.x, b4.x);
            v.y = add_rn(v.y, b4.y);
            v.z = add_rn(v.z, b4.z);
            v.w = add_rn(v.w, b4.w);
        }
        if (relu_x) {
            v.x = relu_nan(v.x);
            v.y = relu_nan(v.y);
            v.z = relu_nan(v.z);
            v.w = relu_nan(v.w);
        }
        if (!(xr + 16 * q < M)) v = zero4;
        *reinterpret_cast<float4*>(&xs[xr + 16 * q][xk]) = v;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h)
        if (!n_ok) wreg[h] = zero4;
    __syncthreads();
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float a4[4] = {wreg[h].x, wreg[h].y, wreg[h].z, wreg[h].w};
        float4 b4m[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) b4m[t] = *reinterpret_cast<const float4*>(&xs[t * 16 + (lane & 15)][16 * h + 4 * kq]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float b = j == 0 ? b4m[t].x : (j == 1 ? b4m[t].y : (j == 2 ? b4m[t].z : b4m[t].w));
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b, acc[t], 0, 0, 0);
            }
    }
    float* __restrict__ p = part + ((size_t)(s * gridDim.x + nb) * (MT * 16)) * LR_NB;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int row = t * 16 + (lane & 15);
        *reinterpret_cast<float4*>(p + (size_t)row * LR_NB + wave * 16 + 4 * (lane >> 4)) =
            make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
}

__global__ void __launch_bounds__(256)
linear_rows_reduce_kernel(const float* __restrict__ part, int S, int nblk, int rows_pad, int M, int N,
                          const float* __restrict__ bias, const float* __restrict__ bias2, int N1, int relu,
                          float* __restrict__ y, int ldy) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;          // (row, neuron)
    if (e >= M * N) return;
    const int m = e / N, n = e - m * N;
    const int nb = n / LR_NB, nl = n - nb * LR_NB;
    const float* __restrict__ p = part + ((size_t)nb * rows_pad + m) * LR_NB + nl;
    const size_t stride = (size_t)nblk * rows_pad * LR_NB;         // between K slices
    float v = 0.0f;
    for (int s0 = 0; s0 < S; s0 += 32) {                           // 32 independent loads (one round trip), ordered adds
        float t[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) t[q] = p[(size_t)min(s0 + q, S - 1) * stride];     // unconditional (clamped): see above
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 32; ++q)
            if (s0 + q < S) v = add_rn(v, t[q]);
    }
    const float* __restrict__ bsel = (bias2 != nullptr && n >= N1) ? bias2 + (n - N1) : (bias != nullptr ? bias + n : nullptr);
    if (bsel != nullptr) v = add_rn(v, *bsel);
    if (relu) v = relu_nan(v);
    y[(size_t)m * ldy + n] = v;
}

// Launch geometry shared by the workspace query and the launcher.
struct LinearRowsPlan {
    int mt, nblk, S, kslice;
    size_t part_floats;
};
inline LinearRowsPlan linear_rows_plan(int M, int K, int N) {
    LinearRowsPlan P;
    P.mt = (M + 15) / 16;
    P.nblk = (N + LR_NB - 1) / LR_NB;
    const int ksteps = (K + LR_KS - 1) / LR_KS;
    int S = (512 + P.nblk - 1) / P.nblk;            // about two workgroups per CU
    if (S > ksteps) S = ksteps;
    if (S < 1) S = 1;
    P.kslice = ((ksteps + S - 1) / S) * LR_KS;
    P.S = (K + P.kslice - 1) / P.kslice;
    P.part_floats = (size_t)P.S * P.nblk * (P.mt * 16) * LR_NB;
    return P;
}

// W2 / bias2 / N2: a second layer on the same input whose N2 outputs follow the first layer's N1 = N columns of y
int launch_linear_rows2(const float* x, int M, int K, const float* W, const float* bias, int N1, const float* W2,
                        const float* bias2, int N2, int relu, float* ws, float* y, int ldy, hipStream_t st) {
    const int N = N1 + N2;
    const LinearRowsPlan P = linear_rows_plan(M, K, N);
    dim3 grid(P.nblk, P.S);
    switch (P.mt) {
        case 1: hipLaunchKernelGGL(linear_rows_partial_kernel<1>, grid, dim3(256), 0, st, x, M, K, W, N, W2, N1, P.kslice, ws); break;
        case 2: hipLaunchKernelGGL(linear_rows_partial_kernel<2>, grid, dim3(256), 0, st, x, M, K, W, N, W2, N1, P.kslice, ws); break;
        case 3: hipLaunchKernelGGL(linear_rows_partial_kernel<3>, grid, dim3(256), 0, st, x, M, K, W, N, W2, N1, P.kslice, ws); break;
        case 4: hipLaunchKernelGGL(linear_rows_partial_kernel<4>, grid, dim3(256), 0, st, x, M, K, W, N, W2, N1, P.kslice, ws); break;
        case 5: hipLaunchKernelGGL(linear_rows_partial_kernel<5>, grid, dim3(256), 0, st, x, M, K, W, N, W2, N1, P.kslice, ws); break;
        case 6: hipLaunchKernelGGL(linear_rows_partial_kernel<6>, grid, dim3(256), 0, st, x, M, K, W, N, W2, N1, P.kslice, ws); break;
        case 7: hipLaunchKernelGGL(linear_rows_partial_kernel<7>, grid, dim3(256), 0, st, x, M, K, W, N, W2, N1, P.kslice, ws); break;
        default: hipLaunchKernelGGL(linear_rows_partial_kernel<8>, grid, dim3(256), 0, st, x, M, K, W, N, W2, N1, P.kslice, ws); break;
    }
    // y == nullptr: the consumer adds the K slices itself while it loads (linear_rows_layout tells it where they are;
    // box_refine_post_kernel does for cls_score | bbox_pred: one launch fewer on a chain of 3-4 us kernels)
    if (y != nullptr)
        hipLaunchKernelGGL(linear_rows_reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, st, (const float*)ws, P.S,
                           P.nblk, P.mt * 16, M, N, bias, bias2, N1, relu, y, ldy);
    return check_launch("linear_rows");
}

// Two layers back to back without the first one's reduction launch: layer A (x [M,K] -> NA, bias/ReLU applied by the
// consumer) leaves its K-slice sums in ws_a, layer B (NA -> N1 | N2, two weight matrices side by side) reads them
// (linear_rows_partial_xpart_kernel) and leaves ITS K-slice sums in ws_b for a consumer that adds them while loading
// (box_refine_post_kernel).  Applies when layer B's K slice is one 64-k step; returns 0 when it does not (nothing is
// launched: the caller uses the launch pairs).
int launch_linear_rows_chain(const float* x, int M, int K, const float* WA, const float* bA, int NA, int reluA, float* ws_a,
                             const float* WB, int N1, const float* WB2, int N2, float* ws_b, hipStream_t st, int* rc) {
    const int NB = N1 + N2;
    const LinearRowsPlan PA = linear_rows_plan(M, K, NA), PB = linear_rows_plan(M, NA, NB);
    *rc = SMOT_OK;
    if (!(PB.kslice == LR_KS && (NA % LR_NB) == 0 && PB.S == PA.nblk)) return 0;
    dim3 ga(PA.nblk, PA.S), gb(PB.nblk, PB.S);
    switch (PA.mt) {
        case 1: hipLaunchKernelGGL(linear_rows_partial_kernel<1>, ga, dim3(256), 0, st, x, M, K, WA, NA, (const float*)nullptr, NA, PA.kslice, ws_a); break;
        case 2: hipLaunchKernelGGL(linear_rows_partial_kernel<2>, ga, dim3(256), 0, st, x, M, K, WA, NA, (const float*)nullptr, NA, PA.kslice, ws_a); break;
        case 3: hipLaunchKernelGGL(linear_rows_partial_kernel<3>, ga, dim3(256), 0, st, x, M, K, WA, NA, (const float*)nullptr, NA, PA.kslice, ws_a); break;
        case 4: hipLaunchKernelGGL(linear_rows_partial_kernel<4>, ga, dim3(256), 0, st, x, M, K, WA, NA, (const float*)nullptr, NA, PA.kslice, ws_a); break;
        case 5: hipLaunchKernelGGL(linear_rows_partial_kernel<5>, ga, dim3(256), 0, st, x, M, K, WA, NA, (const float*)nullptr, NA, PA.kslice, ws_a); break;
        case 6: hipLaunchKernelGGL(linear_rows_partial_kernel<6>, ga, dim3(256), 0, st, x, M, K, WA, NA, (const float*)nullptr, NA, PA.kslice, ws_a); break;
        case 7: hipLaunchKernelGGL(linear_rows_partial_kernel<7>, ga, dim3(256), 0, st, x, M, K, WA, NA, (const float*)nullptr, NA, PA.kslice, ws_a); break;
        default: hipLaunchKernelGGL(linear_rows_partial_kernel<8>, ga, dim3(256), 0, st, x, M, K, WA, NA, (const float*)nullptr, NA, PA.kslice, ws_a); break;
    }
    switch (PA.mt) {
        case 1: hipLaunchKernelGGL(linear_rows_partial_xpart_kernel<1>, gb, dim3(256), 0, st, (const float*)ws_a, PA.S, PA.nblk, bA, reluA, M, NA, WB, NB, WB2, N1, ws_b); break;
        case 2: hipLaunchKernelGGL(linear_rows_partial_xpart_kernel<2>, gb, dim3(256), 0, st, (const float*)ws_a, PA.S, PA.nblk, bA, reluA, M, NA, WB, NB, WB2, N1, ws_b); break;
        case 3: hipLaunchKernelGGL(linear_rows_partial_xpart_kernel<3>, gb, dim3(256), 0, st, (const float*)ws_a, PA.S, PA.nblk, bA, reluA, M, NA, WB, NB, WB2, N1, ws_b); break;
        case 4: hipLaunchKernelGGL(linear_rows_partial_xpart_kernel<4>, gb, dim3(256), 0, st, (const float*)ws_a, PA.S, PA.nblk, bA, reluA, M, NA, WB, NB, WB2, N1, ws_b); break;
        case 5: hipLaunchKernelGGL(linear_rows_partial_xpart_kernel<5>, gb, dim3(256), 0, st, (const float*)ws_a, PA.S, PA.nblk, bA, reluA, M, NA, WB, NB, WB2, N1, ws_b); break;
        case 6: hipLaunchKernelGGL(linear_rows_partial_xpart_kernel<6>, gb, dim3(256), 0, st, (const float*)ws_a, PA.S, PA.nblk, bA, reluA, M, NA, WB, NB, WB2, N1, ws_b); break;
        case 7: hipLaunchKernelGGL(linear_rows_partial_xpart_kernel<7>, gb, dim3(256), 0, st, (const float*)ws_a, PA.S, PA.nblk, bA, reluA, M, NA, WB, NB, WB2, N1, ws_b); break;
        default: hipLaunchKernelGGL(linear_rows_partial_xpart_kernel<8>, gb, dim3(256), 0, st, (const float*)ws_a, PA.S, PA.nblk, bA, reluA, M, NA, WB, NB, WB2, N1, ws_b); break;
    }
    *rc = check_launch("linear_rows_chain");
    return 1;
}

// Where launch_linear_rows2(..., y = nullptr) leaves the partial sums: element (row m, neuron n) of K slice s sits at
// ws[(s * nblk + n / 64) * rows_pad * 64 + m * 64 + n % 64]; the layer's value is their sum IN SLICE ORDER, then + bias.
void linear_rows_layout(int M, int K, int N, int* S, int* nblk, int* rows_pad) {
    const LinearRowsPlan P = linear_rows_plan(M, K, N);
    *S = P.S;
    *nblk = P.nblk;
    *rows_pad = P.mt * 16;
}

int launch_linear_rows(const float* x, int M, int K, const float* W, const float* bias, int N, int relu, float* ws,
                       float* y, int ldy, hipStream_t st) {
    return launch_linear_rows2(x, M, K, W, bias, N, nullptr, nullptr, 0, relu, ws, y, ldy, st);
}

}  // namespace smot

extern "C" int smot_linear_rows_max_rows(void) { return 128; }

extern "C" long long smot_linear_rows_ws_floats(int M, int K, int N) {
    if (M <= 0 || M > 128 || K <= 0 || N <= 0) return 0;
    return (long long)smot::linear_rows_plan(M, K, N).part_floats;
}

extern "C" int smot_linear_rows_fwd(const float* x, int M, int K, const float* W, const float* bias, int N, int relu,
                                    float* ws, float* y, int ldy, smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(M >= 0 && M <= 128 && K > 0 && N > 0 && ldy >= N, "linear_rows: bad sizes M=%d K=%d N=%d ldy=%d", M, K, N, ldy);
    if ((K & 3) != 0) {
        set_error("linear_rows: K=%d is not a multiple of 4 (use the library GEMM)", K);
        return SMOT_ERR_UNSUPPORTED;
    }
    if (M == 0) return SMOT_OK;
    SMOT_REQUIRE(x && W && ws && y, "linear_rows: null pointer");
    SMOT_REQUIRE((((uintptr_t)x | (uintptr_t)W | (uintptr_t)ws) & 15) == 0, "linear_rows: x, W and ws must be 16-byte aligned");
    return launch_linear_rows(x, M, K, W, bias, N, relu, ws, y, ldy, (hipStream_t)stream);
}
