// K2 — depthwise (per-track x per-channel) valid cross-correlation.
//
// Replaces xcorr_depthwise (reference EMM/xcorr.py:37-46): F.conv2d with groups = N*C and
// Rz x Rz filters.  Each (track, channel) plane is an independent Rx*Rx (*) Rz*Rz sliding
// correlation -> Ho*Ho; there is no reuse across planes, so this is VALU/HBM work, not MFMA work
// (arithmetic intensity 2*Ho^2*Rz^2 / (4*(Rx^2+Rz^2+Ho^2)) = 20.9 FLOP/B at 30/15/16: on the fp32
// vector ridge of gfx950).
//
// Fast path (Ho == 16): ONE WAVEFRONT PER PLANE.
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

// Any (Rx, Rz): one workgroup per plane, plane and template in LDS, one thread per output.
__global__ void __launch_bounds__(256)
xcorr_dw_generic_kernel(const float* __restrict__ x, const float* __restrict__ z,
                        float* __restrict__ out, int RX, int RZ) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;
    float* zs = sm + RX * RX;
    const int HO = RX - RZ + 1;
    const size_t plane = blockIdx.x;
    const float* xg = x + plane * RX * RX;
    const float* zg = z + plane * RZ * RZ;
    for (int e = threadIdx.x; e < RX * RX; e += blockDim.x) xs[e] = xg[e];
    for (int e = threadIdx.x; e < RZ * RZ; e += blockDim.x) zs[e] = zg[e];
    __syncthreads();
    for (int o = threadIdx.x; o < HO * HO; o += blockDim.x) {
        const int i = o / HO, j = o - i * HO;
        float acc = 0.0f;
        for (int u = 0; u < RZ; ++u)
            for (int v = 0; v < RZ; ++v) acc = fmaf(xs[(i + u) * RX + j + v], zs[u * RZ + v], acc);
        out[plane * HO * HO + o] = acc;
    }
}

}  // namespace smot

extern "C" int smot_xcorr_dw_fwd(const float* x, const float* z, float* out, int N, int C, int Rx, int Rz,
                                 smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(N >= 0 && C > 0 && Rz > 0 && Rx >= Rz, "xcorr: bad sizes N=%d C=%d Rx=%d Rz=%d", N, C, Rx, Rz);
    if (N == 0) return SMOT_OK;
    SMOT_REQUIRE(x && z && out, "xcorr: null pointer");
    SMOT_REQUIRE((long long)N * C < (1ll << 31), "xcorr: N*C too large");
    const int planes = N * C;
    hipStream_t st = (hipStream_t)stream;
    if (Rx == 30 && Rz == 15) {
        hipLaunchKernelGGL((xcorr_dw_wave_kernel<30, 15>), dim3((planes + 3) / 4), dim3(256), 0, st, x, z, out,
                           planes);
    } else {
        const size_t smem = (size_t)(Rx * Rx + Rz * Rz) * sizeof(float);
        SMOT_REQUIRE(smem <= 160 * 1024, "xcorr: plane too large for LDS (Rx=%d Rz=%d)", Rx, Rz);
        hipLaunchKernelGGL(xcorr_dw_generic_kernel, dim3(planes), dim3(256), smem, st, x, z, out, Rx, Rz);
    }
    return check_launch("xcorr_dw");
}
