// K2 — depthwise (per-track x per-channel) valid cross-correlation.
//
// Replaces xcorr_depthwise (reference EMM/xcorr.py:37-46): F.conv2d with groups = N*C and
// Rz x Rz filters.  Each (track, channel) plane is an independent Rx*Rx (*) Rz*Rz sliding
// correlation -> Ho*Ho; there is no reuse across planes, so this is VALU/HBM work, not MFMA work
// (arithmetic intensity 2*Ho^2*Rz^2 / (4*(Rx^2+Rz^2+Ho^2)) = 20.9 FLOP/B at 30/15/16: on the fp32
// vector ridge of gfx950).
//
// Fast path (Ho == 16): two planes per wavefront, 4x2 output patches per lane (xcorr_patch2.h; the same FMA
// phase runs inside the fused pooling+correlation kernel of sr_xcorr.hip).  The earlier generations of this
// kernel (wave-per-plane with scalar taps, four planes per wave, packed FMA, one plane per wave, 4x4x1 matrix
// instruction) are bit-identical A/B material and live in measure/csrc/xcorr_variants.hip, which is compiled into the
// SMOT_DEBUG library only (libsmot_emm_debug.so).
#include "smot_common.h"
#include "knobs.h"
#include "xcorr_patch2.h"

namespace smot {

// Two-planes-per-wave patch kernel (DEFAULT fast path, Ho == 16).
// Measured on MI355X (tools/ubench/fma_rate.hip): one wave per SIMD issues a v_fmac_f32 only every
// ~5.6 cycles, two or more waves reach ~2.8-3.0; an SGPR multiplicand costs ~4.3 cycles whatever the
// occupancy; v_pk_fma_f32 is ~5 cycles (+13 % FLOPs at best).  So the FMA stream wants plain
// VGPR-operand v_fmac and >= 2 waves per SIMD even at 30 tracks (3840 planes): a wave takes TWO
// planes (1920 waves, ~2 per SIMD), 32 lanes each; lane (q, g) owns the 4x2 output patch rows
// 4q..4q+3, cols 2g..2g+1.  Everything else is the four-plane patch kernel: window row 4q+t is
// read once (8 x ds_read_b64, 8-byte aligned, conflict-free at row stride 36) and feeds up to four
// (output row k, template row t-k) pairs; template rows come from LDS as broadcast ds_read_b128 and
// live in VGPRs; reads of step t+1 are issued before the FMAs of step t; same FMA order per output.
template <int RX, int RZ, int MODE>
__global__ void __launch_bounds__(64, 3)       // <= 168 VGPRs: three waves per SIMD (at four the kernel spilled 5 registers)
xcorr_dw_patch2_kernel(const float* __restrict__ x, const float* __restrict__ z,
                       float* __restrict__ out, int planes) {
    constexpr int HO = RX - RZ + 1;
    static_assert(HO == 16, "patch kernel tiles a 16x16 response");
    constexpr int XS = XP2_XS, XP = XP2_XP, ZS = XP2_ZS, ZP = RZ * XP2_ZS;
    static_assert(RX * XS <= XP && 2 * 7 + RZ + 1 <= XS && RZ <= ZS, "LDS image too small");
    __shared__ __attribute__((aligned(16))) float sm[2 * XP + 2 * ZP];
    float* xs = sm;
    float* zs = sm + 2 * XP;

    const int lane = threadIdx.x;
    const int plane0 = blockIdx.x * 2;

    constexpr int NX4 = (2 * RX * RX / 4 + 63) / 64;     // 8 float4 per lane cover 2 planes
    constexpr int NZ = (2 * RZ * RZ + 63) / 64;          // 8 dwords per lane cover 2 templates
    if (MODE != 2) {
        const long long last4 = (long long)planes * (RX * RX / 4) - 1;
        const float4* __restrict__ xg4 = reinterpret_cast<const float4*>(x);
        float4 sx[NX4];
#pragma unroll
        for (int t = 0; t < NX4; ++t) {
            long long gk = (long long)plane0 * (RX * RX / 4) + lane + 64 * t;
            gk = gk < last4 ? gk : last4;                  // odd plane counts: re-read valid memory
            sx[t] = xg4[gk];
        }
        const long long lastz = (long long)planes * (RZ * RZ) - 1;
        float sz[NZ];
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            long long ge = (long long)plane0 * (RZ * RZ) + lane + 64 * t;
            ge = ge < lastz ? ge : lastz;
            sz[t] = z[ge];
        }
#pragma unroll
        for (int t = 0; t < NX4; ++t) {
            const int k = lane + 64 * t;
            if (k < 2 * RX * RX / 4) {
                const int e0 = 4 * k;
                const int pl = e0 / (RX * RX);
                const int el = e0 - pl * (RX * RX);
                const int r = el / RX;
                const int c0 = el - r * RX;
                const int o0 = pl * XP + r * XS + c0;
                const int o1 = (c0 + 2 < RX) ? o0 + 2 : o0 + XS - c0;
                *reinterpret_cast<float2*>(xs + o0) = make_float2(sx[t].x, sx[t].y);
                *reinterpret_cast<float2*>(xs + o1) = make_float2(sx[t].z, sx[t].w);
            }
        }
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            const int e = lane + 64 * t;
            if (e < 2 * RZ * RZ) {
                const int pl = e / (RZ * RZ);
                const int el = e - pl * (RZ * RZ);
                const int u = el / RZ;
                zs[pl * ZP + u * ZS + (el - u * RZ)] = sz[t];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    xcorr_patch2_compute<RX, RZ, MODE>(xs, zs, lane, out, plane0, planes);
}

// Small templates (the reference's second yaml family: 35x35 search region, 7x7 template -> 29x29): one workgroup
// per plane, thread = (output row, group of four output columns).  The template's RZ*RZ taps live in registers; a
// window row is read once per thread as 4 + RZ - 1 contiguous floats (LDS rows padded to a multiple of four floats,
// zero filled) and feeds 4 * RZ FMAs — 0.4 LDS reads per FMA instead of the generic kernel's two.  Taps accumulate
// u-major / v-minor in one fmaf chain per output: the generic kernel's (and the oracle's) order, bit for bit.
template <int RX, int RZ>
__global__ void __launch_bounds__(256)
xcorr_dw_rowpatch_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ out) {
    constexpr int HO = RX - RZ + 1;
    constexpr int NQ = (HO + 3) / 4;                     // column groups per output row
    constexpr int XS = ((4 * NQ + RZ - 1 + 3) / 4) * 4;  // padded LDS row (floats)
    constexpr int SEG = 4 + RZ - 1;                      // floats a thread needs of a window row
    static_assert(HO * NQ <= 256 && XS >= RX, "one workgroup of 256 threads per plane");
    __shared__ __attribute__((aligned(16))) float xs[RX * XS];
    __shared__ float zs[RZ * RZ];
    const size_t plane = blockIdx.x;
    const float* xg = x + plane * RX * RX;
    const float* zg = z + plane * RZ * RZ;
    for (int e = threadIdx.x; e < RX * XS; e += 256) {
        const int r = e / XS, c = e - r * XS;
        xs[e] = (c < RX) ? xg[r * RX + c] : 0.0f;
    }
    for (int e = threadIdx.x; e < RZ * RZ; e += 256) zs[e] = zg[e];
    __syncthreads();
    const int i = threadIdx.x / NQ, jq = threadIdx.x - i * NQ;
    if (i >= HO) return;
    float tap[RZ * RZ];
#pragma unroll
    for (int t = 0; t < RZ * RZ; ++t) tap[t] = zs[t];
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < RZ; ++u) {
        const float* row = xs + (i + u) * XS + 4 * jq;
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
    float* dst = out + plane * HO * HO + i * HO + 4 * jq;
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (4 * jq + o < HO) dst[o] = acc[o];
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
    timer_mark(0, 0, st);
    if (Rx == 30 && Rz == 15) {
        const dim3 g2((planes + 1) / 2), b64(64);
#ifdef SMOT_DEBUG
        // measurement library only: older generations (A/B) and the phase ablations of the default kernel
        const int var = knobs().xcorr_variant;
        if (var == XV_FILL) {                // default kernel without FMAs (wrong results)
            hipLaunchKernelGGL((xcorr_dw_patch2_kernel<30, 15, 1>), g2, b64, 0, st, x, z, out, planes);
        } else if (var == XV_COMPUTE) {      // default kernel without global loads (wrong results)
            hipLaunchKernelGGL((xcorr_dw_patch2_kernel<30, 15, 2>), g2, b64, 0, st, x, z, out, planes);
        } else if (!launch_xcorr_variant(var, x, z, out, planes, st))
#endif
            SMOT_LAUNCH((xcorr_dw_patch2_kernel<30, 15, 0>), g2, b64, 0, st, x, z, out, planes);
    } else if (Rx == 35 && Rz == 7) {
        SMOT_LAUNCH((xcorr_dw_rowpatch_kernel<35, 7>), dim3(planes), dim3(256), 0, st, x, z, out);
    } else {
        const size_t smem = (size_t)(Rx * Rx + Rz * Rz) * sizeof(float);
        SMOT_REQUIRE(smem <= 160 * 1024, "xcorr: plane too large for LDS (Rx=%d Rz=%d)", Rx, Rz);
        SMOT_LAUNCH(xcorr_dw_generic_kernel, dim3(planes), dim3(256), smem, st, x, z, out, Rx, Rz);
    }
    timer_mark(0, 1, st);
    return check_launch("xcorr_dw");
}
