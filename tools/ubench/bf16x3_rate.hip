// Microbenchmark ("measure - not argue - a 3-way bf16 split of the transformed operands"):
// one K = 32 block of the Winograd towers' GEMM (16 x 16 output tile, 32 input channels) as
//   F32   eight v_mfma_f32_16x16x4_f32 on the lane's eight fp32 B operands (the tower kernel's form), against
//   BF16  the same eight fp32 B operands split on the fly into three bf16 parts each (b = b1 + b2 + b3, truncation split:
//         every part exact, the dropped tail < 2^-24 relative) and SIX v_mfma_f32_16x16x32_bf16 against a pre-split A
//         (a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1; fp32 accumulate; the A parts are weights: split once off-line),
//   SPLIT the split alone (no MFMA), MF6 the six bf16 MFMAs alone (operands ready).
// The B operands change every iteration (they are the transformed activations: built in registers per k-step in the
// tower kernel), so the split is per-block work.  Grid = 256 CUs x 2 waves per SIMD (the tower kernel's occupancy).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/bf16x3_rate.hip -o tools/ubench/bf16x3_rate && tools/ubench/bf16x3_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048;

__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) {      // two bf16 (upper halves of a, b) in one dword
    return (a >> 16) | (b & 0xffff0000u);
}

// mode 0 F32, 1 BF16 (split + 6 MFMAs), 2 SPLIT only, 3 MF6 only, 4 F32 on TWO output tiles (16 MFMAs per operand set: the
// tower kernel's OCT = 2), 5 BF16 on two output tiles (one split, 12 MFMAs), 6 as 5 with the v_cvt_pk_bf16_f32 split,
// 7 the K = 16 instruction: 24 x v_mfma_f32_16x16x16_bf16 only
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, float a, float b) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bv[8];
    for (int i = 0; i < 8; ++i) bv[i] = a + threadIdx.x * 1e-3f + i;
    const float av = b + threadIdx.x * 1e-4f;
    u32x4 a1 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, a2 = a1, a3 = a1;
    unsigned sink = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {                // four independent accumulator tiles per iteration (as the kernel's N-tiles)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(bv[i]) : "v"(av));   // "new" operands
            if (MODE == 0 || MODE == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[i], acc[t], 0, 0, 0);
                    if (MODE == 4) acc[t + 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[i], acc[t + 4], 0, 0, 0);
                }
            } else if (MODE == 7) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                const s16x4 q4 = {(short)(0x3f80 + threadIdx.x), 0x3f80, 0x3f80, 0x3f80};
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(q4, q4, acc[t], 0, 0, 0);
                    acc[t + 4] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(q4, q4, acc[t + 4], 0, 0, 0);
                }
            } else {
                u32x4 p1, p2, p3;
                if (MODE == 6) {
                    // hardware conversion: two fp32 -> packed bf16 pair (round to nearest even); the residuals stay exact
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned ph, pm, pl;
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ph) : "v"(bv[2 * j]), "v"(bv[2 * j + 1]));
                        const float r0 = bv[2 * j] - __uint_as_float(ph << 16), r1 = bv[2 * j + 1] - __uint_as_float(ph & 0xffff0000u);
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pm) : "v"(r0), "v"(r1));
                        const float s0 = r0 - __uint_as_float(pm << 16), s1 = r1 - __uint_as_float(pm & 0xffff0000u);
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pl) : "v"(s0), "v"(s1));
                        p1[j] = ph;
                        p2[j] = pm;
                        p3[j] = pl;
                    }
                } else if (MODE != 3) {
                    unsigned h[8], m[8], l[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const unsigned u = __float_as_uint(bv[i]);
                        h[i] = u & 0xffff0000u;
                        const float r1 = bv[i] - __uint_as_float(h[i]);          // exact
                        m[i] = __float_as_uint(r1) & 0xffff0000u;
                        const float r2 = r1 - __uint_as_float(m[i]);             // exact
                        l[i] = __float_as_uint(r2);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        p1[j] = pack_hi(h[2 * j], h[2 * j + 1]);
                        p2[j] = pack_hi(m[2 * j], m[2 * j + 1]);
                        p3[j] = pack_hi(l[2 * j], l[2 * j + 1]);
                    }
                } else {
                    p1 = p2 = p3 = a1;
                }
                if (MODE == 2) {
                    sink ^= p1[0] ^ p2[1] ^ p3[2] ^ p1[3] ^ p2[0] ^ p3[1] ^ p1[2] ^ p2[3] ^ p3[0] ^ p1[1] ^ p2[2] ^ p3[3];
                } else {
                    const bf16x8 B1 = __builtin_bit_cast(bf16x8, p1), B2 = __builtin_bit_cast(bf16x8, p2), B3 = __builtin_bit_cast(bf16x8, p3);
                    const bf16x8 A1 = __builtin_bit_cast(bf16x8, a1), A2 = __builtin_bit_cast(bf16x8, a2), A3 = __builtin_bit_cast(bf16x8, a3);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B3, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, B1, acc[t], 0, 0, 0);
                    if (MODE == 5 || MODE == 6) {          // the second output-channel tile: same B parts, other A parts
                        acc[t + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, acc[t + 4], 0, 0, 0);
                        acc[t + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B2, acc[t + 4], 0, 0, 0);
                        acc[t + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, B1, acc[t + 4], 0, 0, 0);
                        acc[t + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B3, acc[t + 4], 0, 0, 0);
                        acc[t + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, B2, acc[t + 4], 0, 0, 0);
                        acc[t + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, acc[t + 4], 0, 0, 0);
                    }
                }
            }
        }
    }
    float s = (float)sink;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += bv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, 1.0001f, 0.9999f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, 1.0001f, 0.9999f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // K = 32 blocks per SIMD: 2 waves x ITER x 4 tiles; cycles per block per SIMD at the 2.1 GHz the tower kernel runs at
    const double blocks_per_simd = 2.0 * ITER * 4;
    printf("{\"form\": \"%s\", \"ms\": %.4f, \"ns_per_k32_block_per_simd\": %.2f, \"cycles_at_2.1GHz\": %.1f}\n", name, best,
           best * 1e6 / blocks_per_simd, best * 1e-3 * 2.1e9 / blocks_per_simd);
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    run<0>("F32: 8 x v_mfma_f32_16x16x4_f32", d);
    run<1>("BF16x3: split 8 operands + 6 x v_mfma_f32_16x16x32_bf16", d);
    run<2>("SPLIT only", d);
    run<3>("6 x v_mfma_f32_16x16x32_bf16 only", d);
    run<4>("F32, two output tiles per operand set: 16 x v_mfma_f32_16x16x4_f32", d);
    run<5>("BF16x3, two output tiles: one split + 12 x v_mfma_f32_16x16x32_bf16", d);
    run<6>("BF16x3, two output tiles, v_cvt_pk_bf16_f32 split + 12 MFMAs", d);
    run<7>("24 x v_mfma_f32_16x16x16_bf16 only (K = 16 form of the 12)", d);
    return 0;
}
