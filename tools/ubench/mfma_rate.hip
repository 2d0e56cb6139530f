// Microbenchmark: fp32-input MFMA issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 2048;

template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x = a + threadIdx.x * 1e-6f, y = b;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float x = a + threadIdx.x * 1e-6f, y = b;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int blocks, float* d, double flops_per_inst, int nacc) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double insts = (double)blocks * 4 * ITER * nacc;
    printf("{\"form\": \"%s\", \"acc\": %d, \"blocks\": %d, \"ms\": %.4f, \"TFLOPs\": %.1f, \"cycles_per_inst_per_SIMD_at_2.4GHz\": %.1f}\n",
           name, nacc, blocks, best, insts * flops_per_inst / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 / (insts / 1024.0));
}
int main() {
    float* d;
    (void)hipMalloc(&d, 2048 * 256 * sizeof(float));
    for (int blocks : {256, 512}) {
        run("mfma_f32_16x16x4f32", k16<1>, blocks, d, 2048, 1);
        run("mfma_f32_16x16x4f32", k16<2>, blocks, d, 2048, 2);
        run("mfma_f32_16x16x4f32", k16<4>, blocks, d, 2048, 4);
        run("mfma_f32_32x32x2f32", k32<1>, blocks, d, 4096, 1);
        run("mfma_f32_32x32x2f32", k32<2>, blocks, d, 4096, 2);
        run("mfma_f32_32x32x2f32", k32<4>, blocks, d, 4096, 4);
    }
    return 0;
}
