// Microbenchmark: fp32 FMA issue rates on gfx950 (one number per instruction form).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fma_rate.hip -o tools/ubench/fma_rate && ./fma_rate
// Each wave runs ITER iterations of 16 independent accumulator updates; blocks = 256 CUs x waves/CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int MODE>
__global__ void __launch_bounds__(256) k_fma(float* out, float a, float b) {
    float acc[16];
    v2f pacc[8];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) pacc[i] = (v2f){acc[2 * i], acc[2 * i + 1]};
    float x = a + threadIdx.x * 1e-6f, y = b;
    v2f px = {x, x + 1.0f}, py = {y, y * 0.5f};
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {          // v_fmac_f32, VGPR operands
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
        } else if (MODE == 1) {   // v_fmac_f32 with an SGPR operand
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(b), "v"(x));
        } else if (MODE == 2) {   // v_pk_fma_f32, plain pairs
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(px), "v"(py));
        } else if (MODE == 3) {   // v_pk_fma_f32 with a broadcast (op_sel_hi) operand, as hipcc emits it
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(pacc[i]) : "v"(px), "v"(py));
        } else if (MODE == 4) {   // v_pk_fma_f32 broadcasting the high half
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(pacc[i]) : "v"(px), "v"(py));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += pacc[i].x + pacc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, float* d, double flops_per_inst, int insts_per_iter) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fma<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_fma<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double waves = (double)blocks * 4;
    const double insts = waves * ITER * insts_per_iter;
    const double tf = insts * flops_per_inst / (best * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD assuming 2.4 GHz and waves spread over 1024 SIMDs
    const double cyc = best * 1e-3 * 2.4e9 / (insts / 1024.0);
    printf("{\"form\": \"%s\", \"blocks\": %d, \"ms\": %.4f, \"TFLOPs\": %.1f, \"cycles_per_inst_at_2.4GHz\": %.2f}\n", name,
           blocks, best, tf, cyc);
}

// shader clock during a kernel: s_memtime (shader cycles) against the constant-rate wall clock
__global__ void k_clock(unsigned long long* out, int iters) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(a));
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
    if (a == 12345.f) out[2] = 1;
}

static void clock_probe(const char* label, int iters) {
    unsigned long long* d;
    (void)hipMalloc(&d, 32);
    hipLaunchKernelGGL(k_clock, dim3(1024), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    int wall_khz = 0;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    const double secs = (double)h[1] / (wall_khz * 1e3);
    printf("{\"probe\": \"%s\", \"shader_cycles\": %llu, \"wall_ticks\": %llu, \"wall_khz\": %d, \"kernel_us\": %.1f, "
           "\"shader_MHz\": %.0f}\n", label, h[0], h[1], wall_khz, secs * 1e6, h[0] / secs / 1e6);
    (void)hipFree(d);
}

int main() {
    float* d;
    (void)hipMalloc(&d, 4096 * 256 * sizeof(float));
    clock_probe("cold 10us kernel", 1000);
    clock_probe("second 10us kernel", 1000);
    clock_probe("1ms kernel", 100000);
    clock_probe("10ms kernel", 1000000);
    clock_probe("10us kernel right after load", 1000);
    for (int blocks : {256, 512, 1024}) {   // 1, 2, 4, 8 waves per SIMD
        run<0>("v_fmac_f32 vgpr", blocks, d, 128, 16);
        run<1>("v_fmac_f32 sgpr", blocks, d, 128, 16);
        run<2>("v_pk_fma_f32", blocks, d, 256, 8);
        run<3>("v_pk_fma_f32 op_sel_hi bcast lo", blocks, d, 256, 8);
        run<4>("v_pk_fma_f32 op_sel bcast hi", blocks, d, 256, 8);
    }
    return 0;
}
