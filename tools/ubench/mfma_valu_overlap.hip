// Microbenchmark: do fp32 VALU instructions overlap with v_mfma_f32_16x16x4_f32 on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap
// One iteration = 16 independent MFMAs (the tower kernel's k-step) with V extra instructions of one kind interleaved
// (V/16 after every MFMA, or all V after the 16 MFMAs).  KIND: 0 v_add_f32 (fp32 VALU), 1 v_add_u32 (integer VALU),
// 2 ds_read_b32 (LDS), 3 v_mov_b32.  Grid = 256 CUs x waves/SIMD.  If a kind overlaps with the matrix pipe, the time
// stays at the MFMA-only time until the extra work alone would take longer.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 1024;

template <int KIND, int V, bool INTERLEAVE, bool MFMA>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = a;
    lds[threadIdx.x + 256] = b;
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x = a + threadIdx.x * 1e-6f, y = b;
    float v[8];
    int u[8];
    for (int i = 0; i < 8; ++i) {
        v[i] = x + i;
        u[i] = threadIdx.x + i;
    }
    const int per = V / 16;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MFMA) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
            if (INTERLEAVE) {
#pragma unroll
                for (int e = 0; e < per; ++e) {
                    const int r = (i * per + e) & 7;
                    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(y));
                    if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
                    if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v[r]) : "v"(u[r] & 1020));
                    if (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(y));
                }
            }
        }
        if (!INTERLEAVE) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int r = e & 7;
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(y));
                if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
                if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v[r]) : "v"(u[r] & 1020));
                if (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(y));
            }
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
float run(K kern, int blocks, float* d) {
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
    return best;
}

template <int KIND, int V>
void sweep(const char* kind, float* d) {
    for (int blocks : {256, 512}) {
        const float m = run(k<KIND, 0, false, true>, blocks, d);
        const float vo = run(k<KIND, V, false, false>, blocks, d);
        const float il = run(k<KIND, V, true, true>, blocks, d);
        const float sq = run(k<KIND, V, false, true>, blocks, d);
        printf("{\"kind\": \"%s\", \"extra_per_16_mfma\": %d, \"waves_per_simd\": %d, \"us_mfma_only\": %.1f, "
               "\"us_extra_only\": %.1f, \"us_interleaved\": %.1f, \"us_mfma_then_extra\": %.1f, "
               "\"overlap_frac_interleaved\": %.2f}\n",
               kind, V, blocks / 256, m * 1e3, vo * 1e3, il * 1e3, sq * 1e3, (m + vo - il) / (vo < m ? vo : m));
    }
}

int main() {
    float* d;
    (void)hipMalloc(&d, 2048 * 256 * sizeof(float));
    sweep<0, 16>("v_add_f32", d);
    sweep<0, 32>("v_add_f32", d);
    sweep<0, 64>("v_add_f32", d);
    sweep<1, 32>("v_add_u32", d);
    sweep<1, 64>("v_add_u32", d);
    sweep<3, 32>("v_mov_b32", d);
    sweep<2, 16>("ds_read_b32", d);
    sweep<2, 32>("ds_read_b32", d);
    return 0;
}
