// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for THIS repository's access patterns
// (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming
// read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Every mode is its own kernel name, launched once over a buffer larger than the 256 MiB
// Infinity Cache; the program prints the bytes each mode is KNOWN to touch.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o f -- tools/ubench/fetch_calib     (and again: WRITE_SIZE)
// Modes:  read16  16 B per lane, coalesced stream          read4   4 B per lane, coalesced stream (256 B per wave)
//         window  what the pooling kernels do: 66 rows of 34 floats at an unaligned column of a 176 x 320 plane,
//                 one plane per wave, 4 B per lane (lanes 34..63 idle)
//         write4 / write16  4 / 16 B per lane coalesced stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr size_t BYTES = (size_t)512 << 20;
constexpr int PH = 176, PW = 320, WIN_H = 66, WIN_W = 34;

__global__ void __launch_bounds__(256) read16(const float4* __restrict__ src, size_t n4, float* sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc += (v.x + v.y) + (v.z + v.w);
    }
    if (acc == 12345.678f) *sink = acc;
}

__global__ void __launch_bounds__(256) read4(const float* __restrict__ src, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 12345.678f) *sink = acc;
}

__global__ void __launch_bounds__(64) window(const float* __restrict__ src, int planes, float* sink) {
    const int p = blockIdx.x;
    if (p >= planes) return;
    const int x0 = (p * 7) % (PW - WIN_W), y0 = (p * 5) % (PH - WIN_H);
    const float* base = src + (size_t)p * PH * PW + (size_t)y0 * PW + x0;
    float acc = 0.f;
    if (threadIdx.x < WIN_W)
        for (int r = 0; r < WIN_H; ++r) acc += base[(size_t)r * PW + threadIdx.x];
    if (acc == 12345.678f) *sink = acc;
}

__global__ void __launch_bounds__(256) write4(float* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (float)i;
}

__global__ void __launch_bounds__(256) write16(float4* __restrict__ dst, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float f = (float)i;
        dst[i] = make_float4(f, f, f, f);
    }
}

int main() {
    float *buf, *sink;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 0, BYTES);
    hipDeviceSynchronize();
    const int planes = (int)(BYTES / ((size_t)PH * PW * 4));
    // distinct 64-byte and 128-byte lines of the windowed pattern (the buffer is 256-byte aligned)
    size_t l64 = 0, l128 = 0;
    for (int p = 0; p < planes; ++p) {
        const int x0 = (p * 7) % (PW - WIN_W), y0 = (p * 5) % (PH - WIN_H);
        for (int r = 0; r < WIN_H; ++r) {
            const size_t b0 = (((size_t)p * PH + y0 + r) * PW + x0) * 4, b1 = b0 + WIN_W * 4 - 1;
            l64 += b1 / 64 - b0 / 64 + 1;
            l128 += b1 / 128 - b0 / 128 + 1;
        }
    }
    hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const float4*)buf, BYTES / 16, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(read4, dim3(4096), dim3(256), 0, 0, buf, BYTES / 4, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(window, dim3(planes), dim3(64), 0, 0, buf, planes, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(write4, dim3(4096), dim3(256), 0, 0, buf, BYTES / 4);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(write16, dim3(4096), dim3(256), 0, 0, (float4*)buf, BYTES / 16);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("{\"read16_bytes\": %zu, \"read4_bytes\": %zu, \"window_cell_bytes\": %zu, \"window_bytes_64B_lines\": %zu, "
           "\"window_bytes_128B_lines\": %zu, \"write4_bytes\": %zu, \"write16_bytes\": %zu}\n",
           BYTES, BYTES, (size_t)planes * WIN_H * WIN_W * 4, l64 * 64, l128 * 128, BYTES, BYTES);
    return 0;
}
