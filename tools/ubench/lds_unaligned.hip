// Does ds_read_b128 / ds_read_b64 work at 2-byte and 4-byte alignment on gfx950 (ROCm's default SH_MEM alignment mode), and what
// does it cost?  (Round 6: the B operand of a matrix-pipe cross-correlation is a Toeplitz window — eight consecutive fp16
// values at a lane-dependent half-word offset of a template row.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_unaligned.hip -o tools/ubench/lds_unaligned && tools/ubench/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// every lane reads 16 bytes at byte offset base + lane * stride + skew; result compared on the host
__global__ void check_kernel(unsigned* out, int skew, int stride, int width) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int e = threadIdx.x; e < 4096; e += 64) lds[e] = (unsigned short)(e * 7 + 1);
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)lds + threadIdx.x * stride + skew;
    u32x4 v = {0, 0, 0, 0};
    if (width == 16) {
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    } else {
        u32x2 w;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr) : "memory");
        v[0] = w[0];
        v[1] = w[1];
    }
    for (int q = 0; q < 4; ++q) out[threadIdx.x * 4 + q] = v[q];
}

template <int SKEW>
__global__ void __launch_bounds__(512) rate_kernel(unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int e = threadIdx.x; e < 16384; e += 512) lds[e] = (unsigned short)e;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds + (threadIdx.x & 63) * 30 * 2 + SKEW + (threadIdx.x >> 6) * 2048;   // 30-half stride: a Toeplitz-like walk
    unsigned acc = 0;
    for (int it = 0; it < 2048; ++it) {
        u32x4 a, b, c, d;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:96\n\tds_read_b128 %2, %4 offset:192\n\tds_read_b128 %3, %4 offset:288\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(base) : "memory");
        acc += a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int SKEW>
float time_rate(unsigned* d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<SKEW>, dim3(256), dim3(512), 0, 0, d);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rate_kernel<SKEW>, dim3(256), dim3(512), 0, 0, d);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e3f;
}

// rate of one read shape: WIDTH bytes per lane at byte offset lane * STRIDE + SKEW (+ 4 offsets per iteration), 8 waves per CU
template <int WIDTH, int STRIDE, int SKEW>
__global__ void __launch_bounds__(512) shape_rate_kernel(unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int e = threadIdx.x; e < 16384; e += 512) lds[e] = (unsigned short)e;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds + (threadIdx.x & 63) * STRIDE + SKEW + (threadIdx.x >> 6) * 2048;
    unsigned acc = 0;
    for (int it = 0; it < 2048; ++it) {
        if (WIDTH == 16) {
            u32x4 a, b, c, d;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:16\n\tds_read_b128 %3, %4 offset:1040\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(base) : "memory");
            acc += a[0] ^ b[1] ^ c[2] ^ d[3];
        } else if (WIDTH == 8) {
            u32x2 a, b, c, d;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:1024\n\tds_read_b64 %2, %4 offset:8\n\tds_read_b64 %3, %4 offset:1032\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(base) : "memory");
            acc += a[0] ^ b[1] ^ c[0] ^ d[1];
        } else {
            unsigned a, b, c, d;
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:1024\n\tds_read_b32 %2, %4 offset:4\n\tds_read_b32 %3, %4 offset:1028\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(base) : "memory");
            acc += a ^ b ^ c ^ d;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int WIDTH, int STRIDE, int SKEW>
float time_shape(unsigned* d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((shape_rate_kernel<WIDTH, STRIDE, SKEW>), dim3(256), dim3(512), 0, 0, d);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((shape_rate_kernel<WIDTH, STRIDE, SKEW>), dim3(256), dim3(512), 0, 0, d);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // LDS cycles per wave instruction: 8 waves x 2048 x 4 instructions per CU, 2.4 GHz
    return best * 1e-3f * 2.4e9f / (8.0f * 2048.0f * 4.0f);
}

int main() {
    unsigned* d; (void)hipMalloc(&d, 256 * 512 * 4 * 4);
    unsigned h[256];
    const int cases[][3] = {{0, 16, 16}, {4, 16, 16}, {2, 16, 16}, {6, 20, 16}, {2, 30, 16}, {1, 16, 16}, {0, 8, 8}, {4, 8, 8}, {2, 8, 8}, {2, 30, 8}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, d, c[0], c[1], c[2]);
        hipError_t e = hipDeviceSynchronize();
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int q = 0; q < c[2] / 4; ++q) {
                // expected dword = two halves at half index (l*stride + skew)/2 + 2q (for even byte offsets) — compute by bytes
                unsigned expect = 0;
                for (int byte = 0; byte < 4; ++byte) {
                    const int off = l * c[1] + c[0] + q * 4 + byte;
                    const unsigned short hv = (unsigned short)((off / 2) * 7 + 1);
                    const unsigned bv = (off & 1) ? (hv >> 8) : (hv & 0xff);
                    expect |= bv << (8 * byte);
                }
                if (h[l * 4 + q] != expect) ++bad;
            }
        printf("{\"ds_read_b%d\": {\"skew_bytes\": %d, \"lane_stride_bytes\": %d, \"hip_error\": \"%s\", \"wrong_dwords\": %d}}\n", c[2] * 8, c[0], c[1],
               hipGetErrorString(e), bad);
    }
    printf("{\"rate_us_4x_ds_read_b128_x2048_per_wave\": {\"aligned_16\": %.1f, \"skew_4\": %.1f, \"skew_2\": %.1f}}\n", time_rate<0>(d) , time_rate<4>(d), time_rate<2>(d));
    printf("{\"lds_cycles_per_wave_instruction_at_2.4GHz\": {\"b128_stride16_aligned\": %.1f, \"b128_stride16_skew8\": %.1f, \"b128_stride16_skew4\": %.1f, "
           "\"b64_stride8_aligned\": %.1f, \"b64_stride8_skew4\": %.1f, \"b64_stride8_skew2\": %.1f, \"b32_stride4\": %.1f, \"b32_stride4_skew2\": %.1f, "
           "\"b64_stride4_overlapping_aligned4\": %.1f, \"b64_stride2_overlapping\": %.1f}}\n",
           time_shape<16, 16, 0>(d), time_shape<16, 16, 8>(d), time_shape<16, 16, 4>(d), time_shape<8, 8, 0>(d), time_shape<8, 8, 4>(d),
           time_shape<8, 8, 2>(d), time_shape<4, 4, 0>(d), time_shape<4, 4, 2>(d), time_shape<8, 4, 0>(d), time_shape<8, 2, 0>(d));
    return 0;
}
