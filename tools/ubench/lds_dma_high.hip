// Does a direct-to-LDS load (buffer_load_dwordx4 ... lds, destination base in M0) reach LDS addresses above 64 KB on gfx950?
// One wave: zero 152 KB of LDS, fetch 1 KB of a known pattern to LDS byte address X, read back at X and at X mod 64 KB.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_high.hip -o /tmp/lds_dma_high && /tmp/lds_dma_high
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned* out, unsigned x) {
    extern __shared__ __attribute__((aligned(16))) unsigned sm[];
    const int lane = threadIdx.x;
    for (int e = lane; e < 152 * 256; e += 64) sm[e] = 0u;
    __syncthreads();
    const unsigned long long pa = reinterpret_cast<unsigned long long>(src);
    const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(unsigned)pa), __builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu)),
                      0x7fffffff, 0x00020000};
    typedef __attribute__((address_space(3))) unsigned lds_u;
    const unsigned base = (unsigned)(size_t)(lds_u*)sm;
    const unsigned dst = __builtin_amdgcn_readfirstlane(base + x);
    const unsigned voff = lane * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(dst), "v"(voff), "s"(rs) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = sm[x / 4 + lane * 4];
    out[64 + lane] = sm[(x & 0xffffu) / 4 + lane * 4];
    out[128 + lane] = base;
}
int main() {
    unsigned *src, *out, h[192], hs[256];
    for (int i = 0; i < 256; ++i) hs[i] = 0xabc00000u + i;
    hipMalloc(&src, 1024); hipMalloc(&out, 192 * 4);
    hipMemcpy(src, hs, 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    for (unsigned x : {0x4000u, 0xe000u, 0x10000u, 0x18000u, 0x20000u, 0x25c00u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 152 * 1024, 0, src, out, x);
        hipMemcpy(h, out, 192 * 4, hipMemcpyDeviceToHost);
        int at_x = 0, at_wrap = 0;
        for (int l = 0; l < 64; ++l) { at_x += h[l] == hs[l * 4]; at_wrap += h[64 + l] == hs[l * 4]; }
        printf("{\"lds_byte_address\": %u, \"lanes_found_at_address\": %d, \"lanes_found_at_address_mod_64k\": %d, \"lds_base\": %u}\n", x, at_x, at_wrap, h[128]);
    }
    return 0;
}
