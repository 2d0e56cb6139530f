// Frame pre-processing on the GPU — SURVEY.md §8(f) rank 3.
//
// Replaces the per-frame CPU chain of the reference's inference entry points (demos/demo_inference.py:74-82;
// siammot/data/adapters/augmentation/build_augmentation.py:52-66, image_augmentation.py:21-50):
//   PIL.Image.resize((ow, oh), BILINEAR)  ->  ToTensor (uint8 HWC -> float CHW / 255)
//   ->  Normalize ([BGR, * 255], - mean, / std)  [UPSTREAM maskrcnn_benchmark data/transforms/transforms.py]
// in one launch that reads the uint8 RGB frame (2.8 MB at 720p instead of a 10.8 MB fp32 host-to-device copy)
// and writes the fp32 CHW network input.
//
// Bit-exact with Pillow's 8-bit resampler [THIRD PARTY: src/libImaging/Resample.c]: separable, horizontal pass
// first, triangle filter widened by the down-scaling factor, coefficients quantised to 22 fractional bits
// (the host layer computes them in double precision exactly as Pillow does and passes the integer tables),
// integer accumulation started at one half, clip to [0,255], uint8 intermediate between the passes.  An axis
// that keeps its size gets identity tables (one tap of weight 2^22), which reproduces "pass skipped".
// The float tail follows torch op by op (separately rounded / 255, * 255, - mean, / std).
//
// HBM-bound byte work (720p: 2.76 MB in, 10.8 MB out).  Workgroup = 8 output rows x 128 output columns x 3
// channels; the horizontally resampled rows the tile's vertical taps need are built once in LDS (uint8), then
// every thread resolves 4 output pixels; output stores are coalesced along x.
#include "smot_common.h"

namespace smot {

constexpr int PP_BITS = 22;
constexpr int PP_TH = 8, PP_TW = 128;

struct PreNorm {
    float mean[3], std[3];
    int to_bgr255;
};

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> PP_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ void __launch_bounds__(256)
preprocess_kernel(const unsigned char* __restrict__ frame, int H, int W, const int* __restrict__ xb,
                  const int* __restrict__ xk, int kx, const int* __restrict__ yb, const int* __restrict__ yk, int ky,
                  int OH, int OW, int max_rows, PreNorm Nn, float* __restrict__ out) {
    extern __shared__ unsigned char inter[];          // [rows][PP_TW * 3]
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * PP_TW, y0 = blockIdx.y * PP_TH;
    const int ylast = min(y0 + PP_TH, OH) - 1;
    const int r0 = yb[2 * y0];
    const int r1 = yb[2 * ylast] + yb[2 * ylast + 1];  // bounds are monotonic in y
    const int rows = min(r1 - r0, max_rows);
    const int tw = min(PP_TW, OW - x0);
    // pass 1: horizontal resampling of input rows r0..r1 for the tile's columns
    for (int e = tid; e < rows * tw * 3; e += 256) {
        const int r = e / (tw * 3);
        const int xc = e - r * (tw * 3);
        const int xl = xc / 3, c = xc - xl * 3;
        const int x = x0 + xl;
        const int first = xb[2 * x], cnt = xb[2 * x + 1];
        const unsigned char* __restrict__ src = frame + ((size_t)(r0 + r) * W + first) * 3 + c;
        const int* __restrict__ k = xk + (size_t)x * kx;
        int acc = 1 << (PP_BITS - 1);
        for (int t = 0; t < cnt; ++t) acc += (int)src[t * 3] * k[t];
        inter[r * (PP_TW * 3) + xc] = (unsigned char)clip8(acc);
    }
    __syncthreads();
    // pass 2: vertical resampling + ToTensor + Normalize
#pragma unroll
    for (int j = 0; j < PP_TH * PP_TW / 256; ++j) {
        const int o = tid + 256 * j;
        const int yl = o / PP_TW, xl = o - yl * PP_TW;
        const int y = y0 + yl, x = x0 + xl;
        if (y >= OH || x >= OW) continue;
        const int first = yb[2 * y] - r0, cnt = yb[2 * y + 1];
        const int* __restrict__ k = yk + (size_t)y * ky;
        int acc[3] = {1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1)};
        for (int t = 0; t < cnt; ++t) {
            const unsigned char* p = inter + (first + t) * (PP_TW * 3) + xl * 3;
            const int w = k[t];
            acc[0] += (int)p[0] * w;
            acc[1] += (int)p[1] * w;
            acc[2] += (int)p[2] * w;
        }
#pragma unroll
        for (int oc = 0; oc < 3; ++oc) {
            const int c = Nn.to_bgr255 ? 2 - oc : oc;
            float f = div_rn((float)clip8(acc[c]), 255.0f);
            if (Nn.to_bgr255) f = mul_rn(f, 255.0f);
            f = div_rn(sub_rn(f, Nn.mean[oc]), Nn.std[oc]);
            out[((size_t)oc * OH + y) * OW + x] = f;
        }
    }
}

}  // namespace smot

extern "C" int smot_preprocess_fwd(const unsigned char* frame, int H, int W, const int* xbounds, const int* xcoeffs,
                                   int kx, const int* ybounds, const int* ycoeffs, int ky, int OH, int OW,
                                   int max_tile_rows, const float* mean3, const float* std3, int to_bgr255, float* out,
                                   smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(H > 0 && W > 0 && OH > 0 && OW > 0 && kx > 0 && ky > 0, "preprocess: bad sizes %dx%d -> %dx%d", H, W, OH,
                 OW);
    SMOT_REQUIRE(frame && xbounds && xcoeffs && ybounds && ycoeffs && mean3 && std3 && out, "preprocess: null pointer");
    SMOT_REQUIRE(max_tile_rows > 0, "preprocess: max_tile_rows=%d", max_tile_rows);
    const size_t smem = (size_t)max_tile_rows * PP_TW * 3;
    if (smem > 64 * 1024) {
        set_error("preprocess: %d input rows per 8-row tile need %zu bytes of LDS (down-scaling factor too large)",
                  max_tile_rows, smem);
        return SMOT_ERR_UNSUPPORTED;
    }
    PreNorm Nn;
    for (int i = 0; i < 3; ++i) {
        Nn.mean[i] = mean3[i];       // HOST arrays
        Nn.std[i] = std3[i];
    }
    Nn.to_bgr255 = to_bgr255;
    dim3 grid((OW + PP_TW - 1) / PP_TW, (OH + PP_TH - 1) / PP_TH);
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), smem, (hipStream_t)stream, frame, H, W, xbounds, xcoeffs, kx,
                       ybounds, ycoeffs, ky, OH, OW, max_tile_rows, Nn, out);
    return check_launch("preprocess");
}
