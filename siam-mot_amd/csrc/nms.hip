// NMS — SURVEY.md §8(f) rank 2: the greedy IoU suppression behind boxlist_nms, which the solver
// (reference track_head/track_solver.py:22), the RPN post-processor (operator_patch/rpn_patch.py:53) and
// the box post-processor (box_head/inference.py:174) all call.
//
// Replaces [UPSTREAM] maskrcnn_benchmark _C.nms (csrc/cuda/nms.cu): boxes must arrive sorted by descending
// score; IoU uses the upstream "+1" convention (w = max(x2 - x1 + 1, 0)); box j is suppressed by a kept
// higher-scoring box i when IoU(i, j) > thresh (the CUDA kernel's strict '>'; upstream's CPU fallback uses
// '>=' — the two only differ at exact equality).
//
// Pass 1: the classic 64x64-tile bitmask: mask[i][cb] has bit j set when box cb*64+j (j > i in score order)
//         overlaps box i above the threshold.
// Pass 2: one workgroup resolves the greedy chain on the GPU (upstream copies the mask to the host): 64-row
//         slabs of the mask are staged in LDS; inside a slab one wave walks the 64 candidates in order,
//         OR-ing the rows of kept boxes into a `removed` bitset held one 64-bit word per lane.
// Output: keep[i] in {0,1} per box (score order).  No host synchronisation.
#include "smot_common.h"

namespace smot {

constexpr int NMS_T = 64;

__device__ __forceinline__ float iou_plus1(const float* a, const float* b) {
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float w = fmaxf(right - left + 1.0f, 0.0f), h = fmaxf(bottom - top + 1.0f, 0.0f);
    const float inter = w * h;
    const float sa = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f);
    const float sb = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
    return inter / (sa + sb - inter);
}

__global__ void __launch_bounds__(NMS_T)
nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh, unsigned long long* __restrict__ mask) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    const int nblk = gridDim.x;
    __shared__ float cbox[NMS_T * 4];
    const int ccount = min(n - cb * NMS_T, NMS_T);
    if ((int)threadIdx.x < ccount) {
        const float* src = boxes + (size_t)(cb * NMS_T + threadIdx.x) * 4;
        cbox[threadIdx.x * 4 + 0] = src[0];
        cbox[threadIdx.x * 4 + 1] = src[1];
        cbox[threadIdx.x * 4 + 2] = src[2];
        cbox[threadIdx.x * 4 + 3] = src[3];
    }
    __syncthreads();
    const int i = rb * NMS_T + threadIdx.x;
    if (i >= n) return;
    unsigned long long bits = 0ull;
    if (cb >= rb) {                                   // only later boxes can be suppressed by box i
        const float* me = boxes + (size_t)i * 4;
        const float b4[4] = {me[0], me[1], me[2], me[3]};
        const int j0 = (cb == rb) ? threadIdx.x + 1 : 0;
        for (int j = j0; j < ccount; ++j)
            if (iou_plus1(b4, cbox + j * 4) > thresh) bits |= 1ull << j;
    }
    mask[(size_t)i * nblk + cb] = bits;
}

__global__ void __launch_bounds__(256)
nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int nblk, unsigned char* __restrict__ keep) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long slab[];   // [64 rows][nblk]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // removed[w]: lane w of wave 0 holds word w, w + 64, ... (registers, up to 8 words: n <= 32768)
    unsigned long long removed[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) removed[k] = 0ull;
    for (int c = 0; c < nblk; ++c) {
        const int rows = min(NMS_T, n - c * NMS_T);
        for (int e = tid; e < rows * nblk; e += 256) slab[e] = mask[(size_t)c * NMS_T * nblk + e];
        __syncthreads();
        if (wave == 0) {
            // current word of `removed` for this slab lives in lane (c & 63), slot c >> 6
            unsigned long long cur = 0ull;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k == (c >> 6)) cur = removed[k];
            unsigned lo = __shfl((unsigned)cur, c & 63), hi = __shfl((unsigned)(cur >> 32), c & 63);
            unsigned long long r = ((unsigned long long)hi << 32) | lo;
            for (int j = 0; j < rows; ++j) {
                const bool kept = !((r >> j) & 1ull);          // wave-uniform
                if (lane == 0) keep[c * NMS_T + j] = kept ? 1 : 0;
                if (kept) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int w = lane + 64 * k;
                        if (w < nblk) removed[k] |= slab[j * nblk + w];
                    }
                    r |= slab[j * nblk + c];
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace smot

extern "C" long long smot_nms_ws_bytes(int n) {
    if (n <= 0) return 0;
    const long long nblk = (n + smot::NMS_T - 1) / smot::NMS_T;
    return (long long)n * nblk * 8;
}

extern "C" int smot_nms_fwd(const float* boxes_sorted, int n, float thresh, void* mask_ws, unsigned char* keep,
                            smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(n >= 0, "nms: n=%d", n);
    if (n == 0) return SMOT_OK;
    SMOT_REQUIRE(boxes_sorted && mask_ws && keep, "nms: null pointer");
    SMOT_REQUIRE(((uintptr_t)mask_ws & 7) == 0, "nms: workspace must be 8-byte aligned");
    const int nblk = (n + NMS_T - 1) / NMS_T;
    SMOT_REQUIRE(nblk <= 512, "nms: at most 32768 boxes (got %d)", n);
    const size_t smem = (size_t)NMS_T * nblk * 8;
    SMOT_REQUIRE(smem <= 64 * 1024, "nms: %d boxes need a %zu-byte mask slab (max 64 KiB: 8192 boxes)", n, smem);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(mask_ws);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, nblk), dim3(NMS_T), 0, st, boxes_sorted, n, thresh, mask);
    int rc = check_launch("nms mask");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(256), smem, st, (const unsigned long long*)mask, n, nblk, keep);
    return check_launch("nms scan");
}
