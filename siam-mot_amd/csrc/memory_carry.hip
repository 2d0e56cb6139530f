// Dormant rows of the track memory, carried from frame to frame on the device.
//
// Replaces TrackHead._update_memory_with_dormant_track (reference siammot/modelling/track_head/track_head.py:77-97): the
// reference appends, every frame, the cached (template feature, search region, template box) of every dormant track to the
// memory of the active tracks — `torch.cat` of the templates plus two `cat_boxlist` calls (boxes and three fields each),
// a dozen launches and their host time.  A dormant track's entry never changes while it is dormant, and the memory the
// frame's head just ran on holds it (as an active row if the track went dormant in this frame, as a carried row
// otherwise): the rows are COPIED from that memory to the rows behind the new memory's active rows, in one launch.
//
// Rows: D source rows (indices into the previous memory, by value in the kernel arguments — the host knows the previous
// memory's row order) -> destination rows dst_row0 .. dst_row0 + D - 1, where dst_row0 (the number of active rows) is
// either given by value or read from a device word (the solver's pool state word 4: a launch enqueued BEFORE the host
// has read the frame's record).  Destination rows at or beyond `capacity` are not written.
//
// Pure copies: bit-exact by construction.  HBM-bound: 2 x D x (C rz rz + 11) x 4 bytes; one workgroup per 16 KB piece.
#include "smot_common.h"
#include <cstddef>

namespace smot {

constexpr int MC_MAXROWS = 256;
constexpr int MC_THREADS = 256;
constexpr int MC_VEC_PER_THREAD = 4;      // 16-byte pieces per thread: 16 KB per workgroup

struct MemoryCarryArgs {
    const float* src_z;          // [*, row_floats]
    const float* src_boxes;      // [*, 4]
    const float* src_sr;         // [*, 4]
    const long long* src_ids;    // [*]
    const long long* src_labels; // [*]
    const float* src_scores;     // [*]
    float* dst_z;
    float* dst_boxes;
    float* dst_sr;
    long long* dst_ids;
    long long* dst_labels;
    float* dst_scores;
    const int* dst_row0_dev;     // device word holding the first destination row, or nullptr
    int dst_row0;
    int capacity;                // rows of the destination buffers
    int row_floats;              // C * rz * rz
    short src_rows[MC_MAXROWS];
};

template <bool VEC>
__global__ void __launch_bounds__(MC_THREADS) memory_carry_kernel(MemoryCarryArgs A) {
    const int j = blockIdx.y;
    // (read straight from the kernel-argument segment: a by-value array indexed by a run-time value would be copied to
    // scratch or LDS first — 16 KB of LDS per workgroup in the first build)
    typedef const char __attribute__((address_space(4))) ka_char;
    typedef const short __attribute__((address_space(4))) ka_short;
    ka_char* ka = (ka_char*)__builtin_amdgcn_kernarg_segment_ptr();
    const int s = *(ka_short*)(ka + offsetof(MemoryCarryArgs, src_rows) + 2 * j);
    const int d0 = A.dst_row0_dev ? *A.dst_row0_dev : A.dst_row0;
    const int d = d0 + j;
    if (d0 < 0 || d >= A.capacity) return;
    const float* __restrict__ src = A.src_z + (size_t)s * A.row_floats;
    float* __restrict__ dst = A.dst_z + (size_t)d * A.row_floats;
    if constexpr (VEC) {
        const int nv = A.row_floats >> 2;
        const int base = blockIdx.x * (MC_THREADS * MC_VEC_PER_THREAD) + threadIdx.x;
        // (named registers: an array filled under a condition was promoted to LDS, 16 KB per workgroup, in the first build)
        const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src);
        float4* __restrict__ d4 = reinterpret_cast<float4*>(dst);
        const int e0 = base, e1 = base + MC_THREADS, e2 = base + 2 * MC_THREADS, e3 = base + 3 * MC_THREADS;
        static_assert(MC_VEC_PER_THREAD == 4, "four pieces per thread");
        const float4 v0 = s4[min(e0, nv - 1)];                           // the loads first (clamped, unconditional): in flight together
        const float4 v1 = s4[min(e1, nv - 1)];
        const float4 v2 = s4[min(e2, nv - 1)];
        const float4 v3 = s4[min(e3, nv - 1)];
        if (e0 < nv) d4[e0] = v0;
        if (e1 < nv) d4[e1] = v1;
        if (e2 < nv) d4[e2] = v2;
        if (e3 < nv) d4[e3] = v3;
    } else {
        const int base = blockIdx.x * (MC_THREADS * MC_VEC_PER_THREAD * 4);
        for (int e = base + threadIdx.x; e < min(A.row_floats, base + MC_THREADS * MC_VEC_PER_THREAD * 4); e += MC_THREADS)
            dst[e] = src[e];
    }
    if (blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t < 4) {
            A.dst_boxes[(size_t)d * 4 + t] = A.src_boxes[(size_t)s * 4 + t];
        } else if (t < 8) {
            A.dst_sr[(size_t)d * 4 + t - 4] = A.src_sr[(size_t)s * 4 + t - 4];
        } else if (t == 8) {
            A.dst_ids[d] = A.src_ids[s];
        } else if (t == 9) {
            A.dst_labels[d] = A.src_labels[s];
        } else if (t == 10) {
            A.dst_scores[d] = A.src_scores[s];
        }
    }
}

}  // namespace smot

extern "C" int smot_memory_carry_max_rows(void) { return smot::MC_MAXROWS; }

extern "C" int smot_memory_carry_fwd(const float* src_templates, const float* src_boxes, const float* src_sr,
                                     const int64_t* src_ids, const int64_t* src_labels, const float* src_scores,
                                     int src_rows_total, float* dst_templates, float* dst_boxes, float* dst_sr,
                                     int64_t* dst_ids, int64_t* dst_labels, float* dst_scores, int dst_capacity,
                                     const int* rows, int D, int dst_row0, const int* dst_row0_dev, int row_floats,
                                     smot_stream_t stream) {
    using namespace smot;
    SMOT_REQUIRE(D >= 0 && D <= MC_MAXROWS, "memory_carry: D=%d not in [0,%d]", D, MC_MAXROWS);
    if (D == 0) return SMOT_OK;
    SMOT_REQUIRE(row_floats > 0 && src_rows_total > 0 && src_rows_total <= 32767 && dst_capacity > 0,
                 "memory_carry: bad sizes (row_floats=%d, source rows=%d, capacity=%d)", row_floats, src_rows_total,
                 dst_capacity);
    SMOT_REQUIRE(rows && src_templates && src_boxes && src_sr && src_ids && src_labels && src_scores && dst_templates &&
                     dst_boxes && dst_sr && dst_ids && dst_labels && dst_scores,
                 "memory_carry: null pointer");
    SMOT_REQUIRE(dst_row0_dev || (dst_row0 >= 0 && dst_row0 + D <= dst_capacity),
                 "memory_carry: rows %d..%d beyond the destination's %d rows", dst_row0, dst_row0 + D - 1, dst_capacity);
    MemoryCarryArgs A;
    for (int j = 0; j < D; ++j) {
        SMOT_REQUIRE(rows[j] >= 0 && rows[j] < src_rows_total, "memory_carry: source row %d not in [0,%d)", rows[j],
                     src_rows_total);
        A.src_rows[j] = (short)rows[j];
    }
    A.src_z = src_templates;
    A.src_boxes = src_boxes;
    A.src_sr = src_sr;
    A.src_ids = (const long long*)src_ids;
    A.src_labels = (const long long*)src_labels;
    A.src_scores = src_scores;
    A.dst_z = dst_templates;
    A.dst_boxes = dst_boxes;
    A.dst_sr = dst_sr;
    A.dst_ids = (long long*)dst_ids;
    A.dst_labels = (long long*)dst_labels;
    A.dst_scores = dst_scores;
    A.dst_row0_dev = dst_row0_dev;
    A.dst_row0 = dst_row0;
    A.capacity = dst_capacity;
    A.row_floats = row_floats;
    const bool vec = (row_floats & 3) == 0 && (((uintptr_t)src_templates | (uintptr_t)dst_templates) & 15) == 0;
    const int per_wg = MC_THREADS * MC_VEC_PER_THREAD * 4;
    dim3 grid((row_floats + per_wg - 1) / per_wg, D);
    if (vec)
        hipLaunchKernelGGL(memory_carry_kernel<true>, grid, dim3(MC_THREADS), 0, (hipStream_t)stream, A);
    else
        hipLaunchKernelGGL(memory_carry_kernel<false>, grid, dim3(MC_THREADS), 0, (hipStream_t)stream, A);
    return check_launch("memory_carry");
}
