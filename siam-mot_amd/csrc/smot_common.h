// Shared helpers for libsmot_emm.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/smot_emm.h"

namespace smot {

void set_error(const char* fmt, ...);
extern long long* g_trace;                           // common.hip: phase-trace buffer (smot_debug_trace), or nullptr
// common.hip — kernel timer for bench.py (no-op when idle).  timer_mark(slot, 0) opens a timed region: if this launch
// is sampled, the NEXT SMOT_LAUNCH in the region is issued through hipExtLaunchKernel with a start / stop event
// pair, i.e. the events carry the kernel's own begin / end timestamps (what rocprofv3 reports), not the span of
// two marker packets around it (which ran 2.5-3.5 us longer).  timer_mark(slot, 1) closes the region.
void timer_mark(int slot, int end, hipStream_t st);
bool timer_take(hipEvent_t* start, hipEvent_t* stop);
// common.hip — kernels that use more than 64 KiB of dynamic LDS opt in once per (device, kernel); thread-safe.
int ensure_lds_optin(const void* kernel, size_t bytes, const char* what);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return SMOT_OK;
}

#ifdef SMOT_DEBUG
int smot_debug_launch_flags();      // common.hip: hipExtAnyOrderLaunch when SMOT_ANY_ORDER is set (measurement library only)
#define SMOT_LAUNCH(KERNEL, GRID, BLOCK, SMEM, STREAM, ...)                                                       \
    do {                                                                                                         \
        hipEvent_t smot_e0_, smot_e1_;                                                                           \
        const int smot_fl_ = smot::smot_debug_launch_flags();                                                    \
        if (smot::timer_take(&smot_e0_, &smot_e1_)) {                                                            \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, SMEM, STREAM, smot_e0_, smot_e1_, smot_fl_, __VA_ARGS__); \
        } else if (smot_fl_) {                                                                                   \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, SMEM, STREAM, nullptr, nullptr, smot_fl_, __VA_ARGS__);   \
        } else {                                                                                                 \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SMEM, STREAM, __VA_ARGS__);                                  \
        }                                                                                                        \
    } while (0)
#else
#define SMOT_LAUNCH(KERNEL, GRID, BLOCK, SMEM, STREAM, ...)                                                       \
    do {                                                                                                         \
        hipEvent_t smot_e0_, smot_e1_;                                                                           \
        if (smot::timer_take(&smot_e0_, &smot_e1_)) {                                                            \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, SMEM, STREAM, smot_e0_, smot_e1_, 0, __VA_ARGS__);        \
        } else {                                                                                                 \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SMEM, STREAM, __VA_ARGS__);                                  \
        }                                                                                                        \
    } while (0)
#endif

#define SMOT_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            smot::set_error(__VA_ARGS__);  \
            return SMOT_ERR_BAD_ARG;       \
        }                                  \
    } while (0)

// fp32 ops that must NOT be contracted into FMAs, so that the rounding sequence matches the
// reference's separate torch mul / add kernels.  (In ROCm 7.2 __fmul_rn & co. are plain operators,
// so the guarantee comes from building with -ffp-contract=off — see build.py.)
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }

// torch.max / torch.maximum semantics: NaN propagates.
// ReLU as torch computes it: NaN goes through (fmaxf / v_max_f32 return the OTHER operand for a NaN: a track with a
// non-finite response would come out as a finite box)
__device__ __forceinline__ float relu_nan(float v) { return v < 0.0f ? 0.0f : v; }
__device__ __forceinline__ float max_nan(float a, float b) {
    return (a != a) ? a : ((b != b) ? b : fmaxf(a, b));
}

// torch.clamp semantics: NaN stays NaN
__device__ __forceinline__ float clamp_nan(float v, float lo, float hi) {
    return (v != v) ? v : fminf(fmaxf(v, lo), hi);
}

}  // namespace smot
