// Build-time split between the product library and the measurement library.
//
//   libsmot_emm.so        (default build)  : no environment variable is ever read and no kernel-selection or
//                                            ablation switch exists — smot::knobs() is a compile-time constant and
//                                            every `if (knobs().x)` below folds away.
//   libsmot_emm_debug.so  (-DSMOT_DEBUG)   : the A/B switches between CORRECT kernels and the timing ablations
//                                            (which return WRONG results) are read ONCE, when the library is
//                                            loaded, from SMOT_* environment variables, and can be changed
//                                            afterwards only through smot_debug_set_knob() — never by a stray
//                                            exported variable in the middle of a run, and never with getenv on
//                                            the launch path.  tools/ and the A/B tests load this library
//                                            explicitly; siammot_amd.ops never does on its own.
#pragma once

namespace smot {

enum XcorrVariant { XV_DEFAULT = 0, XV_WAVE, XV_PATCH, XV_PK, XV_ONE, XV_MFMA, XV_FILL, XV_COMPUTE };

struct Knobs {
    // A/B switches between correct kernels
    int no_fuse = 0;         // SMOT_NO_FUSE       : stand-alone pooler + xcorr instead of the fused kernel
    int roi_generic = 0;     // SMOT_ROI_GENERIC   : generic LDS-window ROIAlign for the 15/30 shapes too
    int tower_direct = 0;    // SMOT_TOWER_DIRECT  : direct implicit-GEMM tower instead of Winograd
    int tower_wide = 0;      // SMOT_TOWER_WIDE    : 32-channel tiles in the direct tower at large N
    int decode_split = 0;    // SMOT_DECODE_SPLIT  : thread groups per decode band (0 = automatic; 1, 2 or 4)
    int xcorr_variant = 0;   // SMOT_XCORR_VARIANT : XcorrVariant (wave|patch|pk|one|mfma|fill|compute)
    int decode_two_pass = 0; // SMOT_DECODE_2PASS  : band kernel + separate finalize launch (round-1 structure)
    int fused_gen = 0;       // SMOT_FUSED_GEN     : 0 = current fused pooling kernel, 2 = round-1 kernel
    int fused_order = 0;     // SMOT_FUSED_ORDER   : workgroup -> roi assignment of the pooling kernels (0 = default,
                             //                      1..3 = cost-sorted forms, 4 = grid order; sr_xcorr.hip fx_assign)
    int no_hint = 0;         // SMOT_NO_HINT       : 1 = the pooling + correlation kernel ignores the order hint and ranks
                             //                      its rois itself, 2 = the extraction does not write one either
    int tower_oct = 0;       // SMOT_TOWER_OCT     : 16-channel tiles per Winograd workgroup (0 = default, 1 or 2)
    int tower_bf3 = 1;       // SMOT_TOWER_BF3     : 0 = two-tile Winograd workgroups run the fp32 form of the GEMMs (1 = bf16 x 3)
    int any_order = 0;       // SMOT_ANY_ORDER     : launches carry hipExtAnyOrderLaunch (no barrier between the kernels of a
                             //                      stream: WRONG results — an upper bound on what overlapping the kernels'
                             //                      dispatch ramps and tails could buy, measure/any_order_ab.py)
    // timing ablations: WRONG results, measurement builds only
    int fused_abl = 0;       // SMOT_FUSED_ABL
    int wino_abl = 0;        // SMOT_WINO_ABL
    int tower_abl = 0;       // SMOT_TOWER_ABL
};

#ifdef SMOT_DEBUG
Knobs& knobs_mut();                                   // common.hip
inline const Knobs& knobs() { return knobs_mut(); }
bool launch_xcorr_variant(int variant, const float* x, const float* z, float* out, int planes, hipStream_t st);
#else
inline constexpr Knobs knobs() { return Knobs{}; }
#endif

}  // namespace smot
