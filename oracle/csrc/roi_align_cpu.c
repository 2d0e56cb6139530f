/* TEST / BASELINE INFRASTRUCTURE — not product code (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; see oracle/__init__.py).
 *
 * Plain-C restatement of the legacy (non-"aligned") ROIAlign forward on CPU that the reference calls through
 * maskrcnn_benchmark's `_C.roi_align_forward` (siammot/modelling/track_head/EMM/sr_pool.py:28-31,89 and
 * box_head/box_head.py:46 via the upstream Pooler).  The upstream library (facebookresearch/maskrcnn-benchmark, un-pinned,
 * readme/INSTALL.md:89-92) is absent from /root/reference: this follows its published CPU algorithm
 * (csrc/cpu/ROIAlign_cpu.cpp: pre-computed bilinear taps per roi, then a loop over channels) as SURVEY.md Appendix A1
 * states it — coordinates `start + ph*bin + (iy+.5)*bin/grid`, samples outside [-1, size] contribute 0, clamp at 0,
 * the last cell is its own upper neighbour, `w1*v1 + w2*v2 + w3*v3 + w4*v4` summed in (iy, ix) order, divided by the
 * sample count.  fp32 throughout, every operation rounded separately (build with -ffp-contract=off): bit-identical to
 * oracle/emm_oracle.py::roi_align (tests/test_oracle_golden.py), which is pinned to the reference's golden vectors.
 * Parallel over (roi, channel) with OpenMP — what `at::parallel_for` gives the upstream operator — so that the CPU
 * baseline of bench.py times the reference's algorithm at a fair speed instead of a per-roi Python restatement.
 *
 *   gcc -O3 -fopenmp -ffp-contract=off -shared -fPIC oracle/csrc/roi_align_cpu.c -o oracle/_build/libroi_align_cpu.so
 */
#include <math.h>
#include <stdlib.h>

typedef struct {
    int pos1, pos2, pos3, pos4;
    float w1, w2, w3, w4;
} tap_t;

static void pre_calc(int height, int width, int pooled_h, int pooled_w, float roi_start_h, float roi_start_w,
                     float bin_h, float bin_w, int grid_h, int grid_w, tap_t* pc) {
    int idx = 0;
    for (int ph = 0; ph < pooled_h; ++ph)
        for (int pw = 0; pw < pooled_w; ++pw)
            for (int iy = 0; iy < grid_h; ++iy) {
                const float yy = (roi_start_h + (float)ph * bin_h) + (((float)iy + .5f) * bin_h) / (float)grid_h;
                for (int ix = 0; ix < grid_w; ++ix) {
                    const float xx = (roi_start_w + (float)pw * bin_w) + (((float)ix + .5f) * bin_w) / (float)grid_w;
                    float x = xx, y = yy;
                    tap_t t = {0, 0, 0, 0, 0.f, 0.f, 0.f, 0.f};
                    if (!(y < -1.0f || y > (float)height || x < -1.0f || x > (float)width)) {
                        if (y <= 0) y = 0;
                        if (x <= 0) x = 0;
                        int y_low = (int)y, x_low = (int)x, y_high, x_high;
                        if (y_low >= height - 1) {
                            y_high = y_low = height - 1;
                            y = (float)y_low;
                        } else {
                            y_high = y_low + 1;
                        }
                        if (x_low >= width - 1) {
                            x_high = x_low = width - 1;
                            x = (float)x_low;
                        } else {
                            x_high = x_low + 1;
                        }
                        const float ly = y - (float)y_low, lx = x - (float)x_low;
                        const float hy = 1.f - ly, hx = 1.f - lx;
                        t.pos1 = y_low * width + x_low;
                        t.pos2 = y_low * width + x_high;
                        t.pos3 = y_high * width + x_low;
                        t.pos4 = y_high * width + x_high;
                        t.w1 = hy * hx;
                        t.w2 = hy * lx;
                        t.w3 = ly * hx;
                        t.w4 = ly * lx;
                    }
                    pc[idx++] = t;
                }
            }
}

/* feat [B, C, H, W], rois [R, 5] = (batch index, x1, y1, x2, y2), out [R, C, pooled_h, pooled_w].  Returns 0. */
int roi_align_forward_cpu(const float* feat, int B, int C, int H, int W, const float* rois, int R, float spatial_scale,
                          int pooled_h, int pooled_w, int sampling_ratio, float* out) {
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < R; ++n) {
        const float* roi = rois + 5 * (size_t)n;
        const int b = (int)roi[0];
        if (b < 0 || b >= B) {
            rc = -1;
            continue;
        }
        const float roi_start_w = roi[1] * spatial_scale, roi_start_h = roi[2] * spatial_scale;
        const float roi_end_w = roi[3] * spatial_scale, roi_end_h = roi[4] * spatial_scale;
        const float roi_w = fmaxf(roi_end_w - roi_start_w, 1.f), roi_h = fmaxf(roi_end_h - roi_start_h, 1.f);
        const float bin_h = roi_h / (float)pooled_h, bin_w = roi_w / (float)pooled_w;
        const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)pooled_h);
        const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)pooled_w);
        const float count = (float)(grid_h * grid_w);
        tap_t* pc = (tap_t*)malloc(sizeof(tap_t) * (size_t)pooled_h * pooled_w * grid_h * grid_w);
        pre_calc(H, W, pooled_h, pooled_w, roi_start_h, roi_start_w, bin_h, bin_w, grid_h, grid_w, pc);
        for (int c = 0; c < C; ++c) {
            const float* d = feat + ((size_t)b * C + c) * H * W;
            float* o = out + ((size_t)n * C + c) * pooled_h * pooled_w;
            int idx = 0;
            for (int p = 0; p < pooled_h * pooled_w; ++p) {
                float acc = 0.f;
                for (int s = 0; s < grid_h * grid_w; ++s) {
                    const tap_t t = pc[idx++];
                    acc = acc + (((t.w1 * d[t.pos1] + t.w2 * d[t.pos2]) + t.w3 * d[t.pos3]) + t.w4 * d[t.pos4]);
                }
                o[p] = acc / count;
            }
        }
        free(pc);
    }
    return rc;
}
