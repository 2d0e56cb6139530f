#!/usr/bin/env python
"""Per-operator timing of the EMM hot path on the GPU (HIP events around batches of launches on the
launch stream).  Prints one JSON line per (op, variant, N):  python tools/kernel_bench.py [--tracks 30 100]

An op's time = (event span of B back-to-back launches) / B, min and median over R repetitions, so it
includes the ~1-2 us same-stream kernel boundary but not Python's per-call overhead beyond what the GPU
cannot hide.  Inputs are resident in HBM (and, at these sizes, mostly in the 256 MB Infinity Cache — as
they are in the real pipeline, where each kernel consumes what the previous one just produced).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def timed(fn, batch=20, reps=7):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(batch):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / batch)
    out.sort()
    return out[0], out[len(out) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, nargs="+", default=[30, 100])
    args = ap.parse_args()
    from siammot_amd import ops
    from siammot_amd.config import get_default_cfg
    from siammot_amd.emm import EMM
    from siammot_amd.structures import BoxList
    from siammot_amd.track_utils import build_track_utils
    dev = torch.device("cuda", 0)
    ops.load_library()
    C = bench.CHANNELS
    image_wh = (bench.NET_HW[1], bench.NET_HW[0])
    feats = bench.synthetic_features(7, dev)
    scales = (0.25, 0.125, 0.0625, 0.03125)
    for n in args.tracks:
        boxes = bench.synthetic_boxes(n, image_wh).to(dev)
        cfg = get_default_cfg(channels=C)
        emm = EMM(cfg, build_track_utils(cfg)).eval()
        bench.init_predictor(emm.predictor, boxes.cpu())
        emm = emm.to(dev)
        params = {k: v for k, v in emm.predictor.named_parameters()}
        det = BoxList(boxes, image_wh, mode="xyxy")
        det.add_field("ids", torch.arange(n, device=dev))
        det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
        with torch.no_grad():
            sr = ops.search_region(boxes, 512, 1.0, 0)
            z = ops.roi_align_levels(feats, boxes, boxes, 15, scales, 2)
            x = ops.roi_align_levels(feats, sr, boxes, 30, scales, 2, [128, 64, 32, 16])
            resp = ops.xcorr_depthwise(x, z)
            logits = ops.emm_predictor(resp, params)
            state = emm.extract_cache(feats, det)
            xbytes = 4.0 * n * C * (900 + 225 + 256)

            def rec(op, variant, t, extra=None):
                d = {"op": op, "variant": variant, "tracks": n, "min_us": round(t[0], 2), "median_us": round(t[1], 2)}
                if extra:
                    d.update(extra)
                print(json.dumps(d), flush=True)

            rec("roi_align_sr30", "default", timed(lambda: ops.roi_align_levels(feats, sr, boxes, 30, scales, 2,
                                                                                 [128, 64, 32, 16])))
            rec("roi_align_z15", "default", timed(lambda: ops.roi_align_levels(feats, boxes, boxes, 15, scales, 2)))
            t = timed(lambda: ops.xcorr_depthwise(x, z), batch=50)
            rec("xcorr", "product library", t, {"algorithmic_GBps_at_min": round(xbytes / t[0] / 1e3, 1),
                                                "frac_of_8TBps": round(xbytes / t[0] / 1e3 / 8000.0, 4)})
            # older generations and phase ablations: measurement library only (csrc/knobs.h)
            for var in ("default", "one", "mfma", "pk", "patch", "wave", "fill", "compute"):
                with ops.debug_library(SMOT_XCORR_VARIANT=var):
                    t = timed(lambda: ops.xcorr_depthwise(x, z), batch=50)
                rec("xcorr", var, t, {"algorithmic_GBps_at_min": round(xbytes / t[0] / 1e3, 1),
                                      "frac_of_8TBps": round(xbytes / t[0] / 1e3 / 8000.0, 4)})
            t = timed(lambda: ops.sr_xcorr_fused(feats, boxes, sr, z, 30, 15, scales, 2, 512), batch=50)
            rec("sr_pool+xcorr fused", "default", t)
            rec("predictor", "winograd", timed(lambda: ops.emm_predictor(resp, params)))
            rec("predictor", "direct", timed(lambda: ops.emm_predictor(resp, params, winograd=False)))
            for abl in ("1", "2"):
                with ops.debug_library(SMOT_TOWER_ABL=abl):
                    rec("predictor", "direct ablation %s" % abl,
                        timed(lambda: ops.emm_predictor(resp, params, winograd=False)))
            rec("decode", "default", timed(lambda: ops.emm_decode(logits, sr, boxes, 30, 15, 512)))
            rec("search_region", "default", timed(lambda: ops.search_region(boxes, 512, 1.0, 0)))

            def frame_pair():
                nonlocal state
                zz, ssr, dd = state
                emm(feats, dd, ssr, template_features=zz)
                state = emm.extract_cache(feats, det)
            rec("frame_pair(EMM.forward+extract_cache)", "fused", timed(frame_pair, batch=10))
            with ops.debug_library(SMOT_NO_FUSE=1):
                rec("frame_pair(EMM.forward+extract_cache)", "unfused", timed(frame_pair, batch=10))


if __name__ == "__main__":
    main()
