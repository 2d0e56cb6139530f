#!/usr/bin/env python
"""Frame-pair timing of the OTHER yaml shape family (reference configs/dla/DLA_34_FPN_EMM_AOT.yaml:52-63: template
7x7, search region x5 -> 35x35, response 29x29, no centerness, cosine weight 0.1, pad 256) on the benchmark's maps
(720p, C=128, 30 tracks).  This family runs on the generic kernels (roi_align_levels + stand-alone xcorr +
tower_generic + heads + decode); parity is covered by tests (emm_aot / decode_aot / xcorr_aot goldens).  Prints one
JSON line; run it under rocprofv3 --kernel-trace --stats for the per-kernel table.

    python tools/aot_bench.py [--tracks 30] [--steps 500]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=30)
    ap.add_argument("--steps", type=int, default=500)
    a = ap.parse_args()
    from siammot_amd.config import get_default_cfg
    from siammot_amd.emm import EMM
    from siammot_amd.structures import BoxList
    from siammot_amd.track_utils import build_track_utils
    dev = torch.device("cuda", 0)
    cfg = get_default_cfg(channels=bench.CHANNELS)
    th = cfg.MODEL.TRACK_HEAD
    th.POOLER_RESOLUTION, th.SEARCH_REGION, th.PAD_PIXELS = 7, 5.0, 256
    th.EMM.USE_CENTERNESS, th.EMM.COSINE_WINDOW_WEIGHT = False, 0.1
    image_wh = (bench.NET_HW[1], bench.NET_HW[0])
    boxes = bench.synthetic_boxes(a.tracks, image_wh)
    emm = EMM(cfg, build_track_utils(cfg)).eval()
    bench.init_predictor(emm.predictor, boxes)
    emm = emm.to(dev)
    feats = [bench.synthetic_features(100 + k, dev) for k in range(4)]
    det = BoxList(boxes.to(dev), image_wh, mode="xyxy")
    det.add_field("ids", torch.arange(a.tracks, device=dev))
    det.add_field("labels", torch.ones(a.tracks, dtype=torch.int64, device=dev))
    with torch.no_grad():
        state = emm.extract_cache(feats[3], det)
        for phase_steps in (200, a.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(phase_steps):
                z, sr, d = state
                f = feats[k % 4]
                _, result, _ = emm(f, d, sr, template_features=z)
                state = emm.extract_cache(f, det)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    print(json.dumps({"workload": "AOT shape family frame pair (Rz=7, Rx=35, Ho=29, no centerness) on 720p maps, C=128",
                      "tracks": a.tracks, "steps": a.steps, "ms_per_step": dt / a.steps * 1e3,
                      "frame_pairs_per_s": a.steps / dt, "rx": emm.rx, "rz": emm.rz,
                      "boxes_finite": bool(torch.isfinite(result[0].bbox).all())}))


if __name__ == "__main__":
    main()
