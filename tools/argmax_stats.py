#!/usr/bin/env python
"""Arg-max exact fraction and IoU histogram of the HIP head vs the fp32 CPU oracle at BASELINE.json configs[1]
(SURVEY.md §7 "argmax-exact fraction + IoU histogram").

For every seeded frame pair (fresh random 720p FPN maps, the 30 benchmark boxes jittered per seed, the benchmark's
weights) the whole head runs on the GPU (EMM.extract_cache + EMM.forward through the C ABI) and in the CPU oracle
(oracle/emm_oracle.py, fp32, the reference's torch ops).  Per track: same arg-max cell?  IoU of the boxes, score
difference.  Every disagreement is attributed: the oracle's fp64 scores of the two cells are compared — a gap below
1e-6 is an fp32 rounding tie (library exponentials / summation order differ between torch-CPU and the device),
anything larger would be a ranking error of the kernel (none is tolerated: the script exits non-zero).

    python tools/argmax_stats.py [--pairs 300] [--out gpurun_out/argmax_stats]      (GPU box)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from oracle import emm_oracle as O              # noqa: E402  (checker)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=300)
    ap.add_argument("--tracks", type=int, default=30)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "argmax_stats"))
    args = ap.parse_args()
    from siammot_amd import ops
    from siammot_amd.config import get_default_cfg
    from siammot_amd.emm import EMM
    from siammot_amd.structures import BoxList
    from siammot_amd.track_utils import build_track_utils
    dev = torch.device("cuda:0")
    torch.set_num_threads(bench._cpu_threads())
    n = args.tracks
    image_wh = (1280, 704)
    cfg = get_default_cfg(channels=128)
    emm = EMM(cfg, build_track_utils(cfg)).eval()
    base_boxes = bench.synthetic_boxes(n, image_wh)
    bench.init_predictor(emm.predictor, base_boxes)
    emm = emm.to(dev)
    params_cpu = {k: v.detach().cpu() for k, v in emm.predictor.named_parameters()}
    ocfg = O.EMMConfig(channels=128)
    fe, pr = emm.feature_extractor.pooler_x, emm.predictor
    rows = []
    t0 = time.time()
    with torch.no_grad():
        for seed in range(args.pairs):
            g = torch.Generator().manual_seed(10_000 + seed)
            jitter = (torch.rand((n, 1), generator=g) * 6.0 - 3.0)
            boxes = (base_boxes + jitter).clamp(min=0)
            boxes[:, 2].clamp_(max=image_wh[0] - 1)
            boxes[:, 3].clamp_(max=image_wh[1] - 1)
            gd = torch.Generator(device=dev).manual_seed(20_000 + seed)
            fa = tuple(torch.randn((1, 128, 704 // s, 1280 // s), generator=gd, device=dev) for s in (4, 8, 16, 32, 64))
            fb = tuple(torch.randn((1, 128, 704 // s, 1280 // s), generator=gd, device=dev) for s in (4, 8, 16, 32, 64))
            det = BoxList(boxes.to(dev), image_wh, mode="xyxy")
            det.add_field("ids", torch.arange(n, device=dev))
            det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
            z, sr, d = emm.extract_cache(fa, det)
            bb, conf, idx = ops.emm_track(fb, d[0].bbox, sr[0].bbox, z, pr.param_dict(), emm.rx, emm.rz, tuple(fe.scales),
                                          fe.sampling_ratio, emm.pad_pixels, sigma=emm.sigma,
                                          use_centerness=emm.use_centerness, clip_wh=image_wh, gn_groups=pr.gn_groups,
                                          gn_eps=pr.gn_eps, return_index=True)
            fa_c, fb_c = [t.cpu() for t in fa], [t.cpu() for t in fb]
            z_o, sr_o = O.extract_cache(ocfg, fa_c, boxes)
            bb_o, conf_o, _, inter = O.emm_forward(ocfg, params_cpu, fb_c, boxes, sr_o, z_o, image_wh,
                                                   return_intermediates=True, reference_ops=True)
            idx, idx_o = idx.cpu(), inter["idx"]
            iou = bench.box_iou(bb.cpu().double(), bb_o.double())
            diff = (idx != idx_o).nonzero().flatten().tolist()
            gaps, kind = {}, {}
            if diff:
                # Attribution.  (1) the kernel's OWN logits (same operators, called one by one) decoded by the fp32
                # oracle: if that elects the kernel's cell, the decode is exact and the difference was made upstream
                # (pooling / correlation / Winograd-vs-direct summation order moved the logits by ~1e-5 of their
                # scale and two cells swapped places).  (2) otherwise the two cells' fp64 scores on the kernel's
                # logits: a gap <= 1e-6 is an exponential-rounding tie between torch-CPU and the device.
                resp = ops.sr_xcorr_fused(fb, d[0].bbox, sr[0].bbox, z, emm.rx, emm.rz, tuple(fe.scales), fe.sampling_ratio,
                                          emm.pad_pixels)
                lg = ops.emm_predictor(resp, pr.param_dict(), pr.gn_groups, pr.gn_eps).cpu()[diff]
                xs, ys = O.grid_axes(sr_o[diff], ocfg.rx, ocfg.rz, ocfg.pad_pixels)
                up32 = [O.bicubic_upsample_torch(lg[:, a:b]) for a, b in ((0, 2), (2, 3), (3, 7))]
                _, _, idx_mix = O.decode(up32[0], up32[1], up32[2], xs, ys, boxes[diff], True, 0.4)
                up64 = [O.bicubic_upsample(lg[:, a:b].double()) for a, b in ((0, 2), (2, 3), (3, 7))]
                score64, _ = O.score_map(up64[0], up64[1], up64[2], boxes[diff].double(), True, 0.4)
                up_o = [O.bicubic_upsample(inter[k][diff].double()) for k in ("cls", "center", "reg")]
                score_o, _ = O.score_map(up_o[0], up_o[1], up_o[2], boxes[diff].double(), True, 0.4)
                for j, t in enumerate(diff):
                    gaps[t] = float(score_o[j, idx_o[t]] - score_o[j, idx[t]])       # on the ORACLE's logits
                    if int(idx_mix[j]) == int(idx[t]):
                        kind[t] = 1                                                  # upstream fp32 rounding
                    elif abs(float(score64[j, idx_mix[j]] - score64[j, idx[t]])) <= 1e-6:
                        kind[t] = 2                                                  # decode-level rounding tie
                    else:
                        kind[t] = 3                                                  # unexplained
            for t in range(n):
                rows.append((seed, t, int(idx[t] == idx_o[t]), float(iou[t]), float((conf[t].cpu() - conf_o[t]).abs()),
                             float((bb[t].cpu() - bb_o[t]).abs().max()), gaps.get(t, 0.0), kind.get(t, 0)))
            if (seed + 1) % 50 == 0:
                print("%d pairs, %.0f s" % (seed + 1, time.time() - t0), flush=True)
    a = np.array(rows, dtype=np.float64)
    same = a[:, 2] > 0.5
    one_minus_iou = 1.0 - a[:, 3]
    edges = [0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0 + 1e-9]
    hist = np.histogram(np.clip(one_minus_iou, 0, 1), bins=edges)[0].tolist()
    dis = a[~same]
    summary = {
        "config": "BASELINE.json configs[1]: C=128, net input 704x1280, %d tracks (benchmark boxes +-3 px per seed), "
                  "benchmark weights, fresh N(0,1) FPN maps per seed" % n,
        "frame_pairs": args.pairs, "tracks_total": int(len(a)),
        "argmax_exact": int(same.sum()), "argmax_exact_frac": float(same.mean()),
        "min_iou": float(a[:, 3].min()), "min_iou_among_argmax_exact": float(a[same, 3].min()),
        "tracks_below_1e-3_iou_bar": int((one_minus_iou > 1e-3).sum()),
        "one_minus_iou_histogram": {"bin_edges": edges[:-1] + [1.0], "counts": hist},
        "max_score_err": float(a[:, 4].max()), "max_box_err_px_among_argmax_exact": float(a[same, 5].max()),
        "disagreements": [{"seed": int(r[0]), "track": int(r[1]), "iou": r[3], "fp64_score_gap_on_oracle_logits": r[6],
                           "cause": {1: "upstream fp32 rounding (oracle decode of the kernel's own logits elects the "
                                        "kernel's cell)", 2: "decode rounding tie (fp64 gap <= 1e-6 on the kernel's "
                                        "logits)", 3: "UNEXPLAINED"}[int(r[7])]} for r in dis],
        "disagreements_upstream_fp32_rounding": int((dis[:, 7] == 1).sum()) if len(dis) else 0,
        "disagreements_decode_rounding_ties": int((dis[:, 7] == 2).sum()) if len(dis) else 0,
        "disagreements_unexplained": int((dis[:, 7] == 3).sum()) if len(dis) else 0,
        "seconds": time.time() - t0,
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(summary, open(args.out + ".json", "w"), indent=1)
    with open(args.out + ".md", "w") as f:
        f.write("# Arg-max agreement and IoU vs the fp32 CPU oracle — %s\n\n" % summary["config"])
        f.write("| frame pairs | tracks | arg-max identical | fraction | min IoU | tracks under the 1e-3 IoU bar | max score err |\n|---|---|---|---|---|---|---|\n")
        f.write("| %d | %d | %d | %.6f | %.9f | %d | %.3e |\n\n" % (args.pairs, len(a), same.sum(), same.mean(), a[:, 3].min(),
                                                                   summary["tracks_below_1e-3_iou_bar"], a[:, 4].max()))
        f.write("Histogram of 1 - IoU:\n\n| bin | tracks |\n|---|---|\n")
        for lo, hi, c in zip(edges[:-1], edges[1:], hist):
            f.write("| [%.0e, %.0e) | %d |\n" % (lo, min(hi, 1.0), c))
        f.write("\nDisagreements: %d — upstream fp32 rounding (the oracle's decode of the kernel's own logits elects the "
                "kernel's cell): %d; decode-level rounding ties (fp64 gap <= 1e-6): %d; unexplained: %d\n" % (
                    len(dis), summary["disagreements_upstream_fp32_rounding"], summary["disagreements_decode_rounding_ties"],
                    summary["disagreements_unexplained"]))
        for r in dis:
            f.write("* seed %d track %d: IoU %.6f, fp64 score gap on the oracle's logits %.3e, cause %d\n" % (r[0], r[1], r[3], r[6], r[7]))
    print(json.dumps({k: v for k, v in summary.items() if k != "disagreements"}))
    sys.exit(1 if summary["disagreements_unexplained"] else 0)


if __name__ == "__main__":
    main()
