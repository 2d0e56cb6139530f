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
    from siammot_amd.track_utils import build_track_utils
    dev = torch.device("cuda:0")
    torch.set_num_threads(bench._cpu_threads())
    n = args.tracks
    cfg = get_default_cfg(channels=128)
    emm = EMM(cfg, build_track_utils(cfg)).eval()
    bench.init_predictor(emm.predictor, bench.synthetic_boxes(n, (1280, 704)))
    emm = emm.to(dev)
    t0 = time.time()
    rows = bench.argmax_rows(emm, ops, dev, n, args.pairs, progress=True)      # (the loop lives in bench.py: its own line
                                                                              #  carries the same statistic, measured in-run)
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
