#!/usr/bin/env python
"""Golden vectors AT THE BENCHMARK CONFIGURATIONS, produced by the reference's own EMM code on CPU.

VERDICT r1 "missing #4": the frame-pair fixtures of ``gen_golden.py`` are C=64/32 on 512x384 maps; the
configurations the headline number is quoted on (BASELINE.json configs[1]: C=128, net input 704x1280, the 30
``bench.synthetic_boxes``; configs[2]: 100 tracks) were only compared with the oracle.  This script runs the
UNMODIFIED reference (same stubs as ``gen_golden.py``; EMM/track_core.py:28-98, EMM/sr_pool.py, EMM/xcorr.py,
EMM/feature_extractor.py, track_utils.py) on exactly the tensors ``bench.py`` times:

    features   bench.synthetic_features(100), (101)      (torch CPU generator, seeds as rank 0 of bench.py)
    boxes      bench.synthetic_boxes(N, (1280, 704))
    weights    bench.init_predictor (torch generator seed 1)

and stores, for both frame orders of the timed loop (templates from frame A, tracking on frame B, and the
reverse), only the small outputs: search regions, FPN levels, boxes, scores, the arg-max cell the reference's
``decode_response`` picked (captured by wrapping ``torch.argmax`` while it runs) and the margin between the best
and the second-best penalised score of every track (so a disagreement can be attributed: fp32 library rounding
can only flip an arg-max whose margin is at the 1e-7 level).  Input checksums are stored too, so a consumer can
tell "inputs drifted" (another torch build's generator) from "outputs differ".

Round 3 (VERDICT r2 missing #2, next #4): the same for configs[0] (800x800 net input, 4 tracks), configs[4] (C=256,
1056x1920, 50 tracks) and the second yaml family (DLA_34_FPN_EMM_AOT.yaml: Rz 7, Rx 35, no centerness) at the
configs[1] size — tests/golden_inputs.py::BENCH_CONFIGS names them.

Runs only where /root/reference exists.      Usage:  python oracle/gen_golden_bench.py [n30 n100 cfg0 cfg4 aot_n30]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench                      # noqa: E402  (synthetic_boxes / synthetic_features / init_predictor)
import gen_golden as gg           # noqa: E402  (stubs + reference cfg)

import golden_inputs as gi        # noqa: E402  (BENCH_CONFIGS / BENCH_FAMILIES)


def checksum(tensors):
    return np.array([float(t.double().sum()) for t in tensors] + [float(t.double().abs().sum()) for t in tensors])


def main():
    names = [("n" + a if a.isdigit() else a) for a in sys.argv[1:]] or list(gi.BENCH_CONFIGS)
    gg.install_stubs()
    from siammot.modelling.track_head.EMM import track_core as ref_core
    from siammot.modelling.track_head.track_utils import build_track_utils
    from siammot.utils import registry as ref_registry
    torch.set_grad_enabled(False)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    out_dir = os.path.join(ROOT, "tests", "golden")

    captured = {}
    real_argmax = torch.argmax

    for name in names:
        c = gi.BENCH_CONFIGS[name]
        fam = gi.BENCH_FAMILIES[c["family"]]
        n = c["n"]
        bench.CHANNELS, bench.NET_HW = c["channels"], tuple(c["net_hw"])      # synthetic_features reads them
        image_wh = (bench.NET_HW[1], bench.NET_HW[0])
        feats = [bench.synthetic_features(100 + k, "cpu") for k in range(2)]
        grid = 16 * (int(fam["rz"] * fam["search_region"]) - fam["rz"] + 1)

        def spy(inp, *a, **k):
            r = real_argmax(inp, *a, **k)
            if inp.dim() == 2 and inp.shape[1] == grid * grid:
                captured["idx"] = r.clone()
                top2 = torch.topk(inp, 2, dim=1).values
                captured["margin"] = (top2[:, 0] - top2[:, 1]).clone()
                captured["best"] = top2[:, 0].clone()
            return r

        t0 = time.time()
        boxes = bench.synthetic_boxes(n, image_wh)
        cfg = gg.reference_cfg(dict(fam, channels=c["channels"]))
        track_utils, _ = build_track_utils(cfg)
        emm = ref_registry.SIAMESE_TRACKER["EMM"](cfg, track_utils).eval()
        bench.init_predictor(emm.predictor, boxes)
        sub = gi.bench_channel_subset(c["channels"])
        step = 7 if fam["rz"] == 15 else 3
        out = {"boxes": boxes.numpy(), "feat_checksum": np.stack([checksum(f) for f in feats]),
               "param_checksum": checksum([p for _, p in sorted(emm.predictor.named_parameters())])}
        for tag, (a, b) in (("ab", (0, 1)), ("ba", (1, 0))):
            det = gg.boxlist(boxes.numpy(), image_wh)
            z, sr, det_out = emm.extract_cache(feats[a], det)
            levels = emm.feature_extractor.pooler_z.map_levels([det])
            torch.argmax = spy
            ref_core.torch.argmax = spy
            try:
                _, result, _ = emm(feats[b], det_out, sr, template_features=z)
            finally:
                torch.argmax = real_argmax
            res = result[0]
            assert len(res) == n          # clip_to_image's filtered copy is discarded (track_core.py:177-178)
            out.update({
                "sr_" + tag: sr[0].bbox.numpy(), "levels_" + tag: levels.numpy().astype(np.int32),
                "bb_" + tag: res.bbox.numpy(), "scores_" + tag: res.get_field("scores").numpy(),
                "idx_" + tag: captured["idx"].numpy().astype(np.int64),
                "margin_" + tag: captured["margin"].numpy(), "best_" + tag: captured["best"].numpy(),
                # a thin slice of the template tensor pins the template pooler at this geometry too
                "z_sub_" + tag: z[:, sub].numpy()[:, :, ::step, ::step].copy(),
            })
            print("bench_%s %s: levels %s, min margin %.3e, %.0f s" %
                  (name, tag, np.bincount(levels.numpy().astype(np.int64), minlength=4).tolist(),
                   float(captured["margin"].min()), time.time() - t0), flush=True)
        np.savez_compressed(os.path.join(out_dir, "bench_%s.npz" % name), **out)


if __name__ == "__main__":
    main()
