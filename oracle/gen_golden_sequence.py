#!/usr/bin/env python
"""Closed-loop golden sequences: the REFERENCE's tracking step with the REFERENCE's EMM inside it.

Every frame goes through the reference's own, UNMODIFIED

    CombinedROIHeads.forward           siammot/modelling/roi_heads.py:22-52   (track -> _refine_tracks -> cat ->
                                                                                 solver -> get_track_memory)
    TrackHead                          siammot/modelling/track_head/track_head.py:8-110
    TrackSolver                        siammot/modelling/track_head/track_solver.py:8-108
    TrackPool / TrackUtils             siammot/modelling/track_head/track_utils.py
    EMM (forward + extract_cache)      siammot/modelling/track_head/EMM/*.py
    ROIBoxHead + PostProcessor         siammot/modelling/box_head/*.py        (case "refine")

on CPU, with the ``maskrcnn_benchmark`` symbols stubbed exactly as in gen_golden.py / gen_golden_refine.py (scalar-loop
ROIAlign, numpy NMS, the oracle's own BoxList (oracle/ref_structures.py)).  The tracked box of frame t becomes the template box and the search
region of frame t+1, so rounding differences can compound and an arg-max flip moves a template: this is the parity
the single-frame-pair fixtures cannot give.

The detector is outside the path: the frame's detections are synthetic (tests/golden_inputs.py::SequenceInputs) and
enter ``CombinedROIHeads.forward`` through its ``box`` head — a switch module that returns BoxLists carrying a
``detector_output`` marker unchanged and sends everything else (the propagated tracks ``_refine_tracks`` passes as
proposals) to the real box head; case "plain" has no box head, its switch returns the proposals with their scores
in the (1, 2] band, i.e. ``_refine_tracks`` degenerates to ``score + 1`` (what ``TrackingLoop`` does without a
``refine_tracks`` callable).

Stored per frame (tests/golden/sequence_<case>.npz): the solver's output (boxes, ids, scores, labels), the pool state
(active ids, dormant ids with their last-active frame, max id), the track memory's ids, the raw EMM output (boxes,
scores, ids), the arg-max cell and the margin between best and second-best penalised score of every tracked row, the
refined tracks, and input checksums.  Build container only.

Usage:  python oracle/gen_golden_sequence.py [plain refine]
"""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import golden_inputs as gi                                    # noqa: E402
import gen_golden as gg                                       # noqa: E402
import gen_golden_refine as gr                                # noqa: E402
from oracle import box_head_oracle as BO                      # noqa: E402
from oracle.ref_structures import BoxList                     # noqa: E402


class Cfg(types.SimpleNamespace):
    def clone(self):
        return self


def reference_cfg(case):
    ns = types.SimpleNamespace
    fam = gi.BENCH_FAMILIES[case.get("family", "default")]       # the yaml family's track-head keys
    emm = ns(USE_CENTERNESS=fam["use_centerness"], COSINE_WINDOW_WEIGHT=fam["sigma"], CLS_POS_REGION=0.8, TRACK_LOSS_WEIGHT=1.0,
             POS_RATIO=0.25, HN_RATIO=0.25)          # the last two: target sampler only (built, never called)
    th = ns(POOLER_RESOLUTION=fam["rz"], POOLER_SCALES=fam["scales"], POOLER_SAMPLING_RATIO=2,
            SEARCH_REGION=fam["search_region"], PAD_PIXELS=fam["pad_pixels"], MINIMUM_SREACH_REGION=fam["min_search_wh"],
            MAX_DORMANT_FRAMES=case["max_dormant_frames"],
            EMM=emm, MODEL="EMM", TRACKTOR=False, FG_IOU_THRESHOLD=0.65, BG_IOU_THRESHOLD=0.35, PROPOSAL_PER_IMAGE=256, TRACK_THRESH=case["thresholds"][0],
            START_TRACK_THRESH=case["thresholds"][1], RESUME_TRACK_THRESH=case["thresholds"][2])
    b = case.get("box_head", dict(resolution=7, sampling_ratio=2, mlp_dim=64, num_classes=2, score_thresh=0.05,
                                  nms=0.5, reg_weights=(10.0, 10.0, 5.0, 5.0)))
    return Cfg(INPUT=ns(AMODAL=bool(case.get("amodal", False))), TEST=ns(BBOX_AUG=ns(ENABLED=False)),
               MODEL=ns(TRACK_ON=True, RPN_ONLY=False, CLS_AGNOSTIC_BBOX_REG=False,
                        BACKBONE=ns(CONV_BODY="DLA-34-FPN"), DLA=ns(BACKBONE_OUT_CHANNELS=case["channels"]),
                        ROI_HEADS=ns(USE_FPN=True, BBOX_REG_WEIGHTS=b["reg_weights"], SCORE_THRESH=b["score_thresh"],
                                     NMS=b["nms"], DETECTIONS_PER_IMG=100),
                        ROI_BOX_HEAD=ns(POOLER_RESOLUTION=b["resolution"], POOLER_SCALES=th.POOLER_SCALES,
                                        POOLER_SAMPLING_RATIO=b["sampling_ratio"], MLP_HEAD_DIM=b["mlp_dim"],
                                        NUM_CLASSES=b["num_classes"]),
                        TRACK_HEAD=th))


class BoxSwitch(torch.nn.Module):
    """The ``box`` head of CombinedROIHeads: detector outputs pass, proposals go to ``real`` (None: score band only)."""

    def __init__(self, real):
        super().__init__()
        self.real = real
        self.refined = None

    def forward(self, features, proposals, targets=None):
        p = proposals[0]
        if getattr(p, "detector_output", False):
            return features, proposals, {}
        if self.real is None:
            out = BoxList(p.bbox.clone(), p.size, mode=p.mode)
            for f in p.fields():
                out.add_field(f, p.get_field(f).clone())
            out.add_field("scores", p.get_field("scores") + 1.0)
            return features, [out], {}
        x, res, _ = self.real(features, proposals)
        self.refined = res[0]
        return x, res, {}


def install_stubs(case):
    gr.install_stubs()                     # gen_golden stubs + boxlist_ops + box-head factories
    # target_sampler.py (imported by build_track_head, training only) constructs a Matcher
    sys.modules["maskrcnn_benchmark.modeling.matcher"].Matcher = type("Matcher", (object,), {
        "__init__": lambda self, *a, **k: None})
    if case["refine"]:
        b = case["box_head"]

        def mod(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
        mod("maskrcnn_benchmark.modeling.roi_heads.box_head.roi_box_feature_extractors",
            make_roi_box_feature_extractor=lambda cfg, in_channels: BO.FPN2MLPFeatureExtractor(
                in_channels, b["resolution"], (0.25, 0.125, 0.0625, 0.03125), b["sampling_ratio"], b["mlp_dim"]))
        mod("maskrcnn_benchmark.modeling.roi_heads.box_head.roi_box_predictors",
            make_roi_box_predictor=lambda cfg, dim: BO.FPNPredictor(dim, b["num_classes"]))


def det_boxlist(boxes, scores, image_wh, labels=None):
    bl = BoxList(torch.from_numpy(boxes.copy()), image_wh, mode="xyxy")
    bl.add_field("ids", torch.full((len(boxes),), -1, dtype=torch.int64))
    bl.add_field("labels", torch.ones(len(boxes), dtype=torch.int64) if labels is None else torch.from_numpy(labels.copy()))
    bl.add_field("scores", torch.from_numpy(scores.copy()))
    bl.detector_output = True
    return bl


def run_case(name, save=True):
    case = gi.SEQ_CASES[name]
    install_stubs(case)
    for m in [k for k in sys.modules if k == "siammot" or k.startswith("siammot.")]:
        del sys.modules[m]
    from siammot.modelling import roi_heads as ref_heads
    from siammot.modelling.box_head.box_head import ROIBoxHead
    from siammot.modelling.track_head.EMM import track_core as ref_core
    from siammot.modelling.track_head.track_head import build_track_head
    from siammot.modelling.track_head.track_solver import builder_tracker_solver
    from siammot.modelling.track_head.track_utils import build_track_utils
    from siammot.utils import registry as ref_registry
    torch.set_grad_enabled(False)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    inp = gi.SequenceInputs(name)
    cfg = reference_cfg(case)
    track_utils, track_pool = build_track_utils(cfg)
    track = build_track_head(cfg, track_utils, track_pool).eval()
    track.tracker.predictor.load_state_dict({k: torch.from_numpy(v) for k, v in inp.params.items()})
    solver = builder_tracker_solver(cfg, track_pool)
    real_box = None
    if case["refine"]:
        real_box = ROIBoxHead(cfg, case["channels"]).eval()
        real_box.load_state_dict({k: torch.from_numpy(v) for k, v in inp.box_head_params.items()})
    box = BoxSwitch(real_box)
    heads = ref_heads.CombinedROIHeads(cfg, [("box", box), ("track", track), ("solver", solver)]).eval()

    captured = {}
    real_argmax = torch.argmax
    fam = gi.BENCH_FAMILIES[case.get("family", "default")]
    grid = 16 * (int(fam["rz"] * fam["search_region"]) - fam["rz"] + 1)       # 256 (default family), 464 (AOT)

    def spy(t, *a, **k):
        r = real_argmax(t, *a, **k)
        if t.dim() == 2 and t.shape[1] == grid * grid:                  # decode_response's score map (track_core.py:120)
            top2 = torch.topk(t, 2, dim=1).values
            captured.update(idx=r.clone(), margin=(top2[:, 0] - top2[:, 1]).clone())
        return r

    real_tracker_forward = track.tracker.forward

    def tracker_forward(features, boxes, sr, targets=None, template_features=None):
        torch.argmax = spy
        ref_core.torch.argmax = spy
        try:
            out = real_tracker_forward(features, boxes, sr, targets=targets, template_features=template_features)
        finally:
            torch.argmax = real_argmax
        r = out[1][0]
        captured.update(trk_boxes=r.bbox.clone(), trk_scores=r.get_field("scores").clone(),
                        trk_ids=r.get_field("ids").clone(), tpl_boxes=boxes[0].bbox.clone(), sr_boxes=sr[0].bbox.clone())
        return out
    track.tracker.forward = tracker_forward

    out = {"n_frames": np.int64(case["frames"]),
           "param_checksum": gg_checksum([torch.from_numpy(v) for _, v in sorted(inp.params.items())])}
    memory = None
    events = dict(start=0, suspend=0, resume=0, expire=0)
    t0 = time.time()
    for f in range(case["frames"]):
        feats = [torch.from_numpy(a) for a in inp.features(f)]
        db, ds, dl = inp.detections(f, labels=True)
        dets = [det_boxlist(db, ds, case["image_wh"], dl)]
        captured.clear()
        box.refined = None
        prev_active = set(track_pool._active_ids)
        prev_dormant = set(track_pool._dormant_ids)
        prev_max = track_pool._max_id
        if case.get("given_detections"):
            # INFERENCE.USE_GIVEN_DETECTIONS: the cached detections arrive as `given_detection` (roi_heads.py:23-32, rcnn.py:54)
            memory, result, _ = heads(feats, None, track_memory=memory, given_detection=dets)
        else:
            memory, result, _ = heads(feats, dets, track_memory=memory)    # roi_heads.py:22 (rcnn.py:54 passes these)
        res = result[0]
        act, dorm = set(track_pool._active_ids), dict(track_pool._dormant_ids)
        events["start"] += track_pool._max_id - prev_max
        events["suspend"] += len((prev_active & set(dorm)))
        events["resume"] += len(prev_dormant & act)
        events["expire"] += len(prev_dormant - act - set(dorm))
        p = "f%02d_" % f
        out[p + "feat_checksum"] = gg_checksum(feats)
        out[p + "det_boxes"] = db
        out[p + "det_scores"] = ds
        out[p + "out_boxes"] = res.bbox.numpy().copy()
        out[p + "out_ids"] = res.get_field("ids").numpy().copy()
        out[p + "out_scores"] = res.get_field("scores").numpy().copy()
        out[p + "out_labels"] = res.get_field("labels").numpy().copy()
        out[p + "mem_ids"] = memory[2][0].get_field("ids").numpy().copy()
        out[p + "mem_boxes"] = memory[2][0].bbox.numpy().copy()
        out[p + "mem_sr"] = memory[1][0].bbox.numpy().copy()
        out[p + "pool_active"] = np.array(sorted(act), dtype=np.int64)
        out[p + "pool_dormant"] = np.array(sorted(dorm.items()), dtype=np.int64).reshape(-1, 2)
        out[p + "pool_max_id"] = np.int64(track_pool._max_id)
        if "trk_boxes" in captured:
            out[p + "trk_boxes"] = captured["trk_boxes"].numpy()
            out[p + "trk_scores"] = captured["trk_scores"].numpy()
            out[p + "trk_ids"] = captured["trk_ids"].numpy()
            out[p + "trk_idx"] = captured["idx"].numpy().astype(np.int64)
            out[p + "trk_margin"] = captured["margin"].numpy()
            out[p + "trk_tpl_boxes"] = captured["tpl_boxes"].numpy()
            out[p + "trk_sr_boxes"] = captured["sr_boxes"].numpy()
        if box.refined is not None:
            out[p + "ref_boxes"] = box.refined.bbox.numpy().copy()
            out[p + "ref_scores"] = box.refined.get_field("scores").numpy().copy()
            if case.get("n_foreground", 1) > 1:      # the box head regroups its rows by class (box_head/inference.py:164-191)
                out[p + "ref_ids"] = box.refined.get_field("ids").numpy().copy()
                out[p + "ref_labels"] = box.refined.get_field("labels").numpy().copy()
        n_trk = len(captured.get("trk_ids", ()))
        print("%s f%02d: dets %2d, tracked %2d (min margin %s, scores %s), out %2d, active %2d, dormant %2d, "
              "max id %d, %.0f s" % (
                  name, f, len(db), n_trk,
                  "%.2e" % float(captured["margin"].min()) if n_trk else "-",
                  "%.2f..%.2f" % (float(captured["trk_scores"].min()), float(captured["trk_scores"].max())) if n_trk else "-",
                  len(res), len(act), len(dorm), track_pool._max_id, time.time() - t0), flush=True)
    print(name, "events:", events)
    for k, v in events.items():
        out["events_" + k] = np.int64(v)
    margins = np.concatenate([v for k, v in out.items() if k.endswith("trk_margin")])
    print(name, "seed", case["seed"], "arg-max decisions", len(margins), "min margin %.3e" % margins.min(),
          "below 2e-6:", int((margins < 2e-6).sum()), flush=True)
    if save:
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sequence_%s.npz" % name), **out)
    return float(margins.min())


def gg_checksum(tensors):
    return np.array([float(t.double().sum()) for t in tensors] + [float(t.double().abs().sum()) for t in tensors])


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--probe-seeds":          # python gen_golden_sequence.py --probe-seeds plain 77 78 ...
        for seed in sys.argv[3:]:
            gi.SEQ_CASES[sys.argv[2]]["seed"] = int(seed)
            run_case(sys.argv[2], save=False)
    else:
        for case_name in (sys.argv[1:] or list(gi.SEQ_CASES)):
            run_case(case_name)
