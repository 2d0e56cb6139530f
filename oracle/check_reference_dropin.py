#!/usr/bin/env python
"""Exercise INTEGRATION.md §3 for real: the REFERENCE's own ``build_track_head`` / ``TrackHead`` /
``TrackSolver`` / ``TrackPool`` (imported unmodified from /root/reference; only the absent ``maskrcnn_benchmark``
symbols are stubbed, as in gen_golden.py) constructed with ``siammot_amd.emm.EMM`` swapped in through the reference's
registry.  No kernel is launched (CPU-only container): what is checked is the plumbing a maintainer relies on —

  * ``hip_emm.register("EMM", override=True)`` lands in the dict ``build_track_head`` reads (track_head.py:113-126);
  * the reference ``TrackHead`` then holds ``siammot_amd.emm.EMM``, built from the reference's cfg node and the
    reference's ``TrackUtils`` object;
  * the reference's ``predictor.*`` state_dict loads into it key for key (what DetectronCheckpointer does);
  * ``TrackHead.forward`` in eval mode on the first frame (no track memory) takes the reference's own path
    (reset the pool, return ``({}, None, {})``) and ``get_track_memory`` with no active track builds the empty
    memory with the reference's code (track_head.py:54-67) — both without touching the tracker;
  * with track memory it calls ``EMM.forward(features, template_boxes, sr=..., template_features=...)`` — the call is
    intercepted here and its argument structure checked (the kernels themselves are covered by ``pytest -m gpu``).

Prints one JSON line; exit code 0 on success.  Build container only (needs /root/reference).
"""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import golden_inputs as gi            # noqa: E402
import gen_golden as gg               # noqa: E402


def main():
    gg.install_stubs()                                           # BoxList, cat, LevelMapper, ROIAlign, make_conv3x3, Registry
    from siammot_amd.structures import BoxList, cat_boxlist
    from gen_golden_solver import boxlist_nms                    # CPU restatement of upstream NMS (no device here)
    boxlist_iou = None                                           # training-only symbol of target_sampler.py

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    mod("maskrcnn_benchmark.structures.boxlist_ops", cat_boxlist=cat_boxlist, boxlist_iou=boxlist_iou,
        boxlist_nms=boxlist_nms)
    mod("maskrcnn_benchmark.modeling.matcher", Matcher=type("Matcher", (object,), {
        "__init__": lambda self, *a, **k: None}))

    # --- the reference's own modules, the order INTEGRATION.md §3 prescribes ---------------------------------
    import siammot.modelling.track_head.EMM.track_core as ref_core           # registers the reference "EMM"
    import siammot.modelling.track_head.EMM.target_sampler                   # noqa: F401  TRACKER_SAMPLER["EMM"]
    from siammot.utils import registry as ref_registry
    ref_emm_cls = ref_registry.SIAMESE_TRACKER["EMM"]
    assert ref_emm_cls is ref_core.EMM
    import siammot_amd.emm as hip_emm
    import siammot_amd.registry as our_registry
    assert our_registry.SIAMESE_TRACKER is ref_registry.SIAMESE_TRACKER      # one dict, the reference's
    assert ref_registry.SIAMESE_TRACKER["EMM_HIP"] is hip_emm.EMM
    hip_emm.register("EMM", override=True)
    assert ref_registry.SIAMESE_TRACKER["EMM"] is hip_emm.EMM

    from siammot.modelling.track_head.track_head import TrackHead, build_track_head
    from siammot.modelling.track_head.track_utils import build_track_utils
    from siammot.modelling.track_head.track_solver import builder_tracker_solver

    case = gi.EMM_CASES["default"]
    cfg = gg.reference_cfg(case)
    cfg.MODEL.TRACK_HEAD.MODEL = "EMM"
    cfg.MODEL.TRACK_HEAD.PROPOSAL_PER_IMAGE = 256
    cfg.MODEL.TRACK_HEAD.TRACK_THRESH = 0.4
    cfg.MODEL.TRACK_HEAD.START_TRACK_THRESH = 0.6
    cfg.MODEL.TRACK_HEAD.RESUME_TRACK_THRESH = 0.4
    cfg.MODEL.TRACK_HEAD.EMM.HN_RATIO = 0.25
    cfg.MODEL.TRACK_HEAD.EMM.POS_RATIO = 0.25
    cfg.MODEL.TRACK_HEAD.FG_IOU_THRESHOLD = 0.65
    cfg.MODEL.TRACK_HEAD.BG_IOU_THRESHOLD = 0.35
    cfg.MODEL.TRACK_HEAD.EMM.CLS_POS_REGION = 0.8
    track_utils, track_pool = build_track_utils(cfg)                          # the reference's objects
    head = build_track_head(cfg, track_utils, track_pool)                     # the reference's builder
    assert type(head) is TrackHead and type(head.tracker) is hip_emm.EMM
    assert head.tracker.track_utils is track_utils
    solver = builder_tracker_solver(cfg, track_pool)

    # reference weights -> HIP module, key for key
    ref_emm = ref_emm_cls(cfg, track_utils)
    sd = {k: v for k, v in ref_emm.state_dict().items()}
    missing = head.tracker.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert sorted(head.tracker.state_dict().keys()) == sorted(sd.keys())

    head.eval()
    W, H = case["image_wh"]
    feats = tuple(torch.zeros(s) for s in gi.feature_shapes((W, H), case["channels"]))
    out = head(feats, track_memory=None)                                     # first frame: reference path, no tracker
    assert out == ({}, None, {})
    det = gg.boxlist(gi.emm_case_inputs("default")["boxes"][:3], (W, H))
    det.add_field("scores", torch.tensor([0.9, 0.8, 0.3]))
    det.add_field("ids", torch.full((3,), -1, dtype=torch.int64))
    dets = solver([det])                                                     # the reference solver starts tracks
    started = dets[0].get_field("ids").tolist()
    assert sorted(i for i in started if i >= 0) == [0, 1]

    # with active tracks the reference TrackHead calls tracker.extract_cache / tracker(...): intercept the calls
    calls = {}

    def fake_extract_cache(features, detection):
        calls["extract_cache"] = (len(features), len(detection))
        n = len(detection)
        sr = BoxList(detection.bbox + 512.0, (W + 1024, H + 1024), mode="xyxy")
        for f in detection.fields():
            sr.add_field(f, detection.get_field(f))
        return torch.zeros(n, case["channels"], 15, 15), [sr], [detection]

    def fake_forward(features, boxes, sr, targets=None, template_features=None):
        calls["forward"] = dict(n_feat=len(features), n_boxes=len(boxes[0]), sr_size=tuple(sr[0].size),
                                z_shape=tuple(template_features.shape))
        return {}, boxes, {}
    head.tracker.extract_cache = fake_extract_cache
    head.tracker.forward = fake_forward
    memory = head.get_track_memory(feats, dets)                              # reference code, track_head.py:54-75
    assert calls["extract_cache"] == (5, 2)
    _, tracks, _ = head(feats, track_memory=memory)                          # reference code, track_head.py:37-46
    assert calls["forward"] == dict(n_feat=5, n_boxes=2, sr_size=(W + 1024, H + 1024),
                                    z_shape=(2, case["channels"], 15, 15))
    print(json.dumps({"ok": True, "tracker_class": "%s.%s" % (type(head.tracker).__module__, type(head.tracker).__name__),
                      "track_head_class": "%s.%s" % (TrackHead.__module__, TrackHead.__name__),
                      "state_dict_keys": len(sd), "tracks_started": started}))


if __name__ == "__main__":
    main()
