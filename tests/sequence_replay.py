"""Replay of the closed-loop golden sequences (tests/golden/sequence_<case>.npz, written by
oracle/gen_golden_sequence.py from the reference's own CombinedROIHeads / TrackHead / TrackSolver / TrackPool / EMM)
through this repository's ``TrackingLoop``.

``OracleEMM`` is the CPU stand-in for the head in the ``-m "not gpu"`` suite: the oracle restatement behind the
``EMM`` interface (test infrastructure only — the product never imports it).
"""
import os

import numpy as np
import torch

import golden_inputs as gi
from siammot_amd.config import get_default_cfg
from siammot_amd.structures import BoxList
from siammot_amd.track_head import TrackingLoop as TrackingLoopBase

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sequence_cfg(case):
    cfg = get_default_cfg(channels=case["channels"])
    th = cfg.MODEL.TRACK_HEAD
    th.TRACK_THRESH, th.START_TRACK_THRESH, th.RESUME_TRACK_THRESH = case["thresholds"]
    th.MAX_DORMANT_FRAMES = case["max_dormant_frames"]
    fam = gi.BENCH_FAMILIES[case.get("family", "default")]           # the yaml family's track-head keys
    th.POOLER_RESOLUTION, th.SEARCH_REGION, th.PAD_PIXELS = fam["rz"], fam["search_region"], fam["pad_pixels"]
    th.MINIMUM_SREACH_REGION, th.POOLER_SCALES = fam["min_search_wh"], fam["scales"]
    th.EMM.USE_CENTERNESS, th.EMM.COSINE_WINDOW_WEIGHT = fam["use_centerness"], fam["sigma"]
    cfg.INPUT.AMODAL = bool(case.get("amodal", False))
    if case["refine"]:
        b = case["box_head"]
        cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM = b["mlp_dim"]
        cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES = b["num_classes"]
        cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION = b["resolution"]
        cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO = b["sampling_ratio"]
        cfg.MODEL.ROI_HEADS.SCORE_THRESH = b["score_thresh"]
        cfg.MODEL.ROI_HEADS.NMS = b["nms"]
        cfg.MODEL.ROI_HEADS.BBOX_REG_WEIGHTS = b["reg_weights"]
    return cfg


class OracleEMM(torch.nn.Module):
    """oracle/emm_oracle.py behind ``EMM.forward`` / ``EMM.extract_cache`` (CPU tensors)."""

    def __init__(self, params, channels, track_utils, reference_ops=True, case=None):
        super(OracleEMM, self).__init__()
        self.reference_ops = reference_ops
        from oracle import emm_oracle as O
        self.O = O
        fam = gi.BENCH_FAMILIES[(case or {}).get("family", "default")]
        self.ocfg = O.EMMConfig(channels=channels, rz=fam["rz"], search_region=fam["search_region"], scales=fam["scales"],
                                pad_pixels=fam["pad_pixels"], min_search_wh=fam["min_search_wh"],
                                use_centerness=fam["use_centerness"], sigma=fam["sigma"],
                                amodal=bool((case or {}).get("amodal", False)))
        self.params = {k: torch.from_numpy(v) for k, v in params.items()}
        self.track_utils = track_utils
        self.last = None

    def forward(self, features, boxes, sr, targets=None, template_features=None):
        b = boxes[0]
        bb, conf, _, inter = self.O.emm_forward(self.ocfg, self.params, list(features), b.bbox, sr[0].bbox,
                                                template_features, b.size, return_intermediates=True,
                                                reference_ops=self.reference_ops)
        self.last = inter
        out = b.__class__(bb, b.size, mode="xyxy")
        out.add_field("ids", b.get_field("ids"))
        out.add_field("labels", b.get_field("labels"))
        out.add_field("scores", conf)
        return {}, [out], {}

    def extract_cache(self, features, detection):
        z, sr_bbox = self.O.extract_cache(self.ocfg, list(features), detection.bbox)
        w, h = detection.size
        pad = self.track_utils.pad_pixels
        sr = detection.__class__(sr_bbox, [int(w + 2 * pad), int(h + 2 * pad)], mode="xyxy")
        for f in detection.fields():
            sr.add_field(f, detection.get_field(f))
        return z, [sr], [detection]


class RecordedEMM(torch.nn.Module):
    """The reference's OWN head outputs, replayed: ``forward`` returns the tracked boxes / scores the golden stores for
    the current frame (``trk_boxes`` / ``trk_scores``, after checking that it is asked about the same ids and template
    boxes), ``extract_cache`` the search regions (oracle arithmetic) and placeholder templates.  With it the host glue of
    a whole sequence — TrackHead, solver, pool, dormant-row order, expiry — is pinned against the reference on CPU in
    seconds, whatever the sequence's length and row count (``replay(..., features=False)``; set ``.t`` to the frame)."""

    def __init__(self, golden, track_utils, case):
        super(RecordedEMM, self).__init__()
        self.golden, self.track_utils, self.case, self.t = golden, track_utils, case, 0
        fam = gi.BENCH_FAMILIES[case.get("family", "default")]
        self.e, self.min_wh = fam["search_region"] - 1.0, fam["min_search_wh"]
        self.calls = 0

    def forward(self, features, boxes, sr, targets=None, template_features=None):
        p = "f%02d_" % self.t
        b = boxes[0]
        g = self.golden
        assert b.get_field("ids").tolist() == g[p + "trk_ids"].tolist(), "frame %d: rows handed to the head" % self.t
        assert np.abs(b.bbox.numpy() - g[p + "trk_tpl_boxes"]).max() < 1e-3, "frame %d: template boxes" % self.t
        assert np.abs(sr[0].bbox.numpy() - g[p + "trk_sr_boxes"]).max() < 1e-2, "frame %d: search regions" % self.t
        assert template_features.shape[0] == len(b)
        self.calls += 1
        out = b.__class__(torch.from_numpy(g[p + "trk_boxes"].copy()), b.size, mode="xyxy")
        out.add_field("ids", b.get_field("ids"))
        out.add_field("labels", b.get_field("labels"))
        out.add_field("scores", torch.from_numpy(g[p + "trk_scores"].copy()))
        return {}, [out], {}

    def extract_cache(self, features, detection):
        pad = self.track_utils.pad_pixels
        sr_bbox = torch.from_numpy(gi.np_search_region(detection.bbox.numpy(), pad, self.e, self.min_wh))
        w, h = detection.size
        sr = detection.__class__(sr_bbox, [int(w + 2 * pad), int(h + 2 * pad)], mode="xyxy")
        for f in detection.fields():
            sr.add_field(f, detection.get_field(f))
        # one float per row, tagged with the row's id: a dormant row's entry must come back as it went in
        z = detection.get_field("ids").to(torch.float32).view(-1, 1, 1, 1).clone()
        return z, [sr], [detection]


class RecordedRefine(object):
    """The reference's own box-head outputs for the propagated tracks, replayed (``ref_boxes`` / ``ref_scores`` of the
    golden — the latter are the box head's scores, in its output order; ids / labels follow the class regrouping of
    box_head/inference.py:164-191, which the golden's solver inputs reflect)."""

    def __init__(self, golden, emm, num_classes):
        self.golden, self.emm, self.num_classes = golden, emm, num_classes

    def __call__(self, features, tracks):
        t = tracks[0]
        if len(t) == 0:
            return tracks
        p = "f%02d_" % self.emm.t
        g = self.golden
        track_scores = t.get_field("scores") + 1.0                                  # roi_heads.py:67
        labels, ids = t.get_field("labels"), t.get_field("ids")
        order = torch.cat([torch.nonzero(labels == j).squeeze(1) for j in range(1, self.num_classes)])   # filter_results
        out = t.__class__(torch.from_numpy(g[p + "ref_boxes"].copy()), t.size, mode="xyxy")
        det_scores = torch.from_numpy(g[p + "ref_scores"].copy())
        out.add_field("scores", (det_scores + track_scores) / 2.0)                  # :73-76: input order + output order
        out.add_field("ids", ids[order])
        out.add_field("labels", labels[order])
        return [out]


def iou_rows(a, b):
    """IoU of matching rows (continuous boxes, no +1)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    iw = np.clip(np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0]), 0, None)
    ih = np.clip(np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]), 0, None)
    inter = iw * ih
    ua = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter
    return np.where(ua > 0, inter / np.maximum(ua, 1e-30), 1.0)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, "sequence_%s.npz" % name))


def detections_boxlist(inp, t, device, boxlist_cls=BoxList):
    boxes, scores, labels = inp.detections(t, labels=True)
    bl = boxlist_cls(torch.from_numpy(boxes).to(device), inp.case["image_wh"], mode="xyxy")
    bl.add_field("ids", torch.full((len(boxes),), -1, dtype=torch.int64, device=device))
    bl.add_field("labels", torch.from_numpy(labels).to(device))
    bl.add_field("scores", torch.from_numpy(scores).to(device))
    return bl


TAINTED_IOU = {"crowd": 0.90, "crowd512": 0.90}     # IoU floor of a track downstream of an attributed flip (0.97 elsewhere): a
                       # template that moved by one arg-max cell (0.25-0.5 px) follows its object from there on with an offset;
                       # on the crowd cases' 32-90 px boxes that offset is 0.5-1.5 % of a side per cell, and the box head
                       # regresses from the shifted box (measured: 0.9675 one frame after a flip at a stored margin of 2.4e-6)
FLIP_MARGIN = 3e-6     # an arg-max of the reference whose best and second-best penalised scores are closer than this
                       # can fall on the other cell in another fp32 implementation (tower summation order);
                       # measured flips sit at margins of 1e-7 (profiles/r02_argmax_stats_n100.md)


def probe_tracker(emm):
    """Record the raw output of the head (before refinement / solver) of every frame: wraps ``forward`` of the
    instance and its ``track_raw`` (the Python-composed lean path); the frame entry point hands the rows over through
    ``ProbedTrackingLoop`` (BEFORE anything consumes the head's output).  Returns a dict whose ``last`` entry is ``(boxes, scores)`` or None.

    Tie alignment (round 5).  ``replay`` leaves the reference's raw output of the frame in ``expect`` = (boxes, scores,
    stored arg-max margins, allowed margin): a row that lands on ANOTHER arg-max cell (> 0.05 px away) although nothing
    else differs is an fp32 tie when the reference's own margin between its best and second-best cell is below the
    allowed one (FLIP_MARGIN, 3e-6) — the probe then writes the reference's box and score into that row, in place,
    before refinement / solver read it, and reports it in ``forced``.  Without this a closed loop cannot be compared
    beyond its first tie: in the crowd sequences (8 k arg-max decisions, two dozen with margins under 2e-6) a template
    that moved by half a pixel changes an NMS decision of a NEIGHBOUR within two frames and the two trajectories part
    for good — legitimately, but the frames behind it (where the capacity fallbacks live) would go uncompared.  Rows
    whose margin is not a tie are left alone and fail the comparison."""
    box = {"last": None, "expect": None, "forced": [], "frame": -1}

    def align(bb, conf):
        exp = box["expect"]
        if exp is not None and bb.shape[0] == exp[0].shape[0]:
            gb, gs, margin, allowed, gid = exp
            err = np.abs(bb.detach().cpu().numpy() - gb).max(axis=1)
            rows = [r for r in np.nonzero(err > 0.05)[0].tolist() if margin[r] < allowed]
            if rows:
                idx = torch.tensor(rows, dtype=torch.int64, device=bb.device)
                bb[idx] = torch.from_numpy(gb[rows]).to(bb.device)
                conf[idx] = torch.from_numpy(gs[rows]).to(conf.device)
                box["forced"].extend((box["frame"], int(gid[r]), float(margin[r]), float(err[r])) for r in rows)
        box["last"] = (bb.clone(), conf.clone())                     # (the solver bands scores in place afterwards)
    fwd = emm.forward

    def forward(features, boxes, sr, targets=None, template_features=None):
        out = fwd(features, boxes, sr, targets=targets, template_features=template_features)
        align(out[1][0].bbox, out[1][0].get_field("scores"))
        return out
    emm.forward = forward
    tr = getattr(emm, "track_raw", None)  # the raw-tensor form the Python-composed lean path calls (the HIP head only)
    if tr is not None:
        def track_raw(*a, **k):
            out = tr(*a, **k)
            align(out[0], out[1])
            return out
        emm.track_raw = track_raw
    box["align"] = align                  # the frame entry point's seam: ProbedTrackingLoop (below) calls it
    return box


class ProbedTrackingLoop(TrackingLoopBase):
    """Test-only subclass of the product's TrackingLoop: inside the frame entry point (head, refinement and solver behind
    two library calls) the raw rows of the head and of the box head are handed to the probes of this module — in stream
    order, before anything consumes them.  The product class consults no callback; an instance is switched to this class
    by ``probe_loop``."""
    _probed = True

    def _head_output_enqueued(self, boxes, scores):
        p = self.__dict__.get("_probe_tracker")
        if p is not None:
            p["align"](boxes, scores)

    def _refined_output_enqueued(self, boxes, scores, ids, labels):
        p = self.__dict__.get("_probe_box")
        if p is not None:
            p["hook"](boxes, scores, ids, labels)


def probe_loop(loop, tracker_probe=None, box_probe=None):
    """Switch ``loop`` (a product TrackingLoop) to the probed subclass and attach the probes of ``probe_tracker`` /
    ``probe_box_head``."""
    loop.__class__ = ProbedTrackingLoop
    loop.__dict__["_probe_tracker"] = tracker_probe
    loop.__dict__["_probe_box"] = box_probe
    return loop


def probe_box_head(refine):
    """Record what the box head returns for the propagated tracks in every frame: wraps ``forward`` of
    ``RefineTracks.box`` (boxes, box-head scores, ids) and its ``refine_raw`` (the device-only form; inside the frame entry
    point: ``ProbedTrackingLoop``) — (boxes, None, ids: their scores are averaged already)."""
    rec = {"last": None}
    head = refine.box
    fwd = head.forward

    def forward(features, proposals, targets=None):
        out = fwd(features, proposals)
        rec["last"] = (out[1][0].bbox.clone(), out[1][0].get_field("scores").clone(), out[1][0].get_field("ids").clone())
        return out
    head.forward = forward

    def hook(bb, scores, ids, labels):
        rec["last"] = (bb.clone(), None, ids.clone())
    rr = getattr(head, "refine_raw", None)   # the device-only form the Python-composed lean path calls
    if rr is not None:
        def refine_raw(*a, **k):
            out = rr(*a, **k)
            hook(*out)
            return out
        head.refine_raw = refine_raw
    rec["hook"] = hook                    # the frame entry point's seam: ProbedTrackingLoop
    return rec


SCORE_TOL = 1e-3       # scores in the closed loop.  Single frame pairs agree to 1e-5; in the loop a box error of 1e-3 px
                       # moves the next template, and with the box head in the loop (its regression feeds the next
                       # template too) two fp32 CPU implementations of the SAME head already differ by 1.1e-4 after six
                       # frames (oracle with the reference's library calls vs the explicit restatements)


TIE_MARGIN_LONG = 1e-5      # the fixed tie margin of the long crowd / dormancy runs (44-100 frames, 8 k - 20 k decisions, the box
                            # head in the loop): closed-loop drift of a few 1e-4 in a raw score moves decisions whose stored
                            # margin is a few 1e-6 (measured: 6.5e-6 at most); FLIP_MARGIN everywhere else


def replay(loop, inp, golden, device, frames=None, on_frame=None, probe=None, box_probe=None, prefetch=False,
           features=True, before_frame=None, tie_margin=FLIP_MARGIN):
    """Run the loop over the sequence and compare every frame with the golden: ids, labels, pool state and memory
    ids must be IDENTICAL in every frame; boxes >= 1 - 1e-3 IoU, scores within 1e-4.  With ``probe``
    (``probe_tracker``) the raw head output is compared too, and a tracked row that lands one arg-max cell away from
    the reference's is accepted ONLY when the reference's own stored margin for that row is below ``tie_margin`` — a FIXED
    bound (round 6; rounds 3-5 widened it with the score difference measured so far in the replay): FLIP_MARGIN, or
    TIE_MARGIN_LONG for the long crowd / dormancy runs (errors accumulate in a closed loop; with a random-init box head in
    it — large regressions off noise features — two CPU fp32 implementations of the same head flip a 6e-6 margin after
    nine frames); that track id is then held to
    IoU >= 0.97 (``TAINTED_IOU``: 0.90 on the crowd cases' small boxes) from there on (its template moved by a cell) and
    reported in ``flips``.
    ``prefetch``: every call also gets the NEXT frame's feature maps (``TrackingLoop.forward(..., next_features=)``: the
    next head is launched speculatively behind this frame's extraction) — results must not change.
    Returns a dict of statistics; raises AssertionError (frame, row, stored margins) at the first divergence."""
    n_frames = int(golden["n_frames"]) if frames is None else frames
    pool = loop.solver.track_pool
    if isinstance(loop, TrackingLoopBase) and (probe is not None or box_probe is not None):
        probe_loop(loop, probe, box_probe)        # the frame entry point's raw rows reach the probes through the test subclass
    stats = dict(min_iou=1.0, max_box_err=0.0, max_score_err=0.0, rows=0, frames=n_frames, tracked_rows=0,
                 raw_rows=0, raw_max_box_err=0.0, raw_max_score_err=0.0, flips=[])
    tainted = set()
    loop.reset()
    ahead = None                  # (frame index, device tensors) of the frame prepared one call early (prefetch)
    for t in range(n_frames):
        p = "f%02d_" % t
        if before_frame is not None:
            before_frame(t)
        if not features:                 # (a recorded head: nothing reads the maps, only their device)
            feats = (torch.zeros(1),)
        else:
            feats_np = inp.features(t)
            chk = np.array([float(f.astype(np.float64).sum()) for f in feats_np] +
                           [float(np.abs(f.astype(np.float64)).sum()) for f in feats_np])
            np.testing.assert_allclose(chk, golden[p + "feat_checksum"], rtol=1e-9, err_msg="inputs drifted, frame %d" % t)
        if not features:
            pass
        elif ahead is not None and ahead[0] == t:
            feats = ahead[1]                                                      # the SAME tensors the last call was shown
        else:
            feats = tuple(torch.from_numpy(f).to(device) for f in feats_np)
        if probe is not None:
            probe["last"] = None
            probe["frame"] = t
            probe["expect"] = None
            if (p + "trk_margin") in golden.files:
                probe["expect"] = (golden[p + "trk_boxes"], golden[p + "trk_scores"], golden[p + "trk_margin"],
                                   tie_margin, golden[p + "trk_ids"])
        dets_t = detections_boxlist(inp, t, device, getattr(loop, "boxlist_cls", BoxList))
        if prefetch and t + 1 < n_frames:
            ahead = (t + 1, tuple(torch.from_numpy(f).to(device) for f in inp.features(t + 1)))
            out = loop(feats, dets_t, next_features=ahead[1])
        else:
            out = loop(feats, dets_t)
        has_trk = (p + "trk_margin") in golden.files
        margin = golden[p + "trk_margin"] if has_trk else np.array([np.inf])
        ctx = "case %s frame %d (min stored arg-max margin of the frame %.2e; flips so far %s)" % (
            inp.name, t, float(margin.min()), stats["flips"])
        # ---- the head's raw output (the tracked boxes before the solver) ------------------------------------------
        if probe is not None and has_trk:
            assert probe["last"] is not None, "the head did not run: " + ctx
            rb, rs = probe["last"][0].cpu().numpy(), probe["last"][1].cpu().numpy()
            rs = np.where(rs > 1.0, rs - np.floor(rs), rs)      # (the one-call frame's hook sees the score buffer after the
                                                                # solver banded it in place: + 1 per band)
            gb, gs, gid = golden[p + "trk_boxes"], golden[p + "trk_scores"], golden[p + "trk_ids"]
            assert rb.shape == gb.shape, "tracked rows %s vs %s: %s" % (rb.shape, gb.shape, ctx)
            err = np.abs(rb - gb).max(axis=1)
            for r in np.nonzero(err > 0.05)[0].tolist():
                tid = int(gid[r])
                if tid in tainted:
                    continue
                allowed = tie_margin
                assert margin[r] < allowed, "row %d (id %d) moved by %.3f px although the reference's arg-max margin " \
                    "is %.2e (allowed %.2e): %s" % (r, tid, err[r], margin[r], allowed, ctx)
                stats["flips"].append((t, tid, float(margin[r]), float(err[r])))
                tainted.add(tid)
            clean = np.array([int(i) not in tainted for i in gid])
            if clean.any():
                stats["raw_max_box_err"] = max(stats["raw_max_box_err"], float(err[clean].max()))
                stats["raw_max_score_err"] = max(stats["raw_max_score_err"], float(np.abs(rs - gs)[clean].max()))
            stats["raw_rows"] += len(gid)
        # ---- the box head's output for the propagated tracks (refinement on) ----------------------------------------
        if box_probe is not None and (p + "ref_boxes") in golden.files:
            assert box_probe["last"] is not None, "the box head did not run: " + ctx
            bb, bs, bi = [x.cpu().numpy() if x is not None else None for x in box_probe["last"]]
            if bs is None:
                bs = golden[p + "ref_scores"]
            # (rows come back in input order for one foreground class, regrouped by class otherwise: stored as `ref_ids`)
            want_rows = golden[p + "ref_ids"] if (p + "ref_ids") in golden.files else golden[p + "trk_ids"]
            assert bi.tolist() == want_rows.tolist(), "box-head rows: %s\n got %s\n ref %s" % (
                ctx, bi.tolist(), want_rows.tolist())
            cl = np.array([int(i) not in tainted for i in bi])
            be, se = np.abs(bb - golden[p + "ref_boxes"]).max(axis=1), np.abs(bs - golden[p + "ref_scores"])
            stats["box_head_max_box_err"] = max(stats.get("box_head_max_box_err", 0.0), float(be[cl].max(initial=0.0)))
            stats["box_head_max_score_err"] = max(stats.get("box_head_max_score_err", 0.0), float(se[cl].max(initial=0.0)))
            assert be[cl].max(initial=0.0) < 5e-2 and se[cl].max(initial=0.0) < SCORE_TOL, \
                "box head: box err %.3e px (row %d), score err %.3e (row %d, id %d: got %.6f ref %.6f): %s" % (
                    be.max(), int(be.argmax()), se.max(), int(se.argmax()), int(bi[se.argmax()]), bs[se.argmax()],
                    golden[p + "ref_scores"][se.argmax()], ctx)
            box_probe["last"] = None
        # ---- the frame's result -----------------------------------------------------------------------------------
        ids = out.get_field("ids").cpu().numpy()
        assert ids.tolist() == golden[p + "out_ids"].tolist(), "ids differ: %s\n got %s\n ref %s" % (
            ctx, ids.tolist(), golden[p + "out_ids"].tolist())
        boxes = out.bbox.cpu().numpy()
        scores = out.get_field("scores").cpu().numpy()
        if len(ids):
            iou = iou_rows(boxes, golden[p + "out_boxes"])
            clean = np.array([int(i) not in tainted for i in ids])
            if clean.any():
                stats["min_iou"] = min(stats["min_iou"], float(iou[clean].min()))
                stats["max_box_err"] = max(stats["max_box_err"], float(np.abs(boxes - golden[p + "out_boxes"])[clean].max()))
                assert iou[clean].min() >= 1 - 1e-3, "box IoU %.6f at row %d: %s" % (
                    iou[clean].min(), int(np.nonzero(clean)[0][iou[clean].argmin()]), ctx)
            assert iou.min() >= TAINTED_IOU.get(inp.name, 0.97), "box IoU %.4f of a track downstream of an attributed flip: %s" % (
                iou.min(), ctx)
            stats["max_score_err"] = max(stats["max_score_err"], float(np.abs(scores - golden[p + "out_scores"]).max()))
            sd = np.abs(scores - golden[p + "out_scores"])
            assert sd[clean].max(initial=0.0) < SCORE_TOL, "scores differ by %.3e at row %d (id %d: got %.6f, ref %.6f; box " \
                "err %.3e px): %s" % (sd.max(), int(sd.argmax()), int(ids[sd.argmax()]), scores[sd.argmax()],
                                      golden[p + "out_scores"][sd.argmax()],
                                      np.abs(boxes - golden[p + "out_boxes"])[sd.argmax()].max(), ctx)
            assert out.get_field("labels").cpu().numpy().tolist() == golden[p + "out_labels"].tolist(), ctx
        stats["rows"] += len(ids)
        stats["tracked_rows"] += int((ids >= 0).sum())
        # ---- pool state and the next frame's memory -----------------------------------------------------------------
        assert sorted(pool.get_active_ids()) == golden[p + "pool_active"].tolist(), "active ids: " + ctx
        dorm = sorted((int(k), int(v)) for k, v in pool._dormant_ids.items())
        assert dorm == [tuple(r) for r in golden[p + "pool_dormant"].tolist()], "dormant ids: " + ctx
        assert pool._max_id == int(golden[p + "pool_max_id"]), "max id: " + ctx
        mem = loop.track_memory
        mem_ids = mem[2][0].get_field("ids").cpu().numpy().tolist()
        assert mem_ids == golden[p + "mem_ids"].tolist(), "memory ids: %s\n got %s\n ref %s" % (
            ctx, mem_ids, golden[p + "mem_ids"].tolist())
        if len(mem_ids):
            assert mem[0].shape[0] == len(mem_ids) == len(mem[1][0]), "memory rows: " + ctx
            miou = iou_rows(mem[2][0].bbox.cpu().numpy(), golden[p + "mem_boxes"])
            mclean = np.array([i not in tainted for i in mem_ids])
            assert miou[mclean].min(initial=1.0) >= 1 - 1e-3 and miou.min() >= TAINTED_IOU.get(inp.name, 0.97), "memory boxes: " + ctx
            serr = np.abs(mem[1][0].bbox.cpu().numpy() - golden[p + "mem_sr"]).max(axis=1)
            assert serr[mclean].max(initial=0.0) < 0.2, "memory search regions: " + ctx
        if on_frame is not None:
            on_frame(t, out)
    if probe is not None:
        stats["flips"] = list(probe["forced"]) + stats["flips"]      # rows aligned to the reference's side of a tie + unaligned ones
    return stats
