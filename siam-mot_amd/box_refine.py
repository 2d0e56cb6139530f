"""The detector's half of the tracking step: ``_refine_tracks`` and the box head it calls.

``CombinedROIHeads._refine_tracks`` (reference siammot/modelling/roi_heads.py:60-84) sends the boxes the EMM head
propagated through the box head AS PROPOSALS and averages the two scores; ``RefineTracks`` is that function around any
box head with the reference's call signature ``box(features, [BoxList]) -> (x, [BoxList], {})`` — the reference's own
``ROIBoxHead`` or ``TrackBoxHead`` below — and is what ``TrackingLoop(refine_tracks=...)`` expects.

``TrackBoxHead`` restates ``ROIBoxHead`` at inference (siammot/modelling/box_head/box_head.py:10-56) with the pieces the
reference's yaml files select: [UPSTREAM] ``FPN2MLPFeatureExtractor`` (7x7 ``Pooler`` -> fc6 -> ReLU -> fc7 -> ReLU),
[UPSTREAM] ``FPNPredictor`` (``cls_score`` / ``bbox_pred``) and the reference's own ``PostProcessor``
(box_head/inference.py:11-185: soft-max, ``BoxCoder.decode``, track rows keep only their label's probability + 1 so
that they survive, clip, per-class threshold, NMS on detections only).  The pooler is this repository's HIP ROIAlign
(``poolers.Pooler``); the four ``Linear`` layers are library GEMMs (hipBLASLt through torch) — plain dense layers, not
hand kernels, as the north star assigns them.  Parameter names are upstream's (``feature_extractor.fc6.weight`` ...), so
the ``roi_heads.box.*`` entries of a reference checkpoint load with ``load_state_dict``.
"""
import math

import torch
from torch import nn

from . import ops
from .poolers import Pooler
from .structures import BoxList, boxlist_nms, cat_boxlist


_addmm_relu = getattr(torch, "_addmm_activation", None)      # bias + ReLU in the GEMM's epilogue (hipBLASLt) when offered


def _linear_relu(x, layer):
    if _addmm_relu is not None:
        return _addmm_relu(layer.bias, x, layer.weight.t())
    return torch.addmm(layer.bias, x, layer.weight.t()).relu_()


class BoxCoder(object):
    """[UPSTREAM] maskrcnn_benchmark/modeling/box_coder.py ``decode``: deltas (dx, dy, dw, dh) / weights applied to
    boxes measured with the legacy +1 width, ``dw`` / ``dh`` clamped at log(1000/16)."""

    def __init__(self, weights=(10.0, 10.0, 5.0, 5.0), bbox_xform_clip=math.log(1000.0 / 16)):
        self.weights = tuple(float(w) for w in weights)
        self.bbox_xform_clip = bbox_xform_clip

    def decode(self, rel_codes, boxes):
        boxes = boxes.to(rel_codes.dtype)
        widths = boxes[:, 2] - boxes[:, 0] + 1
        heights = boxes[:, 3] - boxes[:, 1] + 1
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
        pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
        pred_w = torch.exp(dw) * widths[:, None]
        pred_h = torch.exp(dh) * heights[:, None]
        out = torch.zeros_like(rel_codes)
        out[:, 0::4] = pred_ctr_x - 0.5 * pred_w
        out[:, 1::4] = pred_ctr_y - 0.5 * pred_h
        out[:, 2::4] = pred_ctr_x + 0.5 * pred_w - 1
        out[:, 3::4] = pred_ctr_y + 0.5 * pred_h - 1
        return out


class FPN2MLPFeatureExtractor(nn.Module):
    """[UPSTREAM] roi_box_feature_extractors.py ``FPN2MLPFeatureExtractor`` (no GroupNorm variant, as in the yamls)."""

    def __init__(self, in_channels, resolution, scales, sampling_ratio, representation_size, pooler=None):
        super(FPN2MLPFeatureExtractor, self).__init__()
        self.pooler = pooler if pooler is not None else Pooler((resolution, resolution), scales, sampling_ratio)
        self.fc6 = nn.Linear(in_channels * resolution * resolution, representation_size)
        self.fc7 = nn.Linear(representation_size, representation_size)
        self.out_channels = representation_size

    def forward(self, x, proposals):
        x = self.pooler(x, proposals)
        x = x.reshape(x.size(0), -1)
        x = torch.relu(self.fc6(x))
        return torch.relu(self.fc7(x))


class FPNPredictor(nn.Module):
    """[UPSTREAM] roi_box_predictors.py ``FPNPredictor``."""

    def __init__(self, representation_size, num_classes, cls_agnostic_bbox_reg=False):
        super(FPNPredictor, self).__init__()
        self.cls_score = nn.Linear(representation_size, num_classes)
        self.bbox_pred = nn.Linear(representation_size, (2 if cls_agnostic_bbox_reg else num_classes) * 4)

    def forward(self, x):
        return self.cls_score(x), self.bbox_pred(x)


class PostProcessor(nn.Module):
    """box_head/inference.py:11-185.  ``nms_fn(BoxList, thresh) -> BoxList`` defaults to the HIP NMS."""

    def __init__(self, score_thresh=0.05, nms=0.5, box_coder=None, cls_agnostic_bbox_reg=False,
                 amodal_inference=False, nms_fn=None):
        super(PostProcessor, self).__init__()
        self.score_thresh = score_thresh
        self.nms = nms
        self.box_coder = box_coder if box_coder is not None else BoxCoder()
        self.cls_agnostic_bbox_reg = cls_agnostic_bbox_reg
        self.amodal_inference = amodal_inference
        self.nms_fn = nms_fn if nms_fn is not None else boxlist_nms

    def forward(self, x, boxes):
        class_logits, box_regression = x
        if len(boxes) != 1:
            raise RuntimeError("PostProcessor: one image per call")
        box = boxes[0]
        prob = torch.softmax(class_logits, -1)
        device = class_logits.device
        if self.cls_agnostic_bbox_reg:
            box_regression = box_regression[:, -4:]
        proposals = self.box_coder.decode(box_regression.view(len(box), -1), box.bbox)
        if self.cls_agnostic_bbox_reg:
            proposals = proposals.repeat(1, prob.shape[1])
        num_classes = prob.shape[1]
        if box.has_field("ids"):                                                   # inference.py:85-90
            ids = box.get_field("ids")
        else:
            ids = torch.full((len(box),), -1, dtype=torch.int64, device=device)
        if box.has_field("labels"):                                                # inference.py:93-104: tracks
            labels = box.get_field("labels")
            is_track = ids >= 0
            keep = torch.zeros_like(prob)
            keep.scatter_(1, labels.view(-1, 1), 1.0)
            # a track row keeps only its own label's probability, moved to the (1, 2] band
            prob = torch.where(is_track[:, None], (prob + 1.0) * keep, prob)
        boxlist = BoxList(proposals.reshape(-1, 4), box.size, mode="xyxy")
        boxlist.add_field("scores", prob.reshape(-1))
        if not self.amodal_inference:
            boxlist = boxlist.clip_to_image(remove_empty=False)
        return [self.filter_results(boxlist, ids, num_classes)]

    def filter_results(self, boxlist, ids, num_classes):                           # inference.py:137-185
        boxes = boxlist.bbox.reshape(-1, num_classes * 4)
        scores = boxlist.get_field("scores").reshape(-1, num_classes)
        device = scores.device
        result = []
        inds_all = scores > self.score_thresh
        for j in range(1, num_classes):
            inds = inds_all[:, j].nonzero().squeeze(1)
            scores_j, boxes_j, ids_j = scores[inds, j], boxes[inds, j * 4:(j + 1) * 4], ids[inds]
            det_idx = ids_j < 0
            det = BoxList(boxes_j[det_idx], boxlist.size, mode="xyxy")
            det.add_field("scores", scores_j[det_idx])
            det.add_field("ids", ids_j[det_idx])
            if len(det) > 0:
                det = self.nms_fn(det, self.nms)
            trk_idx = ids_j >= 0
            if bool(trk_idx.any()):
                trk = BoxList(boxes_j[trk_idx], boxlist.size, mode="xyxy")
                trk.add_field("scores", scores_j[trk_idx])
                trk.add_field("ids", ids_j[trk_idx])
                det = cat_boxlist([det, trk])
            det.add_field("labels", torch.full((len(det),), j, dtype=torch.int64, device=device))
            result.append(det)
        return cat_boxlist(result)


class TrackBoxHead(nn.Module):
    """``ROIBoxHead`` at inference: ``forward(features, [proposals]) -> (x, [BoxList], {})``."""

    def __init__(self, cfg, in_channels, pooler=None, nms_fn=None):
        super(TrackBoxHead, self).__init__()
        head = cfg.MODEL.ROI_BOX_HEAD
        agnostic = bool(getattr(cfg.MODEL, "CLS_AGNOSTIC_BBOX_REG", False))
        self.feature_extractor = FPN2MLPFeatureExtractor(in_channels, head.POOLER_RESOLUTION, head.POOLER_SCALES,
                                                         head.POOLER_SAMPLING_RATIO, head.MLP_HEAD_DIM, pooler)
        self.predictor = FPNPredictor(head.MLP_HEAD_DIM, head.NUM_CLASSES, agnostic)
        rh = cfg.MODEL.ROI_HEADS
        self.post_processor = PostProcessor(rh.SCORE_THRESH, rh.NMS, BoxCoder(rh.BBOX_REG_WEIGHTS), agnostic,
                                            bool(cfg.INPUT.AMODAL), nms_fn)

    def forward(self, features, proposals, targets=None):
        if self.training:
            raise NotImplementedError("siammot_amd.TrackBoxHead is an inference path (box_head.py:39-43,52-61)")
        x = self.feature_extractor(features, proposals)
        return x, self.post_processor(self.predictor(x), proposals), {}

    # ---- the propagated tracks on raw device tensors: no BoxList, no host synchronisation -----------------------------
    def _track_weights(self):
        """``cls_score`` and ``bbox_pred`` as ONE [K + 4*KR, dim] GEMM operand (+ bias), rebuilt when either parameter
        changes (load_state_dict / .to() bump the version counters or replace the storage)."""
        cs, bp = self.predictor.cls_score, self.predictor.bbox_pred
        stamp = tuple((t.data_ptr(), t._version) for t in (cs.weight, cs.bias, bp.weight, bp.bias))
        hit = self.__dict__.get("_track_w")
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                hit = (stamp, torch.cat((cs.weight, bp.weight), 0).t().contiguous(), torch.cat((cs.bias, bp.bias), 0))
            self.__dict__["_track_w"] = hit
        return hit[1], hit[2]

    def one_call_ok(self, n):
        """The whole refinement of ``n`` rows fits ``smot_box_refine_fwd`` (and so the one-call tracking frame)."""
        fe, pooler = self.feature_extractor, self.feature_extractor.pooler
        return (self.raw_ok(n) and n <= ops.linear_rows_max_rows() and pooler.output_size[0] in (7, 15, 30)
                and pooler.sampling_ratio == 2 and fe.fc6.in_features % 4 == 0 and fe.fc6.out_features % 4 == 0
                and fe.fc7.out_features % 4 == 0)

    def raw_ok(self, n):
        """``refine_raw`` applies: HIP pooler, a threshold no track row can fall under, few enough rows."""
        pp = self.post_processor
        return (isinstance(self.feature_extractor.pooler, Pooler) and pp.score_thresh < 1.0
                and 0 < n <= ops.box_refine_post_max_rows() and self.predictor.cls_score.weight.is_cuda)

    def raw_state_key(self):
        """The mutable state ``raw_ok`` / ``one_call_ok`` rest on (the post-processor's score threshold, where the
        predictor's weights live): the tracking loop caches those verdicts per row count and drops them when this moves
        (``post_processor.score_thresh = 1.0`` or ``.cpu()`` after the first frame must not leave a stale verdict)."""
        st = self.__dict__.get("_raw_state_refs")
        if st is None:                   # (submodule lookups go through nn.Module.__getattr__: once)
            st = self.__dict__["_raw_state_refs"] = (self.post_processor, self.predictor.cls_score)
        w = st[1].weight
        return (float(st[0].score_thresh), w.data_ptr(), w.is_cuda)

    @torch.no_grad()
    def refine_raw(self, features, boxes, conf, ids, labels, image_wh, tracktor=False):
        """The box head on N propagated tracks (every row a track with a label in [1, K)) followed by the score rule of
        ``_refine_tracks`` (roi_heads.py:60-84): HIP pooler -> three library GEMMs -> ONE post-processing launch
        (csrc/box_refine.hip).  Returns ``(boxes [N,4], scores [N] in the (1, 2] band, ids [N], labels [N])`` in the
        box head's output order.  Same numbers as ``forward`` + ``RefineTracks`` (tests/test_box_refine.py)."""
        fe, pp = self.feature_extractor, self.post_processor
        pooler = fe.pooler
        cs, bp = self.predictor.cls_score, self.predictor.bbox_pred
        bc = pp.box_coder
        n = boxes.shape[0]
        if (n <= ops.linear_rows_max_rows() and pooler.output_size[0] in (7, 15, 30) and pooler.sampling_ratio == 2
                and fe.fc6.in_features % 4 == 0 and fe.fc6.out_features % 4 == 0 and fe.fc7.out_features % 4 == 0):
            # the common case (tens of tracks, the yaml's 7x7 / 1024-1024 head): everything behind ONE C-ABI call
            out = ops.box_refine(features, pooler.scales, pooler.output_size[0], pooler.sampling_ratio, boxes, labels, ids,
                                 conf, (fe.fc6.weight, fe.fc6.bias, fe.fc7.weight, fe.fc7.bias, cs.weight, cs.bias,
                                        bp.weight, bp.bias), bc.weights, bc.bbox_xform_clip,
                                 None if pp.amodal_inference else image_wh, tracktor)
            return out
        x = ops.roi_align_levels(features, boxes, boxes, pooler.output_size[0], pooler.scales, pooler.sampling_ratio)
        x = x.view(x.shape[0], -1)
        if x.shape[0] > ops.linear_rows_max_rows():
            ops.FALLBACKS["refine_library_gemm"] += 1
        if x.shape[0] <= ops.linear_rows_max_rows() and x.shape[1] % 4 == 0 and fe.fc6.out_features % 4 == 0 \
                and fe.fc7.out_features % 4 == 0:
            # a handful of rows: weight-streaming kernels on all CUs (csrc/linear_rows.hip); the two predictor layers
            # write the column blocks of one buffer
            h = ops.linear_rows(x, fe.fc6.weight, fe.fc6.bias, relu=True)
            h = ops.linear_rows(h, fe.fc7.weight, fe.fc7.bias, relu=True)
            out = torch.empty((x.shape[0], cs.out_features + bp.out_features), dtype=torch.float32, device=x.device)
            ops.linear_rows(h, cs.weight, cs.bias, out=out)
            ops.linear_rows(h, bp.weight, bp.bias, out=out[:, cs.out_features:])
        else:
            h = _linear_relu(x, fe.fc6)
            h = _linear_relu(h, fe.fc7)
            w, b = self._track_weights()
            out = torch.addmm(b, h, w)
        K = self.predictor.cls_score.out_features
        KR = self.predictor.bbox_pred.out_features // 4
        res = ops.box_refine_post(out, K, KR, boxes, labels, ids, conf, bc.weights, bc.bbox_xform_clip,
                                  None if pp.amodal_inference else image_wh, tracktor)
        return res


class RefineTracks(object):
    """roi_heads.py:60-84 as a callable: ``refine(features, [tracks]) -> [tracks]``.

    The propagated boxes come back from the box head with its regressed box and ``score = (p_label + 1 + track_conf
    + 1) / 2`` in the (1, 2] band (``tracktor``: the box head's score alone).  Like the reference, the matching
    scores are taken in the order the tracks went in and the detection scores in the order the box head returns them
    (grouped by label) — identical orders for the single-class models the reference ships."""

    def __init__(self, box_head, tracktor=False):
        self.box = box_head
        self.tracktor = bool(tracktor)

    def raw_ok(self, n):
        """The device-only form applies (the tracking loop's one-launch path keeps its single synchronisation)."""
        ok = getattr(self.box, "raw_ok", None)
        return ok is not None and ok(n)

    def raw_state_key(self):
        k = getattr(self.box, "raw_state_key", None)
        return k() if k is not None else None

    def refine_raw(self, features, boxes, conf, ids, labels, image_wh):
        return self.box.refine_raw(features, boxes, conf, ids, labels, image_wh, self.tracktor)

    def __call__(self, features, tracks):
        if len(tracks[0]) == 0:
            return tracks
        track_scores = tracks[0].get_field("scores") + 1.0
        _, refined, _ = self.box(features, tracks)
        r = refined[0]
        det_scores = r.get_field("scores")
        scores = det_scores if self.tracktor else (det_scores + track_scores) / 2.0
        out = r.__class__(r.bbox, r.size, mode=r.mode)
        out.add_field("scores", scores)
        out.add_field("ids", r.get_field("ids"))
        out.add_field("labels", r.get_field("labels"))
        return [out]


def build_refine_tracks(cfg, in_channels, box_head=None):
    """The ``refine_tracks`` callable of ``build_tracking_loop`` from a config: the given box head (e.g. the reference's
    ``ROIBoxHead`` with its trained weights) or a fresh ``TrackBoxHead``."""
    if box_head is None:
        box_head = TrackBoxHead(cfg, in_channels).eval()
    return RefineTracks(box_head, bool(getattr(cfg.MODEL.TRACK_HEAD, "TRACKTOR", False)))
