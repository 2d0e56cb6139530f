"""``EMM`` — the Siamese tracker head behind the reference's plug-in interface, on HIP kernels.

Mirrors reference siammot/modelling/track_head/EMM/track_core.py:14-98 (class ``EMM``, registered
in ``SIAMESE_TRACKER``), EMM/feature_extractor.py:9-69 (``EMMFeatureExtractor``, ``EMMPredictor``)
and EMM/sr_pool.py:9-91 (``SRPooler``): same constructor arguments, same ``forward`` /
``extract_cache`` signatures and return structure, same ``state_dict`` keys
(``predictor.{cls_tower.0,cls_tower.1,reg_tower.0,reg_tower.1,cls,center,reg}.*``), so
``TrackHead`` (track_head/track_head.py:8-126) and everything above it run unchanged.

Inference only: the training branch of ``EMM.forward`` (track_core.py:45-47,56-67) is out of scope
and raises.  All arithmetic happens in ``csrc/libsmot_emm.so``; nothing here falls back to eager.

What differs from the reference on purpose (results are the same):
  * ``TrackUtils.pad_feature`` is never materialised — the pooler pads virtually;
  * no ``torch.nonzero`` host syncs for level routing — the kernel maps levels itself;
  * the up-sampled 7x256x256 planes and the location tensor are never written.
"""
import math

import torch
from torch import nn

from . import ops
from .registry import SIAMESE_TRACKER
from .structures import BoxList as _OwnBoxList, cat


class SRPooler(nn.Module):
    """FPN-level-routed pooler (reference EMM/sr_pool.py:9-91).

    ``forward(x, boxes, sr=None)`` keeps the reference contract.  The extra ``pad_pixels`` keyword
    tells the kernel that ``sr`` coordinates live in an image zero-padded by that many pixels while
    ``x`` holds the UN-padded maps (what ``EMM.forward`` passes); with the default 0 it behaves
    exactly like the reference pooler on whatever maps it is given.
    """

    def __init__(self, output_size, scales, sampling_ratio):
        super(SRPooler, self).__init__()
        self.output_size = output_size
        self.scales = tuple(float(s) for s in scales)
        self.sampling_ratio = sampling_ratio

    def forward(self, x, boxes, sr=None, pad_pixels=0):
        if self.output_size[0] != self.output_size[1]:
            raise RuntimeError("SRPooler: square output expected, got %s" % (self.output_size,))
        level_boxes = cat([b.bbox for b in boxes], dim=0)
        rois = level_boxes if sr is None else cat([b.bbox for b in sr], dim=0)
        if len(boxes) != 1:
            raise RuntimeError("SRPooler: one image per call (reference track_core.py:75)")
        pad_cells = [int(pad_pixels / ((2 ** i) * 4)) for i in range(len(self.scales))]
        return ops.roi_align_levels(x, rois, level_boxes, self.output_size[0], self.scales,
                                    self.sampling_ratio, pad_cells)


class EMMFeatureExtractor(nn.Module):
    """Template (Rz) and search-region (Rx = int(Rz * SEARCH_REGION)) poolers
    (reference EMM/feature_extractor.py:9-40)."""

    def __init__(self, cfg):
        super(EMMFeatureExtractor, self).__init__()
        th = cfg.MODEL.TRACK_HEAD
        resolution = th.POOLER_RESOLUTION
        r = th.SEARCH_REGION
        self.pooler_z = SRPooler((resolution, resolution), th.POOLER_SCALES, th.POOLER_SAMPLING_RATIO)
        self.pooler_x = SRPooler((int(resolution * r), int(resolution * r)), th.POOLER_SCALES,
                                 th.POOLER_SAMPLING_RATIO)

    def forward(self, x, proposals, sr=None, pad_pixels=0):
        if sr is not None:
            return self.pooler_x(x, proposals, sr, pad_pixels=pad_pixels)
        return self.pooler_z(x, proposals)


def _conv3x3(in_ch, out_ch, use_gn=False, use_relu=False, gn_groups=32, gn_eps=1e-5):
    """[UPSTREAM] make_conv3x3(kaiming_init=False): normal(std=0.01) weights, zero bias."""
    conv = nn.Conv2d(in_ch, out_ch, kernel_size=3, stride=1, padding=1, bias=not use_gn)
    nn.init.normal_(conv.weight, std=0.01)
    if not use_gn:
        nn.init.constant_(conv.bias, 0)
    mods = [conv]
    if use_gn:
        mods.append(nn.GroupNorm(gn_groups, out_ch, gn_eps, affine=True))
    if use_relu:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods) if len(mods) > 1 else conv


class EMMPredictor(nn.Module):
    """Parameter container with the reference's module tree (EMM/feature_extractor.py:43-69); the
    sub-modules are never called — ``forward`` hands their tensors to the HIP predictor."""

    def __init__(self, cfg):
        super(EMMPredictor, self).__init__()
        body = cfg.MODEL.BACKBONE.CONV_BODY
        if body.startswith("DLA"):
            in_channels = cfg.MODEL.DLA.BACKBONE_OUT_CHANNELS
        elif body.startswith("R-"):
            in_channels = cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS
        else:
            in_channels = 128
        gn = getattr(cfg.MODEL, "GROUP_NORM", None)
        self.gn_groups = gn.NUM_GROUPS if gn is not None else 32
        self.gn_eps = gn.EPSILON if gn is not None else 1e-5
        self.cls_tower = _conv3x3(in_channels, in_channels, True, True, self.gn_groups, self.gn_eps)
        self.reg_tower = _conv3x3(in_channels, in_channels, True, True, self.gn_groups, self.gn_eps)
        self.cls = _conv3x3(in_channels, 2)
        self.center = _conv3x3(in_channels, 1)
        self.reg = _conv3x3(in_channels, 4)

    def param_dict(self):
        """Reference-keyed Parameter objects, collected once (``.to()`` / ``load_state_dict`` update
        the same Parameter objects in place, so the pointers read at call time are always current)."""
        cache = self.__dict__.get("_param_cache")
        if cache is None:
            cache = dict(self.named_parameters())
            self.__dict__["_param_cache"] = cache
        return cache

    def forward_logits(self, x):
        """→ ``[N,7,Ho,Ho]`` (cls0, cls1, center, reg l/t/r/b)."""
        return ops.emm_predictor(x, self.param_dict(), self.gn_groups, self.gn_eps)

    def forward(self, x):
        logits = self.forward_logits(x)
        return logits[:, 0:2], logits[:, 2:3], logits[:, 3:7]


class OrderHint(object):
    """The scheduling hint ``extract_cache`` leaves on its search-region BoxList for the next ``forward`` (include/smot_emm.h,
    ``order_hint``): the cost-sorted list of exactly these rois, written by the extraction launch.  Valid only for the very
    tensors it was made from — ``lookup`` checks identity and in-place versions; any copy, merge (dormant tracks joining
    the memory) or edit of the boxes silently drops it and the head ranks the rois itself, with the same results."""
    __slots__ = ("data", "boxes", "boxes_version", "sr", "sr_version", "scales")

    def __init__(self, data, boxes, sr, scales):
        self.data, self.boxes, self.sr, self.scales = data, boxes, sr, scales
        self.boxes_version, self.sr_version = boxes._version, sr._version

    @staticmethod
    def lookup(sr_boxlist, boxes, sr, scales):
        h = sr_boxlist.__dict__.get("order_hint")
        if (h is not None and h.sr is sr and h.boxes is boxes and h.scales == scales
                and sr._version == h.sr_version and boxes._version == h.boxes_version
                and h.data.shape[0] == boxes.shape[0]):
            return h.data
        return None


class EMM(nn.Module):
    """Drop-in for the reference ``EMM`` (EMM/track_core.py:14-98)."""

    use_order_hint = True      # extract_cache leaves an ``OrderHint`` for the next forward (False: never asks for one)

    def __init__(self, cfg, track_utils):
        super(EMM, self).__init__()
        self.feature_extractor = EMMFeatureExtractor(cfg)
        self.predictor = EMMPredictor(cfg)
        self.track_utils = track_utils
        self.amodal = cfg.INPUT.AMODAL
        self.use_centerness = cfg.MODEL.TRACK_HEAD.EMM.USE_CENTERNESS
        self.pad_pixels = cfg.MODEL.TRACK_HEAD.PAD_PIXELS
        self.sigma = cfg.MODEL.TRACK_HEAD.EMM.COSINE_WINDOW_WEIGHT
        self.rz = cfg.MODEL.TRACK_HEAD.POOLER_RESOLUTION
        self.rx = int(self.rz * cfg.MODEL.TRACK_HEAD.SEARCH_REGION)

    def forward(self, features, boxes, sr, targets=None, template_features=None):
        if self.training:
            raise NotImplementedError("siammot_amd.EMM is an inference path; training "
                                      "(track_core.py:45-47,56-67) is out of scope")
        assert len(boxes) == 1                                           # track_core.py:75
        st = self.__dict__.get("_static")
        if st is None:                   # submodule / attribute lookups go through nn.Module.__getattr__: once
            fe, pr = self.feature_extractor.pooler_x, self.predictor
            st = self.__dict__["_static"] = (pr.param_dict(), tuple(fe.scales), fe.sampling_ratio, pr.gn_groups,
                                             pr.gn_eps)
        params, scales, sampling_ratio, gn_groups, gn_eps = st
        # one library call: pooling -> xcorr -> predictor -> decode (+ the clamp of clip_to_image)
        one = len(sr) == 1
        b0 = boxes[0]
        boxes_bbox = b0.bbox
        sr_bbox = sr[0].bbox if one else cat([b.bbox for b in sr], dim=0)
        hint = OrderHint.lookup(sr[0], boxes_bbox, sr_bbox, scales) if one else None
        out = None
        plan = self._pair_plan(features, boxes_bbox, params, scales, sampling_ratio)
        if plan is not None:
            size = b0.size
            out = plan.track(features, boxes_bbox, sr_bbox, template_features, self.sigma, self.use_centerness,
                             0.0 if self.amodal else float(size[0]), 0.0 if self.amodal else float(size[1]), gn_groups,
                             gn_eps, hint)
        if out is None:                  # anything the plan does not recognise: the general binding (raises or converts)
            out = ops.emm_track(features, boxes_bbox, sr_bbox,
                                template_features, params, self.rx, self.rz, scales, sampling_ratio,
                                self.pad_pixels, sigma=self.sigma, use_centerness=self.use_centerness,
                                clip_wh=None if self.amodal else b0.size,
                                gn_groups=gn_groups, gn_eps=gn_eps, order_hint=hint)
        bb, bb_conf = out
        if b0.__class__ is _OwnBoxList and len(b0) == bb.shape[0]:
            # (this package's container: no re-validation of a tensor the library just wrote; boxes already clamped)
            f = b0.extra_fields
            track_result = [_OwnBoxList._wrap(bb, b0.size, "xyxy", {"ids": f["ids"], "labels": f["labels"], "scores": bb_conf})]
        else:
            track_result = wrap_results_to_boxlist(bb, bb_conf, boxes, amodal=True)   # already clamped
        return {}, track_result, {}

    def _pair_plan(self, features, boxes_bbox, params, scales, sampling_ratio):
        """The cached host plan of this module's two per-frame calls (``ops.PairPlan``), made on first use and again when a
        configuration value, the parameter dict, the library or the device changed; None when the inputs are not plain
        device tensors (the general binding then raises or converts)."""
        plan = self.__dict__.get("_plan")
        tu = self.track_utils
        if plan is not None and plan.dev == boxes_bbox.device and not plan.stale(params, self.rx, self.rz, scales,
                                                                                 sampling_ratio, self.pad_pixels, tu):
            return plan
        if not (isinstance(boxes_bbox, torch.Tensor) and boxes_bbox.is_cuda and self.rx - self.rz + 1 in (16, 29)):
            return None
        try:
            plan = ops.PairPlan(features, boxes_bbox.device, params, self.rx, self.rz, scales, sampling_ratio,
                                self.pad_pixels, tu)
        except (RuntimeError, IndexError, TypeError, AttributeError):
            return None                  # the general binding reports what is wrong with the inputs
        self.__dict__["_plan"] = plan
        return plan

    def track_raw(self, features, boxes, sr, template_features, image_wh, sr_boxlist=None):
        """The inference branch of ``forward`` on raw tensors: template boxes ``[N,4]``, search regions ``[N,4]``,
        templates ``[N,C,rz,rz]`` -> (boxes ``[N,4]``, scores ``[N]``), clamped to the image unless amodal.  No
        BoxList in or out — the tracking loop's per-frame fast path (track_head.TrackingLoop).  ``sr_boxlist``: the
        BoxList ``sr`` came from (it may carry the extraction's order hint)."""
        st = self.__dict__.get("_static")
        if st is None:
            fe, pr = self.feature_extractor.pooler_x, self.predictor
            st = self.__dict__["_static"] = (pr.param_dict(), tuple(fe.scales), fe.sampling_ratio, pr.gn_groups,
                                             pr.gn_eps)
        params, scales, sampling_ratio, gn_groups, gn_eps = st
        hint = OrderHint.lookup(sr_boxlist, boxes, sr, scales) if sr_boxlist is not None else None
        out = ops.emm_track(features, boxes, sr, template_features, params, self.rx, self.rz, scales, sampling_ratio,
                            self.pad_pixels, sigma=self.sigma, use_centerness=self.use_centerness,
                            clip_wh=None if self.amodal else image_wh, gn_groups=gn_groups, gn_eps=gn_eps,
                            order_hint=hint)
        return out

    def _template_pooler(self):
        sz = self.__dict__.get("_static_z")
        if sz is None:                   # (submodule lookups go through nn.Module.__getattr__: once)
            fz = self.feature_extractor.pooler_z
            sz = self.__dict__["_static_z"] = (tuple(fz.scales), fz.sampling_ratio)
        return sz

    def extract_cache(self, features, detection):
        """(template features, [search regions], [detections]) — track_core.py:81-98."""
        det = detection
        tu = self.track_utils
        sz = self._template_pooler()
        r = None
        st = self.__dict__.get("_static")
        if st is not None and sz[0] == st[1] and sz[1] == st[2] and isinstance(det.bbox, torch.Tensor):
            plan = self._pair_plan(features, det.bbox, st[0], st[1], st[2])
            if plan is not None:
                r = plan.extract(features, det.bbox, self.use_order_hint)
        if r is None:
            r = ops.emm_extract_cache(features, det.bbox, self.rz, sz[0], sz[1], tu.pad_pixels, tu.search_expansion,
                                      tu.min_search_wh, hint=self.use_order_hint)
        return self.wrap_cache(r[0], r[1], det, r[2] if len(r) > 2 else None)

    def wrap_cache(self, x, sr_bbox, det, hint=None):
        """(templates, search-region boxes) of ``det``'s rows -> the reference's cache tuple.  ``hint``: the order
        hint the extraction wrote for exactly these rows (``[len(det), HINT_FLOATS]``) or None."""
        tu = self.track_utils
        w, h = det.size
        if det.__class__ is _OwnBoxList:       # (this package's container: no re-validation of the library's own output)
            sr = _OwnBoxList._wrap(sr_bbox, [int(w + tu.pad_pixels * 2), int(h + tu.pad_pixels * 2)], "xyxy",
                                   dict(det.extra_fields))
        else:
            sr = det.__class__(sr_bbox, [int(w + tu.pad_pixels * 2), int(h + tu.pad_pixels * 2)], mode="xyxy")
            for field in det.fields():
                sr.add_field(field, det.get_field(field))
        if hint is not None:
            sr.order_hint = OrderHint(hint, det.bbox, sr_bbox, self._template_pooler()[0])
        return x, [sr], [det]

    def extract_cache_rows(self, features, boxes, n_valid, hint=False):
        """``extract_cache`` on a CAPACITY of boxes ``[M,4]`` whose number of real rows is still on the device
        (``n_valid``: int32 tensor, 1 element — the solver kernel's count): launch only, rows beyond the count are
        skipped by the kernel.  Returns capacity-sized ``(templates [M,C,rz,rz], sr [M,4])``; the caller slices them
        once the count is on the host (``wrap_cache``).  Lets the tracking loop enqueue the template extraction BEFORE
        its one synchronisation of the frame.  ``hint=True`` appends the order hint ``[M,8]`` (or None) of the valid
        rows; the tracking loop does not ask for it: carrying the hint through a frame costs 3-5 us of host work on the
        loop's serial chain and saves the head 0.4-1.5 us at 30-100 tracks (measure/loop_hint_ab.py)."""
        tu = self.track_utils
        sz = self._template_pooler()
        if not (self.rz in (15, 7) and sz[1] == 2):
            return None                                    # no masked kernel for this shape family: caller falls back
        return ops.emm_extract_cache(features, boxes, self.rz, sz[0], sz[1], tu.pad_pixels, tu.search_expansion,
                                     tu.min_search_wh, n_valid=n_valid, hint=hint and self.use_order_hint)


def wrap_results_to_boxlist(bb, bb_conf, boxes, amodal=False):
    """reference track_core.py:165-181 (one image per call)."""
    out = []
    n0 = 0
    whole = len(boxes) == 1 and len(boxes[0]) == bb.shape[0]      # one image: no slicing views needed
    for _boxes in boxes:
        n1 = n0 + len(_boxes)
        tb = _boxes.__class__(bb if whole else bb[n0:n1].reshape(-1, 4), _boxes.size, mode="xyxy")
        tb.add_field("ids", _boxes.get_field("ids"))
        tb.add_field("labels", _boxes.get_field("labels"))
        tb.add_field("scores", bb_conf if whole else bb_conf[n0:n1])
        if not amodal:
            # as the reference: the returned (filtered) copy is discarded — boxes are clamped in
            # place, empty ones are NOT removed (track_core.py:177-178)
            tb.clip_to_image(remove_empty=True)
        out.append(tb)
        n0 = n1
    return out


def register(name="EMM", override=False):
    """Put this class into ``SIAMESE_TRACKER`` (the dict ``build_track_head`` reads,
    track_head.py:118-120).  With the reference importable, ``override=True`` replaces its "EMM"."""
    if name in SIAMESE_TRACKER and not override and SIAMESE_TRACKER[name] is not EMM:
        raise KeyError("SIAMESE_TRACKER[%r] is already taken; pass override=True" % name)
    SIAMESE_TRACKER[name] = EMM
    return EMM


SIAMESE_TRACKER["EMM_HIP"] = EMM
if "EMM" not in SIAMESE_TRACKER:
    SIAMESE_TRACKER["EMM"] = EMM
