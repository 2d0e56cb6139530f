"""Host mirror of the reference's ``TrackUtils`` geometry helpers.

Reference: siammot/modelling/track_head/track_utils.py:8-135, built by ``build_track_utils``
(:258-269).  The EMM module only reads three attributes from whatever ``track_utils`` object it is
given — ``pad_pixels``, ``search_expansion``, ``min_search_wh`` — so the reference's own instance
is a drop-in; this class exists for stand-alone use (no reference on the path) and for tests.
``pad_feature`` is kept for interface parity; the HIP path never calls it (padding is virtual).
"""
import torch
import torch.nn.functional as F


class TrackUtils(object):
    def __init__(self, search_expansion=1.0, min_search_wh=128, pad_pixels=256):
        self.search_expansion = search_expansion
        self.min_search_wh = min_search_wh
        self.pad_pixels = pad_pixels

    def pad_level_cells(self, level):
        """Zero border of FPN level ``level`` in cells (track_utils.py:97-99)."""
        return int(self.pad_pixels / ((2 ** level) * 4))

    def pad_feature(self, f):
        if isinstance(f, (list, tuple)):
            return tuple(F.pad(_f, [self.pad_level_cells(i)] * 4, mode="constant", value=0)
                         for i, _f in enumerate(f))
        return F.pad(f, [self.pad_pixels] * 4, mode="constant", value=0)

    def update_boxes_in_pad_images(self, boxlists):
        out = []
        for bl in boxlists:
            assert bl.mode == "xyxy"
            w, h = bl.size
            nb = bl.__class__(bl.bbox + self.pad_pixels,
                              [int(w + self.pad_pixels * 2), int(h + self.pad_pixels * 2)], mode="xyxy")
            for field in bl.fields():
                nb.add_field(field, bl.get_field(field))
            out.append(nb)
        return out

    def extend_bbox(self, in_box):
        e = self.search_expansion
        for bl in in_box:
            w = bl.bbox[:, 2] - bl.bbox[:, 0] + 1
            h = bl.bbox[:, 3] - bl.bbox[:, 1] + 1
            w_ext = torch.max((self.min_search_wh - w) / (e * 2.), w * (e / 2.))
            h_ext = torch.max((self.min_search_wh - h) / (e * 2.), h * (e / 2.))
            bl.bbox[:, 0] -= w_ext
            bl.bbox[:, 1] -= h_ext
            bl.bbox[:, 2] += w_ext
            bl.bbox[:, 3] += h_ext
        return in_box


def build_track_utils(cfg):
    """TrackUtils from cfg (reference build_track_utils, track_utils.py:258-269).  The reference's builder also
    returns a TrackPool; here the pool is built where the solver is (``siammot_amd.solver.TrackPool``, a
    re-implementation with a device-resident state — the reference's own class works with this head too)."""
    th = cfg.MODEL.TRACK_HEAD
    return TrackUtils(search_expansion=th.SEARCH_REGION - 1.,
                      min_search_wh=th.MINIMUM_SREACH_REGION,
                      pad_pixels=th.PAD_PIXELS)
