"""TEST INFRASTRUCTURE — CPU restatement of the [UPSTREAM] maskrcnn_benchmark pieces the reference's box head is
assembled from (siammot/modelling/box_head/box_head.py:18-21): ``FPN2MLPFeatureExtractor``, ``FPNPredictor`` and
``BoxCoder.decode``.  Upstream's source is not vendored in /root/reference (readme/INSTALL.md:89-92), so the published
algorithm is restated; ``oracle/gen_golden_refine.py`` plugs these into the reference's OWN ``ROIBoxHead`` /
``PostProcessor`` / ``CombinedROIHeads._refine_tracks`` to produce ``tests/golden/refine_tracks.npz``.
Only tests/ and the golden generators import this module.
"""
import math

import torch
from torch import nn

from . import emm_oracle as O


class BoxCoder(object):
    """[UPSTREAM] modeling/box_coder.py (decode only), written loop-wise per class on purpose — a different
    formulation from siammot_amd.box_refine.BoxCoder's strided one."""

    def __init__(self, weights, bbox_xform_clip=math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def decode(self, rel_codes, boxes):
        boxes = boxes.to(rel_codes.dtype)
        w = boxes[:, 2] - boxes[:, 0] + 1
        h = boxes[:, 3] - boxes[:, 1] + 1
        cx = boxes[:, 0] + 0.5 * w
        cy = boxes[:, 1] + 0.5 * h
        wx, wy, ww, wh = self.weights
        out = torch.zeros_like(rel_codes)
        for k in range(rel_codes.shape[1] // 4):
            dx, dy = rel_codes[:, 4 * k] / wx, rel_codes[:, 4 * k + 1] / wy
            dw = torch.clamp(rel_codes[:, 4 * k + 2] / ww, max=self.bbox_xform_clip)
            dh = torch.clamp(rel_codes[:, 4 * k + 3] / wh, max=self.bbox_xform_clip)
            pcx, pcy = dx * w + cx, dy * h + cy
            pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
            out[:, 4 * k] = pcx - 0.5 * pw
            out[:, 4 * k + 1] = pcy - 0.5 * ph
            out[:, 4 * k + 2] = pcx + 0.5 * pw - 1
            out[:, 4 * k + 3] = pcy + 0.5 * ph - 1
        return out


class OraclePooler(nn.Module):
    """[UPSTREAM] modeling/poolers.py ``Pooler`` over the oracle ROIAlign (one image per call)."""

    def __init__(self, resolution, scales, sampling_ratio):
        super(OraclePooler, self).__init__()
        self.resolution, self.scales, self.sampling_ratio = resolution, tuple(scales), sampling_ratio

    def forward(self, x, boxes):
        rois = torch.cat([b.bbox for b in boxes], 0)
        return O.sr_pool(list(x), rois, None, self.resolution, self.scales, self.sampling_ratio)


class FPN2MLPFeatureExtractor(nn.Module):
    def __init__(self, in_channels, resolution, scales, sampling_ratio, dim):
        super(FPN2MLPFeatureExtractor, self).__init__()
        self.pooler = OraclePooler(resolution, scales, sampling_ratio)
        self.fc6 = nn.Linear(in_channels * resolution ** 2, dim)
        self.fc7 = nn.Linear(dim, dim)
        self.out_channels = dim

    def forward(self, x, proposals):
        x = self.pooler(x, proposals)
        x = x.view(x.size(0), -1)
        x = nn.functional.relu(self.fc6(x))
        return nn.functional.relu(self.fc7(x))


class FPNPredictor(nn.Module):
    def __init__(self, dim, num_classes):
        super(FPNPredictor, self).__init__()
        self.cls_score = nn.Linear(dim, num_classes)
        self.bbox_pred = nn.Linear(dim, num_classes * 4)

    def forward(self, x):
        return self.cls_score(x), self.bbox_pred(x)
