"""``Pooler`` — the box head's FPN-level-routed ROIAlign on the same HIP kernel as the EMM poolers.

SURVEY.md §8(f) rank 1.  Mirrors [UPSTREAM] ``maskrcnn_benchmark.modeling.poolers.Pooler`` as the reference
uses it: ``ROIBoxHead`` (siammot/modelling/box_head/box_head.py:17,46) pools ≤300 RPN proposals at 7×7 and
``CombinedROIHeads._refine_tracks`` (siammot/modelling/roi_heads.py:60-84) pools the N track boxes again.
Same contract: ``Pooler(output_size, scales, sampling_ratio)(x, boxes) -> Tensor[R, C, h, w]``; each roi is
assigned a level by ``LevelMapper`` of ITS OWN box (k_min/k_max from the first/last scale), legacy
(non-aligned) ROIAlign with ``sampling_ratio``² samples per bin.  One image per call, as everywhere on
this path (reference EMM/track_core.py:75); no padding (the box head pools the raw maps).
"""
from torch import nn

from . import ops
from .structures import cat


class Pooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio):
        super(Pooler, self).__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        if output_size[0] != output_size[1]:
            raise ValueError("Pooler: square output expected, got %s" % (output_size,))
        self.output_size = tuple(output_size)
        self.scales = tuple(float(s) for s in scales)
        self.sampling_ratio = sampling_ratio

    def forward(self, x, boxes):
        if len(boxes) != 1:
            raise RuntimeError("Pooler: one image per call")
        rois = cat([b.bbox for b in boxes], dim=0)
        return ops.roi_align_levels(x, rois, rois, self.output_size[0], self.scales, self.sampling_ratio)
