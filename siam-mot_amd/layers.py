"""[UPSTREAM] ``maskrcnn_benchmark.layers.ROIAlign`` on the HIP kernel — the layer the reference instantiates in
``SRPooler`` (EMM/sr_pool.py:28-31) and upstream's ``Pooler`` instantiates for the box head.

Same constructor and call signature as upstream's module: ``ROIAlign(output_size, spatial_scale, sampling_ratio)``,
``forward(input [B,C,H,W], rois [R,5] = (image index, x1, y1, x2, y2)) -> [R,C,h,w]``.  ``patch_upstream()`` swaps the
``forward`` of upstream's class for this one so that an unmodified reference runs its poolers on
``smot_roi_align_fwd`` without the ``maskrcnn_benchmark._C`` extension.
"""
from torch import nn

from . import ops


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super(ROIAlign, self).__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        return ops.roi_align(input, rois, self.spatial_scale, self.output_size[0], self.output_size[1],
                             self.sampling_ratio)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)


def patch_upstream():
    """``maskrcnn_benchmark.layers.ROIAlign.forward = siammot_amd.layers.ROIAlign.forward`` (inference only)."""
    import maskrcnn_benchmark.layers as L
    L.ROIAlign.forward = ROIAlign.forward
    return L.ROIAlign
