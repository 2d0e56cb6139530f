"""Host-side box container used on both sides of the EMM boundary.

The reference passes ``maskrcnn_benchmark.structures.bounding_box.BoxList`` objects
through ``EMM.forward`` / ``EMM.extract_cache`` (reference
siammot/modelling/track_head/EMM/track_core.py:5,165-181).  maskrcnn-benchmark is an
un-vendored dependency, so this module restates the part of that interface the hot
path touches: ``bbox``/``size``/``mode``, named fields,
``__len__``/``__getitem__``, ``clip_to_image`` with the upstream ``TO_REMOVE = 1``
convention, ``convert`` and ``resize``.

The EMM module itself is duck-typed: a real upstream ``BoxList`` works as well.
"""
import torch

TO_REMOVE = 1


class BoxList(object):
    """Boxes of one image: ``bbox`` is ``[N, 4]`` float, ``size`` is ``(width, height)``."""

    def __init__(self, bbox, image_size, mode="xyxy"):
        device = bbox.device if isinstance(bbox, torch.Tensor) else torch.device("cpu")
        bbox = torch.as_tensor(bbox, dtype=torch.float32, device=device)
        if bbox.ndimension() != 2:
            raise ValueError("bbox should have 2 dimensions, got {}".format(bbox.ndimension()))
        if bbox.size(-1) != 4:
            raise ValueError("last dimension of bbox should have a size of 4, got {}".format(bbox.size(-1)))
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox = bbox
        self.size = image_size
        self.mode = mode
        self.extra_fields = {}

    @classmethod
    def _wrap(cls, bbox, image_size, mode, fields):
        """Internal constructor for tensors this package produced itself (fp32 ``[N,4]`` already): no conversion, no
        checks; ``fields`` becomes the field dict (insertion order = field order)."""
        o = cls.__new__(cls)
        o.bbox, o.size, o.mode, o.extra_fields = bbox, image_size, mode, fields
        return o

    # ---- fields -------------------------------------------------------------------
    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def _copy_extra_fields(self, other):
        for k, v in other.extra_fields.items():
            self.extra_fields[k] = v

    # ---- geometry -----------------------------------------------------------------
    def _split_into_xyxy(self):
        if self.mode == "xyxy":
            return self.bbox.split(1, dim=-1)
        x, y, w, h = self.bbox.split(1, dim=-1)
        return x, y, x + (w - TO_REMOVE).clamp(min=0), y + (h - TO_REMOVE).clamp(min=0)

    def convert(self, mode):
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        if mode == self.mode:
            return self
        x1, y1, x2, y2 = self._split_into_xyxy()
        if mode == "xyxy":
            out = BoxList(torch.cat((x1, y1, x2, y2), dim=-1), self.size, mode=mode)
        else:
            out = BoxList(torch.cat((x1, y1, x2 - x1 + TO_REMOVE, y2 - y1 + TO_REMOVE), dim=-1),
                          self.size, mode=mode)
        out._copy_extra_fields(self)
        return out

    def resize(self, size, *args, **kwargs):
        """Rescale boxes to a new ``(width, height)`` image size."""
        rw = float(size[0]) / float(self.size[0])
        rh = float(size[1]) / float(self.size[1])
        if rw == rh:
            # [UPSTREAM] BoxList.resize: equal ratios scale the stored coordinates directly, in whatever mode
            # they are (xywh boxes are NOT taken through xyxy and its +-1: different fp32 results otherwise)
            out = BoxList(self.bbox * rw, size, mode=self.mode)
            for k, v in self.extra_fields.items():
                if not isinstance(v, torch.Tensor) and hasattr(v, "resize"):
                    v = v.resize(size, *args, **kwargs)
                out.add_field(k, v)
            return out
        x1, y1, x2, y2 = self._split_into_xyxy()
        out = BoxList(torch.cat((x1 * rw, y1 * rh, x2 * rw, y2 * rh), dim=-1), size, mode="xyxy")
        for k, v in self.extra_fields.items():
            if not isinstance(v, torch.Tensor) and hasattr(v, "resize"):
                v = v.resize(size, *args, **kwargs)
            out.add_field(k, v)
        return out.convert(self.mode)

    def clip_to_image(self, remove_empty=True):
        w, h = self.size[0], self.size[1]
        self.bbox[:, 0].clamp_(min=0, max=w - TO_REMOVE)
        self.bbox[:, 1].clamp_(min=0, max=h - TO_REMOVE)
        self.bbox[:, 2].clamp_(min=0, max=w - TO_REMOVE)
        self.bbox[:, 3].clamp_(min=0, max=h - TO_REMOVE)
        if remove_empty:
            box = self.bbox
            keep = (box[:, 3] > box[:, 1]) & (box[:, 2] > box[:, 0])
            return self[keep]
        return self

    def area(self):
        box = self.bbox
        if self.mode == "xyxy":
            return (box[:, 2] - box[:, 0] + TO_REMOVE) * (box[:, 3] - box[:, 1] + TO_REMOVE)
        return box[:, 2] * box[:, 3]

    # ---- container protocol -------------------------------------------------------
    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            if hasattr(v, "to"):
                v = v.to(device)
            out.add_field(k, v)
        return out

    def __getitem__(self, item):
        out = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item])
        return out

    def __len__(self):
        return self.bbox.shape[0]

    def copy_with_fields(self, fields, skip_missing=False):
        out = BoxList(self.bbox, self.size, self.mode)
        if not isinstance(fields, (list, tuple)):
            fields = [fields]
        for field in fields:
            if self.has_field(field):
                out.add_field(field, self.get_field(field))
            elif not skip_missing:
                raise KeyError("Field '{}' not found in {}".format(field, self))
        return out

    def __repr__(self):
        return "BoxList(num_boxes={}, image_width={}, image_height={}, mode={})".format(
            len(self), self.size[0], self.size[1], self.mode)


def cat(tensors, dim=0):
    """``torch.cat`` that skips the copy for a single tensor (upstream modeling/utils.py)."""
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def cat_boxlist(bboxes):
    """Concatenate BoxLists of one image (same size, mode and field names)."""
    assert isinstance(bboxes, (list, tuple))
    size, mode = bboxes[0].size, bboxes[0].mode
    fields = set(bboxes[0].fields())
    for b in bboxes:
        assert tuple(b.size) == tuple(size) and b.mode == mode and set(b.fields()) == fields
    out = bboxes[0].__class__(cat([b.bbox for b in bboxes], dim=0), size, mode)
    for f in fields:
        out.add_field(f, cat([b.get_field(f) for b in bboxes], dim=0))
    return out


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):
    """[UPSTREAM] ``structures.boxlist_ops.boxlist_nms`` on the HIP NMS kernel (device BoxLists only)."""
    if nms_thresh <= 0:
        return boxlist
    from . import ops
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = ops.nms(boxlist.bbox, boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep].convert(mode)
