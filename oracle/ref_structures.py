"""TEST INFRASTRUCTURE — the oracle's own restatement of [UPSTREAM] maskrcnn-benchmark box containers.

The golden generators (``oracle/gen_golden*.py``) import the reference's modules unmodified from /root/reference and must
satisfy their ``maskrcnn_benchmark.structures.*`` / ``modeling.utils.cat`` imports.  Until round 4 they borrowed the
product's ``siammot_amd.structures`` for that, which put code under test on both sides of every fixture.  This module is
the generators' own, independent restatement (upstream's published algorithm: structures/bounding_box.py,
structures/boxlist_ops.py, modeling/utils.py — facebookresearch/maskrcnn-benchmark, un-pinned ``master`` as the reference's
INSTALL.md:89-92 prescribes).  Nothing under ``siam-mot_amd/`` imports it, and it imports nothing from there
(``tests/test_host.py::test_oracle_never_imports_the_product``).

Only what the reference's hot path and its callers touch is restated:
  * track_core.py:165-181 ``BoxList(bbox, size, mode)``, ``add_field``, ``get_field``, ``clip_to_image``
  * sr_pool.py:40-51,74 ``bbox``, ``area()`` (LevelMapper), ``cat``
  * track_utils.py:109-135, 157-236 ``size``, ``mode``, indexing, ``__len__``, ``copy_with_fields``
  * track_head.py:77-110, track_solver.py:36-108, roi_heads.py:22-84 ``cat_boxlist``, ``fields``, ``convert``
  * box_head/inference.py:46-185 ``resize``-free post-processing (``clip_to_image(remove_empty=False)``)
"""
import torch

_ONE = 1                     # upstream's TO_REMOVE: boxes are inclusive pixel ranges, width = x2 - x1 + 1
_MODES = ("xyxy", "xywh")


class BoxList:
    """[UPSTREAM] structures/bounding_box.py — boxes of ONE image with named per-box fields; ``size`` = (width, height)."""

    def __init__(self, bbox, image_size, mode="xyxy"):
        dev = bbox.device if torch.is_tensor(bbox) else torch.device("cpu")
        t = torch.as_tensor(bbox, dtype=torch.float32, device=dev)
        if t.dim() != 2:
            raise ValueError("bbox should have 2 dimensions, got {}".format(t.dim()))
        if t.shape[-1] != 4:
            raise ValueError("last dimension of bbox should have a size of 4, got {}".format(t.shape[-1]))
        if mode not in _MODES:
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox, self.size, self.mode = t, image_size, mode
        self.extra_fields = {}

    # -- named fields (a plain dict in insertion order, as upstream) --------------------------------------------
    def add_field(self, name, data):
        self.extra_fields[name] = data

    def get_field(self, name):
        return self.extra_fields[name]

    def has_field(self, name):
        return name in self.extra_fields

    def fields(self):
        return list(self.extra_fields)

    def _copy_extra_fields(self, src):
        self.extra_fields.update(src.extra_fields)

    # -- coordinate modes -----------------------------------------------------------------------------------------
    def _corners(self):
        """(x1, y1, x2, y2) as four [N,1] columns, whatever the stored mode."""
        a, b, c, d = self.bbox.split(1, dim=-1)
        if self.mode == "xyxy":
            return a, b, c, d
        return a, b, a + (c - _ONE).clamp(min=0), b + (d - _ONE).clamp(min=0)

    def convert(self, mode):
        if mode not in _MODES:
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        if mode == self.mode:
            return self                                   # upstream returns self, not a copy
        x1, y1, x2, y2 = self._corners()
        cols = (x1, y1, x2, y2) if mode == "xyxy" else (x1, y1, x2 - x1 + _ONE, y2 - y1 + _ONE)
        new = BoxList(torch.cat(cols, dim=-1), self.size, mode=mode)
        new._copy_extra_fields(self)
        return new

    def resize(self, size, *args, **kwargs):
        fx, fy = (float(n) / float(o) for n, o in zip(size, self.size))

        def carry(dst):
            for k, v in self.extra_fields.items():
                if not torch.is_tensor(v) and hasattr(v, "resize"):
                    v = v.resize(size, *args, **kwargs)
                dst.add_field(k, v)
            return dst

        if fx == fy:                                      # upstream: one multiply on the stored coordinates
            return carry(BoxList(self.bbox * fx, size, mode=self.mode))
        x1, y1, x2, y2 = self._corners()
        scaled = torch.cat((x1 * fx, y1 * fy, x2 * fx, y2 * fy), dim=-1)
        return carry(BoxList(scaled, size, mode="xyxy")).convert(self.mode)

    def clip_to_image(self, remove_empty=True):
        """In-place clamp to [0, W-1] x [0, H-1]; with ``remove_empty`` a FILTERED COPY is returned (the caller that
        discards it — track_core.py:177-178 — keeps every row, clamped)."""
        wmax, hmax = self.size[0] - _ONE, self.size[1] - _ONE
        for col, hi in ((0, wmax), (1, hmax), (2, wmax), (3, hmax)):
            self.bbox[:, col].clamp_(min=0, max=hi)
        if not remove_empty:
            return self
        b = self.bbox
        return self[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]

    def area(self):
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + _ONE) * (b[:, 3] - b[:, 1] + _ONE)
        return b[:, 2] * b[:, 3]

    # -- container protocol ---------------------------------------------------------------------------------------
    def to(self, device):
        new = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            new.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return new

    def __getitem__(self, item):
        new = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            new.add_field(k, v[item])
        return new

    def __len__(self):
        return self.bbox.shape[0]

    def copy_with_fields(self, fields, skip_missing=False):
        new = BoxList(self.bbox, self.size, self.mode)
        for f in (fields if isinstance(fields, (list, tuple)) else [fields]):
            if self.has_field(f):
                new.add_field(f, self.get_field(f))
            elif not skip_missing:
                raise KeyError("Field '{}' not found in {}".format(f, self))
        return new

    def __repr__(self):
        return "BoxList(num_boxes={}, image_width={}, image_height={}, mode={})".format(
            len(self), self.size[0], self.size[1], self.mode)


def cat(tensors, dim=0):
    """[UPSTREAM] modeling/utils.py ``cat``: ``torch.cat`` without the copy for a single tensor."""
    assert isinstance(tensors, (list, tuple))
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)


def cat_boxlist(bboxes):
    """[UPSTREAM] structures/boxlist_ops.py ``cat_boxlist``: boxes of one image, same size / mode / field names."""
    assert isinstance(bboxes, (list, tuple))
    assert all(isinstance(b, BoxList) for b in bboxes)
    first = bboxes[0]
    assert all(tuple(b.size) == tuple(first.size) for b in bboxes)
    assert all(b.mode == first.mode for b in bboxes)
    names = set(first.fields())
    assert all(set(b.fields()) == names for b in bboxes)
    out = BoxList(cat([b.bbox for b in bboxes], dim=0), first.size, first.mode)
    for f in names:
        out.add_field(f, cat([b.get_field(f) for b in bboxes], dim=0))
    return out
