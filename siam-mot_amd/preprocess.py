"""Frame pre-processing on the GPU.

Mirrors what the reference's inference entry points do per frame on the CPU
(demos/demo_inference.py:74-82 ``_preprocess``; the test-time chain of
siammot/data/adapters/augmentation/build_augmentation.py:52-66 — ``SiamVideoResize`` -> ``ToTensor`` ->
``Normalize``; ``ImageResize.get_size`` image_augmentation.py:21-42): a uint8 RGB frame goes to the device as
bytes and one HIP kernel produces the fp32 CHW network input, bit-exact with ``PIL.Image.resize(BILINEAR)``
followed by the [UPSTREAM] ToTensor / Normalize arithmetic.

The per-axis resampling tables are Pillow's (precompute_coeffs + normalize_coeffs_8bpc, src/libImaging/
Resample.c [THIRD PARTY]) evaluated here in float64 with numpy, vectorised over the output axis, and cached per
(input size, output size).
"""
import math

import numpy as np
import torch

from . import ops

PRECISION_BITS = 32 - 8 - 2
TILE_ROWS = 8                      # output rows per workgroup of the kernel (csrc/preprocess.hip)


def get_size(image_wh, min_size, max_size, size_divisibility):
    """``ImageResize.get_size`` (image_augmentation.py:21-42) for a single ``min_size`` -> (oh, ow)."""
    w, h = image_wh
    size = min_size
    if max_size is not None:
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > max_size:
            size = int(round(max_size * lo / hi))
    if w < h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    if size_divisibility > 0:
        oh = int(oh / size_divisibility) * size_divisibility
        ow = int(ow / size_divisibility) * size_divisibility
    return oh, ow


def resample_tables(in_size, out_size):
    """Pillow's BILINEAR tables for one axis: bounds ``[out,2]`` int32 (first input index, tap count) and
    coeffs ``[out,ksize]`` int32 (22 fractional bits).  An axis that keeps its size gets identity tables, which
    reproduces Pillow skipping that pass."""
    if in_size == out_size:
        bounds = np.stack((np.arange(out_size), np.ones(out_size, dtype=np.int64)), axis=1).astype(np.int32)
        return bounds, np.full((out_size, 1), 1 << PRECISION_BITS, dtype=np.int32)
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5), 0.0)
    xmax = np.minimum(np.trunc(center + support + 0.5), float(in_size))
    count = (xmax - xmin).astype(np.int64)
    idx = np.arange(ksize, dtype=np.float64)[None, :]
    arg = np.abs((idx + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(arg < 1.0, 1.0 - arg, 0.0)
    w = np.where(idx < count[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for k in range(ksize):                      # Pillow accumulates the taps left to right
        ww = ww + w[:, k]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    coeffs = np.trunc(0.5 + w * float(1 << PRECISION_BITS)).astype(np.int32)    # triangle weights are >= 0
    bounds = np.stack((xmin.astype(np.int64), count), axis=1).astype(np.int32)
    return bounds, coeffs


def max_tile_rows(ybounds, tile=TILE_ROWS):
    """Largest number of input rows any ``tile`` consecutive output rows touch (sizes the kernel's LDS)."""
    first = ybounds[:, 0].astype(np.int64)
    last = first + ybounds[:, 1]
    n = len(first)
    starts = np.arange(0, n, tile)
    ends = np.minimum(starts + tile, n) - 1
    return int((last[ends] - first[starts]).max())


class FramePreprocessor(object):
    """uint8 RGB frame ``[H,W,3]`` -> fp32 ``[3,oh,ow]`` on ``device``; drop-in for the reference's
    ``transform(PIL image)`` at test time (no augmentation)."""

    def __init__(self, min_size, max_size, size_divisibility=32, pixel_mean=(0.485, 0.456, 0.406),
                 pixel_std=(0.229, 0.224, 0.225), to_bgr255=False, device="cuda"):
        self.min_size = int(min_size)
        self.max_size = None if max_size is None else int(max_size)
        self.size_divisibility = int(size_divisibility)
        self.pixel_mean = tuple(float(v) for v in pixel_mean)
        self.pixel_std = tuple(float(v) for v in pixel_std)
        self.to_bgr255 = bool(to_bgr255)
        self.device = torch.device(device)
        self._tables = {}
        self._staging = None

    @classmethod
    def from_cfg(cfg_cls, cfg, device="cuda"):
        """Keys of the reference's test-time transform (build_augmentation.py; defaults.py INPUT.*)."""
        inp = cfg.INPUT
        return cfg_cls(inp.MIN_SIZE_TEST, inp.MAX_SIZE_TEST, cfg.DATALOADER.SIZE_DIVISIBILITY, inp.PIXEL_MEAN,
                       inp.PIXEL_STD, inp.TO_BGR255, device=device)

    def get_size(self, image_wh):
        return get_size(image_wh, self.min_size, self.max_size, self.size_divisibility)

    def tables(self, in_hw, out_hw):
        key = (tuple(in_hw), tuple(out_hw))
        t = self._tables.get(key)
        if t is None:
            xb, xk = resample_tables(in_hw[1], out_hw[1])
            yb, yk = resample_tables(in_hw[0], out_hw[0])
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            t = (dev(xb), dev(xk), dev(yb), dev(yk), max_tile_rows(yb))
            self._tables[key] = t
        return t

    def upload(self, frame):
        """numpy / CPU uint8 frame -> device, through a reusable pinned staging buffer (asynchronous copy on the
        current stream; the staging buffer is reused only after that stream has consumed it)."""
        if isinstance(frame, np.ndarray):
            frame = torch.from_numpy(np.ascontiguousarray(frame))
        if frame.is_cuda:
            return frame
        if frame.dtype != torch.uint8:
            raise RuntimeError("siammot_amd.preprocess: frames must be uint8 RGB, got %s" % frame.dtype)
        if self._staging is None or self._staging.shape != frame.shape:
            self._staging = torch.empty(frame.shape, dtype=torch.uint8).pin_memory()
            self._staged_event = None
        if self._staged_event is not None:
            self._staged_event.synchronize()
        self._staging.copy_(frame)
        dev = self._staging.to(self.device, non_blocking=True)
        self._staged_event = torch.cuda.Event()
        self._staged_event.record()
        return dev

    def __call__(self, frame):
        frame = self.upload(frame)
        h, w = int(frame.shape[0]), int(frame.shape[1])
        out_hw = self.get_size((w, h))
        return ops.preprocess_frame(frame, self.tables((h, w), out_hw), out_hw, self.pixel_mean, self.pixel_std,
                                    self.to_bgr255)
