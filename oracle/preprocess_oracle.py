"""CPU oracle for the inference pre-processing chain — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path
(siam-mot_amd/) never does.

Restates what the reference runs per frame on the CPU before the detector sees it:
  demos/demo_inference.py:74-82  (_preprocess: PIL image -> self.transform)
  siammot/data/adapters/augmentation/build_augmentation.py:52-66  (test-time chain: SiamVideoResize ->
      ToTensor -> Normalize(mean, std, to_bgr255))
  siammot/data/adapters/augmentation/image_augmentation.py:21-50  (ImageResize.get_size / __call__:
      torchvision F.resize(PIL image, (oh, ow)) = PIL.Image.resize(..., BILINEAR))
  [UPSTREAM maskrcnn_benchmark data/transforms/transforms.py] ToTensor (uint8 HWC -> float CHW / 255) and
      Normalize (optionally image[[2,1,0]] * 255, then (x - mean) / std)

The resize restates Pillow's ImagingResample for 8-bit images (src/libImaging/Resample.c [THIRD PARTY, Pillow;
the version in this image is pinned by tests against PIL itself]): separable, horizontal pass first, triangle
filter whose support grows with the down-scaling factor (anti-aliasing), coefficients normalised in double
precision and quantised to 22 fractional bits, integer accumulation started at one half, result clipped to
[0,255] — with a uint8 intermediate image between the passes.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def get_size(image_wh, min_size, max_size, size_divisibility):
    """image_augmentation.py:21-42 — returns (oh, ow)."""
    w, h = image_wh
    size = min_size
    if max_size is not None:
        min_original_size = float(min((w, h)))
        max_original_size = float(max((w, h)))
        if max_original_size / min_original_size * size > max_size:
            size = int(round(max_size * min_original_size / max_original_size))
    if w < h:
        ow = size
        oh = int(size * h / w)
    else:
        oh = size
        ow = int(size * w / h)
    if size_divisibility > 0:
        oh = int(oh / size_divisibility) * size_divisibility
        ow = int(ow / size_divisibility) * size_divisibility
    return oh, ow


def resample_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter over the whole axis.
    Returns (bounds int32 [out,2] = (first input index, tap count), coeffs int32 [out, ksize])."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] /= ww
        for x in range(ksize):
            v = k[x] * (1 << PRECISION_BITS)
            coeffs[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs


def _resample_axis0(img, out_size):
    """Resample axis 0 of a uint8 array [in, ...] -> uint8 [out, ...]."""
    in_size = img.shape[0]
    bounds, coeffs = resample_coeffs(in_size, out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * int(coeffs[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(frame, out_hw):
    """PIL.Image.resize((ow, oh), BILINEAR) on an RGB uint8 HWC array: horizontal pass, then vertical."""
    oh, ow = out_hw
    h, w, _ = frame.shape
    img = frame
    if ow != w:
        img = np.ascontiguousarray(np.swapaxes(_resample_axis0(np.swapaxes(img, 0, 1), ow), 0, 1))
    if oh != h:
        img = _resample_axis0(img, oh)
    return img


def to_tensor_normalize(img_u8, mean, std, to_bgr255):
    """ToTensor + Normalize in fp32, op by op: x/255 ; [BGR, *255] ; -mean ; /std.  Returns [3, H, W] float32."""
    x = np.transpose(img_u8, (2, 0, 1)).astype(np.float32) / np.float32(255)
    if to_bgr255:
        x = x[[2, 1, 0]] * np.float32(255)
    mean = np.asarray(mean, dtype=np.float32)[:, None, None]
    std = np.asarray(std, dtype=np.float32)[:, None, None]
    return ((x - mean) / std).astype(np.float32)


def preprocess(frame, min_size, max_size, size_divisibility, mean, std, to_bgr255):
    """uint8 RGB HWC frame -> float32 [3, oh, ow] network input."""
    h, w, _ = frame.shape
    oh, ow = get_size((w, h), min_size, max_size, size_divisibility)
    return to_tensor_normalize(resize_bilinear_u8(frame, (oh, ow)), mean, std, to_bgr255)
