"""CPU oracle for the SiamMOT EMM tracker-head hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product path (``siam-mot_amd/``) never does and fails loudly
when its HIP library is missing.

This is a restatement, in plain PyTorch CPU ops (dtype-generic: fp32 = the reference's
arithmetic, fp64 = tie-break adjudicator), of the algorithm the reference runs for one
frame pair.  Every function cites the reference lines it follows (paths relative to the
reference root).  ``[UPSTREAM]`` marks semantics that live in facebookresearch/
maskrcnn-benchmark (un-vendored, un-pinned dependency: readme/INSTALL.md:89-92), restated
from that project's published algorithm (csrc/cpu/ROIAlign_cpu.cpp, modeling/poolers.py,
modeling/make_layers.py, structures/bounding_box.py).

PARITY PINNING: the reference ships no tests, fixtures or golden vectors (SURVEY.md §4,
§8c) → by the reference's own suite parity is UNPINNED.  What pins this file instead:
``oracle/gen_golden.py`` imports the reference's own ``xcorr.py`` / ``track_core.py`` /
``sr_pool.py`` / ``feature_extractor.py`` / ``track_utils.py`` UNMODIFIED from
/root/reference (with import stubs for the six absent maskrcnn_benchmark symbols), runs
them on seeded inputs and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` holds this restatement to those vectors.  The [UPSTREAM]
pieces (ROIAlign, LevelMapper, conv3x3+GroupNorm) cannot be executed from upstream source
here; ROIAlign is additionally cross-checked against an independent ``grid_sample``
formulation in ``tests/test_oracle_golden.py``.
"""
import math

import torch
import torch.nn.functional as F

UP_SCALE = 16  # reference track_core.py:69-71 (scale_factor=16) and :73 (up_scale=16)


# ----------------------------------------------------------------------------------------
# geometry (reference siammot/modelling/track_head/track_utils.py)
# ----------------------------------------------------------------------------------------
def pad_cells(pad_pixels, level):
    """Zero-padding, in feature cells, of FPN level ``level`` (track_utils.py:97-99)."""
    return int(pad_pixels / ((2 ** level) * 4))


def pad_features(features, pad_pixels):
    """``TrackUtils.pad_feature`` (track_utils.py:87-107): physical zero padding of every level."""
    out = []
    for i, f in enumerate(features):
        p = pad_cells(pad_pixels, i)
        out.append(F.pad(f, [p, p, p, p], mode="constant", value=0))
    return tuple(out)


def search_region(boxes, pad_pixels, search_expansion, min_search_wh):
    """SR boxes in padded-image coordinates from template boxes ``[N,4]`` (xyxy).

    ``update_boxes_in_pad_images`` (track_utils.py:109-135) then ``extend_bbox``
    (track_utils.py:62-85); ``search_expansion`` = SEARCH_REGION - 1 (track_utils.py:260).
    """
    b = boxes + pad_pixels
    w = b[:, 2] - b[:, 0] + 1
    h = b[:, 3] - b[:, 1] + 1
    w_ext = w * (search_expansion / 2.0)
    h_ext = h * (search_expansion / 2.0)
    min_w_ext = (min_search_wh - w) / (search_expansion * 2.0)
    min_h_ext = (min_search_wh - h) / (search_expansion * 2.0)
    w_ext = torch.max(min_w_ext, w_ext)
    h_ext = torch.max(min_h_ext, h_ext)
    return torch.stack((b[:, 0] - w_ext, b[:, 1] - h_ext, b[:, 2] + w_ext, b[:, 3] + h_ext), dim=1)


def level_mapper(boxes, k_min=2, k_max=5, canonical_scale=224.0, canonical_level=4.0, eps=1e-6):
    """[UPSTREAM] ``LevelMapper.__call__`` (modeling/poolers.py), used at sr_pool.py:38,74.

    ``area`` carries the upstream +1 convention; result is the 0-based pooler index.
    """
    area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    s = torch.sqrt(area)
    lvl = torch.floor(canonical_level + torch.log2(s / canonical_scale + eps))
    lvl = torch.clamp(lvl, min=k_min, max=k_max)
    return lvl.to(torch.int64) - k_min


# ----------------------------------------------------------------------------------------
# ROIAlign (legacy, non-"aligned")  [UPSTREAM csrc/cpu/ROIAlign_cpu.cpp]
# ----------------------------------------------------------------------------------------
def _axis_samples(start, bin_size, n_bins, grid, size, dtype):
    """Per-axis sample bookkeeping: returns (valid, low, high, w_low, w_high), each [n_bins*grid]."""
    p = torch.arange(n_bins, dtype=dtype).repeat_interleave(grid)
    i = torch.arange(grid, dtype=dtype).repeat(n_bins)
    c = start + p * bin_size + (i + 0.5) * bin_size / grid
    valid = ~((c < -1.0) | (c > size))
    c = torch.clamp(c, min=0)
    low = c.to(torch.int64)
    at_edge = low >= size - 1
    low = torch.where(at_edge, torch.full_like(low, size - 1), low)
    high = torch.where(at_edge, low, low + 1)
    c = torch.where(at_edge, low.to(dtype), c)
    l = c - low.to(dtype)
    h = 1.0 - l
    return valid, low, high, l, h


def roi_align(feat, rois, spatial_scale, out_h, out_w, sampling_ratio):
    """[UPSTREAM] ``ROIAlign.forward`` → ``_C.roi_align_forward`` as called at sr_pool.py:28-31,89.

    feat ``[B,C,H,W]``, rois ``[R,5]`` = (batch_idx, x1, y1, x2, y2) → ``[R,C,out_h,out_w]``.
    Mean of ``grid²`` bilinear samples per bin; no half-pixel shift; roi sides clamped to ≥1.
    """
    dtype = feat.dtype
    _, C, H, W = feat.shape
    R = rois.shape[0]
    out = feat.new_zeros((R, C, out_h, out_w))
    for r in range(R):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = [rois[r, k].to(dtype) * spatial_scale for k in range(1, 5)]
        roi_w = torch.clamp(x2 - x1, min=1.0)
        roi_h = torch.clamp(y2 - y1, min=1.0)
        bin_h = roi_h / out_h
        bin_w = roi_w / out_w
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(roi_h) / out_h))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(roi_w) / out_w))
        vy, yl, yh, ly, hy = _axis_samples(y1, bin_h, out_h, gh, H, dtype)
        vx, xl, xh, lx, hx = _axis_samples(x1, bin_w, out_w, gw, W, dtype)
        f = feat[b]
        # four corner gathers, each [C, out_h*gh, out_w*gw]
        v1 = f[:, yl][:, :, xl]
        v2 = f[:, yl][:, :, xh]
        v3 = f[:, yh][:, :, xl]
        v4 = f[:, yh][:, :, xh]
        w1 = hy[:, None] * hx[None, :]
        w2 = hy[:, None] * lx[None, :]
        w3 = ly[:, None] * hx[None, :]
        w4 = ly[:, None] * lx[None, :]
        val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
        val = val * (vy[:, None] & vx[None, :]).to(dtype)
        val = val.reshape(C, out_h, gh, out_w, gw)
        acc = feat.new_zeros((C, out_h, out_w))
        for iy in range(gh):          # same accumulation order as the upstream loops
            for ix in range(gw):
                acc = acc + val[:, :, iy, :, ix]
        out[r] = acc / (gh * gw)
    return out


# The same operator as compiled C (oracle/csrc/roi_align_cpu.c: upstream's published CPU algorithm, OpenMP over rois):
# bit-identical to ``roi_align`` above (tests/test_oracle_golden.py) and ~20x faster — bench.py's cpu_baseline leg times
# the reference's algorithm at the speed its C++ operator would have, not at a Python restatement's.
_ROI_C = {}


def roi_align_c_library():
    """ctypes handle of oracle/_build/libroi_align_cpu.so (built by ``__graft_entry__.build()`` /
    ``oracle/build_c.py``), or None when it has not been built."""
    if "lib" not in _ROI_C:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libroi_align_cpu.so")
        lib = None
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.roi_align_forward_cpu.restype = ctypes.c_int
            lib.roi_align_forward_cpu.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int,
                                                  ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _ROI_C["lib"] = lib
    return _ROI_C["lib"]


def roi_align_c(feat, rois, spatial_scale, out_h, out_w, sampling_ratio):
    """``roi_align`` through the compiled C restatement (fp32 CPU tensors)."""
    lib = roi_align_c_library()
    if lib is None:
        raise RuntimeError("oracle/_build/libroi_align_cpu.so is not built (python oracle/build_c.py)")
    feat = feat.contiguous().float()
    rois = rois.contiguous().float()
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = torch.empty((R, C, out_h, out_w), dtype=torch.float32)
    if R:
        rc = lib.roi_align_forward_cpu(feat.data_ptr(), B, C, H, W, rois.data_ptr(), R, float(spatial_scale), out_h, out_w,
                                       int(sampling_ratio), out.data_ptr())
        if rc:
            raise RuntimeError("roi_align_forward_cpu: roi with a batch index outside the input")
    return out


def sr_pool(features, boxes, sr_boxes, out_size, scales, sampling_ratio, roi_align=None):
    """``SRPooler.forward`` (sr_pool.py:53-91): level from the TEMPLATE box, rois from ``sr_boxes``
    (or the template boxes when ``sr_boxes is None``); one image per call (batch idx 0).
    ``roi_align``: the per-level operator (default: the torch restatement above; ``roi_align_c``: compiled)."""
    roi_align = globals()["roi_align"] if roi_align is None else roi_align
    rois_xyxy = boxes if sr_boxes is None else sr_boxes
    N = rois_xyxy.shape[0]
    rois = torch.cat((rois_xyxy.new_zeros((N, 1)), rois_xyxy), dim=1)
    if len(scales) == 1:
        return roi_align(features[0], rois, scales[0], out_size, out_size, sampling_ratio)
    k_min = -math.log2(scales[0])
    k_max = -math.log2(scales[-1])
    levels = level_mapper(boxes, k_min=k_min, k_max=k_max)
    C = features[0].shape[1]
    out = features[0].new_zeros((N, C, out_size, out_size))
    for lvl, scale in enumerate(scales):      # zip(x, poolers) truncates to len(scales) levels
        idx = torch.nonzero(levels == lvl).squeeze(1)
        if idx.numel() == 0:
            continue
        out[idx] = roi_align(features[lvl], rois[idx], scale, out_size, out_size, sampling_ratio)
    return out


# ----------------------------------------------------------------------------------------
# depthwise cross-correlation (reference EMM/xcorr.py:37-46)
# ----------------------------------------------------------------------------------------
def xcorr_depthwise(x, z):
    """``out[n,c,i,j] = Σ_{u,v} x[n,c,i+u,j+v]·z[n,c,u,v]`` (valid, stride 1), taps summed
    u-major / v-minor — the same order the HIP kernel uses."""
    Rx, Rz = x.shape[-1], z.shape[-1]
    Ho = Rx - Rz + 1
    out = x.new_zeros(x.shape[:2] + (Ho, Ho))
    for u in range(Rz):
        for v in range(Rz):
            out = out + x[:, :, u:u + Ho, v:v + Ho] * z[:, :, u:u + 1, v:v + 1]
    return out


def xcorr_depthwise_conv(x, z):
    """The reference's own formulation (xcorr.py:37-46): one grouped ``F.conv2d`` with
    groups = N·C.  Used by bench.py's ``cpu_baseline`` leg so the CPU number times the ops the
    reference really runs; tests hold it equal to the explicit form above."""
    n, c = z.shape[:2]
    out = F.conv2d(x.reshape(1, n * c, x.shape[2], x.shape[3]),
                   z.reshape(n * c, 1, z.shape[2], z.shape[3]), groups=n * c)
    return out.reshape(n, c, out.shape[2], out.shape[3])


# ----------------------------------------------------------------------------------------
# predictor (reference EMM/feature_extractor.py:43-69; make_conv3x3 / group_norm [UPSTREAM])
# ----------------------------------------------------------------------------------------
PREDICTOR_KEYS = (
    "cls_tower.0.weight", "cls_tower.1.weight", "cls_tower.1.bias",
    "reg_tower.0.weight", "reg_tower.1.weight", "reg_tower.1.bias",
    "cls.weight", "cls.bias", "center.weight", "center.bias", "reg.weight", "reg.bias",
)


def predictor(resp, params, gn_groups=32, gn_eps=1e-5):
    """``EMMPredictor.forward`` (feature_extractor.py:62-69).  ``params`` uses the reference's
    state_dict keys.  Towers: conv3×3 (pad 1, no bias) → GroupNorm(32) → ReLU; heads: conv3×3+bias."""
    def tower(name):
        y = F.conv2d(resp, params[name + ".0.weight"], None, padding=1)
        y = F.group_norm(y, gn_groups, params[name + ".1.weight"], params[name + ".1.bias"], gn_eps)
        return F.relu(y)
    cls_x = tower("cls_tower")
    reg_x = tower("reg_tower")
    cls = F.conv2d(cls_x, params["cls.weight"], params["cls.bias"], padding=1)
    center = F.conv2d(cls_x, params["center.weight"], params["center.bias"], padding=1)
    reg = F.relu(F.conv2d(reg_x, params["reg.weight"], params["reg.bias"], padding=1))
    return cls, center, reg


# ----------------------------------------------------------------------------------------
# bicubic ×16 up-sampling (track_core.py:69-71; torch upsample_bicubic2d, align_corners=False)
# ----------------------------------------------------------------------------------------
def _cubic_taps(n_in, scale, dtype):
    """Tap indices [4, n_out] (border-clamped) and weights [4, n_out]; A = -0.75."""
    A = -0.75
    d = torch.arange(n_in * scale, dtype=dtype)
    src = (d + 0.5) * (1.0 / scale) - 0.5
    f = torch.floor(src)
    t = src - f
    x1, x2 = t, 1.0 - t
    w = torch.stack((
        ((A * (x1 + 1.0) - 5.0 * A) * (x1 + 1.0) + 8.0 * A) * (x1 + 1.0) - 4.0 * A,
        ((A + 2.0) * x1 - (A + 3.0)) * x1 * x1 + 1.0,
        ((A + 2.0) * x2 - (A + 3.0)) * x2 * x2 + 1.0,
        ((A * (x2 + 1.0) - 5.0 * A) * (x2 + 1.0) + 8.0 * A) * (x2 + 1.0) - 4.0 * A,
    ))
    base = f.to(torch.int64)
    idx = torch.stack([torch.clamp(base - 1 + k, 0, n_in - 1) for k in range(4)])
    return idx, w


def bicubic_upsample(planes, scale=UP_SCALE):
    """``F.interpolate(planes, scale_factor=scale, mode='bicubic')``: per output pixel, four
    horizontal 4-tap interpolations then one vertical 4-tap."""
    H, W = planes.shape[-2:]
    iy, wy = _cubic_taps(H, scale, planes.dtype)
    ix, wx = _cubic_taps(W, scale, planes.dtype)
    out = None
    for k in range(4):
        rows = planes[:, :, iy[k], :]                       # [N,C,H*scale,W]
        h = None
        for j in range(4):
            term = rows[:, :, :, ix[j]] * wx[j]
            h = term if h is None else h + term
        h = h * wy[k][:, None]
        out = h if out is None else out + h
    return out


def bicubic_upsample_torch(planes, scale=UP_SCALE):
    """``F.interpolate(..., mode='bicubic')`` itself (track_core.py:69-71) — cpu_baseline leg."""
    return F.interpolate(planes, scale_factor=scale, mode="bicubic")


# ----------------------------------------------------------------------------------------
# locations + decode (track_core.py:184-225, :101-162)
# ----------------------------------------------------------------------------------------
def grid_axes(sr_boxes, rx, rz, pad_pixels, scale=UP_SCALE):
    """Per-track pixel coordinates of the up-sampled response grid, un-padded image space.

    ``get_locations`` (track_core.py:184-225): ``x_k = sr.x1 + (st+k)·(sr.x2-sr.x1)/(rx·scale-1) - pad``
    for ``k ∈ [0, (rx-2·⌊rz/2⌋)·scale)``, ``st = ⌊rz/2⌋·scale``.  Returns (xs [N,G], ys [N,G]).
    """
    full = rx * scale
    st = int(math.floor(rz / 2)) * scale
    k = torch.arange(0, full, dtype=torch.float32).to(sr_boxes.dtype)
    stride_w = (sr_boxes[:, 2] - sr_boxes[:, 0]) / (full - 1)
    stride_h = (sr_boxes[:, 3] - sr_boxes[:, 1]) / (full - 1)
    xs = sr_boxes[:, 0:1] + k[None, :] * stride_w[:, None]
    ys = sr_boxes[:, 1:2] + k[None, :] * stride_h[:, None]
    xs = xs[:, st:full - st] - pad_pixels
    ys = ys[:, st:full - st] - pad_pixels
    return xs, ys


def locations(sr_boxes, rx, rz, pad_pixels, scale=UP_SCALE):
    """The materialised ``[N, G*G, 2]`` tensor of ``get_locations`` (row-major, y outer)."""
    xs, ys = grid_axes(sr_boxes, rx, rz, pad_pixels, scale)
    G = xs.shape[1]
    X = xs[:, None, :].expand(-1, G, -1).reshape(xs.shape[0], -1)
    Y = ys[:, :, None].expand(-1, -1, G).reshape(xs.shape[0], -1)
    return torch.stack((X, Y), dim=2)


def score_map(cls_up, center_up, reg_up, boxes, use_centerness=True, sigma=0.4):
    """Penalised confidence ``[N, G*G]`` and class-1 probability ``[N, G*G]``
    (track_core.py:101-118, :138-162)."""
    N = cls_up.shape[0]
    G = cls_up.shape[-1]
    p = F.softmax(cls_up, dim=1)[:, 1:2]
    conf = p * torch.sigmoid(center_up) if use_centerness else p
    conf = conf.reshape(N, -1)
    tlbr = reg_up.reshape(N, 4, -1)
    box_w = boxes[:, 2] - boxes[:, 0]
    box_h = boxes[:, 3] - boxes[:, 1]
    r_w = tlbr[:, 2] + tlbr[:, 0]
    r_h = tlbr[:, 3] + tlbr[:, 1]
    s_w = r_w / box_w[:, None]
    s_h = r_h / box_h[:, None]
    s_w = torch.max(s_w, 1 / s_w)
    s_h = torch.max(s_h, 1 / s_h)
    penalty = torch.exp((-s_w * s_h + 1) * 0.1)
    hann = torch.hann_window(G, dtype=torch.float).to(cls_up.dtype)      # periodic
    window = torch.outer(hann, hann).reshape(-1)
    score = (conf * penalty) * (1 - sigma) + sigma * window[None, :]
    return score, p.reshape(N, -1)


def decode(cls_up, center_up, reg_up, xs, ys, boxes, use_centerness=True, sigma=0.4):
    """``decode_response`` (track_core.py:101-135) → (bb [N,4], bb_conf [N], idx [N])."""
    N = cls_up.shape[0]
    G = cls_up.shape[-1]
    score, p = score_map(cls_up, center_up, reg_up, boxes, use_centerness, sigma)
    idx = torch.argmax(score, dim=1)
    n = torch.arange(N)
    iy = torch.div(idx, G, rounding_mode="floor")
    ixx = idx - iy * G
    cx = xs[n, ixx]
    cy = ys[n, iy]
    tlbr = reg_up.reshape(N, 4, -1)[n, :, idx]
    bb = torch.stack((cx - tlbr[:, 0], cy - tlbr[:, 1], cx + tlbr[:, 2], cy + tlbr[:, 3]), dim=1)
    return bb, p[n, idx], idx


def clip_boxes(bb, conf, image_wh):
    """The clipping step of wrap_results_to_boxlist (track_core.py:177-178).

    The reference calls ``track_box.clip_to_image(remove_empty=True)`` and DISCARDS the return
    value: [UPSTREAM] BoxList.clip_to_image clamps ``self.bbox`` in place (TO_REMOVE=1) and only
    the returned copy drops empty boxes — so boxes are clamped but never removed.  Returns the
    clamped boxes, conf unchanged, and the (unused by the reference) non-empty mask.
    """
    w, h = image_wh
    bb = bb.clone()
    bb[:, 0].clamp_(min=0, max=w - 1)
    bb[:, 1].clamp_(min=0, max=h - 1)
    bb[:, 2].clamp_(min=0, max=w - 1)
    bb[:, 3].clamp_(min=0, max=h - 1)
    nonempty = (bb[:, 3] > bb[:, 1]) & (bb[:, 2] > bb[:, 0])
    return bb, conf, nonempty


# ----------------------------------------------------------------------------------------
# the two entry points of the boundary (track_core.py:28-98)
# ----------------------------------------------------------------------------------------
class EMMConfig(object):
    """The cfg values the path reads (defaults: siammot/configs/defaults.py:35-82)."""

    def __init__(self, channels=128, rz=15, search_region=2.0, sampling_ratio=2,
                 scales=(0.25, 0.125, 0.0625, 0.03125), pad_pixels=512, min_search_wh=0,
                 use_centerness=True, sigma=0.4, amodal=False, gn_groups=32, gn_eps=1e-5):
        self.channels = channels
        self.rz = rz
        self.rx = int(rz * search_region)                     # feature_extractor.py:27
        self.search_expansion = search_region - 1.0           # track_utils.py:260
        self.sampling_ratio = sampling_ratio
        self.scales = tuple(scales)
        self.pad_pixels = pad_pixels
        self.min_search_wh = min_search_wh
        self.use_centerness = use_centerness
        self.sigma = sigma
        self.amodal = amodal
        self.gn_groups = gn_groups
        self.gn_eps = gn_eps


def extract_cache(cfg, features, det_boxes, roi_align=None):
    """``EMM.extract_cache`` (track_core.py:81-98): template ROIAlign on UNPADDED features +
    search regions.  Returns (z [N,C,rz,rz], sr [N,4])."""
    z = sr_pool(features, det_boxes, None, cfg.rz, cfg.scales, cfg.sampling_ratio, roi_align)
    sr = search_region(det_boxes, cfg.pad_pixels, cfg.search_expansion, cfg.min_search_wh)
    return z, sr


def emm_forward(cfg, params, features, boxes, sr_boxes, template_features, image_wh,
                return_intermediates=False, reference_ops=False, roi_align=None):
    """Inference branch of ``EMM.forward`` (track_core.py:28-79).

    Returns (bb [N,4], conf [N], nonempty [N] bool) after ``wrap_results_to_boxlist`` clipping
    (clamp only — see clip_boxes), plus a dict of every intermediate when asked.
    ``reference_ops=True`` swaps in the torch library calls the reference makes (grouped conv2d,
    F.interpolate) for the explicit restatements — same results to fp32 rounding, much faster on
    CPU (bench.py's cpu_baseline leg).
    """
    xcorr = xcorr_depthwise_conv if reference_ops else xcorr_depthwise
    upsample = bicubic_upsample_torch if reference_ops else bicubic_upsample
    padded = pad_features(features, cfg.pad_pixels)                                   # :49
    x = sr_pool(padded, boxes, sr_boxes, cfg.rx, cfg.scales, cfg.sampling_ratio, roi_align)      # :51
    resp = xcorr(x, template_features)                                                # :53
    cls, center, reg = predictor(resp, params, cfg.gn_groups, cfg.gn_eps)             # :54
    cls_up = upsample(cls)                                                            # :69
    center_up = upsample(center)                                                      # :70
    reg_up = upsample(reg)                                                            # :71
    xs, ys = grid_axes(sr_boxes, cfg.rx, cfg.rz, cfg.pad_pixels)                      # :73
    bb, conf, idx = decode(cls_up, center_up, reg_up, xs, ys, boxes,
                           cfg.use_centerness, cfg.sigma)                             # :76-77
    if cfg.amodal:
        keep = torch.ones(bb.shape[0], dtype=torch.bool)
        out = (bb, conf, keep)
    else:
        out = clip_boxes(bb, conf, image_wh)                                          # :78
    if return_intermediates:
        inter = dict(sr_features=x, response=resp, cls=cls, center=center, reg=reg,
                     bb_raw=bb, conf_raw=conf, idx=idx)
        return out + (inter,)
    return out
