"""ctypes binding of ``csrc/libsmot_emm.so`` (C ABI: include/smot_emm.h) + tensor-level operators.

This is the ONLY compute path of the package: there is no CPU / eager fallback.  If the HIP
library is missing or a tensor is not a contiguous fp32 device tensor, the call raises.

Operator ↔ reference map (reference paths relative to amazon-science/siam-mot):
    roi_align_levels   SRPooler.forward            EMM/sr_pool.py:53-91 (+ track_utils.py:87-107 pad)
    search_region      update_boxes_in_pad_images  track_head/track_utils.py:109-135 + extend_bbox :62-85
    xcorr_depthwise    xcorr_depthwise             EMM/xcorr.py:37-46
    emm_predictor      EMMPredictor.forward        EMM/feature_extractor.py:62-69
    emm_decode         3x F.interpolate + get_locations + decode_response   EMM/track_core.py:69-77
"""
import contextlib
import ctypes
import struct
import weakref
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsmot_emm.so")
# measurement build (-DSMOT_DEBUG): older kernel generations, A/B switches, timing ablations.  Never loaded
# implicitly — only through ``debug_library()`` (tools/, A/B tests).
DEBUG_LIB_PATH = os.path.join(_HERE, "csrc", "libsmot_emm_debug.so")
ABI_VERSION = 13
UP_SCALE = 16          # reference track_core.py:69-73


_lib = None
_c_float_p = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float

_SIGNATURES = {
    "smot_abi_version": (ctypes.c_int, []),
    "smot_build_info": (ctypes.c_int, []),
    "smot_last_error": (ctypes.c_char_p, []),
    "smot_roi_align_levels_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i,
                                                 _vp, _vp, _vp]),
    "smot_roi_align_fwd": (ctypes.c_int, [_vp, _i, _i, _i, _i, _i, _vp, _i, _f, _i, _i, _i, _vp, _vp]),
    "smot_search_region_fwd": (ctypes.c_int, [_vp, _i, _f, _f, _f, _vp, _vp]),
    "smot_xcorr_dw_fwd": (ctypes.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "smot_emm_predictor_fwd": (ctypes.c_int, [_vp, _i, _i, _i] + [_vp] * 12 + [_i, _f, _vp, _vp, _vp, _vp]),
    "smot_debug_trace": (None, [_vp]),
    "smot_emm_tower_pack_floats": (ctypes.c_longlong, [_i]),
    "smot_emm_tower_form": (ctypes.c_int, [_i, _i, _i]),
    "smot_emm_tower_pack": (ctypes.c_int, [_vp, _vp, _i, _vp, _vp]),
    "smot_emm_decode_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _i, _f, _f,
                                           _vp, _vp, _vp, _vp, _vp]),
    "smot_emm_decode_ws_floats": (ctypes.c_int, [_i, _i]),
    "smot_sr_xcorr_fused_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i,
                                               _vp, _vp, _vp]),
    "smot_sr_xcorr_gather_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "smot_preprocess_fwd": (ctypes.c_int, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "smot_nms_ws_bytes": (ctypes.c_longlong, [_i]),
    "smot_nms_fwd": (ctypes.c_int, [_vp, _i, _f, _vp, _vp, _vp]),
    "smot_xcorr_timer_begin": (ctypes.c_int, [_i]),
    "smot_xcorr_timer_end": (ctypes.c_int, [_vp, _vp]),
    "smot_kernel_timer_begin": (ctypes.c_int, [_i, _i, _i]),
    "smot_kernel_timer_bracket_overhead": (ctypes.c_int, [_vp, _i, _vp]),
    "smot_dispatch_floor_fwd": (ctypes.c_int, [_i, _i, _vp]),
    "smot_kernel_timer_end": (ctypes.c_int, [_i, _vp, _vp]),
    "smot_emm_track_ws_floats": (ctypes.c_longlong, [_i, _i, _i, _i]),
    "smot_emm_track_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i,
                                          _vp, _i, _f, _vp, _i, _f, _f, _f, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "smot_emm_order_hint_floats": (ctypes.c_longlong, [_i, _i, _i]),
    "smot_emm_extract_cache_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _f, _f, _f,
                                                  _vp, _vp, _vp, _vp]),
    "smot_emm_extract_cache_masked_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _f, _f, _f,
                                                         _vp, _vp, _vp, _vp]),
    "smot_box_refine_post_max_rows": (ctypes.c_int, []),
    "smot_box_refine_post_fwd": (ctypes.c_int, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _f, _f, _i,
                                                _vp, _vp, _vp, _vp, _vp]),
    "smot_box_refine_ws_floats": (ctypes.c_longlong, [_i, _i, _i, _i, _i, _i, _i]),
    "smot_box_refine_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i,
                                           _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i,
                                           _f, _f, _f, _f, _f, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "smot_linear_rows_max_rows": (ctypes.c_int, []),
    "smot_linear_rows_ws_floats": (ctypes.c_longlong, [_i, _i, _i]),
    "smot_linear_rows_fwd": (ctypes.c_int, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "smot_track_frame_fwd": (ctypes.c_int, [_vp, _vp]),
    "smot_track_solve_max_boxes": (ctypes.c_int, []),
    "smot_emm_order_hint_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "smot_memory_carry_max_rows": (ctypes.c_int, []),
    "smot_memory_carry_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i,
                                             _vp, _i, _vp]),
    "smot_track_solve_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _i,
                                            _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "smot_track_solve_carry_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _i,
                                                  _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                  _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())
_DEBUG_SIGNATURES = {
    "smot_debug_set_knob": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "smot_debug_sr_xcorr_fused_hint_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
}
DEBUG_EXPORTED_SYMBOLS = tuple(_DEBUG_SIGNATURES.keys())


def load_library(path=None):
    """dlopen the HIP library (after torch, so both share one libamdhip64) and type its symbols."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            "siammot_amd: HIP library %s not found — build it with `python siam-mot_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    _lib = _open(path, _SIGNATURES)
    return _lib


def _open(path, signatures):
    lib = ctypes.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.smot_abi_version() != ABI_VERSION:
        raise RuntimeError("siammot_amd: %s has ABI %d, host layer expects %d"
                           % (path, lib.smot_abi_version(), ABI_VERSION))
    return lib


_KNOB_DEFAULTS = {"SMOT_NO_FUSE": "0", "SMOT_ROI_GENERIC": "0", "SMOT_TOWER_DIRECT": "0", "SMOT_TOWER_WIDE": "0",
                  "SMOT_DECODE_SPLIT": "0", "SMOT_XCORR_VARIANT": "default", "SMOT_DECODE_2PASS": "0",
                  "SMOT_FUSED_GEN": "0", "SMOT_TOWER_OCT": "0", "SMOT_TOWER_BF3": "1", "SMOT_FUSED_ORDER": "0", "SMOT_NO_HINT": "0", "SMOT_FUSED_ABL": "0",
                  "SMOT_ANY_ORDER": "0",
                  "SMOT_WINO_ABL": "0",
                  "SMOT_TOWER_ABL": "0"}


@contextlib.contextmanager
def debug_library(**knobs):
    """Measurement build as the current library inside the block (tools/ and A/B tests only), with the given
    switches set (``debug_library(SMOT_XCORR_VARIANT="mfma")``; names as in csrc/knobs.h) and every switch
    back at its default afterwards.  The product library is restored on exit."""
    global _lib
    if not os.path.exists(DEBUG_LIB_PATH):
        raise RuntimeError("siammot_amd: measurement library %s not built (python siam-mot_amd/build.py)" % DEBUG_LIB_PATH)
    load_library()                                  # make sure the product handle exists to return to
    dbg = _open(DEBUG_LIB_PATH, dict(_SIGNATURES, **_DEBUG_SIGNATURES))
    if not (dbg.smot_build_info() & 1):
        raise RuntimeError("siammot_amd: %s is not a measurement build" % DEBUG_LIB_PATH)
    prev, _lib = _lib, dbg

    def set_all(values):
        for k, v in values.items():
            if dbg.smot_debug_set_knob(k.encode(), str(v).encode()) != 0:
                raise RuntimeError("siammot_amd: " + dbg.smot_last_error().decode("utf-8", "replace"))
    try:
        set_all(_KNOB_DEFAULTS)
        set_all(knobs)
        _hint_floats.clear()                        # (answers of the library that depend on its switches)
        yield dbg
    finally:
        set_all(_KNOB_DEFAULTS)
        _lib = prev
        _hint_floats.clear()


def _check(rc, what):
    if rc != 0:
        msg = _lib.smot_last_error().decode("utf-8", "replace")
        raise RuntimeError("siammot_amd.%s failed (code %d): %s" % (what, rc, msg))


def _dev_f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("siammot_amd: %s must be a device (ROCm) tensor — no CPU path exists" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("siammot_amd: %s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream         # (device index) -> hipStream_t as int, ~0.2 us
except AttributeError:                                        # pragma: no cover - older torch
    _raw_stream = None


def _stream(device=None):
    """The current torch stream of ``device`` as a raw handle (launches are enqueued on it)."""
    if _raw_stream is not None:
        idx = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
        return ctypes.c_void_p(_raw_stream(idx))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class _Launch(object):
    """Launch context of one binding: every tensor must live on ONE device, that device is made current for
    the duration of the call (hipLaunchKernelGGL launches on the CURRENT device — a stream handle of another
    device would be invalid there), and ``stream`` is that device's current torch stream.  A no-op switch when
    the device is current already (the common case: ~0.5 us)."""
    __slots__ = ("dev", "prev", "stream")

    def __init__(self, *tensors):
        dev = None
        for t in tensors:
            if t is None:
                continue
            if not (isinstance(t, torch.Tensor) and t.is_cuda):
                raise RuntimeError("siammot_amd: inputs must be device (ROCm) tensors — no CPU path exists")
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise RuntimeError("siammot_amd: inputs live on different devices (%s and %s)" % (dev, t.device))
        if dev is None:
            raise RuntimeError("siammot_amd: no device tensor among the inputs")
        self.dev = dev
        self.prev = None
        self.stream = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.dev.index:
            self.prev = cur
            torch.cuda.set_device(self.dev.index)
        self.stream = _stream(self.dev)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


# ----------------------------------------------------------------------------------------------
def roi_align_levels(features, rois, level_boxes, out_size, scales, sampling_ratio, pad_cells=None,
                     return_levels=False):
    """Level-routed legacy ROIAlign on one image with virtual zero padding.

    features: sequence of ``[1,C,H_l,W_l]`` tensors (only the first ``len(scales)`` are used, as the
    reference's ``zip(x, self.poolers)``); rois ``[R,4]`` xyxy in padded-image pixels; level_boxes
    ``[R,4]`` (template boxes) choose the level; pad_cells: per-level virtual padding in cells.
    Returns ``[R,C,out_size,out_size]`` (and int32 levels when asked).
    """
    lib = load_library()
    L = len(scales)
    feats = [_dev_f32(features[l], "features[%d]" % l) for l in range(L)]
    for f in feats:
        if f.dim() != 4 or f.shape[0] != 1:
            raise RuntimeError("siammot_amd.roi_align_levels: one image per call, got feature shape %s"
                               % (tuple(f.shape),))
    C = feats[0].shape[1]
    rois = _dev_f32(rois, "rois")
    level_boxes = rois if level_boxes is None else _dev_f32(level_boxes, "level_boxes")
    R = rois.shape[0]
    if pad_cells is None:
        pad_cells = [0] * L
    out = torch.empty((R, C, out_size, out_size), dtype=torch.float32, device=rois.device)
    levels = torch.empty((R,), dtype=torch.int32, device=rois.device) if return_levels else None
    fp = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    hs = (ctypes.c_int * L)(*[f.shape[2] for f in feats])
    ws = (ctypes.c_int * L)(*[f.shape[3] for f in feats])
    pc = (ctypes.c_int * L)(*[int(p) for p in pad_cells[:L]])
    sc = (ctypes.c_float * L)(*[float(s) for s in scales])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    with _Launch(rois, level_boxes, *feats) as ln:
        rc = lib.smot_roi_align_levels_fwd(cast(fp), cast(hs), cast(ws), cast(pc), cast(sc), L, C,
                                           _ptr(rois), _ptr(level_boxes), R, out_size, out_size,
                                           int(sampling_ratio), _ptr(out), _ptr(levels), ln.stream)
    _check(rc, "roi_align_levels")
    return (out, levels) if return_levels else out


def roi_align(input, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio, pad_cells=0):
    """[UPSTREAM] ``_C.roi_align_forward(input, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio)``: one feature
    level ``[B,C,H,W]``, rois ``[R,5]`` = (image index, x1, y1, x2, y2) -> ``[R,C,pooled_h,pooled_w]``, allocated here
    with the input's options as upstream does.  ``pad_cells`` > 0: virtual zero border (rois in padded coordinates)."""
    lib = load_library()
    input = _dev_f32(input, "input")
    rois = _dev_f32(rois, "rois")
    if input.dim() != 4 or rois.dim() != 2 or rois.shape[1] != 5:
        raise RuntimeError("siammot_amd.roi_align: input must be [B,C,H,W] and rois [R,5], got %s and %s"
                           % (tuple(input.shape), tuple(rois.shape)))
    B, C, H, W = input.shape
    R = rois.shape[0]
    out = torch.empty((R, C, int(pooled_h), int(pooled_w)), dtype=torch.float32, device=input.device)
    with _Launch(input, rois) as ln:
        rc = lib.smot_roi_align_fwd(_ptr(input), B, C, H, W, int(pad_cells), _ptr(rois), R, float(spatial_scale),
                                    int(pooled_h), int(pooled_w), int(sampling_ratio), _ptr(out), ln.stream)
    _check(rc, "roi_align")
    return out


def search_region(boxes, pad_pixels, search_expansion, min_search_wh):
    """Template boxes ``[N,4]`` → search regions ``[N,4]`` in padded-image coordinates."""
    lib = load_library()
    boxes = _dev_f32(boxes, "boxes")
    sr = torch.empty_like(boxes)
    with _Launch(boxes) as ln:
        rc = lib.smot_search_region_fwd(_ptr(boxes), boxes.shape[0], float(pad_pixels), float(search_expansion),
                                        float(min_search_wh), _ptr(sr), ln.stream)
    _check(rc, "search_region")
    return sr


def xcorr_depthwise(x, kernel):
    """Same contract as the reference ``xcorr_depthwise(x, kernel)`` (EMM/xcorr.py:37-46)."""
    lib = load_library()
    x = _dev_f32(x, "x")
    kernel = _dev_f32(kernel, "kernel")
    if x.dim() != 4 or kernel.dim() != 4 or x.shape[:2] != kernel.shape[:2] \
            or x.shape[2] != x.shape[3] or kernel.shape[2] != kernel.shape[3]:
        raise RuntimeError("siammot_amd.xcorr_depthwise: need x [N,C,Rx,Rx] and kernel [N,C,Rz,Rz], got %s, %s"
                           % (tuple(x.shape), tuple(kernel.shape)))
    N, C, Rx, _ = x.shape
    Rz = kernel.shape[2]
    out = torch.empty((N, C, Rx - Rz + 1, Rx - Rz + 1), dtype=torch.float32, device=x.device)
    with _Launch(x, kernel) as ln:
        rc = lib.smot_xcorr_dw_fwd(_ptr(x), _ptr(kernel), _ptr(out), N, C, Rx, Rz, ln.stream)
    _check(rc, "xcorr_depthwise")
    return out


PREDICTOR_KEYS = (
    "cls_tower.0.weight", "cls_tower.1.weight", "cls_tower.1.bias",
    "reg_tower.0.weight", "reg_tower.1.weight", "reg_tower.1.bias",
    "cls.weight", "cls.bias", "center.weight", "center.bias", "reg.weight", "reg.bias",
)


_TOWER_KEYS = ("cls_tower.0.weight", "reg_tower.0.weight")
_KEY_INDEX = {k: i for i, k in enumerate(PREDICTOR_KEYS)}
_pack_cache = {}


def tower_packed(params):
    """Winograd-transformed tower filters (``smot_emm_tower_pack``) for the matrix-core tower kernel, cached per
    pair of weight tensors and recomputed when either is modified in place (``load_state_dict``) or replaced
    (``.to()``).  ``None`` when the channel count has no packed path."""
    lib = load_library()
    wc = _dev_f32(params["cls_tower.0.weight"], "cls_tower.0.weight")
    wr = _dev_f32(params["reg_tower.0.weight"], "reg_tower.0.weight")
    C = wc.shape[0]
    nfl = lib.smot_emm_tower_pack_floats(C)
    if nfl == 0 or tuple(wc.shape) != (C, C, 3, 3) or tuple(wr.shape) != (C, C, 3, 3):
        return None
    # Keyed on the tensor OBJECTS (weak references), not on addresses: the caching allocator hands the address
    # of a freed weight tensor to the next model's weights, and version counters restart with every tensor.
    key = (wc.data_ptr(), wr.data_ptr())
    ver = (wc._version, wr._version)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is wc and hit[1]() is wr and hit[2] == ver:
        return hit[3]
    packed = torch.empty((nfl,), dtype=torch.float32, device=wc.device)
    with _Launch(wc, wr) as ln:
        _check(lib.smot_emm_tower_pack(_ptr(wc), _ptr(wr), C, _ptr(packed), ln.stream), "tower_pack")
    # the image is cached and may be consumed from another stream: finish it now (once per weight set)
    torch.cuda.current_stream(wc.device).synchronize()
    for k in [k for k, v in _pack_cache.items() if v[0]() is None or v[1]() is None]:
        del _pack_cache[k]
    _pack_cache[key] = (weakref.ref(wc), weakref.ref(wr), ver, packed)
    return packed


def emm_predictor(resp, params, gn_groups=32, gn_eps=1e-5, winograd=True):
    """``resp [N,C,Ho,Ho]`` + reference-keyed ``params`` → logits ``[N,7,Ho,Ho]``
    (cls0, cls1, center, reg l/t/r/b; reg already ReLU'd).  ``winograd=False`` forces the direct fp32
    tower kernel (same result up to fp32 rounding order)."""
    lib = load_library()
    resp = _dev_f32(resp, "resp")
    N, C, Ho, _ = resp.shape
    w = [_dev_f32(params[k], k) for k in PREDICTOR_KEYS]
    expect = {0: (C, C, 3, 3), 3: (C, C, 3, 3), 6: (2, C, 3, 3), 8: (1, C, 3, 3), 10: (4, C, 3, 3)}
    for i, shp in expect.items():
        if tuple(w[i].shape) != shp:
            raise RuntimeError("siammot_amd.emm_predictor: %s has shape %s, expected %s"
                               % (PREDICTOR_KEYS[i], tuple(w[i].shape), shp))
    tower_ws = torch.empty((N, 2 * C, Ho, Ho), dtype=torch.float32, device=resp.device)
    logits = torch.empty((N, 7, Ho, Ho), dtype=torch.float32, device=resp.device)
    packed = tower_packed(params) if (winograd and Ho in (16, 29)) else None       # (29: Winograd in 16 x 16 blocks)
    with _Launch(resp, *w) as ln:
        rc = lib.smot_emm_predictor_fwd(_ptr(resp), N, C, Ho, *[_ptr(t) for t in w], int(gn_groups), float(gn_eps),
                                        _ptr(packed), _ptr(tower_ws), _ptr(logits), ln.stream)
    _check(rc, "emm_predictor")
    return logits


_hann_cache = {}


def hann_window(G, device):
    """``torch.hann_window(G)`` (periodic) evaluated on the CPU — bit-identical to what the CPU
    reference multiplies in (track_core.py:157-158) — cached per device."""
    key = (G, device)
    w = _hann_cache.get(key)
    if w is None:
        w = _hann_cache[key] = torch.hann_window(G, dtype=torch.float).to(device)
    return w


def emm_decode(logits, sr, boxes, rx, rz, pad_pixels, sigma=0.4, use_centerness=True, return_index=False,
               clip_wh=None):
    """Fused up-sample + decode.  logits ``[N,7,Ho,Ho]``, sr/boxes ``[N,4]`` → (bb ``[N,4]``, conf ``[N]``).
    ``clip_wh=(W, H)`` also applies the in-place clamp of ``BoxList.clip_to_image`` (non-amodal)."""
    lib = load_library()
    logits = _dev_f32(logits, "logits")
    sr = _dev_f32(sr, "sr")
    boxes = _dev_f32(boxes, "boxes")
    N, _, Ho, _ = logits.shape
    dev = logits.device
    G = Ho * UP_SCALE
    ws = torch.empty((max(N, 1) * lib.smot_emm_decode_ws_floats(Ho, UP_SCALE),), dtype=torch.float32, device=dev)
    bb = torch.empty((N, 4), dtype=torch.float32, device=dev)
    conf = torch.empty((N,), dtype=torch.float32, device=dev)
    idx = torch.empty((N,), dtype=torch.int64, device=dev) if return_index else None
    with _Launch(logits, sr, boxes) as ln:
        rc = lib.smot_emm_decode_fwd(_ptr(logits), _ptr(sr), _ptr(boxes), _ptr(hann_window(G, dev)), N, Ho, UP_SCALE,
                                     int(rx), int(rz), float(pad_pixels), float(1 - sigma), float(sigma),
                                     int(bool(use_centerness)),
                                     float(clip_wh[0]) if clip_wh is not None else 0.0,
                                     float(clip_wh[1]) if clip_wh is not None else 0.0,
                                     _ptr(ws), _ptr(bb), _ptr(conf), _ptr(idx), ln.stream)
    _check(rc, "emm_decode")
    return (bb, conf, idx) if return_index else (bb, conf)


# ----------------------------------------------------------------------------------------------
# one-call halves of a frame pair
# ----------------------------------------------------------------------------------------------
_cast = lambda a: ctypes.cast(a, ctypes.c_void_p)


def _level_arrays(features, scales):
    L = len(scales)
    feats = [_dev_f32(features[l], "features[%d]" % l) for l in range(L)]
    for f in feats:
        if f.dim() != 4 or f.shape[0] != 1:
            raise RuntimeError("siammot_amd: one image per call, got feature shape %s" % (tuple(f.shape),))
    fp = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    hs = (ctypes.c_int * L)(*[f.shape[2] for f in feats])
    ws = (ctypes.c_int * L)(*[f.shape[3] for f in feats])
    sc = (ctypes.c_float * L)(*[float(s) for s in scales])
    return feats, fp, hs, ws, sc


_ws_cache = {}


def _workspace(device, n_floats, stream=None):
    """Grow-only fp32 scratch per (device, stream): intermediates never leave the library and nothing persists
    semantically between calls, but two streams of one device run concurrently and must not share it."""
    key = (device, stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < n_floats:
        buf = torch.empty((int(n_floats * 1.25) + 1024,), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


# ---- launch state reused across frames --------------------------------------------------------
# A frame pair is ~80 us of GPU work, so the host side of the two one-call entry points has to stay well below
# that: everything that does not change from frame to frame (per-level geometry arrays, the validated parameter
# pointer block, the packed tower filters, the Hann window, the workspace) is cached; per call only the feature /
# box pointers are refreshed and checked.
_F32 = torch.float32


class _LevelGeometry(object):
    """ctypes arrays of one FPN geometry: heights / widths / scales / virtual pad cells, plus the pointer array
    that is refilled every call."""
    __slots__ = ("shapes", "L", "C", "fp", "hs", "ws", "sc", "pc", "a_fp", "a_hs", "a_ws", "a_sc", "a_pc", "keep")

    def __init__(self, shapes, scales, pad_pixels):
        L = len(scales)
        self.shapes, self.L, self.C = shapes, L, shapes[0][1]
        self.fp = (ctypes.c_void_p * L)()
        self.hs = (ctypes.c_int * L)(*[sh[2] for sh in shapes])
        self.ws = (ctypes.c_int * L)(*[sh[3] for sh in shapes])
        self.sc = (ctypes.c_float * L)(*[float(v) for v in scales])
        self.pc = (ctypes.c_int * L)(*[int(pad_pixels / ((2 ** i) * 4)) for i in range(L)])
        self.a_fp, self.a_hs, self.a_ws = ctypes.addressof(self.fp), ctypes.addressof(self.hs), ctypes.addressof(self.ws)
        self.a_sc, self.a_pc = ctypes.addressof(self.sc), ctypes.addressof(self.pc)
        self.keep = [None] * L                   # contiguous copies of strided inputs, alive until the next call


_geom_cache = {}


def _geometry(features, scales, pad_pixels, device):
    """Validate the per-level feature tensors (all on ``device``) and return the cached geometry with fresh
    pointers."""
    L = len(scales)
    shapes = tuple(tuple(features[l].shape) for l in range(L))
    key = (shapes, tuple(scales), pad_pixels)
    g = _geom_cache.get(key)
    if g is None:
        for sh in shapes:
            if len(sh) != 4 or sh[0] != 1 or sh[1] != shapes[0][1]:
                raise RuntimeError("siammot_amd: one image per call and one channel count, got feature shapes %s"
                                   % (shapes,))
        if len(_geom_cache) > 32:
            _geom_cache.clear()
        g = _geom_cache[key] = _LevelGeometry(shapes, scales, pad_pixels)
    fp = g.fp
    for l in range(L):
        f = features[l]
        if not (f.is_cuda and f.dtype is _F32 and f.is_contiguous()):
            f = g.keep[l] = _dev_f32(f, "features[%d]" % l)         # raises, or copies a strided view
        if f.device != device:
            raise RuntimeError("siammot_amd: features[%d] lives on %s, the boxes on %s" % (l, f.device, device))
        fp[l] = f.data_ptr()
    return g


def _geometry_refresh(g, features, device):
    """The per-frame part of ``_geometry`` for a caller that keeps ``g``: same shapes, fp32, contiguous, on ``device`` ->
    the pointer array is refilled and True is returned; anything else returns False (the caller takes the full path,
    which raises or rebuilds)."""
    fp, shapes = g.fp, g.shapes
    if len(features) < g.L:
        return False
    for l in range(g.L):                 # validate every level first: a mismatch must leave the pointer array as it was
        f = features[l]
        if f.shape != shapes[l] or not (f.is_cuda and f.dtype is _F32 and f.is_contiguous()) or f.device != device:
            return False
    for l in range(g.L):
        fp[l] = features[l].data_ptr()
    return True


def _same_device(device, *named):
    for name, t in named:
        if t.device != device:
            raise RuntimeError("siammot_amd: %s lives on %s, the boxes on %s" % (name, t.device, device))


class _ParamBlock(object):
    """The 12 predictor tensors validated once, their pointers as a ctypes block, and the Winograd-packed tower
    filters; revalidated per call by data pointer + version counter (a dozen attribute reads)."""
    __slots__ = ("params", "tensors", "stamp", "pp", "a_pp", "packed", "C", "calls")

    def __init__(self, params):
        self.params = params                     # keeps the dict (and so its id) alive while cached
        self.calls = 0
        self.refresh()

    def refresh(self):
        w = [_dev_f32(self.params[k], k) for k in PREDICTOR_KEYS]
        C = w[0].shape[0]
        expect = {0: (C, C, 3, 3), 3: (C, C, 3, 3), 6: (2, C, 3, 3), 8: (1, C, 3, 3), 10: (4, C, 3, 3)}
        for i, shp in expect.items():
            if tuple(w[i].shape) != shp:
                raise RuntimeError("siammot_amd: %s has shape %s, expected %s" % (PREDICTOR_KEYS[i], tuple(w[i].shape), shp))
        self.tensors, self.C = w, C
        self.stamp = [(t.data_ptr(), t._version) for t in w]
        self.packed = tower_packed(self.params)
        self.pp = (ctypes.c_void_p * 13)(*([t.data_ptr() for t in w] + [self.packed.data_ptr() if self.packed is not None else None]))
        self.a_pp = ctypes.addressof(self.pp)

    def current(self):
        """Revalidate against in-place updates (``load_state_dict`` bumps every tensor's version) and moves
        (``.to()`` replaces every storage): the two tower weights are checked on every call, the whole set every
        32nd — a dozen attribute reads less per frame on the tracking loop's host path."""
        p, st = self.params, self.stamp
        self.calls = calls = self.calls + 1
        keys = PREDICTOR_KEYS if (calls & 31) == 0 else _TOWER_KEYS
        for k in keys:
            t = p[k]
            s = st[_KEY_INDEX[k]]
            if t.data_ptr() != s[0] or t._version != s[1]:
                self.refresh()
                break
        return self


_param_cache = {}


def _param_block(params):
    blk = _param_cache.get(id(params))
    if blk is None or blk.params is not params:
        if len(_param_cache) > 16:
            _param_cache.clear()
        blk = _param_cache[id(params)] = _ParamBlock(params)
        return blk
    return blk.current()


def _chk(t, name, shape):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype is _F32):
        _dev_f32(t, name)
    if tuple(t.shape) != shape:
        raise RuntimeError("siammot_amd: %s has shape %s, expected %s" % (name, tuple(t.shape), shape))
    return t if t.is_contiguous() else t.contiguous()


def emm_track(features, boxes, sr, templates, params, rx, rz, scales, sampling_ratio, pad_pixels,
              sigma=0.4, use_centerness=True, clip_wh=None, gn_groups=32, gn_eps=1e-5, return_index=False,
              winograd=True, order_hint=None):
    """The inference branch of ``EMM.forward`` in ONE library call.  Returns (bb ``[N,4]``, conf ``[N]``).

    ``order_hint``: the ``[N, HINT_FLOATS]`` tensor ``emm_extract_cache(..., hint=True)`` returned TOGETHER WITH exactly
    these ``boxes`` / ``sr`` (include/smot_emm.h: a scheduling side channel, VERIFIED by the kernel against these very
    tensors: a hint of other boxes raises its status word — ``order_hint_status`` — and every returned row is NaN;
    ``siammot_amd.emm.EMM`` also checks tensor identity and versions before it passes one), or None."""
    lib = _lib or load_library()
    if not (isinstance(boxes, torch.Tensor) and boxes.is_cuda):
        _dev_f32(boxes, "boxes")                 # raises: no CPU path
    dev = boxes.device
    g = _geometry(features, scales, pad_pixels, dev)
    N, C = boxes.shape[0], g.C
    boxes = _chk(boxes, "boxes", (N, 4))
    sr = _chk(sr, "sr", (N, 4))
    templates = _chk(templates, "template_features", (N, C, rz, rz))
    blk = _param_block(params)
    _same_device(dev, ("sr", sr), ("template_features", templates), ("predictor weights", blk.tensors[0]))
    if blk.C != C:
        raise RuntimeError("siammot_amd.emm_track: predictor has %d channels, features have %d" % (blk.C, C))
    ho = rx - rz + 1
    a_pp = blk.a_pp
    if not (winograd and ho in (16, 29)) and blk.packed is not None:
        pp = (ctypes.c_void_p * 13)(*([t.data_ptr() for t in blk.tensors] + [None]))    # direct tower kernel
        a_pp = ctypes.addressof(pp)
    stream = _stream(dev)
    work = _workspace(dev, lib.smot_emm_track_ws_floats(N, C, rx, rz), stream.value)
    bb = torch.empty((N, 4), dtype=_F32, device=dev)
    conf = torch.empty((N,), dtype=_F32, device=dev)
    idx = torch.empty((N,), dtype=torch.int64, device=dev) if return_index else None
    if order_hint is None and rx == 30 and rz == 15 and 2 <= N <= 256:
        FALLBACKS["unhinted_head"] += 1
    if order_hint is not None and (tuple(order_hint.shape) != (N, HINT_FLOATS) or order_hint.device != dev
                                   or order_hint.dtype is not _F32 or not order_hint.is_contiguous()):
        raise RuntimeError("siammot_amd.emm_track: order_hint must be the contiguous fp32 [%d, %d] tensor of the "
                           "extraction that made these boxes" % (N, HINT_FLOATS))
    cur = torch.cuda.current_device()           # kernels launch on the CURRENT device: make it the tensors' device
    if cur != dev.index:
        torch.cuda.set_device(dev.index)
    try:
        rc = lib.smot_emm_track_fwd(g.a_fp, g.a_hs, g.a_ws, g.a_pc, g.a_sc, g.L, C,
                                boxes.data_ptr(), sr.data_ptr(), templates.data_ptr(), N, rx, rz, sampling_ratio,
                                a_pp, gn_groups, gn_eps, hann_window(ho * UP_SCALE, dev).data_ptr(),
                                UP_SCALE, pad_pixels, 1 - sigma, sigma, 1 if use_centerness else 0,
                                float(clip_wh[0]) if clip_wh is not None else 0.0,
                                float(clip_wh[1]) if clip_wh is not None else 0.0,
                                work.data_ptr(), bb.data_ptr(), conf.data_ptr(),
                                idx.data_ptr() if idx is not None else None,
                                order_hint.data_ptr() if order_hint is not None else None, stream)
    finally:
        if cur != dev.index:
            torch.cuda.set_device(cur)
    if rc:
        _check(rc, "emm_track")
    return (bb, conf, idx) if return_index else (bb, conf)


HINT_FLOATS = 536                    # SMOT_HINT_FLOATS (include/smot_emm.h): entry header + both finished sample tables + by-roi record
HINT_STATUS_WORD = 534               # SMOT_HINT_STATUS_WORD: the list's status word (entry 0), raised by a head that found the
                                     # hint NOT describing its rois (that head's rows are NaN)
_hint_floats = {}


def order_hint_status(order_hint):
    """Status word of an order hint (``[N, HINT_FLOATS]`` tensor): 0 = every head given this hint found it describing its
    rois.  Synchronises (one 4-byte copy): for error paths and tests, not for the frame loop."""
    return int(order_hint.view(-1)[HINT_STATUS_WORD:HINT_STATUS_WORD + 1].view(torch.int32).item())


def order_hint_floats(N, rz, sampling_ratio):
    """``smot_emm_order_hint_floats``: size of the order hint the extraction writes for ``N`` boxes (0 = none)."""
    key = (N, rz, sampling_ratio)
    v = _hint_floats.get(key)
    if v is None:
        if len(_hint_floats) > 1024:
            _hint_floats.clear()
        v = _hint_floats[key] = int((_lib or load_library()).smot_emm_order_hint_floats(N, rz, sampling_ratio))
    return v


def emm_extract_cache(features, boxes, rz, scales, sampling_ratio, pad_pixels, search_expansion, min_search_wh,
                      n_valid=None, hint=False):
    """``EMM.extract_cache`` in one library call → (templates ``[N,C,rz,rz]``, sr ``[N,4]``); with ``hint=True`` →
    (templates, sr, order hint ``[N, HINT_FLOATS]`` or None when this shape / count writes none): the list the next
    frame's ``emm_track(..., order_hint=)`` reads instead of ranking these search regions in every workgroup.

    ``n_valid``: a device int32 tensor (1 element) holding the number of REAL rows among ``boxes`` (a capacity): rows
    beyond it are skipped on the device and their outputs stay unwritten — the call can be enqueued before the
    host knows the count (ops.track_solve_launch)."""
    lib = _lib or load_library()
    if not (isinstance(boxes, torch.Tensor) and boxes.is_cuda):
        _dev_f32(boxes, "boxes")                 # raises: no CPU path
    dev = boxes.device
    g = _geometry(features, scales, 0, dev)
    N, C = boxes.shape[0], g.C
    boxes = _chk(boxes, "boxes", (N, 4))
    templates = torch.empty((N, C, rz, rz), dtype=_F32, device=dev)
    oh = None
    if hint and boxes.data_ptr() % 16 == 0 and order_hint_floats(N, rz, sampling_ratio) > 0:
        both = torch.empty((N * (HINT_FLOATS + 4),), dtype=_F32, device=dev)       # one allocation: hint | sr
        oh = both[:N * HINT_FLOATS].view(N, HINT_FLOATS)
        sr = both[N * HINT_FLOATS:].view(N, 4)
    else:
        sr = torch.empty((N, 4), dtype=_F32, device=dev)
    cur = torch.cuda.current_device()
    if cur != dev.index:
        torch.cuda.set_device(dev.index)
    try:
        if n_valid is None:
            rc = lib.smot_emm_extract_cache_fwd(g.a_fp, g.a_hs, g.a_ws, g.a_sc, g.L, C, boxes.data_ptr(), N,
                                                rz, sampling_ratio, pad_pixels, search_expansion, min_search_wh,
                                                templates.data_ptr(), sr.data_ptr(),
                                                oh.data_ptr() if oh is not None else None, _stream(dev))
        else:
            rc = lib.smot_emm_extract_cache_masked_fwd(g.a_fp, g.a_hs, g.a_ws, g.a_sc, g.L, C, boxes.data_ptr(), N,
                                                       n_valid.data_ptr(), rz, sampling_ratio, pad_pixels,
                                                       search_expansion, min_search_wh, templates.data_ptr(),
                                                       sr.data_ptr(), oh.data_ptr() if oh is not None else None,
                                                       _stream(dev))
    finally:
        if cur != dev.index:
            torch.cuda.set_device(cur)
    if rc:
        _check(rc, "emm_extract_cache")
    return (templates, sr, oh) if hint else (templates, sr)


try:
    _cur_device = torch._C._cuda_getDevice                    # current device index, ~0.1 us (torch.cuda.current_device: ~0.7)
except AttributeError:                                        # pragma: no cover - older torch
    _cur_device = torch.cuda.current_device


class PairPlan(object):
    """The host side of ``EMM.forward`` / ``EMM.extract_cache`` for a caller that makes the same two calls every frame
    (round 6): what ``emm_track`` / ``emm_extract_cache`` look up, validate and convert on every call —
    the level geometry, the parameter block, the workspace, the Hann window, the hint's size, two dozen scalar arguments —
    is resolved ONCE per (feature shapes, scales, device, shape family) and kept as the constant part of the library
    call's argument tuple; a call then checks what can change between frames (the five feature tensors: shape, dtype,
    layout, device -> their pointers; the box / search-region / template tensors; the parameter tensors' storage and
    version), allocates its outputs and makes the ONE library call.  ``track`` / ``extract`` return None for anything they
    do not recognise (another geometry, a strided tensor, another device, a count the plan was not made for is re-planned):
    the caller then takes the general functions above, which raise or convert.  Results are the general functions', bit
    for bit — same library entry points, same arguments.  There is still no CPU or eager path."""
    __slots__ = ("dev", "dev_index", "g", "gz", "params", "blk", "rx", "rz", "ho", "scales", "sampling_ratio", "pad_pixels",
                 "hann_ptr", "ws", "ws_n", "C", "hint_n", "hint_ok", "lib", "f_track", "f_extract", "tu")

    def __init__(self, features, dev, params, rx, rz, scales, sampling_ratio, pad_pixels, tu):
        self.lib = lib = _lib or load_library()
        self.dev, self.dev_index = dev, dev.index
        self.g = _geometry(features, scales, pad_pixels, dev)           # full validation (raises)
        self.gz = _geometry(features, scales, 0, dev)
        self.params, self.blk = params, _param_block(params)
        self.rx, self.rz, self.ho = rx, rz, rx - rz + 1
        self.scales, self.sampling_ratio, self.pad_pixels = scales, sampling_ratio, pad_pixels
        self.C = self.g.C
        if self.blk.C != self.C:
            raise RuntimeError("siammot_amd.emm_track: predictor has %d channels, features have %d" % (self.blk.C, self.C))
        self.hann_ptr = hann_window(self.ho * UP_SCALE, dev).data_ptr()
        self.ws, self.ws_n = None, -1
        self.hint_n, self.hint_ok = -1, False
        self.f_track, self.f_extract = lib.smot_emm_track_fwd, lib.smot_emm_extract_cache_fwd
        self.tu = (float(tu.pad_pixels), float(tu.search_expansion), float(tu.min_search_wh))

    def stale(self, params, rx, rz, scales, sampling_ratio, pad_pixels, tu):
        return (params is not self.params or rx != self.rx or rz != self.rz or scales != self.scales or
                sampling_ratio != self.sampling_ratio or pad_pixels != self.pad_pixels or self.lib is not _lib or
                self.tu != (float(tu.pad_pixels), float(tu.search_expansion), float(tu.min_search_wh)))

    def track(self, features, boxes, sr, templates, sigma, use_centerness, clip_w, clip_h, gn_groups, gn_eps, order_hint):
        """``emm_track`` (winograd path, no index output): (bb ``[N,4]``, conf ``[N]``) or None."""
        dev, g, rz = self.dev, self.g, self.rz
        if not (boxes.is_cuda and boxes.dtype is _F32 and boxes.is_contiguous() and boxes.device == dev and boxes.dim() == 2):
            return None
        N = boxes.shape[0]
        if not (sr.is_cuda and sr.dtype is _F32 and sr.is_contiguous() and sr.shape == boxes.shape and boxes.shape[1] == 4 and
                templates.is_cuda and templates.dtype is _F32 and templates.is_contiguous() and
                templates.shape == (N, self.C, rz, rz) and sr.get_device() == self.dev_index and
                templates.get_device() == self.dev_index and _cur_device() == self.dev_index):
            return None
        if not _geometry_refresh(g, features, dev):
            return None
        blk = self.blk.current()
        if blk.packed is None or blk.tensors[0].get_device() != self.dev_index or blk.C != self.C:
            return None
        stream = _raw_stream(self.dev_index)
        if N != self.ws_n or self.ws is None or self.ws is not _ws_cache.get((dev, stream)):
            self.ws = _workspace(dev, self.lib.smot_emm_track_ws_floats(N, self.C, self.rx, rz), stream)
            self.ws_n = N
        bb = torch.empty((N, 4), dtype=_F32, device=dev)
        conf = torch.empty((N,), dtype=_F32, device=dev)
        hint_ptr = None
        if order_hint is not None:
            if (order_hint.shape != (N, HINT_FLOATS) or order_hint.get_device() != self.dev_index or
                    order_hint.dtype is not _F32 or not order_hint.is_contiguous()):
                return None
            hint_ptr = order_hint.data_ptr()
        elif self.rx == 30 and rz == 15 and 2 <= N <= 256:
            FALLBACKS["unhinted_head"] += 1
        rc = self.f_track(g.a_fp, g.a_hs, g.a_ws, g.a_pc, g.a_sc, g.L, self.C, boxes.data_ptr(), sr.data_ptr(),
                          templates.data_ptr(), N, self.rx, rz, self.sampling_ratio, blk.a_pp, gn_groups, gn_eps,
                          self.hann_ptr, UP_SCALE, self.pad_pixels, 1 - sigma, sigma, 1 if use_centerness else 0, clip_w,
                          clip_h, self.ws.data_ptr(), bb.data_ptr(), conf.data_ptr(), None, hint_ptr, stream)
        if rc:
            _check(rc, "emm_track")
        return bb, conf

    def extract(self, features, boxes, hint):
        """``emm_extract_cache`` (un-masked form): (templates, sr, order hint or None) or None."""
        dev, g, rz = self.dev, self.gz, self.rz
        if not (boxes.is_cuda and boxes.dtype is _F32 and boxes.is_contiguous() and boxes.device == dev and
                boxes.dim() == 2 and boxes.shape[1] == 4 and _cur_device() == self.dev_index):
            return None
        if not _geometry_refresh(g, features, dev):
            return None
        N = boxes.shape[0]
        templates = torch.empty((N, self.C, rz, rz), dtype=_F32, device=dev)
        oh = None
        if hint:
            if N != self.hint_n:
                self.hint_n, self.hint_ok = N, order_hint_floats(N, rz, self.sampling_ratio) > 0
            if self.hint_ok and boxes.data_ptr() % 16 == 0:
                both = torch.empty((N * (HINT_FLOATS + 4),), dtype=_F32, device=dev)       # one allocation: hint | sr
                oh = both[:N * HINT_FLOATS].view(N, HINT_FLOATS)
                sr = both[N * HINT_FLOATS:].view(N, 4)
        if oh is None:
            sr = torch.empty((N, 4), dtype=_F32, device=dev)
        tu = self.tu
        rc = self.f_extract(g.a_fp, g.a_hs, g.a_ws, g.a_sc, g.L, self.C, boxes.data_ptr(), N, rz, self.sampling_ratio,
                            tu[0], tu[1], tu[2], templates.data_ptr(), sr.data_ptr(),
                            oh.data_ptr() if oh is not None else None, _raw_stream(self.dev_index))
        if rc:
            _check(rc, "emm_extract_cache")
        return templates, sr, oh


TIMER_XCORR, TIMER_TOWER = 0, 1


# Capacity cliffs of the fast paths fall back to slower (correct) ones: every such event is counted here so that a
# benchmark / a service can REPORT them instead of silently running slower.  Keys:
#   refine_library_gemm  box-head refinement of more rows than the weight-streaming kernels take (library GEMMs instead)
#   host_solver          a frame beyond the one-launch solver's capacity (or with fields it does not carry): host path
#   unhinted_head        a pooling + correlation launch that ranked its rois itself although an order hint could exist
#   general_frame        a tracking-loop frame that did not take the one-launch path at all
import collections as _collections
FALLBACKS = _collections.Counter()
# speculative next-frame heads of the tracking loop (TrackingLoop.forward(..., next_features=...)): launched / used as they
# were / discarded because the row count, the memory, the features or the parameters were not what the launch assumed
SPECULATION = _collections.Counter()


TOWER_FORMS = {0: "direct fp32 (no packed path)", 1: "Winograd, one 16-channel tile per workgroup, fp32 matrix instructions",
               2: "Winograd, two tiles per workgroup, fp32 matrix instructions",
               3: "Winograd, two tiles per workgroup, two-part fp16 operands on v_mfma_f32_16x16x32_f16"}


def tower_form(n, channels, ho=16):
    """Which form of the tower kernel ``n`` tracks get from the current library (``smot_emm_tower_form``): 0..3, see
    ``TOWER_FORMS``."""
    return int(load_library().smot_emm_tower_form(int(n), int(channels), int(ho)))


def fused_kernel_name():
    """Name of the kernel ``smot_emm_track_fwd`` runs for search-region pooling + cross-correlation at the
    DLA shape family (what bench.py's roofline and the rocprofv3 summaries in profiles/ refer to)."""
    return "sr_xcorr_fused9_kernel<30,15,2,true,8,false,1>"       # (..., planes per workgroup, no plane-pair FMA phase, matrix-pipe correlation)


def kernel_timer_begin(slot, max_launches, stride=1):
    """Start bracketing the kernels of ``slot`` with HIP events on their launch stream (bench.py)."""
    _check(load_library().smot_kernel_timer_begin(int(slot), int(max_launches), int(stride)), "kernel_timer_begin")


def kernel_timer_bracket_overhead(reps=200):
    """Median span (us) of an empty event bracket on the current stream."""
    us = ctypes.c_double(0.0)
    _check((_lib or load_library()).smot_kernel_timer_bracket_overhead(_stream(), int(reps), ctypes.byref(us)),
           "kernel_timer_bracket_overhead")
    return us.value


def dispatch_floor_us(workgroups, threads, reps=200):
    """Average span of the kernel timer's event bracket around an EMPTY kernel of this launch shape on the current stream
    (``smot_dispatch_floor_fwd``; timer slot 0 must be idle): the bracket's own floor (4.1 us on MI355X), an upper bound on
    the fixed cost of a dispatch — rocprofv3 reads 0.8-1.5 us for the same kernel from the packet's timestamps."""
    lib = _lib or load_library()
    for _ in range(20):
        _check(lib.smot_dispatch_floor_fwd(int(workgroups), int(threads), _stream()), "dispatch_floor")
    kernel_timer_begin(TIMER_XCORR, reps, 1)
    for _ in range(reps):
        _check(lib.smot_dispatch_floor_fwd(int(workgroups), int(threads), _stream()), "dispatch_floor")
    torch.cuda.synchronize()
    ms, cnt = kernel_timer_end(TIMER_XCORR)
    return ms * 1e3 / max(cnt, 1)


def kernel_timer_end(slot):
    """→ (total milliseconds inside the slot's kernels, launches timed)."""
    tot = ctypes.c_double(0.0)
    n = ctypes.c_int(0)
    _check(load_library().smot_kernel_timer_end(int(slot), ctypes.cast(ctypes.byref(tot), ctypes.c_void_p),
                                                ctypes.cast(ctypes.byref(n), ctypes.c_void_p)), "kernel_timer_end")
    return tot.value, n.value


def xcorr_timer_begin(max_launches):
    kernel_timer_begin(TIMER_XCORR, max_launches)


def xcorr_timer_end():
    return kernel_timer_end(TIMER_XCORR)


def sr_xcorr_fused(features, boxes, sr, templates, rx, rz, scales, sampling_ratio, pad_pixels, return_pooled=False):
    """Search-region pooling + depthwise cross-correlation in one kernel → response ``[N,C,Ho,Ho]``
    (and the pooled ``[N,C,rx,rx]`` planes when asked — test hook)."""
    lib = load_library()
    boxes = _dev_f32(boxes, "boxes")
    sr = _dev_f32(sr, "sr")
    templates = _dev_f32(templates, "template_features")
    N = boxes.shape[0]
    feats, fp, hs, ws_, sc = _level_arrays(features, scales)
    C = feats[0].shape[1]
    L = len(scales)
    pc = (ctypes.c_int * L)(*[int(pad_pixels / ((2 ** i) * 4)) for i in range(L)])
    ho = rx - rz + 1
    resp = torch.empty((N, C, ho, ho), dtype=torch.float32, device=boxes.device)
    pooled = torch.empty((N, C, rx, rx), dtype=torch.float32, device=boxes.device) if return_pooled else None
    with _Launch(boxes, sr, templates, *feats) as ln:
        if (int(rx), int(rz)) == (35, 7) and pooled is None:        # the second yaml family's shape: its own entry
            rc = lib.smot_sr_xcorr_gather_fwd(_cast(fp), _cast(hs), _cast(ws_), _cast(pc), _cast(sc), L, C, _ptr(boxes),
                                              _ptr(sr), _ptr(templates), N, int(rx), int(rz), int(sampling_ratio),
                                              _ptr(resp), ln.stream)
        else:
            rc = lib.smot_sr_xcorr_fused_fwd(_cast(fp), _cast(hs), _cast(ws_), _cast(pc), _cast(sc), L, C, _ptr(boxes),
                                             _ptr(sr), _ptr(templates), N, int(rx), int(rz), int(sampling_ratio),
                                             _ptr(resp), _ptr(pooled), ln.stream)
    _check(rc, "sr_xcorr_fused")
    return (resp, pooled) if return_pooled else resp


def nms_keep_mask(boxes, scores, thresh):
    """Greedy NMS as a device-resident boolean mask in the ORIGINAL box order (True = kept) — no host
    synchronisation; ``nms`` below turns it into upstream's index list."""
    lib = load_library()
    boxes = _dev_f32(boxes, "boxes")
    scores = _dev_f32(scores, "scores")
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.bool, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    sorted_boxes = boxes[order].contiguous()
    ws = torch.empty((max(lib.smot_nms_ws_bytes(n) // 8, 1),), dtype=torch.int64, device=boxes.device)
    keep = torch.empty((n,), dtype=torch.uint8, device=boxes.device)
    with _Launch(boxes, scores) as ln:
        rc = lib.smot_nms_fwd(_ptr(sorted_boxes), n, float(thresh), _ptr(ws), _ptr(keep), ln.stream)
    _check(rc, "nms")
    mask = torch.empty((n,), dtype=torch.bool, device=boxes.device)
    mask[order] = keep.bool()
    return mask


def nms(boxes, scores, thresh):
    """[UPSTREAM] ``_C.nms(dets, scores, thresh)`` semantics: indices of the kept boxes, ascending (original
    order), after greedy suppression in descending-score order with the +1 IoU convention.  One host sync
    (the number of kept boxes), as in the reference."""
    lib = load_library()
    boxes = _dev_f32(boxes, "boxes")
    scores = _dev_f32(scores, "scores")
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    sorted_boxes = boxes[order].contiguous()
    ws = torch.empty((max(lib.smot_nms_ws_bytes(n) // 8, 1),), dtype=torch.int64, device=boxes.device)
    keep = torch.empty((n,), dtype=torch.uint8, device=boxes.device)
    with _Launch(boxes, scores) as ln:
        rc = lib.smot_nms_fwd(_ptr(sorted_boxes), n, float(thresh), _ptr(ws), _ptr(keep), ln.stream)
    _check(rc, "nms")
    return order[keep.bool()].sort()[0]


_rec_pinned = {}


class HostRecordRing(object):
    """Two pinned, device-accessible int32 buffers large enough for any solver record, and the event recorded behind
    each launch — owned by ONE tracker (``TrackPool.host_record_ring``), so two tracking loops driven from different
    threads or streams never re-zero or re-record each other's completion word.  The buffers alternate: the host has
    copied the previous frame's record out before the next launch can overwrite it.  A frame that was aborted between
    launch and wait (an exception, KeyboardInterrupt) leaves its buffer marked in flight; the stream is drained
    before that buffer is handed out again, so a late store of the old kernel cannot complete a newer record."""

    def __init__(self, dev, pool_capacity):
        n = 8 + 4 * track_solve_max_boxes() + 3 * pool_capacity
        self.dev = dev
        self.bufs = [torch.zeros((n,), dtype=torch.int32).pin_memory() for _ in range(2)]
        self.views = [b.numpy() for b in self.bufs]          # the same memory as numpy arrays (cheap element access)
        self.in_flight = [False, False]
        self.k = 0
        self.event = torch.cuda.Event()

    def next(self):
        self.k ^= 1
        if self.in_flight[self.k]:
            torch.cuda.current_stream(self.dev).synchronize()
            self.in_flight = [False, False]
        rec = self.bufs[self.k]
        self.views[self.k][3] = 0     # the kernel stores the frame index (>= 1) here last: the host's completion flag
        self.in_flight[self.k] = True
        return rec

    def view(self, rec):
        """The numpy view of one of this ring's buffers."""
        return self.views[0] if rec is self.bufs[0] else self.views[1]

    def record_event(self):
        """Record this ring's event on the device's current stream (behind the solver launch)."""
        self.event.record(torch.cuda.current_stream(self.dev))
        return self.event

    def wait(self, rec, event=True):
        """The frame's one synchronisation: poll the completion word of ``rec``; if it does not show up, fall back to the
        event recorded behind the launch (``record_event``) or — ``event=False``, nothing was recorded — to draining the
        stream (a kernel fault would otherwise spin for ever)."""
        i = 0 if rec is self.bufs[0] else 1
        flag = self.views[i]
        for _ in range(20000):
            if flag[3] != 0:
                break
        else:
            if event:
                self.event.synchronize()
            else:
                torch.cuda.current_stream(self.dev).synchronize()
            if flag[3] == 0:
                raise RuntimeError("siammot_amd.track_solve: the solver kernel finished without completing its record")
        self.in_flight[i] = False


def track_solve(det, trk, trk_score_bias, thresholds, nms_thresh, max_dormant_frames, pool_state, pool_capacity,
                host_record=False, carry=None):
    """``smot_track_solve_fwd``: one launch for TrackSolver.forward + the pool transitions + the active-row filter.

    det / trk: ``(boxes [n,4] xyxy, scores [n], ids [n] int64, labels [n] int64 or None)`` device tensors or ``None``
    for an empty segment.  Launch only (no synchronisation).  Returns ``(fbuf, ibuf, rec, M)``: ``fbuf`` fp32
    ``[10*M]`` = out_boxes | act_boxes | out_scores | act_scores, ``ibuf`` int64 ``[4*M]`` = out_ids | out_labels |
    act_ids | act_labels (capacity M rows each; the first K / A are valid) and ``rec``, the record on the DEVICE
    (``rec[0]`` = K, ``rec[1]`` = A, ...: include/smot_emm.h); ``track_solve_record(rec)`` brings it to the host.
    ``host_record=ring`` (a ``HostRecordRing``): the kernel writes the record straight into the ring's next pinned
    host buffer (``rec`` is then that host tensor; ``ring.record_event()`` behind this launch, ``ring.wait(rec)`` to
    read it — no copy command; the count of active rows for device-side consumers is ``pool_state[4:5]``).
    ``carry`` = (source addresses (templates, boxes, search regions, ids, labels, scores), first source row, rows, guessed
    first destination row, next_templates address, next_sr address, row_floats): ``smot_track_solve_carry_fwd`` — the
    dormant rows of the track memory go behind the active rows in this launch."""
    lib = _lib or load_library()
    segs = []
    dev = pool_state.device
    for seg in (det, trk):
        if seg is None or seg[0].shape[0] == 0:
            segs.append((0, 0, 0, 0, 0))
            continue
        b, s_, i_, l_ = seg
        _check_segment(b, s_, i_, l_, dev)
        segs.append((b.data_ptr(), s_.data_ptr(), i_.data_ptr(), l_.data_ptr() if l_ is not None else 0, b.shape[0]))
    M = segs[0][4] + segs[1][4]
    nrec = 8 + 4 * M + 3 * pool_capacity
    fbuf = torch.empty((10 * max(M, 1),), dtype=_F32, device=dev)
    ibuf = torch.empty((4 * max(M, 1),), dtype=torch.int64, device=dev)
    rec = host_record.next() if host_record else torch.empty((nrec,), dtype=torch.int32, device=dev)
    fp, ip = fbuf.data_ptr(), ibuf.data_ptr()
    cur = torch.cuda.current_device()
    if cur != dev.index:
        torch.cuda.set_device(dev.index)
    try:
        if carry is None:
            rc = lib.smot_track_solve_fwd(segs[0][0], segs[0][1], segs[0][2], segs[0][3], segs[0][4],
                                          segs[1][0], segs[1][1], segs[1][2], segs[1][3], segs[1][4], trk_score_bias,
                                          thresholds[0], thresholds[1], thresholds[2], nms_thresh, max_dormant_frames,
                                          pool_state.data_ptr(), pool_capacity,
                                          fp, fp + 32 * M, ip, ip + 8 * M, fp + 16 * M, ip + 16 * M, ip + 24 * M, fp + 36 * M,
                                          rec.data_ptr(), _stream(dev))
        else:
            src, row0, rows, dst0, nz, nsr, row_floats = carry
            rc = lib.smot_track_solve_carry_fwd(segs[0][0], segs[0][1], segs[0][2], segs[0][3], segs[0][4],
                                                segs[1][0], segs[1][1], segs[1][2], segs[1][3], segs[1][4], trk_score_bias,
                                                thresholds[0], thresholds[1], thresholds[2], nms_thresh, max_dormant_frames,
                                                pool_state.data_ptr(), pool_capacity,
                                                fp, fp + 32 * M, ip, ip + 8 * M, fp + 16 * M, ip + 16 * M, ip + 24 * M,
                                                fp + 36 * M, rec.data_ptr(), src[0], src[1], src[2], src[3], src[4], src[5],
                                                int(row0), int(rows), int(dst0), nz, nsr, int(row_floats), _stream(dev))
    finally:
        if cur != dev.index:
            torch.cuda.set_device(cur)
    if rc:
        _check(rc, "track_solve")
    return fbuf, ibuf, rec, M


def track_solve_record(rec):
    """The solver's record on the host: a copy through a pinned buffer + ONE stream synchronisation (the frame's
    only one).  Everything enqueued on the stream before this call — including launches that consume the
    record's device copy, such as a masked ``emm_extract_cache`` — has completed when it returns."""
    dev, nrec = rec.device, rec.shape[0]
    host = _rec_pinned.get((dev, nrec))
    if host is None:
        if len(_rec_pinned) > 64:
            _rec_pinned.clear()
        host = _rec_pinned[(dev, nrec)] = torch.empty((nrec,), dtype=torch.int32).pin_memory()
    host.copy_(rec, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    return host.numpy().copy()


_rec_events = {}


def track_solve_record_begin(rec):
    """Enqueue the record's copy to a pinned buffer and an event behind it; ``track_solve_record_wait`` blocks on that
    event only.  Launches enqueued in between (the masked template extraction) run while the host is already woken
    up and doing its bookkeeping — they are off the frame's critical path."""
    dev, nrec = rec.device, rec.shape[0]
    host = _rec_pinned.get((dev, nrec))
    if host is None:
        if len(_rec_pinned) > 64:
            _rec_pinned.clear()
        host = _rec_pinned[(dev, nrec)] = torch.empty((nrec,), dtype=torch.int32).pin_memory()
    ev = _rec_events.get(dev)
    if ev is None:
        ev = _rec_events[dev] = torch.cuda.Event()
    host.copy_(rec, non_blocking=True)
    ev.record(torch.cuda.current_stream(dev))
    return host, ev


def track_solve_record_wait(handle):
    host, ev = handle
    ev.synchronize()
    return host.numpy().copy()


def wait_host_record(rec, event, spins=20000):
    """Block until the solver kernel has completed a pinned-memory record: poll its completion word (stored last,
    behind a system-scope fence), falling back to the event behind the launch if it does not show up (a kernel
    fault would otherwise spin for ever).  Polling sees the record ~10 us earlier than an event wake-up."""
    flag = rec.numpy()
    for _ in range(spins):
        if flag[3] != 0:
            return
    event.synchronize()
    if flag[3] == 0:
        raise RuntimeError("siammot_amd.track_solve: the solver kernel finished without completing its record")


def box_refine_post(head_out, num_classes, reg_classes, boxes, labels, ids, track_conf, weights, xform_clip, clip_wh,
                    tracktor=False):
    """``smot_box_refine_post_fwd``: the box head's post-processing of N propagated tracks + the score average of
    ``_refine_tracks`` in one launch (no synchronisation).  ``head_out`` ``[N, K + 4*KR]`` = class logits | box deltas
    (one GEMM over the concatenated ``cls_score`` / ``bbox_pred`` weights).  Returns ``(boxes [N,4], scores [N], ids [N],
    labels [N])`` in the box head's output order."""
    lib = _lib or load_library()
    head_out = _dev_f32(head_out, "head_out")
    boxes = _dev_f32(boxes, "boxes")
    track_conf = _dev_f32(track_conf, "track_conf")
    N = boxes.shape[0]
    dev = boxes.device
    for name, t in (("labels", labels), ("ids", ids)):
        if not (t.is_cuda and t.dtype is torch.int64 and t.is_contiguous() and t.device == dev and t.shape[0] == N):
            raise RuntimeError("siammot_amd.box_refine_post: %s must be a contiguous int64 [N] tensor on the boxes' device" % name)
    if head_out.dim() != 2 or head_out.shape[0] != N or head_out.shape[1] < num_classes + 4 * reg_classes:
        raise RuntimeError("siammot_amd.box_refine_post: head_out %s does not hold %d logits + %d deltas for %d rows"
                           % (tuple(head_out.shape), num_classes, 4 * reg_classes, N))
    out_boxes = torch.empty((N, 4), dtype=_F32, device=dev)
    out_scores = torch.empty((N,), dtype=_F32, device=dev)
    out_ids = torch.empty((N,), dtype=torch.int64, device=dev)
    out_labels = torch.empty((N,), dtype=torch.int64, device=dev)
    cw, ch = (0.0, 0.0) if clip_wh is None else (float(clip_wh[0]), float(clip_wh[1]))
    with _Launch(head_out, boxes, track_conf, labels, ids) as ln:
        rc = lib.smot_box_refine_post_fwd(_ptr(head_out), head_out.shape[1], int(num_classes), int(reg_classes), _ptr(boxes),
                                          _ptr(labels), _ptr(ids), _ptr(track_conf), N, float(weights[0]),
                                          float(weights[1]), float(weights[2]), float(weights[3]), float(xform_clip), cw, ch,
                                          int(bool(tracktor)), _ptr(out_boxes), _ptr(out_scores), _ptr(out_ids),
                                          _ptr(out_labels), ln.stream)
    _check(rc, "box_refine_post")
    return out_boxes, out_scores, out_ids, out_labels


def box_refine(features, scales, pooled, sampling_ratio, boxes, labels, ids, track_conf, layers, weights, xform_clip,
               clip_wh, tracktor=False):
    """``smot_box_refine_fwd``: 7x7 pooler -> fc6 -> fc7 -> cls_score | bbox_pred -> post-processing of N <= 64
    propagated tracks in ONE call (six launches, no synchronisation).  ``layers`` = (fc6.weight, fc6.bias, fc7.weight,
    fc7.bias, cls_score.weight, cls_score.bias, bbox_pred.weight, bbox_pred.bias).  Returns ``(boxes, scores, ids,
    labels)`` in the box head's output order."""
    lib = _lib or load_library()
    L = len(scales)
    feats = [_dev_f32(features[l], "features[%d]" % l) for l in range(L)]
    boxes = _dev_f32(boxes, "boxes")
    track_conf = _dev_f32(track_conf, "track_conf")
    w6, b6, w7, b7, wc, bc, wr, br = [_dev_f32(t, "box head parameter") for t in layers]
    N, C, dev = boxes.shape[0], feats[0].shape[1], boxes.device
    K, KR = wc.shape[0], wr.shape[0] // 4
    for name, t in (("labels", labels), ("ids", ids)):
        if not (t.is_cuda and t.dtype is torch.int64 and t.is_contiguous() and t.device == dev and t.shape[0] == N):
            raise RuntimeError("siammot_amd.box_refine: %s must be a contiguous int64 [N] tensor on the boxes' device" % name)
    if w6.shape[1] != C * pooled * pooled or w7.shape[1] != w6.shape[0] or wc.shape[1] != w7.shape[0] or wr.shape[1] != w7.shape[0]:
        raise RuntimeError("siammot_amd.box_refine: layer shapes do not chain")
    need = int(lib.smot_box_refine_ws_floats(N, C, int(pooled), w6.shape[0], w7.shape[0], K, KR))
    ws = _workspace(dev, need, ("box_refine", _stream(dev).value))        # per (device, stream): streams run concurrently
    out_boxes = torch.empty((N, 4), dtype=_F32, device=dev)
    out_scores = torch.empty((N,), dtype=_F32, device=dev)
    out_ids = torch.empty((N,), dtype=torch.int64, device=dev)
    out_labels = torch.empty((N,), dtype=torch.int64, device=dev)
    fp = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    hs = (ctypes.c_int * L)(*[f.shape[2] for f in feats])
    wsz = (ctypes.c_int * L)(*[f.shape[3] for f in feats])
    sc = (ctypes.c_float * L)(*[float(s_) for s_ in scales])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    cw, ch = (0.0, 0.0) if clip_wh is None else (float(clip_wh[0]), float(clip_wh[1]))
    with _Launch(boxes, track_conf, labels, ids, w6, *feats) as ln:
        rc = lib.smot_box_refine_fwd(cast(fp), cast(hs), cast(wsz), cast(sc), L, C, int(pooled), int(sampling_ratio),
                                     _ptr(boxes), _ptr(labels), _ptr(ids), _ptr(track_conf), N,
                                     _ptr(w6), _ptr(b6), w6.shape[0], _ptr(w7), _ptr(b7), w7.shape[0],
                                     _ptr(wc), _ptr(bc), K, _ptr(wr), _ptr(br), KR,
                                     float(weights[0]), float(weights[1]), float(weights[2]), float(weights[3]),
                                     float(xform_clip), cw, ch, int(bool(tracktor)), _ptr(ws), _ptr(out_boxes),
                                     _ptr(out_scores), _ptr(out_ids), _ptr(out_labels), ln.stream)
    _check(rc, "box_refine")
    return out_boxes, out_scores, out_ids, out_labels


def linear_rows(x, weight, bias=None, relu=False, out=None):
    """``smot_linear_rows_fwd``: ``act(x @ weight.T + bias)`` for x ``[M, K]`` with M <= ``linear_rows_max_rows()`` rows
    (the box head's layers on the propagated tracks: weight streaming on all CUs instead of the library's 128
    workgroups at M = 30).  ``out``: an ``[M, >=N]`` row-major view to write into (column block).  Launch only."""
    lib = _lib or load_library()
    x = _dev_f32(x, "x")
    weight = _dev_f32(weight, "weight")
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise RuntimeError("siammot_amd.linear_rows: x is [%d,%d], weight [%d,%d]" % (M, K, N, weight.shape[1]))
    if bias is not None:
        bias = _dev_f32(bias, "bias")
    y = out if out is not None else torch.empty((M, N), dtype=_F32, device=x.device)
    if y.stride(1) != 1 or y.shape[0] != M or y.shape[1] < N:
        raise RuntimeError("siammot_amd.linear_rows: out must be a row-major [M, >=N] view")
    need = int(lib.smot_linear_rows_ws_floats(M, K, N))
    ws = _workspace(x.device, need, ("linear_rows", _stream(x.device).value))
    with _Launch(x, weight, y) as ln:
        rc = lib.smot_linear_rows_fwd(_ptr(x), M, K, _ptr(weight), _ptr(bias), N, int(bool(relu)), _ptr(ws), _ptr(y),
                                      y.stride(0), ln.stream)
    _check(rc, "linear_rows")
    return y


def linear_rows_max_rows():
    return (_lib or load_library()).smot_linear_rows_max_rows()


def box_refine_post_max_rows():
    return (_lib or load_library()).smot_box_refine_post_max_rows()


_FRAME_PTRS = ("feats", "heights", "widths", "pad_cells", "scales", "predictor_params", "hann",
               "fc6_w", "fc6_b", "fc7_w", "fc7_b", "cls_w", "cls_b", "reg_w", "reg_b", "pool_state",
               # this frame: the head
               "head_ws", "tpl_boxes", "sr", "templates", "order_hint", "trk_ids", "trk_labels", "trk_boxes", "trk_conf",
               # this frame: refinement, detections, solver, next memory
               "refine_ws", "ref_boxes", "ref_scores", "ref_ids", "ref_labels",
               "det_boxes", "det_scores", "det_ids", "det_labels",
               "out_boxes", "out_scores", "out_ids", "out_labels", "act_boxes", "act_ids", "act_labels", "act_scores",
               "record", "next_templates", "next_sr", "next_order_hint",
               # dormant rows carried by the solver's launch (STAGE_CARRY)
               "carry_templates", "carry_boxes", "carry_sr", "carry_ids", "carry_labels", "carry_scores")
_FRAME_INTS = ("n_trk", "stages", "n_det", "num_levels", "C", "rx", "rz", "sampling_ratio", "gn_groups", "up",
               "use_centerness", "refine", "box_pooled", "box_sampling_ratio", "dim6", "dim7", "num_classes", "reg_classes",
               "tracktor", "max_dormant_frames", "pool_capacity", "carry_src_row0", "carry_rows", "carry_dst_row0")
_FRAME_FLOATS = ("track_thresh", "start_thresh", "resume_thresh", "gn_eps", "pad_pixels", "one_minus_sigma", "sigma",
                 "clip_w", "clip_h", "box_wx", "box_wy", "box_ww", "box_wh", "box_xform_clip",
                 "nms_thresh", "search_expansion", "min_search_wh")
STAGE_HEAD, STAGE_REFINE, STAGE_SOLVE, STAGE_EXTRACT, STAGE_CARRY = 1, 2, 4, 8, 16


class FrameArgs(object):
    """``smot_frame_args`` of include/smot_emm.h as one byte buffer: 52 pointers, 24 ints, 17 floats, in the header's
    order.  Fields are plain Python attributes (``__slots__``); ``pack()`` writes all of them with one ``struct.pack_into``
    (a ctypes.Structure costs ~0.4 us per field assignment).  The fields that change from frame to frame are contiguous
    inside each group: ``poke_head`` / ``poke_rest`` rewrite only those ranges of a block that was packed once."""
    __slots__ = _FRAME_PTRS + _FRAME_INTS + _FRAME_FLOATS + ("_buf", "_addr")
    _FMT = struct.Struct("<%dQ%di%df" % (len(_FRAME_PTRS), len(_FRAME_INTS), len(_FRAME_FLOATS)))
    _HEAD0, _HEAD1 = _FRAME_PTRS.index("head_ws"), _FRAME_PTRS.index("trk_conf") + 1
    _REST1 = len(_FRAME_PTRS)
    _INT0 = 8 * len(_FRAME_PTRS)
    _FLT0 = _INT0 + 4 * len(_FRAME_INTS)
    _HEAD_FMT = struct.Struct("<%dQ" % (_HEAD1 - _HEAD0))
    _REST_FMT = struct.Struct("<%dQ" % (_REST1 - _HEAD1))
    _II = struct.Struct("<ii")
    _III = struct.Struct("<iii")
    _FFF = struct.Struct("<fff")
    _CARRY_INT0 = _INT0 + 4 * _FRAME_INTS.index("carry_src_row0")

    def __init__(self):
        for n in _FRAME_PTRS + _FRAME_INTS:
            setattr(self, n, 0)
        for n in _FRAME_FLOATS:
            setattr(self, n, 0.0)
        self._buf = ctypes.create_string_buffer(self._FMT.size + 8)
        self._addr = ctypes.addressof(self._buf)

    def pack(self):
        g = self.__getattribute__
        self._FMT.pack_into(self._buf, 0, *[g(n) or 0 for n in _FRAME_PTRS], *[int(g(n)) for n in _FRAME_INTS],
                            *[float(g(n)) for n in _FRAME_FLOATS])
        return self._addr

    def poke_head(self, ptrs, n_trk, stages):
        """Rewrite the head's per-frame pointers (``head_ws`` .. ``trk_conf``, in that order; None = NULL) and
        ``n_trk``, ``stages``."""
        self._HEAD_FMT.pack_into(self._buf, 8 * self._HEAD0, *ptrs)
        self._II.pack_into(self._buf, self._INT0, n_trk, stages)
        return self._addr

    _HINT_OFF = 8 * _FRAME_PTRS.index("order_hint")
    _Q = struct.Struct("<Q")

    def fix_head(self, n_trk, hint_ptr):
        """Correct the row count and the order-hint pointer of a head range poked on a guess (stage HEAD stands)."""
        self._II.pack_into(self._buf, self._INT0, n_trk, STAGE_HEAD)
        self._Q.pack_into(self._buf, self._HINT_OFF, hint_ptr)

    def poke_rest(self, ptrs, stages, n_det, thresholds, carry=(0, 0, 0), n_trk=None):
        """Rewrite the per-frame pointers behind the head's (``refine_ws`` .. ``carry_scores``), ``stages``,
        ``n_det``, the three solver thresholds and ``carry`` = (carry_src_row0, carry_rows, carry_dst_row0).
        ``n_trk``: the row count of THIS call's propagated tracks — written here too, because the head range may have been
        poked on a guess for a head that never ran (a frame that leaves no track is followed by a call without
        a head; with the guessed count still in the block the solver read one row too many)."""
        self._REST_FMT.pack_into(self._buf, 8 * self._HEAD1, *ptrs)
        if n_trk is None:
            self._II.pack_into(self._buf, self._INT0 + 4, stages, n_det)
        else:
            self._III.pack_into(self._buf, self._INT0, n_trk, stages, n_det)
        self._III.pack_into(self._buf, self._CARRY_INT0, *carry)
        self._FFF.pack_into(self._buf, self._FLT0, *thresholds)
        return self._addr


def track_frame(args, dev, addr=None):
    """``smot_track_frame_fwd``: head [+ box-head refinement] + solver + masked template extraction of ONE tracking frame
    (the stages ``args.stages`` selects) enqueued by one call.  ``args``: a filled ``FrameArgs``, packed here unless the
    caller packed / poked it already and passes the block's address.  Launch only."""
    lib = _lib or load_library()
    cur = torch.cuda.current_device()
    if cur != dev.index:
        torch.cuda.set_device(dev.index)
    try:
        rc = lib.smot_track_frame_fwd(args.pack() if addr is None else addr, _stream(dev))
    finally:
        if cur != dev.index:
            torch.cuda.set_device(cur)
    if rc:
        _check(rc, "track_frame")


def track_frame_addr(lib, addr, dev, stream):
    """``smot_track_frame_fwd`` on a block the caller keeps packed (``FrameArgs.poke_head`` / ``poke_rest``)."""
    cur = torch.cuda.current_device()
    if cur != dev.index:
        torch.cuda.set_device(dev.index)
        try:
            rc = lib.smot_track_frame_fwd(addr, stream)
        finally:
            torch.cuda.set_device(cur)
    else:
        rc = lib.smot_track_frame_fwd(addr, stream)
    if rc:
        _check(rc, "track_frame")


# dormant rows of the track memory copied on the device (TrackingLoop): launches, and frames on which the rows had been
# copied a call early on the guess that nothing changed ("ahead": kept as they were / redone)
MEMORY_CARRY = _collections.Counter()
MEMORY_CARRY_MAX_ROWS = 256
_carry_rows_t = ctypes.c_int * MEMORY_CARRY_MAX_ROWS


def memory_carry(src, src_rows_total, dst, dst_capacity, rows, dst_row0, row_floats, dev, stream, dst_row0_dev=0, lib=None):
    """``smot_memory_carry_fwd`` on bare device addresses: ``src`` / ``dst`` = (templates, boxes, search regions, ids,
    labels, scores) of the memory the frame's head ran on / of the memory being built; ``rows`` the source rows of the
    dormant tracks in the order they are appended (track_head.py:83-86), to rows ``dst_row0`` .. of the destination."""
    lib = lib or _lib or load_library()
    D = len(rows)
    arr = _carry_rows_t(*rows)
    cur = torch.cuda.current_device()
    if cur != dev.index:
        torch.cuda.set_device(dev.index)
    try:
        rc = lib.smot_memory_carry_fwd(src[0], src[1], src[2], src[3], src[4], src[5], int(src_rows_total), dst[0], dst[1], dst[2],
                                       dst[3], dst[4], dst[5], int(dst_capacity), arr, D, int(dst_row0), dst_row0_dev,
                                       int(row_floats), stream)
    finally:
        if cur != dev.index:
            torch.cuda.set_device(cur)
    if rc:
        _check(rc, "memory_carry")
    MEMORY_CARRY["launched"] += 1


def _check_segment(b, s_, i_, l_, dev):
    """The solver's input contract for one segment (boxes, scores, ids, labels or None)."""
    if not (b.is_cuda and b.dtype is _F32 and b.is_contiguous() and s_.dtype is _F32 and s_.is_contiguous()
            and i_.dtype is torch.int64 and i_.is_contiguous()):
        raise RuntimeError("siammot_amd.track_solve: boxes/scores must be contiguous fp32 and ids int64 device tensors")
    if b.device != dev:
        raise RuntimeError("siammot_amd.track_solve: boxes live on %s, the pool state on %s" % (b.device, dev))
    if l_ is not None and not (l_.dtype is torch.int64 and l_.is_contiguous() and l_.device == dev):
        raise RuntimeError("siammot_amd.track_solve: labels must be a contiguous int64 tensor on the boxes' device")


def track_solve_max_boxes():
    return (_lib or load_library()).smot_track_solve_max_boxes()


def preprocess_frame(frame, tables, out_hw, mean, std, to_bgr255):
    """uint8 RGB ``[H,W,3]`` device tensor -> fp32 ``[3,OH,OW]`` network input (resize + ToTensor + Normalize in
    one launch).  ``tables`` = (xbounds, xcoeffs, ybounds, ycoeffs, max_tile_rows) from ``siammot_amd.preprocess``."""
    lib = load_library()
    if not isinstance(frame, torch.Tensor) or not frame.is_cuda or frame.dtype != torch.uint8:
        raise RuntimeError("siammot_amd.preprocess_frame: frame must be a uint8 device (ROCm) tensor — no CPU path exists")
    if frame.dim() != 3 or frame.shape[2] != 3:
        raise RuntimeError("siammot_amd.preprocess_frame: frame must be [H, W, 3], got %s" % (tuple(frame.shape),))
    frame = frame.contiguous()
    H, W, _ = frame.shape
    xb, xk, yb, yk, max_rows = tables
    OH, OW = out_hw
    if xb.shape[0] != OW or yb.shape[0] != OH:
        raise RuntimeError("siammot_amd.preprocess_frame: tables are for %dx%d, asked for %dx%d"
                           % (yb.shape[0], xb.shape[0], OH, OW))
    out = torch.empty((3, OH, OW), dtype=torch.float32, device=frame.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    with _Launch(frame, xb, xk, yb, yk) as ln:
        rc = lib.smot_preprocess_fwd(_ptr(frame), H, W, _ptr(xb), _ptr(xk), xk.shape[1], _ptr(yb), _ptr(yk), yk.shape[1],
                                     OH, OW, int(max_rows), _cast(m), _cast(s), int(bool(to_bgr255)), _ptr(out), ln.stream)
    _check(rc, "preprocess_frame")
    return out
