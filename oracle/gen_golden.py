#!/usr/bin/env python
"""Generate ``tests/golden/*.npz`` by running the REFERENCE's own EMM code on CPU.

Runs only in the build container (needs /root/reference; the GPU box never calls this).
The reference files are imported UNMODIFIED:
    siammot/modelling/track_head/EMM/{xcorr,sr_pool,feature_extractor,track_core}.py
    siammot/modelling/track_head/track_utils.py
    siammot/utils/registry.py
Their six absent ``maskrcnn_benchmark`` imports are satisfied by the stubs below
 plus ``numpy.int = int`` (track_core.py:206 uses the removed alias).
The ROIAlign stub is a deliberately naive scalar-loop transcription of the published
upstream algorithm (csrc/cpu/ROIAlign_cpu.cpp) in numpy float32 — a different
implementation from the vectorised oracle it pins.

Inputs come from ``numpy.random.RandomState`` (bit-stable across numpy versions), so only
outputs are stored; ``tests/golden_inputs.py`` rebuilds the inputs from the same seeds.

Usage:  python oracle/gen_golden.py      (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("SIAMMOT_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_inputs as gi            # noqa: E402  (tests/golden_inputs.py)
from oracle.ref_structures import BoxList, cat   # noqa: E402  (the oracle's OWN restatement: nothing of the product)


# ----------------------------------------------------------------------------------------
# maskrcnn_benchmark stubs
# ----------------------------------------------------------------------------------------
class Registry(dict):
    def register(self, name, module=None):
        if module is not None:
            self[name] = module
            return module

        def deco(fn):
            self[name] = fn
            return fn
        return deco


class LevelMapper(object):
    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min, self.k_max = k_min, k_max
        self.s0, self.lvl0, self.eps = canonical_scale, canonical_level, eps

    def __call__(self, boxlists):
        s = torch.sqrt(cat([b.area() for b in boxlists]))
        lvl = torch.floor(self.lvl0 + torch.log2(s / self.s0 + self.eps))
        lvl = torch.clamp(lvl, min=self.k_min, max=self.k_max)
        return lvl.to(torch.int64) - self.k_min


def _bilinear_cols(feat, H, W, y, x):
    """One bilinear sample for all channels; feat [C,H,W] float32 numpy; scalar float32 y, x."""
    f32 = np.float32
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return np.zeros(feat.shape[0], dtype=f32)
    if y <= 0:
        y = f32(0)
    if x <= 0:
        x = f32(0)
    y_low, x_low = int(y), int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1
        y = f32(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1
        x = f32(x_low)
    else:
        x_high = x_low + 1
    ly, lx = f32(y - f32(y_low)), f32(x - f32(x_low))
    hy, hx = f32(f32(1) - ly), f32(f32(1) - lx)
    w1, w2, w3, w4 = f32(hy * hx), f32(hy * lx), f32(ly * hx), f32(ly * lx)
    return (w1 * feat[:, y_low, x_low] + w2 * feat[:, y_low, x_high]
            + w3 * feat[:, y_high, x_low] + w4 * feat[:, y_high, x_high]).astype(f32)


class ROIAlign(nn.Module):
    """Scalar-loop legacy ROIAlign (slow; golden generation only)."""

    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois):
        f32 = np.float32
        feat = input.detach().numpy().astype(f32)
        rois_np = rois.detach().numpy().astype(f32)
        PH, PW = self.output_size
        _, C, H, W = feat.shape
        out = np.zeros((rois_np.shape[0], C, PH, PW), dtype=f32)
        s = f32(self.spatial_scale)
        for r in range(rois_np.shape[0]):
            b = int(rois_np[r, 0])
            sw, sh = f32(rois_np[r, 1] * s), f32(rois_np[r, 2] * s)
            ew, eh = f32(rois_np[r, 3] * s), f32(rois_np[r, 4] * s)
            rw, rh = max(f32(ew - sw), f32(1)), max(f32(eh - sh), f32(1))
            bh, bw = f32(rh / f32(PH)), f32(rw / f32(PW))
            gh = self.sampling_ratio if self.sampling_ratio > 0 else int(np.ceil(rh / PH))
            gw = self.sampling_ratio if self.sampling_ratio > 0 else int(np.ceil(rw / PW))
            for ph in range(PH):
                for pw in range(PW):
                    acc = np.zeros(C, dtype=f32)
                    for iy in range(gh):
                        y = f32(f32(sh + f32(f32(ph) * bh)) + f32(f32(f32(iy + 0.5) * bh) / f32(gh)))
                        for ix in range(gw):
                            x = f32(f32(sw + f32(f32(pw) * bw)) + f32(f32(f32(ix + 0.5) * bw) / f32(gw)))
                            acc = (acc + _bilinear_cols(feat[b], H, W, y, x)).astype(f32)
                    out[r, :, ph, pw] = acc / f32(gh * gw)
        return torch.from_numpy(out)


def make_conv3x3(in_channels, out_channels, dilation=1, stride=1, use_gn=False, use_relu=False,
                 kaiming_init=True):
    conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=dilation,
                     dilation=dilation, bias=False if use_gn else True)
    if kaiming_init:
        nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    else:
        nn.init.normal_(conv.weight, std=0.01)
    if not use_gn:
        nn.init.constant_(conv.bias, 0)
    module = [conv]
    if use_gn:
        module.append(nn.GroupNorm(32, out_channels, 1e-5, affine=True))
    if use_relu:
        module.append(nn.ReLU(inplace=True))
    if len(module) > 1:
        return nn.Sequential(*module)
    return conv


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("maskrcnn_benchmark")
    mod("maskrcnn_benchmark.structures")
    mod("maskrcnn_benchmark.structures.bounding_box", BoxList=BoxList)
    mod("maskrcnn_benchmark.modeling")
    mod("maskrcnn_benchmark.modeling.utils", cat=cat)
    mod("maskrcnn_benchmark.modeling.poolers", LevelMapper=LevelMapper)
    mod("maskrcnn_benchmark.modeling.make_layers", make_conv3x3=make_conv3x3)
    mod("maskrcnn_benchmark.layers", ROIAlign=ROIAlign)
    mod("maskrcnn_benchmark.utils")
    mod("maskrcnn_benchmark.utils.registry", Registry=Registry)
    if not hasattr(np, "int"):
        np.int = int
    sys.path.insert(0, REFERENCE)


def reference_cfg(case):
    ns = types.SimpleNamespace
    emm = ns(USE_CENTERNESS=case["use_centerness"], COSINE_WINDOW_WEIGHT=case["sigma"],
             CLS_POS_REGION=0.8, TRACK_LOSS_WEIGHT=1.0)
    th = ns(POOLER_RESOLUTION=case["rz"], POOLER_SCALES=case["scales"], POOLER_SAMPLING_RATIO=2,
            SEARCH_REGION=case["search_region"], PAD_PIXELS=case["pad_pixels"],
            MINIMUM_SREACH_REGION=case["min_search_wh"], MAX_DORMANT_FRAMES=1, EMM=emm)
    return ns(INPUT=ns(AMODAL=case["amodal"]),
              MODEL=ns(BACKBONE=ns(CONV_BODY="DLA-34-FPN"), DLA=ns(BACKBONE_OUT_CHANNELS=case["channels"]),
                       TRACK_HEAD=th))


def boxlist(boxes, size, n0=0):
    bl = BoxList(torch.from_numpy(boxes.copy()), size, mode="xyxy")
    bl.add_field("ids", torch.arange(n0, n0 + len(boxes), dtype=torch.int64))
    bl.add_field("labels", torch.ones(len(boxes), dtype=torch.int64))
    return bl


def main():
    install_stubs()
    from siammot.modelling.track_head.EMM import track_core as ref_core
    from siammot.modelling.track_head.EMM.xcorr import xcorr_depthwise as ref_xcorr
    from siammot.modelling.track_head.track_utils import build_track_utils
    from siammot.utils import registry as ref_registry

    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.set_grad_enabled(False)

    # ---- full EMM frame-pair cases (extract_cache on frame A, forward on frame B) ----------
    for name, case in gi.EMM_CASES.items():
        inp = gi.emm_case_inputs(name)
        cfg = reference_cfg(case)
        track_utils, _ = build_track_utils(cfg)
        emm = ref_registry.SIAMESE_TRACKER["EMM"](cfg, track_utils).eval()
        emm.predictor.load_state_dict({k: torch.from_numpy(v) for k, v in inp["params"].items()})
        W, H = case["image_wh"]
        feats_a = tuple(torch.from_numpy(f) for f in inp["features_a"])
        feats_b = tuple(torch.from_numpy(f) for f in inp["features_b"])
        det = boxlist(inp["boxes"], (W, H))

        z, sr, det_out = emm.extract_cache(feats_a, det)
        levels = emm.feature_extractor.pooler_z.map_levels([det])

        # intermediates, re-running the same reference calls EMM.forward makes (track_core.py:49-54)
        padded = track_utils.pad_feature(feats_b)
        x = emm.feature_extractor(padded, det_out, sr)
        resp = ref_xcorr(x, z)
        cls, center, reg = emm.predictor(resp)
        _, result, _ = emm(feats_b, det_out, sr, template_features=z)
        res = result[0]

        # un-clipped decode (what decode_response returns before wrap_results_to_boxlist)
        import torch.nn.functional as F
        up = [F.interpolate(t, scale_factor=16, mode="bicubic") for t in (cls, center, reg)]
        loc = ref_core.get_locations(x, z, sr, shift_xy=(case["pad_pixels"],) * 2, up_scale=16)
        bb_raw, conf_raw = ref_core.decode_response(up[0], up[1], up[2], loc, det_out[0],
                                                    use_centerness=case["use_centerness"], sigma=case["sigma"])
        sub = gi.CHANNEL_SUBSET(case["channels"])
        np.savez_compressed(
            os.path.join(out_dir, "emm_%s.npz" % name),
            z=z.numpy(), sr=sr[0].bbox.numpy(), levels=levels.numpy(),
            x_sub=x[:, sub].numpy(), response=resp.numpy(),
            cls=cls.numpy(), center=center.numpy(), reg=reg.numpy(),
            bb_raw=bb_raw.numpy(), conf_raw=conf_raw.numpy(),
            bb=res.bbox.numpy(), scores=res.get_field("scores").numpy(), ids=res.get_field("ids").numpy(),
            loc_corners=loc[:, [0, loc.shape[1] - 1]].numpy())
        print("emm_%s: kept %d/%d tracks, levels %s" % (name, len(res), len(det), levels.tolist()))

    # ---- operator-level cases --------------------------------------------------------------
    for name in gi.XCORR_CASES:
        x, z = gi.xcorr_case_inputs(name)
        out = ref_xcorr(torch.from_numpy(x), torch.from_numpy(z))
        np.savez_compressed(os.path.join(out_dir, "xcorr_%s.npz" % name), out=out.numpy())
        print("xcorr_%s: out %s" % (name, tuple(out.shape)))

    for name, case in gi.DECODE_CASES.items():
        d = gi.decode_case_inputs(name)
        import torch.nn.functional as F
        cls, center, reg = [torch.from_numpy(d[k]) for k in ("cls", "center", "reg")]
        sr = boxlist(d["sr"], (1, 1))
        boxes = boxlist(d["boxes"], (1, 1))
        up = [F.interpolate(t, scale_factor=16, mode="bicubic") for t in (cls, center, reg)]
        Rx, Rz = case["rx"], case["rz"]
        loc = ref_core.get_locations(torch.zeros(1, 1, Rx, Rx), torch.zeros(1, 1, Rz, Rz), [sr],
                                     shift_xy=(case["pad_pixels"],) * 2, up_scale=16)
        bb, conf = ref_core.decode_response(up[0], up[1], up[2], loc, boxes,
                                            use_centerness=case["use_centerness"], sigma=case["sigma"])
        # also pin the up-sampled planes at a sparse lattice and the winning index
        G = up[0].shape[-1]
        N = cls.shape[0]
        tlbr = up[2].reshape(N, 4, -1)
        score, _ = None, None
        lat = slice(0, G, 37)
        np.savez_compressed(os.path.join(out_dir, "decode_%s.npz" % name),
                            bb=bb.numpy(), conf=conf.numpy(),
                            cls_up_lat=up[0][:, :, lat, lat].numpy(),
                            center_up_lat=up[1][:, :, lat, lat].numpy(),
                            reg_up_lat=up[2][:, :, lat, lat].numpy())
        print("decode_%s: bb[0]=%s conf[0]=%.6f" % (name, bb[0].tolist(), float(conf[0])))


if __name__ == "__main__":
    main()
