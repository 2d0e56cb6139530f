import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
from oracle import emm_oracle as O
DEV = "cuda:0"
n, channels, image_wh = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 128, (1280, 704)
case = dict(gi.EMM_CASES["default"], channels=channels, image_wh=image_wh)
rs = np.random.RandomState(n)
shapes = gi.feature_shapes(image_wh, channels)
feats_a = [rs.standard_normal(s).astype(np.float32) for s in shapes]
feats_b = [rs.standard_normal(s).astype(np.float32) for s in shapes]
sizes = [(32, 64), (64, 128), (100, 200), (160, 320), (320, 640)]
boxes = []
for i in range(n):
    w, h = sizes[i % 5] if i < 10 else sizes[i % 4]
    x1 = rs.uniform(0, image_wh[0] - w - 1); y1 = rs.uniform(0, image_wh[1] - h - 1)
    boxes.append([x1, y1, x1 + w, y1 + h])
boxes = np.array(boxes, dtype=np.float32)
params = gi.predictor_params(rs, channels, boxes)
d = lambda a: torch.from_numpy(np.asarray(a)).float().to(DEV)
fa, fb = [d(f) for f in feats_a], [d(f) for f in feats_b]
P = {k: d(v) for k, v in params.items()}
scales = case["scales"]
B = d(boxes)
sr = ops.search_region(B, 512, 1.0, 0)
z = ops.roi_align_levels(fa, B, B, 15, scales, 2)
x = ops.roi_align_levels(fb, sr, B, 30, scales, 2, [128, 64, 32, 16])
resp = ops.xcorr_depthwise(x, z)
respf = ops.sr_xcorr_fused(fb, B, sr, z, 30, 15, scales, 2, 512)
print("fused vs unfused resp", float((resp - respf).abs().max()), float(resp.abs().max()))
lw = ops.emm_predictor(resp, P); ld = ops.emm_predictor(resp, P, winograd=False)
print("wino vs direct logits", float((lw - ld).abs().max()), float(ld.abs().max()))
bb, conf, idx = ops.emm_decode(ld, sr, B, 30, 15, 512, return_index=True, clip_wh=image_wh)
bt, ct, it = ops.emm_track(fb, B, sr, z, P, 30, 15, scales, 2, 512, clip_wh=image_wh, return_index=True)
print("track vs composition: idx mismatches", int((idx != it).sum()), float((bb - bt).abs().max()))
sample = np.array([0, 1, 2, 3, 4, n - 1])
cfg = O.EMMConfig(channels=channels, rz=15, search_region=2.0, scales=scales, pad_pixels=512, min_search_wh=0, use_centerness=True, sigma=0.4, amodal=False)
t = lambda a: torch.from_numpy(np.asarray(a)).float()
z_ref, sr_ref = O.extract_cache(cfg, [t(f) for f in feats_a], t(boxes[sample]))
print("z", float((z[sample].cpu() - z_ref).abs().max()), "sr", float((sr[sample].cpu() - sr_ref).abs().max()))
bbo, confo, keep, extra = O.emm_forward(cfg, {k: t(v) for k, v in params.items()}, [t(f) for f in feats_b], t(boxes[sample]), sr_ref, z_ref, image_wh, return_intermediates=True)
print("oracle bb", bbo[:3]); print("gpu bb", bt[sample][:3].cpu())
print("x err", float((x[sample].cpu() - extra["sr_features"]).abs().max()))
print("resp err", float((resp[sample].cpu() - extra["response"]).abs().max()), float(extra["response"].abs().max()))
lo = torch.cat((extra["cls"], extra["center"], extra["reg"]), 1)
print("logit err direct", float((ld[sample].cpu() - lo).abs().max()), "wino", float((lw[sample].cpu() - lo).abs().max()), float(lo.abs().max()))
print("idx oracle", extra["idx"].tolist(), "gpu", it[sample].cpu().tolist(), "comp", idx[sample].cpu().tolist())
