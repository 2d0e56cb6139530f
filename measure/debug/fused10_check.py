"""Generation 4 (plan-driven, barrier-free fused pooling + correlation) against generation 3, bit for bit: benchmark
geometry and random geometry (all three width classes, channel tails, 1..260 rois)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
ops.load_library()
scales = (0.25, 0.125, 0.0625, 0.03125)
GEN = dict(SMOT_FUSED_GEN=int(os.environ.get("GEN", "10")))
ok = True
def check(tag, feats, boxes):
    global ok
    sr = ops.search_region(boxes, 512, 1.0, 0)
    z = ops.roi_align_levels(feats, boxes, boxes, 15, scales, 2)
    r3, p3 = ops.sr_xcorr_fused(feats, boxes, sr, z, 30, 15, scales, 2, 512, return_pooled=True)
    junk = torch.full((int(r3.numel() + p3.numel()) * 3 + 1024,), float("nan"), device=dev); del junk   # freed blocks the next outputs reuse hold NaNs
    with ops.debug_library(**GEN):
        r4, p4 = ops.sr_xcorr_fused(feats, boxes, sr, z, 30, 15, scales, 2, 512, return_pooled=True)
        r4b = ops.sr_xcorr_fused(feats, boxes, sr, z, 30, 15, scales, 2, 512)
    torch.cuda.synchronize()
    e = dict(tag=tag, pooled_equal=bool(torch.equal(p3, p4)), resp_equal=bool(torch.equal(r3, r4)), resp_equal_nodebug=bool(torch.equal(r3, r4b)),
             pooled_maxdiff=float((p3 - p4).abs().max()) if p3.numel() else 0.0, resp_maxdiff=float((r3 - r4).abs().max()) if r3.numel() else 0.0)
    ok &= e["pooled_equal"] and e["resp_equal"] and e["resp_equal_nodebug"]
    print(json.dumps(e), flush=True)
for n in (30, 7, 100):
    feats = bench.synthetic_features(1, dev)
    check("bench n=%d" % n, feats, bench.synthetic_boxes(n, (1280, 704)).to(dev))
for n, C in ((1, 16), (2, 16), (30, 13), (65, 7), (130, 16), (260, 8), (40, 24)):
    rs = np.random.RandomState(900 + n)
    g = torch.Generator().manual_seed(n)
    feats = tuple(torch.randn((1, C, 704 // s, 1280 // s), generator=g).to(dev) for s in (4, 8, 16, 32))
    wh = np.exp(rs.uniform(np.log(20), np.log(700), (n, 1))) * np.array([[1.0, 1.7]])
    if n == 40:
        wh = wh * np.array([[6.0, 0.3]])                     # degenerate aspect ratios: windows wider than 64 columns
    xy = rs.uniform(-0.2, 1, (n, 2)) * np.maximum(np.array([1280.0, 704.0]) - wh, 1.0)
    boxes = torch.from_numpy(np.concatenate((xy, xy + wh), 1).astype(np.float32)).to(dev)
    check("random n=%d C=%d" % (n, C), feats, boxes)
print("ALL_EQUAL" if ok else "MISMATCH")
