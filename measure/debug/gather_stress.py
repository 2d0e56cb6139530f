"""Stress of the 35 / 7 fused gather kernel: random boxes / channel counts / map sizes, compared with the two-kernel form."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import siammot_amd.ops as ops
dev = "cuda:0"
rs = np.random.RandomState(int(os.environ.get("SEED", "1")))
scales = (0.25, 0.125, 0.0625, 0.03125)
bad = 0
for it in range(int(os.environ.get("ITERS", "200"))):
    c = int(rs.choice([1, 2, 3, 4, 5, 8, 30, 128]))
    n = int(rs.randint(1, 40))
    Hn, Wn = int(rs.choice([352, 704, 1056])), int(rs.choice([640, 1280, 1920]))
    feats = tuple(torch.randn((1, c, Hn // s_, Wn // s_), device=dev) for s_ in (4, 8, 16, 32))
    wh = np.exp(rs.uniform(np.log(4), np.log(900), (n, 1))) * np.exp(rs.uniform(-1.5, 1.5, (n, 2)))
    xy = rs.uniform(-0.3, 1.1, (n, 2)) * np.array([Wn, Hn])
    boxes = torch.from_numpy(np.concatenate((xy, xy + wh), 1).astype(np.float32)).to(dev)
    pad = int(rs.choice([0, 256, 512]))
    sr = ops.search_region(boxes, pad, float(rs.choice([1.0, 2.0])), 0)
    z = torch.randn((n, c, 7, 7), device=dev)
    pc = [int(pad / ((2 ** i) * 4)) for i in range(4)]
    mode = os.environ.get("MODE", "both")
    want = got = None
    V = os.environ.get("VERBOSE")
    if V: print("case", it, c, n, Hn, Wn, pad, flush=True)
    if mode in ("both", "ref"):
        x = ops.roi_align_levels(feats, sr, boxes, 35, scales, 2, pc)
        if V: torch.cuda.synchronize(); print("  pooled ok", flush=True)
        want = ops.xcorr_depthwise(x, z)
        if V: torch.cuda.synchronize(); print("  xcorr ok", flush=True)
    if mode in ("both", "fused"):
        got = ops.sr_xcorr_fused(feats, boxes, sr, z, 35, 7, scales, 2, pad)
        if V: torch.cuda.synchronize(); print("  fused ok", flush=True)
    torch.cuda.synchronize()
    if mode == "both" and not torch.equal(got, want):
        bad += 1
        print("MISMATCH", it, c, n, Hn, Wn, pad, float((got - want).abs().max()), flush=True)
print("iterations done, mismatches", bad)
