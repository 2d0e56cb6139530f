"""Phase trace (s_memtime ticks, wave 0 of every workgroup) of generation 4 of the fused pooling + correlation kernel."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
ops.load_library()
scales = (0.25, 0.125, 0.0625, 0.03125)
GEN = dict(SMOT_FUSED_GEN=int(os.environ.get("GEN", "10")))
for n in [int(a) for a in sys.argv[1:]] or [30]:
    boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
    feats = bench.synthetic_features(1, dev)
    sr = ops.search_region(boxes, 512, 1.0, 0)
    z = ops.roi_align_levels(feats, boxes, boxes, 15, scales, 2)
    lv = ops.roi_align_levels(feats, boxes, boxes, 15, scales, 2, return_levels=True)[1].cpu().numpy()
    with ops.debug_library(**GEN) as dbg:
        f = lambda: ops.sr_xcorr_fused(feats, boxes, sr, z, 30, 15, scales, 2, 512)
        for _ in range(20): f()
        torch.cuda.synchronize()
        tr = torch.zeros(n * 16 * 8, dtype=torch.int64, device=dev)
        dbg.smot_debug_trace(ops._ptr(tr)); f(); torch.cuda.synchronize(); dbg.smot_debug_trace(None)
    t = tr.view(n, 16, 8).cpu().numpy().astype(np.float64)
    t0 = t[:, :, 0].min()
    ph = {"header": t[:, :, 5] - t[:, :, 0], "loads_issued+park": t[:, :, 2] - t[:, :, 5], "pool": t[:, :, 3] - t[:, :, 2], "xcorr": t[:, :, 4] - t[:, :, 3],
          "total": t[:, :, 4] - t[:, :, 0]}
    nz = t[:, :, 0] > 0
    t0 = t[:, :, 0][nz].min()
    st = np.sort((t[:, :, 0][nz] - t0).ravel()); en = np.sort((t[:, :, 4][nz] - t0).ravel())
    out = {"tracks": n, "gen": GEN, "wg_traced": int(nz.sum()), "wg_total": int(nz.size),
           "start_quantiles": [int(st[int(q * (len(st) - 1))]) for q in (0, .25, .5, .75, .9, 1)],
           "end_quantiles": [int(en[int(q * (len(en) - 1))]) for q in (0, .25, .5, .75, .9, 1)]}
    for k, v in ph.items():
        out[k] = [round(float(v.mean())), int(v.max())]
    sw = (sr[:, 2] - sr[:, 0]).cpu().numpy() * np.array([scales[l] for l in lv])
    wide = sw > 30
    for k in ("pool", "xcorr", "total"):
        out[k + "_wide"] = round(float(ph[k][wide].mean())) if wide.any() else None
        out[k + "_narrow"] = round(float(ph[k][~wide].mean()))
    print(json.dumps(out), flush=True)
