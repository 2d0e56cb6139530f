"""Workgroup -> roi assignment of the fused pooling+xcorr kernel (sr_xcorr.hip fx_assign): kernel time for the benchmark's
box order and a random-size set under each assignment form (measurement library, SMOT_FUSED_ORDER)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
ops.load_library()
scales = (0.25, 0.125, 0.0625, 0.03125)
feats = [bench.synthetic_features(k, dev) for k in range(4)]
FORMS = {4: "grid order", 1: "sorted, item-major", 2: "front/back pairing", 3: "sorted, roi-major"}
for n in [int(a) for a in sys.argv[1:]] or [30, 100]:
    sets = {"benchmark boxes": bench.synthetic_boxes(n, (1280, 704))}
    rs = np.random.RandomState(n)
    wh = np.exp(rs.uniform(np.log(24), np.log(330), (n, 1))) * np.array([[1.0, 2.0]])
    xy = rs.uniform(0, 1, (n, 2)) * (np.array([1280.0, 704.0]) - wh)
    sets["random sizes (log-uniform 24..330 px wide, 1:2)"] = torch.from_numpy(np.concatenate((xy, xy + wh), 1).astype(np.float32))
    for sname, base in sets.items():
        boxes = base.to(dev)
        sr = ops.search_region(boxes, 512, 1.0, 0)
        z = ops.roi_align_levels(feats[0], boxes, boxes, 15, scales, 2)
        ref = None
        for form, fname in FORMS.items():
            with ops.debug_library(SMOT_FUSED_ORDER=form):
                f = lambda k: ops.sr_xcorr_fused(feats[k % 4], boxes, sr, z, 30, 15, scales, 2, 512)
                out = f(0)
                if ref is None:
                    ref = out.clone()
                same = bool(torch.equal(out, ref))
                for k in range(50): f(k)
                torch.cuda.synchronize()
                ts = []
                for rep in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for k in range(200): f(k)
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 200 * 1e3)
            print(json.dumps({"tracks": n, "boxes": sname, "assignment": fname, "fused_us": round(min(ts), 2),
                              "bitwise_equal_to_grid_order": same}), flush=True)
