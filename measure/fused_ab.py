"""A/B of the fused pooling + correlation kernel under measurement-library switches, same session, interleaved:
    python measure/fused_ab.py 30 100 -- SMOT_FUSED_ABL=0 SMOT_FUSED_ABL=3 SMOT_FUSED_ORDER=4
Kernel time = torch events around 200 back-to-back launches over 4 rotating feature sets (min of 5 repeats)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
ops.load_library()
scales = (0.25, 0.125, 0.0625, 0.03125)
args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
counts = [int(a) for a in args[:split]] or [30]
variants = [dict(kv.split("=") for kv in v.split(",")) for v in args[split + 1:]] or [{}]
feats = [bench.synthetic_features(k, dev) for k in range(4)]
for n in counts:
    boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
    sr = ops.search_region(boxes, 512, 1.0, 0)
    z = ops.roi_align_levels(feats[0], boxes, boxes, 15, scales, 2)
    ref = None
    for rep in range(2):
        for var in variants:
            with ops.debug_library(**var):
                f = lambda k: ops.sr_xcorr_fused(feats[k % 4], boxes, sr, z, 30, 15, scales, 2, 512)
                out = f(0)
                ref = out.clone() if ref is None else ref
                same = bool(torch.equal(out, ref))
                for k in range(50): f(k)
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for k in range(200): f(k)
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 200 * 1e3)
            print(json.dumps({"tracks": n, "variant": var, "fused_us": round(min(ts), 2), "bitwise_equal": same}), flush=True)
