"""Tracking loop under the reference's one-frame contract (TrackingLoop.forward(features, detections), no next frame shown)
with and without the early head (round 5: the next call's head launch prepared while the GPU works, enqueued on the next
call's first line), interleaved in one session; JSON lines.   python measure/loop_early_ab.py [tracks ...]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
for N in [int(a) for a in sys.argv[1:]] or [30]:
    for refine, dormant in ((False, 0), (True, 0), (False, max(1, N // 5))):
        for rep in range(3):
            for early in (False, True):
                r = bench.tracking_loop_throughput(N, dev, feats, steps=600, refine=refine, early_head=early, dormant=dormant)
                print(json.dumps({"tracks": N, "refine": refine, "dormant": dormant, "early_head": early,
                                  "ms_per_frame": round(r["ms_per_frame"], 5), "held": r["track_count_held"],
                                  "early_heads": r["early_heads"]}), flush=True)
