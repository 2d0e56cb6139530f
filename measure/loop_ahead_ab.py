"""Tracking loop with and without the speculative next-frame head (TrackingLoop.forward(..., next_features=)), interleaved
in one session; JSON lines."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
for N in [int(a) for a in sys.argv[1:]] or [30]:
    for refine in (False, True):
        for rep in range(3):
            for ahead in (False, True):
                r = bench.tracking_loop_throughput(N, dev, feats, steps=600, refine=refine, ahead=ahead)
                print(json.dumps({"tracks": N, "refine": refine, "next_frame_shown": ahead, "ms_per_frame": round(r["ms_per_frame"], 5),
                                  "held": r["track_count_held"], "speculative_heads": r["speculative_heads"]}), flush=True)
