"""Tracking loop through the frame entry point with / without the extraction's order hint handed to the next head as a bare
address (round 4: no host object, `_LazyMemory.hint_ptr`), same session, interleaved.  usage: [tracks ...]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(k, dev) for k in range(2)]
for N in [int(a) for a in sys.argv[1:]] or [30]:
    for rep in range(3):
        for refine in (False, True):
            for hint in (True, False):
                with torch.no_grad():
                    r = bench.tracking_loop_throughput(N, dev, feats, steps=600, refine=refine, native=True, loop_hint=hint)
                print(json.dumps({"tracks": N, "refine": refine, "loop_hint": hint, "ms_per_frame": round(r["ms_per_frame"], 4),
                                  "held": r["track_count_held"], "native_frames": r["frame_entry_point_frames"]}), flush=True)
