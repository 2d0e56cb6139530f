"""Tracking loop: the frame entry point (two calls on a packed block, default) vs the Python-composed form
(loop.native_frame = False), same session, interleaved.  usage: [tracks]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
feats = [bench.synthetic_features(k, dev) for k in range(2)]
for rep in range(3):
    for refine in (False, True):
        for native in (True, False):
            with torch.no_grad():
                r = bench.tracking_loop_throughput(N, dev, feats, steps=600, refine=refine, native=native)
            print(json.dumps({"tracks": N, "refine": refine, "native": native, "ms_per_frame": round(r["ms_per_frame"], 4),
                              "held": r["track_count_held"], "native_frames": r["frame_entry_point_frames"]}), flush=True)
