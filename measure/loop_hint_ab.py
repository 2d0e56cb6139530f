"""Tracking loop (bench.tracking_loop_throughput) with and without the order hint, same session, interleaved."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import siammot_amd.ops as ops
from siammot_amd.emm import EMM
dev = torch.device("cuda:0")
ops.load_library()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
feats = [bench.synthetic_features(k, dev) for k in range(2)]
for rep in range(2):
    for refine in (False, True):
        for use in (True, False):
            EMM.use_order_hint = use
            with torch.no_grad():
                r = bench.tracking_loop_throughput(N, dev, feats, steps=600, refine=refine)
            print(json.dumps({"refine": refine, "order_hint": use, "ms_per_frame": round(r["ms_per_frame"], 4),
                              "held": r["track_count_held"]}), flush=True)
EMM.use_order_hint = True
