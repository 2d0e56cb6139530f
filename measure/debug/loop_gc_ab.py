"""Tracking loop with the cyclic garbage collector on / off / frozen, same session."""
import gc, json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(k, dev) for k in range(2)]
for rep in range(2):
    for refine in (False, True):
        for mode in ("default", "disabled", "frozen"):
            gc.enable(); gc.unfreeze()
            if mode == "disabled": gc.disable()
            if mode == "frozen": gc.collect(); gc.freeze()
            with torch.no_grad():
                r = bench.tracking_loop_throughput(30, dev, feats, steps=600, refine=refine)
            print(json.dumps({"refine": refine, "gc": mode, "ms_per_frame": round(r["ms_per_frame"], 4), "counts": gc.get_count()}), flush=True)
gc.enable(); gc.unfreeze()
