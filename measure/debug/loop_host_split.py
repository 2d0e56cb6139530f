"""Host-time split of one frame of the one-launch tracking loop: time inside each binding call (enqueue only), the wait for
the solver's record, and the rest (Python glue), from perf_counter stamps around the calls.  usage: [refine]"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
refine = len(sys.argv) > 1 and sys.argv[1] == "refine"
acc = {}
def wrap(obj, name, key=None):
    f = getattr(obj, name); key = key or name
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
for n in ("emm_track", "box_refine", "track_solve", "emm_extract_cache", "track_frame", "_geometry", "_param_block"):
    wrap(ops, n)
wrap(ops.HostRecordRing, "wait", "record_wait")
wrap(ops.FrameArgs, "pack", "args_pack")
from siammot_amd.track_head import TrackingLoop
wrap(TrackingLoop, "_finish_frame", "finish_frame")
wrap(TrackingLoop, "_native_ok", "native_ok")
wrap(TrackingLoop, "_lean_ok", "lean_ok")
if len(sys.argv) > 2 and sys.argv[2] == "python":
    TrackingLoop._native_ok = lambda self, d: False
bench.tracking_loop_throughput(30, dev, feats, steps=50, refine=refine)
acc.clear()
out = bench.tracking_loop_throughput(30, dev, feats, steps=600, refine=refine)
frames = 600 + 30
per = {k: round(v / frames * 1e6, 1) for k, v in acc.items()}
per["frame_us"] = round(out["ms_per_frame"] * 1e3, 1)
per["python_glue_us"] = round(per["frame_us"] - sum(v for k, v in per.items() if k != "frame_us"), 1)
print(json.dumps(per))
