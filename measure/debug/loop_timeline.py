"""Timeline of one frame of the one-launch tracking loop (host stamps relative to the frame's start, mean over 600 frames):
where the host is when — before the first launch, inside the head's binding call, after the solver / extraction are
enqueued, when the record arrives, when the frame is done.  usage: [refine]"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench, gc
gc.disable()
import siammot_amd.ops as ops
from siammot_amd.track_head import TrackingLoop
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
refine = len(sys.argv) > 1 and sys.argv[1] == "refine"
T = time.perf_counter
marks, cur = {}, {}
def stamp(name):
    cur[name] = T()
lib = ops.load_library()
class LibProxy(object):
    def __getattr__(self, k):
        f = getattr(lib, k)
        if k in ("smot_emm_track_fwd", "smot_track_solve_fwd", "smot_emm_extract_cache_masked_fwd", "smot_box_refine_fwd",
                 "smot_track_frame_fwd"):
            def g(*a):
                kk = k + ("#2" if k + ":in" in cur else "")
                stamp(kk + ":in"); r = f(*a); stamp(kk + ":out"); return r
            return g
        return f
ops._lib = LibProxy()
fwd, lean, fin = TrackingLoop.forward, TrackingLoop._step_lean, TrackingLoop._finish_frame
native = TrackingLoop._step_native
def step_native(self, *a, **k):
    stamp("step_native")
    return native(self, *a, **k)
TrackingLoop._step_native = step_native
wait = ops.HostRecordRing.wait
def forward(self, *a, **k):
    cur.clear(); stamp("enter")
    r = fwd(self, *a, **k)
    stamp("done")
    if "wait:out" in cur:
        t0 = cur["enter"]
        for n, v in cur.items():
            marks.setdefault(n, []).append(v - t0)
    return r
def step_lean(self, *a, **k):
    stamp("step_lean")
    return lean(self, *a, **k)
def w(self, rec, *a, **k):
    stamp("wait:in"); r = wait(self, rec, *a, **k); stamp("wait:out"); return r
def finish(self, *a, **k):
    r = fin(self, *a, **k); stamp("finish:out"); return r
TrackingLoop.forward, TrackingLoop._step_lean, TrackingLoop._finish_frame = forward, step_lean, finish
ops.HostRecordRing.wait = w
bench.tracking_loop_throughput(30, dev, feats, steps=50, refine=refine)
marks.clear()
out = bench.tracking_loop_throughput(30, dev, feats, steps=600, refine=refine)
order = sorted(marks, key=lambda n: sum(marks[n]) / len(marks[n]))
print(json.dumps({"frame_us": round(out["ms_per_frame"] * 1e3, 1), "refine": refine,
                  "timeline_us": {n: round(sum(marks[n][30:]) / len(marks[n][30:]) * 1e6, 1) for n in order}}))
