"""Where _finish_frame spends its time (native vs python-composed one-launch path)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
import siammot_amd.ops as ops
from siammot_amd.track_head import TrackingLoop
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
acc = {}
T = time.perf_counter
import gc
if os.environ.get("NOGC"): gc.disable()
def finish(self, features, detections, rec_host, fbuf, ibuf, M, pre):
    t = [T()]
    emm, pool = self.track.tracker, self.solver.track_pool
    ob, ab, osc, asc = fbuf.split((4 * M, 4 * M, M, M)); act_boxes = ab.view(M, 4)
    rec = rec_host.numpy()[:8 + 4 * M + 3 * pool.DEVICE_CAPACITY].copy(); t.append(T())
    K, A = int(rec[0]), int(rec[1]); pool._mirror(rec, M); t.append(T())
    oi, ol, ai, al = ibuf.split((M, M, M, M)); cls = detections.__class__
    out = cls(ob.view(M, 4)[:K], detections.size, mode="xyxy"); out.add_field("ids", oi[:K]); out.add_field("scores", osc[:K]); out.add_field("labels", ol[:K])
    out.host_ids = rec[8 + M:8 + M + K]
    act = cls(act_boxes[:A], detections.size, mode="xyxy"); act.add_field("ids", ai[:A]); act.add_field("scores", asc[:A]); act.add_field("labels", al[:A])
    act.host_ids = rec[8 + 2 * M:8 + 2 * M + A].tolist(); out.active_rows = act; t.append(T())
    memory = emm.wrap_cache(pre[0][:A], pre[1][:A], act); t.append(T())
    pool.note_memory(memory, getattr(memory[2][0], "host_ids", act.host_ids)); self.track_memory = memory; t.append(T())
    for i, n in enumerate(("rec_copy", "mirror", "boxlists", "wrap_cache", "note_memory")):
        acc[n] = acc.get(n, 0.0) + t[i + 1] - t[i]
    return out
TrackingLoop._finish_frame = finish
if len(sys.argv) > 2 and sys.argv[2] == "python":
    TrackingLoop._native_ok = lambda self, d: False
refine = sys.argv[1] == "refine"
bench.tracking_loop_throughput(30, dev, feats, steps=50, refine=refine); acc.clear()
out = bench.tracking_loop_throughput(30, dev, feats, steps=600, refine=refine)
print(json.dumps(dict({k: round(v / 630 * 1e6, 1) for k, v in acc.items()}, frame_us=round(out["ms_per_frame"] * 1e3, 1))))
