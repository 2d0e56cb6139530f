"""Host-time split of TrackingLoop._step_lean (perf_counter stamps between its steps; no extra syncs), bench workload."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from siammot_amd import ops
from siammot_amd.config import get_default_cfg
from siammot_amd.structures import BoxList
from siammot_amd.track_head import build_tracking_loop
dev = torch.device("cuda:0"); n = 30; image_wh = (1280, 704)
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
import math
cols = max(1, int(math.ceil(math.sqrt(n * image_wh[0] / float(image_wh[1]))))); rows = int(math.ceil(n / float(cols)))
bl = []
for i in range(n):
    w, h = bench.TRACK_SIZES[i % 4]; cx = (i % cols + 0.5) * image_wh[0] / cols; cy = (i // cols + 0.5) * image_wh[1] / rows
    x1 = min(max(cx - w / 2, 0), image_wh[0] - w - 1); y1 = min(max(cy - h / 2, 0), image_wh[1] - h - 1); bl.append([x1, y1, x1 + w, y1 + h])
boxes = torch.tensor(bl, dtype=torch.float32, device=dev)
loop = build_tracking_loop(get_default_cfg(channels=128), device=dev, refine_tracks=False)
bench.init_predictor(loop.track.tracker.predictor, boxes.cpu())
with torch.no_grad():
    for name in ("cls", "center", "reg"): getattr(loop.track.tracker.predictor, name).weight.mul_(0.02)
loop.track.tracker.to(dev)
pre = [(boxes + float(j), torch.full((n,), -1, dtype=torch.int64, device=dev), torch.ones(n, dtype=torch.int64, device=dev), torch.full((n,), 0.9, device=dev)) for j in range(2)]
def dets(k):
    b, ids, labels, scores = pre[k & 1]
    d = BoxList(b, image_wh, mode="xyxy"); d.add_field("ids", ids); d.add_field("labels", labels); d.add_field("scores", scores.clone()); return d
out = loop(feats[0], dets(0)); loop.solver.start_thresh, loop.solver.track_thresh = 2.0, 0.0
for k in range(1, 60): loop(feats[k & 1], dets(k))
torch.cuda.synchronize()
T = time.perf_counter
acc = [0.0] * 6; N = 600
emm, solver, pool = loop.track.tracker, loop.solver, loop.solver.track_pool
with torch.no_grad():
    for k in range(N):
        f = feats[k & 1]; t0 = T()
        d = dets(k); t1 = T()
        z, sr, tb = loop.track_memory; tb0 = tb[0]
        bb, conf = emm.track_raw(f, tb0.bbox, sr[0].bbox, z, tb0.size); trk = (bb, conf, tb0.get_field("ids"), tb0.get_field("labels")); t2 = T()
        state = pool.device_state(dev)
        fbuf, ibuf, rec_host, M = ops.track_solve(solver._segment(d), trk, 1.0, (float(solver.track_thresh), float(solver.start_thresh), float(solver.resume_track_thresh)), float(solver.NMS_THRESH), int(pool._max_dormant_frames), state, pool.DEVICE_CAPACITY, host_record=True); t3 = T()
        ev = ops.stream_event(dev)
        ob, ab, osc, asc = fbuf.split((4 * M, 4 * M, M, M)); act_boxes = ab.view(M, 4)
        prec = emm.extract_cache_rows(f, act_boxes, state[4:5]); t4 = T()
        ops.wait_host_record(rec_host, ev); rec = rec_host.numpy()[:8 + 3 * M + 3 * pool.DEVICE_CAPACITY].copy(); t5 = T()
        K, A = int(rec[0]), int(rec[1]); pool._mirror(rec, M)
        oi, ol, ai, al = ibuf.split((M, M, M, M))
        out = BoxList(ob.view(M, 4)[:K], d.size, mode="xyxy"); out.add_field("ids", oi[:K]); out.add_field("scores", osc[:K]); out.add_field("labels", ol[:K]); out.host_ids = rec[8 + M:8 + M + K]
        act = BoxList(act_boxes[:A], d.size, mode="xyxy"); act.add_field("ids", ai[:A]); act.add_field("scores", asc[:A]); act.add_field("labels", al[:A]); act.host_ids = rec[8 + 2 * M:8 + 2 * M + A].tolist(); out.active_rows = act
        memory = emm.wrap_cache(prec[0][:A], prec[1][:A], act)
        pool.note_memory(memory, act.host_ids); loop.track_memory = memory; t6 = T()
        for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))): acc[i] += b - a
torch.cuda.synchronize()
names = ["dets", "head launch (track_raw)", "solver launch", "event + extract launch", "wait for the record", "views + mirror + memory"]
print(json.dumps({k: round(v / N * 1e6, 1) for k, v in zip(names, acc)} | {"sum_us": round(sum(acc) / N * 1e6, 1), "tracks": A}))
