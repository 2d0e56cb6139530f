"""Host-time split of the tracking loop's fast path (perf_counter stamps between its steps; no extra syncs)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from siammot_amd.config import get_default_cfg
from siammot_amd.structures import BoxList
from siammot_amd.track_head import build_tracking_loop
dev = torch.device("cuda:0"); n = 30; image_wh = (1280, 704)
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
boxes = bench.synthetic_boxes(n, image_wh).to(dev)
loop = build_tracking_loop(get_default_cfg(channels=128), device=dev, refine_tracks=False)
bench.init_predictor(loop.track.tracker.predictor, boxes.cpu()); loop.track.tracker.to(dev)
pre = [(boxes + float(j), torch.full((n,), -1, dtype=torch.int64, device=dev), torch.ones(n, dtype=torch.int64, device=dev), torch.full((n,), 0.9, device=dev)) for j in range(2)]
def dets(k):
    b, ids, labels, scores = pre[k & 1]
    d = BoxList(b, image_wh, mode="xyxy"); d.add_field("ids", ids); d.add_field("labels", labels); d.add_field("scores", scores.clone()); return d
for k in range(50): loop(feats[k & 1], dets(k))
torch.cuda.synchronize()
acc = [0.0] * 8
N = 500
T = time.perf_counter
with torch.no_grad():
    for k in range(N):
        f = feats[k & 1]; t0 = T()
        d = dets(k); t1 = T()
        _, tracks, _ = loop.track(f, track_memory=loop.track_memory); t2 = T()
        h = loop.solver.solve_launch(d, tracks[0] if tracks else None, track_score_bias=1.0); t3 = T()
        pre_c = loop.track.tracker.extract_cache_rows(f, h.act_boxes, h.count); t4 = T()
        rec = __import__("siammot_amd.ops", fromlist=["x"]).track_solve_record(h.rec); t5 = T()
        # finish without the sync (already done): replicate solve_finish
        import numpy as np
        M, ref = h.M, h.ref; K, A = int(rec[0]), int(rec[1]); loop.solver.track_pool._mirror(rec, M)
        oi, ol, ai, al = h.ibuf.split((M, M, M, M))
        out = BoxList(h.out_boxes[:K], ref.size); out.add_field("ids", oi[:K]); out.add_field("scores", h.out_scores[:K]); out.add_field("labels", ol[:K]); out.host_ids = rec[8 + M:8 + M + K].astype(np.int64)
        act = BoxList(h.act_boxes[:A], ref.size); act.add_field("ids", ai[:A]); act.add_field("scores", h.act_scores[:A]); act.add_field("labels", al[:A]); act.host_ids = rec[8 + 2 * M:8 + 2 * M + A].tolist(); out.active_rows = act; t6 = T()
        loop.track_memory = loop.track.get_track_memory(f, [out], precomputed=pre_c); t7 = T()
        for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6), (t6, t7))): acc[i] += b - a
torch.cuda.synchronize()
names = ["dets", "head(emm_track)", "solve_launch", "extract_rows", "record+sync", "finish(views)", "get_track_memory"]
print(json.dumps({k: round(v / N * 1e6, 1) for k, v in zip(names, acc)} | {"sum_us": round(sum(acc) / N * 1e6, 1), "tracks": len(loop.track_memory[2][0])}))
