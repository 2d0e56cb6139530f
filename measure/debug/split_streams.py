"""Feasibility: one video stream, its 30 tracks split into G groups that run on G HIP streams (fork/join by events)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from siammot_amd import ops
from siammot_amd.config import get_default_cfg
from siammot_amd.emm import EMM
from siammot_amd.structures import BoxList
from siammot_amd.track_utils import build_track_utils
dev = torch.device("cuda:0")
n = 30
image_wh = (1280, 704)
boxes = bench.synthetic_boxes(n, image_wh)
cfg = get_default_cfg(channels=128)
emm = EMM(cfg, build_track_utils(cfg)).eval()
bench.init_predictor(emm.predictor, boxes)
emm = emm.to(dev)
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
def mkdet(idx):
    d = BoxList(boxes[idx].to(dev), image_wh, mode="xyxy")
    d.add_field("ids", torch.as_tensor(idx, device=dev)); d.add_field("labels", torch.ones(len(idx), dtype=torch.int64, device=dev))
    return d
for G in (1, 2, 3):
    groups = [list(range(g, n, G)) for g in range(G)]          # interleaved: each group gets every box size
    dets = [mkdet(ix) for ix in groups]
    main = torch.cuda.current_stream()
    side = [torch.cuda.Stream(device=dev) for _ in range(G - 1)]
    fork = torch.cuda.Event(); joins = [torch.cuda.Event() for _ in side]
    with torch.no_grad():
        states = [emm.extract_cache(feats[1], d) for d in dets]
        def step(k):
            if side:
                fork.record(main)
            for g in range(G):
                st = main if g == 0 else side[g - 1]
                with torch.cuda.stream(st):
                    if g > 0: st.wait_event(fork)
                    z, sr, d = states[g]
                    emm(feats[k & 1], d, sr, template_features=z)
                    states[g] = emm.extract_cache(feats[k & 1], dets[g])
                    if g > 0: joins[g - 1].record(st)
            for j in joins: main.wait_event(j)
        for k in range(300): step(k)
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            for k in range(1500): step(k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("groups %d: %.1f us per frame pair (%.0f fp/s)" % (G, dt / 1500 * 1e6, 1500 / dt), flush=True)
