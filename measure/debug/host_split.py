"""Where the host time of a frame pair goes: python glue vs the C-ABI call (HIP launches)."""
import os, sys, time, ctypes, torch
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
feats = bench.synthetic_features(100, dev)
det = BoxList(boxes.to(dev), image_wh, mode="xyxy")
det.add_field("ids", torch.arange(n, device=dev)); det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
lib = ops.load_library()
real_track, real_cache = lib.smot_emm_track_fwd, lib.smot_emm_extract_cache_fwd
acc = {"track": 0.0, "cache": 0.0}
def wrap(name, f):
    def g(*a):
        t = time.perf_counter(); r = f(*a); acc[name] += time.perf_counter() - t; return r
    return g
class L(object):
    def __getattr__(self, k):
        if k == "smot_emm_track_fwd": return wrap("track", real_track)
        if k == "smot_emm_extract_cache_fwd": return wrap("cache", real_cache)
        return getattr(lib, k)

with torch.no_grad():
    state = emm.extract_cache(feats, det)
    for k in range(200):
        z, sr, d = state
        emm(feats, d, sr, template_features=z); state = emm.extract_cache(feats, det)
    torch.cuda.synchronize()
    # 1. raw C calls in a burst
    import siammot_amd.ops as O
    orig_load = O.load_library
    O.load_library = lambda *a, **k: L()
    for rep in range(3):
        acc["track"] = acc["cache"] = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(40):
            z, sr, d = state
            emm(feats, d, sr, template_features=z); state = emm.extract_cache(feats, det)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print("burst 40: host total %.1f us/step; inside C: track %.1f us, extract_cache %.1f us; python glue %.1f us"
              % ((t1 - t0) / 40 * 1e6, acc["track"] / 40 * 1e6, acc["cache"] / 40 * 1e6,
                 ((t1 - t0) - acc["track"] - acc["cache"]) / 40 * 1e6), flush=True)
    O.load_library = orig_load
    import cProfile, pstats
    pr = cProfile.Profile()
    for rep in range(5):
        torch.cuda.synchronize()
        pr.enable()
        for k in range(40):
            z, sr, d = state
            emm(feats, d, sr, template_features=z); state = emm.extract_cache(feats, det)
        pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
