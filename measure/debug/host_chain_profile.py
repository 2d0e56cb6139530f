"""Host time on the tracking loop's serial chain, WITHOUT a GPU: ``TrackingLoop._step_native`` with the library call emulated
on CPU memory (tests/test_solver.py::_FrameEntryEmulation, the harness of
test_frame_entry_point_bookkeeping_with_speculation_and_carried_rows_on_cpu).  Per frame two spans are host work the GPU
waits for in the synchronous (reference-contract) loop:
    pre   entry of forward()            -> the head's launch call
    post  return of the last launch call -> return of forward()      (record read, pool mirror, outputs, next memory)
Steady traffic (every track holds).  usage: python measure/debug/host_chain_profile.py [tracks] [frames] [--cprofile] [--sync]
"""
import ctypes, os, sys, time, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_solver as TS
import siammot_amd.ops as ops_
from fake_tracker import FakeTracker
from siammot_amd.solver import TrackPool, TrackSolver
from siammot_amd.structures import BoxList
from siammot_amd.track_head import TrackHead, TrackingLoop

torch.set_grad_enabled(False)          # inference, as the reference's callers run it (demo_inference.py:103)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
frames = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3000
pad, thresholds = 512, (0.0, 2.0, 2.0)
cap = TrackPool.DEVICE_CAPACITY


class NativeFake(FakeTracker):
    rz, rx, pad_pixels, sigma, amodal, use_centerness = 1, 2, pad, 0.4, False, True

    def __init__(self):
        super(NativeFake, self).__init__(pad)
        self.track_utils = types.SimpleNamespace(pad_pixels=pad, search_expansion=1.0, min_search_wh=0.0)
        self.feature_extractor = types.SimpleNamespace(pooler_x=types.SimpleNamespace(scales=(0.25,), sampling_ratio=2))
        self.predictor = types.SimpleNamespace(param_dict=lambda: {}, gn_groups=1, gn_eps=1e-5)


pool = TrackPool(max_dormant_frames=1000)
loop = TrackingLoop(TrackHead(NativeFake(), types.SimpleNamespace(pad_pixels=pad), pool).eval(),
                    TrackSolver(pool, *thresholds, nms_mask_fn=TS._numpy_mask)).eval()
host = TS._HostEmulation(loop, thresholds, 1000)
emu = TS._FrameEntryEmulation(host, pad, 4)
spans = {"in_emu": 0.0}
marks = []
waits = []


def timed_emu(lib, addr, dev, stream):
    t0 = time.perf_counter()
    r = emu(lib, addr, dev, stream)
    t1 = time.perf_counter()
    marks.append((t0, t1))
    return r


geom = types.SimpleNamespace(a_fp=0, a_hs=0, a_ws=0, a_pc=0, a_sc=0, L=1, C=4)
ops_._geometry = lambda features, scales, pad_pixels, dev: geom
ops_._geometry_refresh = lambda g, features, dev: True
ops_._param_block = lambda params: types.SimpleNamespace(a_pp=4711)
ops_._check_segment = lambda *a: None
ops_._stream = lambda dev=None: ctypes.c_void_p(0)
ops_.load_library = lambda: types.SimpleNamespace(smot_emm_track_ws_floats=lambda *a: 64, smot_box_refine_ws_floats=lambda *a: 0)
ops_.track_frame_addr = timed_emu
ops_.memory_carry = host.memory_carry
state = torch.zeros(8 + 3 * cap, dtype=torch.int32)
pool.device_state = lambda dev: state


class Ring(TS._HostEmulation.Ring):
    bufs = [torch.zeros(8 + 4 * 512 + 3 * cap, dtype=torch.int32) for _ in range(2)]
    k = 0

    def next(self):
        self.k ^= 1
        self.bufs[self.k][3] = 0
        return self.bufs[self.k]

    def wait(self, rec, event=True):
        waits.append(time.perf_counter())          # the frame's synchronisation: host work before it runs beside the GPU


ring = Ring()
pool.host_record_ring = lambda dev: ring
# first frame starts n tracks (start threshold 2.0 would start none: one frame at 0.5)
xy = np.stack([(np.arange(n) % 8) * 150.0 + 10, (np.arange(n) // 8) * 160.0 + 10], 1).astype(np.float32)
boxes = torch.from_numpy(np.concatenate([xy, xy + 60], 1))
feats = (torch.zeros(1, 4, 8, 8),)


def dets():
    d = BoxList(boxes.clone(), (1280, 704), mode="xyxy")
    d.add_field("ids", torch.full((n,), -1, dtype=torch.int64))
    d.add_field("labels", torch.ones(n, dtype=torch.int64))
    d.add_field("scores", torch.full((n,), 0.97))
    return d


EARLY = "--sync" not in sys.argv        # TrackingLoop.forward (early head launch, round 5) / _step_native alone (round 4)
loop._lean_ok = lambda d: True
loop._native_ok = lambda d: True
loop.solver.start_thresh = host.solver.start_thresh = 0.5
loop._step_native(feats, dets())
loop.solver.start_thresh = host.solver.start_thresh = 2.0
pre, post, total, beside = [], [], [], []


def run(k):
    for _ in range(k):
        d = dets()
        marks.clear()
        waits.clear()
        t0 = time.perf_counter()
        out = loop(feats, d) if EARLY else loop._step_native(feats, d)
        t1 = time.perf_counter()
        pre.append(marks[0][0] - t0)
        post.append(t1 - waits[-1])
        beside.append(waits[-1] - marks[-1][1])
        total.append((t1 - t0) - sum(b - a for a, b in marks))
    return out


out = run(200)
pre.clear(); post.clear(); total.clear(); beside.clear()
if "--cprofile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    out = run(frames)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
else:
    out = run(frames)
med = lambda v: float(np.median(v)) * 1e6
print("tracks held %d of %d; frames %d" % (int((out.get_field("ids") >= 0).sum()), n, frames))
print("host us per frame (median): pre-launch %.1f, post-record %.1f (on the serial chain); last launch -> wait %.1f (beside the "
      "GPU); all host work outside the library %.1f" % (med(pre), med(post), med(beside), med(total)))
