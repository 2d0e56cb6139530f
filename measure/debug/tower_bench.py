"""Tower kernel timing: winograd vs direct, N in {30, 100}; prints JSON lines."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
rs = np.random.RandomState(0)
boxes = np.array([[0, 0, 80, 120]], dtype=np.float32)
P = {k: torch.from_numpy(v).to(dev) for k, v in gi.predictor_params(rs, 128, boxes).items()}
ABLS = os.environ.get("ABLS", "0").split(",")
OCTS = [int(o) for o in os.environ.get("OCTS", "1,2").split(",")]       # 16-channel tiles per workgroup
for n in [int(t) for t in os.environ.get("TRACKS", "30,100").split(",")]:
    resp = torch.randn(n, 128, 16, 16, device=dev) * 15
    for abl, oct_ in [(a, o) for a in ABLS for o in (OCTS if a != "direct" else [0])]:
      with ops.debug_library(SMOT_WINO_ABL=(0 if abl == "direct" else abl), SMOT_TOWER_OCT=oct_, SMOT_TOWER_BF3=os.environ.get("BF3", "1")):
        f = lambda: ops.emm_predictor(resp, P, winograd=(abl != "direct"))
        for _ in range(200): f()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            ops.kernel_timer_begin(ops.TIMER_TOWER, 300)
            for _ in range(300): f()
            ms, cnt = ops.kernel_timer_end(ops.TIMER_TOWER)
            ts.append(ms / cnt * 1e3)
        print(json.dumps({"tracks": n, "variant": abl, "tiles_per_wg": oct_, "bf3": int(os.environ.get("BF3", "1")) if oct_ == 2 else 0, "tower_event_us_min": round(min(ts), 2), "median": round(sorted(ts)[2], 2)}), flush=True)
if os.environ.get("SWEEP"): sys.exit(0)
# phase trace (s_memtime ticks, 100 MHz constant clock on gfx9: report raw ticks and fractions)
for n, oct_ in [(n, o) for n in [int(t) for t in os.environ.get("TRACKS", "30,100").split(",")] for o in OCTS]:
  with ops.debug_library(SMOT_TOWER_OCT=oct_) as lib:
    resp = torch.randn(n, 128, 16, 16, device=dev) * 15
    grid = (n + 7) // 8 * 8 * 16 // oct_
    tr = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
    lib.smot_debug_trace(ops._ptr(tr))
    ops.emm_predictor(resp, P); torch.cuda.synchronize()
    ops.emm_predictor(resp, P); torch.cuda.synchronize()
    lib.smot_debug_trace(ops._ptr(None))
    t = tr.view(grid, 8).cpu().numpy()
    t = t[t[:, 5] != 0]
    t0 = t[:, 0].min()
    d = np.diff(t[:, :6], axis=1)
    print(json.dumps({"tracks": n, "tiles_per_wg": oct_, "blocks": len(t), "phase_ticks_mean": [round(float(x), 1) for x in d.mean(0)],
                      "phase_ticks_max": [int(x) for x in d.max(0)], "block_total_mean": round(float((t[:, 5] - t[:, 0]).mean()), 1),
                      "kernel_span_ticks": int(t[:, 5].max() - t0), "start_spread": int(t[:, 0].max() - t0)}), flush=True)

    if n == 30 and oct_ == OCTS[-1]:
        tt = tr.view(grid, 8).cpu().numpy()
        hw, xcc = tt[:, 6], tt[:, 7]
        cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7; xc = xcc & 0xF
        key = xc * 10000 + se * 1000 + sh * 100 + cu
        import collections
        groups = collections.defaultdict(list)
        for b in range(grid):
            if tt[b, 5] != 0: groups[int(key[b])].append(b)
        sizes = collections.Counter(len(v) for v in groups.values())
        print(json.dumps({"cus_used": len(groups), "blocks_per_cu_hist": dict(sizes)}))
        pairs = [v for v in groups.values() if len(v) == 2][:12]
        print("sample co-resident pairs (block ids):", pairs)
        diffs = collections.Counter((v[1] - v[0]) for v in groups.values() if len(v) == 2)
        print("pair id differences:", diffs.most_common(6))
        print("xcc of blocks 0..15:", [int(x) for x in xc[:16]])
