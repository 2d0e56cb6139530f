"""Where the tower kernel's duration goes beyond one workgroup's own cycles: start offsets, durations and end times of all
workgroups of one launch (s_memtime stamps of thread 0: slot 0 = first instruction, slot 5 = after the head stores)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
rs = np.random.RandomState(0)
P = {k: torch.from_numpy(v).to(dev) for k, v in gi.predictor_params(rs, 128, np.array([[0, 0, 80, 120]], dtype=np.float32)).items()}
for n in [int(t) for t in os.environ.get("TRACKS", "30").split(",")]:
    resp = torch.randn(n, 128, 16, 16, device=dev) * 15
    with ops.debug_library(SMOT_TOWER_OCT=2, SMOT_TOWER_BF3=1):
        f = lambda: ops.emm_predictor(resp, P)
        for _ in range(30): f()
        torch.cuda.synchronize()
        grid = (n + 7) // 8 * 8 * 8
        tr = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
        lib = ops.load_library()
        lib.smot_debug_trace(ops._ptr(tr))
        rows = []
        for rep in range(5):
            tr.zero_(); f(); torch.cuda.synchronize()
            t = tr.view(grid, 8).cpu().numpy().astype(np.float64); t = t[t[:, 5] != 0]
            t0 = t[:, 0].min()
            rows.append({"workgroups": int(t.shape[0]), "start_offset_mean": float((t[:, 0] - t0).mean()), "start_offset_max": float((t[:, 0] - t0).max()),
                         "duration_mean": float((t[:, 5] - t[:, 0]).mean()), "duration_min": float((t[:, 5] - t[:, 0]).min()), "duration_max": float((t[:, 5] - t[:, 0]).max()),
                         "first_start_to_last_end": float(t[:, 5].max() - t0),
                         "phase_max": [float(x) for x in np.diff(t[:, :6], axis=1).max(0)], "phase_p90": [float(x) for x in np.percentile(np.diff(t[:, :6], axis=1), 90, axis=0)]})
        lib.smot_debug_trace(ops._ptr(None))
        ts = []
        for rep in range(3):
            ops.kernel_timer_begin(ops.TIMER_TOWER, 200)
            for _ in range(200): f()
            ms, cnt = ops.kernel_timer_end(ops.TIMER_TOWER)
            ts.append(ms / cnt * 1e3)
    print(json.dumps({"tracks": n, "tower_us": round(min(ts), 2), "launches": rows[-2:]}))
