"""The 29 x 29 towers (second yaml family): fp32 blocked Winograd against the three-part bf16 experiment (SMOT_TOWER_BF3=2)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
rs = np.random.RandomState(0)
P = {k: torch.from_numpy(v).to(dev) for k, v in gi.predictor_params(rs, 128, np.array([[0, 0, 80, 120]], dtype=np.float32)).items()}
torch.manual_seed(3)
for n in (30,):
    resp = torch.randn(n, 128, 29, 29, device=dev) * 15
    outs = {}
    for name, knobs in (("fp32", {"SMOT_TOWER_BF3": 1}), ("bf16x3", {"SMOT_TOWER_BF3": 2})):
        with ops.debug_library(**knobs):
            f = lambda: ops.emm_predictor(resp, P)
            outs[name] = f().clone()
            for _ in range(50): f()
            torch.cuda.synchronize()
            ts = []
            for rep in range(3):
                ops.kernel_timer_begin(ops.TIMER_TOWER, 100)
                for _ in range(100): f()
                ms, cnt = ops.kernel_timer_end(ops.TIMER_TOWER)
                ts.append(ms / cnt * 1e3)
            again = f()
        print(json.dumps({"tracks": n, "form": name, "tower_us_min": round(min(ts), 2), "repeat_equal": bool((again == outs[name]).all())}), flush=True)
    d = (outs["bf16x3"] - outs["fp32"]).abs()
    scale = outs["fp32"].abs().amax(dim=(0, 2, 3), keepdim=True)
    print(json.dumps({"max_abs": float(d.max()), "max_rel_to_channel_scale": float((d / scale).max())}))
