"""Tower kernel duration by track count for the one-tile fp32 form and the two-tile split form (measurement library)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
C = int(os.environ.get("CH", "128"))
rs = np.random.RandomState(0)
Pn = gi.predictor_params(rs, C, np.array([[0, 0, 80, 120]], dtype=np.float32))
P = {k: torch.from_numpy(v).to(dev) for k, v in Pn.items()}
for n in [int(t) for t in os.environ.get("TRACKS", "1,4,8,12,16,20,24,30,32,40,48,64,100").split(",")]:
    resp = torch.randn(n, C, 16, 16, device=dev) * 15
    row = {"tracks": n, "C": C}
    for name, env in (("one_tile_fp32", dict(SMOT_TOWER_OCT=1)), ("two_tiles_split", dict(SMOT_TOWER_OCT=2, SMOT_TOWER_BF3=1))):
        with ops.debug_library(**env):
            f = lambda: ops.emm_predictor(resp, P)
            for _ in range(50): f()
            torch.cuda.synchronize()
            ts = []
            for rep in range(3):
                ops.kernel_timer_begin(ops.TIMER_TOWER, 200)
                for _ in range(200): f()
                ms, cnt = ops.kernel_timer_end(ops.TIMER_TOWER)
                ts.append(ms / cnt * 1e3)
        row[name + "_us"] = round(min(ts), 2)
    print(json.dumps(row), flush=True)
