import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
rs = np.random.RandomState(0)
P = {k: torch.from_numpy(v).to(dev) for k, v in gi.predictor_params(rs, 128, np.array([[0, 0, 80, 120]], dtype=np.float32)).items()}
torch.manual_seed(1)
for n in (1, 8, 30):
    resp = torch.randn(n, 128, 16, 16, device=dev) * 15
    with ops.debug_library(SMOT_TOWER_OCT=2, SMOT_TOWER_BF3=0):
        ref = ops.emm_predictor(resp, P).clone()
    with ops.debug_library(SMOT_TOWER_OCT=2, SMOT_TOWER_BF3=1, SMOT_WINO_ABL=os.environ.get("ABL", "0")):
        pk = ops.tower_packed(P)
        pk0 = pk.clone()
        outs = []
        for _ in range(4):
            torch.cuda.synchronize()
            outs.append(ops.emm_predictor(resp, P).clone())
        torch.cuda.synchronize()
        print(json.dumps({"packed_unchanged": bool((pk.view(torch.int32) == pk0.view(torch.int32)).all()), "packed_floats": pk.numel(),
                          "bf_nonzero_fraction": float((pk[2 * 128 * 128 * 16:].view(torch.int32) != 0).float().mean())}))
    with ops.debug_library(SMOT_TOWER_OCT=2, SMOT_TOWER_BF3=0):
        ref2 = ops.emm_predictor(resp, P).clone()
    print(json.dumps({"fp32_again_equal": bool((ref2 == ref).all())}))
    torch.cuda.synchronize()
    same = [bool((o == outs[0]).all()) for o in outs]
    print(json.dumps({"n": n, "errs_vs_fp32": [float((o - ref).abs().max()) for o in outs], "eq_1_2": bool((outs[1] == outs[2]).all()), "eq_2_3": bool((outs[2] == outs[3]).all())}))
    d = (outs[0] - ref).abs()
    print(json.dumps({"n": n, "repeat_equal": same, "max_err": float(d.max()), "err_by_track": [round(float(x), 4) for x in d.amax(dim=(1, 2, 3))][:8],
                      "err_by_channel": [round(float(x), 4) for x in d.amax(dim=(0, 2, 3))],
                      "err_by_row": [round(float(x), 3) for x in d.amax(dim=(0, 1, 3))], "err_by_col": [round(float(x), 3) for x in d.amax(dim=(0, 1, 2))]}), flush=True)
