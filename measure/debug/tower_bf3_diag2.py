import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
rs = np.random.RandomState(0)
P = {k: torch.from_numpy(v).to(dev) for k, v in gi.predictor_params(rs, 128, np.array([[0, 0, 80, 120]], dtype=np.float32)).items()}
torch.manual_seed(1)
resp = torch.randn(130, 128, 16, 16, device=dev) * 15
for knobs in ({}, {"SMOT_TOWER_BF3": 0}):
    with ops.debug_library(SMOT_TOWER_OCT=2, **knobs):
        full = ops.emm_predictor(resp, P)
        for nv in (127, 65, 62, 33, 32, 17):
            part = ops.emm_predictor(resp[:nv].contiguous(), P)
            d = (part != full[:nv]).flatten(1).any(1)
            dd = (part - full[:nv]).abs()
            print(json.dumps({"knobs": knobs, "nv": nv, "tracks_differing": [int(i) for i in torch.nonzero(d).flatten()[:20]], "count": int(d.sum()),
                              "max_abs": float(dd.max()), "by_channel": [float(x) for x in dd.amax(dim=(0, 2, 3))]}))
