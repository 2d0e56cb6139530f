"""Placement / race soak of the tower kernel's split form: ONE track replicated n times, every copy must be bit-identical."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
bad = 0; total = 0
for c in (128, 256, 64):
    rs = np.random.RandomState(5 + c)
    P = {k: torch.from_numpy(v).to(dev) for k, v in gi.predictor_params(rs, c, np.array([[0, 0, 80, 120]], dtype=np.float32)).items()}
    one = (rs.standard_normal((1, c, 16, 16)) * 15.0).astype(np.float32)
    for n in (70, 130, 300, 500):
        x = torch.from_numpy(np.repeat(one, n, axis=0)).to(dev)
        for rep in range(int(os.environ.get("REPS", "20"))):
            out = ops.emm_predictor(x, P)
            d = int((out != out[:1]).flatten(1).any(1).sum())
            bad += d; total += n
print(json.dumps({"copies": total, "copies_that_differ_from_the_first": bad}))
