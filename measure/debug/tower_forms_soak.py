"""Soak of the tower kernel's forms: for random track counts / channel counts / response sizes, the form the library picks
against the one-tile fp32 form (rounding-level agreement) and against itself on a second launch (bit equality).  JSON summary."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
rs = np.random.RandomState(int(os.environ.get("SEED", "5")))
boxes = np.array([[0, 0, 80, 120]], dtype=np.float32)
params = {c: {k: torch.from_numpy(v).to(dev) for k, v in gi.predictor_params(rs, c, boxes).items()} for c in (32, 64, 128, 256)}
t0, cases, worst, bad = time.time(), 0, 0.0, []
forms = {}
while time.time() - t0 < float(os.environ.get("SECONDS", "40")):
    c = int(rs.choice([32, 64, 128, 128, 128, 256]))
    ho = int(rs.choice([16, 16, 16, 29]))
    n = int(rs.randint(1, 141 if ho == 16 else 41))
    mag = float(rs.choice([0.01, 1.0, 15.0, 300.0]))
    resp = torch.from_numpy((rs.standard_normal((n, c, ho, ho)) * mag).astype(np.float32)).to(dev)
    a = ops.emm_predictor(resp, params[c])
    b = ops.emm_predictor(resp, params[c])
    with ops.debug_library(SMOT_TOWER_OCT=1):
        one = ops.emm_predictor(resp, params[c])
    torch.cuda.synchronize()
    f = ops.tower_form(n, c, ho)
    forms[f] = forms.get(f, 0) + 1
    scale = one.abs().amax(dim=(0, 2, 3), keepdim=True).clamp_min(1e-30)
    err = float(((a - one).abs() / scale).max())
    worst = max(worst, err)
    if not torch.equal(a, b) or not (err < 3e-6) or not bool(torch.isfinite(a).all()):
        bad.append({"n": n, "c": c, "ho": ho, "mag": mag, "form": f, "repeat_equal": bool(torch.equal(a, b)), "err": err})
    cases += 1
print(json.dumps({"cases": cases, "forms_seen": forms, "worst_rel_err_vs_one_tile_fp32": worst, "failures": bad[:10], "n_failures": len(bad)}))
