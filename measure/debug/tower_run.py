"""A few hundred tower launches under rocprofv3 (PMC passes): python measure/debug/tower_run.py TRACKS KNOB=VALUE ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
n = int(sys.argv[1])
knobs = dict(a.split("=") for a in sys.argv[2:])
rs = np.random.RandomState(0)
P = {k: torch.from_numpy(v).to("cuda:0") for k, v in gi.predictor_params(rs, 128, np.array([[0, 0, 80, 120]], dtype=np.float32)).items()}
resp = torch.randn(n, 128, 16, 16, device="cuda:0") * 15
with ops.debug_library(**knobs):
    for _ in range(60): ops.emm_predictor(resp, P)
    torch.cuda.synchronize()
