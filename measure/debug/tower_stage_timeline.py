"""One stage of the bf16 x 3 tower loop as a per-wave timeline (SMOT_WINO_ABL=15, measurement library): s_memtime stamps of
stage 9 (block 2, stage 1) for all eight waves of every workgroup.  JSON: mean cycles from the stage's first stamp of wave 0."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
rs = np.random.RandomState(0)
Pn = gi.predictor_params(rs, 128, np.array([[0, 0, 80, 120]], dtype=np.float32))
P = {k: torch.from_numpy(v).to(dev) for k, v in Pn.items()}
n = int(os.environ.get("TRACKS", "30"))
resp = torch.randn(n, 128, 16, 16, device=dev) * 15
with ops.debug_library(SMOT_TOWER_OCT=2, SMOT_TOWER_BF3=1, SMOT_WINO_ABL=15):
    f = lambda: ops.emm_predictor(resp, P)
    for _ in range(20): f()
    torch.cuda.synchronize()
    grid = (n + 7) // 8 * 8 * 8
    tr = torch.zeros(grid * 64, dtype=torch.int64, device=dev)
    lib = ops.load_library()
    lib.smot_debug_trace(ops._ptr(tr)); f(); torch.cuda.synchronize(); tr.zero_(); f(); torch.cuda.synchronize()
    lib.smot_debug_trace(ops._ptr(None))
t = tr.view(grid, 8, 8).cpu().numpy().astype(np.float64)
t = t[t[:, 0, 5] != 0]
base = t[:, :, 0].min(axis=1)[:, None, None]
rel = (t[:, :, :6] - base)
names = ["before_barrier1", "after_barrier1", "raw_planes_stored", "vector_phase_done", "after_vmcnt0_and_barrier2", "matrix_and_fetches_issued"]
print(json.dumps({"tracks": n, "workgroups": int(t.shape[0]), "slots": names,
                  "mean_cycles_by_wave": {("wave%d" % w): [round(float(x), 0) for x in rel[:, w, :].mean(0)] for w in range(8)}}))
