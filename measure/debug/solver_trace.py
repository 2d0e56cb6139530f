"""Phase trace (s_memtime ticks) of the one-launch track solver on a synthetic frame."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import siammot_amd.ops as ops
from siammot_amd.solver import TrackPool, TrackSolver
from siammot_amd.structures import BoxList
dev = torch.device("cuda:0")
lib = ops.load_library()
rs = np.random.RandomState(0)
for n_det, n_trk in ((30, 30), (100, 100)):
    pool = TrackPool(max_dormant_frames=30)
    solver = TrackSolver(pool, 0.4, 0.6, 0.4)
    def bl(n, ids, lo):
        xy = rs.uniform(0, 1000, (n, 2)); b = BoxList(torch.tensor(np.concatenate((xy, xy + 60), 1), dtype=torch.float32, device=dev), (1280, 704))
        b.add_field("ids", torch.tensor(ids, dtype=torch.int64, device=dev)); b.add_field("scores", torch.tensor(rs.uniform(lo, lo + 0.3, n), dtype=torch.float32, device=dev))
        b.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev)); return b
    solver([bl(n_trk, [-1] * n_trk, 0.65)])
    ids = sorted(pool.get_active_ids())
    for rep in range(3):
        tr = torch.zeros(16, dtype=torch.int64, device=dev)
        lib.smot_debug_trace(ops._ptr(tr))
        solver.solve(bl(n_det, [-1] * n_det, 0.5), bl(len(ids), ids, 0.5), 1.0)
        lib.smot_debug_trace(ops._ptr(None))
        t = tr.cpu().numpy()
    print(json.dumps({"det": n_det, "trk": len(ids), "phase_ticks(load,band,sort,mask,chain,kept,decide,classify,tables,out)": np.diff(t[:10]).tolist(), "total": int(t[9] - t[0])}))
