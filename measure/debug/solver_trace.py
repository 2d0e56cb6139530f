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
    host = len(sys.argv) > 1 and sys.argv[1] == "host"       # the record in pinned host memory, as in the tracking loop
    for rep in range(3):
        tr = torch.zeros(16, dtype=torch.int64, device=dev)
        lib.smot_debug_trace(ops._ptr(tr))
        if not host:
            solver.solve(bl(n_det, [-1] * n_det, 0.5), bl(len(ids), ids, 0.5), 1.0)
        else:
            ring = pool.host_record_ring(dev)
            fbuf, ibuf, rec_host, M = ops.track_solve(
                solver._segment(bl(n_det, [-1] * n_det, 0.5)), solver._segment(bl(len(ids), ids, 0.5)), 1.0,
                (float(solver.track_thresh), float(solver.start_thresh), float(solver.resume_track_thresh)),
                float(solver.NMS_THRESH), int(pool._max_dormant_frames), pool.device_state(dev), pool.DEVICE_CAPACITY,
                host_record=ring)
            ring.record_event(); ring.wait(rec_host)
            pool._mirror(rec_host.numpy()[:8 + 4 * M + 3 * pool.DEVICE_CAPACITY].copy(), M)
            ids = sorted(pool.get_active_ids())
        lib.smot_debug_trace(ops._ptr(None))
        torch.cuda.synchronize()
        t = tr.cpu().numpy()
    print(json.dumps({"det": n_det, "trk": len(ids), "phase_ticks(load,band,sort,mask,chain,kept,decide,classify,tables,out)": np.diff(t[:10]).tolist(), "total": int(t[9] - t[0]),
                      "last_phase(scan,outputs,tables,fence,flag)": [int(t[10] - t[8]), int(t[11] - t[10]), int(t[12] - t[11]), int(t[13] - t[12]), int(t[9] - t[13])]}))
