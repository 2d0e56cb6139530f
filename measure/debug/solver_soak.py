"""Soak of the one-launch solver (csrc/track_solver.hip) against the literal restatement of the reference
(oracle/solver_oracle.py): many seeds, up to a few hundred boxes per frame, random thresholds and dormant windows —
ids, scores, kept boxes and the pool state must agree frame by frame.  Not part of the suite (minutes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import solver_oracle as SO
from test_solver import _scene
from siammot_amd.solver import TrackPool, TrackSolver
from siammot_amd.structures import BoxList
import siammot_amd.ops as ops

dev = torch.device("cuda:0")
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
launches = [0]
real = ops.track_solve
ops.track_solve = lambda *a, **k: (launches.__setitem__(0, launches[0] + 1), real(*a, **k))[1]
bad = 0
for seed in range(seeds):
    rs = np.random.RandomState(1000 + seed)
    dormant = int(rs.randint(1, 8))
    thr = (float(rs.uniform(0.2, 0.6)), float(rs.uniform(0.5, 0.95)), float(rs.uniform(0.2, 0.6)))
    pool_a, pool_b = TrackPool(max_dormant_frames=dormant), TrackPool(max_dormant_frames=dormant)
    solver = TrackSolver(pool_a, *thr)
    max_det = int(rs.choice([10, 40, 120, 250]))
    worst = 0
    for f in range(60):
        boxes, ids, scores = _scene(rs, pool_b, n_det=int(rs.randint(0, max_det)), n_missing=float(rs.uniform(0.0, 0.4)))
        if len(ids) > ops.track_solve_max_boxes():
            boxes, ids, scores = boxes[:500], ids[:500], scores[:500]
        worst = max(worst, len(ids))
        bl = BoxList(torch.from_numpy(boxes).to(dev), (1280, 704), mode="xyxy")
        bl.add_field("ids", torch.from_numpy(ids).to(dev))
        bl.add_field("scores", torch.from_numpy(scores.copy()).to(dev))
        bl.add_field("labels", torch.ones(len(ids), dtype=torch.int64, device=dev))
        out = solver([bl])[0]
        keep, ref_ids, ref_scores = SO.solve(pool_b, boxes, ids.copy(), scores.copy(), *thr)
        ok = (out.get_field("ids").cpu().tolist() == ref_ids.tolist()
              and np.array_equal(out.get_field("scores").cpu().numpy(), ref_scores)
              and np.array_equal(out.bbox.cpu().numpy(), boxes[keep])
              and pool_a.get_active_ids() == pool_b.get_active_ids()
              and pool_a.get_dormant_ids() == pool_b.get_dormant_ids() and pool_a._max_id == pool_b._max_id)
        if not ok:
            bad += 1
            print("seed %d frame %d MISMATCH (%d boxes)" % (seed, f, len(ids)), flush=True)
            break
    print("seed %d: 60 frames, up to %d boxes, %d ids started, dormant window %d: %s" % (
        seed, worst, pool_a._max_id + 1, dormant, "ok" if ok else "FAILED"), flush=True)
print("solver soak done: %d seeds, %d kernel launches, failures: %d" % (seeds, launches[0], bad))
