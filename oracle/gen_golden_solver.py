#!/usr/bin/env python
"""Generate ``tests/golden/solver_sequence.npz`` by running the REFERENCE's own track solver on CPU.

Runs only in the build container (needs /root/reference).  Imported UNMODIFIED:
    siammot/modelling/track_head/track_solver.py   (TrackSolver)
    siammot/modelling/track_head/track_utils.py    (TrackPool)
Their ``maskrcnn_benchmark`` imports (BoxList, boxlist_nms, cat_boxlist) are satisfied by the oracle's own BoxList (oracle/ref_structures.py)
restatement and by a numpy greedy NMS with upstream's semantics (oracle/solver_oracle.py::nms_indices).
A seeded sequence of random frames (tests/test_solver.py::_scene) is pushed through the solver; per frame the
kept rows, output ids and scores and the pool state are stored.  tests/test_solver.py replays the same sequence
through the oracle restatement and through the product solver.

Usage:  python oracle/gen_golden_solver.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("SIAMMOT_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import solver_oracle as SO                       # noqa: E402
from oracle.ref_structures import BoxList, cat_boxlist       # noqa: E402


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):
    """[UPSTREAM] structures/boxlist_ops.py::boxlist_nms over the numpy NMS."""
    if nms_thresh <= 0:
        return boxlist
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = SO.nms_indices(boxlist.bbox.numpy(), boxlist.get_field(score_field).numpy(), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[torch.from_numpy(keep)].convert(mode)


def install_stubs():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    mod("maskrcnn_benchmark")
    mod("maskrcnn_benchmark.structures")
    bb = mod("maskrcnn_benchmark.structures.bounding_box")
    bb.BoxList = BoxList
    ops_ = mod("maskrcnn_benchmark.structures.boxlist_ops")
    ops_.boxlist_nms = boxlist_nms
    ops_.cat_boxlist = cat_boxlist


def load(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    install_stubs()
    th = os.path.join(REFERENCE, "siammot", "modelling", "track_head")
    solver_mod = load("ref_track_solver", os.path.join(th, "track_solver.py"))
    utils_mod = load("ref_track_utils", os.path.join(th, "track_utils.py"))
    from test_solver import SEQUENCE, _scene
    pool = utils_mod.TrackPool(max_dormant_frames=SEQUENCE["max_dormant_frames"])
    solver = solver_mod.TrackSolver(pool, *SEQUENCE["thresholds"])
    rs = np.random.RandomState(SEQUENCE["seed"])
    out = {}
    for f in range(SEQUENCE["frames"]):
        boxes, ids, scores = _scene(rs, pool, n_det=int(rs.randint(0, 40)), n_missing=0.15)
        bl = BoxList(torch.from_numpy(boxes), (1280, 704), mode="xyxy")
        bl.add_field("ids", torch.from_numpy(ids))
        bl.add_field("scores", torch.from_numpy(scores.copy()))
        res = solver([bl])[0]
        out["f%02d_boxes" % f] = res.bbox.numpy().copy()
        out["f%02d_ids" % f] = res.get_field("ids").numpy().copy()
        out["f%02d_scores" % f] = res.get_field("scores").numpy().copy()
        out["f%02d_active" % f] = np.array(sorted(pool.get_active_ids()), dtype=np.int64)
        out["f%02d_dormant" % f] = np.array(sorted(pool.get_dormant_ids()), dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "solver_sequence.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "frames", SEQUENCE["frames"], "tracks started", pool._max_id + 1)


if __name__ == "__main__":
    main()
