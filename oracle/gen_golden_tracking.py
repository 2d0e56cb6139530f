#!/usr/bin/env python
"""Generate ``tests/golden/tracking_sequence.npz``: the reference's own TrackHead + TrackSolver + TrackPool
(imported UNMODIFIED from /root/reference, ``maskrcnn_benchmark`` stubbed as in gen_golden_solver.py) driven by the
deterministic fake tracker of tests/fake_tracker.py through the inference branch of CombinedROIHeads.forward
(roi_heads.py:38-50, without the detector's box-head refinement: propagated boxes get score + 1).
Build container only.

Usage:  python oracle/gen_golden_tracking.py
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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import gen_golden_solver as G                                  # noqa: E402
from oracle.ref_structures import BoxList, cat_boxlist         # noqa: E402


def main():
    G.install_stubs()
    reg = types.ModuleType("maskrcnn_benchmark.utils")
    sys.modules["maskrcnn_benchmark.utils"] = reg
    regmod = types.ModuleType("maskrcnn_benchmark.utils.registry")
    regmod.Registry = type("Registry", (dict,), {"register": lambda self, name: (lambda f: f)})
    sys.modules["maskrcnn_benchmark.utils.registry"] = regmod
    sys.path.insert(0, REFERENCE)
    th = os.path.join(REFERENCE, "siammot", "modelling", "track_head")
    solver_mod = G.load("ref_track_solver", os.path.join(th, "track_solver.py"))
    utils_mod = G.load("ref_track_utils", os.path.join(th, "track_utils.py"))
    # track_head.py imports `from siammot.utils import registry` at module level: give it a bare module
    pkg = types.ModuleType("siammot"); pkg.__path__ = []
    upkg = types.ModuleType("siammot.utils"); upkg.__path__ = []
    upkg.registry = types.ModuleType("siammot.utils.registry")
    sys.modules.update({"siammot": pkg, "siammot.utils": upkg, "siammot.utils.registry": upkg.registry})
    head_mod = G.load("ref_track_head", os.path.join(th, "track_head.py"))
    from fake_tracker import SEQ, FakeTracker, detections
    pool = utils_mod.TrackPool(max_dormant_frames=SEQ["max_dormant_frames"])
    tu = types.SimpleNamespace(pad_pixels=SEQ["pad"])
    head = head_mod.TrackHead(FakeTracker(SEQ["pad"], BoxList), None, tu, pool).eval()
    solver = solver_mod.TrackSolver(pool, *SEQ["thresholds"])
    rs = np.random.RandomState(SEQ["seed"])
    memory, out = None, {}
    feats = (torch.zeros(1),)
    for f in range(SEQ["frames"]):
        dets = [detections(rs, f, boxlist_cls=BoxList)]
        _, tracks, _ = head(feats, track_memory=memory)                 # roi_heads.py:38
        if tracks is not None:
            t = tracks[0]
            t.add_field("scores", t.get_field("scores") + 1.0)
            dets = [cat_boxlist(dets + tracks)]
        dets = solver(dets)
        memory = head.get_track_memory(feats, dets)
        res = dets[0]
        out["f%02d_boxes" % f] = res.bbox.numpy().copy()
        out["f%02d_ids" % f] = res.get_field("ids").numpy().copy()
        out["f%02d_scores" % f] = res.get_field("scores").numpy().copy()
        out["f%02d_mem_ids" % f] = memory[2][0].get_field("ids").numpy().copy()
        out["f%02d_mem_feat" % f] = memory[0].numpy().copy().reshape(len(memory[2][0]), -1)
    path = os.path.join(ROOT, "tests", "golden", "tracking_sequence.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "tracks started", pool._max_id + 1, "active", len(pool.get_active_ids()))


if __name__ == "__main__":
    main()
