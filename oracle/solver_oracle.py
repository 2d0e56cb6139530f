"""Literal CPU restatement of the reference track solver — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows siammot/modelling/track_head/track_solver.py:18-108 statement by statement on CPU tensors, with the
per-element Python membership tests the reference uses, and a numpy greedy NMS in place of
[UPSTREAM] maskrcnn_benchmark boxlist_nms / _C.nms (descending score, +1 areas, suppress when IoU > thresh as the
CUDA kernel does, kept indices in ascending original order).
"""
import numpy as np
import torch


def nms_indices(boxes, scores, thresh):
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    b = boxes[order].astype(np.float32)
    area = (b[:, 2] - b[:, 0] + np.float32(1)) * (b[:, 3] - b[:, 1] + np.float32(1))
    dead = np.zeros(len(b), bool)
    for i in range(len(b)):
        if dead[i]:
            continue
        w = np.maximum(np.minimum(b[i, 2], b[i + 1:, 2]) - np.maximum(b[i, 0], b[i + 1:, 0]) + np.float32(1), 0)
        h = np.maximum(np.minimum(b[i, 3], b[i + 1:, 3]) - np.maximum(b[i, 1], b[i + 1:, 1]) + np.float32(1), 0)
        inter = (w * h).astype(np.float32)
        dead[i + 1:] |= inter / (area[i] + area[i + 1:] - inter) > np.float32(thresh)
    return np.sort(order[~dead])


def solve(pool, boxes, ids, scores, track_thresh, start_thresh, resume_thresh):
    """One solver call.  boxes [N,4] float32 xyxy, ids [N] int64, scores [N] float32 (numpy; scores are banded in
    place like the reference's tensor).  Returns (kept row indices, ids, scores) of the output BoxList."""
    if len(boxes) == 0:
        return np.zeros(0, np.int64), ids, scores
    all_ids = torch.from_numpy(ids)
    all_scores = torch.from_numpy(scores)
    active_ids = pool.get_active_ids()
    dormant_ids = pool.get_dormant_ids()
    active_mask = torch.tensor([int(x) in active_ids for x in all_ids], dtype=torch.bool)
    all_scores[active_mask] += 1.
    keep = nms_indices(boxes, all_scores.numpy(), 0.5)
    _ids = all_ids[keep].clone()
    _scores = all_scores[keep].clone()
    _scores[_scores >= 2.] = _scores[_scores >= 2.] - 2.
    _scores[_scores >= 1.] = _scores[_scores >= 1.] - 1.
    start_idxs = ((_ids < 0) & (_scores >= start_thresh)).nonzero()
    inactive_idxs = ((_ids >= 0) & (_scores < track_thresh))
    nms_track_ids = set(_ids[_ids >= 0].tolist())
    all_track_ids = set(all_ids[all_ids >= 0].tolist())
    nms_removed_ids = all_track_ids - nms_track_ids
    inactive_ids = set(_ids[inactive_idxs].tolist()) | nms_removed_ids
    dormant_mask = torch.tensor([int(x) in dormant_ids for x in _ids], dtype=torch.bool)
    resume_ids = _ids[dormant_mask & (_scores >= resume_thresh)]
    for _id in resume_ids.tolist():
        pool.resume_track(_id)
    for _idx in start_idxs:
        _ids[_idx] = pool.start_track()
    active_ids = pool.get_active_ids()
    for _id in inactive_ids:
        if _id in active_ids:
            pool.suspend_track(_id)
    _ids[inactive_idxs] = -1
    pool.expire_tracks()
    pool.increment_frame()
    return keep, _ids.numpy(), _scores.numpy()
