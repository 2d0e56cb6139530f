"""TEST INFRASTRUCTURE — a reference-free transcription of the call sequence the reference drives its tracker through.

/root/reference does not exist on the GPU box, so the reference's own ``TrackHead`` and the real HIP kernels can never meet
in one process there (VERDICT r3 missing #4).  This module restates, statement for statement, the inference path of

    CombinedROIHeads.forward                       siammot/modelling/roi_heads.py:22-52
    TrackHead.forward_inference                    siammot/modelling/track_head/track_head.py:37-46
    TrackHead.get_track_memory                     ... track_head.py:54-75
    TrackHead._update_memory_with_dormant_track    ... track_head.py:77-97
    TrackHead._get_track_targets                   ... track_head.py:99-110
    TrackPool                                      siammot/modelling/track_head/track_utils.py:136-255
    TrackSolver.forward / get_nms_boxes            siammot/modelling/track_head/track_solver.py:22-108

on the ORACLE's upstream-style containers (oracle/ref_structures.py: BoxList, cat_boxlist) — none of this repository's
``TrackHead`` / ``TrackPool`` / ``TrackSolver`` / ``BoxList`` is involved.  The tracker is whatever object is handed in: it is
called exactly as the reference calls its ``EMM`` — ``tracker(features, [template boxes], sr=[sr], template_features=z)``
and ``tracker.extract_cache(features, active_tracks)`` — i.e. through the GENERAL ``forward`` / ``extract_cache`` path, with
memories that were concatenated with dormant tracks' cache rows (``torch.cat`` / ``cat_boxlist``) and indexed BoxLists.
tests/test_sequence.py runs it with the oracle head on CPU (which pins this transcription against the reference-generated
``sequence_plain.npz`` bit for bit) and with ``siammot_amd.emm.EMM`` on the device.
"""
import copy

import numpy as np
import torch

from oracle import solver_oracle as SO
from oracle.ref_structures import BoxList, cat_boxlist


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):
    """[UPSTREAM] structures/boxlist_ops.py::boxlist_nms; the NMS itself = the oracle's numpy restatement of upstream's kernel."""
    if nms_thresh <= 0:
        return boxlist
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = SO.nms_indices(boxlist.bbox.detach().cpu().numpy(), boxlist.get_field(score_field).detach().cpu().numpy(), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[torch.from_numpy(keep).to(boxlist.bbox.device)].convert(mode)


class TrackPool(object):
    """track_utils.py:136-255."""

    def __init__(self, max_dormant_frames=1):
        self._max_dormant_frames = max_dormant_frames
        self.reset()

    def reset(self):                                                     # :243-250
        self._active_ids = set()
        self._kill_ids = set()
        self._dormant_ids = {}
        self._cache = {}
        self._max_id = -1
        self._frame_idx = 0

    def suspend_track(self, track_id):                                   # :157-165
        if track_id not in self._active_ids:
            raise ValueError
        self._active_ids.remove(track_id)
        self._dormant_ids[track_id] = self._frame_idx - 1

    def expire_tracks(self):                                             # :167-176
        for track_id, last_active in list(self._dormant_ids.items()):
            if self._frame_idx - last_active >= self._max_dormant_frames:
                self._dormant_ids.pop(track_id)
                self._kill_ids.add(track_id)
                self._cache.pop(track_id, None)

    def increment_frame(self, value=1):                                  # :178-179
        self._frame_idx += value

    def update_cache(self, cache):                                       # :181-198
        template_features, sr, template_boxes = cache
        sr = sr[0]
        template_boxes = template_boxes[0]
        for idx in range(len(template_boxes)):
            if len(template_features) > 0:
                assert len(template_features) == len(sr)
                features = template_features[idx]
            else:
                features = template_features
            search_region = sr[idx: idx + 1]
            box = template_boxes[idx: idx + 1]
            track_id = box.get_field("ids").item()
            self._cache[track_id] = (features, search_region, box)

    def resume_track(self, track_id):                                    # :200-209
        if track_id not in self._dormant_ids or track_id in self._active_ids:
            raise ValueError
        self._active_ids.add(track_id)
        self._dormant_ids.pop(track_id)

    def start_track(self):                                               # :222-230
        new_id = self._max_id + 1
        self._max_id = new_id
        self._active_ids.add(new_id)
        return new_id

    def get_active_ids(self):
        return self._active_ids

    def get_dormant_ids(self):
        return set(self._dormant_ids.keys())

    def get_cache(self):
        return self._cache


class TrackSolver(torch.nn.Module):
    """track_solver.py:9-108."""

    def __init__(self, track_pool, track_thresh, start_track_thresh, resume_track_thresh):
        super(TrackSolver, self).__init__()
        self.track_pool = track_pool
        self.track_thresh = track_thresh
        self.start_thresh = start_track_thresh
        self.resume_track_thresh = resume_track_thresh

    def get_nms_boxes(self, detection):                                  # :22-34
        detection = boxlist_nms(detection, nms_thresh=0.5)
        _ids = detection.get_field('ids')
        _scores = detection.get_field('scores')
        _scores[_scores >= 2.] = _scores[_scores >= 2.] - 2.
        _scores[_scores >= 1.] = _scores[_scores >= 1.] - 1.
        return detection, _ids, _scores

    def forward(self, detection):                                        # :36-108
        assert len(detection) == 1
        detection = detection[0]
        if len(detection) == 0:
            return [detection]
        track_pool = self.track_pool
        all_ids = detection.get_field('ids')
        all_scores = detection.get_field('scores')
        active_ids = track_pool.get_active_ids()
        dormant_ids = track_pool.get_dormant_ids()
        device = all_ids.device
        active_mask = torch.tensor([int(x) in active_ids for x in all_ids], device=device)
        all_scores[active_mask] += 1.
        nms_detection, nms_ids, nms_scores = self.get_nms_boxes(detection)
        combined_detection = nms_detection
        _ids = combined_detection.get_field('ids')
        _scores = combined_detection.get_field('scores')
        start_idxs = ((_ids < 0) & (_scores >= self.start_thresh)).nonzero()
        inactive_idxs = ((_ids >= 0) & (_scores < self.track_thresh))
        nms_track_ids = set(_ids[_ids >= 0].tolist())
        all_track_ids = set(all_ids[all_ids >= 0].tolist())
        nms_removed_ids = all_track_ids - nms_track_ids
        inactive_ids = set(_ids[inactive_idxs].tolist()) | nms_removed_ids
        dormant_mask = torch.tensor([int(x) in dormant_ids for x in _ids], device=device)
        resume_ids = _ids[dormant_mask & (_scores >= self.resume_track_thresh)]
        for _id in resume_ids.tolist():
            track_pool.resume_track(_id)
        for _idx in start_idxs:
            _ids[_idx] = track_pool.start_track()
        active_ids = track_pool.get_active_ids()
        for _id in inactive_ids:
            if _id in active_ids:
                track_pool.suspend_track(_id)
        _ids[inactive_idxs] = -1
        track_pool.expire_tracks()
        track_pool.increment_frame()
        return [combined_detection]


class TrackHead(torch.nn.Module):
    """track_head.py:8-110 (inference)."""

    def __init__(self, tracker, track_utils, track_pool):
        super(TrackHead, self).__init__()
        self.tracker = tracker
        self.track_utils = track_utils
        self.track_pool = track_pool

    def forward(self, features, proposals=None, targets=None, track_memory=None):        # :18-22 / :37-46
        track_boxes = None
        if track_memory is None:
            self.track_pool.reset()
        else:
            (template_features, sr, template_boxes) = track_memory
            if template_features.numel() > 0:
                return self.tracker(features, template_boxes, sr=sr, template_features=template_features)
        return {}, track_boxes, {}

    def get_track_memory(self, features, tracks):                        # :54-75
        assert (len(tracks) == 1)
        active_tracks = self._get_track_targets(tracks[0])
        if len(active_tracks) == 0:
            template_features = torch.tensor([], device=features[0].device)
            sr = copy.deepcopy(active_tracks)
            sr.size = [active_tracks.size[0] + self.track_utils.pad_pixels * 2,
                       active_tracks.size[1] + self.track_utils.pad_pixels * 2]
            track_memory = (template_features, [sr], [active_tracks])
        else:
            track_memory = self.tracker.extract_cache(features, active_tracks)
        track_memory = self._update_memory_with_dormant_track(track_memory)
        self.track_pool.update_cache(track_memory)
        return track_memory

    def _update_memory_with_dormant_track(self, track_memory):           # :77-97
        cache = self.track_pool.get_cache()
        if not cache or track_memory is None:
            return track_memory
        dormant_caches = []
        for dormant_id in self.track_pool.get_dormant_ids():
            if dormant_id in cache:
                dormant_caches.append(cache[dormant_id])
        cached_features = [x[0][None, ...] for x in dormant_caches]
        if track_memory[0] is None:
            if track_memory[1][0] or track_memory[2][0]:
                raise Exception("Unexpected cache state")
            track_memory = [[]] * 3
            buffer_feat = []
        else:
            buffer_feat = [track_memory[0]]
        features = torch.cat(buffer_feat + cached_features)
        sr = cat_boxlist(track_memory[1] + [x[1] for x in dormant_caches])
        boxes = cat_boxlist(track_memory[2] + [x[2] for x in dormant_caches])
        return features, [sr], [boxes]

    def _get_track_targets(self, target):                                # :99-110
        if len(target) == 0:
            return target
        active_ids = self.track_pool.get_active_ids()
        ids = target.get_field('ids').tolist()
        idxs = torch.zeros((len(ids),), dtype=torch.bool, device=target.bbox.device)
        for _i, _id in enumerate(ids):
            if _id in active_ids:
                idxs[_i] = True
        return target[idxs]


class ReferenceLoop(object):
    """roi_heads.py:22-52, inference, ``box`` = a pass-through for the detector's output (the plain golden sequence's switch
    module) and no refinement; exposes what tests/sequence_replay.py::replay reads of a ``TrackingLoop``."""

    boxlist_cls = BoxList

    def __init__(self, tracker, track_utils, thresholds, max_dormant_frames):
        pool = TrackPool(max_dormant_frames=max_dormant_frames)
        self.track = TrackHead(tracker, track_utils, pool)
        self.solver = TrackSolver(pool, *thresholds)
        self.refine_tracks = None
        self.track_memory = None

    def reset(self):
        self.track.track_pool.reset()
        self.track_memory = None

    @staticmethod
    def box(features, proposals, targets=None):
        """The ``box`` slot as the plain golden sequence fills it for PROPAGATED TRACKS (oracle/gen_golden_sequence.py
        BoxSwitch with no real box head): boxes unchanged, scores in the track band (+ 1)."""
        p = proposals[0]
        out = BoxList(p.bbox.clone(), p.size, mode=p.mode)
        for f in p.fields():
            out.add_field(f, p.get_field(f).clone())
        out.add_field("scores", p.get_field("scores") + 1.0)
        return features, [out], {}

    def _refine_tracks(self, features, tracks):                                              # roi_heads.py:60-84
        if len(tracks[0]) == 0:
            return tracks[0]
        track_scores = tracks[0].get_field('scores') + 1.
        _, tracks, _ = self.box(features, tracks)
        det_scores = tracks[0].get_field('scores')
        det_boxes = tracks[0].bbox
        scores = (det_scores + track_scores) / 2.
        boxes = det_boxes
        r_tracks = BoxList(boxes, image_size=tracks[0].size, mode=tracks[0].mode)
        r_tracks.add_field('scores', scores)
        r_tracks.add_field('ids', tracks[0].get_field('ids'))
        r_tracks.add_field('labels', tracks[0].get_field('labels'))
        return [r_tracks]

    def __call__(self, features, detections):
        with torch.no_grad():
            detections = [detections]                                                        # :25-35 (detector output passes)
            y, tracks, loss_track = self.track(features, None, None, self.track_memory)      # :38
            if tracks is not None:                                                           # :43-45
                tracks = self._refine_tracks(features, tracks)
                detections = [cat_boxlist(detections + tracks)]
            detections = self.solver(detections)                                             # :47
            self.track_memory = self.track.get_track_memory(features, detections)            # :50
        return detections[0]
