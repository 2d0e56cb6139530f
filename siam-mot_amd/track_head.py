"""Inference glue around the EMM head: track memory, dormant tracks, the per-frame tracking step.

Mirrors ``TrackHead`` (reference siammot/modelling/track_head/track_head.py:8-126, inference branch) and the
tracking part of ``CombinedROIHeads.forward`` / ``SiamMOT.forward`` (modelling/roi_heads.py:22-52, rcnn.py:34-62).
The reference classes work unchanged with the HIP ``EMM`` module; this restatement exists so that
the whole tracker — head, solver, pool — can run from this repository with a handful of host synchronisations per
frame (the reference writes one boolean per track into a device tensor, track_head.py:103-108).

``TrackingLoop`` is the detector-agnostic frame step: FPN features + this frame's detections in, tracked boxes out.
The box-head refinement of the propagated boxes (``_refine_tracks``, roi_heads.py:60-84) belongs to the detector and
is an optional callable.
"""
import copy

import torch

from .structures import cat_boxlist


class TrackHead(torch.nn.Module):
    def __init__(self, tracker, track_utils, track_pool):
        super(TrackHead, self).__init__()
        self.tracker = tracker
        self.track_utils = track_utils
        self.track_pool = track_pool

    def forward(self, features, proposals=None, targets=None, track_memory=None):
        if self.training:
            raise NotImplementedError("siammot_amd.TrackHead is an inference path (track_head.py:24-35 is training)")
        return self.forward_inference(features, track_memory)

    def forward_inference(self, features, track_memory=None):                      # track_head.py:37-46
        if track_memory is None:
            self.track_pool.reset()
        else:
            template_features, sr, template_boxes = track_memory
            if template_features.numel() > 0:
                return self.tracker(features, template_boxes, sr=sr, template_features=template_features)
        return {}, None, {}

    def reset_track_pool(self):
        self.track_pool.reset()

    def get_track_memory(self, features, tracks):                                  # track_head.py:54-75
        assert len(tracks) == 1
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

    def _update_memory_with_dormant_track(self, track_memory):                     # track_head.py:77-98
        cache = self.track_pool.get_cache()
        if not cache or track_memory is None:
            return track_memory
        dormant = [i for i in self.track_pool.get_dormant_ids() if i in cache]
        dormant_caches = [cache[i] for i in dormant]
        if not dormant_caches:
            return track_memory
        cached_features = [x[0][None, ...] for x in dormant_caches]
        feats = track_memory[0]
        buffer_feat = [feats] if feats.numel() > 0 else []
        features = torch.cat(buffer_feat + cached_features)
        sr = cat_boxlist(list(track_memory[1]) + [x[1] for x in dormant_caches])
        boxes = cat_boxlist(list(track_memory[2]) + [x[2] for x in dormant_caches])
        head_ids = getattr(track_memory[2][0], "host_ids", None)
        if head_ids is not None:                               # keep the ids on the host for update_cache
            boxes.host_ids = list(head_ids) + dormant
        return features, [sr], [boxes]

    def _get_track_targets(self, target):                                          # track_head.py:100-110
        if len(target) == 0:
            return target
        rows = getattr(target, "active_rows", None)            # left by the one-launch solver: already filtered
        if rows is not None:
            return rows
        active = self.track_pool.get_active_ids()
        ids = target.get_field("ids")
        host_ids = getattr(target, "host_ids", None)          # left by siammot_amd.solver.TrackSolver
        if host_ids is None:
            host_ids = ids.tolist()                            # one copy (the reference loops over a device tensor)
        rows = [i for i, t in enumerate(host_ids) if int(t) in active]
        return target[torch.tensor(rows, dtype=torch.int64, device=ids.device)]


class TrackingLoop(torch.nn.Module):
    """One tracking step per frame: ``forward(features, detections) -> BoxList`` with track ids."""

    def __init__(self, track_head, solver, refine_tracks=None):
        super(TrackingLoop, self).__init__()
        self.track = track_head
        self.solver = solver
        self.refine_tracks = refine_tracks
        self.track_memory = None

    def reset(self):                                                               # rcnn.py:37-39
        self.track_memory = None
        self.track.reset_track_pool()

    @torch.no_grad()
    def forward(self, features, detections):
        _, tracks, _ = self.track(features, track_memory=self.track_memory)        # roi_heads.py:38
        fast = getattr(self.solver, "_device_path", None)
        if fast is not None and self.refine_tracks is None and fast(detections, tracks[0] if tracks else None):
            # one launch for merge + solver + pool + active rows: the propagated boxes stay a segment of their own
            # (no concatenation), their +1 score band is applied inside the kernel
            out = self.solver.solve(detections, tracks[0] if tracks else None, track_score_bias=1.0)
            self.track_memory = self.track.get_track_memory(features, [out])
            return out
        dets = [detections]
        if tracks is not None:                                                     # roi_heads.py:43-45
            if self.refine_tracks is not None:
                tracks = self.refine_tracks(features, tracks)
            else:
                t = tracks[0]
                # without a box head the propagated boxes keep their matching score, moved to the (1, 2] band
                t.add_field("scores", t.get_field("scores") + 1.0)
            dets = [cat_boxlist(dets + list(tracks))]
        dets = self.solver(dets)                                                   # roi_heads.py:47
        self.track_memory = self.track.get_track_memory(features, dets)            # roi_heads.py:50, rcnn.py:52
        return dets[0]


def build_tracking_loop(cfg, device="cuda", refine_tracks=None):
    """EMM head + track utils + pool + solver from a config (roi_heads.py:87-101).

    ``refine_tracks(features, [tracks]) -> [tracks]`` is the detector's half of the step (``_refine_tracks``,
    roi_heads.py:60-84: the propagated boxes go through the box head as proposals and come back with the score
    ``(det + track)/2`` in the (1, 2] band).  Without it the propagated boxes keep their matching score + 1 and are
    not refined: the solver's thresholds then see different numbers than in the reference, so ids can differ from
    the reference's on real video — a warning says so once."""
    if refine_tracks is False:            # explicit: a detector-less loop (benchmarks, synthetic streams)
        refine_tracks = None
    elif refine_tracks is None:
        import warnings
        warnings.warn("siammot_amd.build_tracking_loop: no refine_tracks callable — propagated boxes are scored "
                      "track_conf + 1 instead of the reference's box-head refinement (roi_heads.py:60-84); "
                      "reference-equivalent tracking needs the detector's box head", stacklevel=2)
    from .emm import EMM
    from .solver import TrackPool, builder_tracker_solver
    from .track_utils import build_track_utils
    tu = build_track_utils(cfg)
    pool = TrackPool(max_dormant_frames=cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES)
    emm = EMM(cfg, tu).to(device).eval()
    head = TrackHead(emm, tu, pool).eval()
    return TrackingLoop(head, builder_tracker_solver(cfg, pool), refine_tracks).eval()
