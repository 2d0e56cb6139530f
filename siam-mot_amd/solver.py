"""Track solver and track pool with one host synchronisation per frame (second half).

Mirrors ``TrackSolver`` (reference siammot/modelling/track_head/track_solver.py:7-108) and ``TrackPool``
(track_head/track_utils.py:138-250).  The reference merges detections with the boxes propagated from active and
dormant tracks by score-banded NMS and then starts / suspends / resumes / expires track ids — correct, but written
with one device synchronisation per box (``int(x) in active_ids`` over a device tensor, ``.item()`` per cached
track, several ``.tolist()`` / ``.nonzero()``), which caps the frame rate once the tracker head takes 74 us.
Here the score banding and the NMS stay on the device (``ops.nms_keep_mask``: no sync), ids / scores / keep mask
cross to the host in ONE copy, the id bookkeeping runs on numpy arrays, and the results go back in one copy.

Semantics are the reference's, statement by statement (cited inline); tests/test_solver.py runs this module and a
literal restatement of the reference side by side on the same random frames.
"""
import numpy as np
import torch

from . import ops
from .structures import BoxList  # noqa: F401  (duck-typed: any BoxList with bbox / get_field / add_field / __getitem__)


class _CacheEntry(object):
    """(template feature, search region, box) of one track, sliced out of the frame's memory on first use.
    The reference slices three BoxLists / tensors per track per frame (track_utils.py:185-197) although only
    dormant tracks are ever read back (track_head.py:83-86)."""
    __slots__ = ("_src", "_idx", "_val")

    def __init__(self, features, sr, boxes, idx):
        self._src, self._idx, self._val = (features, sr, boxes), idx, None

    def _materialise(self):
        if self._val is None:
            features, sr, boxes = self._src
            i = self._idx
            self._val = (features[i] if len(features) > 0 else features, sr[i: i + 1], boxes[i: i + 1])
            self._src = None
        return self._val

    def __getitem__(self, k):
        return self._materialise()[k]

    def __iter__(self):
        return iter(self._materialise())

    def __len__(self):
        return 3


class TrackPool(object):
    """Track-id life cycle and per-track cache (track_utils.py:138-250), host-side state."""

    DEVICE_CAPACITY = 512           # ids per table of the device-resident state (csrc/track_solver.hip: TS_MAXM)

    def __init__(self, active_ids=None, max_entangle_length=10, max_dormant_frames=1):
        self._active_ids = set()
        self._dormant_ids = {}          # id -> frame index at which it was last active
        self._kill_ids = set()
        self._max_id = -1
        self._embedding = None
        self._cache = {}
        self._frame_idx = 0
        self._max_dormant_frames = max_dormant_frames
        self._max_entangle_length = max_entangle_length
        # The pool as the solver kernel sees it (ops.track_solve): an int32 tensor on the tracker's device, read
        # and rewritten by the kernel every frame.  These Python sets are its MIRROR, refreshed from the record the
        # kernel hands back; a mutation through the methods below marks the device copy stale instead.
        self._dev_state = None
        self._dev_stale = True
        self._pending = None            # (memory, host ids) of the last frame, not yet turned into cache entries

    # ---- device-resident state -------------------------------------------------------------------
    def device_state(self, device):
        """The int32 state tensor on ``device`` (layout: include/smot_emm.h, smot_track_solve_fwd), uploaded from the
        host mirror when the mirror was modified since the last kernel call."""
        cap = self.DEVICE_CAPACITY
        if self._dev_state is None or self._dev_state.device != device:
            self._dev_state = torch.zeros((8 + 3 * cap,), dtype=torch.int32, device=device)
            self._dev_stale = True
        if self._dev_stale:
            if len(self._active_ids) > cap or len(self._dormant_ids) > cap:
                raise RuntimeError("TrackPool: more than %d active / dormant ids — beyond the device-resident solver" % cap)
            h = np.zeros(8 + 3 * cap, dtype=np.int32)
            h[0], h[1], h[2], h[3] = self._max_id, self._frame_idx, len(self._active_ids), len(self._dormant_ids)
            h[8:8 + len(self._active_ids)] = sorted(self._active_ids)
            dorm = sorted(self._dormant_ids.items())
            h[8 + cap:8 + cap + len(dorm)] = [d[0] for d in dorm]
            h[8 + 2 * cap:8 + 2 * cap + len(dorm)] = [d[1] for d in dorm]
            self._dev_state.copy_(torch.from_numpy(h))
            self._dev_stale = False
        return self._dev_state

    def _mirror(self, rec, M):
        """Refresh the host mirror from the kernel's record (header + snapshot of the three tables)."""
        cap = self.DEVICE_CAPACITY
        na, nd = int(rec[4]), int(rec[5])
        base = 8 + 3 * M
        active = set(rec[base:base + na].tolist())
        dormant = dict(zip(rec[base + cap:base + cap + nd].tolist(), rec[base + 2 * cap:base + 2 * cap + nd].tolist()))
        newly_dormant = set(dormant) - set(self._dormant_ids)
        if newly_dormant and self.__dict__.get("_pending") is not None:
            self._flush_pending(only=newly_dormant)    # their entry = their row in the last memory they were active in
        gone = (self._active_ids | set(self._dormant_ids)) - active - set(dormant)
        for tid in gone:                              # expire_tracks: killed ids lose their cache entry
            self._cache.pop(tid, None)
        self._kill_ids |= gone
        self._active_ids, self._dormant_ids = active, dormant
        self._max_id, self._frame_idx = int(rec[2]), int(rec[3])
        if rec[6]:
            raise RuntimeError("TrackPool: the device-resident id tables overflowed (%d ids per table)" % cap)

    def suspend_track(self, track_id):
        if track_id not in self._active_ids:
            raise ValueError
        self._dev_stale = True
        self._active_ids.remove(track_id)
        self._dormant_ids[track_id] = self._frame_idx - 1

    def expire_tracks(self):
        self._dev_stale = True
        for track_id, last_active in list(self._dormant_ids.items()):
            if self._frame_idx - last_active >= self._max_dormant_frames:
                self._dormant_ids.pop(track_id)
                self._kill_ids.add(track_id)
                self._cache.pop(track_id, None)

    def increment_frame(self, value=1):
        self._dev_stale = True
        self._frame_idx += value

    def update_cache(self, cache):
        """Latest (template features, search region, box) per track id — the ids cross to the host in one copy
        (the reference calls ``.item()`` per track, track_utils.py:196)."""
        self._flush_pending()                               # an older, lazily noted memory goes in first
        template_features, sr, template_boxes = cache
        sr, template_boxes = sr[0], template_boxes[0]
        n = len(template_boxes)
        if n == 0:
            return
        ids = getattr(template_boxes, "host_ids", None)     # already on the host after the solver kernel's record
        if ids is None:
            ids = template_boxes.get_field("ids").tolist()
        if len(template_features) > 0:
            assert len(template_features) == len(sr)
        # a dormant track's entry must not be replaced by a lazy view of a memory that contains that very entry
        # (its row is re-appended every frame): materialise entries of tracks that are not active first
        active = self._active_ids
        for idx in range(n):
            tid = ids[idx]
            old = self._cache.get(tid)
            if tid not in active and old is not None:
                continue                                   # dormant: keep the entry it went dormant with
            self._cache[tid] = _CacheEntry(template_features, sr, template_boxes, idx)

    def resume_track(self, track_id):
        if track_id not in self._dormant_ids or track_id in self._active_ids:
            raise ValueError
        self._dev_stale = True
        self._active_ids.add(track_id)
        self._dormant_ids.pop(track_id)

    def kill_track(self, track_id):
        if track_id not in self._active_ids:
            raise ValueError
        self._dev_stale = True
        self._active_ids.remove(track_id)
        self._kill_ids.add(track_id)
        self._cache.pop(track_id, None)

    def start_track(self):
        self._dev_stale = True
        self._max_id += 1
        self._active_ids.add(self._max_id)
        return self._max_id

    def get_active_ids(self):
        return self._active_ids

    def get_dormant_ids(self):
        return set(self._dormant_ids.keys())

    def note_memory(self, cache, host_ids):
        """Lazy form of ``update_cache`` for the tracking loop's fast path: remember this frame's memory and the
        ids of its rows; entries are materialised only for the tracks that go dormant (their last active memory is
        this one) — the reference rebuilds an entry per track per frame although only dormant tracks are ever read
        back (track_head.py:83-86).  ``get_cache()`` still answers for every id in memory."""
        # (the previous frame's pending memory is simply dropped: tracks that went dormant since were materialised
        # from it by _mirror, tracks that are still active have a row in the new memory)
        self._pending = (cache, host_ids)

    def _flush_pending(self, only=None):
        pend = self.__dict__.get("_pending")
        if pend is None:
            return
        (template_features, sr, template_boxes), ids = pend
        sr, template_boxes = sr[0], template_boxes[0]
        active = self._active_ids
        for idx, tid in enumerate(ids):
            if only is not None and tid not in only:
                continue
            if tid not in active and tid in self._cache:
                continue                                   # dormant row re-appended from the cache: keep the original
            self._cache[tid] = _CacheEntry(template_features, sr, template_boxes, idx)
        if only is None:
            self._pending = None

    def get_cache(self):
        self._flush_pending()
        return self._cache

    def activate_tracks(self, track_id):
        self.resume_track(track_id)

    def reset(self):
        self.__init__(max_entangle_length=self._max_entangle_length, max_dormant_frames=self._max_dormant_frames)


class _Solve(object):
    """Handle between ``TrackSolver.solve_launch`` and ``solve_finish``."""
    __slots__ = ("ref", "M", "rec", "out_boxes", "act_boxes", "out_scores", "act_scores", "ibuf")

    def __init__(self, ref, M, rec, out_boxes, act_boxes, out_scores, act_scores, ibuf):
        self.ref, self.M, self.rec = ref, M, rec
        self.out_boxes, self.act_boxes, self.out_scores, self.act_scores, self.ibuf = \
            out_boxes, act_boxes, out_scores, act_scores, ibuf

    @property
    def count(self):
        return self.rec[1:2]                    # number of active rows, on the device


class TrackSolver(torch.nn.Module):
    """Drop-in for the reference ``TrackSolver``: ``forward([BoxList]) -> [BoxList]``.  ``nms_mask_fn(boxes_xyxy,
    scores, thresh) -> bool mask`` defaults to the HIP NMS kernel (device tensors only)."""

    NMS_THRESH = 0.5                                     # track_solver.py:22

    def __init__(self, track_pool, track_thresh=0.3, start_track_thresh=0.5, resume_track_thresh=0.4,
                 nms_mask_fn=None):
        super(TrackSolver, self).__init__()
        self.track_pool = track_pool
        self.track_thresh = track_thresh
        self.start_thresh = start_track_thresh
        self.resume_track_thresh = resume_track_thresh
        self.nms_mask_fn = nms_mask_fn or ops.nms_keep_mask

    def _device_path(self, *boxlists):
        """The one-launch kernel applies: default NMS, device tensors, few enough boxes."""
        if self.nms_mask_fn is not ops.nms_keep_mask or not hasattr(self.track_pool, "device_state"):
            return False
        n = 0
        for b in boxlists:
            if b is None:
                continue
            if not b.bbox.is_cuda:
                return False
            n += len(b)
        return 0 < n <= ops.track_solve_max_boxes()

    @staticmethod
    def _segment(b):
        if b is None or len(b) == 0:
            return None
        bx = b.convert("xyxy").bbox
        sc, ids = b.get_field("scores"), b.get_field("ids")
        lab = b.get_field("labels") if b.has_field("labels") else None
        return (bx if bx.is_contiguous() else bx.contiguous(), sc, ids, lab)

    @torch.no_grad()
    def solve_launch(self, detections, tracks=None, track_score_bias=0.0):
        """Enqueue one frame on the device-resident pool: ``detections`` (+ the boxes the tracker propagated,
        un-concatenated).  ONE kernel launch, NO synchronisation.  Returns a handle for ``solve_finish``; its
        ``act_boxes`` ([M,4], capacity) and ``count`` (device int32 view of the number of active rows) can feed a
        masked ``EMM.extract_cache`` right away."""
        pool = self.track_pool
        ref = detections if detections is not None and len(detections) else tracks
        dev = ref.bbox.device
        fbuf, ibuf, rec, M = ops.track_solve(
            self._segment(detections), self._segment(tracks), float(track_score_bias),
            (float(self.track_thresh), float(self.start_thresh), float(self.resume_track_thresh)),
            float(self.NMS_THRESH), int(pool._max_dormant_frames), pool.device_state(dev), pool.DEVICE_CAPACITY)
        ob, ab, osc, asc = fbuf.split((4 * M, 4 * M, M, M))
        return _Solve(ref, M, rec, ob.view(M, 4), ab.view(M, 4), osc, asc, ibuf)

    def solve_finish(self, h):
        """Synchronise once, mirror the pool, slice the outputs.  The result carries ``host_ids`` (numpy) and
        ``active_rows`` (boxes / ids / labels / scores of the rows whose id is active now) so that ``TrackHead``
        builds the next track memory without touching the device again."""
        rec = ops.track_solve_record(h.rec)
        M, ref = h.M, h.ref
        K, A = int(rec[0]), int(rec[1])
        self.track_pool._mirror(rec, M)
        oi, ol, ai, al = h.ibuf.split((M, M, M, M))
        out = ref.__class__(h.out_boxes[:K], ref.size, mode="xyxy")
        if ref.mode != "xyxy":
            out = out.convert(ref.mode)
        out.add_field("ids", oi[:K])
        out.add_field("scores", h.out_scores[:K])
        out.add_field("labels", ol[:K])
        out.host_ids = rec[8 + M:8 + M + K].astype(np.int64)
        act = ref.__class__(h.act_boxes[:A], ref.size, mode="xyxy")
        act.add_field("ids", ai[:A])
        act.add_field("scores", h.act_scores[:A])
        act.add_field("labels", al[:A])
        act.host_ids = rec[8 + 2 * M:8 + 2 * M + A].tolist()
        out.active_rows = act
        return out

    def solve(self, detections, tracks=None, track_score_bias=0.0):
        return self.solve_finish(self.solve_launch(detections, tracks, track_score_bias))

    @torch.no_grad()
    def forward(self, detection):
        assert len(detection) == 1                        # :50
        detection = detection[0]
        if len(detection) == 0:
            return [detection]
        if self._device_path(detection):
            return [self.solve(detection)]
        pool = self.track_pool
        all_ids = detection.get_field("ids")
        all_scores = detection.get_field("scores")
        device = all_ids.device
        dormant_ids = pool.get_dormant_ids()              # snapshot before any state change (:60)

        # active tracks move one score band up, in place on the input field as in the reference (:63-69)
        active = sorted(pool.get_active_ids())
        if active:
            act = torch.tensor(active, dtype=all_ids.dtype, device=device)
            # N x A comparison: two tiny kernels (torch.isin sorts and costs ~170 us at these sizes)
            all_scores += (all_ids[:, None] == act[None, :]).any(dim=1).to(all_scores.dtype)
        keep_mask = self.nms_mask_fn(detection.convert("xyxy").bbox, all_scores, self.NMS_THRESH)     # :71, :22

        # ---- the one device -> host copy: ids, banded scores, keep mask -----------------------------
        host = torch.stack((all_ids.to(torch.float64), all_scores.to(torch.float64), keep_mask.to(torch.float64)),
                           dim=1).cpu().numpy()
        ids_all = host[:, 0].astype(np.int64)
        keep_idx = np.nonzero(host[:, 2] > 0.5)[0]        # ascending original order, as boxlist_nms returns (:22)
        _ids = ids_all[keep_idx].copy()
        _scores = host[keep_idx, 1].astype(np.float32)
        # back to the [0, 1] range (:30-31), fp32 arithmetic as the reference's tensor ops
        ge2 = _scores >= np.float32(2.0)
        _scores[ge2] = _scores[ge2] - np.float32(2.0)
        ge1 = _scores >= np.float32(1.0)
        _scores[ge1] = _scores[ge1] - np.float32(1.0)

        start_rows = np.nonzero((_ids < 0) & (_scores >= np.float32(self.start_thresh)))[0]          # :78
        inactive_rows = (_ids >= 0) & (_scores < np.float32(self.track_thresh))                       # :81
        nms_track_ids = set(_ids[_ids >= 0].tolist())
        all_track_ids = set(ids_all[ids_all >= 0].tolist())
        inactive_ids = set(_ids[inactive_rows].tolist()) | (all_track_ids - nms_track_ids)           # :82-86
        if dormant_ids:
            dormant_rows = np.isin(_ids, np.fromiter(dormant_ids, dtype=np.int64, count=len(dormant_ids)))
        else:
            dormant_rows = np.zeros(len(_ids), dtype=bool)
        for _id in _ids[dormant_rows & (_scores >= np.float32(self.resume_track_thresh))].tolist():  # :89-92
            pool.resume_track(_id)
        for r in start_rows.tolist():                                                                  # :94-95
            _ids[r] = pool.start_track()
        active_ids = pool.get_active_ids()
        for _id in inactive_ids:                                                                       # :97-100
            if _id in active_ids:
                pool.suspend_track(_id)
        _ids[inactive_rows] = -1                                                                       # :103
        pool.expire_tracks()
        pool.increment_frame()

        # ---- one host -> device copy: kept rows, new ids, rescaled scores ----------------------------
        back = torch.from_numpy(np.stack((keep_idx.astype(np.float64), _ids.astype(np.float64),
                                          _scores.astype(np.float64)), axis=1)).to(device, non_blocking=True)
        out = detection[back[:, 0].to(torch.int64)]
        out.add_field("ids", back[:, 1].to(all_ids.dtype))
        out.add_field("scores", back[:, 2].to(all_scores.dtype))
        out.host_ids = _ids                      # the ids are already on the host: TrackHead filters without a sync
        return [out]


def builder_tracker_solver(cfg, track_pool):
    """Same factory signature as the reference (track_solver.py:111-115)."""
    th = cfg.MODEL.TRACK_HEAD
    return TrackSolver(track_pool, th.TRACK_THRESH, th.START_TRACK_THRESH, th.RESUME_TRACK_THRESH)
