"""Inference glue around the EMM head: track memory, dormant tracks, the per-frame tracking step.

Mirrors ``TrackHead`` (reference siammot/modelling/track_head/track_head.py:8-126, inference branch) and the
tracking part of ``CombinedROIHeads.forward`` / ``SiamMOT.forward`` (modelling/roi_heads.py:22-52, rcnn.py:34-62).
The reference classes work unchanged with the HIP ``EMM`` module; this restatement exists so that
the whole tracker — head, solver, pool — can run from this repository with ONE host synchronisation per frame (the
reference writes one boolean per track into a device tensor, track_head.py:103-108).

``TrackingLoop`` is the detector-agnostic frame step: FPN features + this frame's detections in, tracked boxes out.
Per frame it enqueues the head (3 launches), [the box-head refinement of the propagated boxes (6 launches),] the
one-launch solver and the masked template extraction of the rows the solver leaves active — through the frame entry
point ``smot_track_frame_fwd`` in two calls on an argument block that stays packed between frames (``_step_native``, the
default) or composed from the per-stage bindings (``_step_lean``, ``loop.native_frame = False``); the general path covers
other solvers, CPU tensors and any ``refine_tracks`` callable — then synchronises once on the solver's record.  The
box-head refinement of the propagated boxes
(``_refine_tracks``, roi_heads.py:60-84) belongs to the detector: ``siammot_amd.box_refine.RefineTracks`` wraps any
box head with the reference's call signature.
"""
import copy

import numpy as np

import torch

from . import ops
from .emm import OrderHint
from .structures import BoxList, cat_boxlist


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

    def get_track_memory(self, features, tracks, precomputed=None):                # track_head.py:54-75
        """``precomputed``: capacity-sized ``(templates, sr boxes[, order hint, the boxes tensor they were made from])``
        of the solver's active rows, extracted by a launch that was enqueued before the frame's synchronisation
        (``EMM.extract_cache_rows``)."""
        assert len(tracks) == 1
        active_tracks = self._get_track_targets(tracks[0])
        if precomputed is not None and len(active_tracks) > 0:
            a = len(active_tracks)
            hint = None                   # the hint describes rows 0..a-1 of the tensor the extraction read
            if (len(precomputed) > 3 and precomputed[2] is not None
                    and precomputed[3].data_ptr() == active_tracks.bbox.data_ptr()):
                hint = precomputed[2][:a]
            track_memory = self.tracker.wrap_cache(precomputed[0][:a], precomputed[1][:a], active_tracks,
                                                   *(() if hint is None else (hint,)))
            track_memory = self._update_memory_with_dormant_track(track_memory)
            self.track_pool.update_cache(track_memory)
            return track_memory
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


class _LazyMemory(object):
    """The track memory ``(templates, [search-region BoxList], [template-box BoxList])`` as the tracking loop leaves it for
    the next frame — rows 0 .. n_act-1 the frame's active rows (the solver's and the masked extraction's own outputs), rows
    n_act .. A-1 the dormant tracks' rows, copied behind them on the device (``TrackingLoop._carry_dormant``) —: the tuple's
    six view tensors and two BoxLists are built on first access.  The next frame's head only needs five POINTERS into the
    solver's output buffers and the extraction's outputs (``pointers``) — building the views costs ~8 us of host time on the
    frame's serial chain and, frame after frame, nobody looks at them.  Indexing / iterating / ``len`` behave like the
    tuple."""
    __slots__ = ("_val", "fbuf", "ibuf", "templates", "sr_rows", "M", "A", "size", "sr_size", "host_ids", "cls", "hint_off",
                 "n_act", "dormant")

    def __init__(self, fbuf, ibuf, templates, sr_rows, M, A, size, sr_size, host_ids, cls, hint_off=0, n_act=None, dormant=()):
        self._val = None
        self.fbuf, self.ibuf, self.templates, self.sr_rows = fbuf, ibuf, templates, sr_rows
        self.M, self.A, self.size, self.sr_size, self.host_ids, self.cls = M, A, size, sr_size, host_ids, cls
        self.n_act = A if n_act is None else n_act          # rows 0 .. n_act-1 are active tracks, the rest dormant ones
        self.dormant = dormant                              # ids of rows n_act .. A-1, in row order
        # where in `fbuf` (in floats, behind the frame's float outputs) the order hint lives that the masked extraction wrote
        # for exactly rows 0 .. A-1, or 0: none.  An OFFSET, not an address: a deep copy / pickle of the loop
        # mid-video copies `fbuf` and the hint with it.  It needs no host object — and the head VERIFIES the list against the
        # rows it is given with (csrc/sr_xcorr.hip, fx_verify_hint), so a stale one is reported, not used.
        self.hint_off = hint_off

    @property
    def hint_ptr(self):
        return self.fbuf.data_ptr() + 4 * self.hint_off if self.hint_off else 0

    def hint_status(self):
        """Status word of this memory's order hint (0 = fine / no hint).  Synchronises: error paths and tests only."""
        if not self.hint_off:
            return 0
        w = self.hint_off + ops.HINT_STATUS_WORD
        return int(self.fbuf[w:w + 1].view(torch.int32).item())

    def clear_hint_status(self):
        """Zero the hint's status word on the current stream (after a speculative head that was launched on a guessed row
        count and discarded: its verification failed by construction)."""
        if self.hint_off:
            w = self.hint_off + ops.HINT_STATUS_WORD
            self.fbuf[w:w + 1].zero_()

    def pointers(self):
        """(template boxes, search regions, templates, ids, labels, order hint or 0) of rows 0 .. A-1 as device addresses."""
        fp, ip, M = self.fbuf.data_ptr(), self.ibuf.data_ptr(), self.M
        return (fp + 16 * M, self.sr_rows.data_ptr(), self.templates.data_ptr(), ip + 16 * M, ip + 24 * M,
                fp + 4 * self.hint_off if self.hint_off else 0)

    def carry_pointers(self):
        """(templates, boxes, search regions, ids, labels, scores) as device addresses: ``ops.memory_carry``'s order."""
        fp, ip, M = self.fbuf.data_ptr(), self.ibuf.data_ptr(), self.M
        return self.templates.data_ptr(), fp + 16 * M, self.sr_rows.data_ptr(), ip + 16 * M, ip + 24 * M, fp + 36 * M

    def _materialise(self):
        v = self._val
        if v is None:
            st, M, A = torch.as_strided, self.M, self.A
            fbuf, ibuf = self.fbuf, self.ibuf
            fields = {"ids": st(ibuf, (A,), (1,), 2 * M), "scores": st(fbuf, (A,), (1,), 9 * M),
                      "labels": st(ibuf, (A,), (1,), 3 * M)}
            act = BoxList._wrap(st(fbuf, (A, 4), (4, 1), 4 * M), self.size, "xyxy", fields)
            act.host_ids = self.host_ids.tolist()
            sr = BoxList._wrap(self.sr_rows.narrow(0, 0, A), self.sr_size, "xyxy", dict(fields))
            v = self._val = (self.templates.narrow(0, 0, A), [sr], [act])
        return v

    def __getitem__(self, k):
        return self._materialise()[k]

    def __iter__(self):
        return iter(self._materialise())

    def __len__(self):
        return 3


def _memory_rows(mem):
    """Number of rows of a track memory (0 for None / an empty memory) without building a lazy one."""
    if mem is None:
        return 0
    if type(mem) is _LazyMemory:
        return mem.A
    return len(mem[2][0]) if mem[0].numel() > 0 else 0


class _FramePlan(object):
    """What ``TrackingLoop._frame_plan`` keeps between frames (see there)."""
    __slots__ = ("dev", "emm", "solver", "pool", "refine", "g", "scales", "params", "hann", "rz", "rx", "C", "amodal",
                 "ws_need", "lib", "args", "box_params", "box_ptrs", "box_amodal", "image_wh", "a_pp", "state")


def _GLOBAL_HOOKS():
    m = torch.nn.modules.module
    return (m._global_forward_hooks or m._global_forward_pre_hooks or m._global_backward_hooks
            or m._global_backward_pre_hooks)


class TrackingLoop(torch.nn.Module):
    """One tracking step per frame: ``forward(features, detections) -> BoxList`` with track ids."""

    # The frame entry point runs head, refinement and solver behind two library calls; the only way to look at the head's
    # (or the box head's) raw output of a frame in between is from inside `_step_native`.  The product never does: the two
    # calls below are made only by a subclass that sets `_probed` (tests/sequence_replay.py::ProbedTrackingLoop — the
    # closed-loop replays record the raw rows there).  No attribute of a product object is consulted for a callback.
    _probed = False

    def _head_output_enqueued(self, boxes, scores):                  # pragma: no cover - subclasses only
        raise NotImplementedError

    def _refined_output_enqueued(self, boxes, scores, ids, labels):  # pragma: no cover - subclasses only
        raise NotImplementedError

    def __init__(self, track_head, solver, refine_tracks=None):
        super(TrackingLoop, self).__init__()
        self.track = track_head
        self.solver = solver
        self.refine_tracks = refine_tracks
        self.track_memory = None

    def reset(self):                                                               # rcnn.py:37-39
        self.track_memory = None
        self.track.reset_track_pool()

    def __getstate__(self):
        # per-frame caches (ctypes blocks, library handles, the trusted-memory marker) are rebuilt on demand: a copy or a
        # pickle of the loop carries none of them
        d = self.__dict__.copy()
        for k in ("_plan", "_lean_static", "_own_memory", "_spec_head", "_prev_n_trk", "_early_head", "_prev_K", "_hint_extra"):
            d.pop(k, None)
        return d

    def __setattr__(self, name, value):
        if name in ("track", "solver", "refine_tracks"):                 # what the per-frame caches below were built from
            self.__dict__.pop("_lean_static", None)
            self.__dict__.pop("_plan", None)
        super(TrackingLoop, self).__setattr__(name, value)

    def _lean_ok(self, detections):
        """The per-frame fast path applies: this repository's EMM head, no box-head refinement or one with a
        device-only form (``refine_raw``), the one-launch solver, device tensors."""
        st = self.__dict__.get("_lean_static")
        if st is None:                    # (submodule lookups go through nn.Module.__getattr__: once)
            emm, solver = self.track.tracker, self.solver
            st = self.__dict__["_lean_static"] = (emm, solver, hasattr(emm, "track_raw"),
                                                  getattr(solver, "_device_path", None) is not None,
                                                  ops.track_solve_max_boxes(), self.refine_tracks, {})
        emm, solver, has_raw, has_device_path, max_boxes, refine, refine_ok = st
        if not (has_raw and has_device_path):
            return False
        mem = self.track_memory
        n_mem = (mem.A if type(mem) is _LazyMemory else len(mem[2][0])) if mem is not None else 0
        if refine is not None and n_mem > 0:
            # (what a box head can do for n rows follows from its layer shapes: asked once per row count)
            r = refine_ok.get(n_mem)
            if r is None:
                ok = getattr(refine, "raw_ok", None)
                one = getattr(getattr(refine, "box", None), "one_call_ok", None)
                r = refine_ok[n_mem] = (ok is not None and bool(ok(n_mem)), one is not None and bool(one(n_mem)))
            # the verdict rests on mutable state (the post-processor's score threshold, where the weights live): re-read
            # the two cheap predicates every frame and drop the cached verdicts when they moved
            key = refine_ok.get("key")
            cur = getattr(refine, "raw_state_key", None)
            cur = cur() if cur is not None else None
            if key != cur:
                refine_ok.clear()
                refine_ok["key"] = cur
                ok = getattr(refine, "raw_ok", None)
                one = getattr(getattr(refine, "box", None), "one_call_ok", None)
                r = refine_ok[n_mem] = (ok is not None and bool(ok(n_mem)), one is not None and bool(one(n_mem)))
            if not r[0]:
                return False
        elif refine is not None and getattr(refine, "raw_ok", None) is None:
            return False
        if not detections.bbox.is_cuda or emm.rz not in (15, 7) or detections.mode != "xyxy":     # (the masked extraction's shapes)
            return False
        kernel_fields = solver._KERNEL_FIELDS
        for f in detections.fields():
            if f not in kernel_fields:
                return False                 # extra fields: the general path keeps them
        n = len(detections) + n_mem
        return solver.nms_mask_fn is ops.nms_keep_mask and 0 < n <= max_boxes

    def _step_lean(self, features, detections):
        """One frame with the minimum of host work between the launches: raw tensors into ``ops.emm_track``, the
        solver kernel on un-concatenated segments, the record's copy, the masked template extraction — all enqueued
        before the frame's ONE synchronisation, which waits for the record only (the extraction is still running
        when the host continues) — then views, the pool mirror and a lazy cache note.  Same results as the general path
        (tests/test_solver.py::test_lean_step_equals_general_path)."""
        emm, solver, pool = self.track.tracker, self.solver, self.solver.track_pool
        mem = self.track_memory
        trk = None
        bias = 1.0                                 # propagated boxes keep their matching score, moved to the (1, 2] band
        if mem is None:
            pool.reset()                                                           # track_head.py:39-40
        else:
            z, sr, tb = mem
            tb0 = tb[0]
            if z.numel() > 0:
                bb, conf = emm.track_raw(features, tb0.bbox, sr[0].bbox, z, tb0.size, sr[0])
                trk = (bb, conf, tb0.get_field("ids"), tb0.get_field("labels"))
                if self.refine_tracks is not None:                                 # roi_heads.py:43-44,60-84, on the device
                    trk = self.refine_tracks.refine_raw(features, bb, conf, trk[2], trk[3], tb0.size)
                    bias = 0.0                     # the refined scores are in the band already
        dev = detections.bbox.device
        state = pool.device_state(dev)
        ring = pool.host_record_ring(dev)
        fbuf, ibuf, rec_host, M = ops.track_solve(
            solver._segment(detections), trk, bias,
            (float(solver.track_thresh), float(solver.start_thresh), float(solver.resume_track_thresh)),
            float(solver.NMS_THRESH), int(pool._max_dormant_frames), state, pool.DEVICE_CAPACITY, host_record=ring)
        ring.record_event()                 # behind the solver: its record lands in pinned host memory, no copy command
        ob, ab, osc, asc = fbuf.split((4 * M, 4 * M, M, M))
        act_boxes = ab.view(M, 4)
        pre = emm.extract_cache_rows(features, act_boxes, state[4:5])              # runs while the host wakes up
        ahead = None
        if pre is not None and len(pre) == 2:
            ahead = self._carry_ahead(mem, fbuf, ibuf, pre[0], pre[1], M, state, dev, ops._stream(dev))
        ring.wait(rec_host)                                                        # the frame's one synchronisation
        return self._finish_frame(features, detections, rec_host, fbuf, ibuf, M, pre, carried_ahead=ahead)

    def _carry_ahead(self, mem, fbuf, ibuf, templates, sr_rows, M, state, dev, stream, lib=None):
        """Before the frame's record is read: the dormant rows of the memory this frame's head ran on, copied behind the
        rows the solver is leaving active — the first destination row read from the solver's count on the device — on the
        guess that the dormant tracks stay the ones they were (the usual frame).  ``_carry_dormant`` checks the guess
        against the record and copies again when it did not hold; the copy launch is off the frame's serial chain
        (record -> next head) either way.  Returns (source rows, guessed first destination row) or None."""
        if not (type(mem) is _LazyMemory and mem.A > mem.n_act and self.__dict__.get("device_carry", True)):
            return None
        rows = list(range(mem.n_act, mem.A))
        fp, ip = fbuf.data_ptr(), ibuf.data_ptr()
        ops.memory_carry(mem.carry_pointers(), mem.A, (templates.data_ptr(), fp + 16 * M, sr_rows.data_ptr(), ip + 16 * M,
                                                       ip + 24 * M, fp + 36 * M), M, rows, mem.n_act, templates[0].numel(),
                         dev, stream, dst_row0_dev=state.data_ptr() + 16, lib=lib)
        return rows, mem.n_act

    def _carry_dormant(self, prev, pool, fbuf, ibuf, templates, sr_rows, M, A, ahead=None):
        """The dormant tracks' rows of the next memory, copied on the device from the memory this frame's head ran on
        (``prev``) to rows A .. of the frame's output buffers: one launch instead of the reference's ``torch.cat`` +
        two ``cat_boxlist`` per frame (track_head.py:77-97; ``TrackHead._update_memory_with_dormant_track`` is the general
        form).  Which rows: the reference appends ``cache[id]`` for every id of ``get_dormant_ids()`` — in that set's
        iteration order — that has a cache entry; an entry never changes while its track is dormant, and ``prev`` holds
        it: as an active row when the track went dormant in this frame (``TrackPool._mirror`` made exactly that row its
        entry), as a carried row otherwise.  Returns (ids, source rows, copied ahead) or None: the general form applies
        (no previous memory this code can address, an entry that is not a row of it, more rows than one launch takes).
        ``ahead`` = (source rows, first destination row) of a copy that was enqueued before the record was read."""
        if prev is None or not self.__dict__.get("device_carry", True):
            return None
        order = list(pool.get_dormant_ids())                    # a fresh set of the dict's keys, as track_head.py:83 iterates
        lazy = type(prev) is _LazyMemory
        if lazy and prev.dormant == order:
            # the usual frame with dormant tracks: the same ones as in the last frame, rows n_act .. of the previous memory
            ids, rows = order, list(range(prev.n_act, prev.A))
        else:
            cache, pend = pool._cache, pool.__dict__.get("_pending")
            if lazy:
                prev_ids = prev.host_ids
            else:
                if prev[0].numel() == 0:
                    return None
                prev_ids = getattr(prev[2][0], "host_ids", None)
                if prev_ids is None:
                    return None
            idx = {int(t): i for i, t in enumerate(prev_ids)}
            pend_ids = idx if (pend is not None and pend[0] is prev) else (
                set(int(t) for t in pend[1]) if pend is not None else ())
            ids, rows = [], []
            for d in order:
                if d in cache or d in pend_ids:                                    # track_head.py:84-86
                    r = idx.get(d)
                    if r is None:
                        return None
                    ids.append(d)
                    rows.append(r)
        D = len(rows)
        if D == 0:
            return ids, rows, False
        if D > ops.MEMORY_CARRY_MAX_ROWS or A + D > M:
            return None
        if ahead is not None and ahead[0] == rows and ahead[1] == A:
            ops.MEMORY_CARRY["ahead_kept"] += 1
            return ids, rows, True
        if ahead is not None:
            ops.MEMORY_CARRY["ahead_redone"] += 1
        if lazy:
            src, n_prev = prev.carry_pointers(), prev.A
        else:
            z, sr0, tb0 = prev[0], prev[1][0], prev[2][0]
            n_prev = len(prev_ids)
            if not (tb0.has_field("ids") and tb0.has_field("labels") and tb0.has_field("scores")):
                return None
            t = (z, tb0.bbox, sr0.bbox, tb0.get_field("ids"), tb0.get_field("labels"), tb0.get_field("scores"))
            want = (torch.float32, torch.float32, torch.float32, torch.int64, torch.int64, torch.float32)
            for x, dt in zip(t, want):
                if not (x.is_cuda and x.device == fbuf.device and x.dtype is dt and x.is_contiguous() and x.shape[0] == n_prev):
                    return None
            if z[0].numel() != templates[0].numel() or tb0.bbox.shape[1] != 4 or sr0.bbox.shape[1] != 4:
                return None
            src = tuple(x.data_ptr() for x in t)
        fp, ip = fbuf.data_ptr(), ibuf.data_ptr()
        dev = fbuf.device
        ops.memory_carry(src, n_prev, (templates.data_ptr(), fp + 16 * M, sr_rows.data_ptr(), ip + 16 * M, ip + 24 * M,
                                       fp + 36 * M), M, rows, A, templates[0].numel(), dev, ops._stream(dev))
        return ids, rows, False

    def _finish_frame(self, features, detections, rec_host, fbuf, ibuf, M, pre, P=None, hint_off=0, carried_ahead=None,
                      pre_out=None):
        """After the record arrived: mirror the pool, slice the outputs, build the next track memory.  This is host work on
        the frame's serial chain: ten strided views straight off the two output buffers (no intermediate splits), the
        record through the ring's numpy view, BoxLists of this package's own class without re-validation."""
        if P is not None:
            emm, pool = P.emm, P.pool
        else:
            emm, pool = self.track.tracker, self.solver.track_pool
        ring = pool._ring
        view = ring.view(rec_host) if ring is not None and (rec_host is ring.bufs[0] or rec_host is ring.bufs[1]) else rec_host.numpy()
        if (pre_out is not None and view[6] == 4 and view[7] == M and view[0] == pre_out[0] and view[1] >= 1 and not pool._dormant_ids
                and not self.__dict__.get("_hint_extra") and pool._last_tables is not None and self.__dict__.get("lazy_memory", True)
                and pool.__dict__.get("mirror_skip", True)):
            # The steady frame, in as few bytecodes as it takes (this is the serial chain: the GPU waits for the next head):
            # the kernel says the id tables stand (record word 6 == 4: no overflow, no NaN score, nothing started /
            # suspended / resumed / expired), there are no dormant tracks, the output has the row count its views were
            # built for.  Same state as the general code below leaves: pool counters, output, the unbuilt next memory.
            K, A = pre_out[0], int(view[1])
            pool._max_id, pool._frame_idx = int(view[2]), int(view[3])
            out = pre_out[1]
            out.host_ids = view[8 + M:8 + M + K].copy()
            host_ids = view[8 + 2 * M:8 + 2 * M + A].copy()
            size = detections.size
            pad2 = emm.track_utils.pad_pixels * 2
            memory = _LazyMemory(fbuf, ibuf, pre[0], pre[1], M, A, size, [int(size[0] + pad2), int(size[1] + pad2)], host_ids,
                                 BoxList, hint_off if A >= 2 else 0, A, [])
            pool._pending = (memory, host_ids)                      # = pool.note_memory
            d = self.__dict__
            d["_carry_ahead_kept"] = False
            d["track_memory"] = d["_own_memory"] = memory
            return out
        rec = view[:8 + 4 * M + 3 * pool.DEVICE_CAPACITY].copy()
        K, A = int(rec[0]), int(rec[1])
        if int(rec[7]) != M:
            # the record's layout is sized by the row count the KERNEL saw: a mismatch means the argument block named another
            # count than this call allocated and parsed for — never parse such a record
            raise RuntimeError("siammot_amd: the solver ran on %d rows, this frame was laid out for %d" % (int(rec[7]), M))
        if rec[6] & 2:
            # a propagated track came in with a NaN score: what a head writes whose order hint failed the kernel's
            # verification (include/smot_emm.h, order_hint) — ask the hint; NaN features give NaN scores too and pass
            self._raise_on_bad_hint(self.__dict__.get("track_memory"))
        pool._mirror(rec, M)
        cls, size = detections.__class__, detections.size
        st = torch.as_strided
        # fbuf = out_boxes [4M] | act_boxes [4M] | out_scores [M] | act_scores [M]; ibuf = out_ids | out_labels | act_ids | act_labels
        own = cls is BoxList
        if not (pre_out is not None and pre_out[0] == K):
            ob, oi, osc, ol = (st(fbuf, (K, 4), (4, 1), 0), st(ibuf, (K,), (1,), 0), st(fbuf, (K,), (1,), 8 * M),
                               st(ibuf, (K,), (1,), M))
        if pre_out is not None and pre_out[0] == K:
            out = pre_out[1]              # the output's views were built before the record arrived, on the right row count
        elif own:
            out = BoxList._wrap(ob, size, "xyxy", {"ids": oi, "scores": osc, "labels": ol})
        else:
            out = cls(ob, size, mode="xyxy")
            out.add_field("ids", oi)
            out.add_field("scores", osc)
            out.add_field("labels", ol)
        out.host_ids = rec[8 + M:8 + M + K]
        hint = pre[2] if pre is not None and len(pre) > 2 else None
        self.__dict__["_carry_ahead_kept"] = False
        if own and A > 0 and pre is not None and hint is None and self.__dict__.get("lazy_memory", True):
            # the usual frame: the active rows — with the dormant tracks' rows copied behind them on the device — ARE the next
            # memory, left unbuilt (see _LazyMemory); `out.active_rows` (read by the general path's
            # TrackHead._get_track_targets only) is not needed either
            carried = ((), (), False)
            if pool._dormant_ids:
                carried = self._carry_dormant(self.__dict__.get("track_memory"), pool, fbuf, ibuf, pre[0], pre[1], M, A,
                                              carried_ahead)
            if carried is not None:
                dormant, D = carried[0], len(carried[0])
                host_ids = rec[8 + 2 * M:8 + 2 * M + A]
                if D:
                    host_ids = np.concatenate([host_ids, np.asarray(dormant, dtype=host_ids.dtype)])
                    self.__dict__["_carry_ahead_kept"] = carried[2]
                pad2 = emm.track_utils.pad_pixels * 2
                # the extraction's order hint describes the memory's rows when it ranked exactly them: the active rows plus
                # the dormant rows that were in place behind them when it ran (carried inside the solver's launch, and the
                # guess about them held: `carried[2]`) — `_hint_extra` is how many such rows it was told about
                hint_ok = A + D >= 2 and self.__dict__.get("_hint_extra", 0) == D and (D == 0 or carried[2])
                memory = _LazyMemory(fbuf, ibuf, pre[0], pre[1], M, A + D, size, [int(size[0] + pad2), int(size[1] + pad2)],
                                     host_ids, cls, hint_off if hint_ok else 0, A, list(dormant))
                pool.note_memory(memory, memory.host_ids)
                self.__dict__["track_memory"] = memory
                self.__dict__["_own_memory"] = memory
                return out
            ops.FALLBACKS["dormant_rows_on_the_host"] += 1
        ab, ai, asc, al = (st(fbuf, (A, 4), (4, 1), 4 * M), st(ibuf, (A,), (1,), 2 * M), st(fbuf, (A,), (1,), 9 * M),
                           st(ibuf, (A,), (1,), 3 * M))
        if own:
            act = BoxList._wrap(ab, size, "xyxy", {"ids": ai, "scores": asc, "labels": al})
        else:
            act = cls(ab, size, mode="xyxy")
            act.add_field("ids", ai)
            act.add_field("scores", asc)
            act.add_field("labels", al)
        act.host_ids = rec[8 + 2 * M:8 + 2 * M + A].tolist()
        out.active_rows = act
        if A == 0 or pre is None:
            # no active row (the reference's empty memory) or no masked kernel for this pooler shape: general code
            self.track_memory = self.track.get_track_memory(features, [out])
            return out
        # (the extraction ranked exactly the rows of `act` — the first A of act_boxes — for the next frame's head)
        hint = hint.narrow(0, 0, A) if hint is not None else None
        zv, srv = pre[0].narrow(0, 0, A), pre[1].narrow(0, 0, A)
        if own and hint is None:
            pad2 = emm.track_utils.pad_pixels * 2                                  # = EMM.wrap_cache
            sr = BoxList._wrap(srv, [int(size[0] + pad2), int(size[1] + pad2)], "xyxy", dict(act.extra_fields))
            memory = (zv, [sr], [act])
        else:
            memory = emm.wrap_cache(zv, srv, act, hint)
        if pool._dormant_ids:
            memory = self.track._update_memory_with_dormant_track(memory)
        pool.note_memory(memory, getattr(memory[2][0], "host_ids", act.host_ids))
        self.__dict__["track_memory"] = memory
        self.__dict__["_own_memory"] = memory          # built here from this package's own kernels' outputs: the next
        return out                                     # frame's head takes its tensors without re-checking them

    @staticmethod
    def _raise_on_bad_hint(mem):
        """``mem``: the track memory the frame's head ran on.  Raises when the order hint that went with it failed the head's
        verification (the head's rows are NaN then)."""
        status = 0
        if type(mem) is _LazyMemory:
            status = mem.hint_status()
        elif mem is not None:
            h = getattr(mem[1][0], "order_hint", None)
            if h is not None:
                status = ops.order_hint_status(h.data)
        if status:
            raise RuntimeError("siammot_amd.TrackingLoop: the order hint handed to this frame's head does not describe the "
                               "track memory's rows (status %d): the head's output is NaN.  The memory was edited, re-ordered "
                               "or replaced without dropping its hint" % status)

    # ---- the same frame through the frame entry point: two calls on a block that stays packed -------------------------
    def _native_ok(self, detections):
        """``smot_track_frame_fwd`` applies: the one-launch path's conditions, the 15x15 template pooler's masked kernel
        and — with refinement — a box head whose whole chain fits ``smot_box_refine_fwd``."""
        if not self.__dict__.get("native_frame", True):
            return False              # (``loop.native_frame = False``: the Python-composed form, _step_lean)
        P = self.__dict__.get("_plan")
        if P is None:
            emm = self.track.tracker
            fz = emm.feature_extractor.pooler_z
            if not (emm.rz in (15, 7) and fz.sampling_ratio == 2):
                return False
        st = self.__dict__.get("_lean_static")
        if st is not None and st[5] is not None:                         # with refinement: the box head's whole chain must
            geom = st[6].get("geom")                                     # fit smot_box_refine_fwd (asked by _lean_ok)
            if geom is None:
                # smot_frame_args carries ONE level geometry (scales, level count) and ONE clip pair for the head and the
                # refinement: the frame entry point applies only when the box head's pooler sits on the search pooler's
                # levels (the reference keeps MODEL.ROI_BOX_HEAD.POOLER_SCALES and MODEL.TRACK_HEAD.POOLER_SCALES apart) and
                # both stages clip alike (INPUT.AMODAL in the reference's cfg; RefineTracks accepts any box head).  Otherwise
                # the Python-composed form, which handles the two independently.
                emm, box = self.track.tracker, st[5].box
                bx = getattr(getattr(box, "feature_extractor", None), "pooler", None)
                pp_ = getattr(box, "post_processor", None)
                geom = st[6]["geom"] = bool(
                    bx is not None and pp_ is not None
                    and tuple(float(v) for v in bx.scales) == tuple(float(v) for v in emm.feature_extractor.pooler_x.scales)
                    and bool(pp_.amodal_inference) == bool(emm.amodal))
            if not geom:
                return False
            mem = self.track_memory
            n_mem = _memory_rows(mem)
            if n_mem > 0:
                r = st[6].get(n_mem)
                if r is None or not r[1]:
                    return False
        elif st is None and self.refine_tracks is not None:
            return False
        return True

    def _frame_plan(self, dev, features):
        """Everything of a frame's argument block that stands while the video's geometry and the model stand: packed once
        (``ops.FrameArgs``), re-packed when a feature shape, a weight pointer or the device changes."""
        P = self.__dict__.get("_plan")
        if P is not None and P.dev == dev and ops._geometry_refresh(P.g, features, dev):
            return P
        emm, solver, pool = self.track.tracker, self.solver, self.solver.track_pool
        tu = emm.track_utils
        fe, pr = emm.feature_extractor.pooler_x, emm.predictor
        P = _FramePlan()
        P.dev, P.emm, P.solver, P.pool, P.refine = dev, emm, solver, pool, self.refine_tracks
        P.g = g = ops._geometry(features, tuple(fe.scales), emm.pad_pixels, dev)       # validates (raises) and caches
        P.scales = tuple(fe.scales)
        P.params = pr.param_dict()
        P.hann = ops.hann_window((emm.rx - emm.rz + 1) * ops.UP_SCALE, dev)
        P.rz, P.rx, P.C, P.amodal = emm.rz, emm.rx, g.C, emm.amodal
        P.ws_need, P.lib = {}, ops.load_library()
        a = P.args = ops.FrameArgs()
        a.feats, a.heights, a.widths, a.pad_cells, a.scales, a.num_levels, a.C = g.a_fp, g.a_hs, g.a_ws, g.a_pc, g.a_sc, g.L, g.C
        a.hann = P.hann.data_ptr()
        a.rx, a.rz, a.sampling_ratio = emm.rx, emm.rz, fe.sampling_ratio
        a.gn_groups, a.gn_eps, a.up = pr.gn_groups, pr.gn_eps, ops.UP_SCALE
        a.use_centerness = 1 if emm.use_centerness else 0
        a.pad_pixels, a.one_minus_sigma, a.sigma = emm.pad_pixels, 1 - emm.sigma, emm.sigma
        a.nms_thresh, a.max_dormant_frames, a.pool_capacity = solver.NMS_THRESH, pool._max_dormant_frames, pool.DEVICE_CAPACITY
        a.search_expansion, a.min_search_wh = tu.search_expansion, tu.min_search_wh
        P.box_params, P.box_ptrs = None, None
        if self.refine_tracks is not None:
            box = self.refine_tracks.box
            fx, pp_, cs, bp = box.feature_extractor, box.post_processor, box.predictor.cls_score, box.predictor.bbox_pred
            P.box_params = (fx.fc6.weight, fx.fc6.bias, fx.fc7.weight, fx.fc7.bias, cs.weight, cs.bias, bp.weight, bp.bias)
            a.refine, a.tracktor = 1, 1 if self.refine_tracks.tracktor else 0
            a.box_pooled, a.box_sampling_ratio = fx.pooler.output_size[0], fx.pooler.sampling_ratio
            a.dim6, a.dim7, a.num_classes, a.reg_classes = fx.fc6.out_features, fx.fc7.out_features, cs.out_features, bp.out_features // 4
            bc = pp_.box_coder
            a.box_wx, a.box_wy, a.box_ww, a.box_wh = bc.weights
            a.box_xform_clip = bc.bbox_xform_clip
            P.box_amodal = bool(pp_.amodal_inference)
        P.image_wh = None
        P.a_pp = None
        P.state = None
        self.__dict__["_plan"] = P
        return P

    def _step_native(self, features, detections, next_features=None):
        """``_step_lean`` through ``smot_track_frame_fwd``: same kernels, same arguments, same single synchronisation, two
        binding calls per frame on an argument block that is packed once per video (``_frame_plan``) and of which only two
        short ranges are rewritten per call.  A frame is a serial chain — host work before the first launch, the GPU chain,
        the record, host bookkeeping — so the FIRST call (stage HEAD) goes out as soon as the head's nine pointers stand,
        and the detections' segment, the output buffers and the remaining stages' call are prepared while the head runs.
        (The first version of this path prepared all 80 fields before ONE call: the first kernel started 20 us later than
        in the Python-composed form and the frame was slower, 0.165 vs 0.14 ms.)

        ``next_features`` (optional: the NEXT frame's feature maps, complete on this stream — a streaming caller has them
        while this frame's solver runs): the next frame's head is enqueued behind this frame's extraction BEFORE the host
        waits for the record, on the guess that the number of tracks stays what it was; the host's record -> first-launch
        path (~30 us of Python) then runs while the GPU works on that head instead of leaving it idle.  The next call
        takes the head's output as it is when the guess held (same memory, same row count, same feature tensors, same
        parameters) and launches the head again when it did not — the speculative launch only ever wrote its own buffers,
        so results are the synchronous path's bit for bit either way."""
        dev = detections.bbox.device
        P = self._frame_plan(dev, features)
        a, emm, solver, pool = P.args, P.emm, P.solver, P.pool
        blk = ops._param_block(P.params)           # revalidated per frame (a dozen attribute reads): load_state_dict / .to()
        mem = self.track_memory
        if mem is None:
            pool.reset()                                                           # track_head.py:39-40
        size = detections.size
        repack = False
        if blk.a_pp != P.a_pp:
            a.predictor_params, P.a_pp, repack = blk.a_pp, blk.a_pp, True
        if size != P.image_wh:
            P.image_wh = size
            cw, ch = (0.0, 0.0) if P.amodal else (float(size[0]), float(size[1]))
            if P.box_params is not None and P.box_amodal:
                cw = ch = 0.0                   # (head and box head share the flag in the reference's cfg: INPUT.AMODAL)
            a.clip_w, a.clip_h, repack = cw, ch, True
        if P.box_params is not None:
            ptrs = tuple([p_.data_ptr() for p_ in P.box_params])
            if ptrs != P.box_ptrs:
                P.box_ptrs = ptrs
                a.fc6_w, a.fc6_b, a.fc7_w, a.fc7_b, a.cls_w, a.cls_b, a.reg_w, a.reg_b = ptrs
                repack = True
        state = pool.device_state(dev)
        if state is not P.state:
            P.state, a.pool_state, repack = state, state.data_ptr(), True
        if repack:
            a.pack()
        n_trk = 0
        tf = ti = None
        stream = ops._stream(dev)
        head_ptrs = None
        if type(mem) is _LazyMemory and mem._val is None:
            # the memory this loop left behind, untouched since: five addresses, no tensor is built
            n_trk = mem.A
            head_ptrs = mem.pointers()
        elif mem is not None and mem[0].numel() > 0:
            z, sr, tb = mem
            tb0, sr0 = tb[0], sr[0]
            tbb, srb = tb0.bbox, sr0.bbox
            n_trk = tbb.shape[0]
            ids_t, lab_t = tb0.get_field("ids"), tb0.get_field("labels")
            if mem is not self.__dict__.get("_own_memory"):      # a memory this loop did not build itself: full checks
                tbb, srb = ops._chk(tbb, "template boxes", (n_trk, 4)), ops._chk(srb, "sr", (n_trk, 4))
                z = ops._chk(z, "template_features", (n_trk, P.C, P.rz, P.rz))
                if not (ids_t.is_contiguous() and lab_t.is_contiguous() and ids_t.dtype is torch.int64 and lab_t.dtype is torch.int64):
                    ids_t, lab_t = ids_t.to(torch.int64).contiguous(), lab_t.to(torch.int64).contiguous()
                ops._same_device(dev, ("template boxes", tbb), ("sr", srb), ("template_features", z), ("ids", ids_t),
                                 ("labels", lab_t))
            hint = OrderHint.lookup(sr0, tb0.bbox, sr0.bbox, P.scales) if "order_hint" in sr0.__dict__ else None
            head_ptrs = (tbb.data_ptr(), srb.data_ptr(), z.data_ptr(), ids_t.data_ptr(), lab_t.data_ptr(),
                         hint.data_ptr() if hint is not None else 0)
        spec = self.__dict__.pop("_spec_head", None)
        # (usable only while the memory is the unbuilt lazy one: once somebody has looked at its tensors they may have been
        # edited in place, and the speculative head read what was there before)
        if spec is not None and not (head_ptrs is not None and not repack and spec[0] is mem and spec[1] == n_trk
                                     and type(mem) is _LazyMemory and mem._val is None
                                     and spec[2] is features and spec[3] == blk.a_pp):
            if spec[5] and spec[0] is mem:
                mem.clear_hint_status()                   # (see the other discard below)
            ops.SPECULATION["early_discarded" if spec[7] else "discarded"] += 1
            spec = None                                   # the guess did not hold: the head runs again, below
        tf_cap = n_trk                    # rows the head's output buffer `tf` is laid out for: boxes | scores | refined boxes | refined scores
        if spec is not None:
            tf, tf_cap = spec[4], spec[6]                 # the head of this frame has been running since the last call /
            need = P.ws_need.get(n_trk)                   # since this call's first line (early head: TrackingLoop.forward)
            if need is None:
                need = P.ws_need[n_trk] = (
                    int(P.lib.smot_emm_track_ws_floats(n_trk, P.C, P.rx, P.rz)),
                    int(P.lib.smot_box_refine_ws_floats(n_trk, P.C, a.box_pooled, a.dim6, a.dim7, a.num_classes, a.reg_classes))
                    if a.refine else 0)
            ops.SPECULATION["early_used" if spec[7] else "used"] += 1
        elif head_ptrs is not None:
            p_tbb, p_sr, p_z, p_ids, p_lab, p_hint = head_ptrs
            need = P.ws_need.get(n_trk)
            if need is None:
                need = P.ws_need[n_trk] = (
                    int(P.lib.smot_emm_track_ws_floats(n_trk, P.C, P.rx, P.rz)),
                    int(P.lib.smot_box_refine_ws_floats(n_trk, P.C, a.box_pooled, a.dim6, a.dim7, a.num_classes, a.reg_classes))
                    if a.refine else 0)
            tf = torch.empty((10 * n_trk,), dtype=torch.float32, device=dev)
            p = tf.data_ptr()
            if p_hint == 0 and 2 <= n_trk <= 256 and P.rz == 15:
                ops.FALLBACKS["unhinted_head"] += 1
            addr = a.poke_head((ops._workspace(dev, need[0], stream.value).data_ptr(), p_tbb, p_sr, p_z, p_hint, p_ids,
                                p_lab, p, p + 16 * n_trk), n_trk, ops.STAGE_HEAD)
            ops.track_frame_addr(P.lib, addr, dev, stream)                         # the head is running from here on
        if self._probed and n_trk > 0:              # (a test-only subclass: tests/sequence_replay.ProbedTrackingLoop)
            self._head_output_enqueued(tf[:4 * n_trk].view(n_trk, 4), tf[4 * tf_cap:4 * tf_cap + n_trk])
        # ---- while the head runs: detections, output buffers, the remaining stages --------------------------------------
        seg = solver._segment(detections)
        n_det = 0
        d0 = d1 = d2 = d3 = 0
        if seg is not None:
            db, dsc, did, dlab = seg
            ops._check_segment(db, dsc, did, dlab, dev)
            n_det = db.shape[0]
            d0, d1, d2, d3 = db.data_ptr(), dsc.data_ptr(), did.data_ptr(), (dlab.data_ptr() if dlab is not None else 0)
        M = n_det + n_trk
        ring = pool.host_record_ring(dev)
        # float outputs [10 M] and, behind them at a 32-byte boundary, the order hint [M, HINT_FLOATS] the masked extraction ranks
        # the NEXT frame's rois into (one wave of one extra workgroup of that launch; 2..256 rows) — handed to the next
        # head as a bare address when the memory stays the extraction's own output (_LazyMemory)
        hint_off = ((10 * M + 7) & ~7) if (2 <= M <= 256 and self.__dict__.get("loop_order_hint", True)) else 0
        fbuf = torch.empty((hint_off + ops.HINT_FLOATS * M,) if hint_off else (10 * M,), dtype=torch.float32, device=dev)
        ibuf = torch.empty((4 * M,), dtype=torch.int64, device=dev)
        templates = torch.empty((M, P.C, P.rz, P.rz), dtype=torch.float32, device=dev)
        sr_next = torch.empty((M, 4), dtype=torch.float32, device=dev)
        rec_host = ring.next()
        fp, ip = fbuf.data_ptr(), ibuf.data_ptr()
        rw = r0 = r1 = r2 = r3 = 0
        stages = ops.STAGE_SOLVE | ops.STAGE_EXTRACT
        if n_trk > 0:
            if a.refine:
                ti = torch.empty((2 * n_trk,), dtype=torch.int64, device=dev)
                p = tf.data_ptr()
                rw = ops._workspace(dev, need[1], ("refine", stream.value)).data_ptr()
                r0, r1, r2, r3 = p + 20 * tf_cap, p + 36 * tf_cap, ti.data_ptr(), ti.data_ptr() + 8 * n_trk
                stages |= ops.STAGE_REFINE
        # the dormant rows of the memory this frame's head ran on go behind the rows the solver leaves active, in the solver's
        # own launch (extra workgroups beside its one: no launch, no time on the chain) — on the guess that the dormant
        # tracks stay the ones they were and the active count what it was (the usual frame); _carry_dormant checks both
        # against the record and copies again, with the stand-alone kernel, when the guess did not hold
        carried_ahead = None
        cz = cb = cs = ci = cl = cc = 0
        carry = (0, 0, 0)
        if (type(mem) is _LazyMemory and mem.A > mem.n_act and self.__dict__.get("device_carry", True)
                and (P.C * P.rz * P.rz) % 4 == 0 and self.__dict__.get("carry_in_solver", True)):
            cz, cb, cs, ci, cl, cc = mem.carry_pointers()
            carry = (mem.n_act, mem.A - mem.n_act, mem.n_act)
            carried_ahead = (list(range(mem.n_act, mem.A)), mem.n_act)
            stages |= ops.STAGE_CARRY
            ops.MEMORY_CARRY["in_the_solver_launch"] += 1
        # rows behind the active ones that the extraction's order hint ranks as well (smot_track_frame_fwd: the rows carried
        # in the solver's launch stand in act_boxes when the extraction runs)
        self.__dict__["_hint_extra"] = carry[1] if (stages & ops.STAGE_CARRY) else 0
        addr = a.poke_rest((rw, r0, r1, r2, r3, d0, d1, d2, d3,
                            fp, fp + 32 * M, ip, ip + 8 * M,                       # out_boxes, out_scores, out_ids, out_labels
                            fp + 16 * M, ip + 16 * M, ip + 24 * M, fp + 36 * M,    # act_boxes, act_ids, act_labels, act_scores
                            rec_host.data_ptr(), templates.data_ptr(), sr_next.data_ptr(), (fp + 4 * hint_off) if hint_off else 0,
                            cz, cb, cs, ci, cl, cc),
                           stages, n_det, (solver.track_thresh, solver.start_thresh, solver.resume_track_thresh), carry, n_trk)
        ops.track_frame_addr(P.lib, addr, dev, stream)
        if self._probed and n_trk > 0 and a.refine:
            self._refined_output_enqueued(tf[5 * tf_cap:5 * tf_cap + 4 * n_trk].view(n_trk, 4), tf[9 * tf_cap:9 * tf_cap + n_trk],
                                          ti[:n_trk], ti[n_trk:])
        hint_ptr = (fp + 4 * hint_off) if hint_off else 0
        spec_tf = None
        if carried_ahead is None:
            carried_ahead = self._carry_ahead(mem, fbuf, ibuf, templates, sr_next, M, state, dev, stream, P.lib)
        # (a wrong guess costs the GPU a whole head: the guess is made only while the count has been holding — this frame
        # had as many tracks as the frame before)
        steady = self.__dict__.get("_prev_n_trk") == n_trk
        self.__dict__["_prev_n_trk"] = n_trk
        if (next_features is not None and steady and n_trk >= 1 and type(mem) is _LazyMemory and detections.__class__ is BoxList
                and self.__dict__.get("lazy_memory", True) and ops._geometry_refresh(P.g, next_features, dev)):
            # the next frame's head on this frame's outputs, guessing that n_trk rows stay active (the steady state: the
            # memory of this frame was the previous frame's active rows, untouched): rows 0 .. n_trk-1 of act_boxes / ids /
            # labels, of the extraction's search regions and templates, and its order hint
            need = P.ws_need[n_trk]
            # (... and that the dormant tracks stay the ones they were: their rows went behind the active rows just above)
            spec_tf = torch.empty((10 * n_trk,), dtype=torch.float32, device=dev)
            p = spec_tf.data_ptr()
            spec_hint = hint_ptr if (n_trk >= 2 and (carried_ahead is None or (stages & ops.STAGE_CARRY))) else 0
            addr = a.poke_head((ops._workspace(dev, need[0], stream.value).data_ptr(), fp + 16 * M, sr_next.data_ptr(),
                                templates.data_ptr(), spec_hint, ip + 16 * M,
                                ip + 24 * M, p, p + 16 * n_trk), n_trk, ops.STAGE_HEAD)
            ops.track_frame_addr(P.lib, addr, dev, stream)
            ops.SPECULATION["launched"] += 1
            ops._geometry_refresh(P.g, features, dev)     # the plan's pointer array names THIS frame's maps again
        early = None
        if (spec_tf is None and M >= 1 and detections.__class__ is BoxList and self.__dict__.get("early_head", True)
                and self.__dict__.get("lazy_memory", True)):
            # No speculative head: the NEXT call's head launch is prepared NOW, while the GPU works on this frame — its
            # output buffer, workspace and the nine pointers of the head's argument range (all of them addresses inside THIS
            # frame's output buffers: the next memory, if it stays the unbuilt lazy one) — so that the next call can enqueue
            # its head on its FIRST line (TrackingLoop.forward) and validate afterwards: what used to be ~12 us of host work
            # between the record and the first launch of the next frame — the GPU idle — runs beside the GPU here.  The row
            # count is this frame's, corrected after the record.
            nm = P.ws_need.get(M)
            if nm is None:
                nm = P.ws_need[M] = (
                    int(P.lib.smot_emm_track_ws_floats(M, P.C, P.rx, P.rz)),
                    int(P.lib.smot_box_refine_ws_floats(M, P.C, a.box_pooled, a.dim6, a.dim7, a.num_classes, a.reg_classes))
                    if a.refine else 0)
            e_tf = torch.empty((10 * M,), dtype=torch.float32, device=dev)
            p = e_tf.data_ptr()
            e_ws = ops._workspace(dev, nm[0], stream.value)
            a.poke_head((e_ws.data_ptr(), fp + 16 * M, sr_next.data_ptr(),
                         templates.data_ptr(), hint_ptr, ip + 16 * M, ip + 24 * M, p, p + 16 * M), max(n_trk, 1), ops.STAGE_HEAD)
            # (the launch happens in the NEXT call: everything its argument block names must stay alive until then — the
            # parameter block `blk` owns the host array of weight pointers the block's `predictor_params` points at, and
            # ops' caches may drop both it and the workspace tensor when other modules / loops come and go in between:
            # without these references the early launch read a freed pointer array, "predictor: null pointer" once in a
            # full test session)
            early = (e_tf, stream.value, blk.a_pp, (blk, getattr(blk, "pp", None), getattr(blk, "packed", None), tuple(getattr(blk, "tensors", ()))), e_ws)   # (a refresh of `blk`
            # replaces its pointer array and packed filters: the objects themselves are held, not just their owner)
        # the frame's output BoxList, built NOW on the guess that it has as many rows as the last frame's (four strided views:
        # ~6 us that would otherwise sit between the record and the return)
        pre_out = None
        Kg = self.__dict__.get("_prev_K")
        if Kg is not None and Kg <= M and detections.__class__ is BoxList:
            st = torch.as_strided
            pre_out = (Kg, BoxList._wrap(st(fbuf, (Kg, 4), (4, 1), 0), detections.size, "xyxy",
                                         {"ids": st(ibuf, (Kg,), (1,), 0), "scores": st(fbuf, (Kg,), (1,), 8 * M),
                                          "labels": st(ibuf, (Kg,), (1,), M)}))
        ring.wait(rec_host, event=False)                                           # the frame's one synchronisation
        out = self._finish_frame(features, detections, rec_host, fbuf, ibuf, M, (templates, sr_next), P, hint_off=hint_off,
                                 carried_ahead=carried_ahead, pre_out=pre_out)
        self.__dict__["_prev_K"] = len(out.bbox)
        if spec_tf is not None:
            # valid for exactly the memory _finish_frame just built, if it is the lazy one over these buffers with n_trk rows
            # — and its dormant rows, if any, are the ones that were copied before that head was enqueued
            m2 = self.__dict__.get("track_memory")
            if (type(m2) is _LazyMemory and m2.fbuf is fbuf and m2.A == n_trk
                    and (m2.A == m2.n_act or self.__dict__.get("_carry_ahead_kept"))
                    and (not spec_hint or m2.hint_off)):      # (a hint that turned out not to describe the rows: that head's output is NaN)
                self.__dict__["_spec_head"] = (m2, n_trk, next_features, P.a_pp, spec_tf, spec_hint, n_trk, False)
            else:
                ops.SPECULATION["discarded"] += 1        # the row count changed, or the dormant rows are not the ones copied ahead
                # the discarded head VERIFIED the extraction's hint against a row count that was not the frame's and raised
                # its status word: the head that runs again on the true rows must find it clear
                if spec_hint and type(m2) is _LazyMemory and m2.fbuf is fbuf:
                    m2.clear_hint_status()
        self.__dict__["_early_head"] = None
        if early is not None:
            m2 = self.__dict__.get("track_memory")
            if type(m2) is _LazyMemory and m2.fbuf is fbuf and m2.A >= 1:
                # the head's argument range stands as poked above; the record brought the row count (and whether the
                # extraction's hint describes all rows: not when dormant rows were carried behind them)
                use_hint = hint_ptr if m2.hint_off else 0
                if m2.A != max(n_trk, 1) or use_hint != hint_ptr:
                    a.fix_head(m2.A, use_hint)
                self.__dict__["_early_head"] = (m2, m2.A, early[0], early[1], early[2], use_hint, M, P, early[3], early[4])
        return out

    def __call__(self, features, detections, next_features=None):
        # nn.Module.__call__ is three frames and a dozen hook-table reads deep (~3 us on the frame's serial chain): with no
        # hook registered on this module the call goes straight to forward(); with one, through the usual machinery
        d = self.__dict__
        if d["_forward_hooks"] or d["_forward_pre_hooks"] or d["_backward_hooks"] or d["_backward_pre_hooks"] or _GLOBAL_HOOKS():
            return super(TrackingLoop, self).__call__(features, detections, next_features=next_features)
        return self.forward(features, detections, next_features)

    def forward(self, features, detections, next_features=None):
        # inference only: nothing here may record an autograd graph.  (A ``@torch.no_grad()`` decorator costs 2-7 us per
        # call — a context object, two switches of the grad mode — on the serial chain; callers run inference under
        # ``torch.no_grad()`` anyway, as the reference does: demo_inference.py:103, inferencer.py:56.)
        if torch.is_grad_enabled():
            with torch.no_grad():
                return self._forward(features, detections, next_features)
        return self._forward(features, detections, next_features)

    def _forward(self, features, detections, next_features=None):
        eh = self.__dict__.get("_early_head")
        if eh is not None:
            # the head of THIS frame, prepared by the last call while the GPU was busy (see _step_native): enqueued before
            # anything else — the checks that it is the right launch follow while it runs (the memory object untouched, the
            # same stream and device, the maps' geometry here; parameters and packing in _step_native, which launches the
            # head again when any of them moved: the early launch wrote its own output buffer only)
            self.__dict__["_early_head"] = None
            mem = self.__dict__.get("track_memory")
            P = eh[7]
            stream = ops._stream(P.dev)
            if (eh[0] is mem and mem._val is None and P is self.__dict__.get("_plan")
                    and stream.value == eh[3] and ops._geometry_refresh(P.g, features, P.dev)):
                ops.track_frame_addr(P.lib, P.args._addr, P.dev, stream)
                ops.SPECULATION["early_launched"] += 1
                self.__dict__["_spec_head"] = (mem, eh[1], features, eh[4], eh[2], eh[5], eh[6], True)
        if self._lean_ok(detections):
            if self._native_ok(detections):
                if next_features is None:
                    return self._step_native(features, detections)
                return self._step_native(features, detections, next_features=next_features)
            return self._step_lean(features, detections)
        ops.FALLBACKS["general_frame"] += 1
        _, tracks, _ = self.track(features, track_memory=self.track_memory)        # roi_heads.py:38
        fast = getattr(self.solver, "_device_path", None)
        if fast is not None and self.refine_tracks is None and fast(detections, tracks[0] if tracks else None):
            # one launch for merge + solver + pool + active rows: the propagated boxes stay a segment of their own
            # (no concatenation), their +1 score band is applied inside the kernel
            h = self.solver.solve_launch(detections, tracks[0] if tracks else None, track_score_bias=1.0)
            # the next frame's templates / search regions of the rows the solver leaves active: enqueued NOW, on
            # the capacity with the count still on the device — the GPU never waits for the host inside a frame
            rows = getattr(self.track.tracker, "extract_cache_rows", None)
            pre = rows(features, h.act_boxes, h.count) if rows is not None else None
            if pre is not None:
                pre = tuple(pre) + (h.act_boxes,)
            mem_before = self.track_memory
            out = self.solver.solve_finish(h)                                      # the frame's one synchronisation
            if out.nan_track_scores:
                # (on this path too a hint that failed the head's verification is REPORTED — the head's rows are
                # NaN and would otherwise just drop out of the tracking; NaN features give NaN scores too and pass)
                self._raise_on_bad_hint(mem_before)
            self.track_memory = self.track.get_track_memory(features, [out], precomputed=pre)
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
