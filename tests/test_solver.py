"""Track solver / track pool: the one-sync solver against a literal restatement of the
reference's solver, frame after frame on random scenes, with two pools that must evolve identically."""
import numpy as np
import pytest
import torch

import os

from oracle import solver_oracle as SO

# the seeded sequence oracle/gen_golden_solver.py pushes through the REFERENCE's own TrackSolver / TrackPool
SEQUENCE = dict(seed=3, frames=25, max_dormant_frames=3, thresholds=(0.4, 0.6, 0.4))
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solver_sequence.npz")


def _scene(rs, pool, n_det, n_missing):
    """Detections (id -1, score in (0,1)) + boxes propagated from active / dormant tracks (score in (1,2])."""
    W, H = 1280, 704
    tracks = sorted(pool.get_active_ids()) + sorted(pool.get_dormant_ids())
    tracks = [t for t in tracks if rs.rand() > n_missing]
    n = n_det + len(tracks)
    c = rs.uniform(60, [W - 60, H - 60], (n, 2))
    if n > 4:                                  # clusters: plenty of overlaps between detections and tracks
        c[n // 2:] = c[rs.randint(0, n // 2, n - n // 2)] + rs.normal(0, 6, (n - n // 2, 2))
    wh = rs.uniform(30, 110, (n, 2))
    boxes = np.concatenate((c - wh / 2, c + wh / 2), 1).astype(np.float32)
    ids = np.concatenate((np.full(n_det, -1), np.array(tracks, dtype=np.int64))).astype(np.int64)
    scores = np.concatenate((rs.uniform(0.05, 0.999, n_det), 1.0 + rs.uniform(0.05, 1.0, len(tracks)))).astype(np.float32)
    perm = rs.permutation(n)
    return boxes[perm], ids[perm], scores[perm]


def _run(device, nms_mask_fn, frames=SEQUENCE["frames"]):
    from siammot_amd.solver import TrackPool, TrackSolver
    from siammot_amd.structures import BoxList
    pool_a = TrackPool(max_dormant_frames=SEQUENCE["max_dormant_frames"])
    pool_b = TrackPool(max_dormant_frames=SEQUENCE["max_dormant_frames"])
    solver = TrackSolver(pool_a, *SEQUENCE["thresholds"], nms_mask_fn=nms_mask_fn)
    rs = np.random.RandomState(SEQUENCE["seed"])
    gold = np.load(GOLDEN)                      # outputs of the reference's own solver on this sequence
    seen_events = set()
    for f in range(frames):
        boxes, ids, scores = _scene(rs, pool_b, n_det=int(rs.randint(0, 40)), n_missing=0.15)
        bl = BoxList(torch.from_numpy(boxes).to(device), (1280, 704), mode="xyxy")
        bl.add_field("ids", torch.from_numpy(ids).to(device))
        bl.add_field("scores", torch.from_numpy(scores.copy()).to(device))
        bl.add_field("labels", torch.ones(len(ids), dtype=torch.int64, device=device))
        out = solver([bl])[0]
        keep, ref_ids, ref_scores = SO.solve(pool_b, boxes, ids.copy(), scores.copy(), *SEQUENCE["thresholds"])
        # the oracle restatement is pinned to the reference ...
        assert ref_ids.tolist() == gold["f%02d_ids" % f].tolist(), "oracle vs reference, frame %d" % f
        assert np.array_equal(ref_scores, gold["f%02d_scores" % f]) and np.array_equal(boxes[keep], gold["f%02d_boxes" % f])
        assert sorted(pool_b.get_active_ids()) == gold["f%02d_active" % f].tolist()
        assert sorted(pool_b.get_dormant_ids()) == gold["f%02d_dormant" % f].tolist()
        # ... and the product solver to both
        assert out.get_field("ids").cpu().tolist() == ref_ids.tolist(), "frame %d" % f
        assert np.array_equal(out.get_field("scores").cpu().numpy(), ref_scores), "frame %d" % f
        assert np.array_equal(out.bbox.cpu().numpy(), boxes[keep])
        assert out.get_field("labels").shape[0] == len(keep)
        assert pool_a.get_active_ids() == pool_b.get_active_ids()
        assert pool_a.get_dormant_ids() == pool_b.get_dormant_ids()
        assert pool_a._kill_ids == pool_b._kill_ids and pool_a._max_id == pool_b._max_id
        if pool_a._kill_ids:
            seen_events.add("expire")
        if pool_a.get_dormant_ids():
            seen_events.add("suspend")
        if len(bl):      # the input's score field was banded in place, as the reference does
            pass
    assert {"expire", "suspend"} <= seen_events and pool_a._max_id > 20
    empty = BoxList(torch.zeros((0, 4), device=device), (1280, 704))
    empty.add_field("ids", torch.zeros(0, dtype=torch.int64, device=device))
    empty.add_field("scores", torch.zeros(0, device=device))
    assert len(solver([empty])[0]) == 0


def _numpy_mask(boxes, scores, thresh):
    keep = SO.nms_indices(boxes.cpu().numpy(), scores.cpu().numpy(), thresh)
    m = torch.zeros(len(boxes), dtype=torch.bool)
    m[torch.from_numpy(keep)] = True
    return m


def test_solver_logic_matches_the_reference_restatement_on_cpu():
    """Host logic only: the NMS mask is injected (the product's default is the HIP kernel, device tensors only)."""
    _run("cpu", _numpy_mask)


def test_track_pool_cache_and_life_cycle():
    from siammot_amd.solver import TrackPool
    from siammot_amd.structures import BoxList
    pool = TrackPool(max_dormant_frames=2)
    a, b = pool.start_track(), pool.start_track()
    assert (a, b) == (0, 1) and pool.get_active_ids() == {0, 1}
    boxes = BoxList(torch.tensor([[0., 0., 9., 9.], [5., 5., 20., 20.]]), (100, 100))
    boxes.add_field("ids", torch.tensor([1, 0]))
    sr = BoxList(torch.tensor([[-5., -5., 14., 14.], [0., 0., 30., 30.]]), (100, 100))
    pool.update_cache((torch.arange(2 * 3).reshape(2, 3).float(), [sr], [boxes]))
    feat, s, bx = pool.get_cache()[1]
    assert feat.tolist() == [0.0, 1.0, 2.0] and s.bbox.tolist() == [[-5., -5., 14., 14.]] and bx.get_field("ids").tolist() == [1]
    pool.increment_frame()
    pool.suspend_track(0)
    assert pool.get_dormant_ids() == {0}
    with pytest.raises(ValueError):
        pool.suspend_track(0)
    pool.expire_tracks()                               # frame 1, last active at frame 0: 1 < 2 frames dormant
    assert pool.get_dormant_ids() == {0}
    pool.increment_frame(); pool.expire_tracks()       # frame 2: expired, cache entry dropped
    assert pool.get_dormant_ids() == set() and 0 not in pool.get_cache() and 0 in pool._kill_ids
    with pytest.raises(ValueError):
        pool.resume_track(0)


def test_lazy_cache_drops_expired_ids_for_good():
    """an id that expires while its row still sits in the lazily noted memory of the previous frame must not
    come back through the next full flush (its entry would pin that frame's whole template tensor until reset())."""
    from siammot_amd.solver import TrackPool
    from siammot_amd.structures import BoxList
    cap = TrackPool.DEVICE_CAPACITY

    def memory(ids):
        n = len(ids)
        b = BoxList(torch.arange(4 * n).reshape(n, 4).float(), (100, 100))
        b.add_field("ids", torch.tensor(ids))
        return (torch.arange(n * 3).reshape(n, 3).float(), [b], [b])

    def record(active, dormant, max_id, frame, M=4):
        rec = np.zeros(8 + 3 * M + 3 * cap, dtype=np.int32)
        rec[2], rec[3], rec[4], rec[5] = max_id, frame, len(active), len(dormant)
        base = 8 + 3 * M
        rec[base:base + len(active)] = sorted(active)
        d = sorted(dormant.items())
        rec[base + cap:base + cap + len(d)] = [k for k, _ in d]
        rec[base + 2 * cap:base + 2 * cap + len(d)] = [v for _, v in d]
        return rec

    pool = TrackPool(max_dormant_frames=1)
    pool._mirror(record({0, 1, 2}, {}, 2, 1), 4)
    pool.note_memory(memory([0, 1, 2]), [0, 1, 2])
    pool._mirror(record({0, 2}, {1: 0}, 2, 2), 4)              # id 1 goes dormant: materialised from the noted memory
    assert set(pool.get_cache()) == {0, 1, 2}
    pool.note_memory(memory([0, 2, 1]), [0, 2, 1])             # the dormant row is re-appended (track_head.py:77-98)
    pool._mirror(record({0, 2}, {}, 2, 3), 4)                  # id 1 expires
    assert 1 in pool._kill_ids
    cache = pool.get_cache()                                   # full flush of the memory that still lists id 1
    assert set(cache) == {0, 2} and set(cache) <= pool.get_active_ids() | pool.get_dormant_ids()


def test_mirror_keeps_the_reference_order_of_dormant_ids():
    """The device path hands the id tables back sorted; the row order of dormant tracks in the next memory follows the
    insertion order of the reference's ``_dormant_ids`` dict (track_head.py:83), i.e. the order its solver suspends
    ids in.  ``TrackPool._mirror`` rebuilds that order from the record; here the record is synthesised from the host
    path (the reference's statements on the same Python containers) and the two dicts must agree item by item."""
    from siammot_amd.solver import TrackPool, TrackSolver
    from siammot_amd.structures import BoxList
    cap = TrackPool.DEVICE_CAPACITY
    pool_a = TrackPool(max_dormant_frames=3)
    pool_b = TrackPool(max_dormant_frames=3)
    solver = TrackSolver(pool_a, 0.4, 0.6, 0.4, nms_mask_fn=_numpy_mask)
    rs = np.random.RandomState(5)
    multi = 0
    for f in range(60):
        tracks = [t for t in list(pool_a.get_active_ids()) + list(pool_a.get_dormant_ids()) if rs.rand() > 0.1]
        rs.shuffle(tracks)
        n_det = int(rs.randint(0, 12))
        n = n_det + len(tracks)
        c = rs.uniform(60, [1200, 640], (n, 2))
        if n > 4:
            c[n // 2:] = c[rs.randint(0, n // 2, n - n // 2)] + rs.normal(0, 8, (n - n // 2, 2))
        wh = rs.uniform(30, 110, (n, 2))
        boxes = np.concatenate((c - wh / 2, c + wh / 2), 1).astype(np.float32)
        ids = np.concatenate((np.full(n_det, -1), np.array(tracks, dtype=np.int64))).astype(np.int64)
        scores = np.concatenate((rs.uniform(0.05, 0.999, n_det), 1.0 + rs.uniform(0.05, 1.0, len(tracks)))).astype(np.float32)
        bl = BoxList(torch.from_numpy(boxes), (1280, 704), mode="xyxy")
        bl.add_field("ids", torch.from_numpy(ids))
        bl.add_field("scores", torch.from_numpy(scores.copy()))
        before = set(pool_a._dormant_ids)
        keep_mask = _numpy_mask(torch.from_numpy(boxes), torch.from_numpy(
            scores + np.isin(ids, list(pool_a.get_active_ids())).astype(np.float32)), 0.5).numpy()
        out = solver([bl])[0]
        rows = np.nonzero(keep_mask)[0]
        final = out.get_field("ids").numpy()
        assert len(rows) == len(final)
        M = n
        rec = np.zeros(8 + 4 * M + 3 * cap, dtype=np.int32)
        rec[8 + 3 * M + 3 * cap:] = ids
        rec[0], rec[2], rec[3] = len(rows), pool_a._max_id, pool_a._frame_idx
        rec[4], rec[5] = len(pool_a._active_ids), len(pool_a._dormant_ids)
        rec[8:8 + len(rows)] = rows
        rec[8 + M:8 + M + len(rows)] = final
        base = 8 + 3 * M
        rec[base:base + rec[4]] = sorted(pool_a._active_ids)
        d = sorted(pool_a._dormant_ids.items())
        rec[base + cap:base + cap + len(d)] = [k for k, _ in d]
        rec[base + 2 * cap:base + 2 * cap + len(d)] = [v for _, v in d]
        pool_b._mirror(rec, M)
        assert list(pool_b._dormant_ids.items()) == list(pool_a._dormant_ids.items()), "frame %d" % f
        assert list(pool_b.get_dormant_ids()) == list(pool_a.get_dormant_ids())
        assert pool_b._active_ids == pool_a._active_ids and pool_b._max_id == pool_a._max_id
        multi += len(set(pool_a._dormant_ids) - before) > 1
    assert multi >= 5 and pool_a._max_id > 100       # several frames suspended more than one id at once


class _HostEmulation(object):
    """The device side of ``TrackingLoop._step_lean`` on CPU tensors — test infrastructure: the one-launch solver's buffers
    and record (include/smot_emm.h, smot_track_solve_fwd) written by a twin host solver on a twin pool, the masked
    extraction by the fake tracker, ``smot_memory_carry_fwd`` by byte copies between the same addresses.  What runs
    unmodified is the loop's host logic: the pool mirror, the lazy cache, the row bookkeeping of the carried memory."""

    def __init__(self, loop, thresholds, max_dormant_frames):
        from siammot_amd.solver import TrackPool, TrackSolver
        self.loop = loop
        self.pool = TrackPool(max_dormant_frames=max_dormant_frames)
        self.solver = TrackSolver(self.pool, *thresholds, nms_mask_fn=_numpy_mask)
        self.carries = self.ahead = 0

    class Ring(object):
        bufs = (None, None)

        def record_event(self):
            pass

        def wait(self, rec, event=True):
            pass

    def track_solve(self, det, trk, bias, thresholds, nms_thresh, max_dormant, state, cap, host_record=False):
        from siammot_amd.structures import BoxList
        segs = [x for x in (det, (trk[0], trk[1] + bias, trk[2], trk[3]) if trk is not None else None) if x is not None]
        boxes = torch.cat([x[0] for x in segs])
        scores = torch.cat([x[1] for x in segs]).clone()
        ids = torch.cat([x[2] for x in segs]).clone()
        labels = torch.cat([x[3] for x in segs])
        M = boxes.shape[0]
        active_before = self.pool.get_active_ids()
        banded = scores + torch.tensor([float(int(t) in active_before) for t in ids])
        rows = np.nonzero(_numpy_mask(boxes, banded, nms_thresh).numpy())[0]
        bl = BoxList(boxes, (1280, 704), mode="xyxy")
        bl.add_field("ids", ids)
        bl.add_field("scores", scores)
        bl.add_field("labels", labels)
        in_ids = ids.numpy().copy()
        tables_before = (set(self.pool._active_ids), dict(self.pool._dormant_ids))
        out = self.solver([bl])[0]
        K = len(out)
        assert K == len(rows) and np.array_equal(out.bbox.numpy(), boxes.numpy()[rows])
        final = out.get_field("ids").numpy()
        active = self.pool.get_active_ids()
        act = [p for p in range(K) if int(final[p]) in active]
        A = len(act)
        fbuf = torch.full((10 * M,), float("nan"))
        ibuf = torch.full((4 * M,), -99, dtype=torch.int64)
        fbuf[:4 * K] = out.bbox.reshape(-1)
        fbuf[4 * M:4 * M + 4 * A] = out.bbox[act].reshape(-1)
        fbuf[8 * M:8 * M + K] = out.get_field("scores")
        fbuf[9 * M:9 * M + A] = out.get_field("scores")[act]
        ibuf[:K] = out.get_field("ids")
        ibuf[M:M + K] = out.get_field("labels")
        ibuf[2 * M:2 * M + A] = out.get_field("ids")[act]
        ibuf[3 * M:3 * M + A] = out.get_field("labels")[act]
        rec = np.zeros(8 + 4 * M + 3 * cap, dtype=np.int32)
        pool = self.pool
        rec[0], rec[1], rec[2], rec[3] = K, A, pool._max_id, pool._frame_idx
        rec[4], rec[5], rec[7] = len(pool._active_ids), len(pool._dormant_ids), M
        rec[8:8 + K] = rows
        rec[8 + M:8 + M + K] = final
        rec[8 + 2 * M:8 + 2 * M + A] = final[act]
        base = 8 + 3 * M
        rec[base:base + rec[4]] = sorted(pool._active_ids)
        d = sorted(pool._dormant_ids.items())
        rec[base + cap:base + cap + len(d)] = [k for k, _ in d]
        rec[base + 2 * cap:base + 2 * cap + len(d)] = [v for _, v in d]
        rec[8 + 3 * M + 3 * cap:] = in_ids
        # record word 6, bit 2 (csrc/track_solver.hip): nothing started, resumed, was suspended or expired — the tables stand
        if tables_before == (set(pool._active_ids), dict(pool._dormant_ids)):
            rec[6] |= 4
        state[4] = A                                     # the count a launch enqueued behind the solver reads on the device
        return fbuf, ibuf, torch.from_numpy(rec), M

    def memory_carry(self, src, n_src, dst, cap, rows, row0, row_floats, dev, stream, dst_row0_dev=0, lib=None):
        import ctypes
        if dst_row0_dev:
            row0 = ctypes.c_int.from_address(dst_row0_dev).value
            self.ahead += 1
        assert all(0 <= r < n_src for r in rows)
        for j, r in enumerate(rows):
            if row0 + j >= cap:
                continue
            for k, nbytes in enumerate((4 * row_floats, 16, 16, 8, 8, 4)):
                ctypes.memmove(dst[k] + (row0 + j) * nbytes, src[k] + r * nbytes, nbytes)
        self.carries += 1


def test_lean_frame_bookkeeping_with_carried_dormant_rows_equals_the_general_path_on_cpu(monkeypatch):
    """The host logic of the lean per-frame step — record -> pool mirror -> lazy cache -> which rows of the memory the
    head just ran on are the dormant tracks' rows of the next one (``TrackingLoop._carry_dormant``; the reference
    re-concatenates them from its cache, track_head.py:77-97) — with the device side emulated on CPU tensors, against the
    general path (TrackHead / TrackSolver / TrackPool, pinned to the reference's classes by the test above) on 180 frames in
    which tracks start, go dormant, are carried for several frames, resume and expire: outputs, memory (templates, boxes,
    search regions, ids, labels, scores, row order), pool and cache identical in every frame."""
    import types
    import siammot_amd.ops as ops_
    from fake_tracker import FakeTracker, detections
    from siammot_amd.solver import TrackPool, TrackSolver
    from siammot_amd.track_head import TrackHead, TrackingLoop, _LazyMemory
    pad, thresholds, max_dormant = 512, (0.4, 0.6, 0.4), 4

    class LeanFake(FakeTracker):
        rz = 15

        def __init__(self):
            super(LeanFake, self).__init__(pad)
            self.track_utils = types.SimpleNamespace(pad_pixels=pad)

        def track_raw(self, features, boxes, sr, z, size, sr_boxlist):
            assert torch.allclose(z[:, :, 0, 0], sr - pad + 1000.0, atol=1e-2)              # carried rows stay in step
            ids = sr_boxlist.get_field("ids")
            return boxes + 2.0, (((ids * 37) % 100).to(torch.float32) / 100.0) * 0.9 + 0.05

        def extract_cache_rows(self, features, boxes, n_valid):
            return (boxes + 1000.0)[:, :, None, None].clone(), boxes + pad

    loops = []
    for lean in (True, False):
        pool = TrackPool(max_dormant_frames=max_dormant)
        head = TrackHead(LeanFake(), types.SimpleNamespace(pad_pixels=pad), pool).eval()
        loops.append(TrackingLoop(head, TrackSolver(pool, *thresholds, nms_mask_fn=_numpy_mask)).eval())
    emu = _HostEmulation(loops[0], thresholds, max_dormant)
    monkeypatch.setattr(ops_, "track_solve", emu.track_solve)
    monkeypatch.setattr(ops_, "memory_carry", emu.memory_carry)
    monkeypatch.setattr(ops_, "_stream", lambda dev=None: None)
    pa, pb = loops[0].solver.track_pool, loops[1].solver.track_pool
    state = torch.zeros(8, dtype=torch.int32)
    monkeypatch.setattr(pa, "device_state", lambda dev: state, raising=False)
    monkeypatch.setattr(pa, "host_record_ring", lambda dev: _HostEmulation.Ring(), raising=False)
    fb0 = ops_.FALLBACKS["dormant_rows_on_the_host"]
    kept0, redone0 = ops_.MEMORY_CARRY["ahead_kept"], ops_.MEMORY_CARRY["ahead_redone"]
    rs = [np.random.RandomState(21), np.random.RandomState(21)]
    feats = (torch.zeros(1),)
    carried_frames = lazy_frames = resumed = 0
    for f in range(180):
        if f == 120:
            # a calmer stretch: no track starts, none is dropped for its score, none resumes or expires — tracks only go
            # dormant when the NMS removes their propagated row; the dormant rows are then the same ones frame after frame
            # and the copy made before the record was read is the right one
            for sv in (loops[0].solver, loops[1].solver, emu.solver):
                sv.track_thresh, sv.start_thresh, sv.resume_track_thresh = 0.0, 2.0, 2.0
                sv.track_pool._max_dormant_frames = 1000
        a = loops[0]._step_lean(feats, detections(rs[0], f))
        b = loops[1](feats, detections(rs[1], f))
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), "frame %d" % f
        assert torch.equal(a.get_field("scores"), b.get_field("scores"))
        ma, mb = loops[0].track_memory, loops[1].track_memory
        lazy = type(ma) is _LazyMemory
        lazy_frames += lazy
        carried_frames += lazy and ma.A > ma.n_act
        assert torch.equal(ma[0], mb[0]) and torch.equal(ma[1][0].bbox, mb[1][0].bbox), "memory, frame %d" % f
        assert torch.equal(ma[2][0].bbox, mb[2][0].bbox)
        for fld in ("ids", "labels", "scores"):
            assert torch.equal(ma[2][0].get_field(fld), mb[2][0].get_field(fld)), "memory %s, frame %d" % (fld, f)
            assert torch.equal(ma[1][0].get_field(fld), mb[1][0].get_field(fld))
        assert [int(t) for t in ma[2][0].host_ids] == mb[2][0].get_field("ids").tolist()
        assert pa.get_active_ids() == pb.get_active_ids() and list(pa._dormant_ids.items()) == list(pb._dormant_ids.items())
        assert pa._kill_ids == pb._kill_ids and pa._max_id == pb._max_id
        if f % 7 == 3:                                    # (a full flush of the lazy cache: every 7th frame only)
            ca, cb = pa.get_cache(), pb.get_cache()
            for tid in pa.get_dormant_ids():
                if tid in cb:
                    assert tid in ca and torch.equal(ca[tid][0], cb[tid][0]) and torch.equal(ca[tid][1].bbox, cb[tid][1].bbox)
                    assert torch.equal(ca[tid][2].bbox, cb[tid][2].bbox)
    assert carried_frames >= 40 and emu.carries >= 40 and pa._kill_ids and pa._max_id > 25, (carried_frames, emu.carries)
    # the copy enqueued before the record was read (on the guess that the dormant tracks stay the ones they were) was kept
    # on some frames and redone on others
    mc = ops_.MEMORY_CARRY
    assert emu.ahead >= 40 and mc["ahead_kept"] - kept0 >= 20 and mc["ahead_redone"] - redone0 >= 20, (emu.ahead, dict(mc))
    # every frame with active rows leaves the memory unbuilt; no frame fell back to the host form
    assert ops_.FALLBACKS["dormant_rows_on_the_host"] == fb0 and lazy_frames >= 170, (lazy_frames,)


class _FrameEntryEmulation(object):
    """``smot_track_frame_fwd`` on CPU memory — test infrastructure: the argument block is read as the library reads it
    (include/smot_emm.h, ``ops.FrameArgs``) and the stages it selects run on the arrays its pointers name — the head as the
    fake tracker (boxes + 2, a score hashed from the id), the solver as ``_HostEmulation.track_solve`` (twin host solver),
    SMOT_STAGE_CARRY as csrc/track_solver.hip does it (small fields behind the count the solver determined, templates and
    search regions at the caller's guess), the masked extraction as the fake tracker's.  Calls execute in the order they
    are made: the stream order of the real thing."""

    def __init__(self, host_emu, pad, row_floats):
        import siammot_amd.ops as ops_
        self.ops, self.host, self.pad, self.rf = ops_, host_emu, pad, row_floats
        self.names = ops_._FRAME_PTRS + ops_._FRAME_INTS + ops_._FRAME_FLOATS
        self.heads = self.carried = self.no_head_frames = 0

    @staticmethod
    def _arr(ptr, n, ctype, dtype):
        import ctypes
        return np.frombuffer((ctype * n).from_address(ptr), dtype=dtype) if n > 0 else np.zeros(0, dtype)

    def f32(self, ptr, n):
        import ctypes
        return self._arr(ptr, n, ctypes.c_float, np.float32)

    def i64(self, ptr, n):
        import ctypes
        return self._arr(ptr, n, ctypes.c_int64, np.int64)

    def __call__(self, lib, addr, dev, stream):
        import ctypes
        ops_ = self.ops
        fmt = ops_.FrameArgs._FMT
        a = dict(zip(self.names, fmt.unpack_from((ctypes.c_char * fmt.size).from_address(addr))))
        stages, n_trk, n_det, rf = a["stages"], a["n_trk"], a["n_det"], self.rf
        assert a["C"] * a["rz"] * a["rz"] == rf
        if stages & ops_.STAGE_HEAD and n_trk > 0:
            tb, ids = self.f32(a["tpl_boxes"], 4 * n_trk), torch.from_numpy(self.i64(a["trk_ids"], n_trk).copy())
            self.f32(a["trk_boxes"], 4 * n_trk)[:] = tb + np.float32(2.0)
            self.f32(a["trk_conf"], n_trk)[:] = ((((ids * 37) % 100).to(torch.float32) / 100.0) * 0.9 + 0.05).numpy()
            self.heads += 1
        if not stages & ops_.STAGE_SOLVE:
            return
        M = n_det + n_trk
        if n_trk == 0 and n_det > 0:
            self.no_head_frames += 1                 # a frame with rows and no propagated track: no head ran for it
        det = trk = None
        if n_det:
            det = (torch.from_numpy(self.f32(a["det_boxes"], 4 * n_det).reshape(n_det, 4).copy()),
                   torch.from_numpy(self.f32(a["det_scores"], n_det).copy()),
                   torch.from_numpy(self.i64(a["det_ids"], n_det).copy()), torch.from_numpy(self.i64(a["det_labels"], n_det).copy()))
        if n_trk:
            trk = (torch.from_numpy(self.f32(a["trk_boxes"], 4 * n_trk).reshape(n_trk, 4).copy()),
                   torch.from_numpy(self.f32(a["trk_conf"], n_trk).copy()),
                   torch.from_numpy(self.i64(a["trk_ids"], n_trk).copy()), torch.from_numpy(self.i64(a["trk_labels"], n_trk).copy()))
        state = torch.zeros(8, dtype=torch.int32)
        fbuf, ibuf, rec, M2 = self.host.track_solve(det, trk, 1.0, (a["track_thresh"], a["start_thresh"], a["resume_thresh"]),
                                                    a["nms_thresh"], a["max_dormant_frames"], state, a["pool_capacity"])
        assert M2 == M
        fb, ib = fbuf.numpy(), ibuf.numpy()
        K, A = int(rec[0]), int(rec[1])
        self.f32(a["out_boxes"], 4 * M)[:4 * K] = fb[:4 * K]
        self.f32(a["act_boxes"], 4 * M)[:4 * A] = fb[4 * M:4 * M + 4 * A]
        self.f32(a["out_scores"], M)[:K] = fb[8 * M:8 * M + K]
        self.f32(a["act_scores"], M)[:A] = fb[9 * M:9 * M + A]
        self.i64(a["out_ids"], M)[:K] = ib[:K]
        self.i64(a["out_labels"], M)[:K] = ib[M:M + K]
        self.i64(a["act_ids"], M)[:A] = ib[2 * M:2 * M + A]
        self.i64(a["act_labels"], M)[:A] = ib[3 * M:3 * M + A]
        r = rec.numpy()
        np.frombuffer((ctypes.c_int32 * len(r)).from_address(a["record"]), dtype=np.int32)[:] = r
        np.frombuffer((ctypes.c_int32 * 8).from_address(a["pool_state"]), dtype=np.int32)[4] = A
        nz, nsr = self.f32(a["next_templates"], M * rf).reshape(M, rf), self.f32(a["next_sr"], 4 * M).reshape(M, 4)
        if stages & ops_.STAGE_CARRY and a["carry_rows"] > 0:
            r0, D, guess = a["carry_src_row0"], a["carry_rows"], a["carry_dst_row0"]
            n_src = r0 + D
            src_z = self.f32(a["carry_templates"], n_src * rf).reshape(n_src, rf)
            src_b, src_sr = self.f32(a["carry_boxes"], 4 * n_src).reshape(n_src, 4), self.f32(a["carry_sr"], 4 * n_src).reshape(n_src, 4)
            src_i, src_l, src_s = self.i64(a["carry_ids"], n_src), self.i64(a["carry_labels"], n_src), self.f32(a["carry_scores"], n_src)
            ab, ai = self.f32(a["act_boxes"], 4 * M).reshape(M, 4), self.i64(a["act_ids"], M)
            al, asc = self.i64(a["act_labels"], M), self.f32(a["act_scores"], M)
            for j in range(D):
                if guess + j < M:                                   # the carrying workgroups: to the caller's guess
                    nz[guess + j], nsr[guess + j] = src_z[r0 + j], src_sr[r0 + j]
                if A + j < M:                                       # workgroup 0: behind the count it determined
                    ab[A + j], ai[A + j], al[A + j], asc[A + j] = src_b[r0 + j], src_i[r0 + j], src_l[r0 + j], src_s[r0 + j]
            self.carried += 1
        if stages & ops_.STAGE_EXTRACT:
            act = self.f32(a["act_boxes"], 4 * M).reshape(M, 4)[:A]
            nz[:A] = np.repeat(act + np.float32(1000.0), rf // 4, axis=1) if rf != 4 else act + np.float32(1000.0)
            nsr[:A] = act + np.float32(self.pad)


@pytest.mark.parametrize("mode", ["ahead", "early", "early_no_track_left"])
def test_frame_entry_point_bookkeeping_with_speculation_and_carried_rows_on_cpu(monkeypatch, mode):
    """``TrackingLoop._step_native`` — the default path: two calls of the frame entry point per frame on a block that stays
    packed, the next frame's head launched before the record is read when the caller shows the next features, the dormant
    rows carried inside the solver's launch on a guess — with the library call emulated on CPU memory
    (``_FrameEntryEmulation`` reads the argument block as the library does), against the general path on 200 frames of
    churning and calm traffic: outputs identical in every frame, memory and pool identical whenever they are compared
    (every third frame: looking at a memory builds it, and a built memory takes no speculative head — on purpose), both
    outcomes of both guesses occur.
    ``mode == "early"`` (round 5): the reference's one-frame contract — ``TrackingLoop.forward(features, detections)``, no
    next frame shown —: every call prepares the NEXT call's head launch while the (emulated) GPU works and the next call
    enqueues it on its first line, validating afterwards; the row count poked on a guess is corrected from the record.
    Same comparison; the early head must be the one used on (nearly) every frame whose memory nobody looked at.
    ``mode == "early_no_track_left"`` (round 6): the same contract on traffic whose first frames — and a stretch in
    the middle — carry only detections below the start threshold: frames with rows (M >= 1) that leave NO active track, so
    that the head range poked for the next call names a head that never runs.  The next call has no head; the row count
    the solver sees must be that call's own (``poke_rest`` writes it), the record's row count is checked against the
    frame's layout, and every frame equals the general path."""
    import ctypes
    import types
    import siammot_amd.ops as ops_
    from fake_tracker import FakeTracker, detections
    from siammot_amd.solver import TrackPool, TrackSolver
    from siammot_amd.track_head import TrackHead, TrackingLoop, _LazyMemory
    pad, thresholds, max_dormant = 512, (0.4, 0.6, 0.4), 4
    cap = TrackPool.DEVICE_CAPACITY

    class NativeFake(FakeTracker):
        rz, rx, pad_pixels, sigma, amodal, use_centerness = 1, 2, pad, 0.4, False, True

        def __init__(self):
            super(NativeFake, self).__init__(pad)
            self.track_utils = types.SimpleNamespace(pad_pixels=pad, search_expansion=1.0, min_search_wh=0.0)
            self.feature_extractor = types.SimpleNamespace(pooler_x=types.SimpleNamespace(scales=(0.25,), sampling_ratio=2))
            self.predictor = types.SimpleNamespace(param_dict=lambda: {}, gn_groups=1, gn_eps=1e-5)

    loops = []
    for lean in (True, False):
        pool = TrackPool(max_dormant_frames=max_dormant)
        head = TrackHead(NativeFake(), types.SimpleNamespace(pad_pixels=pad), pool).eval()
        loops.append(TrackingLoop(head, TrackSolver(pool, *thresholds, nms_mask_fn=_numpy_mask)).eval())
    loops[1]._lean_ok = lambda d: False                     # general path
    host = _HostEmulation(loops[0], thresholds, max_dormant)
    emu = _FrameEntryEmulation(host, pad, 4)
    geom = types.SimpleNamespace(a_fp=0, a_hs=0, a_ws=0, a_pc=0, a_sc=0, L=1, C=4)
    monkeypatch.setattr(ops_, "_geometry", lambda features, scales, pad_pixels, dev: geom)
    monkeypatch.setattr(ops_, "_geometry_refresh", lambda g, features, dev: True)
    monkeypatch.setattr(ops_, "_param_block", lambda params: types.SimpleNamespace(a_pp=4711))
    monkeypatch.setattr(ops_, "_check_segment", lambda *a: None)
    monkeypatch.setattr(ops_, "_stream", lambda dev=None: ctypes.c_void_p(0))
    monkeypatch.setattr(ops_, "load_library", lambda: types.SimpleNamespace(
        smot_emm_track_ws_floats=lambda *a: 64, smot_box_refine_ws_floats=lambda *a: 0))
    monkeypatch.setattr(ops_, "track_frame_addr", emu)
    monkeypatch.setattr(ops_, "memory_carry", host.memory_carry)
    pa, pb = loops[0].solver.track_pool, loops[1].solver.track_pool
    state = torch.zeros(8 + 3 * cap, dtype=torch.int32)
    monkeypatch.setattr(pa, "device_state", lambda dev: state, raising=False)

    class Ring(_HostEmulation.Ring):
        def next(self):
            return torch.zeros(8 + 4 * 512 + 3 * cap, dtype=torch.int32)
    monkeypatch.setattr(pa, "host_record_ring", lambda dev: Ring(), raising=False)
    for key in ("launched", "used", "discarded", "early_launched", "early_used", "early_discarded"):
        ops_.SPECULATION[key] += 0
    no_track = mode == "early_no_track_left"
    if no_track:
        mode = "early"

    def traffic(r, f):
        d = detections(r, f)
        if no_track and (f < 3 or 60 <= f < 66 or 140 <= f < 143):
            d.add_field("scores", torch.full_like(d.get_field("scores"), 0.05))       # below every threshold: nothing starts / resumes
        return d
    if mode == "early":
        loops[0]._lean_ok = lambda d: True
        loops[0]._native_ok = lambda d: True
    sp0, mc0 = dict(ops_.SPECULATION), dict(ops_.MEMORY_CARRY)
    fb0 = ops_.FALLBACKS["dormant_rows_on_the_host"]
    rs = [np.random.RandomState(33), np.random.RandomState(33)]
    feats = (torch.zeros(1, 4, 8, 8),)
    compared = 0
    for f in range(200):
        if f == 110:
            for sv in (loops[0].solver, loops[1].solver, host.solver):      # the calm stretch (see the test above)
                sv.track_thresh, sv.start_thresh, sv.resume_track_thresh = 0.0, 2.0, 2.0
                sv.track_pool._max_dormant_frames = 1000
        heads0 = emu.heads
        if mode == "early":
            a = loops[0](feats, traffic(rs[0], f))
        else:
            a = loops[0]._step_native(feats, traffic(rs[0], f), next_features=feats)
        assert emu.heads - heads0 <= 2                       # (a head launched early / ahead and, at most, launched again)
        b = loops[1](feats, traffic(rs[1], f))
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), "frame %d" % f
        assert torch.equal(a.get_field("scores"), b.get_field("scores")) and torch.equal(a.get_field("labels"), b.get_field("labels"))
        assert pa.get_active_ids() == pb.get_active_ids() and list(pa._dormant_ids.items()) == list(pb._dormant_ids.items())
        assert pa._kill_ids == pb._kill_ids and pa._max_id == pb._max_id
        if f % 3 == 2 or f > 190:
            ma, mb = loops[0].track_memory, loops[1].track_memory
            if no_track and len(mb[2][0]) == 0:
                assert len(ma[2][0]) == 0                     # no track left: an empty memory on both paths
                continue
            assert type(ma) is _LazyMemory
            assert torch.equal(ma[0], mb[0]) and torch.equal(ma[1][0].bbox, mb[1][0].bbox), "memory, frame %d" % f
            assert torch.equal(ma[2][0].bbox, mb[2][0].bbox)
            for fld in ("ids", "labels", "scores"):
                assert torch.equal(ma[2][0].get_field(fld), mb[2][0].get_field(fld)), "memory %s, frame %d" % (fld, f)
            assert [int(t) for t in ma[2][0].host_ids] == mb[2][0].get_field("ids").tolist()
            compared += 1
    sp = {k: ops_.SPECULATION[k] - sp0.get(k, 0) for k in ("launched", "used", "discarded", "early_launched", "early_used",
                                                           "early_discarded")}
    mc = {k: ops_.MEMORY_CARRY[k] - mc0.get(k, 0) for k in ("in_the_solver_launch", "ahead_kept", "ahead_redone", "launched")}
    if no_track:
        assert emu.no_head_frames >= 3, emu.no_head_frames      # calls without a head behind frames with rows did occur
        return
    if mode == "early":
        # every frame whose memory was left unbuilt (two of three: the comparison builds the third) started with the head
        # that the call before had prepared, whatever happened to the row count in between; none was launched twice
        assert sp["launched"] == 0 and sp["early_launched"] >= 110 and sp["early_used"] == sp["early_launched"], sp
        assert sp["early_discarded"] == 0 and emu.heads <= 200, (sp, emu.heads)
    else:
        # (the last frame's speculative head may still be waiting for a call that never comes)
        assert sp["launched"] - sp["used"] - sp["discarded"] in (0, 1) and sp["used"] >= 20 and sp["discarded"] >= 10, sp
    assert mc["in_the_solver_launch"] >= 60 and mc["ahead_kept"] >= 20 and mc["ahead_redone"] >= 20, mc
    assert emu.carried == mc["in_the_solver_launch"] and compared >= 60
    assert ops_.FALLBACKS["dormant_rows_on_the_host"] == fb0 and pa._kill_ids and pa._max_id > 25


@pytest.mark.gpu
def test_solver_on_the_device_with_the_hip_nms():
    _run("cuda:0", None)


def test_tracking_loop_matches_the_reference_classes():
    """TrackHead + solver + pool of this repository replay the sequence that oracle/gen_golden_tracking.py pushed
    through the reference's own TrackHead / TrackSolver / TrackPool (same deterministic fake tracker)."""
    import types
    from fake_tracker import SEQ, FakeTracker, detections
    from siammot_amd.solver import TrackPool, TrackSolver
    from siammot_amd.track_head import TrackHead, TrackingLoop
    gold = np.load(os.path.join(os.path.dirname(GOLDEN), "tracking_sequence.npz"))
    pool = TrackPool(max_dormant_frames=SEQ["max_dormant_frames"])
    head = TrackHead(FakeTracker(SEQ["pad"]), types.SimpleNamespace(pad_pixels=SEQ["pad"]), pool).eval()
    loop = TrackingLoop(head, TrackSolver(pool, *SEQ["thresholds"], nms_mask_fn=_numpy_mask)).eval()
    rs = np.random.RandomState(SEQ["seed"])
    feats = (torch.zeros(1),)
    for f in range(SEQ["frames"]):
        res = loop(feats, detections(rs, f))
        assert res.get_field("ids").tolist() == gold["f%02d_ids" % f].tolist(), "frame %d" % f
        assert np.array_equal(res.get_field("scores").numpy(), gold["f%02d_scores" % f])
        assert np.array_equal(res.bbox.numpy(), gold["f%02d_boxes" % f])
        mem = loop.track_memory
        assert mem[2][0].get_field("ids").tolist() == gold["f%02d_mem_ids" % f].tolist(), "memory, frame %d" % f
        assert np.array_equal(mem[0].numpy().reshape(len(mem[2][0]), -1), gold["f%02d_mem_feat" % f])
    assert pool._max_id > 20 and len(pool.get_active_ids()) > 5
    loop.reset()
    assert loop.track_memory is None and pool.get_active_ids() == set()


@pytest.mark.gpu
def test_tracking_loop_with_the_hip_head_runs_and_stays_consistent():
    """The real EMM head inside the loop (random features and weights: the propagated boxes are arbitrary, the
    bookkeeping must still hold): unique ids, memory aligned with the pool, dormant tracks carried and expired."""
    import golden_inputs as gi
    from fake_tracker import detections
    from siammot_amd.config import get_default_cfg
    from siammot_amd.track_head import build_tracking_loop
    cfg = get_default_cfg(channels=32)
    cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 2
    loop = build_tracking_loop(cfg, device="cuda:0", refine_tracks=False)
    pool = loop.track.track_pool
    rs = np.random.RandomState(5)
    shapes = gi.feature_shapes((1280, 704), 32)
    started = 0
    for f in range(12):
        feats = tuple(torch.from_numpy(rs.standard_normal(s).astype(np.float32)).cuda() for s in shapes)
        out = loop(feats, detections(rs, f).to("cuda:0"))
        ids = out.get_field("ids").cpu().numpy()
        tracked = ids[ids >= 0]
        assert len(np.unique(tracked)) == len(tracked)
        assert set(tracked.tolist()) <= pool.get_active_ids()
        z, sr, boxes = loop.track_memory
        assert z.shape[0] == len(sr[0]) == len(boxes[0]) and (z.numel() == 0 or tuple(z.shape[1:]) == (32, 15, 15))
        mem_ids = set(boxes[0].get_field("ids").cpu().tolist())
        assert mem_ids == pool.get_active_ids() | (pool.get_dormant_ids() & set(pool.get_cache()))
        started = max(started, pool._max_id + 1)
    assert started >= 10


@pytest.mark.gpu
def test_one_launch_solver_replays_the_reference_tracking_sequence():
    """The device-resident path (one kernel for merge + NMS + decisions + pool + active rows, un-concatenated
    segments, +1 band applied in the kernel) through TrackingLoop on the sequence the reference's own TrackHead /
    TrackSolver / TrackPool produced (tests/golden/tracking_sequence.npz).  Ids, scores and boxes are compared per
    id: the kernel keeps the pool tables in its own order, so dormant rows may sit in another order in the memory
    than the reference's Python-set iteration gives — the per-track content must be identical."""
    import types
    from fake_tracker import SEQ, FakeTracker, detections
    from siammot_amd.solver import TrackPool, TrackSolver
    from siammot_amd.track_head import TrackHead, TrackingLoop
    dev = torch.device("cuda:0")
    gold = np.load(os.path.join(os.path.dirname(GOLDEN), "tracking_sequence.npz"))
    pool = TrackPool(max_dormant_frames=SEQ["max_dormant_frames"])

    class HostFake(FakeTracker):
        """The stand-in tracker's own arithmetic stays on the CPU (as in the golden run: torch's device kernels round
        ``x / 100 * 0.9 + 0.05`` differently in the last bit); only its results live on the device."""

        def forward(self, features, boxes, sr, targets=None, template_features=None):
            _, out, _ = FakeTracker.forward(self, features, [boxes[0].to("cpu")], [sr[0].to("cpu")],
                                            template_features=template_features.cpu())
            return {}, [out[0].to(dev)], {}

        def extract_cache(self, features, detection):
            f, sr, d = FakeTracker.extract_cache(self, features, detection.to("cpu"))
            keep = detection                     # the solver's active rows (with their host_ids) stay the memory boxes
            return f.to(dev), [sr[0].to(dev)], [keep]
    head = TrackHead(HostFake(SEQ["pad"]), types.SimpleNamespace(pad_pixels=SEQ["pad"]), pool).eval()
    solver = TrackSolver(pool, *SEQ["thresholds"])
    loop = TrackingLoop(head, solver).eval()
    rs = np.random.RandomState(SEQ["seed"])
    feats = (torch.zeros(1, device=dev),)
    launches = []
    import siammot_amd.ops as ops
    real = ops.track_solve
    ops.track_solve = lambda *a, **k: (launches.append(1), real(*a, **k))[1]
    try:
        for f in range(SEQ["frames"]):
            res = loop(feats, detections(rs, f).to(dev))
            ids = res.get_field("ids").cpu().numpy()
            g_ids = gold["f%02d_ids" % f]
            # detections keep their order; propagated tracks follow in memory order: compare as (id, score, box) sets
            # for tracked rows and positionally for the detection rows in front
            def rows(i, s, b):
                return sorted((int(a), float(c), tuple(map(float, d))) for a, c, d in zip(i, s, b))
            assert rows(ids, res.get_field("scores").cpu().numpy(), res.bbox.cpu().numpy()) == \
                rows(g_ids, gold["f%02d_scores" % f], gold["f%02d_boxes" % f]), "frame %d" % f
            mem = loop.track_memory
            m_ids = mem[2][0].get_field("ids").cpu().numpy()
            assert sorted(m_ids.tolist()) == sorted(gold["f%02d_mem_ids" % f].tolist()), "memory ids, frame %d" % f
            got = dict(zip(m_ids.tolist(), mem[0].cpu().numpy().reshape(len(m_ids), -1).tolist()))
            want = dict(zip(gold["f%02d_mem_ids" % f].tolist(), gold["f%02d_mem_feat" % f].tolist()))
            assert got == want, "memory features, frame %d" % f
            assert mem[2][0].host_ids is not None and list(mem[2][0].host_ids) == m_ids.tolist()
    finally:
        ops.track_solve = real
    assert len(launches) == SEQ["frames"]                      # every frame took the one-launch path
    assert pool._max_id > 20 and len(pool.get_active_ids()) > 5


@pytest.mark.gpu
def test_one_launch_solver_edge_cases():
    """Empty segments, only tracks, ids resumed and suspended in one frame, host-side pool edits between frames
    (the device copy is refreshed), more boxes than the kernel takes (falls back to the multi-kernel path)."""
    from siammot_amd.solver import TrackPool, TrackSolver
    from siammot_amd.structures import BoxList
    import siammot_amd.ops as ops
    dev = torch.device("cuda:0")

    def bl(boxes, ids, scores):
        b = BoxList(torch.tensor(boxes, dtype=torch.float32, device=dev).reshape(-1, 4), (1280, 704))
        b.add_field("ids", torch.tensor(ids, dtype=torch.int64, device=dev))
        b.add_field("scores", torch.tensor(scores, dtype=torch.float32, device=dev))
        b.add_field("labels", torch.ones(len(ids), dtype=torch.int64, device=dev))
        return b
    pool = TrackPool(max_dormant_frames=3)
    solver = TrackSolver(pool, 0.4, 0.6, 0.4)
    out = solver([bl([[0, 0, 50, 50], [200, 200, 260, 260], [400, 100, 450, 190]], [-1, -1, -1], [0.9, 0.7, 0.3])])[0]
    assert out.get_field("ids").tolist() == [0, 1, -1] and pool.get_active_ids() == {0, 1} and pool._frame_idx == 1
    assert out.active_rows.get_field("ids").tolist() == [0, 1] and out.host_ids.tolist() == [0, 1, -1]
    # only propagated tracks (no detections): track 1 falls below the track threshold -> dormant
    trk = bl([[2, 2, 52, 52], [202, 202, 262, 262]], [0, 1], [0.8, 0.2])
    out = solver.solve(None, trk, track_score_bias=1.0)
    assert out.get_field("ids").tolist() == [0, -1] and pool.get_active_ids() == {0} and pool.get_dormant_ids() == {1}
    assert torch.allclose(out.get_field("scores"), torch.tensor([0.8, 0.2], device=dev))
    assert trk.get_field("scores").tolist() == pytest.approx([2.8, 2.2])          # banded in place (+1 bias, +1 active)
    # host-side edit between frames: the mirror is authoritative until the next kernel call uploads it
    pool.resume_track(1)
    out = solver([bl([[0, 0, 50, 50], [200, 200, 260, 260]], [0, 1], [1.9, 1.9])])[0]
    assert out.get_field("ids").tolist() == [0, 1] and pool.get_active_ids() == {0, 1}
    # an overlapping, higher-scoring detection removes track 0 in NMS -> suspended; the detection starts id 2
    out = solver([bl([[0, 0, 50, 50], [1, 1, 51, 51], [200, 200, 260, 260]], [0, -1, 1], [1.5, 0.95, 1.9])])[0]
    assert pool.get_active_ids() == {0, 1} and out.get_field("ids").tolist() == [0, 1]      # active band wins NMS
    # too many boxes for the single-workgroup kernel: the multi-kernel path takes over, same semantics
    n = ops.track_solve_max_boxes() + 10
    rs = np.random.RandomState(0)
    xy = rs.uniform(0, 1000, (n, 2))
    big = bl(np.concatenate((xy, xy + 20), 1).tolist(), [-1] * n, rs.uniform(0.1, 0.5, n).tolist())
    assert not solver._device_path(big)
    out = solver([big])[0]
    assert len(out) > 0 and (out.get_field("ids") < 0).all()


@pytest.mark.gpu
def test_reset_between_videos_leaves_nothing_behind():
    """``TrackingLoop.reset()`` (rcnn.py:37-39) between two videos: the second video tracked after a first one must come out
    exactly as on a fresh loop — device-resident pool, record ring, frame plan, lazily noted memory and the id-table
    snapshot of the mirror all start over (ids restart at 0)."""
    import golden_inputs as gi
    from fake_tracker import detections
    from siammot_amd.config import get_default_cfg
    from siammot_amd.track_head import build_tracking_loop
    dev = torch.device("cuda:0")
    cfg = get_default_cfg(channels=32)
    cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 3
    cfg.MODEL.TRACK_HEAD.TRACK_THRESH = 0.5
    cfg.MODEL.TRACK_HEAD.RESUME_TRACK_THRESH = 0.5
    torch.manual_seed(21)
    used, fresh = [build_tracking_loop(cfg, device=dev, refine_tracks=False) for _ in range(2)]
    with torch.no_grad():
        for name in ("cls", "center", "reg"):
            getattr(used.track.tracker.predictor, name).weight.mul_(20.0)
    fresh.track.tracker.load_state_dict(used.track.tracker.state_dict())
    shapes = gi.feature_shapes((1280, 704), 32)
    rs_f = np.random.RandomState(4)
    frames = [tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes) for _ in range(7)]
    rs = np.random.RandomState(8)
    for f in range(7):                                   # first video, on the used loop only
        used(frames[f], detections(rs, f).to(dev))
    assert used.solver.track_pool._max_id > 0
    used.reset()
    ra, rb = np.random.RandomState(15), np.random.RandomState(15)
    for f in range(7):                                   # second video on both
        a = used(frames[6 - f], detections(ra, f + 3).to(dev))
        b = fresh(frames[6 - f], detections(rb, f + 3).to(dev))
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), "frame %d" % f
        assert torch.equal(a.get_field("scores"), b.get_field("scores")), "frame %d" % f
        pa, pb = used.solver.track_pool, fresh.solver.track_pool
        assert pa.get_active_ids() == pb.get_active_ids() and pa._dormant_ids == pb._dormant_ids and pa._max_id == pb._max_id
    ma, mb = used.track_memory, fresh.track_memory
    assert torch.equal(ma[0], mb[0]) and torch.equal(ma[2][0].get_field("ids"), mb[2][0].get_field("ids"))


@pytest.mark.gpu
def test_unbuilt_track_memory_feeds_the_next_head_by_address():
    """Frames without dormant tracks leave the next memory UNBUILT (track_head._LazyMemory): the following frame's head takes
    five device addresses from it, no view tensor and no BoxList is made unless somebody reads ``loop.track_memory``.  A
    loop whose memory is never looked at must produce what the general path produces, frame by frame, and the memory it
    finally builds on access must equal the general path's."""
    import golden_inputs as gi
    from siammot_amd.config import get_default_cfg
    from siammot_amd.structures import BoxList
    from siammot_amd.track_head import build_tracking_loop, _LazyMemory
    dev = torch.device("cuda:0")
    cfg = get_default_cfg(channels=32)
    torch.manual_seed(11)
    loops = [build_tracking_loop(cfg, device=dev, refine_tracks=False) for _ in range(2)]
    loops[1].track.tracker.load_state_dict(loops[0].track.tracker.state_dict())
    loops[1]._lean_ok = lambda d: False
    boxes = torch.tensor([[100.0 + 180 * i, 90.0 + 70 * (i % 3), 170.0 + 180 * i, 250.0 + 70 * (i % 3)] for i in range(6)])
    shapes = gi.feature_shapes((1280, 704), 32)
    rs_f = np.random.RandomState(2)

    def dets():
        d = BoxList(boxes.clone().to(dev), (1280, 704))
        d.add_field("scores", torch.full((6,), 0.95, device=dev))
        d.add_field("labels", torch.ones(6, dtype=torch.int64, device=dev))
        d.add_field("ids", torch.full((6,), -1, dtype=torch.int64, device=dev))
        return d
    unbuilt = 0
    for f in range(8):
        feats = tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes)
        a, b = [lp(feats, dets()) for lp in loops]
        if f == 0:
            for lp in loops:                     # from here on no track is started or suspended: the memory is the active rows
                lp.solver.start_thresh, lp.solver.track_thresh = 2.0, 0.0
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), "frame %d" % f
        assert torch.equal(a.get_field("scores"), b.get_field("scores")), "frame %d" % f
        mem = loops[0].__dict__["track_memory"]                   # (looked at without touching its contents)
        unbuilt += type(mem) is _LazyMemory and mem._val is None
    assert unbuilt == 8 and not loops[0].solver.track_pool.get_dormant_ids()
    ma, mb = loops[0].track_memory, loops[1].track_memory          # built now
    assert len(ma) == 3 and torch.equal(ma[0], mb[0]) and torch.equal(ma[1][0].bbox, mb[1][0].bbox)
    assert torch.equal(ma[2][0].bbox, mb[2][0].bbox) and torch.equal(ma[2][0].get_field("ids"), mb[2][0].get_field("ids"))
    assert ma[1][0].size == mb[1][0].size and ma[2][0].get_field("labels").tolist() == mb[2][0].get_field("labels").tolist()
    assert loops[0].solver.track_pool.get_cache().keys() == loops[1].solver.track_pool.get_cache().keys()


@pytest.mark.gpu
def test_a_frame_on_the_host_solver_between_one_launch_frames_keeps_the_dormant_templates():
    """A loop that leaves the one-launch path for single frames (more than 512 boxes, detections with extra fields, ...) runs
    the host solver there: tracks suspended in such a frame must enter the cache with the row of the LAST memory they were
    active in — the lazily noted one (TrackPool.note_memory) — not with an older entry.  (Found by
    measure/debug/loop_equiv_soak.py at a seed whose track count crossed 512: the host-path transitions did not flush the
    pending memory first and the dormant rows of the next memories carried templates one frame too old.)"""
    import golden_inputs as gi
    from fake_tracker import detections
    from siammot_amd.config import get_default_cfg
    from siammot_amd.track_head import build_tracking_loop
    dev = torch.device("cuda:0")
    cfg = get_default_cfg(channels=32)
    cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 3
    cfg.MODEL.TRACK_HEAD.TRACK_THRESH = 0.5
    cfg.MODEL.TRACK_HEAD.RESUME_TRACK_THRESH = 0.5
    torch.manual_seed(3)
    loops = [build_tracking_loop(cfg, device=dev, refine_tracks=False) for _ in range(2)]
    with torch.no_grad():
        for name in ("cls", "center", "reg"):
            getattr(loops[0].track.tracker.predictor, name).weight.mul_(20.0)
    loops[1].track.tracker.load_state_dict(loops[0].track.tracker.state_dict())
    # the reference loop: general path + host solver in every frame
    loops[1]._lean_ok = lambda d: False
    loops[1].solver._device_path = lambda *a, **k: False
    lean_ok, device_path = loops[0]._lean_ok, loops[0].solver._device_path
    shapes = gi.feature_shapes((1280, 704), 32)
    rs_f = np.random.RandomState(9)
    rs = [np.random.RandomState(5), np.random.RandomState(5)]
    host_frames = dormant_after_host = 0
    for f in range(30):
        host = f % 3 == 2                                   # every third frame of loop 0 goes through the host solver
        loops[0]._lean_ok = (lambda d: False) if host else lean_ok
        loops[0].solver._device_path = (lambda *a, **k: False) if host else device_path
        feats = tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes)
        a, b = [lp(feats, detections(r, f).to(dev)) for lp, r in zip(loops, rs)]
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), "frame %d" % f
        ma, mb = loops[0].track_memory, loops[1].track_memory
        assert torch.equal(ma[2][0].get_field("ids"), mb[2][0].get_field("ids")), "memory ids, frame %d" % f
        assert torch.equal(ma[0], mb[0]) and torch.equal(ma[1][0].bbox, mb[1][0].bbox), "memory, frame %d" % f
        pa, pb = loops[0].solver.track_pool, loops[1].solver.track_pool
        assert pa.get_active_ids() == pb.get_active_ids() and pa._dormant_ids == pb._dormant_ids
        host_frames += host
        dormant_after_host += bool(host and pa.get_dormant_ids())
    assert host_frames == 10 and dormant_after_host > 0


@pytest.mark.gpu
@pytest.mark.parametrize("row_shape", [(32, 15, 15), (128, 7, 7), (3, 7, 7)])
def test_memory_carry_copies_the_named_rows(row_shape):
    """``smot_memory_carry_fwd``: D rows of a source memory (templates, boxes, search regions, ids, labels, scores) to rows
    dst_row0 .. of the destination buffers — first row by value and read from a device word; rows beyond the capacity are
    dropped; nothing else is written (16-byte and scalar copy forms)."""
    import siammot_amd.ops as ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    n_src, cap = 23, 17
    rf = int(np.prod(row_shape))
    src = [torch.randn((n_src,) + row_shape, generator=g), torch.randn(n_src, 4, generator=g), torch.randn(n_src, 4, generator=g),
           torch.randint(0, 1 << 40, (n_src,), generator=g), torch.randint(0, 9, (n_src,), generator=g), torch.rand(n_src, generator=g)]
    src = [t.to(dev) for t in src]

    def fresh():
        return [torch.full((cap,) + row_shape, -7.0, device=dev), torch.full((cap, 4), -7.0, device=dev),
                torch.full((cap, 4), -7.0, device=dev), torch.full((cap,), -7, dtype=torch.int64, device=dev),
                torch.full((cap,), -7, dtype=torch.int64, device=dev), torch.full((cap,), -7.0, device=dev)]
    rows = [22, 0, 5, 5, 11]
    for row0, on_device in ((4, False), (9, True), (14, True)):
        dst = fresh()
        want = [t.clone() for t in dst]
        for j, r in enumerate(rows):
            if row0 + j < cap:
                for w, s_ in zip(want, src):
                    w[row0 + j] = s_[r]
        word = torch.tensor([0, 0, 0, 0, row0, 0], dtype=torch.int32, device=dev)
        ops.memory_carry([t.data_ptr() for t in src], n_src, [t.data_ptr() for t in dst], cap, rows, 0 if on_device else row0, rf,
                         dev, ops._stream(dev), dst_row0_dev=(word.data_ptr() + 16) if on_device else 0)
        torch.cuda.synchronize()
        for w, d in zip(want, dst):
            assert torch.equal(w, d)
    with pytest.raises(RuntimeError):                    # by value, the rows must fit
        ops.memory_carry([t.data_ptr() for t in src], n_src, [t.data_ptr() for t in fresh()], cap, rows, 14, rf, dev,
                         ops._stream(dev))
    with pytest.raises(RuntimeError):                    # a source row that does not exist
        ops.memory_carry([t.data_ptr() for t in src], n_src, [t.data_ptr() for t in fresh()], cap, [23], 0, rf, dev,
                         ops._stream(dev))


@pytest.mark.gpu
@pytest.mark.parametrize("guess", [5, 2, 9])
def test_solver_launch_carries_the_dormant_rows(guess):
    """``smot_track_solve_carry_fwd``: the solver's own results are those of the plain launch bit for bit; the dormant rows'
    boxes / ids / labels / scores stand behind the active rows it determined (whatever the caller guessed); their templates
    and search regions stand at the GUESSED rows (right when the guess equals the count; a wrong guess is the caller's to
    repair, smot_memory_carry_fwd) and nothing else of the next memory's buffers is written."""
    import siammot_amd.ops as ops
    from siammot_amd.solver import TrackPool
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    n_det, n_src, D, row0 = 5, 7, 3, 2
    row_shape = (32, 15, 15)
    rf = int(np.prod(row_shape))
    det_boxes = torch.tensor([[10 + 200 * i, 20, 90 + 200 * i, 180] for i in range(n_det)], dtype=torch.float32, device=dev)
    src = [torch.randn((n_src,) + row_shape, generator=g), torch.randn(n_src, 4, generator=g), torch.randn(n_src, 4, generator=g),
           torch.randint(0, 1 << 40, (n_src,), generator=g), torch.randint(0, 9, (n_src,), generator=g), torch.rand(n_src, generator=g)]
    src = [t.to(dev) for t in src]
    cap = TrackPool.DEVICE_CAPACITY

    def solve(carry):
        state = torch.zeros(8 + 3 * cap, dtype=torch.int32, device=dev)
        state[0] = -1
        det = (det_boxes, torch.full((n_det,), 0.9, device=dev), torch.full((n_det,), -1, dtype=torch.int64, device=dev),
               torch.ones(n_det, dtype=torch.int64, device=dev))
        nz = torch.full((n_det,) + row_shape, -7.0, device=dev)
        nsr = torch.full((n_det, 4), -7.0, device=dev)
        c = None
        if carry:
            c = ([t.data_ptr() for t in src], row0, D, guess, nz.data_ptr(), nsr.data_ptr(), rf)
        fbuf, ibuf, rec, M = ops.track_solve(det, None, 1.0, (0.4, 0.6, 0.4), 0.5, 3, state, cap, carry=c)
        torch.cuda.synchronize()
        return fbuf.clone(), ibuf.clone(), rec.cpu().numpy().copy(), M, nz, nsr, state.cpu()
    f0, i0, r0, M, nz0, nsr0, st0 = solve(False)
    f1, i1, r1, _, nz1, nsr1, st1 = solve(True)
    K, A = int(r0[0]), int(r0[1])
    assert K == n_det and A == n_det and M == n_det          # five far-apart detections start five tracks
    # (record and state are compared where the launch defines them: header, kept rows, final ids, active ids, the id table)
    na = int(r0[4])
    assert na == n_det and int(r0[5]) == 0
    for lo, n in ((0, 8), (8, K), (8 + M, K), (8 + 2 * M, A), (8 + 3 * M, na)):
        assert np.array_equal(r0[lo:lo + n], r1[lo:lo + n]), lo
    assert torch.equal(st0[:5], st1[:5]) and torch.equal(st0[8:8 + na], st1[8:8 + na])
    # the plain launch's outputs, untouched: kept rows, active rows (the buffers' other words are uninitialised)
    for off, n in ((0, 4 * K), (4 * M, 4 * A), (8 * M, K), (9 * M, A)):
        assert torch.equal(f0[off:off + n], f1[off:off + n])
    for off, n in ((0, K), (M, K), (2 * M, A), (3 * M, A)):
        assert torch.equal(i0[off:off + n], i1[off:off + n])
    # (capacity M = 5 rows: behind five active rows there is no room — nothing may be written beyond the buffers; the rows
    # that fit are checked with a guess below the count, where the templates land on rows the extraction overwrites later)
    want_z, want_sr = torch.full_like(nz1, -7.0), torch.full_like(nsr1, -7.0)
    for j in range(D):
        if guess + j < M:
            want_z[guess + j] = src[0][row0 + j]
            want_sr[guess + j] = src[2][row0 + j]
    assert torch.equal(nz1, want_z) and torch.equal(nsr1, want_sr)
    assert torch.equal(nz0, torch.full_like(nz0, -7.0))


@pytest.mark.gpu
def test_solver_launch_appends_the_dormant_rows_behind_the_active_count():
    """The same with room behind the active rows: eight detections of which the NMS keeps five (three sit on top of
    others), none below the start threshold -> five active rows, capacity eight: the three carried rows' boxes / ids /
    labels / scores stand at rows 5 .. 7 of the act_* arrays, their templates / search regions at the guessed rows."""
    import siammot_amd.ops as ops
    from siammot_amd.solver import TrackPool
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(6)
    n_src, D, row0 = 9, 3, 4
    row_shape = (128, 7, 7)
    rf = int(np.prod(row_shape))
    b = [[10 + 200 * i, 20, 90 + 200 * i, 180] for i in range(5)] + [[12 + 200 * i, 22, 92 + 200 * i, 182] for i in range(3)]
    det_boxes = torch.tensor(b, dtype=torch.float32, device=dev)
    scores = torch.tensor([0.9] * 5 + [0.8] * 3, device=dev)
    src = [torch.randn((n_src,) + row_shape, generator=g), torch.randn(n_src, 4, generator=g), torch.randn(n_src, 4, generator=g),
           torch.randint(0, 1 << 40, (n_src,), generator=g), torch.randint(0, 9, (n_src,), generator=g), torch.rand(n_src, generator=g)]
    src = [t.to(dev) for t in src]
    cap = TrackPool.DEVICE_CAPACITY
    for guess in (5, 4):
        state = torch.zeros(8 + 3 * cap, dtype=torch.int32, device=dev)
        state[0] = -1
        det = (det_boxes, scores.clone(), torch.full((8,), -1, dtype=torch.int64, device=dev), torch.ones(8, dtype=torch.int64, device=dev))
        nz = torch.full((8,) + row_shape, -7.0, device=dev)
        nsr = torch.full((8, 4), -7.0, device=dev)
        fbuf, ibuf, rec, M = ops.track_solve(det, None, 1.0, (0.4, 0.6, 0.4), 0.5, 3, state, cap,
                                             carry=([t.data_ptr() for t in src], row0, D, guess, nz.data_ptr(), nsr.data_ptr(), rf))
        torch.cuda.synchronize()
        r = rec.cpu().numpy()
        K, A = int(r[0]), int(r[1])
        assert (K, A, M) == (5, 5, 8) and int(state[4]) == 5
        act_boxes, act_scores = fbuf[4 * M:8 * M].view(M, 4), fbuf[9 * M:10 * M]
        act_ids, act_labels = ibuf[2 * M:3 * M], ibuf[3 * M:4 * M]
        sel = slice(row0, row0 + D)
        assert torch.equal(act_boxes[A:A + D], src[1][sel]) and torch.equal(act_scores[A:A + D], src[5][sel])
        assert torch.equal(act_ids[A:A + D], src[3][sel]) and torch.equal(act_labels[A:A + D], src[4][sel])
        assert act_ids[:A].tolist() == [0, 1, 2, 3, 4]
        want_z, want_sr = torch.full_like(nz, -7.0), torch.full_like(nsr, -7.0)
        want_z[guess:guess + D], want_sr[guess:guess + D] = src[0][sel], src[2][sel]
        assert torch.equal(nz, want_z) and torch.equal(nsr, want_sr)


@pytest.mark.gpu
@pytest.mark.parametrize("native", [True, False])
def test_lean_step_equals_general_path(native):
    """TrackingLoop's lean per-frame step (raw tensors, masked template extraction before the synchronisation, lazy
    cache) against the general path (TrackHead / EMM modules, BoxLists, eager cache) on the same sequence with the
    real HIP head: outputs, track memory and pool must be identical frame by frame — including frames in which
    tracks go dormant, are carried in the memory, are resumed and expire."""
    import golden_inputs as gi
    from fake_tracker import detections
    from siammot_amd.config import get_default_cfg
    from siammot_amd.track_head import build_tracking_loop
    dev = torch.device("cuda:0")
    cfg = get_default_cfg(channels=32)
    cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 3
    # the freshly initialised head scores every track ~0.5: thresholds at 0.5 make tracks go dormant, come back, expire
    cfg.MODEL.TRACK_HEAD.TRACK_THRESH = 0.5
    cfg.MODEL.TRACK_HEAD.RESUME_TRACK_THRESH = 0.5
    loops = [build_tracking_loop(cfg, device=dev, refine_tracks=False) for _ in range(2)]
    with torch.no_grad():
        for name in ("cls", "center", "reg"):
            getattr(loops[0].track.tracker.predictor, name).weight.mul_(20.0)
    loops[1].track.tracker.load_state_dict(loops[0].track.tracker.state_dict())
    loops[1]._lean_ok = lambda d: False                     # general path
    import siammot_amd.ops as ops_
    carried0 = ops_.MEMORY_CARRY["launched"]
    lean_frames = []
    loops[0].native_frame = native              # one library call per frame (smot_track_frame_fwd) / composed in Python
    which = "_step_native" if native else "_step_lean"
    real = getattr(loops[0], which)
    setattr(loops[0], which, lambda f, d: (lean_frames.append(1), real(f, d))[1])
    shapes = gi.feature_shapes((1280, 704), 32)
    rs_f = np.random.RandomState(9)
    rs = [np.random.RandomState(5), np.random.RandomState(5)]
    seen_dormant = False
    for f in range(16):
        feats = tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes)
        outs = [lp(feats, detections(r, f).to(dev)) for lp, r in zip(loops, rs)]
        a, b = outs
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), "frame %d" % f
        assert torch.equal(a.get_field("scores"), b.get_field("scores"))
        ma, mb = loops[0].track_memory, loops[1].track_memory
        assert torch.equal(ma[0], mb[0]) and torch.equal(ma[1][0].bbox, mb[1][0].bbox), "memory, frame %d" % f
        assert torch.equal(ma[2][0].bbox, mb[2][0].bbox)
        assert torch.equal(ma[2][0].get_field("ids"), mb[2][0].get_field("ids"))
        for fld in ("labels", "scores"):                    # (dormant rows are copied on the device on the lean paths)
            assert torch.equal(ma[2][0].get_field(fld), mb[2][0].get_field(fld)), "memory %s, frame %d" % (fld, f)
            assert torch.equal(ma[1][0].get_field(fld), mb[1][0].get_field(fld)), "memory sr %s, frame %d" % (fld, f)
        assert [int(t) for t in ma[2][0].host_ids] == mb[2][0].get_field("ids").tolist()
        pa, pb = loops[0].solver.track_pool, loops[1].solver.track_pool
        assert pa.get_active_ids() == pb.get_active_ids() and pa._dormant_ids == pb._dormant_ids
        assert pa._kill_ids == pb._kill_ids and pa._max_id == pb._max_id
        ca, cb = pa.get_cache(), pb.get_cache()
        for tid in pa.get_dormant_ids():
            if tid in cb:
                assert tid in ca and torch.equal(ca[tid][0], cb[tid][0]) and torch.equal(ca[tid][1].bbox, cb[tid][1].bbox)
        seen_dormant |= bool(pa.get_dormant_ids())
    assert len(lean_frames) == 16 and seen_dormant and loops[0].solver.track_pool._kill_ids
    import siammot_amd.ops as ops_
    assert ops_.MEMORY_CARRY["launched"] > carried0, "no frame copied its dormant rows on the device"


@pytest.mark.gpu
def test_frames_that_leave_no_track_are_followed_by_calls_without_a_head():
    """the default tracking loop prepares the NEXT call's head launch at the end of every frame with rows;
    a frame whose detections all stay below the start threshold leaves no track, the next call has an empty memory and no
    head — the row count the solver sees must be that call's own (it used to keep the guessed count of the head that never
    ran: one row too many, a record laid out for another count).  Video start and a stretch in the middle, real HIP head,
    against the general path: outputs, pool and memory identical in every frame."""
    import golden_inputs as gi
    from fake_tracker import detections
    from siammot_amd.config import get_default_cfg
    from siammot_amd.track_head import build_tracking_loop
    dev = torch.device("cuda:0")
    cfg = get_default_cfg(channels=32)
    cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 1
    loops = [build_tracking_loop(cfg, device=dev, refine_tracks=False) for _ in range(2)]
    loops[1].track.tracker.load_state_dict(loops[0].track.tracker.state_dict())
    loops[1]._lean_ok = lambda d: False                     # general path
    shapes = gi.feature_shapes((1280, 704), 32)
    rs_f = np.random.RandomState(19)
    rs = [np.random.RandomState(6), np.random.RandomState(6)]
    empties = 0
    for f in range(14):
        feats = tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes)
        outs = []
        for lp, r in zip(loops, rs):
            d = detections(r, f)
            if f < 3 or 7 <= f < 11:
                d.add_field("scores", torch.full_like(d.get_field("scores"), 0.05))       # nothing starts; tracks starve
                d = d[:0] if f in (8, 9) else d                                           # ... and frames with no row at all
            outs.append(lp(feats, d.to(dev)))
        a, b = outs
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), "frame %d" % f
        assert torch.equal(a.get_field("scores"), b.get_field("scores"))
        pa, pb = loops[0].solver.track_pool, loops[1].solver.track_pool
        assert pa.get_active_ids() == pb.get_active_ids() and pa._dormant_ids == pb._dormant_ids and pa._max_id == pb._max_id
        ma, mb = loops[0].track_memory, loops[1].track_memory
        assert len(ma[2][0]) == len(mb[2][0])
        if len(mb[2][0]):
            assert torch.equal(ma[0], mb[0]) and torch.equal(ma[2][0].bbox, mb[2][0].bbox), "memory, frame %d" % f
        else:
            empties += 1
    assert empties >= 3 and loops[0].solver.track_pool._max_id >= 5, (empties, loops[0].solver.track_pool._max_id)


@pytest.mark.gpu
def test_next_frame_shown_equals_the_synchronous_loop_on_random_traffic():
    """``TrackingLoop.forward(..., next_features=)`` launches the next frame's head a call early, guessing that the track
    count holds.  Against a second loop that is never shown the next frame, on 60 frames of random traffic (detections
    drop out, false positives start tracks, tracks go dormant, resume and expire — the guess fails often): outputs
    identical in every frame, memory and pool identical whenever they are compared (every fifth frame: looking at the
    memory builds it, which makes the loop discard that frame's speculative head — exercised on purpose), and both
    outcomes of the guess occur."""
    import golden_inputs as gi
    import siammot_amd.ops as ops_
    from fake_tracker import detections
    from siammot_amd.config import get_default_cfg
    from siammot_amd.track_head import build_tracking_loop
    dev = torch.device("cuda:0")
    cfg = get_default_cfg(channels=32)
    # (speculation needs a memory that is the extraction's own output: frames without dormant tracks.  A low track
    # threshold and one dormant frame keep such frames common while tracks still start, get lost, come back and expire)
    cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 1
    cfg.MODEL.TRACK_HEAD.TRACK_THRESH = float(__import__("os").environ.get("SMOT_TEST_TT", "0.35"))
    cfg.MODEL.TRACK_HEAD.RESUME_TRACK_THRESH = 0.5
    loops = [build_tracking_loop(cfg, device=dev, refine_tracks=False) for _ in range(2)]
    with torch.no_grad():
        for name in ("cls", "center", "reg"):
            getattr(loops[0].track.tracker.predictor, name).weight.mul_(20.0)
    loops[1].track.tracker.load_state_dict(loops[0].track.tracker.state_dict())
    for lp in loops:
        lp.native_frame = True
    shapes = gi.feature_shapes((1280, 704), 32)
    rs_f = np.random.RandomState(9)
    frames = 60
    feats = [tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes) for _ in range(6)]
    rs = [np.random.RandomState(5), np.random.RandomState(5)]
    ops_.SPECULATION.clear()
    for f in range(frames):
        fa, fb = feats[f % 6], feats[(f + 1) % 6]
        a = loops[0](fa, detections(rs[0], f % 40).to(dev), next_features=fb if f + 1 < frames else None)
        b = loops[1](fa, detections(rs[1], f % 40).to(dev))
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), "frame %d" % f
        assert torch.equal(a.get_field("scores"), b.get_field("scores")), "frame %d" % f
        pa, pb = loops[0].solver.track_pool, loops[1].solver.track_pool
        assert pa.get_active_ids() == pb.get_active_ids() and pa._dormant_ids == pb._dormant_ids and pa._max_id == pb._max_id
        if f % 5 == 4:
            ma, mb = loops[0].track_memory, loops[1].track_memory
            assert torch.equal(ma[0], mb[0]) and torch.equal(ma[1][0].bbox, mb[1][0].bbox) and torch.equal(ma[2][0].bbox, mb[2][0].bbox)
            assert torch.equal(ma[2][0].get_field("ids"), mb[2][0].get_field("ids"))
    sp = dict(ops_.SPECULATION)
    print("speculative heads on random traffic:", sp)
    assert sp.get("launched", 0) >= 4 and sp.get("used", 0) >= 1 and sp.get("discarded", 0) >= 1, sp
    assert sp.get("used", 0) + sp.get("discarded", 0) == sp["launched"]

