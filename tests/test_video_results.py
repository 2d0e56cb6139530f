"""Frame iterators, result wire format and the JSON result cache: CPU tests of the host logic and
one GPU test that feeds a device BoxList coming out of the HIP head through the result path with a single
device->host copy."""
import json
import os

import numpy as np
import pytest
import torch


def _write_frames(folder, n, hw=(48, 64)):
    from PIL import Image
    rs = np.random.RandomState(3)
    frames = []
    for i in range(n):
        # smooth content survives JPEG nearly unchanged; the index is encoded in the mean brightness
        base = np.full(hw + (3,), 40 + 20 * i, dtype=np.uint8)
        base[:, : hw[1] // 2, 0] += 10
        frames.append(base)
        Image.fromarray(base, "RGB").save(os.path.join(folder, "%06d.jpg" % i), quality=95)
    (open(os.path.join(folder, "notes.txt"), "w")).write("not a frame")
    return frames


def test_image_folder_iterator_contract(tmp_path):
    """demos/video_iterator.py:99-125: sorted *.jpg, (frame_idx, RGB uint8 [H,W,3]), len = frames produced,
    frame_idxs selects and sorts."""
    from siammot_amd.video import ImageFolderIterator, build_video_iterator
    frames = _write_frames(str(tmp_path), 5)
    it = build_video_iterator(str(tmp_path))
    assert isinstance(it, ImageFolderIterator) and len(it) == 5 and it.video_len() == 5
    got = list(it())
    assert [i for i, _ in got] == [0, 1, 2, 3, 4]
    for (i, f), ref in zip(got, frames):
        assert f.dtype == np.uint8 and f.shape == ref.shape
        assert np.abs(f.astype(int) - ref.astype(int)).max() <= 6        # JPEG round trip
        assert abs(float(f.mean()) - float(ref.mean())) < 1.0
    sub = ImageFolderIterator(str(tmp_path), frame_idxs=[3, 1])
    assert len(sub) == 2 and [i for i, _ in sub()] == [1, 3]
    assert list(ImageFolderIterator(str(tmp_path), frame_idxs=[])()) == []
    with pytest.raises(FileNotFoundError):
        ImageFolderIterator(str(tmp_path / "missing"))


def test_container_readers_fail_loudly_without_a_decoder(tmp_path):
    from siammot_amd import video
    p = tmp_path / "clip.mp4"
    p.write_bytes(b"\x00" * 16)
    for kind in ("decord", "cv2"):
        try:
            __import__(kind)
            continue                                      # decoder present on this box: nothing to assert
        except ImportError:
            pass
        with pytest.raises(RuntimeError, match="not installed"):
            video.build_video_iterator(str(p), video_decode=kind)


# ---- container readers against decoder stand-ins (decord / cv2 / ffmpeg-python are not installable here) ----------------
_FAKE_VIDEOS = {}          # path -> dict(frames=[T,H,W,3] RGB uint8, rotate=degrees or None, fail_at=frame index or None)


def _install_fake_decoders(monkeypatch):
    """Modules ``decord``, ``cv2`` and ``ffmpeg`` with exactly the calls demos/video_iterator.py makes, backed by arrays:
    ``decord.VideoReader(path, ctx=cpu(0))`` (len, [i].asnumpy() -> RGB), ``cv2.VideoCapture(path)`` (isOpened, get(
    CAP_PROP_FRAME_COUNT), set(CAP_PROP_POS_FRAMES, i), read() -> (ok, BGR)), ``ffmpeg.probe(path)`` -> streams[0].tags."""
    import sys
    import types

    class _Frame(object):
        def __init__(self, a):
            self.a = a

        def asnumpy(self):
            return self.a.copy()

    class VideoReader(object):
        def __init__(self, path, ctx=None):
            self.v = _FAKE_VIDEOS[path]

        def __len__(self):
            return len(self.v["frames"])

        def __getitem__(self, i):
            return _Frame(self.v["frames"][int(i)])

    class VideoCapture(object):
        def __init__(self, path):
            self.v = _FAKE_VIDEOS.get(path)
            self.pos = 0

        def isOpened(self):
            return self.v is not None

        def get(self, prop):
            assert prop == cv2.CAP_PROP_FRAME_COUNT
            return float(len(self.v["frames"]))

        def set(self, prop, value):
            assert prop == cv2.CAP_PROP_POS_FRAMES
            self.pos = int(value)

        def read(self):
            if self.pos >= len(self.v["frames"]) or self.pos == self.v.get("fail_at"):
                return False, None
            f = self.v["frames"][self.pos][:, :, ::-1].copy()         # OpenCV hands out BGR
            self.pos += 1
            return True, f
    decord = types.ModuleType("decord")
    decord.VideoReader, decord.cpu = VideoReader, (lambda i=0: ("cpu", i))
    cv2 = types.ModuleType("cv2")
    cv2.VideoCapture, cv2.CAP_PROP_FRAME_COUNT, cv2.CAP_PROP_POS_FRAMES = VideoCapture, 7, 1
    ffmpeg = types.ModuleType("ffmpeg")

    def probe(path):
        rot = _FAKE_VIDEOS[path].get("rotate")
        return {"streams": [{"tags": ({"rotate": str(rot)} if rot is not None else {"language": "und"})}]}
    ffmpeg.probe = probe
    for name, mod in (("decord", decord), ("cv2", cv2), ("ffmpeg", ffmpeg)):
        monkeypatch.setitem(sys.modules, name, mod)


def _fake_clip(path, n=7, hw=(6, 10), rotate=None, fail_at=None, seed=0):
    rs = np.random.RandomState(seed)
    _FAKE_VIDEOS[path] = dict(frames=rs.randint(0, 256, (n,) + hw + (3,)).astype(np.uint8), rotate=rotate, fail_at=fail_at)
    return _FAKE_VIDEOS[path]["frames"]


@pytest.mark.parametrize("rotate", [None, 0, 90, 180, 270])
def test_container_readers_on_decoder_stand_ins(monkeypatch, rotate):
    """``DecordVideoIterator`` / ``CV2VideoIterator`` (demos/video_iterator.py:9-82) executed — the decoder libraries are
    absent from this image, so against stand-ins with the calls the reference makes: frame ids, RGB order (OpenCV's BGR
    flipped), the rotation tag of the container applied as the reference applies it, ``frame_idxs`` selected and sorted,
    a failing read ends the stream, ``build_video_iterator`` picks the reader by ``video_decode``."""
    from siammot_amd import video
    _install_fake_decoders(monkeypatch)
    frames = _fake_clip("clip.mp4", rotate=rotate)
    want = [np.rot90(f, k=(-(rotate // 90)) % 4) if rotate else f for f in frames]
    for kind, cls in (("decord", video.DecordVideoIterator), ("cv2", video.CV2VideoIterator)):
        it = video.build_video_iterator("clip.mp4", video_decode=kind)
        assert type(it) is cls and len(it) == 7
        got = list(it())
        assert [i for i, _ in got] == list(range(7))
        for (_, f), w in zip(got, want):
            assert f.dtype == np.uint8 and np.array_equal(f, w)
        sub = cls("clip.mp4", frame_idxs=[5, 2, 3])
        assert len(sub) == 3 and [i for i, _ in sub()] == [2, 3, 5]
        assert all(np.array_equal(f, want[i]) for i, f in sub())
    assert video.DecordVideoIterator("clip.mp4").video_len() == 7
    _fake_clip("broken.mp4", fail_at=4)
    assert [i for i, _ in video.CV2VideoIterator("broken.mp4")()] == [0, 1, 2, 3]        # read() fails: the stream ends
    with pytest.raises(AssertionError, match="Cannot open"):
        video.CV2VideoIterator("missing.mp4")


REFERENCE = os.environ.get("SIAMMOT_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "demos", "video_iterator.py")), reason="reference checkout not present")
def test_container_readers_equal_the_reference_iterators_on_the_same_stand_ins(monkeypatch):
    """The reference's UNMODIFIED demos/video_iterator.py imported on the same decoder stand-ins (its top-level ``import
    decord / cv2 / ffmpeg`` resolve to them): its ``DecordVideoIterator`` / ``CV2VideoIterator`` / ``build_video_iterator``
    and this package's yield the same frame ids and the same arrays — every rotation tag, subsets, a failing read."""
    import importlib.util
    from siammot_amd import video
    _install_fake_decoders(monkeypatch)
    spec = importlib.util.spec_from_file_location("ref_video_iterator", os.path.join(REFERENCE, "demos", "video_iterator.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    checked = 0
    for k, rotate in enumerate((0, 90, 180, 270)):
        path = "clip_%d.mp4" % rotate
        _fake_clip(path, n=6, hw=(5, 9), rotate=rotate, fail_at=(4 if rotate == 180 else None), seed=10 + k)
        for kind in ("decord", "cv2"):
            for idxs in (None, [4, 0, 3]):
                mine = video.build_video_iterator(path, video_decode=kind)
                theirs = ref.build_video_iterator(path, video_decode=kind)
                if idxs is not None:
                    mine, theirs = type(mine)(path, frame_idxs=idxs), type(theirs)(path, frame_idxs=idxs)
                a, b = list(mine()), list(theirs())
                assert len(mine) == len(theirs) and [int(i) for i, _ in a] == [int(i) for i, _ in b]
                for (_, fa), (_, fb) in zip(a, b):
                    assert np.array_equal(fa, fb)
                    checked += 1
    assert checked > 60


def test_prefetch_preserves_order_and_surfaces_errors():
    from siammot_amd.video import ArrayVideoIterator, prefetch
    frames = [np.full((4, 4, 3), i, dtype=np.uint8) for i in range(7)]
    out = list(prefetch(ArrayVideoIterator(frames)(), depth=2))
    assert [i for i, _ in out] == list(range(7)) and all(int(f[0, 0, 0]) == i for i, f in out)

    def bad():
        yield 0, frames[0]
        raise ValueError("decoder died")
    g = prefetch(bad(), depth=1)
    assert next(g)[0] == 0
    with pytest.raises(ValueError, match="decoder died"):
        next(g)


def _entity(frame, tid, conf, box=(1.0, 2.0, 3.0, 4.0)):
    from siammot_amd.results import AnnoEntity
    e = AnnoEntity()
    e.bbox, e.confidence, e.labels, e.id, e.frame_num, e.time = list(box), conf, {"person": conf}, tid, frame, frame / 30.0
    return e


def test_result_sample_json_cache_and_postprocess(tmp_path):
    """inferencer.py:118-153: <output_dir>/<id>.json is written once and re-used; short / unconfident tracks are
    filtered out afterwards."""
    from siammot_amd.results import ResultSample, cached_video_result, mot_challenge_rows, postprocess_tracks
    runs = []

    def run():
        runs.append(1)
        s = ResultSample("vid/a", 1920, 1080, 30.0)
        for f in range(6):
            s.add_entity(_entity(f, 0, 0.9))          # long and confident: kept
            s.add_entity(_entity(f, 1, 0.5))          # long but unconfident
            if f < 3:
                s.add_entity(_entity(f, 2, 0.95))     # confident but short
            s.add_entity(_entity(f, -1, 0.99))        # untracked detection
        return s
    out_dir = str(tmp_path / "out")
    a = cached_video_result(out_dir, "vid/a", run)
    b = cached_video_result(out_dir, "vid/a", run)
    assert runs == [1]                                # second call came from the JSON cache
    assert os.path.exists(os.path.join(out_dir, "vid/a.json"))
    assert json.load(open(os.path.join(out_dir, "vid/a.json")))["metadata"]["resolution"] == {"width": 1920, "height": 1080}
    assert [e.to_dict() for e in a.entities] == [e.to_dict() for e in b.entities] and len(b) == 6
    kept = postprocess_tracks(b, track_len=5, track_conf=0.7)
    assert sorted({e.id for e in kept.entities}) == [0] and len(kept.entities) == 6
    rows = mot_challenge_rows(kept.entities)
    assert rows[0] == "1,0,1.00,2.00,3.00,4.00,0.9000,-1,-1,-1" and len(rows) == 6


def test_boxlist_resize_equal_ratio_keeps_the_mode():
    """[UPSTREAM] BoxList.resize: with equal ratios the stored coordinates are scaled as they are (xywh boxes do
    not take the xyxy round trip and its +-1)."""
    from siammot_amd.structures import BoxList
    b = BoxList(torch.tensor([[10.0, 20.0, 30.0, 40.0]]), (100, 50), mode="xywh")
    r = b.resize((200, 100))
    assert r.mode == "xywh" and r.bbox.tolist() == [[20.0, 40.0, 60.0, 80.0]]
    r2 = b.resize((200, 50))                              # unequal ratios: through xyxy (TO_REMOVE = 1)
    assert r2.mode == "xywh" and r2.bbox.tolist() == [[20.0, 20.0, 59.0, 40.0]]


@pytest.mark.gpu
def test_device_boxlist_from_the_hip_head_to_mot_rows_with_one_host_copy(tmp_path):
    """EMM.forward on the GPU -> resize to the source frame -> boxlists_to_entities -> MOT rows, with device->host
    synchronisation counted by torch's sync debug mode: exactly one copy (the reference's version synchronises
    three times per BOX, boxlists_to_entities.py:25-33)."""
    import warnings
    import golden_inputs as gi
    from siammot_amd.config import get_default_cfg
    from siammot_amd.emm import EMM
    from siammot_amd.results import ResultSample, boxlists_to_entities, mot_challenge_rows, to_original_xywh
    from siammot_amd.structures import BoxList
    from siammot_amd.track_utils import build_track_utils
    dev = "cuda:0"
    case = gi.EMM_CASES["default"]
    inp = gi.emm_case_inputs("default")
    cfg = get_default_cfg(channels=case["channels"])
    emm = EMM(cfg, build_track_utils(cfg)).to(dev).eval()
    emm.predictor.load_state_dict({k: torch.from_numpy(v) for k, v in inp["params"].items()})
    n = len(case["boxes"])
    det = BoxList(torch.from_numpy(inp["boxes"]).to(dev), case["image_wh"], mode="xyxy")
    det.add_field("ids", torch.arange(n, device=dev))
    det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
    with torch.no_grad():
        z, sr, d = emm.extract_cache(tuple(torch.from_numpy(f).to(dev) for f in inp["features_a"]), det)
        _, result, _ = emm(tuple(torch.from_numpy(f).to(dev) for f in inp["features_b"]), d, sr, template_features=z)
    torch.cuda.synchronize()
    out = result[0]
    assert out.bbox.is_cuda
    orig_wh = (1280, 960)                                  # the source frame: 2.5x the network input, both axes
    torch.cuda.set_sync_debug_mode("warn")
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ents = boxlists_to_entities([to_original_xywh(out, orig_wh)], 7, [0.25])
        syncs = [x for x in w if "synchroniz" in str(x.message).lower()]
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert len(syncs) == 1, "expected ONE device->host copy, saw %d: %s" % (len(syncs), [str(x.message)[:80] for x in syncs])
    # contents: the reference's per-box loop on the host copy
    ref = out.to("cpu").resize(orig_wh).convert("xywh")
    assert len(ents) == n
    for j, e in enumerate(ents):
        assert e.bbox == ref.bbox[j].tolist() and e.confidence == ref.get_field("scores")[j].item()
        assert e.id == j and e.frame_num == 7 and e.time == 0.25 and e.labels == {"person": e.confidence}
    s = ResultSample("clip", *orig_wh, fps=4.0)
    for e in ents:
        s.add_entity(e)
    path = str(tmp_path / "clip.json")
    s.dump(path)
    rows = mot_challenge_rows(ResultSample.load(path).entities)
    assert len(rows) == n and rows[0].startswith("8,0,")


@pytest.mark.gpu
def test_frame_sequence_runner_over_an_image_folder(tmp_path):
    """Folder of JPEG frames -> ImageFolderIterator -> FramePreprocessor (GPU) -> synthetic detector -> TrackingLoop
    (HIP head + solver) -> ResultSample: every frame produces entities, ids persist across frames, and the JSON
    cache round-trips."""
    from siammot_amd.config import get_default_cfg
    from siammot_amd.preprocess import FramePreprocessor
    from siammot_amd.results import FrameSequenceRunner, cached_video_result
    from siammot_amd.structures import BoxList
    from siammot_amd.track_head import build_tracking_loop
    from siammot_amd.video import ImageFolderIterator
    dev = torch.device("cuda:0")
    _write_frames(str(tmp_path), 6, hw=(96, 128))
    cfg = get_default_cfg(channels=32)
    loop = build_tracking_loop(cfg, device=dev, refine_tracks=False)
    pre = FramePreprocessor(min_size=192, max_size=256, size_divisibility=32, device=dev)
    boxes = torch.tensor([[20.0, 30.0, 80.0, 150.0], [120.0, 40.0, 200.0, 170.0]], device=dev)
    g = torch.Generator().manual_seed(0)
    feats = None

    def detector(image):
        nonlocal feats
        _, h, w = image.shape
        if feats is None:
            feats = tuple(torch.randn((1, 32, h // s, w // s), generator=g).to(dev) for s in (4, 8, 16, 32, 64))
        d = BoxList(boxes.clone(), (w, h), mode="xyxy")
        d.add_field("ids", torch.full((2,), -1, dtype=torch.int64, device=dev))
        d.add_field("labels", torch.ones(2, dtype=torch.int64, device=dev))
        d.add_field("scores", torch.tensor([0.9, 0.8], device=dev))
        # (fresh tensors per call, as a detector's forward returns them: the look-ahead runner refuses re-used output buffers)
        return tuple(f.clone() for f in feats), d
    runner = FrameSequenceRunner(detector, loop, pre)
    res = cached_video_result(str(tmp_path / "out"), "folder", lambda: runner.run_video(
        "folder", ImageFolderIterator(str(tmp_path)), fps=30.0))
    assert (res.width, res.height) == (128, 96) and len(res) == 6
    per_frame = [sorted(e.id for e in res.get_entities_for_frame_num(f)) for f in range(6)]
    # frame 0 starts tracks 0 and 1 from the two detections; later frames carry the two detections (matched to a
    # track or, when the random-feature tracker loses one, started as a new id) — every box has an id >= 0
    assert per_frame[0] == [0, 1] and all(len(p) >= 2 and min(p) >= 0 for p in per_frame)
    again = cached_video_result(str(tmp_path / "out"), "folder", lambda: 1 / 0)     # served from the cache
    assert len(again.entities) == len(res.entities)
    ids = [fid for fid, _ in runner.process_frame_sequence(ImageFolderIterator(str(tmp_path), frame_idxs=[0, 2])())]
    assert ids == [0, 2]
    # the same video with the detector one frame ahead of the tracker (every tracker call is shown the next frame's
    # features: the next head is launched speculatively): the same entities, field for field
    ahead = runner.run_video("folder", ImageFolderIterator(str(tmp_path)), fps=30.0, lookahead=True)
    assert len(ahead.entities) == len(res.entities)
    for e, w in zip(ahead.entities, res.entities):
        assert (e.id, e.frame_num, e.bbox, e.confidence, e.labels) == (w.id, w.frame_num, w.bbox, w.confidence, w.labels)


def test_lookahead_refuses_a_detector_that_reuses_its_output_buffers():
    """with ``lookahead=True`` the detector runs on frame t+1 before the tracker consumes frame t; a detector
    that writes into persistent output buffers would make the tracker follow the wrong frame silently.  The runner
    compares storage addresses and raises; without look-ahead the same detector is served."""
    import numpy as np
    from siammot_amd.results import FrameSequenceRunner
    from siammot_amd.structures import BoxList

    class Loop(object):
        seen = []

        def reset(self):
            pass

        def __call__(self, feats, dets, next_features=None):
            self.seen.append(float(feats[0].flatten()[0]))
            return dets
    persistent = (torch.zeros(1, 4, 2, 2),)

    def boxes():
        d = BoxList(torch.tensor([[1.0, 2.0, 5.0, 7.0]]), (8, 8), mode="xyxy")
        d.add_field("ids", torch.tensor([-1]))
        d.add_field("labels", torch.tensor([1]))
        d.add_field("scores", torch.tensor([0.9]))
        return d

    def reusing(image):
        persistent[0].add_(1.0)
        return persistent, boxes()

    def fresh(image):
        persistent[0].add_(1.0)
        return (persistent[0].clone(),), boxes()
    frames = [(i, np.zeros((8, 8, 3), np.uint8)) for i in range(3)]
    loop = Loop()
    with pytest.raises(RuntimeError, match="re-uses its output buffers"):
        list(FrameSequenceRunner(reusing, loop, lambda f: torch.zeros(3, 8, 8)).process_frame_sequence(iter(frames), lookahead=True))
    assert len(list(FrameSequenceRunner(reusing, loop, lambda f: torch.zeros(3, 8, 8)).process_frame_sequence(iter(frames)))) == 3
    loop.seen.clear()
    persistent[0].zero_()
    out = list(FrameSequenceRunner(fresh, loop, lambda f: torch.zeros(3, 8, 8)).process_frame_sequence(iter(frames), lookahead=True))
    assert [i for i, _ in out] == [0, 1, 2] and loop.seen == [1.0, 2.0, 3.0]        # every frame tracked on ITS maps


# ---- the wire format against the reference's own code (tests/golden/results_wire.json, oracle/gen_golden_results.py) -------
def _wire_cases():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "results_wire.json")
    return json.load(open(path))


@pytest.mark.parametrize("name", ["person_720p", "two_classes_1080p"])
def test_wire_format_equals_what_the_reference_functions_return(name):
    """VERDICT r3 missing #5 / SURVEY §8(f) rank 4: ``boxlists_to_entities`` after ``resize(original).convert('xywh')``
    (inferencer.py:65-68), ``convert_given_detections_to_boxlist`` + resize to the network frame (:47-55) and
    ``_postprocess_tracks`` (:133-153) — the fixture holds what the REFERENCE's unmodified functions returned on these
    inputs (gluoncv's containers stubbed as attribute bags); every field must be equal, floats bit for bit."""
    import torch
    from siammot_amd import results as R
    from siammot_amd.structures import BoxList
    c = _wire_cases()[name]
    sample = R.ResultSample(name)
    for k, (fr, want) in enumerate(zip(c["frames"], c["entities"])):
        n = len(fr["scores"])
        bl = BoxList(torch.tensor(fr["boxes"], dtype=torch.float32).reshape(n, 4), tuple(c["net_wh"]), mode="xyxy")
        bl.add_field("scores", torch.tensor(fr["scores"], dtype=torch.float32))
        bl.add_field("labels", torch.tensor(fr["labels"], dtype=torch.int64))
        bl.add_field("ids", torch.tensor(fr["ids"], dtype=torch.int64))
        ents = R.boxlists_to_entities([R.to_original_xywh(bl, c["orig_wh"])], c["first_frame"] + k, [c["timestamps"][k]],
                                      class_table=c["class_table"])
        assert len(ents) == len(want)
        for e, w in zip(ents, want):
            assert e.bbox == w["bbox"] and e.confidence == w["confidence"] and e.labels == w["labels"]
            assert (e.id, e.frame_num, e.time) == (w["id"], w["frame_num"], w["time"])
            sample.add_entity(e)
    g = c["given"]
    given = [e for e in sample.entities if e.frame_num == c["first_frame"] + g["frame"]]
    bl = R.given_detections_to_boxlist(given, c["orig_wh"][0], c["orig_wh"][1], class_table=c["class_table"])
    assert bl.mode == g["mode"] and list(bl.size) == g["size"] and bl.bbox.tolist() == g["bbox"]
    assert bl.get_field("labels").tolist() == g["labels"] and bl.get_field("scores").tolist() == g["scores"]
    assert bl.get_field("ids").tolist() == g["ids"]
    assert bl.resize((c["net_wh"][0], c["net_wh"][1])).bbox.tolist() == g["bbox_net"]
    pp = c["postprocess"]
    kept = R.postprocess_tracks(sample, track_len=pp["track_len"], track_conf=pp["track_conf"])
    assert [[e.id, e.frame_num] for e in kept.entities] == pp["kept"]
