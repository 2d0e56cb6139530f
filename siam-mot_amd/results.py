"""Result wire format: track BoxLists -> per-box entities / MOT-challenge rows.

Mirrors siammot/utils/boxlists_to_entities.py:6-37 (``boxlists_to_entities``) and the conversion step of
siammot/engine/inferencer.py:65-70 (resize to the original frame, xywh).  The reference builds every entity
with three ``.item()`` / ``.tolist()`` calls per box — three device synchronisations per box per frame once
the BoxList lives on the GPU; here each frame's boxes, scores, labels and ids cross to the host in ONE copy.
``AnnoEntity`` is gluoncv's class in the reference (an un-vendored dependency); the fields the reference
fills are restated on a plain object.
"""
import json
import os

import torch


class AnnoEntity(object):
    """The fields ``boxlists_to_entities`` fills (gluoncv_motion_dataset.dataset.AnnoEntity [THIRD PARTY])."""
    __slots__ = ("bbox", "confidence", "labels", "id", "frame_num", "time")

    def __init__(self):
        self.bbox = None          # [x, y, w, h]
        self.confidence = None
        self.labels = None        # {class name: confidence}
        self.id = -1
        self.frame_num = None
        self.time = None

    def to_dict(self):
        return {"bbox": self.bbox, "confidence": self.confidence, "labels": self.labels, "id": self.id,
                "frame_num": self.frame_num, "time": self.time}


def boxlist_to_host(boxlist):
    """(bbox [N,4], scores [N], labels [N], ids [N]) as python lists with a single device->host copy."""
    n = len(boxlist)
    if n == 0:
        return [], [], [], []
    cols = [boxlist.bbox.to(torch.float64), boxlist.get_field("scores").to(torch.float64)[:, None],
            boxlist.get_field("labels").to(torch.float64)[:, None]]
    if boxlist.has_field("ids"):
        cols.append(boxlist.get_field("ids").to(torch.float64)[:, None])
    else:
        cols.append(torch.full((n, 1), -1.0, dtype=torch.float64, device=boxlist.bbox.device))
    host = torch.cat(cols, dim=1).cpu()          # float64 holds fp32 boxes/scores and int ids < 2^53 exactly
    bbox = host[:, 0:4].to(torch.float32).tolist()
    scores = host[:, 4].to(torch.float32).tolist()
    return bbox, scores, host[:, 5].to(torch.int64).tolist(), host[:, 6].to(torch.int64).tolist()


def boxlists_to_entities(boxlists, firstframe_idx, timestamps, class_table=None):
    """Same contract as the reference function: one entity per box, ``labels = {class_table[label-1]: score}``,
    ``id`` from the ``ids`` field (or -1), ``frame_num = firstframe_idx + i``, ``time = timestamps[i]``."""
    if not isinstance(boxlists, list):
        boxlists = [boxlists]
    if class_table is None:
        class_table = ["person"]
    entities = []
    for i, boxlist in enumerate(boxlists):
        bbox, scores, labels, ids = boxlist_to_host(boxlist)
        for j in range(len(bbox)):
            e = AnnoEntity()
            e.bbox = bbox[j]
            e.confidence = scores[j]
            e.labels = {class_table[labels[j] - 1]: scores[j]}
            e.id = ids[j]
            e.frame_num = firstframe_idx + i
            e.time = timestamps[i]
            entities.append(e)
    return entities


def given_detections_to_boxlist(entities, video_width, video_height, class_table=None, boxlist_cls=None):
    """``convert_given_detections_to_boxlist`` (boxlists_to_entities.py:40-60): cached / public detections of one frame
    (entities with xywh boxes in the ORIGINAL frame) -> an xyxy BoxList with labels (1-based index of the entity's first
    label in ``class_table``), scores and ids = -1.  The caller resizes it to the network's input frame and moves it to
    the device (inferencer.py:50-55)."""
    if boxlist_cls is None:
        from .structures import BoxList as boxlist_cls
    if class_table is None:
        class_table = ["person"]
    boxes = torch.as_tensor([e.bbox for e in entities]).reshape(-1, 4)
    labels = torch.tensor([class_table.index(list(e.labels.keys())[0]) + 1 for e in entities], dtype=torch.int64)
    scores = torch.tensor([e.confidence for e in entities])
    ids = torch.tensor([-1 for _ in entities], dtype=torch.int64)
    boxlist = boxlist_cls(boxes, [video_width, video_height], mode="xywh").convert("xyxy")
    boxlist.add_field("labels", labels)
    boxlist.add_field("scores", scores)
    boxlist.add_field("ids", ids)
    return boxlist


def to_original_xywh(boxlist, orig_wh):
    """inferencer.py:65-66 / demo_inference.py:108: back to the source frame's size, xywh mode."""
    return boxlist.resize([orig_wh[0], orig_wh[1]]).convert("xywh")


def mot_challenge_rows(entities):
    """MOTChallenge text rows ``frame, id, x, y, w, h, conf, -1, -1, -1`` (1-based frames), tracked boxes only."""
    rows = []
    for e in entities:
        if e.id < 0:
            continue
        x, y, w, h = e.bbox
        rows.append("%d,%d,%.2f,%.2f,%.2f,%.2f,%.4f,-1,-1,-1" % (e.frame_num + 1, e.id, x, y, w, h, e.confidence))
    return rows


class ResultSample(object):
    """What ``do_inference`` accumulates per video (inferencer.py:24-75: ``sample.get_copy_without_entities()`` +
    ``add_entity``) and ``DatasetInference._inference_on_video`` dumps / loads as ``<output_dir>/<sample id>.json``
    (inferencer.py:118-132).  The reference's container is gluoncv's ``DataSample`` [THIRD PARTY, un-vendored]; the
    members the reference path uses are restated: id, width / height / fps metadata, the entity list, ``dump`` /
    ``load`` as JSON, and the two queries ``_postprocess_tracks`` needs."""

    def __init__(self, sample_id, width=None, height=None, fps=None):
        self.id = sample_id
        self.width, self.height, self.fps = width, height, fps
        self.entities = []

    def add_entity(self, entity):
        self.entities.append(entity)

    def get_copy_without_entities(self):
        return ResultSample(self.id, self.width, self.height, self.fps)

    def get_entities_with_id(self, track_id):
        return [e for e in self.entities if e.id == track_id]

    def get_entities_for_frame_num(self, frame_num):
        return [e for e in self.entities if e.frame_num == frame_num]

    def __len__(self):
        return len({e.frame_num for e in self.entities})

    def to_dict(self):
        return {"id": self.id, "metadata": {"resolution": {"width": self.width, "height": self.height},
                                            "fps": self.fps},
                "entities": [e.to_dict() for e in self.entities]}

    def dump(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(self.to_dict(), f)
        os.replace(tmp, path)                 # a crashed run never leaves a half-written cache behind

    @classmethod
    def load(cls, path):
        with open(path) as f:
            d = json.load(f)
        res = d.get("metadata", {}).get("resolution", {})
        s = cls(d["id"], res.get("width"), res.get("height"), d.get("metadata", {}).get("fps"))
        for ed in d["entities"]:
            e = AnnoEntity()
            for k in AnnoEntity.__slots__:
                setattr(e, k, ed.get(k))
            s.entities.append(e)
        return s


def cached_video_result(output_dir, sample_id, run):
    """``DatasetInference._inference_on_video`` (inferencer.py:118-132): load ``<output_dir>/<id>.json`` if it
    exists, otherwise call ``run()`` (-> ResultSample) and dump its result there."""
    cache_path = os.path.join(output_dir, "{}.json".format(sample_id))
    if os.path.exists(cache_path):
        return ResultSample.load(cache_path)
    result = run()
    result.dump(cache_path)
    return result


def postprocess_tracks(tracks, track_len=5, track_conf=0.7):
    """``DatasetInference._postprocess_tracks`` (inferencer.py:134-153): keep tracks that last at least
    ``track_len`` frames with a mean confidence of at least ``track_conf``."""
    import numpy as np
    ids = set()                                   # (a set, iterated as such: the reference's output order, inferencer.py:139-146)
    for e in tracks.entities:
        if e.id not in ids and e.id >= 0:
            ids.add(e.id)
    out = tracks.get_copy_without_entities()
    for i in ids:
        ents = tracks.get_entities_with_id(i)
        conf = np.mean([e.confidence for e in ents])
        if len(ents) >= track_len and conf >= track_conf:
            for e in ents:
                out.add_entity(e)
    return out


def _storage_overlap(feats_a, dets_a, feats_b, dets_b):
    """True when any tensor of frame b lives in the storage of a tensor of frame a (a detector re-using its outputs)."""
    def ptrs(feats, dets):
        out = set()
        for f in (feats.values() if hasattr(feats, "values") else feats):       # (a dict of FPN maps iterates its KEYS)
            if isinstance(f, torch.Tensor) and f.numel():
                out.add(f.untyped_storage().data_ptr())
        fields = getattr(dets, "extra_fields", None)
        for t in [getattr(dets, "bbox", None)] + (list(fields.values()) if isinstance(fields, dict) else []):
            if isinstance(t, torch.Tensor) and t.numel():
                out.add(t.untyped_storage().data_ptr())
        return out
    return bool(ptrs(feats_a, dets_a) & ptrs(feats_b, dets_b))


class FrameSequenceRunner(object):
    """The per-video loop of the reference's entry points (demos/demo_inference.py:94-122 ``process`` /
    ``process_frame_sequence``; engine/inferencer.py:24-75 ``do_inference``) around this repository's pieces:

        frame (RGB uint8) -> FramePreprocessor (GPU) -> ``detector(image) -> (FPN features, detections BoxList)``
        -> TrackingLoop (HIP head + solver + track memory) -> resize to the source frame, xywh -> entities.

    ``detector`` is the PyTorch-ROCm backbone + RPN + box head (outside this repository's scope); anything with
    that call signature works.  Per frame the results cross to the host in one copy (``boxlist_to_host``)."""

    def __init__(self, detector, tracking_loop, preprocessor, class_table=None):
        self.detector = detector
        self.loop = tracking_loop
        self.preprocess = preprocessor
        self.class_table = class_table

    def process(self, frame):
        orig_h, orig_w = int(frame.shape[0]), int(frame.shape[1])
        image = self.preprocess(frame)
        features, detections = self.detector(image)
        out = self.loop(features, detections)
        return to_original_xywh(out, (orig_w, orig_h))

    def process_frame_sequence(self, frame_iterator, lookahead=False):
        """``frame_iterator``: what calling a video iterator returns.  Yields ``(frame_id, BoxList)`` like the
        reference; the track pool is reset first (rcnn.py:37-39).

        ``lookahead``: the detector runs one frame ahead of the tracker and every tracker call is shown the next frame's
        feature maps (``TrackingLoop.forward(..., next_features=)``: the next frame's head is enqueued behind this frame's
        extraction while the host still waits for this frame's record).  Same results frame for frame; each frame's
        result arrives one detector pass later.  The detector must return FRESH tensors per call: one that writes into
        persistent output buffers (graph-captured, pre-allocated FPN outputs) would overwrite the pending frame's maps
        and detections before the tracker consumes them — detected through the storage addresses and refused
        (RuntimeError): run such a detector with ``lookahead=False``."""
        self.loop.reset()
        if not lookahead:
            for frame_id, frame in frame_iterator:
                yield frame_id, self.process(frame)
            return
        pending = None                      # (frame_id, (w, h), features, detections) of the frame the tracker takes next
        for frame_id, frame in frame_iterator:
            size = (int(frame.shape[1]), int(frame.shape[0]))
            features, detections = self.detector(self.preprocess(frame))
            if pending is not None:
                pid, psize, pf, pd = pending
                if _storage_overlap(pf, pd, features, detections):
                    # the pending frame's maps / detections have just been overwritten: nothing correct can be produced
                    raise RuntimeError(
                        "FrameSequenceRunner(lookahead=True): the detector returned frame %r in the storage of frame %r, "
                        "which the tracker had not consumed yet — it re-uses its output buffers; call it with "
                        "lookahead=False or make it return fresh tensors" % (frame_id, pid))
                yield pid, to_original_xywh(self.loop(pf, pd, next_features=features), psize)
            pending = (frame_id, size, features, detections)
        if pending is not None:
            pid, psize, pf, pd = pending
            yield pid, to_original_xywh(self.loop(pf, pd), psize)

    def run_video(self, sample_id, video_iterator, fps=None, prefetch_depth=2, lookahead=False):
        """-> ResultSample with one entity per tracked / detected box per frame (do_inference).  ``lookahead``: see
        ``process_frame_sequence``."""
        from .video import prefetch
        result = [None]

        def frames():
            for frame_id, frame in prefetch(video_iterator(), prefetch_depth):
                if result[0] is None:
                    result[0] = ResultSample(sample_id, int(frame.shape[1]), int(frame.shape[0]), fps)
                yield frame_id, frame
        for frame_id, boxes in self.process_frame_sequence(frames(), lookahead=lookahead):
            t = frame_id / fps if fps else None
            for e in boxlists_to_entities([boxes], frame_id, [t], self.class_table):
                result[0].add_entity(e)
        return result[0] if result[0] is not None else ResultSample(sample_id, None, None, fps)
