#!/usr/bin/env python
"""Generate ``tests/golden/results_wire.json`` by running the REFERENCE's own result wire-format code on CPU (VERDICT r3
missing #5: the format was restated only).

Imported UNMODIFIED from /root/reference:
    siammot/utils/boxlists_to_entities.py   boxlists_to_entities, convert_given_detections_to_boxlist
and, as an unbound method on a plain namespace (its class needs gluoncv's dataset machinery to be constructed):
    siammot/engine/inferencer.py::DatasetInference._postprocess_tracks
The absent third-party symbols are satisfied by: the oracle's own BoxList (oracle/ref_structures.py) and a stand-in for
``gluoncv.torch.data.gluoncv_motion_dataset.dataset.AnnoEntity`` / ``DataSample`` — attribute bags with the four
container methods the reference calls (``add_entity``, ``get_copy_without_entities``, ``get_entities_with_id``,
``entities``); gluoncv 0.9's classes are exactly that for the purposes of these functions (their JSON schema is NOT
exercised here and stays unpinned: DESIGN.md §4).

What is stored, per case: the inputs (boxes / scores / labels / ids per frame, frame sizes, first frame index, time
stamps, class table) and what the reference returned — every entity's bbox / confidence / labels / id / frame_num / time
after ``do_inference``'s own ``resize([w, h]).convert('xywh')`` (inferencer.py:65-66), the BoxList that
``convert_given_detections_to_boxlist`` builds back from them (+ its resize to the network input, inferencer.py:52-54), and
the ids that survive ``_postprocess_tracks`` (inferencer.py:133-153).

Usage:  python oracle/gen_golden_results.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("SIAMMOT_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle.ref_structures import BoxList                    # noqa: E402


class AnnoEntity(object):
    """Stand-in for gluoncv's AnnoEntity: an attribute bag (the reference assigns bbox / confidence / labels / id / frame_num /
    time and reads them back)."""

    def __init__(self, time=None, id=None):
        self.time, self.id = time, id
        self.frame_num = None
        self.confidence = 1.0
        self.labels = None
        self.bbox = None


class DataSample(object):
    """Stand-in for gluoncv's DataSample: the four container operations ``_postprocess_tracks`` uses."""

    def __init__(self, sample_id):
        self.id = sample_id
        self.entities = []

    def add_entity(self, e):
        self.entities.append(e)

    def get_copy_without_entities(self):
        return DataSample(self.id)

    def get_entities_with_id(self, track_id):
        return [e for e in self.entities if e.id == track_id]


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    mod("maskrcnn_benchmark")
    mod("maskrcnn_benchmark.structures")
    mod("maskrcnn_benchmark.structures.bounding_box", BoxList=BoxList)
    for n in ("gluoncv", "gluoncv.torch", "gluoncv.torch.data", "gluoncv.torch.data.gluoncv_motion_dataset"):
        mod(n)
    mod("gluoncv.torch.data.gluoncv_motion_dataset.dataset", AnnoEntity=AnnoEntity, DataSample=DataSample)


def load(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def postprocess_source():
    """The reference's ``_postprocess_tracks`` as a plain function: its text is cut out of inferencer.py and compiled here,
    because importing the module needs gluoncv / tqdm dataset code (the body itself touches numpy and the sample only)."""
    src = open(os.path.join(REFERENCE, "siammot", "engine", "inferencer.py")).read().splitlines()
    i0 = next(i for i, l in enumerate(src) if l.strip().startswith("def _postprocess_tracks"))
    i1 = next(i for i in range(i0 + 1, len(src)) if src[i].startswith("    def "))
    body = "\n".join(l[4:] for l in src[i0:i1])                       # de-indent the method
    body = body.replace("(self, tracks: DataSample)", "(self, tracks)")
    ns = {"np": np}
    exec(compile(body, "inferencer.py:_postprocess_tracks", "exec"), ns)
    return ns["_postprocess_tracks"]


def main():
    install_stubs()
    b2e = load("ref_boxlists_to_entities", os.path.join(REFERENCE, "siammot", "utils", "boxlists_to_entities.py"))
    post = postprocess_source()
    rs = np.random.RandomState(17)
    cases = {}
    for name, (net_wh, orig_wh, table, frames) in {
            "person_720p": ((1280, 704), (1280, 720), None, 7),
            "two_classes_1080p": ((1280, 704), (1920, 1080), ["person", "vehicle"], 6)}.items():
        nclass = 1 if table is None else len(table)
        first, stamps = 40, [round(40 / 30.0 + k / 30.0, 6) for k in range(frames)]
        sample = DataSample(name)
        frames_in, ents_out = [], []
        for k in range(frames):
            n = int(rs.randint(0, 6)) if k != 2 else 0                  # an empty frame among them
            xy = rs.uniform(0, [net_wh[0] - 120, net_wh[1] - 200], (n, 2))
            wh = rs.uniform(20, [120, 200], (n, 2))
            boxes = np.concatenate((xy, xy + wh), 1).astype(np.float32)
            scores = rs.uniform(0.05, 0.99, n).astype(np.float32)
            labels = rs.randint(1, nclass + 1, n).astype(np.int64)
            ids = np.where(rs.rand(n) < 0.75, rs.randint(0, 4, n), -1).astype(np.int64)
            bl = BoxList(torch.from_numpy(boxes.copy()), net_wh, mode="xyxy")
            bl.add_field("scores", torch.from_numpy(scores.copy()))
            bl.add_field("labels", torch.from_numpy(labels.copy()))
            bl.add_field("ids", torch.from_numpy(ids.copy()))
            out = bl.resize([orig_wh[0], orig_wh[1]]).convert("xywh")                     # inferencer.py:65-66
            ents = b2e.boxlists_to_entities([out], first + k, [stamps[k]], class_table=table)    # :68
            for e in ents:
                sample.add_entity(e)
            frames_in.append(dict(boxes=boxes.tolist(), scores=scores.tolist(), labels=labels.tolist(), ids=ids.tolist()))
            ents_out.append([dict(bbox=[float(v) for v in e.bbox], confidence=float(e.confidence), labels=e.labels, id=int(e.id),
                                  frame_num=int(e.frame_num), time=e.time) for e in ents])
        # the way back: cached detections -> BoxList in the network's frame (inferencer.py:47-55)
        det_frame = 1
        given = [e for e in sample.entities if e.frame_num == first + det_frame]
        g = b2e.convert_given_detections_to_boxlist(given, orig_wh[0], orig_wh[1], class_table=table)
        g_net = g.resize((net_wh[0], net_wh[1]))
        kept = post(types.SimpleNamespace(_track_len=3, _track_conf=0.3), sample)          # inferencer.py:133-153
        cases[name] = dict(net_wh=list(net_wh), orig_wh=list(orig_wh), class_table=table, first_frame=first, timestamps=stamps,
                           frames=frames_in, entities=ents_out,
                           given=dict(frame=det_frame, bbox=g.bbox.tolist(), mode=g.mode, size=list(g.size),
                                      labels=g.get_field("labels").tolist(), scores=g.get_field("scores").tolist(),
                                      ids=g.get_field("ids").tolist(), bbox_net=g_net.bbox.tolist()),
                           postprocess=dict(track_len=3, track_conf=0.3,
                                            kept=[[int(e.id), int(e.frame_num)] for e in kept.entities]))
        print(name, "frames", frames, "entities", sum(len(f) for f in ents_out), "kept after post-processing", len(kept.entities))
    path = os.path.join(ROOT, "tests", "golden", "results_wire.json")
    with open(path, "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
