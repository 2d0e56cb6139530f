"""Result wire format: track BoxLists -> per-box entities / MOT-challenge rows.

Mirrors siammot/utils/boxlists_to_entities.py:6-37 (``boxlists_to_entities``) and the conversion step of
siammot/engine/inferencer.py:65-70 (resize to the original frame, xywh).  The reference builds every entity
with three ``.item()`` / ``.tolist()`` calls per box — three device synchronisations per box per frame once
the BoxList lives on the GPU; here each frame's boxes, scores, labels and ids cross to the host in ONE copy.
``AnnoEntity`` is gluoncv's class in the reference (an un-vendored dependency); the fields the reference
fills are restated on a plain object.
"""
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
