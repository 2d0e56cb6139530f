"""Seeded inputs shared by ``oracle/gen_golden.py`` (which runs the reference on them) and the
parity tests (which run the oracle and the HIP path on them).

Everything comes from ``numpy.random.RandomState`` (legacy MT19937 stream, bit-stable across
numpy versions and machines), so ``tests/golden/*.npz`` only has to store OUTPUTS.
"""
import numpy as np

F32 = np.float32

# ---- full frame-pair cases --------------------------------------------------------------
EMM_CASES = {
    # DLA-34-FPN defaults (siammot/configs/defaults.py:35-82) at a reduced channel count / image
    "default": dict(channels=64, rz=15, search_region=2.0, pad_pixels=512, min_search_wh=0,
                    use_centerness=True, sigma=0.4, amodal=False,
                    scales=(0.25, 0.125, 0.0625, 0.03125), image_wh=(512, 384), seed=1234,
                    boxes=[(40.3, 50.7, 71.9, 114.2),        # 32x64   -> level 0
                           (150.5, 60.25, 214.0, 187.5),     # 64x128  -> level 0
                           (250.0, 100.0, 349.5, 299.0),     # 100x200 -> level 1
                           (300.2, 30.1, 459.7, 349.9),      # 160x320 -> level 2
                           (5.0, 200.0, 60.0, 380.0),        # hugging the left/bottom border
                           (-20.0, -30.0, 480.0, 400.0),     # larger than the image -> level 3
                           (-160.0, 40.0, -100.0, 140.0)]),  # left of the image: SR mostly virtual pad,
                                                             # result clipped away (remove_empty)
    # configs/dla/DLA_34_FPN_EMM_AOT.yaml:52-63 shape family: Rz=7, Rx=35, pad 256, no centerness
    "aot": dict(channels=32, rz=7, search_region=5.0, pad_pixels=256, min_search_wh=64,
                use_centerness=False, sigma=0.1, amodal=True,
                scales=(0.25, 0.125, 0.0625, 0.03125), image_wh=(512, 384), seed=4321,
                boxes=[(100.0, 100.0, 111.5, 123.0),         # tiny: min_search_wh kicks in
                       (200.5, 50.5, 260.0, 170.0),
                       (300.0, 150.0, 460.0, 370.0),
                       (20.0, 20.0, 180.0, 340.0)]),
}


def CHANNEL_SUBSET(C):
    return [0, C // 3, C - 1]


def feature_shapes(image_wh, channels):
    W, H = image_wh
    return [(1, channels, H // s, W // s) for s in (4, 8, 16, 32, 64)]


def predictor_params(rs, C, boxes):
    """Reference state_dict keys with structured values (§8d)."""
    def conv(o, i):
        return (rs.standard_normal((o, i, 3, 3)) * 0.05).astype(F32)
    mw = float(np.mean(boxes[:, 2] - boxes[:, 0]))
    mh = float(np.mean(boxes[:, 3] - boxes[:, 1]))
    p = {}
    for t in ("cls_tower", "reg_tower"):
        p[t + ".0.weight"] = conv(C, C)
        p[t + ".1.weight"] = rs.uniform(0.5, 1.5, C).astype(F32)
        p[t + ".1.bias"] = (rs.standard_normal(C) * 0.1).astype(F32)
    p["cls.weight"] = conv(2, C)
    p["cls.bias"] = np.array([0.1, -0.1], dtype=F32)
    p["center.weight"] = conv(1, C)
    p["center.bias"] = np.array([0.05], dtype=F32)
    p["reg.weight"] = conv(4, C)
    p["reg.bias"] = np.array([0.5 * mw, 0.5 * mh, 0.5 * mw, 0.5 * mh], dtype=F32)
    return p


def emm_case_inputs(name):
    case = EMM_CASES[name]
    rs = np.random.RandomState(case["seed"])
    shapes = feature_shapes(case["image_wh"], case["channels"])
    feats_a = [rs.standard_normal(s).astype(F32) for s in shapes]
    feats_b = [rs.standard_normal(s).astype(F32) for s in shapes]
    boxes = np.array(case["boxes"], dtype=F32)
    params = predictor_params(rs, case["channels"], boxes)
    return dict(features_a=feats_a, features_b=feats_b, boxes=boxes, params=params)


# ---- operator-level cases ---------------------------------------------------------------
XCORR_CASES = {
    "c128": dict(n=2, c=128, rx=30, rz=15, seed=11),
    "aot": dict(n=1, c=32, rx=35, rz=7, seed=12),
}


def xcorr_case_inputs(name):
    c = XCORR_CASES[name]
    rs = np.random.RandomState(c["seed"])
    x = rs.standard_normal((c["n"], c["c"], c["rx"], c["rx"])).astype(F32)
    z = rs.standard_normal((c["n"], c["c"], c["rz"], c["rz"])).astype(F32)
    return x, z


DECODE_CASES = {
    "default": dict(n=8, rx=30, rz=15, pad_pixels=512, use_centerness=True, sigma=0.4,
                    search_expansion=1.0, seed=21),
    "aot": dict(n=3, rx=35, rz=7, pad_pixels=256, use_centerness=False, sigma=0.1,
                search_expansion=4.0, seed=22),
}


def np_search_region(boxes, pad_pixels, e, min_wh=0.0):
    b = (boxes + F32(pad_pixels)).astype(F32)
    w = b[:, 2] - b[:, 0] + F32(1)
    h = b[:, 3] - b[:, 1] + F32(1)
    w_ext = np.maximum((F32(min_wh) - w) / F32(e * 2.0), w * F32(e / 2.0))
    h_ext = np.maximum((F32(min_wh) - h) / F32(e * 2.0), h * F32(e / 2.0))
    return np.stack((b[:, 0] - w_ext, b[:, 1] - h_ext, b[:, 2] + w_ext, b[:, 3] + h_ext), 1).astype(F32)


def decode_case_inputs(name):
    """Logits fed directly: cls, center ~ N(0,2); reg ~ |N(0,1)|·0.5·box side."""
    c = DECODE_CASES[name]
    rs = np.random.RandomState(c["seed"])
    n = c["n"]
    ho = c["rx"] - c["rz"] + 1
    wh = rs.uniform(24.0, 260.0, (n, 2))
    xy = rs.uniform(0.0, 900.0, (n, 2))
    boxes = np.concatenate((xy, xy + wh), 1).astype(F32)
    cls = (rs.standard_normal((n, 2, ho, ho)) * 2.0).astype(F32)
    center = (rs.standard_normal((n, 1, ho, ho)) * 2.0).astype(F32)
    side = np.stack((wh[:, 0], wh[:, 1], wh[:, 0], wh[:, 1]), 1)[:, :, None, None]
    reg = (np.abs(rs.standard_normal((n, 4, ho, ho))) * 0.5 * side).astype(F32)
    sr = np_search_region(boxes, c["pad_pixels"], c["search_expansion"])
    return dict(cls=cls, center=center, reg=reg, boxes=boxes, sr=sr)


# ---- box head / _refine_tracks (second half) ---------------------
REFINE_CASE = dict(image_wh=(512, 384), channels=32, resolution=7, scales=(0.25, 0.125, 0.0625, 0.03125),
                   sampling_ratio=2, mlp_dim=64, num_classes=3, score_thresh=0.05, nms=0.5,
                   reg_weights=(10.0, 10.0, 5.0, 5.0), seed=31)


# the box head of the reference's yamls (configs/dla/DLA_34_FPN_EMM.yaml:25-33): 7x7 pooler on the 128-channel DLA maps,
# 1024-1024 MLP, person / background — the sizes the weight-streaming GEMM kernels are built for (the
# small case above pins the head's LOGIC, this one its arithmetic at the shipped width)
REFINE_CASE_YAML = dict(image_wh=(1280, 704), channels=128, resolution=7, scales=(0.25, 0.125, 0.0625, 0.03125),
                        sampling_ratio=2, mlp_dim=1024, num_classes=2, score_thresh=0.05, nms=0.5,
                        reg_weights=(10.0, 10.0, 5.0, 5.0), seed=41)
REFINE_CASES = {"small": REFINE_CASE, "yaml": REFINE_CASE_YAML}


def refine_case_inputs(c=None):
    """FPN maps, a box head's weights (upstream parameter names) and two proposal sets: seven propagated tracks
    (ids >= 0, labels in {1, 2}, matching scores) and sixteen proposals of which seven carry an id."""
    c = REFINE_CASE if c is None else c
    rs = np.random.RandomState(c["seed"])
    feats = [rs.standard_normal(s).astype(F32) for s in feature_shapes(c["image_wh"], c["channels"])[:4]]
    d_in = c["channels"] * c["resolution"] ** 2
    params = {
        "feature_extractor.fc6.weight": (rs.standard_normal((c["mlp_dim"], d_in)) / np.sqrt(d_in)).astype(F32),
        "feature_extractor.fc6.bias": (0.1 * rs.standard_normal(c["mlp_dim"])).astype(F32),
        "feature_extractor.fc7.weight": (rs.standard_normal((c["mlp_dim"], c["mlp_dim"])) / 8.0).astype(F32),
        "feature_extractor.fc7.bias": (0.1 * rs.standard_normal(c["mlp_dim"])).astype(F32),
        "predictor.cls_score.weight": (rs.standard_normal((c["num_classes"], c["mlp_dim"])) / 2.0).astype(F32),
        "predictor.cls_score.bias": np.zeros(c["num_classes"], F32),
        "predictor.bbox_pred.weight": (rs.standard_normal((4 * c["num_classes"], c["mlp_dim"])) / 4.0).astype(F32),
        "predictor.bbox_pred.bias": np.zeros(4 * c["num_classes"], F32),
    }
    W, H = c["image_wh"]
    wh = np.exp(rs.uniform(np.log(12), np.log(260), (16, 2)))
    xy = rs.uniform(0, 1, (16, 2)) * np.array([W - 40.0, H - 40.0]) - 10.0
    boxes = np.concatenate((xy, xy + wh), 1).astype(F32)
    boxes[3] = [W - 30.0, H - 25.0, W + 40.0, H + 30.0]           # sticks out of the image: clipped by the head
    track_boxes = boxes[:7]
    track_ids = np.array([4, 0, 9, 2, 11, 5, 7], np.int64)
    track_labels = np.minimum(np.array([1, 2, 1, 1, 2, 1, 2], np.int64), c["num_classes"] - 1)
    track_scores = rs.uniform(0.3, 1.0, 7).astype(F32)
    mixed_boxes = boxes[rs.permutation(16)]
    order = rs.permutation(16)
    mixed_ids = np.full(16, -1, np.int64)
    for k, row in enumerate(order[:7]):
        mixed_ids[row] = track_ids[k]
    return dict(features=feats, params=params, track_boxes=track_boxes, track_ids=track_ids,
                track_labels=track_labels, track_scores=track_scores, mixed_boxes=mixed_boxes, mixed_ids=mixed_ids)


# ---- closed-loop tracking sequences -----------------------------------
# The reference's CombinedROIHeads.forward (roi_heads.py:22-52) with its own TrackHead / TrackSolver / TrackPool AND
# its own EMM in the loop, over synthetic 720p-shaped maps (oracle/gen_golden_sequence.py).  Frames are a smooth
# rotation through three independent noise fields, so a template extracted at frame t still correlates with frame
# t+1 (as on video) while every frame differs.
SEQ_CASES = {
    # propagated boxes keep their matching score + 1 (a box head that returns its proposals unchanged)
    "plain": dict(channels=128, image_wh=(1280, 704), frames=24, seed=101, thresholds=(0.4, 0.6, 0.4),
                  max_dormant_frames=3, n_objects=12, refine=False, cls_bias=(1.0, -1.0), reg_gain=1.0),
    # propagated boxes go through a box head as proposals (_refine_tracks, roi_heads.py:60-84)
    "refine": dict(channels=128, image_wh=(1280, 704), frames=20, seed=78, thresholds=(0.4, 0.6, 0.4),
                   max_dormant_frames=2, n_objects=10, refine=True, cls_bias=(1.0, -1.0), reg_gain=1.0,
                   box_head=dict(resolution=7, sampling_ratio=2, mlp_dim=64, num_classes=2, score_thresh=0.05,
                                 nms=0.5, reg_weights=(10.0, 10.0, 5.0, 5.0))),
    # the second yaml family in the loop (configs/dla/DLA_34_FPN_EMM_AOT.yaml:52-63: 7x7 templates, search region x5 ->
    # 35x35, 29x29 response, no centerness, cosine window 0.1, pad 256, TRACK_THRESH 0.6 / START_TRACK_THRESH 0.95): small
    # objects (that data set's airborne objects), a detector whose scores reach the start threshold
    "aot": dict(channels=128, image_wh=(1280, 704), frames=12, seed=330, thresholds=(0.6, 0.95, 0.4),
                max_dormant_frames=2, n_objects=7, refine=False, cls_bias=(-0.7, 0.7), reg_gain=1.0, family="aot",
                object_sizes=[(30, 30), (44, 36), (24, 28), (60, 48), (36, 40), (52, 60)], reg_size=(40.0, 40.0),
                det_scores=(0.85, 0.995)),
    # INPUT.AMODAL: True and INFERENCE.USE_GIVEN_DETECTIONS (configs/dla/DLA_34_FPN_EMM_MOT17.yaml:19-20,51-52): neither
    # the head nor the box head clips to the image, the detections enter through CombinedROIHeads.forward's
    # `given_detection` argument (roi_heads.py:23-32) and may stick out of the frame, objects leave through the border
    "amodal": dict(channels=128, image_wh=(1280, 704), frames=18, seed=213, thresholds=(0.4, 0.6, 0.4),
                   max_dormant_frames=3, n_objects=10, refine=True, cls_bias=(1.0, -1.0), reg_gain=1.0, amodal=True,
                   given_detections=True, edge_fraction=0.3,
                   box_head=dict(resolution=7, sampling_ratio=2, mlp_dim=64, num_classes=2, score_thresh=0.05,
                                 nms=0.5, reg_weights=(10.0, 10.0, 5.0, 5.0))),
    # ---- round 5 (VERDICT r4 "next" #1): the regimes where the capacity fallbacks live, pinned to the reference ----------
    # crowd: > 128 propagated rows in every frame from the twelfth on (30+ frames), box head in the loop — beyond the weight-streaming
    # refinement kernels' row count (library GEMMs) and, with the detections, near the one-launch solver's box count
    "crowd": dict(channels=128, image_wh=(1280, 704), frames=44, seed=501, thresholds=(0.4, 0.6, 0.4),
                  max_dormant_frames=3, n_objects=64, refine=True, cls_bias=(1.0, -1.0), reg_gain=1.0, layout="grid",
                  velocity=0.8, object_sizes=[(60, 120), (68, 132), (56, 112), (40, 80), (64, 128), (90, 150)],
                  reg_size=(62.0, 124.0), box_reg_gentle=16.0,
                  box_head=dict(resolution=7, sampling_ratio=2, mlp_dim=256, num_classes=2, score_thresh=0.05,
                                nms=0.5, reg_weights=(10.0, 10.0, 5.0, 5.0))),
    # crowd512: > 256 rois in the head (no order hint beyond 256) and > 512 boxes into the solver (host solver beyond the
    # one-launch kernel's capacity); no box head
    "crowd512": dict(channels=128, image_wh=(1280, 704), frames=9, seed=502, thresholds=(0.4, 0.6, 0.4),
                     max_dormant_frames=3, n_objects=200, refine=False, cls_bias=(1.0, -1.0), reg_gain=1.0, layout="grid",
                     velocity=0.5, object_sizes=[(36, 64), (40, 70), (32, 60), (44, 72)], reg_size=(38.0, 66.0),
                     false_positives=(90, 130)),
    # longdormant: the yaml's MAX_DORMANT_FRAMES 30 (configs/dla/DLA_34_FPN_EMM.yaml:20-22): dormant tracks are searched
    # from their stale entries for 30 frames, resume or expire (track_utils.py:152-178); objects disappear for 6-28 frames
    # (resume) and for good (expire)
    "longdormant": dict(channels=128, image_wh=(1280, 704), frames=72, seed=503, thresholds=(0.4, 0.6, 0.4),
                        max_dormant_frames=30, n_objects=6, refine=False, cls_bias=(1.6, -1.6), reg_gain=1.0,
                        long_gaps=True),
    # dormant256: the longdormant regime with three times the objects and MAX_DORMANT_FRAMES 100 (a legal value of the key; the
    # yamls use 30): the dormant rows keep accumulating, > 256 of them towards the end — more than one carry
    # launch takes (siammot_amd.ops.MEMORY_CARRY_MAX_ROWS): the loop falls back to the reference's host concatenation of the
    # dormant rows (TrackHead._update_memory_with_dormant_track, track_head.py:77-97) for those frames
    "dormant256": dict(channels=128, image_wh=(1280, 704), frames=100, seed=505, thresholds=(0.4, 0.6, 0.4),
                       max_dormant_frames=100, n_objects=18, refine=False, cls_bias=(1.6, -1.6), reg_gain=1.0,
                       long_gaps=True),
    # multiclass: configs[4] maps (R-50-FPN: 256 channels, 1080p -> 1056 x 1920), two foreground classes (person + vehicle)
    # and the box head at the yamls' width in the loop: PostProcessor.filter_results regroups the refined tracks by class
    # (box_head/inference.py:164-191) while _refine_tracks keeps the matching scores in input order (roi_heads.py:67-76)
    "multiclass": dict(channels=256, image_wh=(1920, 1056), frames=14, seed=504, thresholds=(0.4, 0.6, 0.4),
                       max_dormant_frames=3, n_objects=12, refine=True, cls_bias=(1.0, -1.0), reg_gain=1.0, n_foreground=2,
                       box_head=dict(resolution=7, sampling_ratio=2, mlp_dim=1024, num_classes=3, score_thresh=0.05,
                                     nms=0.5, reg_weights=(10.0, 10.0, 5.0, 5.0))),
}
# Random-init regression heads predict a box of about the bias size whatever the object: most objects are near that
# size (straddling the FPN level 1 / 2 boundary at sqrt(area) = 224) so that tracks and detections keep meeting in
# the solver's NMS; the small and the large object exercise levels 0 and 3 whenever a track starts on them.
SEQ_OBJECT_SIZES = [(150, 300), (170, 340), (140, 290), (64, 128), (160, 330), (430, 540)]
SEQ_REG_SIZE = (160.0, 320.0)


class SequenceInputs(object):
    """Seeded inputs of one closed-loop case: ``features(t)`` (5 FPN levels, fp32 numpy), ``detections(t)``
    (boxes, scores), ``params`` (EMM predictor state_dict) and, for the refine case, ``box_head_params``."""

    def __init__(self, name):
        self.case = c = SEQ_CASES[name]
        self.name = name
        rs = np.random.RandomState(c["seed"])
        shapes = feature_shapes(c["image_wh"], c["channels"])
        self._fields = [[rs.standard_normal(s).astype(F32) for s in shapes] for _ in range(3)]
        self._objects(rs)
        reg_size = c.get("reg_size", SEQ_REG_SIZE)
        boxes = np.array([[0, 0, reg_size[0], reg_size[1]]], dtype=F32)
        self.params = predictor_params(rs, c["channels"], boxes)
        self.params["cls.bias"] = np.array(c["cls_bias"], dtype=F32)
        for k in ("reg.weight",):
            self.params[k] = (self.params[k] * F32(c["reg_gain"])).astype(F32)
        if c["refine"]:
            b = c["box_head"]
            d_in = c["channels"] * b["resolution"] ** 2
            m, k = b["mlp_dim"], b["num_classes"]
            self.box_head_params = {
                "feature_extractor.fc6.weight": (rs.standard_normal((m, d_in)) / np.sqrt(d_in)).astype(F32),
                "feature_extractor.fc6.bias": (0.1 * rs.standard_normal(m)).astype(F32),
                # (scales normalised by the layer width — 8.0 = sqrt(64) at the width of the round-3/4 cases, whose values stay
                # bit for bit what they were — so that a 256- or 1024-wide head has the same gain)
                "feature_extractor.fc7.weight": (rs.standard_normal((m, m)) / np.sqrt(m)).astype(F32),
                "feature_extractor.fc7.bias": (0.1 * rs.standard_normal(m)).astype(F32),
                # a box head that nudges an already tracked box by about a percent of its size and scores it around
                # 0.5, as a trained head does; large random regressions off noise features make the closed loop
                # chaotic (two fp32 CPU implementations of the same head then part ways within ten frames).  The crowd
                # case (44 frames, 250 rows) divides the regression by another `box_reg_gentle`: the loop's gain on a
                # rounding difference must stay below one over its length
                "predictor.cls_score.weight": (rs.standard_normal((k, m)) / np.sqrt(m)).astype(F32),
                "predictor.cls_score.bias": np.zeros(k, F32),
                "predictor.bbox_pred.weight": (rs.standard_normal((4 * k, m)) / (8.0 * np.sqrt(m) * c.get("box_reg_gentle", 1.0))).astype(F32),
                "predictor.bbox_pred.bias": np.zeros(4 * k, F32),
            }

    def _objects(self, rs):
        c = self.case
        W, H = c["image_wh"]
        n, T = c["n_objects"], c["frames"]
        sizes = c.get("object_sizes", SEQ_OBJECT_SIZES)
        self.obj_wh = np.array([sizes[i % len(sizes)] for i in range(n)], dtype=np.float64)
        lo = self.obj_wh / 2 + 4
        hi = np.array([W, H]) - self.obj_wh / 2 - 4
        if c.get("edge_fraction"):               # amodal: centres may start that fraction of a box beyond the border
            lo = lo - (0.5 + c["edge_fraction"]) * self.obj_wh
            hi = hi + (0.5 + c["edge_fraction"]) * self.obj_wh
        self.obj_c0 = lo + rs.uniform(0, 1, (n, 2)) * (hi - lo)
        vmax = c.get("velocity", 2.5)
        self.obj_vel = rs.uniform(-vmax, vmax, (n, 2))
        if c.get("layout") == "grid":            # crowd: one object per cell of a grid over the image, jittered inside it
            cols = int(np.ceil(np.sqrt(n * W / float(H))))
            rows = int(np.ceil(n / float(cols)))
            cw, ch = W / float(cols), H / float(rows)
            cell = np.stack(((np.arange(n) % cols + 0.5) * cw, (np.arange(n) // cols + 0.5) * ch), 1)
            self.obj_c0 = np.clip(cell + (self.obj_c0 - lo) / np.maximum(hi - lo, 1) * 0.2 * np.array([cw, ch]), lo, hi)
        self.obj_first = np.where(np.arange(n) % 4 == 3, rs.randint(2, T // 2, n), 0)       # late arrivals
        self.obj_last = np.where(np.arange(n) % 5 == 2, rs.randint(T // 2, T - 3, n), T)    # leave for good
        # detector drop-outs: bursts of 1-3 frames
        self.obj_seen = np.ones((n, T), dtype=bool)
        for i in range(n):
            for _ in range(int(rs.randint(0, 3))):
                t0, ln = int(rs.randint(1, T - 3)), int(rs.randint(1, 4))
                self.obj_seen[i, t0:t0 + ln] = False
            self.obj_seen[i, :self.obj_first[i]] = False
            self.obj_seen[i, self.obj_last[i]:] = False
        self._det_seed = int(rs.randint(0, 2 ** 31 - 1))
        if c.get("long_gaps"):                   # long dormancy: every object also disappears once for 6-28 frames
            g = np.random.RandomState(self._det_seed ^ 0x5a5a)
            for i in range(n):
                t0, ln = int(g.randint(4, T // 2)), int(g.randint(6, 29))
                self.obj_seen[i, t0:t0 + ln] = False
        # foreground class of every object (1 .. n_foreground): no random draw, the existing cases' streams stay as they are
        self.obj_label = 1 + np.arange(n) % int(c.get("n_foreground", 1))

    def features(self, t):
        th, ph = 0.12 * t, 0.7 * t
        c0, c1, c2 = F32(np.cos(th)), F32(np.sin(th) * np.cos(ph)), F32(np.sin(th) * np.sin(ph))
        return [((c0 * f0) + (c1 * f1)) + (c2 * f2) for f0, f1, f2 in zip(*self._fields)]

    def detections(self, t, labels=False):
        """(boxes, scores) of frame t; with ``labels`` also the class of every row (1 for the one-class cases)."""
        c = self.case
        W, H = c["image_wh"]
        rs = np.random.RandomState(self._det_seed + 7919 * t)
        ctr = self.obj_c0 + self.obj_vel * t
        boxes = np.concatenate((ctr - self.obj_wh / 2, ctr + self.obj_wh / 2), 1) + rs.uniform(-1.5, 1.5, (len(ctr), 4))
        boxes = boxes[self.obj_seen[:, t]]
        lo_fp, hi_fp = c.get("false_positives", (0, 3))
        nfp = int(rs.randint(lo_fp, hi_fp))
        fc = rs.uniform(60, [W - 60.0, H - 60.0], (nfp, 2))
        fwh = rs.uniform(30, 110, (nfp, 2))
        boxes = np.concatenate((boxes, np.concatenate((fc - fwh / 2, fc + fwh / 2), 1)), 0)
        if not c.get("amodal"):
            boxes[:, 0::2] = np.clip(boxes[:, 0::2], 0, W - 1)
            boxes[:, 1::2] = np.clip(boxes[:, 1::2], 0, H - 1)
        lo_s, hi_s = c.get("det_scores", (0.45, 0.99))
        scores = rs.uniform(lo_s, hi_s, len(boxes))
        if c.get("false_positives"):             # the flood of false positives stays under the start threshold
            scores[len(boxes) - nfp:] = rs.uniform(0.06, 0.55, nfp)
        if labels:
            lab = np.concatenate((self.obj_label[self.obj_seen[:, t]], 1 + (t + np.arange(nfp)) % int(c.get("n_foreground", 1))))
            return boxes.astype(F32), scores.astype(F32), lab.astype(np.int64)
        return boxes.astype(F32), scores.astype(F32)


# ---- the benchmark configurations (BASELINE.json configs) whose reference outputs oracle/gen_golden_bench.py stores -----
BENCH_FAMILIES = {
    # siammot/configs/defaults.py:35-82 / configs/dla/DLA_34_FPN_EMM.yaml
    "default": dict(rz=15, search_region=2.0, pad_pixels=512, min_search_wh=0, use_centerness=True, sigma=0.4,
                    amodal=False, scales=(0.25, 0.125, 0.0625, 0.03125)),
    # configs/dla/DLA_34_FPN_EMM_AOT.yaml:52-63
    "aot": dict(rz=7, search_region=5.0, pad_pixels=256, min_search_wh=0, use_centerness=False, sigma=0.1,
                amodal=False, scales=(0.25, 0.125, 0.0625, 0.03125)),
}
BENCH_CONFIGS = {
    "n30": dict(channels=128, net_hw=(704, 1280), n=30, family="default"),       # configs[1]
    "n100": dict(channels=128, net_hw=(704, 1280), n=100, family="default"),     # configs[2]
    "cfg0": dict(channels=128, net_hw=(800, 800), n=4, family="default"),        # configs[0]: 256x256 frame -> 800x800
    "cfg4": dict(channels=256, net_hw=(1056, 1920), n=50, family="default"),     # configs[4]: R-50-FPN, 1080p
    "aot_n30": dict(channels=128, net_hw=(704, 1280), n=30, family="aot"),       # second yaml family at configs[1] size
}


def bench_channel_subset(C):
    return [0, C // 2 - 1, C - 1]
