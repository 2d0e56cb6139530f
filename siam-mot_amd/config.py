"""yacs-free config carrying exactly the keys the EMM path reads.

Reference: siammot/configs/defaults.py:35-82 (values) and the reads at
EMM/track_core.py:23-26, EMM/feature_extractor.py:17-20,47-52, track_utils.py:258-269.
A real yacs ``cfg`` from the reference works in its place (attribute access only).
"""


class CfgNode(dict):
    """Attribute-style nested dict (just enough of yacs.CfgNode for attribute reads/writes)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return CfgNode({k: (v.clone() if isinstance(v, CfgNode) else v) for k, v in self.items()})


def get_default_cfg(conv_body="DLA-34-FPN", channels=128):
    cfg = CfgNode()
    # INPUT.* / DATALOADER.*: the DLA_34_FPN_EMM.yaml values (reference configs/dla/DLA_34_FPN_EMM.yaml:4-8,38)
    cfg.INPUT = CfgNode(AMODAL=False, MIN_SIZE_TEST=800, MAX_SIZE_TEST=1280, PIXEL_MEAN=[0.485, 0.456, 0.406],
                        PIXEL_STD=[0.229, 0.224, 0.225], TO_BGR255=False)
    cfg.DATALOADER = CfgNode(SIZE_DIVISIBILITY=32)
    cfg.MODEL = CfgNode()
    cfg.MODEL.BACKBONE = CfgNode(CONV_BODY=conv_body)
    cfg.MODEL.DLA = CfgNode(BACKBONE_OUT_CHANNELS=channels)
    cfg.MODEL.RESNETS = CfgNode(BACKBONE_OUT_CHANNELS=256)
    cfg.MODEL.GROUP_NORM = CfgNode(DIM_PER_GP=-1, NUM_GROUPS=32, EPSILON=1e-5)
    th = CfgNode()
    th.TRACKTOR = False
    th.POOLER_SCALES = (0.25, 0.125, 0.0625, 0.03125)
    th.POOLER_RESOLUTION = 15
    th.POOLER_SAMPLING_RATIO = 2
    th.PAD_PIXELS = 512
    th.SEARCH_REGION = 2.0
    th.MINIMUM_SREACH_REGION = 0
    th.MODEL = "EMM"
    th.TRACK_THRESH = 0.4
    th.START_TRACK_THRESH = 0.6
    th.RESUME_TRACK_THRESH = 0.4
    th.MAX_DORMANT_FRAMES = 1
    th.EMM = CfgNode(USE_CENTERNESS=True, POS_RATIO=0.25, HN_RATIO=0.25, TRACK_LOSS_WEIGHT=1.0,
                     CLS_POS_REGION=0.8, COSINE_WINDOW_WEIGHT=0.4)
    cfg.MODEL.TRACK_HEAD = th
    # the box head the tracker calls back into (_refine_tracks, roi_heads.py:60-84): DLA_34_FPN_EMM.yaml:25-33;
    # thresholds / regression weights are [UPSTREAM] maskrcnn_benchmark defaults (the reference's yamls keep them)
    cfg.MODEL.CLS_AGNOSTIC_BBOX_REG = False
    cfg.MODEL.ROI_HEADS = CfgNode(USE_FPN=True, SCORE_THRESH=0.05, NMS=0.5, DETECTIONS_PER_IMG=100,
                                  BBOX_REG_WEIGHTS=(10.0, 10.0, 5.0, 5.0))
    cfg.MODEL.ROI_BOX_HEAD = CfgNode(POOLER_RESOLUTION=7, POOLER_SCALES=(0.25, 0.125, 0.0625, 0.03125),
                                     POOLER_SAMPLING_RATIO=2, MLP_HEAD_DIM=1024, NUM_CLASSES=2)
    return cfg
