"""SURVEY.md §8(f) rank 1, second half: ``_refine_tracks`` and the box head it calls, against
``tests/golden/refine_tracks.npz`` — outputs of the reference's OWN ``CombinedROIHeads._refine_tracks`` /
``ROIBoxHead`` / ``PostProcessor`` (oracle/gen_golden_refine.py)."""
import os
import types

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import box_head_oracle as BO
from oracle import solver_oracle as SO
from siammot_amd.box_refine import BoxCoder, RefineTracks, TrackBoxHead, build_refine_tracks
from siammot_amd.structures import BoxList

GOLD = os.path.join(os.path.dirname(__file__), "golden", "refine_tracks.npz")


def _cfg(c=None):
    c = gi.REFINE_CASE if c is None else c
    ns = types.SimpleNamespace
    return ns(INPUT=ns(AMODAL=False),
              MODEL=ns(CLS_AGNOSTIC_BBOX_REG=False,
                       ROI_HEADS=ns(BBOX_REG_WEIGHTS=c["reg_weights"], SCORE_THRESH=c["score_thresh"], NMS=c["nms"]),
                       ROI_BOX_HEAD=ns(POOLER_RESOLUTION=c["resolution"], POOLER_SCALES=c["scales"],
                                       POOLER_SAMPLING_RATIO=c["sampling_ratio"], MLP_HEAD_DIM=c["mlp_dim"],
                                       NUM_CLASSES=c["num_classes"]),
                       TRACK_HEAD=ns(TRACKTOR=False)))


def _cpu_nms(boxlist, thresh):
    keep = SO.nms_indices(boxlist.bbox.numpy(), boxlist.get_field("scores").numpy(), thresh)
    return boxlist[torch.from_numpy(keep)]


def _proposals(boxes, ids, dev, labels=None, scores=None, image_wh=None):
    bl = BoxList(torch.from_numpy(boxes.copy()).to(dev), image_wh or gi.REFINE_CASE["image_wh"], mode="xyxy")
    bl.add_field("ids", torch.from_numpy(ids.copy()).to(dev))
    if labels is not None:
        bl.add_field("labels", torch.from_numpy(labels.copy()).to(dev))
    if scores is not None:
        bl.add_field("scores", torch.from_numpy(scores.copy()).to(dev))
    return bl


def _check(head, dev, tol):
    c, inp, g = gi.REFINE_CASE, gi.refine_case_inputs(), np.load(GOLD)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in inp["params"].items()}, strict=True)
    head = head.to(dev).eval()
    feats = [torch.from_numpy(f).to(dev) for f in inp["features"]]
    with torch.no_grad():
        refine = RefineTracks(head)
        r = refine(feats, [_proposals(inp["track_boxes"], inp["track_ids"], dev, inp["track_labels"],
                                      inp["track_scores"])])[0]
        x, res, losses = head(feats, [_proposals(inp["mixed_boxes"], inp["mixed_ids"], dev)])
    assert losses == {}
    assert r.get_field("ids").cpu().numpy().tolist() == g["refine_ids"].tolist()
    assert r.get_field("labels").cpu().numpy().tolist() == g["refine_labels"].tolist()      # regrouped by label
    np.testing.assert_allclose(r.bbox.cpu().numpy(), g["refine_bbox"], rtol=0, atol=tol * 100)
    np.testing.assert_allclose(r.get_field("scores").cpu().numpy(), g["refine_scores"], rtol=0, atol=tol)
    sc = r.get_field("scores").cpu().numpy()
    assert (sc > 1.0).all() and (sc <= 2.0).all()                  # the band TrackSolver reads as "track"
    m = res[0]
    assert m.get_field("ids").cpu().numpy().tolist() == g["mixed_ids"].tolist()
    assert m.get_field("labels").cpu().numpy().tolist() == g["mixed_labels"].tolist()
    np.testing.assert_allclose(m.bbox.cpu().numpy(), g["mixed_bbox"], rtol=0, atol=tol * 100)
    np.testing.assert_allclose(m.get_field("scores").cpu().numpy(), g["mixed_scores"], rtol=0, atol=tol)
    np.testing.assert_allclose(x[:, ::9].cpu().numpy(), g["mixed_x_sub"], rtol=0, atol=tol * 10)
    assert int((g["mixed_ids"] >= 0).sum()) > 7 and len(g["mixed_ids"]) < 32      # id rows kept per class, dets pruned


def test_box_head_host_logic_matches_reference_golden():
    """Everything above the pooler and the NMS kernel (MLP, soft-max, BoxCoder, the track branch of the post-processor,
    per-class filtering, ``_refine_tracks`` score averaging and regrouping) with the oracle's pooler / numpy NMS
    injected: no GPU, no HIP library."""
    c = gi.REFINE_CASE
    head = TrackBoxHead(_cfg(), c["channels"], pooler=BO.OraclePooler(c["resolution"], c["scales"], c["sampling_ratio"]),
                        nms_fn=_cpu_nms)
    _check(head, "cpu", 2e-6)


def test_box_head_parameter_names_are_upstreams():
    c = gi.REFINE_CASE
    head = TrackBoxHead(_cfg(), c["channels"], pooler=torch.nn.Identity())
    assert sorted(head.state_dict().keys()) == sorted(gi.refine_case_inputs()["params"].keys())
    with pytest.raises(NotImplementedError):
        head.train()(None, None)


def test_box_coder_matches_the_loopwise_restatement():
    rs = np.random.RandomState(5)
    boxes = torch.from_numpy(np.abs(rs.standard_normal((40, 4))).cumsum(1).astype(np.float32) * 30)
    codes = torch.from_numpy(rs.standard_normal((40, 12)).astype(np.float32) * 4)
    codes[0, 2] = 100.0                                              # clamp at log(1000/16)
    a = BoxCoder((10.0, 10.0, 5.0, 5.0)).decode(codes, boxes)
    b = BO.BoxCoder((10.0, 10.0, 5.0, 5.0)).decode(codes, boxes)
    assert torch.equal(a, b)


def test_refine_tracks_empty_and_tracktor():
    empty = BoxList(torch.zeros(0, 4), (10, 10))
    empty.add_field("scores", torch.zeros(0))
    assert RefineTracks(None)(None, [empty])[0] is empty

    def box(features, tracks):
        t = tracks[0]
        out = BoxList(t.bbox + 1.0, t.size)
        out.add_field("scores", torch.full((len(t),), 1.25))
        out.add_field("ids", t.get_field("ids"))
        out.add_field("labels", t.get_field("labels"))
        return None, [out], {}
    t = _proposals(np.zeros((2, 4), np.float32), np.array([3, 4]), "cpu", np.array([1, 1]),
                   np.array([0.5, 0.9], np.float32))
    assert RefineTracks(box, tracktor=True)(None, [t])[0].get_field("scores").tolist() == [1.25, 1.25]
    assert RefineTracks(box)(None, [t])[0].get_field("scores").tolist() == pytest.approx([1.375, 1.575])
    cfg = _cfg()
    assert isinstance(build_refine_tracks(cfg, 32, box_head=box), RefineTracks)


@pytest.mark.gpu
def test_box_head_on_the_hip_pooler_and_nms_matches_reference_golden():
    """The product configuration: HIP ROIAlign 7x7 + hipBLASLt linears + HIP NMS on the device."""
    c = gi.REFINE_CASE
    _check(TrackBoxHead(_cfg(), c["channels"]), "cuda", 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tracktor,one_call", [(False, True), (True, True), (False, False)])
def test_refine_raw_one_launch_post_processing_matches_the_reference_golden(tracktor, one_call, monkeypatch):
    """``RefineTracks.refine_raw`` (HIP pooler -> three GEMMs -> smot_box_refine_post_fwd: no BoxList, no host
    synchronisation) on the seven propagated tracks of the golden case — three classes, labels {1, 2}: the rows come
    back grouped by label and the matching scores are paired in input order, as the reference's
    ``_refine_tracks`` (roi_heads.py:60-84) over its own ``ROIBoxHead`` produced them; TRACKTOR = the box head's score
    alone, against the general path."""
    c, inp, g = gi.REFINE_CASE, gi.refine_case_inputs(), np.load(GOLD)
    dev = "cuda"
    head = TrackBoxHead(_cfg(), c["channels"])
    head.load_state_dict({k: torch.from_numpy(v) for k, v in inp["params"].items()}, strict=True)
    head = head.to(dev).eval()
    feats = [torch.from_numpy(f).to(dev) for f in inp["features"]]
    refine = RefineTracks(head, tracktor=tracktor)
    assert refine.raw_ok(7)
    if not one_call:          # the stage-wise form (more than 128 rows, other layer widths): library GEMMs + post kernel
        import siammot_amd.ops as ops_
        monkeypatch.setattr(ops_, "linear_rows_max_rows", lambda: 0)
    args = [torch.from_numpy(inp[k].copy()).to(dev) for k in ("track_boxes", "track_scores", "track_ids", "track_labels")]
    refine.refine_raw(feats, *args, c["image_wh"])           # first call: builds the concatenated head weights
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")                  # any host synchronisation inside raises
    try:
        bb, sc, ids, lab = refine.refine_raw(feats, *args, c["image_wh"])
    finally:
        torch.cuda.set_sync_debug_mode("default")
    with torch.no_grad():
        r = refine(feats, [_proposals(inp["track_boxes"], inp["track_ids"], dev, inp["track_labels"],
                                      inp["track_scores"])])[0]
    assert ids.cpu().tolist() == r.get_field("ids").cpu().tolist() == g["refine_ids"].tolist()
    assert lab.cpu().tolist() == r.get_field("labels").cpu().tolist() == g["refine_labels"].tolist()
    np.testing.assert_allclose(bb.cpu().numpy(), r.bbox.cpu().numpy(), rtol=0, atol=2e-4)
    np.testing.assert_allclose(sc.cpu().numpy(), r.get_field("scores").cpu().numpy(), rtol=0, atol=2e-6)
    if not tracktor:
        np.testing.assert_allclose(bb.cpu().numpy(), g["refine_bbox"], rtol=0, atol=2e-3)
        np.testing.assert_allclose(sc.cpu().numpy(), g["refine_scores"], rtol=0, atol=2e-5)
    assert float(sc.min()) > 1.0 and float(sc.max()) <= 2.0


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,relu", [(30, 6272, 1024, True), (30, 1024, 1024, True), (7, 1024, 10, False),
                                        (64, 6272, 64, False), (1, 64, 3, True), (33, 100, 130, False), (16, 36, 2, False),
                                        (100, 6272, 1024, True), (128, 1024, 1024, True), (65, 1024, 10, False), (97, 200, 70, True)])
def test_linear_rows_matches_the_library_gemm(M, K, N, relu):
    """``smot_linear_rows_fwd`` (split-K weight streaming on the fp32 matrix cores + slice-ordered reduction) against a
    float64 reference: the box head's layers at the sizes of DLA_34_FPN_EMM.yaml (fc6 6272 -> 1024, fc7, predictor) and
    edge shapes (one row, 64 rows, more than 64 rows — five to eight row tiles, the 100 tracks of BASELINE.json configs[2]
    and the 128-row capacity —, K not a multiple of the step, N below a tile, a column-block output)."""
    import siammot_amd.ops as ops
    g = torch.Generator().manual_seed(M * 1000 + N)
    x = torch.randn((M, K), generator=g).cuda()
    w = (torch.randn((N, K), generator=g) / K ** 0.5).cuda()
    b = torch.randn((N,), generator=g).cuda()
    ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.clamp(min=0)
    y = ops.linear_rows(x, w, b, relu=relu)
    assert y.shape == (M, N)
    scale = float((x.double().abs() @ w.double().abs().t()).max())
    assert float((y.double() - ref).abs().max()) <= 2e-6 * scale
    assert torch.equal(y, ops.linear_rows(x, w, b, relu=relu))                     # deterministic
    buf = torch.full((M, N + 5), -7.0, device="cuda")
    ops.linear_rows(x, w, None, relu=False, out=buf[:, 3:])                        # column block, no bias
    assert torch.equal(buf[:, :3], torch.full((M, 3), -7.0, device="cuda")) and float(buf[:, 3 + N:].max()) == -7.0
    assert float((buf[:, 3:3 + N].double() - (ref - b.double() if not relu else x.double() @ w.double().t())).abs().max()) <= 2e-6 * scale
    with pytest.raises(RuntimeError):
        ops.linear_rows(torch.randn((3, 6), device="cuda"), torch.randn((4, 6), device="cuda"))      # K % 4 != 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,dim6,dim7,classes", [(30, 1024, 1024, 2), (64, 1024, 1024, 2), (1, 1024, 1024, 2), (17, 128, 64, 3),
                                                 (30, 96, 100, 2), (9, 256, 1024, 14), (30, 64, 128, 16),
                                                 (100, 1024, 1024, 2), (128, 1024, 1024, 2), (71, 128, 64, 3)])
def test_one_call_refinement_equals_the_stage_wise_composition_bitwise(n, dim6, dim7, classes):
    """``smot_box_refine_fwd`` skips two reduction launches where the layer shapes allow (the head's launch adds fc7's K
    slices while it loads, the post-processing kernel adds the head's): the slices are added in the order and with the
    bias / ReLU placement of the reduction kernel, so the result must equal pooler -> linear_rows x 3 -> box_refine_post
    bit for bit — at the yaml's 1024-1024 head, at 64 rows and one row, at widths where the chain applies with fewer
    slices, where it does not (fc7 width not a multiple of 64), and at heads of 70 / 80 columns (more than one neuron
    block: both reductions are launches again)."""
    import siammot_amd.ops as ops
    g = torch.Generator().manual_seed(n * 7 + dim7)
    C, scales = 16, (0.25, 0.125, 0.0625, 0.03125)
    feats = tuple(torch.randn((1, C, 352 // s_, 640 // s_), generator=g).cuda() for s_ in (1, 2, 4, 8))
    xy = torch.rand((n, 2), generator=g) * torch.tensor([1000.0, 500.0])
    wh = 20.0 + torch.rand((n, 2), generator=g) * 200.0
    boxes = torch.cat((xy, xy + wh), 1).cuda()
    labels = torch.randint(1, classes, (n,), generator=g).cuda()
    ids = torch.arange(n).cuda()
    conf = torch.rand((n,), generator=g).cuda()

    def lin(o, i):
        return (torch.randn((o, i), generator=g) / i ** 0.5).cuda(), torch.randn((o,), generator=g).mul(0.1).cuda()
    w6, b6 = lin(dim6, C * 49)
    w7, b7 = lin(dim7, dim6)
    wc, bc = lin(classes, dim7)
    wr, br = lin(4 * classes, dim7)
    weights, clip = (10.0, 10.0, 5.0, 5.0), 4.135
    one = ops.box_refine(feats, scales, 7, 2, boxes, labels, ids, conf, (w6, b6, w7, b7, wc, bc, wr, br), weights, clip,
                         (1280, 704))
    x = ops.roi_align_levels(feats, boxes, boxes, 7, scales, 2).reshape(n, -1)
    h6 = ops.linear_rows(x, w6, b6, relu=True)
    h7 = ops.linear_rows(h6, w7, b7, relu=True)
    ho = torch.empty((n, 5 * classes), device="cuda")
    ops.linear_rows(h7, wc, bc, out=ho[:, :classes])
    ops.linear_rows(h7, wr, br, out=ho[:, classes:])
    staged = ops.box_refine_post(ho, classes, classes, boxes, labels, ids, conf, weights, clip, (1280, 704))
    for a, b in zip(one, staged):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_tracking_loop_with_refine_tracks_runs_the_reference_order():
    """TrackingLoop(refine_tracks=RefineTracks(box head)): the propagated boxes go through the box head as proposals
    (roi_heads.py:43-45), come back with scores in the (1, 2] band, and the tracks keep their ids over frames."""
    import bench
    from siammot_amd.config import get_default_cfg
    from siammot_amd.track_head import build_tracking_loop
    cfg = get_default_cfg(channels=bench.CHANNELS)
    cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM = 64
    image_wh = (bench.NET_HW[1], bench.NET_HW[0])
    torch.manual_seed(0)
    head = TrackBoxHead(cfg, bench.CHANNELS).to("cuda").eval()
    with torch.no_grad():
        head.predictor.bbox_pred.weight.mul_(0.02)                 # small regressions: tracks stay on their objects
        head.predictor.cls_score.bias.copy_(torch.tensor([-4.0, 4.0]))
    calls = []
    refine = RefineTracks(head)

    def counted(features, tracks):
        out = refine(features, tracks)
        calls.append((len(tracks[0]), float(out[0].get_field("scores").min()), float(out[0].get_field("scores").max())))
        return out
    loop = build_tracking_loop(cfg, "cuda", refine_tracks=counted)
    boxes = torch.tensor([[100.0 + 150 * i, 80.0 + 60 * (i % 3), 160.0 + 150 * i, 220.0 + 60 * (i % 3)] for i in range(7)])
    bench.init_predictor(loop.track.tracker.predictor, boxes)
    with torch.no_grad():
        for name in ("cls", "center", "reg"):                      # a response dominated by the cosine window
            getattr(loop.track.tracker.predictor, name).weight.mul_(0.02)
    loop.track.tracker.to("cuda")
    feats = [bench.synthetic_features(100 + k, "cuda") for k in range(2)]
    ids_seen = []
    for f in range(4):
        det = BoxList(boxes.clone().to("cuda"), image_wh)
        det.add_field("scores", torch.full((7,), 0.99, device="cuda"))
        det.add_field("labels", torch.ones(7, dtype=torch.int64, device="cuda"))
        det.add_field("ids", torch.full((7,), -1, dtype=torch.int64, device="cuda"))
        out = loop(feats[f % 2], det)
        ids_seen.append(sorted(i for i in out.get_field("ids").tolist() if i >= 0))
    assert ids_seen[0] == list(range(7))
    assert len(calls) == 3 and all(n == 7 for n, _, _ in calls)     # every later frame refined the 7 propagated boxes
    assert all(lo > 1.0 and hi <= 2.0 for _, lo, hi in calls)
    assert ids_seen[-1] == list(range(7))                           # the refined tracks win the NMS against detections


@pytest.mark.gpu
def test_yaml_sized_box_head_matches_the_reference_golden():
    """the refinement at the SHIPPED head width (configs/dla/DLA_34_FPN_EMM.yaml:25-33: 7x7 pooler on
    128-channel maps, 1024-1024 MLP, two classes) against ``tests/golden/refine_tracks_yaml.npz`` = the reference's own
    ``ROIBoxHead`` / ``PostProcessor`` / ``_refine_tracks`` on these tensors (oracle/gen_golden_refine.py yaml) — through
    the general path (HIP pooler, library GEMMs, the reference-shaped post-processor), the device-only ``refine_raw``
    (weight-streaming GEMM kernels behind one C-ABI call) and the box head's plain ``forward`` on mixed proposals."""
    c = gi.REFINE_CASE_YAML
    inp = gi.refine_case_inputs(c)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "refine_tracks_yaml.npz"))
    dev = "cuda:0"
    head = TrackBoxHead(_cfg(c), c["channels"])
    head.load_state_dict({k: torch.from_numpy(v) for k, v in inp["params"].items()}, strict=True)
    head = head.to(dev).eval()
    feats = [torch.from_numpy(f).to(dev) for f in inp["features"]]
    wh = c["image_wh"]
    tol = 3e-5            # fp32 dot products of 6,272 and 1,024 terms in another order than torch-CPU's
    with torch.no_grad():
        refine = RefineTracks(head)
        tr = _proposals(inp["track_boxes"], inp["track_ids"], dev, inp["track_labels"], inp["track_scores"], wh)
        r = refine(feats, [tr])[0]
        assert refine.raw_ok(len(tr)) and head.one_call_ok(len(tr))
        rb, rs_, ri, rl = refine.refine_raw(feats, tr.bbox, tr.get_field("scores"), tr.get_field("ids"), tr.get_field("labels"), wh)
        x, res, _ = head(feats, [_proposals(inp["mixed_boxes"], inp["mixed_ids"], dev, image_wh=wh)])
    for bb, sc, ids, labels in ((r.bbox, r.get_field("scores"), r.get_field("ids"), r.get_field("labels")), (rb, rs_, ri, rl)):
        assert ids.cpu().numpy().tolist() == g["refine_ids"].tolist() and labels.cpu().numpy().tolist() == g["refine_labels"].tolist()
        np.testing.assert_allclose(bb.cpu().numpy(), g["refine_bbox"], rtol=0, atol=tol * 100)
        np.testing.assert_allclose(sc.cpu().numpy(), g["refine_scores"], rtol=0, atol=tol)
    m = res[0]
    assert m.get_field("ids").cpu().numpy().tolist() == g["mixed_ids"].tolist()
    assert m.get_field("labels").cpu().numpy().tolist() == g["mixed_labels"].tolist()
    np.testing.assert_allclose(m.bbox.cpu().numpy(), g["mixed_bbox"], rtol=0, atol=tol * 100)
    np.testing.assert_allclose(m.get_field("scores").cpu().numpy(), g["mixed_scores"], rtol=0, atol=tol)
    np.testing.assert_allclose(x[:, ::9].cpu().numpy(), g["mixed_x_sub"], rtol=0, atol=tol * 10)


def test_yaml_sized_box_head_host_logic_matches_reference_golden():
    """The same fixture on CPU with the oracle's pooler and the numpy NMS (no HIP library): pins the host logic at the
    shipped width in the ``-m "not gpu"`` suite."""
    c = gi.REFINE_CASE_YAML
    inp = gi.refine_case_inputs(c)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "refine_tracks_yaml.npz"))
    head = TrackBoxHead(_cfg(c), c["channels"], pooler=BO.OraclePooler(c["resolution"], c["scales"], c["sampling_ratio"]),
                        nms_fn=_cpu_nms)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in inp["params"].items()}, strict=True)
    head = head.eval()
    feats = [torch.from_numpy(f) for f in inp["features"]]
    with torch.no_grad():
        r = RefineTracks(head)(feats, [_proposals(inp["track_boxes"], inp["track_ids"], "cpu", inp["track_labels"],
                                                  inp["track_scores"], c["image_wh"])])[0]
    assert r.get_field("ids").numpy().tolist() == g["refine_ids"].tolist()
    np.testing.assert_allclose(r.bbox.numpy(), g["refine_bbox"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(r.get_field("scores").numpy(), g["refine_scores"], rtol=0, atol=1e-5)
