"""CPU-side checks: the C-ABI library loads and exports everything include/smot_emm.h declares,
the host mirror keeps the reference's interface, and the product path refuses to run without a
device (no CPU fallback).  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(debug=False):
    """Symbols include/smot_emm.h declares; with ``debug`` also those inside ``#ifdef SMOT_DEBUG`` blocks (the
    measurement library's extra entry points)."""
    src = open(os.path.join(ROOT, "include", "smot_emm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    if not debug:
        src = re.sub(r"#ifdef SMOT_DEBUG.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smot_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import siammot_amd.ops as ops
    lib_path = ops.LIB_PATH
    if not os.path.exists(lib_path):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(lib_path)
    declared = _header_symbols()
    assert len(declared) >= 8
    for name in declared:
        assert hasattr(lib, name), "libsmot_emm.so lacks %s declared in include/smot_emm.h" % name
    assert sorted(ops.EXPORTED_SYMBOLS) == declared       # the ctypes table covers the whole header
    ops.load_library()
    assert lib.smot_abi_version() == ops.ABI_VERSION
    assert lib.smot_build_info() == 0                     # the product library is not a measurement build
    assert lib.smot_memory_carry_max_rows() == ops.MEMORY_CARRY_MAX_ROWS     # (the binding's row array has that many entries)
    assert not hasattr(lib, "smot_debug_set_knob")


def test_product_library_reads_no_environment_and_has_no_ablation_kernels():
    """VERDICT r1 #8 / ADVICE: switches that change which kernel runs (or make it compute something else) must
    not exist in the product library, and nothing on the launch path may call getenv."""
    import subprocess
    import siammot_amd.ops as ops
    nm = subprocess.run(["nm", "-D", "--undefined-only", ops.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in nm, "libsmot_emm.so imports getenv"
    raw = open(ops.LIB_PATH, "rb").read()
    for name in (b"SMOT_FUSED_ABL", b"SMOT_WINO_ABL", b"SMOT_TOWER_ABL", b"SMOT_NO_FUSE", b"SMOT_DECODE_SPLIT",
                 b"SMOT_XCORR_VARIANT", b"xcorr_dw_wave_kernel", b"xcorr_dw_mfma_kernel", b"xcorr_dw_pk_kernel"):
        assert name not in raw, "%s found in the product library" % name.decode()


def test_measurement_library_exports_the_debug_entry_points():
    import siammot_amd.ops as ops
    if not os.path.exists(ops.DEBUG_LIB_PATH):
        pytest.skip("measurement library not built")
    lib = ctypes.CDLL(ops.DEBUG_LIB_PATH)
    for name in _header_symbols(debug=True):
        assert hasattr(lib, name), "libsmot_emm_debug.so lacks %s" % name
    assert sorted(ops.EXPORTED_SYMBOLS + ops.DEBUG_EXPORTED_SYMBOLS) == _header_symbols(debug=True)
    assert lib.smot_build_info() & 1
    lib.smot_debug_set_knob.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.smot_last_error.restype = ctypes.c_char_p
    assert lib.smot_debug_set_knob(b"SMOT_DECODE_SPLIT", b"3") == -1      # validated where it is set
    assert b"SMOT_DECODE_SPLIT" in lib.smot_last_error()
    assert lib.smot_debug_set_knob(b"SMOT_NOT_A_SWITCH", b"1") == -1
    assert lib.smot_debug_set_knob(b"SMOT_DECODE_SPLIT", b"2") == 0
    assert lib.smot_debug_set_knob(b"SMOT_DECODE_SPLIT", b"0") == 0
    with ops.debug_library(SMOT_XCORR_VARIANT="mfma") as dbg:
        assert ops.load_library() is dbg
    assert ops.load_library().smot_build_info() == 0      # product library restored


def test_argument_errors_are_reported_without_a_device():
    """Argument validation happens before any launch, so it is testable on CPU."""
    import siammot_amd.ops as ops
    lib = ops.load_library()
    null = ctypes.c_void_p(0)
    rc = lib.smot_xcorr_dw_fwd(null, null, null, 2, 4, 10, 15, null)          # Rx < Rz
    assert rc == -1 and b"xcorr" in lib.smot_last_error()
    rc = lib.smot_xcorr_dw_fwd(null, null, null, 0, 4, 30, 15, null)          # empty batch is fine
    assert rc == 0
    rc = lib.smot_emm_decode_fwd(null, null, null, null, 1, 16, 12, 30, 15, 512.0, 0.6, 0.4, 1, 0.0, 0.0,
                                 null, null, null, null, null)                  # up not a power of two
    assert rc == -2 and b"up=12" in lib.smot_last_error()
    rc = lib.smot_emm_decode_fwd(null, null, null, null, 1, 16, 16, 30, 14, 512.0, 0.6, 0.4, 1, 0.0, 0.0,
                                 null, null, null, null, null)                  # even rz / Ho mismatch
    assert rc == -1
    rc = lib.smot_emm_predictor_fwd(null, 1, 100, 16, *([null] * 12), 32, 1e-5, null, null, null, null)
    assert rc == -1 and b"divisible" in lib.smot_last_error()
    # fp32 image + (C % 32 == 0) the two-part fp16 image: C/32 + 1 rotated blocks of 8192 floats per 16-channel tile + a
    # four-word header (largest |w|, the scale's inverse)
    assert lib.smot_emm_tower_pack_floats(128) == 2 * 128 * 128 * 16 + 16 * 5 * 8192 + 4
    assert lib.smot_emm_tower_pack_floats(48) == 2 * 48 * 48 * 16 and lib.smot_emm_tower_pack_floats(100) == 0
    # which form of the towers a track count gets (host-side arithmetic): since round 6 two tiles per workgroup on two-part
    # fp16 operands at every track count (one form: a track's logits do not depend on how many tracks the frame has); no
    # packed path for channel counts that are not powers of two
    assert [lib.smot_emm_tower_form(n, 128, 16) for n in (1, 16, 17, 30, 100)] == [3, 3, 3, 3, 3]
    assert lib.smot_emm_tower_form(30, 96, 16) == 0 and lib.smot_emm_tower_form(30, 128, 15) == 0
    assert lib.smot_emm_tower_form(30, 128, 29) == 3
    assert lib.smot_emm_tower_pack(null, null, 100, null, null) == -1
    assert lib.smot_emm_decode_ws_floats(16, 16) == 2 * 5 * 17 + 2        # 17 band records of 5 words + a ticket


def test_product_path_has_no_cpu_fallback():
    import siammot_amd.ops as ops
    with pytest.raises(RuntimeError, match="device"):
        ops.xcorr_depthwise(torch.zeros(1, 2, 30, 30), torch.zeros(1, 2, 15, 15))
    with pytest.raises(RuntimeError, match="device"):
        ops.search_region(torch.zeros(1, 4), 512, 1.0, 0)
    with pytest.raises(RuntimeError, match="not found"):
        ops.load_library("/nonexistent/libsmot_emm.so")


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "siam-mot_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, "%s mentions the oracle" % f


def test_oracle_never_imports_the_product():
    """The golden generators satisfy the reference's ``maskrcnn_benchmark`` imports with the oracle's OWN restatement
    (oracle/ref_structures.py), not with the package under test.  The one script that is ABOUT the
    product — check_reference_dropin.py, which plugs siammot_amd.emm.EMM into the reference's TrackHead — is exempt."""
    odir = os.path.join(ROOT, "oracle")
    for f in sorted(os.listdir(odir)):
        if not f.endswith(".py") or f == "check_reference_dropin.py":
            continue
        for ln in open(os.path.join(odir, f)):
            code = ln.split("#")[0]
            assert not (("import" in code) and "siammot_amd" in code), "oracle/%s imports the product: %s" % (f, ln.strip())
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_structures_probe", os.path.join(odir, "ref_structures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from siammot_amd import structures as S
    assert m.BoxList is not S.BoxList
    # the two restatements of upstream's container agree where the reference touches it
    import torch
    b = torch.tensor([[-5.0, 3.0, 2000.0, 40.0], [10.0, 20.0, 5.0, 800.0], [1.0, 2.0, 3.0, 4.0]])
    for cls in (m.BoxList, S.BoxList):
        bl = cls(b.clone(), (1280, 704), "xyxy")
        bl.add_field("ids", torch.arange(3))
        kept = bl.clip_to_image(remove_empty=True)
        assert bl.bbox.tolist() == [[0.0, 3.0, 1279.0, 40.0], [10.0, 20.0, 5.0, 703.0], [1.0, 2.0, 3.0, 4.0]]
        assert kept.get_field("ids").tolist() == [0, 2] and len(bl) == 3          # filtered COPY; the original keeps every row
        assert bl.area().tolist() == [1280.0 * 38.0, -4.0 * 684.0, 9.0]
        x = bl.convert("xywh")
        assert x.bbox[2].tolist() == [1.0, 2.0, 3.0, 3.0] and x.convert("xyxy").bbox[2].tolist() == [1.0, 2.0, 3.0, 4.0]
    cat2 = m.cat_boxlist([m.BoxList(b[:1].clone(), (1280, 704)), m.BoxList(b[1:].clone(), (1280, 704))])
    assert len(cat2) == 3 and m.cat([b]) is b


def test_emm_module_mirrors_reference_interface():
    import inspect
    from siammot_amd.config import get_default_cfg
    from siammot_amd.emm import EMM, EMMFeatureExtractor, EMMPredictor, SRPooler
    from siammot_amd.registry import SIAMESE_TRACKER
    from siammot_amd.track_utils import build_track_utils
    cfg = get_default_cfg()
    tu = build_track_utils(cfg)
    assert (tu.search_expansion, tu.min_search_wh, tu.pad_pixels) == (1.0, 0, 512)
    emm = SIAMESE_TRACKER["EMM_HIP"](cfg, tu)
    assert isinstance(emm, EMM)
    # reference signatures: track_core.py:16,28,81
    assert list(inspect.signature(EMM.__init__).parameters) == ["self", "cfg", "track_utils"]
    assert list(inspect.signature(EMM.forward).parameters) == ["self", "features", "boxes", "sr", "targets",
                                                               "template_features"]
    assert list(inspect.signature(EMM.extract_cache).parameters) == ["self", "features", "detection"]
    assert list(inspect.signature(SRPooler.forward).parameters)[:4] == ["self", "x", "boxes", "sr"]
    assert list(inspect.signature(EMMFeatureExtractor.forward).parameters)[:4] == ["self", "x", "proposals", "sr"]
    # reference state_dict keys (SURVEY.md §5, Appendix C probe4)
    keys = sorted(emm.state_dict().keys())
    assert keys == sorted("predictor." + k for k in (
        "cls_tower.0.weight", "cls_tower.1.weight", "cls_tower.1.bias", "reg_tower.0.weight",
        "reg_tower.1.weight", "reg_tower.1.bias", "cls.weight", "cls.bias", "center.weight", "center.bias",
        "reg.weight", "reg.bias"))
    assert emm.rx == 30 and emm.rz == 15 and emm.pad_pixels == 512 and emm.sigma == 0.4
    assert isinstance(emm.predictor, EMMPredictor)
    sd = emm.state_dict()
    assert sd["predictor.cls_tower.0.weight"].shape == (128, 128, 3, 3)
    assert sd["predictor.reg.weight"].shape == (4, 128, 3, 3)
    # R-50-FPN body picks RESNETS.BACKBONE_OUT_CHANNELS (feature_extractor.py:49-50)
    cfg2 = get_default_cfg(conv_body="R-50-FPN")
    assert EMMPredictor(cfg2).cls.weight.shape == (2, 256, 3, 3)


def test_boxlist_conventions():
    from siammot_amd.structures import BoxList, cat_boxlist
    b = BoxList(torch.tensor([[-5.0, 10.0, 700.0, 20.0], [30.0, 40.0, 30.0, 80.0], [1.0, 2.0, 3.0, 4.0]]),
                (640, 480), "xyxy")
    b.add_field("ids", torch.tensor([7, 8, 9]))
    kept = b.clip_to_image(remove_empty=True)
    assert b.bbox[0].tolist() == [0.0, 10.0, 639.0, 20.0]          # clamped in place to [0, W-1]
    assert kept.get_field("ids").tolist() == [7, 9]                 # zero-width box dropped in the copy only
    assert len(b) == 3
    assert b.area().tolist()[2] == 9.0                              # +1 convention
    xywh = b.convert("xywh")
    assert xywh.bbox[2].tolist() == [1.0, 2.0, 3.0, 3.0]
    assert torch.equal(xywh.convert("xyxy").bbox, b.bbox)
    r = b.resize((1280, 960))
    assert r.bbox[2].tolist() == [2.0, 4.0, 6.0, 8.0] and r.size == (1280, 960)
    c = cat_boxlist([b[[0]], b[[2]]])
    assert len(c) == 2 and c.get_field("ids").tolist() == [7, 9]


def test_track_utils_matches_oracle_geometry():
    from oracle import emm_oracle as O
    from siammot_amd.structures import BoxList
    from siammot_amd.track_utils import TrackUtils
    tu = TrackUtils(search_expansion=1.0, min_search_wh=0, pad_pixels=512)
    boxes = torch.tensor([[10.0, 20.0, 73.5, 148.0], [300.0, 100.0, 459.0, 419.0]])
    bl = BoxList(boxes.clone(), (1280, 704))
    bl.add_field("ids", torch.tensor([0, 1]))
    sr = tu.extend_bbox(tu.update_boxes_in_pad_images([bl]))[0]
    assert sr.size == [2304, 1728] and sr.get_field("ids").tolist() == [0, 1]
    assert torch.equal(sr.bbox, O.search_region(boxes, 512, 1.0, 0))
    assert [tu.pad_level_cells(i) for i in range(5)] == [128, 64, 32, 16, 8]
    f = tu.pad_feature((torch.ones(1, 2, 4, 6), torch.ones(1, 2, 2, 3)))
    assert f[0].shape == (1, 2, 260, 262) and f[1].shape == (1, 2, 130, 131)


def test_bench_workload_definition():
    import bench
    b = bench.synthetic_boxes(30, (1280, 704))
    assert b.shape == (30, 4)
    assert (b[:, 0] >= 0).all() and (b[:, 1] >= 0).all() and (b[:, 2] < 1280).all() and (b[:, 3] < 704).all()
    from oracle import emm_oracle as O
    assert sorted(set(O.level_mapper(b).tolist())) == [0, 1, 2]
    # algorithmic bytes of the graded kernel: 707,072 B per track at C=128
    assert 4 * 128 * (30 * 30 + 15 * 15 + 16 * 16) == 707072


def test_result_wire_format_matches_the_reference_contract():
    """siammot/utils/boxlists_to_entities.py:6-37 restated with one host copy per frame."""
    from siammot_amd.results import boxlists_to_entities, mot_challenge_rows, to_original_xywh
    from siammot_amd.structures import BoxList
    b = BoxList(torch.tensor([[10.0, 20.0, 29.0, 59.0], [0.5, 0.25, 100.5, 50.25]]), (1280, 704), mode="xyxy")
    b.add_field("scores", torch.tensor([0.9, 0.123456]))
    b.add_field("labels", torch.tensor([1, 2]))
    b.add_field("ids", torch.tensor([7, -1]))
    out = to_original_xywh(b, (1280, 720))
    ents = boxlists_to_entities([out, out], 5, [0.2, 0.24], class_table=["person", "vehicle"])
    assert len(ents) == 4
    e = ents[0]
    ref_box = out.bbox[0].tolist()
    assert e.bbox == ref_box and e.id == 7 and e.frame_num == 5 and e.time == 0.2
    assert e.confidence == out.get_field("scores")[0].item() and e.labels == {"person": e.confidence}
    assert ents[1].id == -1 and list(ents[1].labels) == ["vehicle"] and ents[3].frame_num == 6
    assert abs(e.bbox[2] - 20.0) < 1e-5 and abs(e.bbox[3] - 40.0 * 720 / 704) < 0.2     # xywh with the +1 convention
    rows = mot_challenge_rows(ents)
    assert len(rows) == 2 and rows[0].startswith("6,7,10.00,")
    empty = BoxList(torch.zeros((0, 4)), (10, 10))
    empty.add_field("scores", torch.zeros(0))
    empty.add_field("labels", torch.zeros(0, dtype=torch.int64))
    assert boxlists_to_entities(empty, 0, [0.0]) == []


def test_wait_host_record_polls_then_falls_back_to_the_event():
    """The tracking loop's one synchronisation (ops.wait_host_record): returns as soon as the record's completion
    word is non-zero, falls back to the event behind the launch when it is not, and raises when even the event's
    completion leaves the record incomplete (a faulted kernel must not read as an empty frame)."""
    import torch
    from siammot_amd import ops

    class Ev(object):
        def __init__(self, rec=None, value=0):
            self.rec, self.value, self.calls = rec, value, 0

        def synchronize(self):
            self.calls += 1
            if self.rec is not None:
                self.rec[3] = self.value
    done = torch.zeros(16, dtype=torch.int32)
    done[3] = 7
    ev = Ev()
    ops.wait_host_record(done, ev)
    assert ev.calls == 0                                   # polled: the event was never touched
    late = torch.zeros(16, dtype=torch.int32)
    ev = Ev(late, 3)
    ops.wait_host_record(late, ev, spins=50)
    assert ev.calls == 1 and int(late[3]) == 3
    never = torch.zeros(16, dtype=torch.int32)
    with pytest.raises(RuntimeError):
        ops.wait_host_record(never, Ev(never, 0), spins=50)


def test_order_hint_is_bound_to_the_tensors_it_was_made_from():
    """``emm.OrderHint`` (the scheduling hint ``extract_cache`` leaves for the next ``forward``) is handed to the kernel only
    for the very tensors it describes: identity of both box tensors, their in-place version counters, the row count and the
    pooler scales are all checked — pure host logic, no kernel involved."""
    import torch
    from siammot_amd.emm import OrderHint
    from siammot_amd.structures import BoxList
    boxes, sr = torch.rand(5, 4), torch.rand(5, 4)
    hint = torch.zeros(5, 8)
    scales = (0.25, 0.125)
    srb = BoxList(sr, (100, 100))
    assert OrderHint.lookup(srb, boxes, srb.bbox, scales) is None                   # nothing attached
    srb.order_hint = OrderHint(hint, boxes, srb.bbox, scales)
    assert OrderHint.lookup(srb, boxes, srb.bbox, scales) is hint
    assert OrderHint.lookup(srb, boxes.clone(), srb.bbox, scales) is None           # other template boxes (equal values)
    assert OrderHint.lookup(srb, boxes, srb.bbox.clone(), scales) is None           # a copy of the search regions
    assert OrderHint.lookup(srb, boxes, srb.bbox, (0.25, 0.0625)) is None           # another pooler
    assert OrderHint.lookup(srb, boxes[:4], srb.bbox, scales) is None               # another row count (a view: other object)
    srb.bbox[0, 0] += 1.0                                                           # in-place edit: the hint's copy is stale
    assert OrderHint.lookup(srb, boxes, srb.bbox, scales) is None
    srb.order_hint = OrderHint(hint, boxes, srb.bbox, scales)                       # re-made after the edit: valid again
    assert OrderHint.lookup(srb, boxes, srb.bbox, scales) is hint
    boxes.mul_(1.0)                                                                 # any in-place op on the template boxes
    assert OrderHint.lookup(srb, boxes, srb.bbox, scales) is None


def test_frame_args_buffer_matches_the_header_struct():
    """``ops.FrameArgs`` packs ``smot_frame_args`` by hand (one struct.pack_into per frame): its field order, the
    pointers-then-ints-then-floats grouping and the total size must be exactly the C struct of include/smot_emm.h — a
    drift would corrupt every argument behind it without any error."""
    import re
    import siammot_amd.ops as ops
    src = open(os.path.join(ROOT, "include", "smot_emm.h")).read()
    body = src[src.index("typedef struct smot_frame_args {"):src.index("} smot_frame_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        kind = "ptr" if "*" in decl else ("int" if re.match(r"int\b", decl) else ("float" if re.match(r"float\b", decl) else None))
        assert kind is not None, decl
        names = re.sub(r"^(const\s+)?(float|int64_t|int)\s*(const\s*)?\**\s*(const\s*)?\**", "", decl) if kind == "ptr" else decl.split(None, 1)[1]
        for n in names.split(","):
            fields.append((n.strip().lstrip("*").strip(), kind))
    assert [n for n, k in fields if k == "ptr"] == list(ops._FRAME_PTRS)
    assert [n for n, k in fields if k == "int"] == list(ops._FRAME_INTS)
    assert [n for n, k in fields if k == "float"] == list(ops._FRAME_FLOATS)
    kinds = [k for _, k in fields]
    assert kinds == sorted(kinds, key={"ptr": 0, "int": 1, "float": 2}.get)         # grouped: no padding inside
    assert ops.FrameArgs._FMT.size == 8 * len(ops._FRAME_PTRS) + 4 * (len(ops._FRAME_INTS) + len(ops._FRAME_FLOATS))
    a = ops.FrameArgs()
    a.n_trk, a.clip_w, a.det_labels = 7, 1280.0, None
    a.pack()
    import struct
    vals = ops.FrameArgs._FMT.unpack_from(a._buf, 0)
    assert vals[len(ops._FRAME_PTRS) + ops._FRAME_INTS.index("n_trk")] == 7
    assert vals[len(ops._FRAME_PTRS) + len(ops._FRAME_INTS) + ops._FRAME_FLOATS.index("clip_w")] == 1280.0
    # the two per-frame ranges a packed block gets rewritten with (TrackingLoop._step_native): every value must land in
    # the field it is meant for and nothing else may move
    P, I, F = ops._FRAME_PTRS, ops._FRAME_INTS, ops._FRAME_FLOATS
    head = P[P.index("head_ws"):P.index("trk_conf") + 1]
    rest = P[P.index("trk_conf") + 1:]
    assert head == ("head_ws", "tpl_boxes", "sr", "templates", "order_hint", "trk_ids", "trk_labels", "trk_boxes", "trk_conf")
    assert rest[0] == "refine_ws" and rest[20] == "next_order_hint" and rest[-1] == "carry_scores" and len(rest) == 27
    a.poke_head(tuple(1000 + i for i in range(len(head))), 11, ops.STAGE_HEAD)
    a.poke_rest(tuple(2000 + i for i in range(len(rest))), ops.STAGE_SOLVE | ops.STAGE_EXTRACT, 5, (0.25, 0.5, 0.75),
                carry=(24, 6, 24))
    v2 = ops.FrameArgs._FMT.unpack_from(a._buf, 0)
    want = list(vals)
    for i, n in enumerate(head):
        want[P.index(n)] = 1000 + i
    for i, n in enumerate(rest):
        want[P.index(n)] = 2000 + i
    want[len(P) + I.index("n_trk")], want[len(P) + I.index("stages")], want[len(P) + I.index("n_det")] = 11, 12, 5
    for n, x in zip(("carry_src_row0", "carry_rows", "carry_dst_row0"), (24, 6, 24)):
        want[len(P) + I.index(n)] = x
    for n, x in zip(("track_thresh", "start_thresh", "resume_thresh"), (0.25, 0.5, 0.75)):
        want[len(P) + len(I) + F.index(n)] = x
    assert list(v2) == want
    src_c = open(os.path.join(ROOT, "include", "smot_emm.h")).read()
    for name, val in (("SMOT_STAGE_HEAD", ops.STAGE_HEAD), ("SMOT_STAGE_REFINE", ops.STAGE_REFINE),
                      ("SMOT_STAGE_SOLVE", ops.STAGE_SOLVE), ("SMOT_STAGE_EXTRACT", ops.STAGE_EXTRACT),
                      ("SMOT_STAGE_CARRY", ops.STAGE_CARRY)):
        assert re.search(r"#define\s+%s\s+%d\b" % (name, val), src_c), name


def test_matrix_pipe_correlation_operand_layout_is_the_correlation():
    """csrc/xcorr_f16x2.h on paper (no device): the 30x30 * 15x15 -> 16x16 correlation as 15 products A_i [16 x 32] . B_i [32 x 16]
    with B_i[c][x] = Z[i][c - x] read as FOUR ALIGNED DWORDS from an even or an odd copy of template row i.  The test rebuilds
    both copies and every lane's window from the header's own address formulas (clamp, parity select, dword index, the value a
    lane packs with its right neighbour's) and checks (1) window = Toeplitz definition for every lane and row, exactly,
    (2) two fp16 parts of power-of-two-scaled operands with three part products reproduce an fp64 correlation within the
    bound the GPU tests hold the kernel to."""
    rs = np.random.RandomState(5)
    X = rs.standard_normal((30, 30)).astype(np.float32) * 37.0
    Z = rs.standard_normal((15, 15)).astype(np.float32) * 0.013

    def scale_of(a):                                   # xh_pow2_scale: largest |a| * s in [2^13, 2^14)
        e = (np.abs(a).max().view(np.uint32) >> 23) & 0xFF
        k = int(np.clip(140 - int(e), -100, 100))
        return np.float32(2.0) ** k

    def split(a):                                      # a1 = RNE11(a), a2 = RNE11(a - a1)
        a1 = a.astype(np.float16)
        a2 = (a - a1.astype(np.float32)).astype(np.float16)
        return a1, a2

    sx, sz = scale_of(X), scale_of(Z)
    x1, x2 = split(X * sx)
    z1, z2 = split(Z * sz)
    # ---- the template's rows as xh_template_store writes them: lane j of a row packs (Z[j], Z[j+1]) — lane 15 holds 0 and
    # its right neighbour is lane 0 — even j -> dword (8 + j) / 2 of the even copy, odd j -> dword (7 + j) / 2 of the odd copy
    # (lane 15 -> dword 3); rows of 16 dwords, everything else zero
    def copies(zp):
        even = np.zeros((15, 32), np.float16)          # halves of the even copy: local half h at index h
        odd = np.zeros((15, 32), np.float16)           # halves of the odd copy: index h holds local half h + 1
        for i in range(15):
            vals = list(zp[i]) + [np.float16(0)]       # lane 15: 0
            for j in range(16):
                mine, nxt = vals[j], vals[(j + 1) % 16]
                if j % 2 == 0:
                    dw = (8 + j) >> 1
                    even[i, 2 * dw], even[i, 2 * dw + 1] = mine, nxt
                else:
                    dw = 3 if j == 15 else (7 + j) >> 1
                    odd[i, 2 * dw], odd[i, 2 * dw + 1] = mine, nxt
        return even, odd

    out = np.zeros((16, 16), np.float64)
    ref_rows = np.zeros((15, 48), np.float64)          # the 48-half zero-padded rows of the definition: template at 16..30
    for (zp, tag) in ((z1, 1), (z2, 2)):
        even, odd = copies(zp)
        ref_rows[:] = 0
        ref_rows[:, 16:31] = zp.astype(np.float64)
        for i in range(15):
            B = np.zeros((32, 16), np.float64)
            for lane in range(64):
                x, kq = lane & 15, lane >> 4
                sw = min(max(8 * kq - x + 16, 8), 31)
                loc = sw - 8
                if loc & 1:
                    window = odd[i, 2 * ((loc - 1) >> 1): 2 * ((loc - 1) >> 1) + 8]
                else:
                    window = even[i, 2 * (loc >> 1): 2 * (loc >> 1) + 8]
                assert window.shape == (8,)
                want = ref_rows[i, 8 * kq - x + 16: 8 * kq - x + 24]          # B_i[8 kq + e][x] = Z[i][8 kq + e - x]
                assert np.array_equal(window.astype(np.float64), want), (tag, i, lane)
                B[8 * kq: 8 * kq + 8, x] = window
            for (xp, xtag) in ((x1, 1), (x2, 2)):
                if tag == 2 and xtag == 2:
                    continue                            # x2 z2 is dropped
                A = np.zeros((16, 32), np.float64)      # rows i .. i + 15, columns 30, 31 zero
                A[:, :30] = xp[i:i + 16].astype(np.float64)
                out += A @ B
    out = out / float(sx) / float(sz)
    ref = np.zeros((16, 16))
    den = np.zeros((16, 16))
    for y in range(16):
        for x in range(16):
            w = X[y:y + 15, x:x + 15].astype(np.float64)
            ref[y, x] = (w * Z).sum()
            den[y, x] = (np.abs(w) * np.abs(Z)).sum()
    assert np.abs(out - ref).max() <= 2.5e-7 * den.max() and (np.abs(out - ref) <= 6e-7 * den).all(), np.abs((out - ref) / den).max()
