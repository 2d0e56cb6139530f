"""Frame pre-processing: the oracle is pinned against Pillow itself (the library the
reference calls), the product-side tables against the oracle, the HIP kernel against PIL + torch."""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as PO

PIL_Image = pytest.importorskip("PIL.Image")

SIZES = [((720, 1280), (704, 1280)), ((90, 160), (64, 128)), ((1080, 1920), (704, 1280)), ((256, 256), (800, 800)),
         ((480, 640), (96, 128)), ((100, 37), (31, 77))]


def _frame(h, w, seed=0):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)


def _torch_chain(img_u8, mean, std, to_bgr255):
    """ToTensor + Normalize with the torch ops the reference's transforms run (transforms.py [UPSTREAM])."""
    t = torch.from_numpy(np.array(img_u8, copy=True)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    if to_bgr255:
        t = t[[2, 1, 0]] * 255
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
    return t.sub_(m).div_(s)


@pytest.mark.parametrize("in_hw,out_hw", SIZES)
def test_oracle_resize_equals_pillow(in_hw, out_hw):
    a = _frame(*in_hw)
    ref = np.asarray(PIL_Image.fromarray(a, "RGB").resize((out_hw[1], out_hw[0]), PIL_Image.BILINEAR))
    assert np.array_equal(PO.resize_bilinear_u8(a, out_hw), ref)


def test_oracle_normalize_equals_torch_ops():
    a = _frame(40, 50, 3)
    for mean, std, bgr in (((0.485, 0.456, 0.406), (0.229, 0.224, 0.225), False),
                           ((102.9801, 115.9465, 122.7717), (1.0, 1.0, 1.0), True)):
        got = PO.to_tensor_normalize(a, mean, std, bgr)
        assert np.array_equal(got, _torch_chain(a, mean, std, bgr).numpy())


def test_get_size_follows_the_reference_rule():
    from siammot_amd.preprocess import get_size
    for wh in ((1280, 720), (1920, 1080), (256, 256), (720, 1280), (641, 480), (3840, 2160)):
        assert get_size(wh, 800, 1280, 32) == PO.get_size(wh, 800, 1280, 32)
    assert get_size((1280, 720), 800, 1280, 32) == (704, 1280)       # SURVEY.md §8d net-input sizes
    assert get_size((1920, 1080), 800, 1280, 32) == (704, 1280)
    assert get_size((256, 256), 800, 1280, 32) == (800, 800)


@pytest.mark.parametrize("n_in,n_out", [(720, 704), (1280, 1280), (1920, 1280), (256, 800), (37, 77), (2160, 704), (5, 3)])
def test_product_tables_equal_the_oracle_tables(n_in, n_out):
    """Two independent evaluations of Pillow's coefficient rule (vectorised numpy vs the scalar loop)."""
    from siammot_amd.preprocess import max_tile_rows, resample_tables
    b, k = resample_tables(n_in, n_out)
    if n_in == n_out:
        assert (b[:, 0] == np.arange(n_out)).all() and (b[:, 1] == 1).all() and (k == 1 << 22).all()
        return
    bo, ko = PO.resample_coeffs(n_in, n_out)
    assert np.array_equal(b, bo) and np.array_equal(k, ko)
    assert (k.sum(1) - (1 << 22)).__abs__().max() <= k.shape[1]          # weights sum to one (up to rounding)
    assert max_tile_rows(b) >= b[:, 1].max()


@pytest.mark.gpu
@pytest.mark.parametrize("in_hw,out_hw", SIZES)
@pytest.mark.parametrize("bgr", [False, True])
def test_kernel_equals_pillow_plus_torch(in_hw, out_hw, bgr):
    import siammot_amd.ops as ops
    from siammot_amd.preprocess import FramePreprocessor
    mean, std = ((102.9801, 115.9465, 122.7717), (1.0, 1.0, 1.0)) if bgr else ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    pre = FramePreprocessor(800, 1280, 32, mean, std, bgr, device="cuda:0")
    a = _frame(*in_hw, seed=7)
    got = ops.preprocess_frame(pre.upload(a), pre.tables(in_hw, out_hw), out_hw, mean, std, bgr)
    ref_img = np.asarray(PIL_Image.fromarray(a, "RGB").resize((out_hw[1], out_hw[0]), PIL_Image.BILINEAR))
    ref = _torch_chain(ref_img, mean, std, bgr)
    assert tuple(got.shape) == (3,) + tuple(out_hw)
    assert torch.equal(got.cpu(), ref), "max abs diff %g" % float((got.cpu() - ref).abs().max())


@pytest.mark.gpu
def test_preprocessor_call_follows_the_resize_rule_and_rejects_bad_input():
    import siammot_amd.ops as ops
    from siammot_amd.config import get_default_cfg
    from siammot_amd.preprocess import FramePreprocessor
    pre = FramePreprocessor.from_cfg(get_default_cfg(), device="cuda:0")
    a = _frame(720, 1280, 11)
    out = pre(a)
    assert tuple(out.shape) == (3, 704, 1280) and out.dtype == torch.float32 and out.is_cuda
    ref = torch.from_numpy(PO.preprocess(a, 800, 1280, 32, pre.pixel_mean, pre.pixel_std, False))
    assert torch.equal(out.cpu(), ref)
    out2 = pre(torch.from_numpy(a).to("cuda:0"))                 # device frames skip the staging copy
    assert torch.equal(out2, out)
    with pytest.raises(RuntimeError, match="uint8"):
        ops.preprocess_frame(torch.zeros((4, 4, 3), device="cuda:0"), pre.tables((4, 4), (4, 4)), (4, 4),
                             pre.pixel_mean, pre.pixel_std, False)
    with pytest.raises(RuntimeError, match="LDS"):               # 8 output rows would need > 64 KiB of input rows
        big = torch.zeros((4000, 8, 3), dtype=torch.uint8, device="cuda:0")
        ops.preprocess_frame(big, pre.tables((4000, 8), (8, 8)), (8, 8), pre.pixel_mean, pre.pixel_std, False)
