"""Soak of the pooling + correlation kernel with the correlation on the matrix pipe (round 6): random rois of every window
class (<= 32, 33..64, > 64 columns, all-border), channel tails, with and without an order hint; every launch twice (bitwise
equal: no race), pooled planes bitwise equal to the stand-alone pooler, responses within the fp64 bound of the tests.
    python measure/debug/fused_mm_soak.py [iterations]"""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import siammot_amd.ops as ops
dev = "cuda:0"
ops.load_library()
scales = (0.25, 0.125, 0.0625, 0.03125)
BOUND = 6e-7
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(2026)
worst, worst_fma, launches, hinted, wide, chunked = 0.0, 0.0, 0, 0, 0, 0
for it in range(iters):
    c = int(rs.choice([3, 8, 12, 17, 32, 64, 128]))
    n = int(rs.choice([1, 2, 5, 17, 30, 64, 100]))
    if n * c > 6400: n = max(1, 6400 // c)
    h, w = int(rs.choice([96, 184])), int(rs.choice([160, 320]))
    mag = float(10.0 ** rs.uniform(-3, 3))
    feats = [torch.from_numpy((rs.standard_normal((1, c, h // 2 ** l, w // 2 ** l)) * mag).astype(np.float32)).to(dev) for l in range(4)]
    kind = rs.randint(0, 4, n)
    bw = np.where(kind == 0, rs.uniform(10, 60, n), np.where(kind == 1, rs.uniform(60, 400, n), np.where(kind == 2, rs.uniform(500, 1200, n), rs.uniform(20, 100, n))))
    bh = np.where(kind == 2, rs.uniform(8, 20, n), bw * rs.uniform(0.5, 2.5, n))
    cx = np.where(kind == 3, rs.uniform(-3000, -1500, n), rs.uniform(0, w * 4.0, n))
    cy = rs.uniform(0, h * 4.0, n)
    b = torch.from_numpy(np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).astype(np.float32)).to(dev)
    use_hint = bool(rs.randint(0, 2)) and 2 <= n <= 256
    if use_hint:
        z, sr, hint = ops.emm_extract_cache(feats, b, 15, scales, 2, 512, 1.0, 0, hint=True)
    else:
        z, sr = ops.emm_extract_cache(feats, b, 15, scales, 2, 512, 1.0, 0)
        hint = None
    z = z * float(10.0 ** rs.uniform(-2, 2))

    def run():
        if hint is None:
            return ops.sr_xcorr_fused(feats, b, sr, z, 30, 15, scales, 2, 512, return_pooled=True)
        fl, fp, hs, ws_, sc = ops._level_arrays(feats, scales)
        pc = (ctypes.c_int * 4)(128, 64, 32, 16)
        r = torch.empty((n, c, 16, 16), device=dev)
        with ops.debug_library() as dbg:
            rc = dbg.smot_debug_sr_xcorr_fused_hint_fwd(ops._cast(fp), ops._cast(hs), ops._cast(ws_), ops._cast(pc), ops._cast(sc), 4, c,
                                                        ops._ptr(b), ops._ptr(sr), ops._ptr(z), n, ops._ptr(r), ops._ptr(hint), ops._stream(torch.device(dev)))
            assert rc == 0
        return r, None
    r1, p1 = run()
    r2, _ = run()
    assert torch.equal(r1, r2) or bool((torch.isnan(r1) == torch.isnan(r2)).all() and torch.equal(torch.nan_to_num(r1), torch.nan_to_num(r2))), "iteration %d: two launches differ" % it
    x = ops.roi_align_levels(feats, sr, b, 30, scales, 2, [128, 64, 32, 16])
    if p1 is not None:
        assert torch.equal(p1, x), "iteration %d: pooled planes differ from the stand-alone pooler" % it
    if hint is not None:
        ru, _ = ops.sr_xcorr_fused(feats, b, sr, z, 30, 15, scales, 2, 512, return_pooled=True)
        assert torch.equal(ru, r1), "iteration %d: hinted and un-hinted launches differ" % it
        hinted += 1
    x64 = x.double().reshape(1, n * c, 30, 30)
    z64 = z.double().reshape(n * c, 1, 15, 15)
    ref = torch.nn.functional.conv2d(x64, z64, groups=n * c).reshape(r1.shape)
    den = torch.nn.functional.conv2d(x64.abs(), z64.abs(), groups=n * c).reshape(r1.shape)
    err = (r1.double() - ref).abs()
    assert bool(torch.isfinite(r1).all()), "iteration %d: non-finite response" % it
    assert bool((err <= BOUND * den).all()), "iteration %d: bound exceeded, worst ratio %.3e" % (it, float((err / den.clamp_min(1e-300))[den > 0].max()))
    if bool((den > 0).any()):
        worst = max(worst, float((err / den.clamp_min(1e-300))[den > 0].max()))
        ef = (ops.xcorr_depthwise(x, z).double() - ref).abs()            # the fp32 FMA chain (stand-alone operator) on the same planes
        worst_fma = max(worst_fma, float((ef / den.clamp_min(1e-300))[den > 0].max()))
    launches += 2 + (1 if hint is not None else 0)
    wide += int(((kind == 1)).sum()); chunked += int((kind == 2).sum())
print(json.dumps({"iterations": iters, "launches": launches, "hinted_iterations": hinted, "rois_with_large_boxes": wide, "rois_with_extreme_aspect": chunked,
                  "worst_err_over_sum_abs": worst, "worst_err_of_the_fp32_fma_operator_on_the_same_planes": worst_fma, "bound": BOUND, "result": "every launch repeated bit for bit; pooled planes = stand-alone pooler; hinted = un-hinted; all within the bound"}))
