"""Round 6: the fused pooling + correlation kernel with the correlation on the matrix pipe (xcorr_f16x2.h) against an fp64
evaluation of the same pooled planes, beside the fp32 FMA kernel's error (the stand-alone operator)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "siam-mot_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from siammot_amd import ops as OPS

DEV = "cuda:0"
SCALES = (0.25, 0.125, 0.0625, 0.03125)


def err64(resp, pooled, z):
    n, c = pooled.shape[:2]
    x64 = pooled.double().reshape(1, n * c, 30, 30)
    z64 = z.double().reshape(n * c, 1, 15, 15)
    ref = torch.nn.functional.conv2d(x64, z64, groups=n * c).reshape(n, c, 16, 16)
    den = torch.nn.functional.conv2d(x64.abs(), z64.abs(), groups=n * c).reshape(n, c, 16, 16)
    e = (resp.double() - ref).abs() / den.clamp_min(1e-300)
    e = torch.where(den > 0, e, torch.zeros_like(e))
    return float(e.max())


def main():
    rs = np.random.RandomState(5)
    out = []
    for (n, c, h, w, mag) in ((30, 128, 184, 320, 1.0), (7, 20, 96, 128, 1e-4), (5, 9, 96, 128, 3e3), (100, 128, 184, 320, 1.0)):
        feats = [torch.from_numpy((rs.standard_normal((1, c, h // 2 ** l, w // 2 ** l)) * mag).astype(np.float32)).to(DEV) for l in range(4)]
        wh = rs.uniform(20, 300, (n, 2))
        xy = rs.uniform(0, 1, (n, 2)) * np.array([w * 4.0, h * 4.0])
        b = torch.from_numpy(np.concatenate([xy - wh / 2, xy + wh / 2], 1).astype(np.float32)).to(DEV)
        sr = OPS.search_region(b, 512, 1.0, 0)
        z = torch.from_numpy(rs.standard_normal((n, c, 15, 15)).astype(np.float32)).to(DEV)
        r, p = OPS.sr_xcorr_fused(feats, b, sr, z, 30, 15, SCALES, 2, 512, return_pooled=True)
        r2 = OPS.xcorr_depthwise(p, z)
        torch.cuda.synchronize()
        out.append({"n": n, "c": c, "mag": mag, "fused_err_over_sum_abs": err64(r, p, z), "fma_err_over_sum_abs": err64(r2, p, z),
                    "max_abs_diff_fused_vs_fma": float((r - r2).abs().max()), "resp_absmax": float(r2.abs().max()),
                    "nonfinite": int((~torch.isfinite(r)).sum())})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
