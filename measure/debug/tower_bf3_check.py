"""Split form of the Winograd towers (round 6: fp16 x 2; rounds 4-5: bf16 x 3) (SMOT_TOWER_BF3=1, measurement library) against the fp32 form:
   logit error of both against an fp64 evaluation of the same predictor, and the tower kernel's duration.  JSON lines."""
import json, os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
import siammot_amd.ops as ops
dev = "cuda:0"
rs = np.random.RandomState(0)
boxes = np.array([[0, 0, 80, 120]], dtype=np.float32)
Pn = gi.predictor_params(rs, 128, boxes)
P = {k: torch.from_numpy(v).to(dev) for k, v in Pn.items()}


def ref64(resp):
    x = resp.double()
    p = {k: v.double() for k, v in P.items()}
    out = []
    feats = {}
    for t in ("cls_tower", "reg_tower"):
        y = F.conv2d(x, p[t + ".0.weight"], padding=1)
        y = F.group_norm(y, 32, p[t + ".1.weight"], p[t + ".1.bias"], 1e-5)
        feats[t] = F.relu(y)
    cls = F.conv2d(feats["cls_tower"], p["cls.weight"], p["cls.bias"], padding=1)
    cen = F.conv2d(feats["cls_tower"], p["center.weight"], p["center.bias"], padding=1)
    reg = F.relu(F.conv2d(feats["reg_tower"], p["reg.weight"], p["reg.bias"], padding=1))
    return torch.cat([cls, cen, reg], 1)


for n in [int(t) for t in os.environ.get("TRACKS", "30,100").split(",")]:
    torch.manual_seed(n)
    resp = torch.randn(n, 128, 16, 16, device=dev) * 15
    r64 = ref64(resp)
    outs = {}
    forms = [("fp32", dict(SMOT_TOWER_OCT=2, SMOT_TOWER_BF3=0)), ("fp16x2", dict(SMOT_TOWER_OCT=2, SMOT_TOWER_BF3=1)),
             ("fp32_one_tile", dict(SMOT_TOWER_OCT=1, SMOT_TOWER_BF3=0))]
    for a in [x for x in os.environ.get("ABLS", "").split(",") if x]:      # timing ablations of the bf16 x 3 form (wrong results)
        forms.append(("fp16x2_abl%s" % a, dict(SMOT_TOWER_OCT=2, SMOT_TOWER_BF3=1, SMOT_WINO_ABL=a)))
    for name, env in forms:
        with ops.debug_library(**env):
            f = lambda: ops.emm_predictor(resp, P)
            o = f()
            torch.cuda.synchronize()
            outs[name] = o
            for _ in range(100): f()
            torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                ops.kernel_timer_begin(ops.TIMER_TOWER, 300)
                for _ in range(300): f()
                ms, cnt = ops.kernel_timer_end(ops.TIMER_TOWER)
                ts.append(ms / cnt * 1e3)
            # phase trace (s_memtime ticks at 100 MHz): start, loop start, loop end, transform, heads, end
            grid = (n + 7) // 8 * 8 * 16 // int(env["SMOT_TOWER_OCT"])
            tr = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
            lib = ops.load_library()
            lib.smot_debug_trace(ops._ptr(tr)); f(); torch.cuda.synchronize(); tr.zero_(); f(); torch.cuda.synchronize()
            lib.smot_debug_trace(ops._ptr(None))
            t = tr.view(grid, 8).cpu().numpy(); t = t[t[:, 5] != 0]
            phases = [round(float(x), 1) for x in np.diff(t[:, :6], axis=1).mean(0)]
            if str(env.get("SMOT_WINO_ABL")) == "11":
                phases += [{"in_stage_barriers": round(float(t[:, 6].mean()), 1), "in_vmcnt_waits": round(float(t[:, 7].mean()), 1)}]
        e = (o.double() - r64).abs()
        scale = r64.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)
        print(json.dumps({"tracks": n, "form": name, "tower_us_min": round(min(ts), 2), "tower_us_median": round(sorted(ts)[2], 2),
                          "max_abs_err_vs_fp64": float(e.max()), "mean_abs_err_vs_fp64": float(e.mean()),
                          "max_err_over_channel_scale": float((e.amax(dim=(0, 2, 3)) / scale).max()),
                          "finite": bool(torch.isfinite(o).all()), "phase_ticks_mean": phases}), flush=True)
    d = (outs["fp16x2"] - outs["fp32"]).abs()
    print(json.dumps({"tracks": n, "fp16x2_vs_fp32_max_abs": float(d.max()), "equal_fraction": float((d == 0).float().mean())}), flush=True)
