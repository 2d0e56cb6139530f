"""Phase trace (s_memtime ticks) of the fused pooling+xcorr kernel and the template pooler, bench workload."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
lib = ops.load_library()
scales = (0.25, 0.125, 0.0625, 0.03125)
for n in [int(a) for a in sys.argv[1:]] or [30]:
    boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
    feats = bench.synthetic_features(1, dev)
    sr = ops.search_region(boxes, 512, 1.0, 0)
    z = ops.roi_align_levels(feats, boxes, boxes, 15, scales, 2)
    lv = ops.roi_align_levels(feats, boxes, boxes, 15, scales, 2, return_levels=True)[1].cpu().numpy()
    _, sr_h, hint = ops.emm_extract_cache(feats, boxes, 15, scales, 2, 512, 1.0, 0, hint=True)
    fl, fp, hs, ws_, sc = ops._level_arrays(feats, scales)
    pc = (ctypes.c_int * 4)(128, 64, 32, 16)
    resp_h = torch.empty((n, 128, 16, 16), device=dev)

    def hinted():
        with ops.debug_library() as dbg:
            dbg.smot_debug_trace(ops._ptr(tr) if tracing else None)
            rc = dbg.smot_debug_sr_xcorr_fused_hint_fwd(ops._cast(fp), ops._cast(hs), ops._cast(ws_), ops._cast(pc), ops._cast(sc), 4, 128,
                                                        ops._ptr(boxes), ops._ptr(sr_h), ops._ptr(z), n, ops._ptr(resp_h), ops._ptr(hint),
                                                        ops._stream(dev))
            dbg.smot_debug_trace(None)
            assert rc == 0
    tracing = False
    runs = {"fused<30,15,true>": lambda: ops.sr_xcorr_fused(feats, boxes, sr, z, 30, 15, scales, 2, 512),
            "fused<30,15,true>+hint": hinted,
            "pool<15>": lambda: ops.roi_align_levels(feats, boxes, boxes, 15, scales, 2),
            "pool<30>": lambda: ops.roi_align_levels(feats, sr, boxes, 30, scales, 2, [128, 64, 32, 16])}
    for name, f in runs.items():
        for _ in range(20): f()
        torch.cuda.synchronize()
        grid = n * 16
        tr = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
        tracing = True
        lib.smot_debug_trace(ops._ptr(tr)); f(); torch.cuda.synchronize(); lib.smot_debug_trace(ops._ptr(None))
        tracing = False
        t = tr.view(n, 16, 8).cpu().numpy().astype(np.float64)
        ran = t[:, :, 4] > 0                                   # workgroups that did not return at the split
        d = np.diff(t[:, :, :5], axis=2)
        t0 = t[:, :, 0].min()
        out = {"tracks": n, "kernel": name, "workgroups_working": int(ran.sum()), "workgroups_launched": int(ran.size),
               "phase_ticks_mean(tables,z,pool,xcorr)": [round(float(x)) for x in d[ran].mean(0)],
               "phase_max": [int(x) for x in d[ran].max(0)], "span": int(t[:, :, 4].max() - t0),
               "start_spread": int(t[:, :, 0].max() - t0), "total_mean": round(float((t[:, :, 4] - t[:, :, 0])[ran].mean())),
               "total_max": int((t[:, :, 4] - t[:, :, 0])[ran].max()),
               "assignment_ticks_mean": round(float((t[:, :, 5] - t[:, :, 0])[ran].mean()))}
        for L in range(4):
            m = (lv == L)[:, None] & ran
            if m.any(): out["pool_ticks_level%d" % L] = round(float(d[:, :, 2][m].mean()))
        print(json.dumps(out), flush=True)
