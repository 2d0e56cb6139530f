"""Phase trace (s_memtime ticks) of the pooling + correlation kernel with an order hint, per measurement-library switch:
    python measure/debug/fused_trace_ab.py 30 100 -- SMOT_FUSED_ABL=0 SMOT_FUSED_ABL=8
(ABL 8 = the fp32 FMA correlation of rounds 2-5, 0 = the matrix-pipe correlation)."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
ops.load_library()
scales = (0.25, 0.125, 0.0625, 0.03125)
args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
counts = [int(a) for a in args[:split]] or [30]
variants = [dict(kv.split("=") for kv in v.split(",")) for v in args[split + 1:]] or [{}]
for n in counts:
    boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
    feats = bench.synthetic_features(1, dev)
    z = ops.roi_align_levels(feats, boxes, boxes, 15, scales, 2)
    _, sr_h, hint = ops.emm_extract_cache(feats, boxes, 15, scales, 2, 512, 1.0, 0, hint=True)
    fl, fp, hs, ws_, sc = ops._level_arrays(feats, scales)
    pc = (ctypes.c_int * 4)(128, 64, 32, 16)
    resp_h = torch.empty((n, 128, 16, 16), device=dev)
    for var in variants:
        with ops.debug_library(**var) as dbg:
            def run(tr=None):
                dbg.smot_debug_trace(ops._ptr(tr) if tr is not None else None)
                rc = dbg.smot_debug_sr_xcorr_fused_hint_fwd(ops._cast(fp), ops._cast(hs), ops._cast(ws_), ops._cast(pc), ops._cast(sc), 4, 128,
                                                            ops._ptr(boxes), ops._ptr(sr_h), ops._ptr(z), n, ops._ptr(resp_h), ops._ptr(hint),
                                                            ops._stream(dev))
                dbg.smot_debug_trace(None)
                assert rc == 0
            for _ in range(20): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 200 * 1e3
            tr = torch.zeros(n * 16 * 8, dtype=torch.int64, device=dev)
            run(tr); torch.cuda.synchronize()
        t = tr.view(n, 16, 8).cpu().numpy().astype(np.float64)
        ran = (t[:, :, 4] > 0) & (t[:, :, 0] > 0)
        d = np.diff(t[:, :, :5], axis=2)
        t0 = t[:, :, 0][ran].min()
        end = t[:, :, 4][ran]
        starts = np.sort(t[:, :, 0][ran] - t0)
        ends = np.sort(end - t0)
        print(json.dumps({"tracks": n, "variant": var, "same_feature_set_us": round(us, 2),
                          "phase_ticks_mean(tables,z,pool,xcorr)": [round(float(x)) for x in d[ran].mean(0)],
                          "phase_p90": [round(float(x)) for x in np.percentile(d[ran], 90, axis=0)],
                          "phase_max": [int(x) for x in d[ran].max(0)], "span": int(end.max() - t0), "workgroups_traced": int(ran.sum()),
                          "start_percentiles(50,90,100)": [int(np.percentile(starts, q)) for q in (50, 90, 100)],
                          "end_percentiles(50,90,99,100)": [int(np.percentile(ends, q)) for q in (50, 90, 99, 100)], "slowest_workgroups(roi,cgrp,total,tables,z,pool,xcorr)": [
                              [int(i // 16), int(i % 16), int((t[:, :, 4] - t[:, :, 0]).reshape(-1)[i])] + [int(v) for v in d.reshape(-1, 4)[i]]
                              for i in np.argsort(-(np.where(ran, t[:, :, 4] - t[:, :, 0], 0)).reshape(-1))[:10]],
                          "head_ticks_mean(start->assigned,->level+bins,->wave0 at table barrier,->tables)": [
                              round(float((t[:, :, 5] - t[:, :, 0])[ran].mean())), round(float((t[:, :, 6] - t[:, :, 5])[ran].mean())),
                              round(float((t[:, :, 7] - t[:, :, 6])[ran].mean())), round(float((t[:, :, 1] - t[:, :, 7])[ran].mean()))],
                          "total_by_roi_mean": [round(float(v)) for v in np.where(ran, t[:, :, 4] - t[:, :, 0], np.nan).mean(1)],
                          "total_mean": round(float((t[:, :, 4] - t[:, :, 0])[ran].mean())),
                          "total_max": int((t[:, :, 4] - t[:, :, 0])[ran].max())}), flush=True)
