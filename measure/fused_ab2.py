"""A/B of the fused pooling + correlation kernel under measurement-library switches, same session, interleaved, timed by
the DISPATCH'S OWN begin/end timestamps (the library's kernel timer: hipExtLaunchKernel events, what rocprofv3 reports) —
`fused_ab.py` times 200 back-to-back Python calls with torch events, which at 30 tracks is bound by the host's enqueue
rate (≈17 µs per call), not by the kernel (round 4: the no-correlation ablation read 17.7 µs there).
    python measure/fused_ab2.py 30 100 -- SMOT_FUSED_ABL=0 SMOT_FUSED_ABL=2 hint=1,SMOT_FUSED_GEN=0
`hint=1` runs the launch with the order hint / roi plans written by the extraction (the benchmarked configuration).
Prints the median and the minimum over 5 repeats of the mean of 100 timed launches over 4 rotating feature sets."""
import ctypes, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
ops.load_library()
scales = (0.25, 0.125, 0.0625, 0.03125)
args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
counts = [int(a) for a in args[:split]] or [30]
variants = [dict(kv.split("=") for kv in v.split(",")) for v in args[split + 1:]] or [{}]
feats = [bench.synthetic_features(k, dev) for k in range(4)]
pc = (ctypes.c_int * 4)(128, 64, 32, 16)
for n in counts:
    boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
    sr = ops.search_region(boxes, 512, 1.0, 0)
    z = ops.roi_align_levels(feats[0], boxes, boxes, 15, scales, 2)
    ref = None
    for rep in range(2):
        for var in variants:
            var = dict(var)
            hinted = var.pop("hint", "0") != "0"
            with ops.debug_library(**var) as dbg:
                la = [ops._level_arrays(f, scales) for f in feats]
                resp = torch.empty((n, 128, 16, 16), device=dev)
                if hinted:
                    _, sr_h, hint = ops.emm_extract_cache(feats[0], boxes, 15, scales, 2, 512, 1.0, 0, hint=True)

                    def f(k):
                        fl, fp, hs, ws_, sc = la[k % 4]
                        rc = dbg.smot_debug_sr_xcorr_fused_hint_fwd(ops._cast(fp), ops._cast(hs), ops._cast(ws_), ops._cast(pc),
                                                                    ops._cast(sc), 4, 128, ops._ptr(boxes), ops._ptr(sr_h), ops._ptr(z), n,
                                                                    ops._ptr(resp), ops._ptr(hint), ops._stream(dev))
                        assert rc == 0, dbg.smot_last_error()
                        return resp
                else:
                    f = lambda k: ops.sr_xcorr_fused(feats[k % 4], boxes, sr, z, 30, 15, scales, 2, 512)
                out = f(0)
                torch.cuda.synchronize()
                ref = out.clone() if ref is None else ref
                same = bool(torch.equal(out, ref))
                for k in range(50): f(k)
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    ops.kernel_timer_begin(ops.TIMER_XCORR, 100, 1)
                    for k in range(100): f(k)
                    torch.cuda.synchronize()
                    ms, cnt = ops.kernel_timer_end(ops.TIMER_XCORR)
                    ts.append(ms / max(cnt, 1) * 1e3)
            print(json.dumps({"tracks": n, "variant": var, "hint": hinted, "fused_us_median": round(statistics.median(ts), 2),
                              "fused_us_min": round(min(ts), 2), "bitwise_equal": same}), flush=True)
