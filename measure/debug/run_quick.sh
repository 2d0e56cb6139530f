# scratch: one GPU-box session for the kernel under work (edit freely; measure/gpu_round.sh is the full round)
set -x
python - <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
import siammot_amd.ops as ops, bench
dev = torch.device('cuda:0')
scales = (0.25, 0.125, 0.0625, 0.03125)
feats = [bench.synthetic_features(k, dev) for k in range(4)]
for n in (30, 16):
    boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
    sr = ops.search_region(boxes, 512, 1.0, 0)
    z = ops.roi_align_levels(feats[0], boxes, boxes, 15, scales, 2)
    ref = ops.sr_xcorr_fused(feats[0], boxes, sr, z, 30, 15, scales, 2, 512)
    for gen in (0, 3):
        with ops.debug_library(SMOT_FUSED_GEN=gen):
            f = lambda i: ops.sr_xcorr_fused(feats[i % 4], boxes, sr, z, 30, 15, scales, 2, 512)
            same = bool(torch.equal(f(0), ref))
            for i in range(50): f(i)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(5):
                ops.kernel_timer_begin(ops.TIMER_XCORR, 300)
                for i in range(300): f(i)
                ms, cnt = ops.kernel_timer_end(ops.TIMER_XCORR)
                best = min(best, ms / cnt * 1e3)
        print(json.dumps({"tracks": n, "variant": "pipelined wide windows" if gen == 3 else "default", "fused_kernel_us": round(best, 2), "bitwise_equal": same}), flush=True)
PY
