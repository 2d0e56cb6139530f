# scratch: one GPU-box session for the kernel under work (edit freely; tools/gpu_round.sh is the full round)
set -x
python - <<'PY'
import sys, json, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import siammot_amd.ops as ops, bench
import golden_inputs as gi
dev = 'cuda'
for n in (16, 30, 40):
    rs = np.random.RandomState(1)
    boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
    sr = ops.search_region(boxes, 512, 1.0, 0)
    logits = torch.randn(n, 7, 16, 16, device=dev) * 2
    for split in (1, 2, 4):
        with ops.debug_library(SMOT_DECODE_SPLIT=split):
            f = lambda: ops.emm_decode(logits, sr, boxes, 30, 15, 512, clip_wh=(1280, 704))
            for _ in range(50): f()
            torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200): f()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 200 * 1e3)
        print(json.dumps({"tracks": n, "decode_split": split, "us_per_call_back_to_back": round(min(ts), 2)}), flush=True)
PY
