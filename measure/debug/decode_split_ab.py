"""Decode kernel alone (random logits) per thread-group count, interleaved:  python measure/debug/decode_split_ab.py 30 45 50 64 100"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import siammot_amd.ops as ops
dev = "cuda:0"
ops.load_library()
for n in [int(a) for a in sys.argv[1:]] or [30, 100]:
    g = torch.Generator().manual_seed(n)
    logits = torch.randn((n, 7, 16, 16), generator=g).to(dev)
    logits[:, 3:] = logits[:, 3:].abs() * 20
    wh = torch.rand((n, 2), generator=g) * 200 + 30
    xy = torch.rand((n, 2), generator=g) * torch.tensor([1000.0, 500.0])
    boxes = torch.cat([xy, xy + wh], 1).to(dev)
    sr = ops.search_region(boxes, 512, 1.0, 0)
    ref = None
    for rep in range(2):
        for split in (1, 2):
            with ops.debug_library(SMOT_DECODE_SPLIT=str(split)):
                f = lambda: ops.emm_decode(logits, sr, boxes, 30, 15, 512, return_index=True, clip_wh=(1280, 704))
                out = f()
                ref = out if ref is None else ref
                same = all(bool(torch.equal(a, b)) for a, b in zip(out, ref))
                for _ in range(30): f()
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(200): f()
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 200 * 1e3)
            print(json.dumps({"tracks": n, "thread_groups_per_band": split, "decode_call_us": round(min(ts), 2), "bitwise_equal": same}), flush=True)
