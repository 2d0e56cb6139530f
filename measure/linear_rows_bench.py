"""Box-head linear layers at M = 30 rows: csrc/linear_rows.hip against the library GEMM (torch / hipBLASLt), same session."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import siammot_amd.ops as ops
ops.load_library()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for K, N, relu in ((6272, 1024, True), (1024, 1024, True), (1024, 10, False)):
    x = torch.randn((M, K), device="cuda"); w = torch.randn((N, K), device="cuda") / K ** 0.5; b = torch.randn((N,), device="cuda")
    fns = {"linear_rows": lambda: ops.linear_rows(x, w, b, relu=relu),
           "library": lambda: (torch._addmm_activation(b, x, w.t()) if relu else torch.addmm(b, x, w.t()))}
    for name, f in fns.items():
        for _ in range(20): f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        print(json.dumps({"rows": M, "K": K, "N": N, "impl": name, "us_per_call_back_to_back": round(min(ts), 2)}), flush=True)
