# scratch: one GPU-box session for the kernel under work (edit freely; tools/gpu_round.sh is the full round)
set -x
python -m pytest tests/test_hip_parity.py -m gpu -q --no-header --tb=short -x -k "xcorr or aot or predictor" 2>&1 | tail -3
python - <<'PY'
import torch, json, sys
sys.path.insert(0, '.')
import siammot_amd.ops as ops
ops.load_library()
for n, rx, rz in ((30, 30, 15), (100, 30, 15), (30, 35, 7), (100, 35, 7)):
    x = torch.randn(n, 128, rx, rx, device='cuda'); z = torch.randn(n, 128, rz, rz, device='cuda')
    for _ in range(50): ops.xcorr_depthwise(x, z)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        ops.kernel_timer_begin(ops.TIMER_XCORR, 300)
        for _ in range(300): ops.xcorr_depthwise(x, z)
        ms, cnt = ops.kernel_timer_end(ops.TIMER_XCORR)
        best = min(best, ms / cnt * 1e3)
    print(json.dumps({"op": "xcorr_depthwise", "tracks": n, "rx": rx, "rz": rz, "kernel_us": round(best, 2)}), flush=True)
PY
python tools/aot_bench.py 2>&1 | tail -1 | cut -c1-300
