set -x
mkdir -p gpurun_out
python -m pytest tests/test_solver.py tests/test_video_results.py -m gpu -q --no-header --tb=short 2>&1 | tail -30
python - <<'PY'
import sys, os, json, time
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
print(json.dumps(bench.tracking_loop_throughput(30, dev, feats)))
print(json.dumps(bench.tracking_loop_throughput(100, dev, feats)))
PY
timeout 600 python tools/argmax_stats.py --pairs 1000 --out gpurun_out/r02_argmax_stats 2>&1 | tail -2 | cut -c1-600
