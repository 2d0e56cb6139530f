"""The tracking loop under rocprofv3: per-kernel durations of one frame (head + solver + masked template extraction).
    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -o loop -- python measure/debug/loop_kernels.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
print(bench.tracking_loop_throughput(int(sys.argv[1]) if len(sys.argv) > 1 else 30, dev, feats, steps=500))
