"""200 template extractions (EMM.extract_cache's launch) for rocprofv3: python extract_run.py TRACKS HINT(0|1)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
n, hint = int(sys.argv[1]), bool(int(sys.argv[2]))
feats = [bench.synthetic_features(k, dev) for k in range(4)]
boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
for i in range(230):
    ops.emm_extract_cache(feats[i % 4], boxes, 15, (0.25, 0.125, 0.0625, 0.03125), 2, 512.0, 1.0, 0.0, hint=hint)
    if i % 8 == 7: torch.cuda.synchronize()
torch.cuda.synchronize()
