"""cProfile of the one-launch tracking loop (bench.tracking_loop_throughput): where the host time of a frame goes."""
import cProfile, os, pstats, sys, io
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
refine = len(sys.argv) > 1 and sys.argv[1] == "refine"
bench.tracking_loop_throughput(30, dev, feats, steps=50, refine=refine)        # warm
pr = cProfile.Profile()
pr.enable()
out = bench.tracking_loop_throughput(30, dev, feats, steps=600, refine=refine)
pr.disable()
print({k: out[k] for k in ("ms_per_frame", "track_count_held")})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().split("\n")[:60]))
