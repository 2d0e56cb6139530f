"""The tracking loop with dormant tracks in the memory (bench.tracking_loop_throughput(..., dormant=)): device copy of the
dormant rows / the reference's host form / with the next frame shown.  JSON lines; run under rocprofv3 --kernel-trace for
the kernels of the frame.  usage: [tracks] [dormant] [modes: d,h,a]"""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
import siammot_amd.ops as ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 6
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["d", "h", "a", "0", "0a"]
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
for m in modes:
    ops.FALLBACKS.clear(); ops.MEMORY_CARRY.clear()
    kw = {"d": dict(dormant=nd), "h": dict(dormant=nd, device_carry=False), "a": dict(dormant=nd, ahead=True),
          "0": dict(), "0a": dict(ahead=True)}[m]
    r = bench.tracking_loop_throughput(n, dev, feats, steps=400, **kw)
    print(json.dumps({"tracks": n, "mode": m, "ms_per_frame": round(r["ms_per_frame"], 5), "active": r["active_tracks"],
                      "dormant": r["dormant_tracks"], "memory_rows": r["memory_rows"], "held": r["track_count_held"],
                      "spec": r["speculative_heads"], "carry": dict(ops.MEMORY_CARRY), "fallbacks": dict(ops.FALLBACKS)}), flush=True)
