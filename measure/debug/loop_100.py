import json, sys, torch
sys.path.insert(0, '/root/repo')
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
for refine in (False, True):
    for ahead in (False, True):
        ops.FALLBACKS.clear()
        r = bench.tracking_loop_throughput(100, dev, feats, steps=400, refine=refine, ahead=ahead)
        print(json.dumps({"tracks": 100, "refine": refine, "next_frame_shown": ahead, "ms_per_frame": round(r["ms_per_frame"], 5), "held": r["track_count_held"],
                          "tracked": r["tracked_in_last_frame"], "native_frames": r["frame_entry_point_frames"], "spec": r["speculative_heads"], "fallbacks": dict(ops.FALLBACKS)}), flush=True)
