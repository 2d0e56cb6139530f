# round-4 end-of-round session, second edition (after the device-carried track memory): the artifacts that changed —
# GPU tests, the bench line (default + driver form), kernel statistics of the frame-pair loop and of the loops, the
# dormant-track loops at 30 and 100 rows.  The counters / traffic stamp, the C = 256 and second-yaml-family profiles and
# the arg-max statistics of measure/gpu_r04_final.sh stand (their kernels' sources are unchanged).
#   gpurun --timeout 900 -- 'bash measure/gpu_r04_final2.sh'
TAG=r04
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -3 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-200
timeout 500 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json; cut -c1-300 gpurun_out/${TAG}_bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_line.json"))
t = d["tracking_loop"]; w = t["with_dormant_tracks"]
print("loop", t["ms_per_frame"], t["with_refinement"]["ms_per_frame"], "shown", t["next_frame_shown"]["ms_per_frame"], t["next_frame_shown"]["with_refinement"]["ms_per_frame"])
print("dormant", w["ms_per_frame"], "host", w["host_form"]["ms_per_frame"], "shown", w["next_frame_shown"]["ms_per_frame"], w["active_tracks"], w["dormant_tracks"])
print("fused", d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["traffic"], "tower", d["roofline_tower"]["avg_launch_us"])
for k, v in (d.get("other_configs") or {}).items():
    print(k, v.get("ms_per_step"), v.get("error"))
print("parity", json.dumps(d.get("parity"))[:400])
PY
bash measure/gpu_r04_bench.sh ${TAG} | cut -c1-300
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}a -o a -- python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --no-graph --extra-streams 0 --no-other-configs > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}a/a_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: bench.py --steps 300 --extra-streams 0 (frame-pair loop)" > /dev/null 2>&1; head -9 gpurun_out/${TAG}_kernel_stats.md | cut -c1-170
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}b -o b -- python $R/bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-parity --no-graph --no-other-configs > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}b/b_results.db --by-grid --md gpurun_out/${TAG}_loop_kernel_stats.md --title "${TAG}: bench.py --steps 100 incl. multi-stream and tracking loops (with refinement, with dormant tracks), rows per launch grid" > /dev/null 2>&1
rm -rf gpurun_out/prof_${TAG}a gpurun_out/prof_${TAG}b
timeout 200 python measure/debug/loop_dormant.py 100 20 d,h,a,0,0a > gpurun_out/${TAG}_loop_dormant_n100.jsonl 2>&1; grep '^{' gpurun_out/${TAG}_loop_dormant_n100.jsonl | cut -c1-300
