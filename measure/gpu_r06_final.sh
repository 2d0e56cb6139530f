# round-6 evidence session: full GPU suite, default bench line, driver-form line, rocprofv3 kernel stats of the bench (30 tracks,
# 100 tracks, C = 256 / 1080p), AOT family stats, tracking-loop kernel stats
mkdir -p gpurun_out
TAG=${1:-r06}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
grep -n "passed\|failed\|^FAILED" gpurun_out/${TAG}_pytest_gpu.log | tail -5
timeout 500 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; grep '^{' gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench_line.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_form.log 2>&1; grep '^{' gpurun_out/${TAG}_bench_driver_form.log | tail -1 > gpurun_out/${TAG}_bench_line_driver_form.json
timeout 300 python bench.py --tracks 100 --no-other-configs > gpurun_out/${TAG}_bench_n100.log 2>&1; grep '^{' gpurun_out/${TAG}_bench_n100.log | tail -1 > gpurun_out/${TAG}_bench_line_n100.json
python - <<PY
import json
for f in ("bench_line", "bench_line_driver_form", "bench_line_n100"):
    try:
        d = json.load(open("gpurun_out/${TAG}_%s.json" % f))
        print(f, json.dumps(d.get("summary"))[:900])
    except Exception as e:
        print(f, "unreadable", e)
PY
stats() {   # name, command...
  local name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -o $name -- "$@" > $R/gpurun_out/${TAG}_${name}_prof.log 2>&1 )
  python tools/rocpd_stats.py gpurun_out/prof_$name/${name}_results.db --md gpurun_out/${TAG}_${name}_kernel_stats.md --title "${TAG} $name: $*" > /dev/null 2>&1
  echo "== $name"; head -10 gpurun_out/${TAG}_${name}_kernel_stats.md | awk -F'|' 'NR>4{print substr($2,1,58), $3, $4, $5}'
  rm -rf gpurun_out/prof_$name
}
stats n30 python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --no-tracking-loop --no-other-configs
stats n100 python $R/bench.py --tracks 100 --steps 200 --warmup 30 --no-cpu-baseline --no-parity --no-tracking-loop --no-other-configs
stats cfg4 python $R/bench.py --tracks 50 --channels 256 --net-hw 1056 1920 --steps 150 --warmup 20 --no-cpu-baseline --no-parity --no-tracking-loop --no-other-configs
stats aot python $R/tools/aot_bench.py
stats loop python $R/measure/debug/loop_kernels.py
