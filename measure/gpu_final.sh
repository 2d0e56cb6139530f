# end-of-round session: GPU tests, the default bench line, kernel stats of the frame-pair loop and of the bench with the
# tracking loops (rows per launch grid), arg-max statistics at 30 tracks.   gpurun --timeout 1500 -- 'bash measure/gpu_final.sh r03'
TAG=${1:-r03}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -3 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-200
timeout 500 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json; cut -c1-400 gpurun_out/${TAG}_bench_line.json
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_driver_form.log 2>&1; tail -1 gpurun_out/${TAG}_bench_driver_form.log | cut -c1-300
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}a -o a -- python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --no-graph --extra-streams 0 > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}a/a_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: bench.py --steps 300 --extra-streams 0 (frame-pair loop)" > /dev/null 2>&1; head -9 gpurun_out/${TAG}_kernel_stats.md | cut -c1-170
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}b -o b -- python $R/bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-parity --no-graph > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}b/b_results.db --by-grid --md gpurun_out/${TAG}_loop_kernel_stats.md --title "${TAG}: bench.py --steps 100 incl. multi-stream and tracking loops (with refinement), rows per launch grid" > /dev/null 2>&1
grep -E "linear_rows|box_refine|fused9_kernel<7|track_solve" gpurun_out/${TAG}_loop_kernel_stats.md | cut -c1-200
rm -rf gpurun_out/prof_${TAG}a gpurun_out/prof_${TAG}b
timeout 600 python tools/argmax_stats.py --pairs ${PAIRS:-1000} --out gpurun_out/${TAG}_argmax_stats > gpurun_out/${TAG}_argmax.log 2>&1; tail -3 gpurun_out/${TAG}_argmax.log | cut -c1-300
