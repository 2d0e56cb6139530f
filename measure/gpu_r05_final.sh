# round-5 end-of-round session (all artifacts the bench line and DESIGN.md point at):
#   gpurun --timeout 2400 -- 'bash measure/gpu_r05_final.sh'
TAG=r05
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -3 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-200
# PMC counters first: the traffic figure the bench line quotes is stamped with the kernel's source hash
bash measure/gpu_pmc.sh ${TAG}_n30 30 > /dev/null 2>&1; mv gpurun_out/${TAG}_n30_pmc_counters.md gpurun_out/${TAG}_pmc_counters_n30.md
bash measure/gpu_pmc.sh ${TAG}_n100 100 > /dev/null 2>&1; mv gpurun_out/${TAG}_n100_pmc_counters.md gpurun_out/${TAG}_pmc_counters_n100.md
cp gpurun_out/${TAG}_pmc_counters_n30.md gpurun_out/${TAG}_pmc_counters_n100.md profiles/
python tools/make_traffic_json.py profiles/${TAG}_pmc_counters_n30.md profiles/${TAG}_pmc_counters_n100.md | cut -c1-200
cp profiles/xcorr_traffic.json gpurun_out/xcorr_traffic.json
timeout 700 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json; cut -c1-300 gpurun_out/${TAG}_bench_line.json
bash measure/gpu_r04_bench.sh ${TAG}
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}a -o a -- python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --no-graph --extra-streams 0 --no-other-configs > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}a/a_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: bench.py --steps 300 --extra-streams 0 (frame-pair loop)" > /dev/null 2>&1; head -9 gpurun_out/${TAG}_kernel_stats.md | cut -c1-170
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}b -o b -- python $R/bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-parity --no-graph --no-other-configs > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}b/b_results.db --by-grid --md gpurun_out/${TAG}_loop_kernel_stats.md --title "${TAG}: bench.py --steps 100 incl. multi-stream and tracking loops (with refinement), rows per launch grid" > /dev/null 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}c -o c -- python $R/bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-parity --no-graph --extra-streams 0 --no-other-configs --other-config-worker --channels 256 --net-hw 1056 1920 --tracks 50 > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}c/c_results.db --md gpurun_out/${TAG}_cfg4_kernel_stats.md --title "${TAG}: BASELINE.json configs[4] (C=256, 1056x1920, 50 tracks): bench.py frame-pair loop" > /dev/null 2>&1; head -8 gpurun_out/${TAG}_cfg4_kernel_stats.md | cut -c1-170
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}d -o d -- python $R/tools/aot_bench.py --steps 300 > $R/gpurun_out/${TAG}_aot_bench.log 2>&1 )
grep '^{' gpurun_out/${TAG}_aot_bench.log | tail -1 > gpurun_out/${TAG}_aot_bench.json; cut -c1-300 gpurun_out/${TAG}_aot_bench.json
python tools/rocpd_stats.py gpurun_out/prof_${TAG}d/d_results.db --md gpurun_out/${TAG}_aot_kernel_stats.md --title "${TAG}: tools/aot_bench.py --steps 300 — second yaml family (Rz=7, Rx=35, Ho=29), 30 tracks" > /dev/null 2>&1; head -10 gpurun_out/${TAG}_aot_kernel_stats.md | cut -c1-170
rm -rf gpurun_out/prof_${TAG}a gpurun_out/prof_${TAG}b gpurun_out/prof_${TAG}c gpurun_out/prof_${TAG}d
[ -n "$SKIP_ARGMAX" ] && exit 0
timeout 600 python tools/argmax_stats.py --pairs ${PAIRS:-1000} --tracks 30 --out gpurun_out/${TAG}_argmax_stats > gpurun_out/${TAG}_argmax.log 2>&1; tail -1 gpurun_out/${TAG}_argmax.log | cut -c1-400
timeout 600 python tools/argmax_stats.py --pairs 100 --tracks 100 --out gpurun_out/${TAG}_argmax_stats_n100 > gpurun_out/${TAG}_argmax_n100.log 2>&1; tail -1 gpurun_out/${TAG}_argmax_n100.log | cut -c1-400
timeout 300 python measure/loop_early_ab.py 30 > gpurun_out/${TAG}_loop_early_ab.jsonl 2>&1; grep '^{' gpurun_out/${TAG}_loop_early_ab.jsonl | cut -c1-200
