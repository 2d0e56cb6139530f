# second yaml family after the GroupNorm + heads kernel's halo-in-the-load-shadow change: its tests, trace, bench + kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_hip_parity.py tests/test_sequence.py -m gpu -q --no-header -x -k "blocked or aot or second_yaml or 29" 2>&1 | tail -3
timeout 120 python measure/debug/gn_heads_trace.py 30 > gpurun_out/r04_gn_heads_trace_b.jsonl 2>&1; grep '^{' gpurun_out/r04_gn_heads_trace_b.jsonl | cut -c1-400
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04d -o d -- python $R/tools/aot_bench.py --steps 300 > $R/gpurun_out/r04_aot_bench.log 2>&1 )
grep '^{' gpurun_out/r04_aot_bench.log | tail -1 > gpurun_out/r04_aot_bench.json; cut -c1-300 gpurun_out/r04_aot_bench.json
python tools/rocpd_stats.py gpurun_out/prof_r04d/d_results.db --md gpurun_out/r04_aot_kernel_stats.md --title "r04: tools/aot_bench.py --steps 300 — second yaml family (Rz=7, Rx=35, Ho=29), 30 tracks" > /dev/null 2>&1; head -10 gpurun_out/r04_aot_kernel_stats.md | cut -c1-170
rm -rf gpurun_out/prof_r04d
