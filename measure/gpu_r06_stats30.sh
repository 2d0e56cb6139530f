mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_q -o q -- python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --no-tracking-loop --no-other-configs > $R/gpurun_out/r06q_prof.log 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_q/q_results.db --md gpurun_out/r06q_kernel_stats.md --title "r06q" > /dev/null 2>&1
head -12 gpurun_out/r06q_kernel_stats.md | awk -F'|' 'NR>4{print substr($2,1,70), $3, $4, $5}'
rm -rf gpurun_out/prof_q
