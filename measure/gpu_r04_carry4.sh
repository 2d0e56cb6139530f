# round 4: kernels of the dormant-track loop (next frame shown) after the small fields' preload in the solver launch
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 80 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_carry -o c -- python $R/measure/debug/loop_dormant.py 30 6 a > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_carry/c_results.db --md gpurun_out/r04_loop_dormant_kernel_stats.md --title "r04: tracking loop, 24 active + 6 dormant tracks, next frame shown (measure/debug/loop_dormant.py 30 6 a)" > /dev/null 2>&1; head -10 gpurun_out/r04_loop_dormant_kernel_stats.md | cut -c1-150
rm -rf gpurun_out/prof_carry
