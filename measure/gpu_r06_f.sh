mkdir -p gpurun_out
export TMPDIR=/tmp
for H in 1 0; do
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_x$H -o x$H -- python $GRAFT_REPO_ROOT/measure/debug/extract_run.py 30 $H > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_x$H/x${H}_results.db --md gpurun_out/r06f_extract_hint$H.md --title "extract hint=$H" > /dev/null 2>&1
grep "fused9_kernel<15" gpurun_out/r06f_extract_hint$H.md | awk -F'|' '{print "hint='$H'", $3, $4, $5, $6}'
rm -rf gpurun_out/prof_x$H
done
