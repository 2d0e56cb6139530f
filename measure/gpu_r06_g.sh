mkdir -p gpurun_out
export TMPDIR=/tmp
N=${1:-30}; shift
for K in "$@"; do
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_k -o k -- python $GRAFT_REPO_ROOT/measure/debug/pair_run.py $N ${K//,/ } > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_k/k_results.db --md gpurun_out/r06_knob.md --title "$K" > /dev/null 2>&1
echo "== $K"; head -9 gpurun_out/r06_knob.md | awk -F'|' 'NR>4{print substr($2,1,52), $3, $4, $5}'
rm -rf gpurun_out/prof_k
done
