# scratch: one GPU-box session for the kernel under work (edit freely; gpu_round.sh is the full round)
set -x
TAG=${1:-q}
python -m pytest tests/test_hip_parity.py -m gpu -q --no-header --tb=short -x -k "predictor or aot or emm" 2>&1 | tail -5
python tools/aot_bench.py 2>&1 | tail -2 | cut -c1-400
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/tools/aot_bench.py --steps 300 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}/${TAG}_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: tools/aot_bench.py --steps 300 (AOT shape family)" 2>&1 | tail -1; head -14 gpurun_out/${TAG}_kernel_stats.md | cut -c1-100,150-230
