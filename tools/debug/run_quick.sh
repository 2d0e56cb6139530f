# scratch: one GPU-box session for the kernel under work (edit freely; tools/gpu_round.sh is the full round)
set -x
python -m pytest tests/test_hip_parity.py -m gpu -q --no-header --tb=short -x -k "decode or emm or bench" 2>&1 | tail -3
export TMPDIR=/tmp
TAG=r02dd
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --extra-streams 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}/${TAG}_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: bench.py --steps 300" 2>&1 | tail -1; head -9 gpurun_out/${TAG}_kernel_stats.md | cut -c1-60,150-260
