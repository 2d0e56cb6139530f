mkdir -p gpurun_out
TAG=${1:-r06e}
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short -x -k "decode or benchmark or frame_pair or golden or emm_case or sequence_plain or near_ties or many_tracks" > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --extra-streams 0 --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}/${TAG}_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: bench.py --steps 300" 2>&1 | tail -1; head -9 gpurun_out/${TAG}_kernel_stats.md | cut -c1-140
rm -rf gpurun_out/prof_${TAG}
