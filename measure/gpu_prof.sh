# rocprofv3 kernel-trace of the bench: (a) the frame-pair loop alone, (b) the bench with the tracking loops (refinement on)
# usage: gpurun --timeout 900 -- 'bash measure/gpu_prof.sh TAG'
TAG=${1:-prof}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}a -o ${TAG}a -- python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --extra-streams 0 > $R/gpurun_out/${TAG}a_prof_bench.log 2>&1 )
grep '"metric"' gpurun_out/${TAG}a_prof_bench.log | cut -c1-200
python tools/rocpd_stats.py gpurun_out/prof_${TAG}a/${TAG}a_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: bench.py --steps 300 --extra-streams 0 (frame-pair loop)" 2>&1 | tail -2
head -14 gpurun_out/${TAG}_kernel_stats.md | cut -c1-220
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}b -o ${TAG}b -- python $R/bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-parity > $R/gpurun_out/${TAG}b_prof_bench.log 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}b/${TAG}b_results.db --md gpurun_out/${TAG}_loop_kernel_stats.md --title "${TAG}: bench.py --steps 100 incl. multi-stream and tracking loops (with refinement)" 2>&1 | tail -2
head -40 gpurun_out/${TAG}_loop_kernel_stats.md | cut -c1-200
grep -o '"tracking_loop".*' gpurun_out/${TAG}b_prof_bench.log | cut -c1-1200
rm -rf gpurun_out/prof_${TAG}a gpurun_out/prof_${TAG}b
