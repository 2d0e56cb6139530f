# round 4: dormant rows carried on the device, copy enqueued before the record is read — targeted tests, loop timings, kernels
#   gpurun --timeout 600 -- 'bash measure/gpu_r04_carry2.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_sequence.py tests/test_solver.py -m gpu -q --no-header -rf --tb=short -x > gpurun_out/r04_pytest_carry.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_pytest_carry.log; tail -4 gpurun_out/r04_pytest_carry.log | cut -c1-300
( AHEAD=1 PEEK=5 MAXD=30 timeout 120 python measure/debug/loop_equiv_soak.py 200; echo "exit $?"
  SWITCH=1 MAXD=8 timeout 120 python measure/debug/loop_equiv_soak.py 200; echo "exit $?" ) > gpurun_out/r04_carry_soak.log 2>&1
grep -v "^$" gpurun_out/r04_carry_soak.log | grep -v amdgpu.ids | tail -8 | cut -c1-400
timeout 200 python measure/debug/loop_dormant.py 30 6 > gpurun_out/r04_loop_dormant.jsonl 2>&1; cat gpurun_out/r04_loop_dormant.jsonl | grep '^{' | cut -c1-400
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_carry -o c -- python $R/measure/debug/loop_dormant.py 30 6 a > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_carry/c_results.db --md gpurun_out/r04_loop_dormant_kernel_stats.md --title "r04: tracking loop, 24 active + 6 dormant tracks, next frame shown (measure/debug/loop_dormant.py 30 6 a)" > /dev/null 2>&1; head -16 gpurun_out/r04_loop_dormant_kernel_stats.md | cut -c1-200
rm -rf gpurun_out/prof_carry
