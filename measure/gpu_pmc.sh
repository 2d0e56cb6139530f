set -x
mkdir -p gpurun_out
TAG=${1:-r01}
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --prewarm-ms 0 --no-cpu-baseline --no-kernel-timer --extra-streams 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_sq -o sq -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_fetch -o fetch -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_write -o write -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_tcc -o tcc -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_tcc.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/pmc_${TAG}_*/ | head -20
tail -2 gpurun_out/pmc_${TAG}_sq.log | cut -c1-200
