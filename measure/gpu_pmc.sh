# PMC counters of the bench's kernels, separate --pmc passes (no --stats / trace domains beside them).
# usage: gpurun --timeout 900 -- 'bash measure/gpu_pmc.sh TAG [TRACKS]'   -> gpurun_out/${TAG}_pmc_counters.md
mkdir -p gpurun_out
TAG=${1:-r03}
N=${2:-30}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --tracks $N --steps 40 --warmup 5 --prewarm-ms 0 --no-cpu-baseline --no-parity --no-kernel-timer --no-tracking-loop --extra-streams 0"
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_${TAG}_sq -o sq -- $B > $R/gpurun_out/pmc_${TAG}_sq.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${TAG}_fetch -o fetch -- $B > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $R/gpurun_out/pmc_${TAG}_write -o write -- $B > $R/gpurun_out/pmc_${TAG}_write.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d $R/gpurun_out/pmc_${TAG}_tcc -o tcc -- $B > $R/gpurun_out/pmc_${TAG}_tcc.log 2>&1
cd $R
python tools/rocpd_pmc.py gpurun_out/pmc_${TAG}_sq/sq_results.db gpurun_out/pmc_${TAG}_fetch/fetch_results.db gpurun_out/pmc_${TAG}_write/write_results.db gpurun_out/pmc_${TAG}_tcc/tcc_results.db --md gpurun_out/${TAG}_pmc_counters.md > /dev/null 2>&1
grep -A24 "fused9_kernel<30" gpurun_out/${TAG}_pmc_counters.md | head -26
rm -rf gpurun_out/pmc_${TAG}_sq gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_tcc
