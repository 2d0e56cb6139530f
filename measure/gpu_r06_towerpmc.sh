# PMC counters of the tower kernel alone (separate --pmc passes): gpurun_out/${TAG}_tower_pmc.md
mkdir -p gpurun_out
TAG=${1:-r06}
N=${2:-30}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/measure/debug/tower_run.py $N SMOT_TOWER_OCT=2"
cd /tmp
timeout 90 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $R/gpurun_out/pmc_${TAG}_tcc -o tcc -- $B > $R/gpurun_out/pmc_${TAG}_tcc.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_${TAG}_sq -o sq -- $B > $R/gpurun_out/pmc_${TAG}_sq.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc_${TAG}_valu -o valu -- $B > $R/gpurun_out/pmc_${TAG}_valu.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${TAG}_fetch -o fetch -- $B > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
cd $R
python tools/rocpd_pmc.py $(ls gpurun_out/pmc_${TAG}_*/*_results.db) --md gpurun_out/${TAG}_tower_pmc.md > /dev/null 2>&1
grep -A26 "tower_wino" gpurun_out/${TAG}_tower_pmc.md | head -40
tail -3 gpurun_out/pmc_${TAG}_tcc.log
rm -rf gpurun_out/pmc_${TAG}_sq gpurun_out/pmc_${TAG}_valu gpurun_out/pmc_${TAG}_tcc gpurun_out/pmc_${TAG}_fetch
