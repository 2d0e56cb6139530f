mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for F in "fp32 SMOT_TOWER_OCT=2 SMOT_TOWER_BF3=0" "bf3 SMOT_TOWER_OCT=2 SMOT_TOWER_BF3=1"; do
  set -- $F; TAG=$1; shift
  B="python $R/measure/debug/tower_run.py 30 $@"
  timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $R/gpurun_out/pmc_tw_${TAG}_tcc -o tcc -- $B > $R/gpurun_out/pmc_tw_${TAG}_tcc.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc_tw_${TAG}_sq -o sq -- $B > $R/gpurun_out/pmc_tw_${TAG}_sq.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmc_tw_${TAG}_fetch -o fetch -- $B > $R/gpurun_out/pmc_tw_${TAG}_fetch.log 2>&1
  cd $R
  python tools/rocpd_pmc.py gpurun_out/pmc_tw_${TAG}_tcc/tcc_results.db gpurun_out/pmc_tw_${TAG}_sq/sq_results.db gpurun_out/pmc_tw_${TAG}_fetch/fetch_results.db --md gpurun_out/r04_tower_${TAG}_pmc.md > gpurun_out/pmc_tw_${TAG}.log 2>&1
  grep -B2 -A22 "tower_wino" gpurun_out/r04_tower_${TAG}_pmc.md | head -40
  cd /tmp
done
rm -rf $R/gpurun_out/pmc_tw_*_tcc $R/gpurun_out/pmc_tw_*_sq $R/gpurun_out/pmc_tw_*_fetch
