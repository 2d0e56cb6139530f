# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/ubench/fetch_calib.hip); two separate --pmc passes.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $R/gpurun_out/calib_fetch -o fetch -- $R/tools/ubench/fetch_calib > $R/gpurun_out/calib_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $R/gpurun_out/calib_write -o write -- $R/tools/ubench/fetch_calib > $R/gpurun_out/calib_write.log 2>&1
cd $R
grep read16_bytes gpurun_out/calib_fetch.log
python tools/rocpd_pmc.py gpurun_out/calib_fetch/fetch_results.db gpurun_out/calib_write/write_results.db --skip 0 --filter "" --md gpurun_out/calib_counters.md | grep -v "^$" | head -60
