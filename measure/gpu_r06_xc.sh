# round 6: the matrix-pipe correlation prototype, aligned-read form (tools/ubench/xcorr_f16x2.hip)
mkdir -p gpurun_out
( timeout 120 tools/ubench/xcorr_f16x2 3840; timeout 120 tools/ubench/xcorr_f16x2 12800 ) > gpurun_out/r06_ubench_xcorr_f16x2.jsonl 2>&1
cat gpurun_out/r06_ubench_xcorr_f16x2.jsonl
