mkdir -p gpurun_out
python measure/debug/fused10_check.py > gpurun_out/r04_f10_check.jsonl 2>&1; grep -v amdgpu.ids gpurun_out/r04_f10_check.jsonl | tail -14
python measure/fused_ab2.py 30 100 -- hint=1 SMOT_FUSED_GEN=10 > gpurun_out/r04_f10_ab.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r04_f10_ab.jsonl | tail -14
python measure/debug/fused10_trace.py 30 100 > gpurun_out/r04_f10_trace.jsonl 2>&1; grep -v amdgpu.ids gpurun_out/r04_f10_trace.jsonl | tail -4
