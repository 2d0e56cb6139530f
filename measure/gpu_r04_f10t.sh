mkdir -p gpurun_out
python measure/debug/fused10_trace.py 30 100 > gpurun_out/r04_f10_trace.jsonl 2>&1; grep -v amdgpu.ids gpurun_out/r04_f10_trace.jsonl | tail -4
