mkdir -p gpurun_out
timeout 300 python measure/fused_ab.py 30 100 -- SMOT_FUSED_ABL=0 SMOT_FUSED_ABL=8 > gpurun_out/r06mm_ab.jsonl 2>&1
cat gpurun_out/r06mm_ab.jsonl | tail -8
timeout 300 python measure/debug/fused_trace.py 30 100 > gpurun_out/r06mm_trace.jsonl 2>&1
grep "fused<30" gpurun_out/r06mm_trace.jsonl
