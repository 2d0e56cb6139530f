mkdir -p gpurun_out
timeout 300 python measure/debug/fused_trace_ab.py 30 100 -- SMOT_FUSED_ABL=0 SMOT_FUSED_ABL=8 SMOT_FUSED_ABL=9 > gpurun_out/r06mm_trace_ab.jsonl 2>&1
cat gpurun_out/r06mm_trace_ab.jsonl
