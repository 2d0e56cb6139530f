mkdir -p gpurun_out
timeout 300 python measure/debug/fused_trace_ab.py 30 -- SMOT_FUSED_ABL=0 > gpurun_out/r06mm_trace_ab2.jsonl 2>&1
cut -c1-2500 gpurun_out/r06mm_trace_ab2.jsonl
