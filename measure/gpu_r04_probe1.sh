# round 4, probe 1: where the fused kernel's time goes before the redesign (no-correlation ablation, traces)
mkdir -p gpurun_out
python measure/fused_ab.py 30 100 -- SMOT_FUSED_ABL=0 SMOT_FUSED_ABL=2 SMOT_FUSED_ORDER=4 > gpurun_out/r04p1_fused_ab.jsonl 2>&1
cat gpurun_out/r04p1_fused_ab.jsonl | tail -14
python measure/debug/fused_trace.py 30 > gpurun_out/r04p1_fused_trace.jsonl 2>&1
cat gpurun_out/r04p1_fused_trace.jsonl | tail -5
