# generation 4 (measure/csrc/sr_xcorr_plan.hip): phase decomposition by ablation (abl bits: 1 no window loads, 2 no template
# loads, 4 no pooling arithmetic, 8 no correlation) and the staggered-start experiment (abl >> 8 = delay of waves 4-7 in 512-cycle units)
mkdir -p gpurun_out
python measure/fused_ab2.py 30 -- hint=1 SMOT_FUSED_GEN=10 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=1 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=3 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=4 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=8 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=12 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=15 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=7 > gpurun_out/r04_gen4_decomposition.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r04_gen4_decomposition.jsonl | grep tracks | tail -9
python measure/fused_ab2.py 30 100 -- SMOT_FUSED_GEN=10 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=1024 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=2048 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=4096 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=8192 > gpurun_out/r04_gen4_stagger.jsonl 2>&1
python measure/debug/fused10_trace.py 30 100 > gpurun_out/r04_gen4_trace.jsonl 2>&1
