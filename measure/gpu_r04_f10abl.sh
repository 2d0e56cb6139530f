mkdir -p gpurun_out
python measure/fused_ab2.py 30 100 -- SMOT_FUSED_GEN=10 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=1024 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=2048 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=4096 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=6144 SMOT_FUSED_GEN=10,SMOT_FUSED_ABL=8192 > gpurun_out/r04_f10_abl.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r04_f10_abl.jsonl | grep tracks | tail -12
