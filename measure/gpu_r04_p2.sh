mkdir -p gpurun_out
python measure/fused_ab2.py 30 100 -- hint=1 hint=1,SMOT_FUSED_ABL=5 SMOT_FUSED_ABL=0 SMOT_FUSED_ABL=5 > gpurun_out/r04_p2_ab.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r04_p2_ab.jsonl | grep tracks | tail -16
