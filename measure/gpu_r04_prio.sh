mkdir -p gpurun_out
python measure/fused_ab2.py 30 100 -- hint=1 hint=1,SMOT_FUSED_ABL=21 hint=1,SMOT_FUSED_ABL=22 hint=1,SMOT_FUSED_ABL=23 > gpurun_out/r04_prio_ab.jsonl 2>&1; grep -v amdgpu.ids gpurun_out/r04_prio_ab.jsonl | grep tracks | tail -16
