# round 4, probe 2: the fused kernel by its own dispatch timestamps: with / without correlation, hinted / un-hinted
mkdir -p gpurun_out
python measure/fused_ab2.py 30 100 -- SMOT_FUSED_ABL=0 SMOT_FUSED_ABL=2 hint=1 hint=1,SMOT_FUSED_ABL=2 SMOT_FUSED_ORDER=4 > gpurun_out/r04p2_fused_ab2.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r04p2_fused_ab2.jsonl | tail -24
