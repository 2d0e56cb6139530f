mkdir -p gpurun_out
ABL=0 timeout 300 python measure/debug/tower_bf3_diag.py > gpurun_out/r04_tower_bf3_diag.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r04_tower_bf3_diag.jsonl | tail
