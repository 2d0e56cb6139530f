mkdir -p gpurun_out
TRACKS=30 ABLS=12 timeout 600 python measure/debug/tower_bf3_check.py > gpurun_out/r04_tower_bf3.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r04_tower_bf3.jsonl | tail -20
