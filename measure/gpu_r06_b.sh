# round 6, session b: tower A/B only (ABL 13 = vector split of rounds 4-5)
mkdir -p gpurun_out
TAG=${1:-r06b}
TRACKS=${TRACKS:-30} ABLS=${ABLS:-13} timeout 600 python measure/debug/tower_bf3_check.py > gpurun_out/${TAG}_tower.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/${TAG}_tower.jsonl | cut -c1-420 | tail -14
