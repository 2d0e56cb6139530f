mkdir -p gpurun_out
SWEEP=1 TRACKS=4,8,12,16,20,24,28,30,32,36,40,48,56,64,80,100 OCTS=1,2 BF3=1 timeout 900 python measure/debug/tower_bench.py > gpurun_out/r04_tower_sweep.jsonl 2>&1
SWEEP=1 TRACKS=16,30,32,64,100 OCTS=2 BF3=0 timeout 600 python measure/debug/tower_bench.py >> gpurun_out/r04_tower_sweep.jsonl 2>&1
grep tracks gpurun_out/r04_tower_sweep.jsonl | tail -60
