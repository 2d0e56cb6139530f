# round-5 session F: the loop soak (default path == general path, bit for bit, on 1,500 frames of random traffic with the
# yaml's 30-frame dormancy; no false alarm of the hint's verification), the loop A/B and the bench lines at HEAD (the loop
# legs take ready-made detection BoxLists and report the median of three chunks).
#   gpurun --timeout 1500 -- 'bash measure/gpu_r05_f.sh'
TAG=r05
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python measure/debug/loop_soak_r05.py 1500 > gpurun_out/${TAG}_loop_soak.log 2>&1; echo "soak exit $?" >> gpurun_out/${TAG}_loop_soak.log; tail -9 gpurun_out/${TAG}_loop_soak.log | cut -c1-600
timeout 300 python measure/loop_early_ab.py 30 > gpurun_out/${TAG}_loop_early_ab.jsonl 2>&1; grep '^{' gpurun_out/${TAG}_loop_early_ab.jsonl | cut -c1-200
timeout 700 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json; cut -c1-300 gpurun_out/${TAG}_bench_line.json
bash measure/gpu_r04_bench.sh ${TAG} | cut -c1-400
