# round-4 check after a kernel change: GPU tests, arg-max statistics at 30 / 100 tracks, the driver's form of the bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu.log 2>&1; tail -3 gpurun_out/r04_pytest_gpu.log
timeout 600 python tools/argmax_stats.py --pairs 1000 --tracks 30 --out gpurun_out/r04_argmax_stats > gpurun_out/r04_argmax.log 2>&1; tail -3 gpurun_out/r04_argmax.log
timeout 600 python tools/argmax_stats.py --pairs 100 --tracks 100 --out gpurun_out/r04_argmax_stats_n100 > gpurun_out/r04_argmax_n100.log 2>&1; tail -3 gpurun_out/r04_argmax_n100.log
bash measure/gpu_r04_bench.sh r04
