mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sequence.py tests/test_solver.py -m gpu -q -x 2>&1 | tail -3
python measure/loop_hint_ab2.py 30 100 > gpurun_out/r04_loop_hint_ab.jsonl 2>&1; grep -v amdgpu.ids gpurun_out/r04_loop_hint_ab.jsonl | tail -24
