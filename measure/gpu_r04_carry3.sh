# round 4: the carried rows' small fields loaded at the solver's start — tests + the dormant loops
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_sequence.py tests/test_solver.py -m gpu -q --no-header --tb=short -x 2>&1 | tail -3
timeout 60 python measure/debug/loop_dormant.py 30 6 d,a > gpurun_out/r04_loop_dormant_c.jsonl 2>&1; grep '^{' gpurun_out/r04_loop_dormant_c.jsonl | cut -c1-330
