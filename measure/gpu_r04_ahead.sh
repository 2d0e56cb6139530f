mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sequence.py -m gpu -q -k "ahead or order_hint_to_the_next_head" 2>&1 | tail -5
timeout 600 python measure/loop_ahead_ab.py 30 100 > gpurun_out/r04_loop_ahead_ab.jsonl 2>&1; grep tracks gpurun_out/r04_loop_ahead_ab.jsonl | tail -30
