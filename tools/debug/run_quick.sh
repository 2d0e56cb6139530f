# scratch: one GPU-box session for the kernel under work (edit freely; tools/gpu_round.sh is the full round)
set -x
python -m pytest tests/test_solver.py tests/test_video_results.py tests/test_box_refine.py -m gpu -q --no-header --tb=short -x -p no:cacheprovider 2>&1 | tail -5
python tools/debug/lean_split.py 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['tracking_loop'])"
python tools/debug/solver_trace.py 2>&1 | grep "^{" | head -2 | cut -c1-300
