# round 6, session a: matrix-pipe residuals — exactness + rate ubench, tower A/B (ABL 13 = vector split of rounds 4-5), tower tests
mkdir -p gpurun_out
timeout 120 tools/ubench/mfma_residual > gpurun_out/r06a_ubench_mfma_residual.jsonl 2>&1
cat gpurun_out/r06a_ubench_mfma_residual.jsonl
TRACKS=30,100 ABLS=13 timeout 600 python measure/debug/tower_bf3_check.py > gpurun_out/r06a_tower.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r06a_tower.jsonl | cut -c1-420 | tail -14
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short -x -k "tower or predictor or forms" > gpurun_out/r06a_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r06a_pytest.log
grep -n "passed\|failed\|^E  " gpurun_out/r06a_pytest.log | cut -c1-400 | tail -12
