mkdir -p gpurun_out
TAG=${1:-r06c}
timeout 600 python measure/debug/tower_forms_by_tracks.py > gpurun_out/${TAG}_forms.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/${TAG}_forms.jsonl | tail -16
timeout 1200 python -m pytest tests -m gpu -q --no-header -rf --tb=short -k "tower or predictor or forms or fused or frame_pair or benchmark" > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
grep -n "passed\|failed\|^E  \|^FAILED" gpurun_out/${TAG}_pytest.log | cut -c1-300 | tail -30
