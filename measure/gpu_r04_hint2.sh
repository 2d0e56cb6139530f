mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_sequence.py -m gpu -q -x -k "hint or fused or order or sequence or closed or benchmark" 2>&1 | tail -4
python measure/fused_ab2.py 30 100 -- hint=1 SMOT_FUSED_GEN=0 > gpurun_out/r04_hint2_ab.jsonl 2>&1; grep -v amdgpu.ids gpurun_out/r04_hint2_ab.jsonl | grep tracks | tail -8
python measure/debug/fused_trace.py 30 2>&1 | grep -v amdgpu.ids | grep "fused<30" 
