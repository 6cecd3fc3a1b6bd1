mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_half_gpu.py tests/test_train_full_gpu.py tests/test_train_fused_gpu.py -q -m gpu -s 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > gpurun_out/pytest_fe.log
echo "pytest rc ${PIPESTATUS[0]}"; grep -E "f16 training step|passed|failed|Error|assert" gpurun_out/pytest_fe.log | tail -n 14
