mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_train_fused_gpu.py -x -q -m gpu -k "rebuilds or marching" 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > gpurun_out/pytest_fe.log
echo "pytest rc ${PIPESTATUS[0]}"; tail -n 12 gpurun_out/pytest_fe.log
