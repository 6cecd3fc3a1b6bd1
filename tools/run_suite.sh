mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > gpurun_out/pytest_fe.log
echo "pytest rc ${PIPESTATUS[0]}"; tail -n 4 gpurun_out/pytest_fe.log
timeout -k 10 300 python tools/ab_pool.py > gpurun_out/ab_pool.log 2>&1; tail -n 7 gpurun_out/ab_pool.log
