mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_half_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | cut -c1-900 > gpurun_out/pytest_fe.log
echo "pytest rc ${PIPESTATUS[0]}"; tail -n 12 gpurun_out/pytest_fe.log
timeout -k 10 300 python tools/ab_sweep_flags.py fused_dw_bwd > gpurun_out/ab_flags.log 2>&1; tail -n 6 gpurun_out/ab_flags.log
