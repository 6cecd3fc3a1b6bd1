mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_half_gpu.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > gpurun_out/pytest_fe.log
echo "pytest rc ${PIPESTATUS[0]}"; grep -E "f16 training step|passed|failed|Error|assert" gpurun_out/pytest_fe.log | tail -n 10
timeout -k 10 300 python tools/ab_sweep_flags.py conv0_in_dgrad > gpurun_out/ab_flags.log 2>&1; tail -n 6 gpurun_out/ab_flags.log
