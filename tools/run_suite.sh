mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_train_fused_gpu.py tests/test_train_full_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | cut -c1-600 > gpurun_out/pytest_fe.log
echo "pytest rc ${PIPESTATUS[0]}"; tail -n 6 gpurun_out/pytest_fe.log
timeout -k 10 200 python tools/launcher_table.py conv0_two_pass 2>&1 | grep -v amdgpu > gpurun_out/launcher_table.log; grep "wgrad\|^sum" gpurun_out/launcher_table.log
