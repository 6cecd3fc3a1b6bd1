# GPU-box script: the whole GPU suite, then the bench rehearsals; logs under gpurun_out/ (progress lines keep the run visibly alive)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_train_fused_gpu.py tests/test_train_full_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | cut -c1-600 > gpurun_out/pytest_fused.log
echo "pytest fused rc ${PIPESTATUS[0]}"; tail -n 25 gpurun_out/pytest_fused.log
timeout -k 10 300 python tools/ab_flags.py fused_pw_wgrad 3 > gpurun_out/ab_flags.log 2>&1; tail -n 8 gpurun_out/ab_flags.log
