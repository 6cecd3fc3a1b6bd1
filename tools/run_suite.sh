mkdir -p gpurun_out
timeout -k 10 300 python tools/launcher_calls.py 100 > gpurun_out/launcher_calls.log 2>&1; grep -E "dw_bwd|sum of" gpurun_out/launcher_calls.log
timeout -k 10 600 python -m pytest tests/test_train_fused_gpu.py tests/test_half_gpu.py -q -m gpu -k "marching or one_marching" 2>&1 | tail -n 3
