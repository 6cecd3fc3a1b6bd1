mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_train_fused_gpu.py tests/test_train_head_gpu.py tests/test_train_full_gpu.py tests/test_train_workflow_gpu.py -x -q -m gpu 2>&1 | tail -n 4
timeout -k 10 300 python tools/launcher_calls.py 100 > gpurun_out/launcher_calls.log 2>&1; grep -E "lstm|sum of" gpurun_out/launcher_calls.log
