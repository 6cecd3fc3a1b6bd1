mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_train_fused_gpu.py -x -q -m gpu -k "rebuilds" 2>&1 | tail -n 2
timeout -k 10 300 python tools/launcher_calls.py 100 > gpurun_out/launcher_calls.log 2>&1; grep -E "conv0|sum of" gpurun_out/launcher_calls.log
timeout -k 10 300 python tools/ab_flags.py conv0_in_dgrad 3 > gpurun_out/ab_flags.log 2>&1; tail -n 6 gpurun_out/ab_flags.log
