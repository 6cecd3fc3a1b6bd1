mkdir -p gpurun_out
timeout -k 10 300 python tools/launcher_calls.py 60 > gpurun_out/launcher_calls.log 2>&1; grep -E "conv0|sum of" gpurun_out/launcher_calls.log
