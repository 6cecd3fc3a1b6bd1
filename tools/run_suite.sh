mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "stacked" 2>&1 | tail -n 5
