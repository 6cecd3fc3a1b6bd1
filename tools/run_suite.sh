mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_frontend_gpu.py tests/test_fullsize_gpu.py tests/test_predict_e2e_gpu.py tests/test_train_workflow_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | cut -c1-600 > gpurun_out/pytest_fe.log
echo "pytest rc ${PIPESTATUS[0]}"; tail -n 15 gpurun_out/pytest_fe.log
for i in 1 2; do timeout -k 10 120 python bench.py --workload frontend --no-cpu-baseline > gpurun_out/bench_fe_$i.json 2> gpurun_out/bench_fe_$i.err; cut -c1-700 gpurun_out/bench_fe_$i.json; done
