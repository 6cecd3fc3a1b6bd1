mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > gpurun_out/pytest_full.log
echo "pytest rc ${PIPESTATUS[0]}"; tail -n 5 gpurun_out/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3
