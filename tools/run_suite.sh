# GPU-box script: the whole GPU suite, then the default bench; logs under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids | cut -c1-600 > gpurun_out/pytest_full.log
echo "pytest rc ${PIPESTATUS[0]}"; tail -n 8 gpurun_out/pytest_full.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc $?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print("predict", d["value"], d["ms_per_step"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
print("per layer", d["roofline"]["per_layer_ms_per_step"])
s = d["secondary"]; print("train", s["value"], s["ms_per_step"], s["roofline"].get("launcher"), s["roofline"].get("frac"))
print("sweep", d["secondary2"]["value"], d["secondary2"]["ms_per_step"])
print("cpu", d["cpu_baseline"])
PY
