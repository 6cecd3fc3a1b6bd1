mkdir -p gpurun_out
for i in 1 2; do timeout -k 10 300 python bench.py --workload train --no-cpu-baseline > gpurun_out/bench_train_$i.json 2> gpurun_out/bench_train_$i.err; python - <<PY
import json
d = json.load(open("gpurun_out/bench_train_$i.json")); r = d["roofline"]
print("train", d["value"], d["ms_per_step"], r.get("launcher"), r.get("frac"), r.get("kernel_ms"), r.get("kernel_in_timed_steps"))
PY
done
