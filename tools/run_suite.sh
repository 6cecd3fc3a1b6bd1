mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/exp_base.json 2> gpurun_out/exp_base.err && \
cp orcai_amd/liborcai_hip.so /tmp/keep.so && cp orcai_amd/liborcai_hip_exp.so orcai_amd/liborcai_hip.so && touch orcai_amd/liborcai_hip.so && \
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/exp_split.json 2> gpurun_out/exp_split.err ; cp /tmp/keep.so orcai_amd/liborcai_hip.so
python - <<'PY'
import json
for n in ("exp_base", "exp_split"):
    try:
        d = json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, d["value"], d["ms_per_step"], r["kernel_ms"], r["per_layer_ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
