"""profiles/rNN_pmc_traffic.json from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/evidence_rNN_b.sh.

usage: python tools/make_pmc_traffic.py gpurun_out/ev3 profiles/r03_pmc_traffic.json
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE tallies the
128-byte requests of wide (16 B per lane) reads at 64 bytes (MI355X_MICROARCH.md, HBM section), so it is doubled.
"""
import json
import subprocess
import sys
from pathlib import Path

here = Path(__file__).resolve().parent


def summarise(*dirs):
    out = subprocess.run([sys.executable, str(here / "summarize_pmc.py"), *map(str, dirs)], check=True, capture_output=True, text=True).stdout
    return json.loads(out)


def main():
    ev, dst = Path(sys.argv[1]), Path(sys.argv[2])
    result = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 1 --no-cpu-baseline` "
                      "(frontend: --steps 2); hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, mean over the launches of the symbol"}
    for workload, suffix in (("predict", ""), ("frontend", "_fe"), ("hpsearch_f16_set3", "_h"), ("train", "_t")):
        if not (ev / f"pmc_fetch{suffix}").exists():
            continue
        d = summarise(ev / f"pmc_fetch{suffix}", ev / f"pmc_write{suffix}")
        kernels = {}
        for sym, rec in d.get("FETCH_SIZE", {}).items():
            if sym.startswith("at::") or sym.startswith("__amd"):
                continue
            w = d.get("WRITE_SIZE", {}).get(sym)
            if w is None:
                continue
            kernels[sym] = {"launches": rec["launches"], "fetch_size_kib": round(rec["mean_per_launch"], 1), "write_size_kib": round(w["mean_per_launch"], 1),
                            "hbm_bytes_per_launch": round((2 * rec["mean_per_launch"] + w["mean_per_launch"]) * 1024)}
        result[workload] = {"kernels": kernels}
    dst.write_text(json.dumps(result, indent=1, sort_keys=True) + "\n")
    print(dst)


if __name__ == "__main__":
    main()
