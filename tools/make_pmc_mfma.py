"""profiles/r01_pmc_mfma.json from the rocprofv3 --pmc pass of tools/evidence_b.sh (SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES,
SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE on `bench.py --steps 1 --warmup 1`).

usage: python tools/make_pmc_mfma.py gpurun_out/ev/pmc_mfma profiles/r01_pmc_mfma.json
MFMA utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) with cycles = GRBM_GUI_ACTIVE / 8: rocprofv3 sums
the GRBM counter over the 8 XCDs (it reads 8 x 2.4 GHz x the kernel duration).  Cross-check: SQ_INSTS_MFMA x 32 cycles
(v_mfma_f32_16x16x4_f32: 8 passes, MI355X_MICROARCH.md) over the same denominator."""
import json
import subprocess
import sys
from pathlib import Path

here = Path(__file__).resolve().parent
SIMDS = 256 * 4
XCDS = 8


def main():
    src, dst = sys.argv[1], Path(sys.argv[2])
    d = json.loads(subprocess.run([sys.executable, str(here / "summarize_pmc.py"), src], check=True, capture_output=True, text=True).stdout)
    out = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "kernels": {}}
    for sym, busy in d.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).items():
        if sym.startswith(("at::", "__amd")) or busy["mean_per_launch"] == 0:
            continue
        act = d["GRBM_GUI_ACTIVE"][sym]["mean_per_launch"]
        insts = d["SQ_INSTS_MFMA"][sym]["mean_per_launch"]
        out["kernels"][sym] = {"launches": busy["launches"], "mfma_busy_cycles": round(busy["mean_per_launch"]), "gui_active_cycles": round(act),
                               "mfma_wave_instructions": round(insts), "mfma_util_counter": round(busy["mean_per_launch"] / (act / XCDS * SIMDS), 4),
                               "mfma_util_from_instruction_count": round(insts * 32 / (act / XCDS * SIMDS), 4)}
    dst.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print(dst)


if __name__ == "__main__":
    main()
