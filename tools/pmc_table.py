"""Per-kernel reading of a tools/summarize_pmc.py JSON (tools/pmc_issue_r04.sh): per-wave instruction counts and the share of a wave's life spent
issuing / waiting.  usage: pmc_table.py file.json [substring ...]"""
import json, sys
d = json.load(open(sys.argv[1]))
subs = sys.argv[2:]
waves = d["SQ_WAVES"]
rows = []
for k in waves:
    if subs and not any(s in k for s in subs):
        continue
    g = lambda c: d.get(c, {}).get(k, {}).get("mean_per_launch", float("nan"))
    w, wc = g("SQ_WAVES"), g("SQ_WAVE_CYCLES")
    rows.append((g("SQ_BUSY_CYCLES") * waves[k]["launches"], k, waves[k]["launches"], w, wc, g))
for _, k, n, w, wc, g in sorted(rows, key=lambda r: -r[0])[: (len(rows) if subs else 14)]:
    gui = g("GRBM_GUI_ACTIVE")
    print(f"{k}  x{n}")
    print(f"   waves {w:.0f}, cycles per wave {wc / w:.0f}, GPU-active cycles per launch {gui:.3g}, resident waves per SIMD {wc / gui / 1024 * 1:.2f}" if gui == gui else f"   waves {w:.0f}")
    print("   per wave: VALU %.0f  MFMA %.0f  SALU %.0f  LDS %.0f  SMEM %.0f  VMEM rd %.0f wr %.0f" % tuple(g(c) / w for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")))
    print("   share of wave cycles: waiting (any instruction) %.2f  VALU issuing %.3f  LDS issuing %.3f  VMEM issuing %.3f  scalar %.3f | LDS wait %.3f | MFMA busy / SIMD-active %.3f, VALU active / SIMD-active %.3f, LDS bank conflict cycles / LDS active %.3f" % (
        g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc, g("SQ_ACTIVE_INST_SCA") / wc, g("SQ_WAIT_INST_LDS") / wc,
        g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CU_CYCLES") * 4 if g("SQ_BUSY_CU_CYCLES") == g("SQ_BUSY_CU_CYCLES") else float("nan")), g("SQ_ACTIVE_INST_VALU") / (g("SQ_BUSY_CU_CYCLES") * 4), g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_ACTIVE_INST_LDS"), 1)))
