"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin), filtered by a
substring of the mangled name.  usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 | python tools/kres.py ftile"""
import re, subprocess, sys

pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur, rows = None, []
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|TotalSGPRs): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0] + ("Spill" if "Spill" in k else "")] = v
for r in rows:
    if pat in r["name"]:
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r["name"]], capture_output=True, text=True).stdout.strip()
        except Exception:
            name = r["name"]
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"{name:58s} vgpr {r.get('VGPRs','?'):>4} scratch {r.get('ScratchSize','?'):>4} occ {r.get('Occupancy','?'):>2} lds {r.get('LDS','?'):>6} spill {r.get('VGPRsSpill','?')}")
