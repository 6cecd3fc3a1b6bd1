"""Per-step summary of a rocprofv3 *kernel_stats.csv: usage kstats.py FILE STEPS [TOP]"""
import csv
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from summarize_pmc import demangle  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 28
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time {tot / 1e6 / steps:.3f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms {int(r['Calls']) / steps:6.1f} calls {float(r['AverageNs']) / 1e3:8.1f} us  {demangle(r['Name'])[:100]}")
