#!/usr/bin/env python3
"""Where a step's wall time is NOT inside a kernel: from a rocprofv3 --kernel-trace CSV, the busy fraction of the device over the steady-state window
and the idle gaps grouped by the kernel that FOLLOWS the gap (the launch that arrived late or ramped slowly).

usage: python tools/gap_analysis.py <dir with *_kernel_trace.csv> [--last-frac 0.5] [--top 25]
The window is the last `last-frac` of the dispatches (warm-up, construction and capture excluded).  Kernels of concurrent streams overlap: busy time is the
union of the intervals, `sum` their plain sum."""
import csv
import re
import sys
from collections import defaultdict
from pathlib import Path


def short(name: str) -> str:
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:
        n = int(m.group(1))
        rest = name[len(m.group(0)):]
        return rest[:n] + " " + rest[n:n + 24]
    return name.split("(")[0][:80]


def main():
    d = Path(sys.argv[1])
    frac = float(sys.argv[sys.argv.index("--last-frac") + 1]) if "--last-frac" in sys.argv else 0.5
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    f = next(d.rglob("*kernel_trace.csv"))
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
    rows.sort()
    rows = rows[int(len(rows) * (1 - frac)):]
    t0, t1 = rows[0][0], max(e for _, e, _ in rows)
    busy, cur_end, gaps, gap_n, ksum, kn = 0, rows[0][0], defaultdict(int), defaultdict(int), defaultdict(int), defaultdict(int)
    for s, e, n in rows:
        ksum[n] += e - s
        kn[n] += 1
        if s > cur_end:
            gaps[n] += s - cur_end
            gap_n[n] += 1
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    wall = t1 - t0
    print(f"{f.name}: {len(rows)} dispatches over {wall / 1e6:.2f} ms; device busy {busy / 1e6:.2f} ms = {busy / wall:.3f}; sum of durations {sum(ksum.values()) / 1e6:.2f} ms; idle {(wall - busy) / 1e6:.2f} ms")
    big = [g for g in gaps.values()]
    print(f"gaps: {sum(gap_n.values())} totalling {sum(big) / 1e6:.2f} ms; mean {sum(big) / max(1, sum(gap_n.values())) / 1e3:.2f} us")
    print(f"{'kernel':70s} {'n':>6s} {'total ms':>9s} {'mean us':>8s}")
    for n, t in sorted(ksum.items(), key=lambda kv: -kv[1])[:top]:
        print(f"{n[:70]:70s} {kn[n]:6d} {t / 1e6:9.3f} {t / kn[n] / 1e3:8.1f}")
    print(f"{'gap BEFORE kernel':70s} {'n':>6s} {'gap ms':>8s} {'mean us':>8s} {'kernel mean us':>14s}")
    for n, g in sorted(gaps.items(), key=lambda kv: -kv[1])[:top]:
        print(f"{n[:70]:70s} {gap_n[n]:6d} {g / 1e6:8.3f} {g / gap_n[n] / 1e3:8.2f} {ksum[n] / kn[n] / 1e3:14.1f}")


if __name__ == "__main__":
    main()
