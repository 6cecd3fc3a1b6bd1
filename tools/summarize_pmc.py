"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel symbol, launches and mean counter value per launch.

usage: python tools/summarize_pmc.py <dir-with-*_counter_collection.csv> [...]   -> JSON on stdout
Counter values are summed over the dimensions rocprofv3 reports per dispatch (one row per dispatch x instance).
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^>(]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    out = {}
    for d in sys.argv[1:]:
        for path in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            per_dispatch = defaultdict(float)
            kern = {}
            counter = None
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    key = row["Dispatch_Id"]
                    per_dispatch[key] += float(row["Counter_Value"])
                    kern[key] = short(row["Kernel_Name"])
                    counter = row["Counter_Name"]
            agg = defaultdict(list)
            for k, v in per_dispatch.items():
                agg[kern[k]].append(v)
            out.setdefault(counter, {}).update({k: {"launches": len(v), "mean_per_launch": sum(v) / len(v), "total": sum(v)} for k, v in agg.items()})
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
