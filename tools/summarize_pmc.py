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


_DEMANGLED = {}


def demangle(name: str) -> str:
    """rocprofv3 leaves symbols with _Float16 parameters mangled: run them through llvm-cxxfilt when it is there."""
    if not name.startswith("_Z"):
        return name
    if name not in _DEMANGLED:
        import shutil
        import subprocess

        tool = shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
        try:
            _DEMANGLED[name] = subprocess.run([tool, name], capture_output=True, text=True, check=True).stdout.strip() or name
        except (OSError, subprocess.CalledProcessError):
            _DEMANGLED[name] = name
    return _DEMANGLED[name]


def parse_itanium_kernel(name: str):
    """`_ZN12_GLOBAL__N_1<len><kernel>I<literal template arguments>E...` -> `kernel<2, false, true>`: the demanglers of this image give up on
    symbols with _Float16 parameters (DF16_), which is every kernel of the f16 path.  Only integer / bool literal arguments are understood."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if not m:
        return None
    n, i = int(m.group(1)), m.end()
    kernel, rest = name[i : i + n], name[i + n :]
    if not rest.startswith("I"):
        return kernel
    args, j = [], 1
    while j < len(rest) and rest[j] != "E":
        a = re.match(r"L([ibjlm])(n?)(\d+)E", rest[j:])
        if not a:
            return kernel
        v = int(a.group(3)) * (-1 if a.group(2) else 1)
        args.append(("true" if v else "false") if a.group(1) == "b" else str(v))
        j += a.end()
    return f"{kernel}<{', '.join(args)}>"


def short(name: str) -> str:
    if name.startswith("_Z"):
        p = parse_itanium_kernel(name)
        if p and demangle(name) == name:  # the demangler could not read it
            return p
    name = demangle(name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^>(]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    out = {}
    for d in sys.argv[1:]:
        for path in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            per_dispatch = defaultdict(float)  # (counter, dispatch) -> value summed over the reported instances
            kern = {}
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    key = (row["Counter_Name"], row["Dispatch_Id"])
                    per_dispatch[key] += float(row["Counter_Value"])
                    kern[row["Dispatch_Id"]] = short(row["Kernel_Name"])
            agg = defaultdict(lambda: defaultdict(list))
            for (counter, disp), v in per_dispatch.items():
                agg[counter][kern[disp]].append(v)
            for counter, per_kernel in agg.items():
                out.setdefault(counter, {}).update({k: {"launches": len(v), "mean_per_launch": sum(v) / len(v), "total": sum(v)} for k, v in per_kernel.items()})
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
