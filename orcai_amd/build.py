"""Builds liborcai_hip.so (all HIP kernels + the C ABI) in-tree with hipcc for gfx950.

The built library is git-ignored but travels to the GPU box with the source snapshot.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
LIB_PATH = PKG_DIR / "liborcai_hip.so"
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: liborcai_hip.so cannot be built")


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def is_stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = sources() + sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    """Compile every csrc/*.hip into one shared library.  Objects are cached per source."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    objs = []
    hdr_mtime = max([p.stat().st_mtime for p in list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))] or [0.0])
    for src in sources():
        obj = objdir / (src.stem + ".o")
        objs.append(obj)
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_mtime):
            continue
        cmd = [hipcc, "-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", f"-I{INCLUDE}", f"-I{CSRC}", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH)] + [str(o) for o in objs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
