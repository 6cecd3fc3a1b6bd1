"""The C ABI is one contract stated three times: include/orcai_hip.h (the declarations a reference maintainer binds), the ctypes
table in orcai_amd/_native.py, and the symbols liborcai_hip.so actually exports.  No compute calls: runs without a GPU."""

import ctypes
import re
from pathlib import Path

from orcai_amd import _native as N

ROOT = Path(__file__).resolve().parent.parent
C_TYPES = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
           "double": ctypes.c_double}


def header_prototypes() -> dict:
    """{name: (return ctype, [argument ctypes])} parsed from every *.h under include/ (comments stripped)."""
    out = {}
    for h in sorted((ROOT / "include").glob("*.h")):
        text = re.sub(r"/\*.*?\*/", " ", h.read_text(), flags=re.S)
        for ret, name, args in re.findall(r"\b(int|size_t|const char\s*\*)\s*(orcai_\w+)\s*\(([^)]*)\)\s*;", text):
            argtypes = []
            for a in [x.strip() for x in args.split(",")]:
                if a in ("void", ""):
                    continue
                if "*" in a or "[" in a:  # an array parameter is a pointer
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = re.sub(r"\bconst\b", "", a).split()[0]
                    argtypes.append(C_TYPES[base])
            rt = ctypes.c_char_p if "char" in ret else C_TYPES[ret]
            out[name] = (rt, argtypes)
    return out


def _same(a, b) -> bool:
    pointerish = (ctypes.c_void_p, ctypes.c_char_p)
    if a in pointerish or (isinstance(a, type) and issubclass(a, ctypes._Pointer)):
        return b in pointerish or (isinstance(b, type) and issubclass(b, ctypes._Pointer))
    return ctypes.sizeof(a) == ctypes.sizeof(b) and a._type_.lower() == b._type_.lower() or a is b


def test_header_and_ctypes_table_declare_the_same_functions():
    proto = header_prototypes()
    assert len(proto) >= 60
    assert set(proto) == set(N.exported_symbols()), sorted(set(proto) ^ set(N.exported_symbols()))
    for name, (ret, args) in proto.items():
        r2, a2 = N._SIGNATURES[name]
        assert _same(ret, r2), name
        assert len(args) == len(a2), (name, len(args), len(a2))
        for i, (x, y) in enumerate(zip(args, a2)):
            assert _same(x, y), (name, i, x, y)


def test_library_exports_every_declared_symbol():
    lib = N.lib()  # raises NativeLibraryError when the library or a symbol is missing (no CPU fallback)
    for name in header_prototypes():
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.orcai_version()


def test_bench_kernel_symbols_are_in_the_newest_pmc_traffic_table():
    """bench.py's `roofline.traffic` is looked up by kernel symbol in profiles/rNN_pmc_traffic.json.  The symbol bench_predict.kernel_symbol()
    builds for the bracketed layers must be one the newest table (made from rocprofv3 --pmc passes of the same command at the same
    kernels) knows -- a renamed template argument list would otherwise turn `traffic` into a silent None or, worse, a stale hit."""
    import json

    import bench_predict as bp

    f = bp.traffic_file()
    assert f is not None
    kernels = json.loads(f.read_text())["predict"]["kernels"]
    for label in ("conv0+b1/sep_a", "b1/sep_b+pool_res", "b2/sep_a", "b2/sep_b", "b2/pool_res", "lstm1/rec", "lstm1/gemm"):  # labels the inference step launches
        sym = bp.PredictWorkload.kernel_symbol(label)
        assert sym in kernels, (label, sym, f.name, sorted(k for k in kernels if "sepconv" in k or "conv0" in k))
    fe = json.loads(f.read_text())["frontend"]["kernels"]
    import bench

    assert bench.FrontendWorkload.kernel_symbol in fe, (bench.FrontendWorkload.kernel_symbol, sorted(fe))


def test_training_bench_symbols_are_in_the_newest_pmc_traffic_table():
    """The same for the two training workloads: every kernel symbol the newest committed bench lines rank (bench_predict's call -> symbol maps, written
    into `symbol_ms_per_*_instrumented`; launcher names stand for calls with more than one kernel and are skipped) is a key of the table's section the
    line's `traffic` is read from, and the top symbol's traffic is what the table holds."""
    import json
    from pathlib import Path

    import bench_predict as bp

    f = bp.traffic_file()
    table = json.loads(f.read_text())
    prof = Path(f).parent
    rnd = f.name.split("_")[0]
    for workload, key, section in (("train", "symbol_ms_per_step_instrumented", "train"), ("hpsearch", "symbol_ms_per_sweep_step_instrumented", "hpsearch_f16_set3")):
        line = json.loads((prof / f"{rnd}_bench_{workload}.json").read_text())
        roof, kernels = line["roofline"], table[section]["kernels"]
        ranked = [s for s in roof[key] if not s.startswith("orcai_")]
        assert len(ranked) >= 6 and all(s in kernels for s in ranked), (workload, [s for s in ranked if s not in kernels])
        assert roof["kernel"] == ranked[0]
        assert bp.measured_traffic_symbol(roof["kernel"], section) == kernels[roof["kernel"]]["hbm_bytes_per_launch"]
