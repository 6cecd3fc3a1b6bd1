#!/usr/bin/env python3
"""Convert a reference-trained orcAI model to this package's weight store.

    python tools/keras_to_npz.py MODEL_DIR            # reads MODEL_DIR/<name>.keras or MODEL_DIR/model_weights.h5
                                                      # writes MODEL_DIR/<name>.weights.npz

Needs ``h5py`` (NOT present in the build image -- run it where the reference's environment exists; keras itself is not needed:
the ``.keras`` archive is a zip whose ``model.weights.h5`` member is read directly).  What is read, and how variable paths map
to npz names, is defined in ``orcai_amd/keras_layout.py`` (reference ``io.py:386-404``, ``architectures.py:162-241``).  After the
rename every array is checked against ``ResNetLSTM.variable_spec()`` (name set and shapes) before anything is written; optimizer
state in the archive is ignored (``load_orcai_model`` only needs the forward weights; training resumes with fresh Adam moments).
"""

from __future__ import annotations

import io
import json
import sys
import zipfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from orcai_amd import keras_layout  # noqa: E402


def _read_h5_datasets(h5file) -> dict:
    out = {}

    def visit(name, obj):
        if hasattr(obj, "shape") and hasattr(obj, "dtype"):
            out[name] = np.asarray(obj)

    h5file.visititems(visit)
    return out


def load_keras_arrays(path: Path, n_blocks: int, architecture: str) -> dict:
    try:
        import h5py
    except ImportError as e:  # the build image: say what is missing instead of failing obscurely
        raise SystemExit(f"tools/keras_to_npz.py needs h5py to read {path.name}: {e}") from e
    if path.suffix == ".keras":
        with zipfile.ZipFile(path) as z:
            blob = z.read("model.weights.h5")
        with h5py.File(io.BytesIO(blob), "r") as f:
            return keras_layout.from_keras3_paths(_read_h5_datasets(f), n_blocks, architecture)
    with h5py.File(path, "r") as f:
        if "layers" in f:  # a Keras 3 ``*.weights.h5``
            return keras_layout.from_keras3_paths(_read_h5_datasets(f), n_blocks, architecture)
        dec = lambda v: v.decode() if isinstance(v, bytes) else str(v)  # noqa: E731
        layer_names = [dec(n) for n in f.attrs["layer_names"]]
        weight_names = {ln: [dec(w) for w in f[ln].attrs["weight_names"]] for ln in layer_names}
        arrays = {f"{ln}/{w}": np.asarray(f[ln][w]) for ln in layer_names for w in weight_names[ln]}
        return keras_layout.from_legacy_h5(arrays, layer_names, weight_names, n_blocks, architecture)


def convert(model_dir: Path) -> Path:
    from orcai_amd.architectures import build_model
    from orcai_amd.auxiliary import Messenger
    from orcai_amd.io import WEIGHTS_SUFFIX, read_json

    model_dir = Path(model_dir)
    param = read_json(model_dir / "orcai_parameter.json")
    shape = read_json(model_dir / "model_shape.json")
    name = param["name"]
    src = model_dir / f"{name}.keras"
    if not src.exists():
        src = model_dir / "model_weights.h5"
    if not src.exists():
        raise SystemExit(f"neither {name}.keras nor model_weights.h5 in {model_dir}")
    weights = load_keras_arrays(src, len(param["model"]["filters"]), param["architecture"])
    model = build_model(tuple(shape["input_shape"]), param, msgr=Messenger(verbosity=0))
    model.set_weights_dict(weights)  # raises on a missing name or a shape mismatch
    out = model_dir / (name + WEIGHTS_SUFFIX)
    model.save_weights(out)
    print(json.dumps({"source": src.name, "written": out.name, "variables": len(weights), "parameters": model.count_params()}))
    return out


if __name__ == "__main__":
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    convert(Path(sys.argv[1]))
